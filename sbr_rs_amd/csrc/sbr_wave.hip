/* sbr_wave.hip — the recurrent pass of SMALL minibatches at d <= 32: one wavefront per sequence.
 *
 * The reference's own schedule is one optimiser step per subsequence (sequence_model.rs:111-169) at embedding_dim 32
 * (lib.rs:22-58): a step is then ONE chain of up to max_sequence_length dependent LSTM cells in each direction.  On the
 * MFMA tile kernels (sbr_kernels.hip) such a step costs a full 16-row tile per time step — 15 of the 16 rows are
 * padding — and two barriers; here the recurrence runs on the vector ALU of a single wave, with no barrier at all:
 *
 *   forward   z_t = bW + x_t Wx + h_{t-1} Wh is a k-ascending fma chain whose x part comes first, so
 *             P_t = bW + x_t Wx does not depend on the recurrence: all 256 threads of the workgroup compute P_t for every
 *             t into LDS up front (and copy the gathered rows to X for the dense-gradient GEMM); then wave 0 walks t:
 *             lane l continues the chains of gate columns l and l + 64 over h_{t-1} (read from LDS as broadcast
 *             16-byte pieces), the gate pre-activations cross lanes through LDS, every lane evaluates the cell of unit
 *             l mod d and the lane groups share the stores of G / C / H.
 *   backward  the gate values, cell states and dloss/dh rows of every step are staged into LDS by all 256 threads;
 *             wave 0 walks t downwards: cell backward in lane u = l mod d, dz to LDS and HBM, then lane c owns column c
 *             of dz W^T (row c of W in registers: 4d values) — columns < d are dX of the row, columns >= d the recurrent
 *             dh, which goes back through LDS.
 *
 * Arithmetic is the contract's, operation for operation (sbr_numerics.h; fma chains in k / j order from the bias / from
 * 0), so the results are bit-identical to the tile kernels' and the oracle's; tests/test_parity_gpu.py runs both forms.
 * A workgroup per sequence: up to SBR_WAVE_MAX_SEQ sequences per step take this path (launch_wave_* return false
 * otherwise and the caller launches the tile kernels). */
#include <cstdlib>

#include "sbr_kernels.h"
#include "sbr_numerics.h"

namespace sbr {
namespace {

#ifndef SBR_WAVE_MAX_SEQ
#define SBR_WAVE_MAX_SEQ 64
#endif
constexpr size_t WAVE_LDS_LIMIT = 150 * 1024;

__device__ __forceinline__ float4 ld4w(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4w(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// LDS traffic between the lanes of ONE wave: the hardware executes a wave's LDS instructions in order, so a write followed
// by another lane's read needs no barrier — only the compiler must keep the program order
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

template <int D, int NG>
__global__ __launch_bounds__(256) void lstm_fwd_wave_kernel(ModelView m, MbView mb, float* H, WorkView w, int Tm) {
    constexpr int NGD = NG * D;
    constexpr int SLOTS = (NGD + 63) / 64;  // gate columns per lane of the recurrence wave
    constexpr int GROUPS = 64 / D;          // lane groups; every group evaluates the d cells
    constexpr int Q = D / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int nsteps = mb.steps[b];
    float* Xs = lds;            // [Tm][D]   gathered input rows
    float* Ps = Xs + Tm * D;    // [Tm][NGD] bias + x part of the pre-activations
    float* Zs = Ps + Tm * NGD;  // [NGD]     pre-activations of the current step
    float* Hs = Zs + NGD;       // [D]       h_{t-1}
    int* Rs = reinterpret_cast<int*>(Hs + D);  // [Tm] packed row of step t
    for (int t = tid; t < nsteps; t += 256) Rs[t] = mb.off[t] + b;
    if (tid < D) Hs[tid] = 0.0f;  // h_{-1} = 0
    __syncthreads();
    for (int idx = tid; idx < nsteps * Q; idx += 256) {
        const int t = idx / Q, c4 = (idx % Q) * 4;
        const int r = Rs[t];
        const float4 v = ld4w(m.E + (size_t)mb.in_idx[r] * D + c4);
        st4w(&Xs[t * D + c4], v);
        st4w(w.X + (size_t)r * D + c4, v);
    }
    __syncthreads();
    {
        constexpr int NSTREAM = 256 / NGD;  // threads (stream, j): column j of steps stream, stream + NSTREAM, ...
        const int stream = tid / NGD, j = tid % NGD;
        if (stream < NSTREAM) {
            float wx[D];
#pragma unroll
            for (int k = 0; k < D; ++k) wx[k] = m.W[(size_t)k * NGD + j];
            const float bj = m.bW[j];
            for (int t = stream; t < nsteps; t += NSTREAM) {
                float acc = bj;
#pragma unroll
                for (int k = 0; k < D; k += 4) {
                    const float4 x = ld4w(&Xs[t * D + k]);
                    acc = sbr_fma(x.x, wx[k], acc);
                    acc = sbr_fma(x.y, wx[k + 1], acc);
                    acc = sbr_fma(x.z, wx[k + 2], acc);
                    acc = sbr_fma(x.w, wx[k + 3], acc);
                }
                Ps[t * NGD + j] = acc;
            }
        }
    }
    __syncthreads();
    if (tid >= 64) return;
    // ---- the recurrence: wave 0 alone, no barriers
    const int lane = tid;
    const int u = lane % D, grp = lane / D;
    float wh[SLOTS][D];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = s * 64 + lane < NGD ? s * 64 + lane : 0;
#pragma unroll
        for (int k = 0; k < D; ++k) wh[s][k] = m.W[(size_t)(D + k) * NGD + j];
    }
    float c_prev = 0.0f;
    for (int t = 0; t < nsteps; ++t) {
        float z[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) z[s] = Ps[t * NGD + (s * 64 + lane < NGD ? s * 64 + lane : 0)];
#pragma unroll
        for (int k = 0; k < D; k += 4) {
            const float4 hv = ld4w(&Hs[k]);
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                z[s] = sbr_fma(hv.x, wh[s][k], z[s]);
                z[s] = sbr_fma(hv.y, wh[s][k + 1], z[s]);
                z[s] = sbr_fma(hv.z, wh[s][k + 2], z[s]);
                z[s] = sbr_fma(hv.w, wh[s][k + 3], z[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (s * 64 + lane < NGD) Zs[s * 64 + lane] = z[s];
        wave_lds_fence();
        const float zi = NG == 4 ? Zs[u] : 0.0f;
        const float zf = Zs[(NG - 3) * D + u], zg = Zs[(NG - 2) * D + u], zo = Zs[(NG - 1) * D + u];
        float gate[4], cc, hh;
        sbr_lstm_cell_fwd(zi, zf, zg, zo, c_prev, NG == 3, &gate[0], &gate[1], &gate[2], &gate[3], &cc, &hh);
        c_prev = cc;
        const int r = __builtin_amdgcn_readfirstlane(Rs[t]);
        float* Grow = w.G + (size_t)r * 4 * D;
#pragma unroll
        for (int g0 = 0; g0 < 4; g0 += GROUPS) {  // the lane groups share the four gate rows
            const int g = g0 + grp;
            float v = gate[g0];
#pragma unroll
            for (int q = 1; q < GROUPS; ++q) v = grp == q ? gate[(g0 + q) & 3] : v;
            if (g < 4) Grow[g * D + u] = v;
        }
        if (grp == 0) {
            w.C[(size_t)r * D + u] = cc;
            H[(size_t)r * D + u] = hh;
            Hs[u] = hh;
        }
        wave_lds_fence();
    }
}

template <int D, int NG>
__global__ __launch_bounds__(256) void lstm_bwd_wave_kernel(ModelView m, MbView mb, BlockView blk, WorkView w, int Tm) {
    constexpr int NGD = NG * D;
    constexpr int GROUPS = 64 / D;
    constexpr int Q = D / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int nsteps = mb.steps[b];
    float* Gs = lds;               // [Tm][4D] gate values
    float* Cs = Gs + Tm * 4 * D;   // [Tm][D]  cell states
    float* DHs = Cs + Tm * D;      // [Tm][D]  dloss/dh
    float* Zs = DHs + Tm * D;      // [NGD]    dz of the current step
    float* DHr = Zs + NGD;         // [D]      recurrent dh from step t+1
    int* Rs = reinterpret_cast<int*>(DHr + D);
    for (int t = tid; t < nsteps; t += 256) Rs[t] = mb.off[t] + b;
    if (tid < D) DHr[tid] = 0.0f;
    __syncthreads();
    for (int idx = tid; idx < nsteps * Q; idx += 256) {
        const int t = idx / Q, c4 = (idx % Q) * 4;
        const size_t r = (size_t)Rs[t];
        const float g = blk.coef[r];
        const float4 en = ld4w(m.E + (size_t)blk.neg[r] * D + c4);
        const float4 ep = ld4w(m.E + (size_t)blk.out_idx[r] * D + c4);
        float4 dh;  // g*E[neg] - g*E[pos]: two rounded products, one subtraction (the oracle's order)
        dh.x = g * en.x - g * ep.x;
        dh.y = g * en.y - g * ep.y;
        dh.z = g * en.z - g * ep.z;
        dh.w = g * en.w - g * ep.w;
        st4w(&DHs[t * D + c4], dh);
        st4w(&Cs[t * D + c4], ld4w(w.C + r * D + c4));
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) st4w(&Gs[t * 4 * D + g4 * D + c4], ld4w(w.G + r * 4 * D + g4 * D + c4));
    }
    __syncthreads();
    if (tid >= 64) return;
    const int lane = tid;
    const int u = lane % D, grp = lane / D;
    const int col = lane < 2 * D ? lane : 0;  // column of dz W^T this lane owns (lanes >= 2d idle along)
    float wt[NGD];                            // W^T[j][col] = W[col][j]: row `col` of W
#pragma unroll
    for (int j = 0; j < NGD; j += 4) {
        const float4 v = ld4w(m.W + (size_t)col * NGD + j);
        wt[j] = v.x; wt[j + 1] = v.y; wt[j + 2] = v.z; wt[j + 3] = v.w;
    }
    float dc = 0.0f;
    for (int t = nsteps - 1; t >= 0; --t) {
        const bool carried = t + 1 < nsteps;  // the sequence has a step t+1: recurrent dh and dc exist
        const float recv = carried ? DHr[u] : 0.0f;
        const float dh = DHs[t * D + u] + recv;
        const float gi = Gs[t * 4 * D + u], gf = Gs[t * 4 * D + D + u], gg = Gs[t * 4 * D + 2 * D + u], go = Gs[t * 4 * D + 3 * D + u];
        const float cc = Cs[t * D + u];
        const float cp = t > 0 ? Cs[(t - 1) * D + u] : 0.0f;
        float dz[4], dco;
        sbr_lstm_cell_bwd(dh, carried ? dc : 0.0f, gi, gf, gg, go, cc, cp, NG == 3, &dz[0], &dz[1], &dz[2], &dz[3], &dco);
        dc = dco;
        const int r = __builtin_amdgcn_readfirstlane(Rs[t]);
        float* dZrow = w.dZ + (size_t)r * NGD;
#pragma unroll
        for (int g0 = 0; g0 < 4; g0 += GROUPS) {  // the lane groups share the gate rows (coupled: the input gate has none)
            const int g = g0 + grp;
            float v = dz[g0];
#pragma unroll
            for (int q = 1; q < GROUPS; ++q) v = grp == q ? dz[(g0 + q) & 3] : v;
            const int jrow = NG == 4 ? g : g - 1;
            if (g < 4 && jrow >= 0) {
                dZrow[jrow * D + u] = v;
                Zs[jrow * D + u] = v;
            }
        }
        wave_lds_fence();
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NGD; j += 4) {
            const float4 zv = ld4w(&Zs[j]);
            acc = sbr_fma(zv.x, wt[j], acc);
            acc = sbr_fma(zv.y, wt[j + 1], acc);
            acc = sbr_fma(zv.z, wt[j + 2], acc);
            acc = sbr_fma(zv.w, wt[j + 3], acc);
        }
        if (lane < D) blk.dX[(size_t)r * D + lane] = acc;
        else if (lane < 2 * D) DHr[lane - D] = acc;
        wave_lds_fence();
    }
}

int wave_mode() {  // SBR_WAVE: 0 never, 1 whenever the shape allows, unset: up to SBR_WAVE_MAX_SEQ sequences per step
    const char* e = std::getenv("SBR_WAVE");  // read per call: the tests force either form
    return e ? (std::atoi(e) ? 1 : 0) : -1;
}
bool wave_shape_ok(int d, int ng, int B, size_t lds_bytes) {
    if ((d != 16 && d != 32) || (ng != 3 && ng != 4) || B <= 0 || lds_bytes > WAVE_LDS_LIMIT) return false;
    const int mode = wave_mode();
    return mode == 1 || (mode < 0 && B <= SBR_WAVE_MAX_SEQ);
}
template <class K>
void allow_lds(K kernel, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

bool launch_wave_forward(const ModelView& m, const MbView& mb, float* H, const WorkView& w, int tm_host, hipStream_t s) {
    const int d = m.d, ngd = m.ng * m.d;
    const size_t lds = ((size_t)tm_host * (d + ngd) + ngd + d + tm_host) * 4;
    if (!wave_shape_ok(d, m.ng, mb.B, lds)) return false;
#define SBR_WAVE_FWD(DD, NN)                                                                                   \
    {                                                                                                          \
        allow_lds(lstm_fwd_wave_kernel<DD, NN>, lds);                                                          \
        hipLaunchKernelGGL((lstm_fwd_wave_kernel<DD, NN>), dim3(mb.B), dim3(256), lds, s, m, mb, H, w, tm_host);       \
    }
    if (d == 32 && m.ng == 4) SBR_WAVE_FWD(32, 4)
    else if (d == 32) SBR_WAVE_FWD(32, 3)
    else if (m.ng == 4) SBR_WAVE_FWD(16, 4)
    else SBR_WAVE_FWD(16, 3)
#undef SBR_WAVE_FWD
    return true;
}

bool launch_wave_backward(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, int tm_host, int b_host,
                          hipStream_t s) {
    const int d = m.d, ngd = m.ng * m.d;
    const size_t lds = ((size_t)tm_host * 6 * d + ngd + d + tm_host) * 4;
    if (!wave_shape_ok(d, m.ng, b_host, lds)) return false;
#define SBR_WAVE_BWD(DD, NN)                                                                                   \
    {                                                                                                          \
        allow_lds(lstm_bwd_wave_kernel<DD, NN>, lds);                                                          \
        hipLaunchKernelGGL((lstm_bwd_wave_kernel<DD, NN>), dim3(b_host), dim3(256), lds, s, m, mb, blk, w, tm_host);   \
    }
    if (d == 32 && m.ng == 4) SBR_WAVE_BWD(32, 4)
    else if (d == 32) SBR_WAVE_BWD(32, 3)
    else if (m.ng == 4) SBR_WAVE_BWD(16, 4)
    else SBR_WAVE_BWD(16, 3)
#undef SBR_WAVE_BWD
    return true;
}

}  // namespace sbr
