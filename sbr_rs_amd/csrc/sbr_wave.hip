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
 *             lane l continues the chains of gate columns l and l + 64 over h_{t-1}.  d = 32 with four gates (the
 *             reference's configuration): one packed fma per k with h_k from a v_readlane, the rational tanh of the lane's
 *             own two pre-activations, numerators / denominators across the half-waves by v_permlane32_swap — no LDS
 *             between two steps.  Other shapes: h_{t-1} and the pre-activations cross lanes through LDS, every lane
 *             evaluates the cell of unit l mod d.  The lane groups share the stores of G / C / H.
 *   backward  the gate values, cell states, tanh(c_t) and dloss/dh rows of every step are staged into LDS by all 256
 *             threads (the transcendental of the cell backward does not depend on the gradient coming down the sequence);
 *             wave 0 walks t downwards: cell backward in lane u = l mod d, dz to LDS and HBM, then lane c owns column c
 *             of dz W^T (row c of W in registers: 4d values) — columns < d are dX of the row, columns >= d the recurrent
 *             dh, which returns to the unit lanes by v_permlane32_swap / v_permlane16_swap.
 *   also here lstm_dw_block_kernel: the dense gradient of such steps with one wave per 32 x 32 output block.
 *
 * Arithmetic is the contract's, operation for operation (sbr_numerics.h; fma chains in k / j order from the bias / from
 * 0), so the results are bit-identical to the tile kernels' and the oracle's; tests/test_parity_gpu.py runs both forms.
 * A workgroup per sequence, any max_sequence_length (segments of SBR_WAVE_SEG steps are staged in LDS at a time): up to
 * SBR_WAVE_MAX_SEQ sequences per step (twice that at d = 16) take this path (launch_wave_*
 * return false otherwise and the caller launches the tile kernels).  The machine model behind the design — a lone wave
 * issues a dependent vector instruction every 8.8 cycles, so a step costs its instruction count — is measured by
 * tools/valu_chain_ubench.hip; numbers in profiles/r03_small_steps.md. */
#include <cstdlib>

#include "sbr_kernels.h"
#include "sbr_numerics.h"

namespace sbr {
namespace {

#ifndef SBR_WAVE_MAX_SEQ
/* ms per step, len <= 64, wave / tile form of the recurrent kernels (everything else equal): d = 32: 1 sequence 0.126 / 0.216,
 * 256 0.224 / 0.437, 2 048 0.336 / 0.458, 4 096 0.629 / 0.603, 8 192 1.01 / 0.705; d = 16: 2 048 0.230 / 0.373, 4 096 0.411 / 0.477,
 * 8 192 0.501 / 0.506 — the wave form up to 2 048 sequences per step at d = 32, 4 096 at d = 16 */
#define SBR_WAVE_MAX_SEQ 2048
#endif
constexpr size_t WAVE_LDS_LIMIT = 150 * 1024;
#ifndef SBR_WAVE_SEG
#define SBR_WAVE_SEG 128 /* time steps of a sequence staged in LDS at a time (80 / 112 KB forward / backward at d = 32) */
#endif
#define SBR_WAVE_RSRC_FLAGS 0x00020000 /* word 3 of a raw buffer resource on gfx9-class targets (32-bit data format, no swizzle) */
#ifndef SBR_DW_BLOCK_MAX_CHUNKS
#define SBR_DW_BLOCK_MAX_CHUNKS 128 /* x 1-2 block rows = workgroups: the chip's CUs once */
#endif

__device__ __forceinline__ float4 ld4w(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4w(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_splat(float x) { return (v2f){x, x}; }
// sbr_tanh_pq (sbr_approx.h) on two arguments at once: packed f32 arithmetic, the same operations per element
__device__ __forceinline__ void tanh_pq2(v2f x, v2f* p, v2f* q) {
    // the clamp of sbr_tanh_pq (two comparison + select pairs; a NaN stays NaN) as one v_med3_f32 — which would turn a NaN into
    // -C — and one unordered comparison + select that puts the NaN back: the same bits for every input, four issue slots
    // fewer per value on a wave that pays ~7 cycles for each
    const float cx = __builtin_amdgcn_fmed3f(x.x, -SBR_TANH_CLAMP, SBR_TANH_CLAMP), cy = __builtin_amdgcn_fmed3f(x.y, -SBR_TANH_CLAMP, SBR_TANH_CLAMP);
    x.x = x.x != x.x ? x.x : cx;
    x.y = x.y != x.y ? x.y : cy;
    const v2f x2 = x * x;
    v2f n = pk_splat(-2.76076847742355e-16f);
    n = pk_fma(n, x2, pk_splat(2.00018790482477e-13f));
    n = pk_fma(n, x2, pk_splat(-8.60467152213735e-11f));
    n = pk_fma(n, x2, pk_splat(5.12229709037114e-08f));
    n = pk_fma(n, x2, pk_splat(1.48572235717979e-05f));
    n = pk_fma(n, x2, pk_splat(6.37261928875436e-04f));
    n = pk_fma(n, x2, pk_splat(4.89352455891786e-03f));
    *p = n * x;
    v2f dq = pk_splat(1.19825839466702e-06f);
    dq = pk_fma(dq, x2, pk_splat(1.18534705686654e-04f));
    dq = pk_fma(dq, x2, pk_splat(2.26843463243900e-03f));
    dq = pk_fma(dq, x2, pk_splat(4.89352518554385e-03f));
    *q = dq;
}
// LDS traffic between the lanes of ONE wave: the hardware executes a wave's LDS instructions in order, so a write followed
// by another lane's read needs no barrier — only the compiler must keep the program order
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// Both kernels walk a sequence in SEGMENTS of at most `seg` time steps (SBR_WAVE_SEG: what of a sequence is staged in LDS at
// a time — any max_sequence_length fits): all four waves stage a segment, wave 0 runs the recurrence over it with its state
// in registers while the others wait at the segment's closing barrier.
template <int D, int NG>
__global__ __launch_bounds__(256) void lstm_fwd_wave_kernel(ModelView m, MbView mb, float* H, WorkView w, int seg) {
    constexpr int NGD = NG * D;
    constexpr int SLOTS = (NGD + 63) / 64;  // gate columns per lane of the recurrence wave
    constexpr int GROUPS = 64 / D;          // lane groups; every group evaluates the d cells
    constexpr int Q = D / 4;
    constexpr bool FAST = D == 32 && NG == 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int nsteps = mb.steps[b];
    float* Xs = lds;             // [seg][D]   gathered input rows
    float* Ps = Xs + seg * D;    // [seg][NGD] bias + x part of the pre-activations
    float* Zs = Ps + seg * NGD;  // [NGD]      pre-activations of the current step (general path)
    float* Hs = Zs + NGD;        // [D]        h_{t-1} (general path)
    int* Rs = reinterpret_cast<int*>(Hs + D);  // [seg] packed row of the segment's steps
    const int lane = tid & 63;
    const int u = lane % D, grp = lane / D;
    const bool recur = tid < 64;  // wave 0 runs the recurrence
    if (tid < D) Hs[tid] = 0.0f;  // h_{-1} = 0
    // weights of the recurrence wave: the h rows of this lane's gate columns
    v2f w2[FAST ? D : 1];
    float wh[FAST ? 1 : SLOTS][FAST ? 1 : D];
    if (recur) {
        if constexpr (FAST) {
#pragma unroll
            for (int k = 0; k < D; ++k) w2[k] = (v2f){m.W[(size_t)(D + k) * NGD + lane], m.W[(size_t)(D + k) * NGD + 64 + lane]};
        } else {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int j = s * 64 + lane < NGD ? s * 64 + lane : 0;
#pragma unroll
                for (int k = 0; k < D; ++k) wh[s][k] = m.W[(size_t)(D + k) * NGD + j];
            }
        }
    }
    // weights of the x half: thread (stream, j) forms column j of steps stream, stream + NSTREAM, ... of every segment
    constexpr int NSTREAM = 256 / NGD;
    const int stream = tid / NGD, jx = tid % NGD;
    float wx[D];
    float bj = 0.0f;
    if (stream < NSTREAM) {
#pragma unroll
        for (int k = 0; k < D; ++k) wx[k] = m.W[(size_t)k * NGD + jx];
        bj = m.bW[jx];
    }
    float c_prev = 0.0f, h_prev = 0.0f;
    for (int t0 = 0; t0 < nsteps; t0 += seg) {
        const int ns = nsteps - t0 < seg ? nsteps - t0 : seg;
        for (int i = tid; i < ns; i += 256) Rs[i] = mb.off[t0 + i] + b;
        __syncthreads();
        for (int idx = tid; idx < ns * Q; idx += 256) {
            const int i = idx / Q, c4 = (idx % Q) * 4;
            const int r = Rs[i];
            const float4 v = ld4w(m.E + (size_t)mb.in_idx[r] * D + c4);
            st4w(&Xs[i * D + c4], v);
            st4w(w.X + (size_t)r * D + c4, v);
        }
        __syncthreads();
        if (stream < NSTREAM) {
            for (int i = stream; i < ns; i += NSTREAM) {
                float acc = bj;
#pragma unroll
                for (int k = 0; k < D; k += 4) {
                    const float4 x = ld4w(&Xs[i * D + k]);
                    acc = sbr_fma(x.x, wx[k], acc);
                    acc = sbr_fma(x.y, wx[k + 1], acc);
                    acc = sbr_fma(x.z, wx[k + 2], acc);
                    acc = sbr_fma(x.w, wx[k + 3], acc);
                }
                Ps[i * NGD + jx] = acc;
            }
        }
        __syncthreads();
        // ---- the recurrence over the segment: wave 0 alone, no barriers
        if (recur) {
            if constexpr (FAST) {
                // d = 32, four gates: lane l owns gate columns l and 64 + l — (i_u, g_u) in lanes u < 32, (f_u, o_u) in lanes 32 + u.
                // Both chains advance in one packed fma per k; every lane evaluates the rational tanh of ITS two pre-activations
                // (two, not four, per lane), and the numerators / denominators cross the wave halves with v_permlane32_swap —
                // no LDS round trip between the chain and the cell.
                const bool low = lane < 32;
                for (int i = 0; i < ns; ++i) {
                    v2f z2 = (v2f){Ps[i * NGD + lane], Ps[i * NGD + 64 + lane]};
                    // h_{t-1}[k] lives in lane k: a v_readlane per k puts it into a scalar register.  Each costs an issue slot of
                    // its own (17 cycles per readlane + fma pair, tools/valu_chain_ubench.hip), but there is no LDS write -> 16-byte
                    // read round trip between the cell and the chain any more: 44.1 -> 40.1 us per MovieLens subsequence
                    // (all reads first, into scalar registers of their own: read next to its use, every fma waits out the
                    // scalar-write hazard — 42 s_nop per step in the compiler's schedule)
                    float hk[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) hk[k] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(h_prev), k));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < D; ++k) z2 = pk_fma(pk_splat(hk[k]), w2[k], z2);
                    // arguments of the rational tanh: i, f, o are sigmoids (tanh of half the pre-activation), g is a tanh
                    const float half1 = 0.5f * z2.y;
                    v2f p2, q2;
                    tanh_pq2((v2f){0.5f * z2.x, low ? z2.y : half1}, &p2, &q2);
                    const v2u sp0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p2.x), __float_as_uint(p2.x), false, false);
                    const v2u sq0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(q2.x), __float_as_uint(q2.x), false, false);
                    const v2u sp1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p2.y), __float_as_uint(p2.y), false, false);
                    const v2u sq1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(q2.y), __float_as_uint(q2.y), false, false);
                    // .x = the value of lane u (lower half), .y = the value of lane 32 + u, in both halves
                    const float pi = __uint_as_float(sp0.x), pf = __uint_as_float(sp0.y), qi = __uint_as_float(sq0.x), qf = __uint_as_float(sq0.y);
                    const float pg = __uint_as_float(sp1.x), po = __uint_as_float(sp1.y), qg = __uint_as_float(sq1.x), qo = __uint_as_float(sq1.y);
                    // sbr_lstm_cell_fwd from here on, operation for operation
                    const float q_if = qi * qf, q_go = qg * qo;
                    const float r = 1.0f / (q_if * q_go);
                    const float r_if = r * q_go, r_go = r * q_if;
                    const float gf = sbr_fma(0.5f, pf * (r_if * qi), 0.5f);
                    const float gi = sbr_fma(0.5f, pi * (r_if * qf), 0.5f);
                    const float gg = pg * (r_go * qo);
                    const float go = sbr_fma(0.5f, po * (r_go * qg), 0.5f);
                    const float cc = sbr_fma(gf, c_prev, gi * gg);
                    c_prev = cc;
                    const int r_ = __builtin_amdgcn_readfirstlane(Rs[i]);
                    float* Grow = w.G + (size_t)r_ * 4 * D;
                    Grow[lane] = low ? gi : gf;
                    Grow[64 + lane] = low ? gg : go;
                    h_prev = go * sbr_tanhf(cc);
                    if (low) {
                        w.C[(size_t)r_ * D + u] = cc;
                        H[(size_t)r_ * D + u] = h_prev;
                    }
                }
            } else {
                for (int i = 0; i < ns; ++i) {
                    float z[SLOTS];
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) z[s] = Ps[i * NGD + (s * 64 + lane < NGD ? s * 64 + lane : 0)];
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
                    const int r = __builtin_amdgcn_readfirstlane(Rs[i]);
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
        }
        __syncthreads();  // the segment's rows in LDS are free again
    }
}

template <int D, int NG>
__global__ __launch_bounds__(256) void lstm_bwd_wave_kernel(ModelView m, MbView mb, BlockView blk, WorkView w, int seg) {
    constexpr int NGD = NG * D;
    constexpr int GROUPS = 64 / D;
    constexpr int Q = D / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int nsteps = mb.steps[b];
    float* Gs = lds;                     // [seg][4D]   gate values
    float* Cs = Gs + seg * 4 * D;        // [seg + 1][D] cell states; row 0 = the step before the segment (zeros before step 0)
    float* DHs = Cs + (seg + 1) * D;     // [seg][D]    dloss/dh
    float* TCs = DHs + seg * D;          // [seg][D]    tanh(c_t)
    float* Zs = TCs + seg * D;           // [NGD]       dz of the current step
    int* Rs = reinterpret_cast<int*>(Zs + NGD);  // [seg] packed row of the segment's steps
    const int lane = tid & 63;
    const int u = lane % D, grp = lane / D;
    const bool recur = tid < 64;
    const int col = lane < 2 * D ? lane : 0;  // column of dz W^T this lane owns (lanes >= 2d idle along)
    float wt[NGD];                            // W^T[j][col] = W[col][j]: row `col` of W
    if (recur) {
#pragma unroll
        for (int j = 0; j < NGD; j += 4) {
            const float4 v = ld4w(m.W + (size_t)col * NGD + j);
            wt[j] = v.x; wt[j + 1] = v.y; wt[j + 2] = v.z; wt[j + 3] = v.w;
        }
    }
    float dc = 0.0f, rec = 0.0f;  // carry of the cell state gradient, recurrent dh of unit u (every lane group holds both)
    const int nseg = (nsteps + seg - 1) / seg;
    for (int sg = nseg - 1; sg >= 0; --sg) {  // segments from the end of the sequence
        const int t0 = sg * seg;
        const int ns = nsteps - t0 < seg ? nsteps - t0 : seg;
        for (int i = tid; i < ns; i += 256) Rs[i] = mb.off[t0 + i] + b;
        __syncthreads();
        for (int idx = tid; idx < ns * Q; idx += 256) {
            const int i = idx / Q, c4 = (idx % Q) * 4;
            const size_t r = (size_t)Rs[i];
            const float g = blk.coef[r];
            const float4 en = ld4w(m.E + (size_t)blk.neg[r] * D + c4);
            const float4 ep = ld4w(m.E + (size_t)blk.out_idx[r] * D + c4);
            float4 dh;  // g*E[neg] - g*E[pos]: two rounded products, one subtraction (the oracle's order)
            dh.x = g * en.x - g * ep.x;
            dh.y = g * en.y - g * ep.y;
            dh.z = g * en.z - g * ep.z;
            dh.w = g * en.w - g * ep.w;
            st4w(&DHs[i * D + c4], dh);
            const float4 cv = ld4w(w.C + r * D + c4);
            st4w(&Cs[(i + 1) * D + c4], cv);
            // tanh(c_t): the one transcendental of the cell backward does not depend on the gradient coming down the sequence,
            // so all 256 threads evaluate it here instead of the lone recurrence wave (which pays ~7 cycles per instruction)
            st4w(&TCs[i * D + c4], make_float4(sbr_tanhf(cv.x), sbr_tanhf(cv.y), sbr_tanhf(cv.z), sbr_tanhf(cv.w)));
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) st4w(&Gs[i * 4 * D + g4 * D + c4], ld4w(w.G + r * 4 * D + g4 * D + c4));
        }
        for (int c4 = tid * 4; c4 < D; c4 += 1024)  // c of the step before the segment
            st4w(&Cs[c4], t0 > 0 ? ld4w(w.C + (size_t)(mb.off[t0 - 1] + b) * D + c4) : make_float4(0.f, 0.f, 0.f, 0.f));
        __syncthreads();
        if (recur) {
            for (int i = ns - 1; i >= 0; --i) {
                const int t = t0 + i;
                const bool carried = t + 1 < nsteps;  // the sequence has a step t+1: recurrent dh and dc exist
                const float dh = DHs[i * D + u] + (carried ? rec : 0.0f);
                const float gi = Gs[i * 4 * D + u], gf = Gs[i * 4 * D + D + u], gg = Gs[i * 4 * D + 2 * D + u], go = Gs[i * 4 * D + 3 * D + u];
                const float tc = TCs[i * D + u];
                const float cp = Cs[i * D + u];  // c_{t-1}
                // sbr_lstm_cell_bwd, operation for operation (tanh(c_t) was formed in the staging phase)
                const float d_o = dh * tc;
                const float dcv = sbr_fma(dh * go, 1.0f - tc * tc, carried ? dc : 0.0f);
                const float di = dcv * gg, dg = dcv * gi;
                float df = dcv * cp;
                dc = dcv * gf;
                float dz[4];
                if (NG == 3) { df = df - di; dz[0] = 0.0f; } else { dz[0] = di * (gi * (1.0f - gi)); }
                dz[1] = df * (gf * (1.0f - gf));
                dz[2] = dg * (1.0f - gg * gg);
                dz[3] = d_o * (go * (1.0f - go));
                const int r = __builtin_amdgcn_readfirstlane(Rs[i]);
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
                float acc = 0.0f;  // column `col` of dz W^T, j ascending from 0; dz read back as broadcast 16-byte pieces
#pragma unroll
                for (int j = 0; j < NGD; j += 4) {
                    const float4 zv = ld4w(&Zs[j]);
                    acc = sbr_fma(zv.x, wt[j], acc);
                    acc = sbr_fma(zv.y, wt[j + 1], acc);
                    acc = sbr_fma(zv.z, wt[j + 2], acc);
                    acc = sbr_fma(zv.w, wt[j + 3], acc);
                }
                if (lane < D) blk.dX[(size_t)r * D + lane] = acc;
                // recurrent dh of unit u = column d + u, held by lane d + u: into every lane group, inside the vector ALU
                if constexpr (D == 32) {
                    const v2u sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
                    rec = __uint_as_float(sw.y);  // .y = the value of the upper half-wave's lane 32 + u
                } else {
                    const v2u sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
                    // .y = the value of the odd 16-lane row of each pair: lane 16 + u for lanes 0..31; then the lower half-wave's
                    // value into the upper one (lanes >= 32 own no column)
                    const v2u lo = __builtin_amdgcn_permlane32_swap(sw.y, sw.y, false, false);
                    rec = __uint_as_float(lo.x);
                }
                wave_lds_fence();  // the next step's dz must not overwrite Zs before every lane has read it
            }
        }
        __syncthreads();  // the segment's rows in LDS are free again
    }
}

// ------------------------------------------------------------------------------------------------
// Dense gradient of a SMALL step: dW[k][j] = sum_r xh[r][k] dz[r][j] per 1 024-row chunk, one WAVE per 32 x 32 block of
// the output.  The chunk's chain (rows ascending, from 0: SBR_DW_CHUNK_ROWS) is 512 dependent v_mfma_f32_32x32x2_f32 for
// every accumulator, whoever computes it; the tile kernel of sbr_kernels.hip gives a wave four accumulators (and at
// d = 32 half of its 128 x 128 tile lies outside the matrix), so a chunk costs it 4 x 512 MFMAs back to back — 55-130 us
// — which is what a step of a few hundred sequences then waits for.  Here a chunk is 512 MFMAs deep (14 us): one
// workgroup per chunk and block ROW, one wave per block — the waves of a workgroup sit on different SIMDs, each with an
// MFMA pipe to itself (two dependent chains on one SIMD take turns: measured 141 cycles per MFMA and wave) — 64-row slabs of
// [32 columns of xh | dz] staged through LDS (double-buffered in LDS, two slabs ahead in registers), operands read back
// one float per lane and MFMA, eight MFMAs' reads ahead.  Rows past the chunk's end, first-step rows'
// h_{t-1} and padding columns are zeros (requests beyond the buffer resources), i.e. +0 on every chain.  The bias row
// (column sums of dz, a plain add chain over the rows) is formed by the waves of the first block row from the staged dz.
// ------------------------------------------------------------------------------------------------
typedef float f32x16w __attribute__((ext_vector_type(16)));
typedef unsigned v4uw __attribute__((ext_vector_type(4)));
template <int D, int NG>
struct DwBlockCfg {
    static constexpr int K2 = 2 * D, NGD = NG * D;
    static constexpr int KB = (K2 + 31) / 32, JB = (NGD + 31) / 32;  // blocks; a workgroup = one block ROW of one chunk
    static constexpr int NL = 4;                                     // loader waves (one per SIMD, beside the MFMA waves)
    static constexpr int NT = (JB + NL + 1) * 64;                    // + the wave that forms the bias row
    static constexpr int KP = 32, JP = JB * 32, LD = KP + JP;        // floats per staged row: the block row's 32 columns of xh, all of dz
    static constexpr int S = 64;                                     // rows per slab
    static constexpr size_t lds_bytes = (size_t)2 * S * LD * 4 + SBR_DW_CHUNK_ROWS * 4;
};
template <int D, int NG>
__global__ __launch_bounds__((DwBlockCfg<D, NG>::NT)) void lstm_dw_block_kernel(ModelView m, MbView mb, BlockView blk, WorkView w) {
    using Cfg = DwBlockCfg<D, NG>;
    constexpr int K2 = Cfg::K2, NGD = Cfg::NGD, JB = Cfg::JB, NT = Cfg::NT, KP = Cfg::KP, LD = Cfg::LD, S = Cfg::S;
    constexpr int NLT = Cfg::NL * 64;                         // loader threads
    constexpr int QX = 32 / 4, QZ = NGD / 4;                  // 16-byte pieces per row of the xh block row and of dZ
    constexpr int IX = (S * QX + NLT - 1) / NLT, IZ = (S * QZ + NLT - 1) / NLT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* buf0 = lds;
    float* buf1 = lds + S * LD;
    int* s_prev = reinterpret_cast<int*>(lds + 2 * S * LD);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = blockIdx.x % Cfg::KB;
    const int c = blockIdx.x / Cfg::KB;
    const int r0 = c * SBR_DW_CHUNK_ROWS;
    const int nr = mb.R - r0 < SBR_DW_CHUNK_ROWS ? mb.R - r0 : SBR_DW_CHUNK_ROWS;
    const int nslabs = (nr + S - 1) / S;
    for (int i = tid; i < SBR_DW_CHUNK_ROWS; i += NT) s_prev[i] = i < nr ? mb.prev_row[r0 + i] : -1;
    // padding columns of both buffers (d = 16 with three gates: dz columns 48..63) stay zero: cleared once
    for (int i = tid; i < 2 * S * LD; i += NT) lds[i] = 0.0f;
    __syncthreads();
    float* part = w.partials + (size_t)c * (K2 + 1) * NGD;
    if (wave < JB) {
        // ---- MFMA wave: block (kb, jb = wave); per slab 32 dependent MFMAs, operands from LDS, eight MFMAs' reads ahead
        const int jb = wave, l31 = lane & 31, par = lane >> 5;
        f32x16w acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
        __syncthreads();  // slab 0 staged
        for (int slab = 0; slab < nslabs; ++slab) {
            const float* cur = (slab & 1) ? buf1 : buf0;
            const float* ap = cur + par * LD + l31;
            const float* bp = cur + par * LD + KP + jb * 32 + l31;
            constexpr int G = 8;
            float av[2][G], bv[2][G];
#pragma unroll
            for (int q = 0; q < G; ++q) { av[0][q] = ap[2 * q * LD]; bv[0][q] = bp[2 * q * LD]; }
#pragma unroll
            for (int g = 0; g < S / 2 / G; ++g) {
                if (g + 1 < S / 2 / G) {
#pragma unroll
                    for (int q = 0; q < G; ++q) {
                        av[(g + 1) & 1][q] = ap[2 * ((g + 1) * G + q) * LD];
                        bv[(g + 1) & 1][q] = bp[2 * ((g + 1) * G + q) * LD];
                    }
                }
#pragma unroll
                for (int q = 0; q < G; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][q], bv[g & 1][q], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
        const int j = jb * 32 + l31;
        if (j < NGD) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int kk = kb * 32 + (q & 3) + 8 * (q >> 2) + 4 * par;
                if (kk < K2) part[(size_t)kk * NGD + j] = acc[q];
            }
        }
        return;
    }
    if (wave == JB + Cfg::NL) {
        // ---- bias wave: row k = 2d of the gradient = column sums of dz, a plain add chain over the rows of the chunk (rows past
        // its end add +0), read from the staged slabs; lane l owns columns l and 64 + l.  (A lone wave pays ~7 cycles per
        // instruction: folded into the MFMA waves these adds cost a third of the MFMA rate, folded into a loader wave they made
        // it the slowest wave of the workgroup.)  Only the first block row's workgroup has work here.
        float bias0 = 0.0f, bias1 = 0.0f;
        __syncthreads();  // slab 0 staged
        for (int slab = 0; slab < nslabs; ++slab) {
            if (kb == 0) {
                const float* cur = (slab & 1) ? buf1 : buf0;
                const float* z0 = cur + KP + (lane < NGD ? lane : 0);
                const float* z1 = cur + KP + (64 + lane < NGD ? 64 + lane : 0);
#pragma unroll 16
                for (int i = 0; i < S; ++i) {
                    bias0 = bias0 + z0[i * LD];
                    bias1 = bias1 + z1[i * LD];
                }
            }
            __syncthreads();
        }
        if (kb == 0) {
            if (lane < NGD) part[(size_t)K2 * NGD + lane] = bias0;
            if (64 + lane < NGD) part[(size_t)K2 * NGD + 64 + lane] = bias1;
        }
        return;
    }
    // ---- loader waves: stage the slabs (rows past the chunk, h_{t-1} of first steps, columns outside the matrix: requests beyond
    // the buffer resources, i.e. zeros)
    const int lt = tid - JB * 64;  // loader thread id
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)w.X, (short)0, mb.R * D * 4, SBR_WAVE_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)blk.H, (short)0, mb.R * D * 4, SBR_WAVE_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void*)w.dZ, (short)0, mb.R * NGD * 4, SBR_WAVE_RSRC_FLAGS);
    struct Regs { v4uw xr[IX], hr[IX], zr[IZ]; };
    Regs ra, rb;  // two slabs in flight towards LDS (a request has two slabs' MFMAs to land under)
    auto fetch = [&](Regs& rg, int slab) {
#pragma unroll
        for (int it = 0; it < IX; ++it) {
            const int p = lt + it * NLT, lr = slab * S + p / QX, k4 = kb * 32 + (p % QX) * 4;  // column of xh = [x | h]
            const bool ok = p < S * QX && lr < nr;
            const int pr = s_prev[lr < SBR_DW_CHUNK_ROWS ? lr : 0];
            // one of the two requests of a piece falls outside its resource and returns zeros: the piece is their OR
            rg.xr[it] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok && k4 < D ? (unsigned)((r0 + lr) * D + k4) * 4u : OOB, 0, 0);
            rg.hr[it] = __builtin_amdgcn_raw_buffer_load_b128(rsH, ok && k4 >= D && k4 < K2 && pr >= 0 ? (unsigned)(pr * D + k4 - D) * 4u : OOB, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < IZ; ++it) {
            const int p = lt + it * NLT, lr = slab * S + p / QZ, c4 = (p % QZ) * 4;
            const bool ok = p < S * QZ && lr < nr;
            rg.zr[it] = __builtin_amdgcn_raw_buffer_load_b128(rsZ, ok ? (unsigned)((r0 + lr) * NGD + c4) * 4u : OOB, 0, 0);
        }
    };
    auto stage = [&](const Regs& rg, float* buf) {
#pragma unroll
        for (int it = 0; it < IX; ++it) {
            const int p = lt + it * NLT, row = p / QX, c4 = (p % QX) * 4;
            if (p < S * QX) *reinterpret_cast<v4uw*>(&buf[row * LD + c4]) = rg.xr[it] | rg.hr[it];
        }
#pragma unroll
        for (int it = 0; it < IZ; ++it) {
            const int p = lt + it * NLT, row = p / QZ, c4 = (p % QZ) * 4;
            if (p < S * QZ) *reinterpret_cast<v4uw*>(&buf[row * LD + KP + c4]) = rg.zr[it];
        }
    };
    // slab s is computed from LDS buffer s & 1 while slab s + 1 goes from registers into the other buffer (its readers passed
    // the barrier at the end of slab s - 1) and slab s + 3 is requested into the registers that have just been emptied;
    // requests past the last slab fall outside the chunk and return zeros, which nobody reads
    fetch(ra, 0);
    stage(ra, buf0);
    fetch(ra, 1);
    fetch(rb, 2);
    __syncthreads();  // slab 0 staged
    for (int slab = 0; slab < nslabs; slab += 2) {
        stage(ra, buf1);
        fetch(ra, slab + 3);
        __syncthreads();
        if (slab + 1 < nslabs) {
            stage(rb, buf0);
            fetch(rb, slab + 4);
            __syncthreads();
        }
    }
}

int wave_mode() {  // SBR_WAVE: 0 never, 1 whenever the shape allows, unset: up to SBR_WAVE_MAX_SEQ sequences per step
    const char* e = std::getenv("SBR_WAVE");  // read per call: the tests force either form
    return e ? (std::atoi(e) ? 1 : 0) : -1;
}
bool wave_shape_ok(int d, int ng, int B, size_t lds_bytes) {
    if ((d != 16 && d != 32) || (ng != 3 && ng != 4) || B <= 0 || lds_bytes > WAVE_LDS_LIMIT) return false;
    const int mode = wave_mode();
    return mode == 1 || (mode < 0 && B <= (d == 16 ? 2 * SBR_WAVE_MAX_SEQ : SBR_WAVE_MAX_SEQ));
}
// dynamic LDS beyond the default limit has to be granted per kernel; `granted` is the instantiation's high-water mark, so a
// step of a few microseconds does not pay for the call again
// (per device: the attribute belongs to the function on the current device, and one process may drive several)
template <class K>
void allow_lds(K kernel, size_t bytes, size_t* granted) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev >= 0 && dev < 64 ? dev : 0;
    if (bytes <= granted[dev]) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    granted[dev] = bytes;
}

}  // namespace

bool launch_wave_forward(const ModelView& m, const MbView& mb, float* H, const WorkView& w, int tm_host, hipStream_t s) {
    const int d = m.d, ngd = m.ng * m.d;
    const int seg = tm_host < SBR_WAVE_SEG ? tm_host : SBR_WAVE_SEG;
    const size_t lds = ((size_t)seg * (d + ngd) + ngd + d + seg) * 4;
    if (!wave_shape_ok(d, m.ng, mb.B, lds)) return false;
#define SBR_WAVE_FWD(DD, NN)                                                                                   \
    {                                                                                                          \
        static size_t granted[64] = {0};                                                                             \
        allow_lds(lstm_fwd_wave_kernel<DD, NN>, lds, granted);                                                \
        hipLaunchKernelGGL((lstm_fwd_wave_kernel<DD, NN>), dim3(mb.B), dim3(256), lds, s, m, mb, H, w, seg);           \
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
    const int seg = tm_host < SBR_WAVE_SEG ? tm_host : SBR_WAVE_SEG;
    const size_t lds = ((size_t)seg * 7 * d + d + ngd + seg) * 4;
    if (!wave_shape_ok(d, m.ng, b_host, lds)) return false;
#define SBR_WAVE_BWD(DD, NN)                                                                                   \
    {                                                                                                          \
        static size_t granted[64] = {0};                                                                             \
        allow_lds(lstm_bwd_wave_kernel<DD, NN>, lds, granted);                                                \
        hipLaunchKernelGGL((lstm_bwd_wave_kernel<DD, NN>), dim3(b_host), dim3(256), lds, s, m, mb, blk, w, seg);       \
    }
    if (d == 32 && m.ng == 4) SBR_WAVE_BWD(32, 4)
    else if (d == 32) SBR_WAVE_BWD(32, 3)
    else if (m.ng == 4) SBR_WAVE_BWD(16, 4)
    else SBR_WAVE_BWD(16, 3)
#undef SBR_WAVE_BWD
    return true;
}

bool launch_wave_dense_gradient(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, int rows_host,
                                hipStream_t s) {
    const int d = m.d;
    if ((d != 16 && d != 32) || (m.ng != 3 && m.ng != 4) || rows_host <= 0) return false;
    const char* e = std::getenv("SBR_DW_BLOCK");  // 0 never, 1 always (tests), unset: while the waves fit the SIMDs about once
    const int mode = e ? (std::atoi(e) ? 1 : 0) : -1;
    const int nch = (rows_host + SBR_DW_CHUNK_ROWS - 1) / SBR_DW_CHUNK_ROWS;
    if (mode == 0 || (mode < 0 && nch > SBR_DW_BLOCK_MAX_CHUNKS)) return false;
    if ((size_t)rows_host * m.ng * d * 4 >= ((size_t)1 << 31)) return false; /* the kernel's 32-bit buffer offsets */
#define SBR_DW_BLOCK_LAUNCH(DD, NN)                                                                                          \
    {                                                                                                                        \
        using Cfg = DwBlockCfg<DD, NN>;                                                                                      \
        static size_t granted[64] = {0};                                                                                           \
        allow_lds(lstm_dw_block_kernel<DD, NN>, Cfg::lds_bytes, granted);                                                    \
        hipLaunchKernelGGL((lstm_dw_block_kernel<DD, NN>), dim3(nch * Cfg::KB), dim3(Cfg::NT), Cfg::lds_bytes, s, m, mb, blk, w); \
    }
    if (d == 32 && m.ng == 4) SBR_DW_BLOCK_LAUNCH(32, 4)
    else if (d == 32) SBR_DW_BLOCK_LAUNCH(32, 3)
    else if (m.ng == 4) SBR_DW_BLOCK_LAUNCH(16, 4)
    else SBR_DW_BLOCK_LAUNCH(16, 3)
#undef SBR_DW_BLOCK_LAUNCH
    return true;
}

}  // namespace sbr
