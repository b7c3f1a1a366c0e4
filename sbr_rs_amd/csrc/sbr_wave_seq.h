/* sbr_wave_seq.h — the recurrent pass of ONE sequence on one wavefront (forward and backward), as device functions: the bodies of
 * lstm_fwd_wave_kernel / lstm_bwd_wave_kernel (sbr_wave.hip: a workgroup of 256 threads per sequence, the design notes are there)
 * (the one-launch step runs of sbr_steps.hip use the same forms out of LDS-resident operands; the reference's own schedule,
 * /root/reference/src/models/sequence_model.rs:111-169).  `lds`: the caller's dynamic shared memory;
 * forward needs (seg (D + NG D) + NG D + D + seg) floats, backward (7 seg D + D + NG D + seg). */
#ifndef SBR_WAVE_SEQ_H
#define SBR_WAVE_SEQ_H

#include "sbr_kernels.h"
#include "sbr_numerics.h"

namespace sbr {
namespace wave_seq {

#define SBR_WAVE_SEQ_FWD_LDS_FLOATS(seg, d, ngd) ((size_t)(seg) * ((d) + (ngd)) + (ngd) + (d) + (seg))
#define SBR_WAVE_SEQ_BWD_LDS_FLOATS(seg, d, ngd) ((size_t)(seg) * 7 * (d) + (d) + (ngd) + (seg))

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
__device__ __forceinline__ void lstm_fwd_wave_seq(const ModelView& m, const MbView& mb, float* H, const WorkView& w, int seg, int b, float* lds) {
    constexpr int NGD = NG * D;
    constexpr int SLOTS = (NGD + 63) / 64;  // gate columns per lane of the recurrence wave
    constexpr int GROUPS = 64 / D;          // lane groups; every group evaluates the d cells
    constexpr int Q = D / 4;
    constexpr bool FAST = D == 32 && NG == 4;
    const int tid = threadIdx.x;
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
__device__ __forceinline__ void lstm_bwd_wave_seq(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, int seg, int b, float* lds) {
    constexpr int NGD = NG * D;
    constexpr int GROUPS = 64 / D;
    constexpr int Q = D / 4;
    const int tid = threadIdx.x;
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


}  // namespace wave_seq
}  // namespace sbr
#endif
