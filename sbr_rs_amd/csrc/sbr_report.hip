// sbr_report.hip — the loss figure the reference's `fit` returns, on the device.
//
// /root/reference/src/models/sequence_model.rs:157 adds `loss.value().scalar_sum()` of the node losses[loss_idx] BEFORE :160 runs
// `loss.forward()` on it.  The nodes are the running sums of lstm.rs:322-328 / ewma.rs:337-343 — summed_losses[k] =
// summed_losses[k-1].clone() + loss_k, the SAME Rc node — so a forward pass over a sequence of s steps leaves its prefix sums
// L_0 .. L_{s-1} in nodes 0 .. s-1 and leaves the nodes above untouched.  What a worker accumulates for a sequence of s steps is
// therefore L_{s-1} of its most recent earlier sequence with AT LEAST s steps (0 if there was none since the graph was built, i.e.
// since the start of this fit call), in f32, and `fit` returns the sum over the workers of accumulator / (1 + examples)
// (:173-177).  The engine's sbr_fit_end reports the true sums; this file keeps the reference's figure beside it so that a caller
// behind INTEGRATION.md can return what the crate returns.
//
// Per minibatch (sequences b = 0..B-1 in packed order = steps descending, so "the most recent sequence with at least as many
// steps" of sequence b > 0 is sequence b - 1):
//   P_b(k)   = l_0 + l_1 + ... + l_k                          t-ascending f32 chain over sequence b's loss terms
//   x_0      = node[steps_0 - 1]                               what the previous minibatches left
//   x_b      = P_{b-1}(steps_b - 1)                            b > 0
//   total    = ((total + x_0) + x_1) + ...                     strictly sequential f32 chain over b
//   node[k]  = P_b(k) of the LAST b with steps_b > k            (k < steps_0; the nodes above keep their values)
// The x_b are independent of each other and only the chain over b is sequential: ONE wave walks it, 64 values per load,
// v_readlane + v_add_f32 per value (a lone wave issues a dependent VALU instruction every ~9 cycles), on the sorter stream
// underneath BPTT.  The CPU checker restates it sequentially, one sequence at a time (tests compare the two bit for bit).
//
// state layout: state[0] = the partition's accumulator (loss_value of sequence_model.rs:105); state[1 + 2k] = node k;
// state[2 + 2k] = the value node k takes at the end of the minibatch in flight (staging: x_0 is read before the nodes move).

#include <hip/hip_runtime.h>

#include "sbr_kernels.h"

namespace sbr {

// one thread per sequence: the running sums of the sequence, t ascending; px[b] = x_{b+1}; the nodes' next values
__global__ __launch_bounds__(256) void seq_loss_kernel(MbView mb, const float* loss, float* px, float* state) {
    __builtin_amdgcn_s_setprio(3); /* rides the ordering's stream underneath BPTT's older MFMA waves, like the ordering itself */
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= mb.B) return;
    const int n = mb.steps[b];
    const int nxt = b + 1 < mb.B ? mb.steps[b + 1] : 0;
    float s = 0.0f;
    int t = 0;
#pragma unroll 8
    for (; t < nxt; ++t) s = s + loss[mb.off[t] + b]; /* the common case: the next sequence is as long, nothing else to do */
    if (nxt) px[b] = s;
    for (; t < n; ++t) {
        s = s + loss[mb.off[t] + b];
        state[2 + 2 * t] = s;
    }
}

// One wave; `px` in global memory or LDS.  Tiles of LAG_TILE sequences: first the tile's x values with all their loads in flight
// together (they do not depend on the chain; a chunk-at-a-time version paid two dependent memory round trips per 64 sequences:
// 0.5 ms for 8 192 sequences), staged through `xs` (LDS, LAG_TILE floats); then the chain, 64 values per LDS read.  At the end the
// staged node values move into the nodes.
#define LAG_TILE 4096
__device__ __forceinline__ void lagged_chain(const float* px, int B, int s0, float* state, int lane, float* xs) {
    float acc = state[0];
    for (int tile0 = 0; tile0 < B; tile0 += LAG_TILE) {
        const int n = B - tile0 < LAG_TILE ? B - tile0 : LAG_TILE;
        const int nchunks = (n + 63) / 64;
#pragma unroll 8
        for (int c = 0; c < nchunks; ++c) {
            const int j = tile0 + c * 64 + lane;
            float x = 0.0f;
            if (j < tile0 + n) x = j == 0 ? state[1 + 2 * (s0 - 1)] : px[j - 1];
            xs[c * 64 + lane] = x;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float v = xs[lane];
        for (int c = 0; c < nchunks; ++c) {
            const float vn = xs[(c + 1 < nchunks ? c + 1 : c) * 64 + lane];  // next chunk's values while this chunk's chain runs
            // (a lane-to-lane form — one v_add_f32 with a wave_shr:1 DPP operand per element instead of v_readlane + v_add — was
            // measured and is no faster: beside the GEMM's back-to-back 64-cycle MFMAs this wave's vector instructions wait for the
            // pipe one MFMA pass at a time either way; NOTES.md, round 4)
            const int cnt = n - c * 64 < 64 ? n - c * 64 : 64;
            if (cnt == 64) {
#pragma unroll
                for (int l = 0; l < 64; ++l) acc = acc + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l));
            } else {
                for (int l = 0; l < cnt; ++l) acc = acc + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l));
            }
            v = vn;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    for (int t = lane; t < s0; t += 64) state[1 + 2 * t] = state[2 + 2 * t]; /* x_0 was read above: same wave, in order */
    if (lane == 0) state[0] = acc;
}

__global__ __launch_bounds__(64) void lagged_chain_kernel(const int* steps, const float* px, int B, float* state) {
    // one wave under BPTT's MFMA waves, which are older and win the SIMD's issue arbitration: at priority 0 the chain of 8 192
    // sequences took 0.44 ms of elapsed time for ~0.05 ms of dependent adds (profiles/r04_*kernel_stats*)
    __shared__ float xs[LAG_TILE];
    __builtin_amdgcn_s_setprio(3);
    lagged_chain(px, B, steps[0], state, threadIdx.x, xs);
}

// header of the exchange block: row count, loss sum (reporting only: order-free f64 reduction, compared with a tolerance) and
// example count.  loss_acc / ex_acc non-null (single device): the plan's accumulators take the header in the same launch.
// lag_state non-null (a small step, everything on one stream): the lagged figure of up to SBR_HEADER_LAG_MAX_B sequences in the
// same launch as well — a small step is a chain of ~5 us launches and cannot afford two more.
__global__ __launch_bounds__(1024) void block_header_kernel(uint32_t* header, int R, const double* part_loss,
                                                          const unsigned int* part_tries, int nparts, double* loss_acc,
                                                          unsigned long long* ex_acc, MbView mb, const float* loss, float* lag_state) {
    /* one workgroup, usually underneath BPTT on the ordering's stream: at priority 0 it took 1.5 ms of elapsed time at 50 000
     * sequences per step (every instruction waits for an MFMA pass of the older waves) and held the key ordering up by as much */
    __builtin_amdgcn_s_setprio(3);
    __shared__ double part[16];
    __shared__ unsigned int tpart[16];
    const int nthreads = (int)blockDim.x, nwaves = nthreads >> 6; /* 256, or 1 024 when several sequences' loss chains are walked here */
    /* the lagged figure's staging (24 KB) is DYNAMIC shared memory, asked for only by the small-step launch that uses it: as static
     * memory it kept the header launch of a 50 000-sequence step off the chip until a BPTT workgroup retired (64-sequence tiles
     * leave less than 24 KB of LDS per CU) — 1.5 ms during which the key ordering behind it on the stream could not start */
    extern __shared__ float lag_lds[];
    float* spx = lag_lds;
    float* sx = lag_lds + SBR_HEADER_LAG_MAX_B;
    double acc = 0.0;
    unsigned int tacc = 0;
    for (int i = threadIdx.x; i < nparts; i += nthreads) {
        acc += part_loss[i];
        tacc += part_tries[i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        acc += __shfl_xor(acc, off, 64);
        tacc += __shfl_xor(tacc, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6] = acc;
        tpart[threadIdx.x >> 6] = tacc;
    }
    if (lag_state) { /* a wave per sequence, 64 loss terms per coalesced load, then the t-ascending chain by v_readlane: one
                      * sequence per step (the reference's schedule) must not wait for ~100 dependent load-add pairs.  Lane l keeps
                      * the running sum after term base + l: the next sequence's x and the nodes' next values come from there */
        const int lane = threadIdx.x & 63;
        for (int b = threadIdx.x >> 6; b < mb.B; b += nwaves) {
            const int n = mb.steps[b];
            const int nxt = b + 1 < mb.B ? mb.steps[b + 1] : 0;
            float s = 0.0f;
            for (int base = 0; base < n; base += 64) {
                const int t = base + lane;
                const float v = t < n ? loss[mb.off[t] + b] : 0.0f;
                const int cnt = n - base < 64 ? n - base : 64;
                float mine = 0.0f;
                for (int l = 0; l < cnt; ++l) {
                    s = s + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l));
                    mine = lane == l ? s : mine;
                }
                if (t < n && t >= nxt) lag_state[2 + 2 * t] = mine;
                if (t + 1 == nxt) spx[b] = mine;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        header[0] = (uint32_t)R;
        unsigned int tsum = tpart[0];
        double lsum = part[0];
        for (int i = 1; i < nwaves; ++i) { tsum += tpart[i]; lsum += part[i]; }
        header[1] = tsum; /* negatives scored in this minibatch (reporting only) */
        header[2] = header[3] = 0;
        *reinterpret_cast<double*>(header + 4) = lsum;
        *reinterpret_cast<unsigned long long*>(header + 6) = (unsigned long long)R;
        if (loss_acc) { /* single device: the plan's accumulators take the header here (accumulate_loss_kernel with one block) */
            loss_acc[0] += lsum;
            loss_acc[1] += lsum;
            ex_acc[0] += (unsigned long long)R;
            ex_acc[1] += header[1];
            ex_acc[2] += (unsigned long long)R;
        }
    }
    /* (the barrier above orders the waves' node staging and spx before this wave's reads: one workgroup, one CU) */
    if (lag_state && threadIdx.x < 64 && mb.B > 0) lagged_chain(spx, mb.B, mb.steps[0], lag_state, threadIdx.x, sx);
}

void launch_block_header_parts(uint32_t* header, int rows_host, const double* part_loss, const unsigned int* part_tries, int nparts,
                               double* loss_acc, unsigned long long* ex_acc, const MbView& mb, const float* loss, float* lag_state,
                               hipStream_t s) {
    const size_t lds = lag_state ? (size_t)(SBR_HEADER_LAG_MAX_B + (SBR_HEADER_LAG_MAX_B < LAG_TILE ? LAG_TILE : SBR_HEADER_LAG_MAX_B)) * 4 : 0;
    /* a small step walks its sequences' loss chains here, a wave per sequence: sixteen waves when there are more than four sequences
     * (MovieLens-100K at 16 sequences per step: 16 -> 6 us of a 266 us step) */
    const int threads = lag_state && mb.B > 4 ? 1024 : 256;
    hipLaunchKernelGGL(block_header_kernel, dim3(1), dim3(threads), lds, s, header, rows_host, part_loss, part_tries, nparts, loss_acc, ex_acc,
                       mb, loss, lag_state);
}

void launch_seq_loss(const MbView& mb, const float* loss, float* px, float* lag_state, int b_host, hipStream_t s) {
    if (b_host <= 0) return;
    hipLaunchKernelGGL(seq_loss_kernel, dim3((b_host + 255) / 256), dim3(256), 0, s, mb, loss, px, lag_state);
}

void launch_lagged_chain(const MbView& mb, const float* px, int b_host, float* state, hipStream_t s) {
    if (b_host <= 0) return;
    hipLaunchKernelGGL(lagged_chain_kernel, dim3(1), dim3(64), 0, s, mb.steps, px, b_host, state);
}

}  // namespace sbr
