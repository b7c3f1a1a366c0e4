// sbr_kernels.hip — gfx950 (CDNA4, MI355X) kernels of the sequence-recommender hot path.
//
// Every kernel reproduces, bit for bit, the association orders fixed in sbr_numerics.h and
// restated sequentially by the CPU oracle (oracle/sbr_oracle.c):
//   * length-d score dots   : 16-byte-per-lane partials + xor butterfly inside a d/4-lane group
//   * LSTM gate / BPTT GEMMs: v_mfma_f32_16x16x4_f32, one accumulator per output, k ascending
//                             (an f32 MFMA is a k-ordered fmaf chain on gfx950)
//   * dense gradients       : v_mfma_f32_32x32x2_f32 over fixed 1024-row chunks, chunk partials
//                             added in chunk order
//   * sparse row gradients  : (row, source) keys in key order (sbr_sort.hip), per-row in-order reduction,
//                             one Adagrad read-modify-write per touched row
// Reference call sites replaced: the wyrm graph built by Parameters::build
// (/root/reference/src/models/lstm.rs:258-337, ewma.rs:266-352), driven by fit_sequence_model
// (/root/reference/src/models/sequence_model.rs:111-169), predict_single (lstm.rs:338-350) and
// the ranking loop of mrr_score (/root/reference/src/evaluation.rs:27-43).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see sbr_rs_amd/build.py).

#include "sbr_kernels.h"

#include <atomic>
#include <cstdlib>
#include <cstring>

#include "../../include/sbr_hip.h"
#include "sbr_device.h"
#include "sbr_numerics.h"

namespace sbr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define EWMA_CHUNK_SEQS 256
#ifndef SBR_FWD_RT
#define SBR_FWD_RT 4   /* 16-row tiles per sequence-resident workgroup (d >= 64; smaller d: 2) */
#endif
#ifndef SBR_FWD_WPE
#define SBR_FWD_WPE 3 /* min waves per SIMD asked of the compiler when a wave owns several unit tiles */
#endif
#define SBR_MAX_T 1024 /* longest max_sequence_length the sequence-resident kernels take (their per-step row offsets live in LDS: 4 KB); beyond it, per-step launches */
/* Two forms of the sequence-resident kernels: 32-sequence tiles at two workgroups per CU (128 registers per wave) and
 * 64-sequence tiles at one (256 registers; half the weight traffic per row, but nothing to fill a workgroup's barrier
 * waits with).  Since the address arithmetic left the vector ALU the 32-sequence form is the faster one up to ~1 500
 * 64-sequence tiles and equal beyond (ms forward / BPTT on the bench workload, 32- vs 64-sequence tiles: 30 000
 * sequences 2.21 / 2.31 vs 2.78 / 2.48; 50 000 (782 tiles) 3.92 / 3.99 vs 4.03 / 4.34; 100 000 (1 563 tiles)
 * 7.73-7.96 / 7.78-8.04 vs 7.66-7.76 / 7.91-7.93).  SBR_SEQ_RT = 2 / 4 forces one form. */
#ifndef SBR_FWD_RT4_MIN_TILES
#define SBR_FWD_RT4_MIN_TILES 1500
#endif
#ifndef SBR_SEQ_RT1_MAX_TILES
/* Below this many 32-sequence tiles the recurrent kernels take 16-SEQUENCE tiles (96 / 80 registers per wave): with few tiles
 * the kernel time is the longest tile's chain of dependent steps, and a 16-row step costs half the MFMA time of a 32-row
 * step on its CU.  M interactions/s of the whole step, 32- vs 16-sequence tiles, bench workload (profiles/r03_batch_sweep_tiles.md):
 * 256 sequences 3.4 / 5.8, 1 024 12.7 / 21.7, 4 096 43.7 / 66.4, 8 192 74.4 / 96.5, 16 384 (512 tiles) 100.5 / 104.0,
 * 50 000 122.3 / 114.7.  Since BPTT folds its tile list, keeps four weight blocks in flight and reads its dz operand a block
 * ahead (all of which the 16-sequence form gains most from), ms per step 32- / 16-sequence tiles on one box: 8 192 3.53 / 2.63,
 * 12 000 4.02 / 3.45, 20 000 5.96 / 5.85, 24 000 6.08 / 5.74, 30 000 7.60 / 7.27, 35 000 9.57 / 9.57, 40 000 9.82 / 9.92,
 * 50 000 13.66 / 13.79: the 16-sequence form up to 1 000 32-sequence tiles (32 000 sequences per step). */
#define SBR_SEQ_RT1_MAX_TILES 1000
#endif
#ifndef SBR_BWD_RT4_MIN_TILES
#define SBR_BWD_RT4_MIN_TILES 3000
#endif
#ifndef SBR_DW_LDS_PAD
#define SBR_DW_LDS_PAD 0
#endif
#ifndef SBR_DW_WPE
#define SBR_DW_WPE 4 /* waves per SIMD the dense-gradient kernel's register budget is set for (110 registers at 16-row slabs) */
#endif
/* rows staged through LDS per barrier pair of the dense-gradient kernel: 16 (20 KB of LDS and 110 registers per workgroup,
 * four workgroups per CU) measured 1-3 % ahead of 32 alone and beside the sparse update, 64 (two workgroups per CU) 10 % behind */
#ifndef SBR_DW_SLAB
#define SBR_DW_SLAB 16
#endif
#ifndef SBR_BWD256_MIN_TILES
#define SBR_BWD256_MIN_TILES 320
#endif
#ifndef SBR_FWD_UPW
#define SBR_FWD_UPW 1  /* 16-unit tiles per wave */
#endif

// word 3 of a raw buffer resource on gfx9-class targets (32-bit data format, no swizzle)
#define SBR_BUFFER_RSRC_FLAGS 0x00020000

// (16-byte loads / stores, the group all-reduce of the dot order, the optimiser element update, the packed-f32 helpers and the
// SmallTail of a one-sequence step are in sbr_device.h: sbr_steps.hip shares them)


// Streaming (non-temporal, `nt`) forms of the 16-byte row accesses, per kernel family by measurement
// (profiles/r05_streaming_gathers.md; the SBR_NT_* macros are build-time A/B hooks, tools/build_variant.sh):
//   score kernels (table rows gathered once, h rows read once): ON while the step's h rows fit the Infinity Cache (stream_policy,
//     launch_score) — the launch is 10-12 % shorter in the step (0.49-0.55 ->
//     0.55-0.59 of HBM; the cache-cold 4 M-item table 0.43 -> 0.51) because its 0.5 GB of one-touch rows no longer displace what
//     the forward pass left for BPTT and what the Infinity Cache holds of the table;
//   sparse update's parameter / optimiser-state rows (read, rewritten once per step): ON (step -0.5 %, the GEMM beside it -1.5 %);
//   forward gather: OFF (its rows are the score kernel's target rows one step later: cached, they are hits there);
//   BPTT's re-gather: neutral; the EWMA scans: OFF (the backward scan re-reads what the forward scan gathered: 1.36 -> 1.57 ms).
typedef float v4f_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4s(const float* p) {
    const v4f_nt v = __builtin_nontemporal_load(reinterpret_cast<const v4f_nt*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4s(float* p, float4 v) {
    v4f_nt t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<v4f_nt*>(p));
}
#ifndef SBR_NT_SCORE
#define SBR_NT_SCORE 1
#endif
#ifndef SBR_NT_UPD
#define SBR_NT_UPD 1
#endif
#ifndef SBR_NT_UPD_ST
#define SBR_NT_UPD_ST 1
#endif
/* cache-policy word of the raw buffer accesses of the recurrent kernels' per-row activations (2 = nt; build-time A/B hooks).
 * The forward pass writes 12 d floats per packed row — gates, cell states, the copy of the gathered input rows, h (1.5 GB at the
 * headline batch) — of which only h (131 MB) is read again soon: by the score kernel, next on the stream.  Gates, cell states and
 * the input copy streamed past the caches leave h in the 256 MB Infinity Cache: the score launch reads it there instead of from
 * HBM, 110 -> 92 us (0.56 -> 0.67 of the HBM roofline on its algorithmic bytes; cache-cold 4 M-item table 0.51 -> 0.62); the forward
 * pass pays ~1.5 % (0.675 -> 0.69 ms), the step is unchanged within noise.  BPTT's dz (0.5 GB, read once by the dense-gradient GEMM)
 * streamed: BPTT -1.4 %.  Streaming dx, BPTT's own reads or the GEMM's reads: nothing.  profiles/r05_streaming_gathers.md. */
#ifndef SBR_AUX_FWD_ST
#define SBR_AUX_FWD_ST 2 /* forward: gates, cell states, copy of the gathered input rows */
#endif
#ifndef SBR_AUX_FWD_G
#define SBR_AUX_FWD_G SBR_AUX_FWD_ST
#endif
#ifndef SBR_AUX_FWD_C
#define SBR_AUX_FWD_C SBR_AUX_FWD_ST
#endif
#ifndef SBR_AUX_FWD_X
#define SBR_AUX_FWD_X SBR_AUX_FWD_ST
#endif
#ifndef SBR_AUX_BWD_LD
#define SBR_AUX_BWD_LD 0 /* BPTT: its reads of the gates and cell states */
#endif
#ifndef SBR_AUX_BWD_ST
#define SBR_AUX_BWD_ST 0 /* BPTT: dx */
#endif
#ifndef SBR_AUX_BWD_Z
#define SBR_AUX_BWD_Z 2 /* BPTT: dz */
#endif
#ifndef SBR_AUX_DW_LD
#define SBR_AUX_DW_LD 0 /* dense-gradient GEMM: x / dz rows */
#endif
__device__ __forceinline__ float4 ld4_score(const float* p) { return SBR_NT_SCORE ? ld4s(p) : ld4(p); }
__device__ __forceinline__ float4 ld4_upd(const float* p) { return SBR_NT_UPD ? ld4s(p) : ld4(p); }
__device__ __forceinline__ void st4_upd(float* p, float4 v) { if (SBR_NT_UPD_ST) st4s(p, v); else st4(p, v); }

// ------------------------------------------------------------------------------------------------
// K1+K3+K4: gather + negative sampling + loss + dloss/dh  (the HBM-roofline kernel)
// One d/4-lane group per packed row; 16 B per lane per gathered embedding row.
// ------------------------------------------------------------------------------------------------
// U rows per lane group are in flight together (their retry rounds in lockstep).  U = 1 at large launches: the kernel is
// bandwidth-bound there and extra rows per group only add retry rounds (a group runs max-over-its-rows rounds; 0.72 -> 0.76 ms at
// 50 000 sequences per step with U = 2, round 2).  U = 2 at small launches (the quality-neutral 8 192 sequences per step =
// 256 K rows): a wave then walks ~18 passes of ~1.75 dependent memory round trips each and the kernel is THEIR LATENCY, not the
// bytes — half the passes with twice the rows in flight.  Same arithmetic per row, same bits.
// SPEC (the one-workgroup launch of a one-sequence step): all of a row's candidates are gathered together and then tested in
// order — the launch is a handful of dependent memory round trips and nothing else, and WARP's retries were up to four of them
// (~6 us of a 19 us launch); the extra rows are a few KB.  Same tests in the same order on the same values.
template <int D, int U, bool PF, bool SPEC = false, bool NT = false>
__device__ __forceinline__ void score_warp_rows(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w,
                                                uint64_t epoch_key, int wave, int nwaves, double* loss_out, unsigned int* tries_out) {
    constexpr int L = D / 4;
    constexpr int GPW = 64 / L;
    constexpr int RPW = GPW * U;  // rows per wave and pass
    const int lane = threadIdx.x & 63;
    const int lg = lane % L;
    const int grp = lane / L;
    const int max_tries = m.loss == SBR_LOSS_WARP ? SBR_WARP_MAX_TRIES : 1;
    auto ldrow = [](const float* p) { return NT ? ld4_score(p) : ld4(p); };  // nt: the launch's cache policy (launch_score)
    static_assert(SBR_WARP_MAX_TRIES <= 8, "one candidate per lane of an 8-lane group");
    const bool spread = L >= 8 && max_tries > 1;
    double loss_part = 0.0;   // reporting only: order-free f64 partial sums per workgroup
    unsigned int tries_part = 0;
    // Software pipeline over the passes of a wave: the ids, the counter and the h row of the NEXT pass are requested while this
    // pass's table rows are in flight, so a pass starts with its positive and first candidate rows already addressable — one
    // dependent memory round trip fewer per pass.
    const int last_row = mb.R - 1;
    int base = wave * RPW;
    uint32_t pi_n[U], ctr_n[U];
    float4 h_n[U];
    if constexpr (PF) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r0 = base + u * GPW + grp;
            const int rr0 = r0 < mb.R ? r0 : last_row;
            pi_n[u] = mb.out_idx[rr0];
            ctr_n[u] = mb.ctr[rr0];
            h_n[u] = ldrow(blk.H + (size_t)rr0 * D + 4 * lg);
        }
    }
    for (; base < mb.R; base += nwaves * RPW) {
        float4 h[U], ep[U], ec[U];
        uint32_t pi[U], ctr[U], draws[U], cand[U], nj[U], tries[U];
        float bp[U], bc[U], pos[U], neg[U];
        bool done[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (PF) {
                h[u] = h_n[u];
                pi[u] = pi_n[u];
                ctr[u] = ctr_n[u];
            } else {  // (no ids a pass ahead: their registers are what lets seven waves per SIMD be resident at U = 2)
                const int r0 = base + u * GPW + grp;
                const int rr0 = r0 < mb.R ? r0 : last_row;
                pi[u] = mb.out_idx[rr0];
                ctr[u] = mb.ctr[rr0];
                h[u] = ldrow(blk.H + (size_t)rr0 * D + 4 * lg);
            }
            ep[u] = ldrow(m.E + (size_t)pi[u] * D + 4 * lg);
            bp[u] = m.b[pi[u]];
            // WARP: lane lg of the group draws candidate lg & 7 — the five draws of the row cost one evaluation of the
            // 64-bit hash instead of one per try (every lane would compute the same value); try k reads lane k's
            draws[u] = sbr_neg_draw(epoch_key, ctr[u], spread ? (uint32_t)(lg & 7) : 0u, m.num_items);
            // first candidate is always scored: issue its gather together with the positive's
            cand[u] = spread ? (uint32_t)__shfl((int)draws[u], grp * L, 64) : draws[u];
            ec[u] = ldrow(m.E + (size_t)cand[u] * D + 4 * lg);
            bc[u] = m.b[cand[u]];
        }
        if constexpr (PF) {
#pragma unroll
            for (int u = 0; u < U; ++u) {  // the next pass's ids and h rows (clamped: the request count does not depend on the pass)
                const int rn = base + nwaves * RPW + u * GPW + grp;
                const int rrn = rn < mb.R ? rn : last_row;
                pi_n[u] = mb.out_idx[rrn];
                ctr_n[u] = mb.ctr[rrn];
                h_n[u] = ldrow(blk.H + (size_t)rrn * D + 4 * lg);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pos[u] = bp[u] + group_allreduce<L>(dot4(h[u], ep[u]));
            done[u] = false;
            nj[u] = 0;
            tries[u] = 0;
            neg[u] = 0.0f;
        }
        if constexpr (SPEC) {
            float4 ecs[U][SBR_WARP_MAX_TRIES];
            float bcs[U][SBR_WARP_MAX_TRIES];
            uint32_t cds[U][SBR_WARP_MAX_TRIES];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                cds[u][0] = cand[u];
                ecs[u][0] = ec[u];
                bcs[u][0] = bc[u];
#pragma unroll
                for (int k = 1; k < SBR_WARP_MAX_TRIES; ++k) {
                    cds[u][k] = spread ? (uint32_t)__shfl((int)draws[u], grp * L + k, 64) : sbr_neg_draw(epoch_key, ctr[u], (uint32_t)k, m.num_items);
                    ecs[u][k] = ldrow(m.E + (size_t)cds[u][k] * D + 4 * lg);
                    bcs[u][k] = m.b[cds[u][k]];
                }
            }
#pragma unroll
            for (int k = 0; k < SBR_WARP_MAX_TRIES; ++k) {
                if (k < max_tries) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float sc = bcs[u][k] + group_allreduce<L>(dot4(h[u], ecs[u][k]));
                        if (!done[u]) {
                            nj[u] = cds[u][k];
                            neg[u] = sc;
                            ++tries[u];
                            if (sbr_warp_violates(pos[u], sc)) done[u] = true;
                        }
                    }
                }
            }
        } else
        for (int k = 0; k < max_tries; ++k) {
            if (k > 0) {
                bool all_done = true;
#pragma unroll
                for (int u = 0; u < U; ++u) all_done = all_done && done[u];
                if (__all(all_done)) break;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    cand[u] = spread ? (uint32_t)__shfl((int)draws[u], grp * L + k, 64) : sbr_neg_draw(epoch_key, ctr[u], (uint32_t)k, m.num_items);
                    if (!done[u]) {
                        ec[u] = ldrow(m.E + (size_t)cand[u] * D + 4 * lg);
                        bc[u] = m.b[cand[u]];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float sc = bc[u] + group_allreduce<L>(dot4(h[u], ec[u]));
                if (!done[u]) {
                    nj[u] = cand[u];
                    neg[u] = sc;
                    ++tries[u];
                    if (sbr_warp_violates(pos[u], sc)) done[u] = true;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = base + u * GPW + grp;
            float g, l;
            if (m.loss == SBR_LOSS_BPR) l = sbr_loss_bpr(pos[u], neg[u], &g);
            else l = sbr_loss_hinge(pos[u], neg[u], &g);
            // dloss/dh = g (E[neg] - E[pos]) is not written here: the backward kernels re-form it
            // from (neg, coef) with two row gathers, which keeps this kernel read-only on the table
            if (r < mb.R && lg == 0) {
                blk.neg[r] = nj[u];
                blk.coef[r] = g;
                blk.in_idx[r] = mb.in_idx[r];
                blk.out_idx[r] = pi[u];
                w.loss[r] = l;
                w.tries[r] = tries[u];
                loss_part += (double)l;
                tries_part += tries[u];
            }
        }
    }
    *loss_out = loss_part;
    *tries_out = tries_part;
}

// the workgroup's share of the reported loss / tries (order-free sums) -> part_loss / part_tries [block]; thread 0 returns the sums
template <bool ALL>  // ALL: every thread gets the sums (the small tail follows); otherwise thread 0 only, as the store needs them
__device__ __forceinline__ void score_partials(const WorkView& w, int block, double loss_part, unsigned int tries_part, double* lsum,
                                               unsigned int* tsum) {
    __shared__ double s_loss[4];
    __shared__ unsigned int s_tries[4];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        loss_part += __shfl_xor(loss_part, off, 64);
        tries_part += __shfl_xor(tries_part, off, 64);
    }
    if (lane == 0) {
        s_loss[threadIdx.x >> 6] = loss_part;
        s_tries[threadIdx.x >> 6] = tries_part;
    }
    __syncthreads();
    if (ALL || threadIdx.x == 0) {
        *lsum = s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3];
        *tsum = s_tries[0] + s_tries[1] + s_tries[2] + s_tries[3];
    }
    if (threadIdx.x == 0) {
        w.part_loss[block] = *lsum;
        w.part_tries[block] = *tsum;
    }
}

// (seven waves per SIMD = 72 registers is what the d = 128 rows need in flight; below that a row is 64-256 B, the two-row form
// wants 73-80 registers, and a scratch reload inside the pass loop would wait for every gather outstanding: six waves there)
template <int D, int U, bool PF, bool NT>
__global__ __launch_bounds__(256, PF ? 1 : (D >= 128 ? 7 : 6)) void score_kernel(ModelView m, MbView mb, BlockView blk, WorkView w, uint64_t epoch_key) {
    double loss_part, lsum = 0.0;
    unsigned int tries_part, tsum = 0;
    score_warp_rows<D, U, PF, false, NT>(m, mb, blk, w, epoch_key, (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), (int)((gridDim.x * blockDim.x) >> 6),
                              &loss_part, &tries_part);
    score_partials<false>(w, blockIdx.x, loss_part, tries_part, &lsum, &tsum);
}
template <int D, int U, bool NT = false>
__device__ __forceinline__ void score_single_rows(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w,
                                                  uint64_t epoch_key, int wave, int nwaves, double* loss_out, unsigned int* tries_out);
#define SBR_SCORE_SINGLE_U 4 /* rows per lane group and pass of score_single_kernel */
template <int D>
__device__ __forceinline__ void score_single_rows_fwd(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w,
                                                      uint64_t epoch_key, int wave, int nwaves, double* loss_out, unsigned int* tries_out) {
    score_single_rows<D, SBR_SCORE_SINGLE_U>(m, mb, blk, w, epoch_key, wave, nwaves, loss_out, tries_out);
}
// one workgroup: the score pass of a one-sequence step, then its SmallTail
// (NT = 256 up to 64 rows; sixteen waves beyond: up to 128 rows are then ONE pass of the score loop and the ranking of the keys
// takes a quarter of the time — but a 1 024-thread workgroup costs ~3 us more to start and drain, which a ten-row step notices)
template <int D, int NT>
__device__ __forceinline__ void score_tail_body(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, uint64_t epoch_key,
                                                const SmallTail& tail) {
    constexpr int NW = NT / 64;
    double loss_part;
    unsigned int tries_part;
    if (m.loss == SBR_LOSS_WARP) score_warp_rows<D, 1, false, true>(m, mb, blk, w, epoch_key, (int)(threadIdx.x >> 6), NW, &loss_part, &tries_part);
    else score_single_rows_fwd<D>(m, mb, blk, w, epoch_key, (int)(threadIdx.x >> 6), NW, &loss_part, &tries_part);
    __shared__ double s_loss[NW];
    __shared__ unsigned int s_tries[NW];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        loss_part += __shfl_xor(loss_part, off, 64);
        tries_part += __shfl_xor(tries_part, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_loss[threadIdx.x >> 6] = loss_part;
        s_tries[threadIdx.x >> 6] = tries_part;
    }
    __syncthreads();
    double lsum = s_loss[0];
    unsigned int tsum = s_tries[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) { lsum += s_loss[i]; tsum += s_tries[i]; }
    if (threadIdx.x == 0) {
        w.part_loss[0] = lsum;
        w.part_tries[0] = tsum;
    }
    small_tail<NT>(mb, blk, w, tail, lsum, tsum);
}
template <int D, int NT>
__global__ __launch_bounds__(NT) void score_tail_kernel(ModelView m, MbView mb, BlockView blk, WorkView w, uint64_t epoch_key, SmallTail tail) {
    score_tail_body<D, NT>(m, mb, blk, w, epoch_key, tail);
}

// Single-negative losses (hinge, BPR: one candidate, no retry loop): U rows per lane group and pass, all their gathers
// in flight together, the ids of the next pass requested before this pass's rows (vector memory operations retire in
// order, so they have long arrived when the next pass starts).  Without the retry loop the pass is pure memory-level
// parallelism: 1.25 -> 1.1 ms at d = 256 against a 10 M-row table.  (With WARP's retry rounds in lockstep the same
// idea loses — more rounds per row than a two-row group needs — so WARP keeps score_kernel.)
template <int D, int U, bool NT>
__device__ __forceinline__ void score_single_rows(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w,
                                                  uint64_t epoch_key, int wave, int nwaves, double* loss_out, unsigned int* tries_out) {
    constexpr int L = D / 4;
    constexpr int GPW = 64 / L;
    constexpr int RPW = GPW * U;  // rows per wave and pass
    const int lane = threadIdx.x & 63;
    const int lg = lane % L;
    const int grp = lane / L;
    const int R = mb.R, last = mb.R - 1;
    const float* E = launder(m.E);
    const float* bias = launder(m.b);
    const float* Hh = launder(blk.H);
    const uint32_t* out_idx = launder(mb.out_idx);
    const uint32_t* in_idx = launder(mb.in_idx);
    const uint32_t* ctrs = launder(mb.ctr);
    double loss_part = 0.0;   // reporting only: order-free f64 partial sums per workgroup
    unsigned int tries_part = 0;
    uint32_t pi_next[U], ctr_next[U], in_next[U];
    auto request_ids = [&](int base) {  // rows past R are clamped to the last row (valid memory, results not stored)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = base + u * GPW + grp;
            const int rr = r < R ? r : last;
            pi_next[u] = out_idx[rr];
            ctr_next[u] = ctrs[rr];
            in_next[u] = in_idx[rr];
        }
    };
    int base = wave * RPW;
    if (base < R) request_ids(base);
    for (; base < R; base += nwaves * RPW) {
        uint32_t pi[U], cand[U], ini[U];
        float4 h[U], ep[U], ec[U];
        float bp[U], bc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pi[u] = pi_next[u];
            ini[u] = in_next[u];
            cand[u] = sbr_neg_draw(epoch_key, ctr_next[u], 0u, m.num_items);
        }
        request_ids(base + nwaves * RPW);  // clamped: the request count does not depend on the pass
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = base + u * GPW + grp;
            const int rr = r < R ? r : last;
            h[u] = (NT ? ld4s : ld4)(Hh + (size_t)rr * D + 4 * lg);
            ep[u] = (NT ? ld4s : ld4)(E + (size_t)pi[u] * D + 4 * lg);
            bp[u] = bias[pi[u]];
            ec[u] = (NT ? ld4s : ld4)(E + (size_t)cand[u] * D + 4 * lg);
            bc[u] = bias[cand[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = base + u * GPW + grp;
            const float pos = bp[u] + group_allreduce<L>(dot4(h[u], ep[u]));
            const float neg = bc[u] + group_allreduce<L>(dot4(h[u], ec[u]));
            float g, l;
            if (m.loss == SBR_LOSS_BPR) l = sbr_loss_bpr(pos, neg, &g);
            else l = sbr_loss_hinge(pos, neg, &g);
            if (r < R && lg == 0) {
                blk.neg[r] = cand[u];
                blk.coef[r] = g;
                blk.in_idx[r] = ini[u];
                blk.out_idx[r] = pi[u];
                w.loss[r] = l;
                w.tries[r] = 1u;
                loss_part += (double)l;
                tries_part += 1u;
            }
        }
    }
    *loss_out = loss_part;
    *tries_out = tries_part;
}
template <int D, int U, bool NT>
__global__ __launch_bounds__(256) void score_single_kernel(ModelView m, MbView mb, BlockView blk, WorkView w, uint64_t epoch_key) {
    double loss_part, lsum = 0.0;
    unsigned int tries_part, tsum = 0;
    score_single_rows<D, U, NT>(m, mb, blk, w, epoch_key, (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), (int)((gridDim.x * blockDim.x) >> 6),
                                &loss_part, &tries_part);
    score_partials<false>(w, blockIdx.x, loss_part, tries_part, &lsum, &tsum);
}

// dloss/dh of packed row r, elements u..u+3:  g*E[neg] - g*E[pos]  (two rounded products, one
// subtraction — the oracle's order)
__device__ __forceinline__ float4 dh_loss4(const ModelView& m, const BlockView& blk, size_t r, int u, int D) {
    const float g = blk.coef[r];
    const float4 en = ld4(m.E + (size_t)blk.neg[r] * D + u);
    const float4 ep = ld4(m.E + (size_t)blk.out_idx[r] * D + u);
    float4 dh;
    dh.x = g * en.x - g * ep.x;
    dh.y = g * en.y - g * ep.y;
    dh.z = g * en.z - g * ep.z;
    dh.w = g * en.w - g * ep.w;
    return dh;
}

template <int D>
__global__ __launch_bounds__(256) void materialize_dh_kernel(ModelView m, BlockView blk, int R, float* dH) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int r = idx / (D / 4);
    if (r >= R) return;
    const int u = (idx % (D / 4)) * 4;
    st4(dH + (size_t)r * D + u, dh_loss4(m, blk, (size_t)r, u, D));
}

// ------------------------------------------------------------------------------------------------
// K2': EWMA forward / backward scans — one d/4-lane group per sequence
// ------------------------------------------------------------------------------------------------
#ifndef EWMA_U
#define EWMA_U 4 /* time steps whose gathers are in flight together in the EWMA scans */
#endif
template <int D>
__global__ __launch_bounds__(256) void ewma_forward_kernel(ModelView m, MbView mb, float* H) {
    constexpr int L = D / 4;
    constexpr int GPW = 64 / L;
    const int lane = threadIdx.x & 63, lg = lane % L, grp = lane / L;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    float a[4], oma[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = sbr_sigmoidf(m.alpha[4 * lg + j]);
        oma[j] = 1.0f - a[j];
    }
    for (int b = wave * GPW + grp; b < mb.B; b += nwaves * GPW) {
        const int n = mb.steps[b];
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        // the gathers do not depend on the scan: the rows of EWMA_U steps are requested together (their indices first), then the
        // dependent fma chain runs over them — one sequence per step (the reference's schedule) would otherwise pay two
        // dependent memory round trips per time step
        for (int t0 = 0; t0 < n; t0 += EWMA_U) {
            int r[EWMA_U];
            uint32_t it[EWMA_U];
            float4 x[EWMA_U];
#pragma unroll
            for (int q = 0; q < EWMA_U; ++q) r[q] = mb.off[t0 + q < n ? t0 + q : n - 1] + b;
#pragma unroll
            for (int q = 0; q < EWMA_U; ++q) it[q] = mb.in_idx[r[q]];
#pragma unroll
            for (int q = 0; q < EWMA_U; ++q) x[q] = ld4(m.E + (size_t)it[q] * D + 4 * lg);
#pragma unroll
            for (int q = 0; q < EWMA_U; ++q) {
                if (t0 + q < n) {
                    if (t0 + q == 0) {
                        s = x[q];
                    } else {
                        s.x = sbr_fma(a[0], s.x, oma[0] * x[q].x);
                        s.y = sbr_fma(a[1], s.y, oma[1] * x[q].y);
                        s.z = sbr_fma(a[2], s.z, oma[2] * x[q].z);
                        s.w = sbr_fma(a[3], s.w, oma[3] * x[q].w);
                    }
                    st4(H + (size_t)r[q] * D + 4 * lg, s);
                }
            }
        }
    }
}

template <int D>
__device__ __forceinline__ void ewma_backward_seq(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, int b, int n,
                                                  int lg, const float (&a)[4], const float (&oma)[4]);
template <int D>
__global__ __launch_bounds__(256) void ewma_backward_kernel(ModelView m, MbView mb, BlockView blk, WorkView w) {
    constexpr int L = D / 4;
    constexpr int GPW = 64 / L;
    const int lane = threadIdx.x & 63, lg = lane % L, grp = lane / L;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    float a[4], oma[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = sbr_sigmoidf(m.alpha[4 * lg + j]);
        oma[j] = 1.0f - a[j];
    }
    for (int b = wave * GPW + grp; b < mb.B; b += nwaves * GPW) ewma_backward_seq<D>(m, mb, blk, w, b, mb.steps[b], lg, a, oma);
}

// ------------------------------------------------------------------------------------------------
// EWMA with a single-negative loss (hinge, BPR: BASELINE configs[4]) — ONE pass per sequence.
// The scan does not depend on the scores and the negative of a single-negative loss is a hash of the row counter, so the lane
// group that walks a sequence keeps s_t in registers, gathers x_t, the target row and the negative row of EWMA_U steps
// together, scores each step against the s_t it has just formed (ewma.rs:302-335: the scan node and the two dot nodes of a
// step) and writes H once — for the sparse update and the backward scan; the separate scan launch, its H round trip (write
// 4d B + read 4d B per row) and the score pass's own index traffic are gone.  The lane group then continues with the backward
// scan of the same sequence (ewma_backward_seq): its x / target / negative rows and its H rows were touched moments ago by this
// very group, newest first in the order the backward scan wants them — out of the Infinity Cache when few enough sequences are in
// flight (launch_ewma_sequences).
// Same arithmetic, same bits as ewma_forward_kernel + score_single_kernel + ewma_backward_kernel.
// ------------------------------------------------------------------------------------------------
// Packing invariant both scans rest on (sbr_engine.hip pack_sequences / the epoch packer: in_idx = item[t], out_idx = item[t + 1] of
// ONE item slice): the target of step t is the input of step t + 1, out_idx[row(t, b)] == in_idx[row(t + 1, b)].  A step's target
// row therefore IS the next step's input row: it is gathered once and handed on in registers (two table rows per step forward —
// target and negative — instead of three; input, negative and the h row backward instead of four).
// The ids of the NEXT batch of EWMA_U steps travel underneath the rows of the current one (forward, the offsets of the steps are
// scalar loads: every live lane of the wave is at the same time step), so a batch is ONE dependent round trip — its rows — where
// the loop used to pay three (offsets, ids, rows).
template <int D>
__device__ __forceinline__ void ewma_backward_seq(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, int b, int n,
                                                  int lg, const float (&a)[4], const float (&oma)[4]) {
    float carry[4] = {0.f, 0.f, 0.f, 0.f};
    float da[4] = {0.f, 0.f, 0.f, 0.f};
    auto step_off = [&](int t) { return mb.off[t < 0 ? 0 : t]; };  // (per lane: the groups of a wave start at their own n - 1)
    // ids of a batch: steps t0, t0 - 1, ... (q ascending = t descending); rows of steps below 0 are row b (step 0: valid memory)
    int r[EWMA_U], rp[EWMA_U];
    uint32_t ii[EWMA_U], ni[EWMA_U];
    float g[EWMA_U];
    auto request_ids = [&](int t0, int (&r_)[EWMA_U], int (&rp_)[EWMA_U], uint32_t (&ii_)[EWMA_U], uint32_t (&ni_)[EWMA_U], float (&g_)[EWMA_U]) {
#pragma unroll
        for (int q = 0; q < EWMA_U; ++q) {
            const int t = t0 - q;
            const int o = step_off(t), op = step_off(t - 1);
            r_[q] = t >= 0 ? o + b : b;
            rp_[q] = t > 0 ? op + b : b;
        }
#pragma unroll
        for (int q = 0; q < EWMA_U; ++q) {
            ii_[q] = mb.in_idx[r_[q]];
            ni_[q] = blk.neg[r_[q]];
            g_[q] = blk.coef[r_[q]];
        }
    };
    request_ids(n - 1, r, rp, ii, ni, g);
    // the target row of the LAST step is no step's input: gathered on its own
    float4 xnext = ld4(m.E + (size_t)blk.out_idx[step_off(n - 1) + b] * D + 4 * lg);
    for (int t0 = n - 1; t0 >= 0; t0 -= EWMA_U) {
        float4 x[EWMA_U], sp[EWMA_U], en[EWMA_U];
#pragma unroll
        for (int q = 0; q < EWMA_U; ++q) {
            x[q] = ld4(m.E + (size_t)ii[q] * D + 4 * lg);
            sp[q] = ld4(blk.H + (size_t)rp[q] * D + 4 * lg);
            en[q] = ld4(m.E + (size_t)ni[q] * D + 4 * lg);
        }
        int rn[EWMA_U], rpn[EWMA_U];
        uint32_t iin[EWMA_U], nin[EWMA_U];
        float gn[EWMA_U];
        request_ids(t0 - EWMA_U, rn, rpn, iin, nin, gn);
#pragma unroll
        for (int q = 0; q < EWMA_U; ++q) {
            const int t = t0 - q;
            if (t < 0) continue;
            const float4 ep = q == 0 ? xnext : x[q - 1];  // target of step t = input of step t + 1
            // dloss/dh: g*E[neg] - g*E[pos], two rounded products and one subtraction (dh_loss4)
            float ds[4] = {g[q] * en[q].x - g[q] * ep.x, g[q] * en[q].y - g[q] * ep.y, g[q] * en[q].z - g[q] * ep.z,
                           g[q] * en[q].w - g[q] * ep.w};
            if (t != n - 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ds[j] = ds[j] + carry[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) ds[j] = ds[j] + 0.0f;
            }
            float4 dx;
            if (t > 0) {
                const float xs[4] = {x[q].x, x[q].y, x[q].z, x[q].w};
                const float sps[4] = {sp[q].x, sp[q].y, sp[q].z, sp[q].w};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = oma[j] * ds[j];
                    carry[j] = a[j] * ds[j];
                    da[j] = sbr_fma(ds[j], sps[j] - xs[j], da[j]);
                }
                dx = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                dx = make_float4(ds[0], ds[1], ds[2], ds[3]);
            }
            st4(blk.dX + (size_t)r[q] * D + 4 * lg, dx);
        }
        xnext = x[EWMA_U - 1];
#pragma unroll
        for (int q = 0; q < EWMA_U; ++q) { r[q] = rn[q]; rp[q] = rpn[q]; ii[q] = iin[q]; ni[q] = nin[q]; g[q] = gn[q]; }
    }
    st4(w.dab + (size_t)b * D + 4 * lg, make_float4(da[0], da[1], da[2], da[3]));
}

// TAIL (one workgroup, a one-sequence step): the step's SmallTail follows in the same launch
template <int D, bool TAIL>
__device__ __forceinline__ void ewma_seq_body(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, uint64_t epoch_key,
                                              const SmallTail& tail, int wave, int nwaves, int part) {
    constexpr int L = D / 4;
    constexpr int GPW = 64 / L;
    const int lane = threadIdx.x & 63, lg = lane % L, grp = lane / L;
    float a[4], oma[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = sbr_sigmoidf(m.alpha[4 * lg + j]);
        oma[j] = 1.0f - a[j];
    }
    double loss_part = 0.0;   // reporting only: order-free f64 partial sums per workgroup
    unsigned int tries_part = 0;
    for (int b = wave * GPW + grp; b < mb.B; b += nwaves * GPW) {
        const int n = mb.steps[b];
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        auto step_off = [&](int t) {  // off[t], a scalar load (t is the same in every live lane); clamped to the array
            const int ts = __builtin_amdgcn_readfirstlane(t);
            return mb.off[ts > mb.Tm ? mb.Tm : ts];
        };
        // ids of a batch of EWMA_U steps (rows of steps past the sequence's end: row b, step 0's — valid memory, never used)
        // (the negative of a step is a 64-bit hash of its counter: lane q of the group evaluates step q's — one evaluation per batch
        // instead of EWMA_U, each the same value in every lane — and the others read it from there)
        static_assert(EWMA_U <= L, "one lane of the group per step of a batch");
        int r[EWMA_U];
        uint32_t pi[EWMA_U], ctr;
        auto request_ids = [&](int t0, int (&r_)[EWMA_U], uint32_t (&pi_)[EWMA_U], uint32_t* ctr_) {
#pragma unroll
            for (int q = 0; q < EWMA_U; ++q) {
                const int o = step_off(t0 + q);
                r_[q] = t0 + q < n ? o + b : b;
            }
            int rmine = r_[0];
#pragma unroll
            for (int q = 1; q < EWMA_U; ++q) rmine = (lg & (EWMA_U - 1)) == q ? r_[q] : rmine;
#pragma unroll
            for (int q = 0; q < EWMA_U; ++q) pi_[q] = mb.out_idx[r_[q]];
            *ctr_ = mb.ctr[rmine];
        };
        static_assert((EWMA_U & (EWMA_U - 1)) == 0, "lg & (EWMA_U - 1) picks the lane's step");
        request_ids(0, r, pi, &ctr);
        // the input of step 0 is no step's target: gathered on its own (see the packing invariant above ewma_backward_seq)
        uint32_t pprev = mb.in_idx[b];
        float4 xprev = ld4(m.E + (size_t)pprev * D + 4 * lg);
        for (int t0 = 0; t0 < n; t0 += EWMA_U) {
            uint32_t cand[EWMA_U];
            float4 ep[EWMA_U], ec[EWMA_U];
            float bp[EWMA_U], bc[EWMA_U];
            const uint32_t cmine = sbr_neg_draw(epoch_key, ctr, 0u, m.num_items);
#pragma unroll
            for (int q = 0; q < EWMA_U; ++q) {
                cand[q] = (uint32_t)__shfl((int)cmine, grp * L + q, 64);
                ep[q] = ld4(m.E + (size_t)pi[q] * D + 4 * lg);
                bp[q] = m.b[pi[q]];
                ec[q] = ld4(m.E + (size_t)cand[q] * D + 4 * lg);
                bc[q] = m.b[cand[q]];
            }
            int rn[EWMA_U];
            uint32_t pin[EWMA_U], ctrn;
            request_ids(t0 + EWMA_U, rn, pin, &ctrn);  // the next batch's ids, underneath this batch's rows
#pragma unroll
            for (int q = 0; q < EWMA_U; ++q) {
                if (t0 + q < n) {
                    const float4 x = q == 0 ? xprev : ep[q - 1];  // input of step t = target of step t - 1
                    if (t0 + q == 0) {
                        s = x;
                    } else {
                        s.x = sbr_fma(a[0], s.x, oma[0] * x.x);
                        s.y = sbr_fma(a[1], s.y, oma[1] * x.y);
                        s.z = sbr_fma(a[2], s.z, oma[2] * x.z);
                        s.w = sbr_fma(a[3], s.w, oma[3] * x.w);
                    }
                    st4(blk.H + (size_t)r[q] * D + 4 * lg, s);
                    const float pos = bp[q] + group_allreduce<L>(dot4(s, ep[q]));
                    const float neg = bc[q] + group_allreduce<L>(dot4(s, ec[q]));
                    float g, l;
                    if (m.loss == SBR_LOSS_BPR) l = sbr_loss_bpr(pos, neg, &g);
                    else l = sbr_loss_hinge(pos, neg, &g);
                    if (lg == 0) {
                        blk.neg[r[q]] = cand[q];
                        blk.coef[r[q]] = g;
                        blk.in_idx[r[q]] = q == 0 ? pprev : pi[q - 1];
                        blk.out_idx[r[q]] = pi[q];
                        w.loss[r[q]] = l;
                        w.tries[r[q]] = 1u;
                        loss_part += (double)l;
                        tries_part += 1u;
                    }
                }
            }
            xprev = ep[EWMA_U - 1];
            pprev = pi[EWMA_U - 1];
#pragma unroll
            for (int q = 0; q < EWMA_U; ++q) { r[q] = rn[q]; pi[q] = pin[q]; }
            ctr = ctrn;
        }
        /* lane 0's stores to blk.neg / coef / out_idx are read back by the whole group below */
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        ewma_backward_seq<D>(m, mb, blk, w, b, n, lg, a, oma);
    }
    __shared__ double s_loss[4];
    __shared__ unsigned int s_tries[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        loss_part += __shfl_xor(loss_part, off, 64);
        tries_part += __shfl_xor(tries_part, off, 64);
    }
    if (lane == 0) {
        s_loss[threadIdx.x >> 6] = loss_part;
        s_tries[threadIdx.x >> 6] = tries_part;
    }
    __syncthreads();
    if (TAIL || threadIdx.x == 0) {
        const double lsum = s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3];
        const unsigned int tsum = s_tries[0] + s_tries[1] + s_tries[2] + s_tries[3];
        if (threadIdx.x == 0) {
            w.part_loss[part] = lsum;
            w.part_tries[part] = tsum;
        }
        if constexpr (TAIL) small_tail<256>(mb, blk, w, tail, lsum, tsum);
    }
}
template <int D, bool TAIL = false>
__global__ __launch_bounds__(256) void ewma_seq_kernel(ModelView m, MbView mb, BlockView blk, WorkView w, uint64_t epoch_key, SmallTail tail, int active_blocks) {
    /* the grid is the score launch's (block_header_kernel adds that many loss partials); the sequences are walked by the first
     * active_blocks workgroups only — see launch_ewma_sequences — and the others leave a zero partial */
    if ((int)blockIdx.x >= active_blocks) {
        if (threadIdx.x == 0) { w.part_loss[blockIdx.x] = 0.0; w.part_tries[blockIdx.x] = 0u; }
        return;
    }
    ewma_seq_body<D, TAIL>(m, mb, blk, w, epoch_key, tail, (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6),
                                  (int)((active_blocks * blockDim.x) >> 6), (int)blockIdx.x);
}

// dalpha: chunk partials (chain over sequences inside a chunk), then chain across chunks
__global__ void ewma_dab_chunk_kernel(const float* dab, int B, int D, float* partials) {
    const int c = blockIdx.x;
    const int k = threadIdx.x;
    if (k >= D) return;
    int b0 = c * EWMA_CHUNK_SEQS, b1 = b0 + EWMA_CHUNK_SEQS;
    if (b1 > B) b1 = B;
    float pc = 0.0f;
    for (int b = b0; b < b1; ++b) pc = pc + dab[(size_t)b * D + k];
    partials[(size_t)c * D + k] = pc;
}
// both in one launch when all sequences fit one chunk (small steps: a launch is ~5 us of a ~60 us step)
__global__ void ewma_dab_final_kernel(const float* dab, int B, int D, const float* alpha, float* dense) {
    const int k = threadIdx.x;
    if (k >= D) return;
    float pc = 0.0f;
    for (int b = 0; b < B; ++b) pc = pc + dab[(size_t)b * D + k];
    const float a = sbr_sigmoidf(alpha[k]);
    dense[k] = pc * (a * (1.0f - a));
}
__global__ void ewma_dense_final_kernel(const float* partials, int nchunks, int D, const float* alpha, float* dense) {
    const int k = threadIdx.x;
    if (k >= D) return;
    float tot = 0.0f;
    for (int c = 0; c < nchunks; ++c) {
        const float pc = partials[(size_t)c * D + k];
        tot = c == 0 ? pc : tot + pc;
    }
    const float a = sbr_sigmoidf(alpha[k]);
    dense[k] = tot * (a * (1.0f - a));
}

// ------------------------------------------------------------------------------------------------
// K2: LSTM step, forward.  Grid = (32-row tiles of step t) x (16-unit tiles); one wave per gate.
// Wave g computes z_g = [x_t ; h_{t-1}] Wp_g + b_g for the workgroup's 32 rows x 16 units on
// v_mfma_f32_16x16x4_f32 (accumulator seeded with the bias, k ascending).  The A operand goes
// through LDS in two halves that share one buffer: the gathered x_t rows (k < D) first, the h_{t-1}
// rows (k >= D) — already in registers — second; the gate pre-activations then meet in the same
// buffer and the cell is applied by all threads.  Weight fragments (Wp packing, 16 B per lane per
// 4 k-steps) are requested a half ahead.  16.5 KiB of LDS and ~80 VGPRs per workgroup keep 6+
// workgroups resident per CU, so staging, MFMA and epilogue phases of different workgroups overlap.
// ------------------------------------------------------------------------------------------------
template <int D, int NG>
__global__ __launch_bounds__(NG * 64) void lstm_fwd_step_kernel(ModelView m, MbView mb, int t, float* H, WorkView w) {
    constexpr int LDA = D + 2;
    constexpr int NSH = D / 16;       // weight k-blocks (16 k each) per half
    constexpr int RT = 2;             // 16-row tiles per workgroup
    constexpr int ROWS = 16 * RT;
    constexpr int NT = NG * 64;
    constexpr int LDZ = NG * 16 + 1;
    constexpr int PF = NSH < 8 ? NSH : 8;
    constexpr int LDS_FLOATS = ROWS * (LDA > LDZ ? LDA : LDZ);  // A halves and the gate exchange share one buffer
    __shared__ float As[LDS_FLOATS];
    float* Zs = As;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j16 = lane & 15;
    const int kq = lane >> 4;
    const int ut = blockIdx.y;
    const int row_begin = mb.off[t];
    const int bt = mb.off[t + 1] - row_begin;
    const int b0 = blockIdx.x * ROWS;
    const int nrows = bt - b0 < ROWS ? bt - b0 : ROWS;
    const int prev_begin = t > 0 ? mb.off[t - 1] : 0;
    const float* wp = m.Wp + (((size_t)(ut * NG + g) * (2 * NSH)) * 64 + lane) * 4;
    float4 bf[PF];
#pragma unroll
    for (int S = 0; S < PF; ++S) bf[S] = ld4(wp + (size_t)S * 256);
    const float bias = m.bW[g * D + ut * 16 + j16];
    // both halves of the A tile are requested up front
    constexpr int NV = ROWS * (D / 4);
    constexpr int ITER = (NV + NT - 1) / NT;
    float4 xv[ITER], hv[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int idx = tid + it * NT;
        const int i = idx / (D / 4);
        const int c4 = (idx % (D / 4)) * 4;
        xv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        hv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < NV && i < nrows) {
            xv[it] = ld4(m.E + (size_t)mb.in_idx[row_begin + b0 + i] * D + c4);
            if (blockIdx.y == 0) st4(w.X + (size_t)(row_begin + b0 + i) * D + c4, xv[it]);  // copy for the dense-gradient GEMM
            if (t > 0) hv[it] = ld4(H + (size_t)(prev_begin + b0 + i) * D + c4);
        }
    }
    auto stage = [&](const float4* v) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int idx = tid + it * NT;
            if (idx < NV) {
                const int i = idx / (D / 4);
                const int c4 = (idx % (D / 4)) * 4;
                float2* dst = reinterpret_cast<float2*>(&As[i * LDA + c4]);
                dst[0] = make_float2(v[it].x, v[it].y);
                dst[1] = make_float2(v[it].z, v[it].w);
            }
        }
    };
    f32x4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){bias, bias, bias, bias};
    auto mma_half = [&](int half) {
#pragma unroll
        for (int S0 = 0; S0 < NSH; S0 += PF) {
            float4 cur[PF];
#pragma unroll
            for (int S = 0; S < PF; ++S) cur[S] = bf[S];
            // request the next group of weight k-blocks (possibly of the other half)
            const int nextS = half * NSH + S0 + PF;
            if (nextS < 2 * NSH) {
#pragma unroll
                for (int S = 0; S < PF; ++S) bf[S] = ld4(wp + (size_t)(nextS + S) * 256);
            }
#pragma unroll
            for (int S = 0; S < PF; ++S) {
                float av[RT][4];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const float* arow = &As[(rt * 16 + j16) * LDA + 16 * (S0 + S) + kq];
                    av[rt][0] = arow[0]; av[rt][1] = arow[4]; av[rt][2] = arow[8]; av[rt][3] = arow[12];
                }
                // independent accumulators alternate so that no MFMA waits on its predecessor
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][0], cur[S].x, acc[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][1], cur[S].y, acc[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][2], cur[S].z, acc[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][3], cur[S].w, acc[rt], 0, 0, 0);
            }
        }
    };
    stage(xv);
    __syncthreads();
    mma_half(0);
    __syncthreads();
    stage(hv);
    __syncthreads();
    mma_half(1);
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Zs[(rt * 16 + kq * 4 + reg) * LDZ + g * 16 + j16] = acc[rt][reg];
    __syncthreads();
    for (int e = tid; e < ROWS * 16; e += NT) {
        const int i = e >> 4;
        const int uu = e & 15;
        if (i < nrows) {
            const int u = ut * 16 + uu;
            const size_t r = (size_t)(row_begin + b0 + i);
            const float cprev = t > 0 ? w.C[(size_t)(prev_begin + b0 + i) * D + u] : 0.0f;
            const float* z = &Zs[i * LDZ + uu];
            float zi, zf, zg, zo;
            if (NG == 4) { zi = z[0]; zf = z[16]; zg = z[32]; zo = z[(NG - 1) * 16]; }
            else { zi = 0.0f; zf = z[0]; zg = z[16]; zo = z[32]; }
            float gi, gf, gg, go, cc, hh;
            sbr_lstm_cell_fwd(zi, zf, zg, zo, cprev, NG == 3, &gi, &gf, &gg, &go, &cc, &hh);
            float* G = w.G + r * 4 * D;
            G[u] = gi; G[D + u] = gf; G[2 * D + u] = gg; G[3 * D + u] = go;
            w.C[r * D + u] = cc;
            H[r * D + u] = hh;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2, sequence-resident form: one workgroup owns a tile of 32 sequences (rows b0..b0+31 of the
// length-sorted minibatch) for ALL its time steps — sequences are independent, so no grid-wide
// synchronisation is needed, h_{t-1} never leaves LDS and c_{t-1} never leaves registers.  Wave w
// owns unit tile w with all gates (cell is lane-local); per step it computes
// z = [x_t ; h_{t-1}] Wp + b on v_mfma_f32_16x16x4_f32 (bias-seeded accumulators, k ascending),
// applies the cell, writes G/C/H rows for BPTT and h_t into LDS.  The gather of x_{t+1} is in
// flight during the MFMAs of step t; weights stream from L2 in the Wp fragment packing.  Different
// workgroups drift out of phase over the steps, so one workgroup's gathers/epilogue overlap
// another's MFMAs (per-step launches start every workgroup in lockstep and serialise the phases).
// ------------------------------------------------------------------------------------------------

// Two LSTM cells at once on packed f32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two IEEE
// operations per issue slot).  The operation sequence per cell is exactly sbr_lstm_cell_fwd's / sbr_tanh_pq's
// (sbr_numerics.h, sbr_approx.h) — same bits; only min/max and the divisions stay scalar.  The cell epilogue is
// VALU time that the f32 MFMAs cannot hide, so halving its instruction count is worth doing by hand.
template <bool COUPLED>
__device__ __forceinline__ void lstm_cell_fwd_x2(v2f zi, v2f zf, v2f zg, v2f zo, v2f c_prev, v2f* gi, v2f* gf, v2f* gg, v2f* go,
                                                  v2f* c, v2f* h) {
    const v2f half = pk_splat(0.5f), one = pk_splat(1.0f);
    v2f pi = pk_splat(0.0f), qi = one, pf, qf, pg, qg, po, qo;
    if (!COUPLED) tanh_pq_x2(half * zi, &pi, &qi);
    tanh_pq_x2(half * zf, &pf, &qf);
    tanh_pq_x2(zg, &pg, &qg);
    tanh_pq_x2(half * zo, &po, &qo);
    const v2f q_if = qi * qf, q_go = qg * qo;
    const v2f den = q_if * q_go;
    const v2f r = (v2f){1.0f / den.x, 1.0f / den.y};
    const v2f r_if = r * q_go, r_go = r * q_if;
    const v2f f = pk_fma(half, pf * (r_if * qi), half);
    const v2f i = COUPLED ? one - f : pk_fma(half, pi * (r_if * qf), half);
    const v2f g = pg * (r_go * qo);
    const v2f o = pk_fma(half, po * (r_go * qg), half);
    const v2f cc = pk_fma(f, c_prev, i * g);
    v2f pc, qc;
    tanh_pq_x2(cc, &pc, &qc);
    const v2f tc = (v2f){pc.x / qc.x, pc.y / qc.y};
    *gi = i; *gf = f; *gg = g; *go = o; *c = cc;
    *h = o * tc;
}

template <int D, int NG, int RT, int UPW>
__global__ __launch_bounds__((D / 16 / UPW) * 64, UPW == 1 ? (RT >= 4 ? 2 : 4) : SBR_FWD_WPE) void lstm_fwd_seq_kernel(ModelView m, MbView mb, float* H, WorkView w, int ntiles) {
    constexpr int K2 = 2 * D;
    constexpr int LDA = K2 + 2;
    constexpr int NS = K2 / 16;
    constexpr int UT = D / 16;
    constexpr int NW = UT / UPW;      // waves; wave v owns unit tiles v*UPW .. v*UPW+UPW-1
    constexpr int NT = NW * 64;
    constexpr int ROWS = 16 * RT;
    constexpr int NV = ROWS * (D / 4);
    constexpr int ITER = (NV + NT - 1) / NT;
    __shared__ float As[ROWS * LDA];
    __shared__ int s_off[SBR_MAX_T + 2];
    const int tid0 = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const bool stream_act = w.stream_activations != 0;  // kernel argument: a scalar
    // Per-lane values derived from the thread id are re-derived where they are used (the id passes through an
    // empty asm), so that the compiler does not keep dozens of loop-invariant address pieces alive across the
    // time loop and spill them
    auto thread_id = [&]() { int t_ = tid0; asm volatile("" : "+v"(t_)); return t_; };
    int tid = tid0;
    int lane = tid & 63;
    int j16 = lane & 15;
    int kq = lane >> 4;
    // tiles are sorted by length; fold the list so that consecutive resident slots of a CU get
    // long/short/long/short ... and every CU ends up with about the same number of steps
    const int nslot = 256;
    const int q = (int)blockIdx.x / nslot, c = (int)blockIdx.x % nslot;
    const int fold_w = ntiles - q * nslot < nslot ? ntiles - q * nslot : nslot;
    const int tile = (q & 1) && ntiles <= w.fold_max_tiles ? q * nslot + (fold_w - 1 - c) : (int)blockIdx.x;
    const int b0 = tile * ROWS;
    const int nsteps = mb.steps[b0];
    // the row offsets of the steps are read from LDS inside the time loop (a global read there would be a
    // vector-memory operation queued behind the gathers and the stores)
    for (int idx = tid; idx <= nsteps; idx += NT) s_off[idx] = mb.off[idx];
    float bias[UPW][NG];
#pragma unroll
    for (int p = 0; p < UPW; ++p)
#pragma unroll
        for (int g = 0; g < NG; ++g) bias[p][g] = m.bW[g * D + (wv * UPW + p) * 16 + j16];
    float cst[RT][UPW][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int p = 0; p < UPW; ++p)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) cst[rt][p][reg] = 0.0f;
    for (int idx = tid; idx < ROWS * D; idx += NT) As[(idx / D) * LDA + D + (idx % D)] = 0.0f;  // h_{-1} = 0
    __syncthreads();
    auto step_rows = [&](int t, int* row_begin, int* nrows) {  // rows of this tile alive at step t
        const int rb = __builtin_amdgcn_readfirstlane(s_off[t]);
        int nr = __builtin_amdgcn_readfirstlane(s_off[t + 1]) - rb - b0;
        *row_begin = rb;
        *nrows = nr < ROWS ? nr : ROWS;
    };
    float4 xn[ITER];
    uint32_t xidx[ITER];  // item ids of the NEXT gather, requested one step before the gather that needs them
    // (the loaded ids are not touched until that gather is issued: any use right after the request would make
    // the wave wait for them, and with them for every vector-memory operation issued before)
    auto prefetch_idx = [&](int t) {  // step t's input item of this thread's rows (t < nsteps)
        int rb, nr;
        step_rows(t, &rb, &nr);
        const uint32_t* in_idx = launder(mb.in_idx);
#pragma unroll
        for (int it = 0; it < ITER; ++it) {  // branch-free: lanes of finished sequences read a valid position
            const int i = (tid + it * NT) / (D / 4);
            const bool live = tid + it * NT < NV && i < nr;
            xidx[it] = in_idx[live ? rb + b0 + i : rb];
        }
    };
    auto prefetch_x = [&]() {  // lanes of finished sequences gather a valid row too: their cells compute on, unobserved
        const float* E = launder(m.E);
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int c4 = ((tid + it * NT) % (D / 4)) * 4;
            xn[it] = ld4(E + ((size_t)xidx[it] * D + c4));
        }
    };
    int row_begin, nrows;
    step_rows(0, &row_begin, &nrows);
    // Vector memory operations of a wave retire IN ORDER (loads and stores share one counter), so a weight
    // fragment requested after an HBM gather or after the epilogue's stores cannot be consumed before those
    // have completed.  The weights do not change between time steps: the fragment ring therefore runs across
    // steps — the first PF k-blocks of step t+1 are requested at the end of step t's MFMA loop, AHEAD of the
    // gather of x_{t+1} and of the epilogue's stores, and the gather is issued after the last fragment request
    // of the step, with the cell epilogue and a barrier to land under.
    constexpr int PF = NS < 4 ? NS : (RT >= 4 ? 4 : 2);
    f32x4 ring[PF][UPW][NG];
    // Weight fragments come through a buffer resource: the address is {resource (scalar registers), one per-lane
    // byte offset that never changes, a scalar offset for (wave, gate, k-block)} — the request costs no vector ALU
    // work.  (As a global load with a 64-bit per-lane address it was 14 VALU instructions per k-block, on the
    // pipe the MFMAs run on.)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)m.Wp, (short)0, K2 * NG * D * 4, SBR_BUFFER_RSRC_FLAGS);
    const uint32_t lane_byte = (uint32_t)(tid0 & 63) * 16u;
    auto load_b = [&](f32x4 (*dst)[NG], int S) {
#pragma unroll
        for (int p = 0; p < UPW; ++p)
#pragma unroll
            for (int g = 0; g < NG; ++g)
                dst[p][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane_byte, ((((wv * UPW + p) * NG + g) * NS + S) * 1024), 0));
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) load_b(ring[j], j);
    prefetch_idx(0);
    prefetch_x();
    if (nsteps > 1) prefetch_idx(1);
#ifdef SBR_PROF_FWD
    long long prof[6] = {0, 0, 0, 0, 0, 0};
#define PROF_MARK(k) { const long long now_ = clock64(); prof[k] += now_ - prof_t; prof_t = now_; }
    long long prof_t = clock64();
#else
#define PROF_MARK(k)
#endif
    // x of the step (row_begin, nrows) describe goes from registers into LDS and into X (the copy of the
    // gathered input rows that the dense-gradient GEMM streams instead of re-gathering E, which lets that GEMM
    // run concurrently with the sparse update of E)
    // Stores of a step go through buffer resources that span exactly the tile's live rows of that step: base =
    // first row of the tile, num_records = live rows x row bytes.  Lanes of finished sequences fall outside the
    // range and the hardware drops their stores — branch-free without dump rows, and the address is {resource, an
    // invariant per-lane byte offset, a scalar / immediate offset}: no per-store vector ALU work (64-bit per-lane
    // addresses and live/dump selects were a fifth of the epilogue's VALU instructions).
    auto row_rsrc = [&](float* base, int row_floats, int first_row, int live_rows) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)first_row * row_floats), (short)0, live_rows * row_floats * 4,
                                                 SBR_BUFFER_RSRC_FLAGS);
    };
    auto stage = [&]() {
        tid = thread_id();
        const __amdgpu_buffer_rsrc_t rsX = row_rsrc(w.X, D, row_begin + b0, nrows);
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int idx = tid + it * NT;
            if (idx < NV) {
                const int i = idx / (D / 4);
                const int c4 = (idx % (D / 4)) * 4;
                float2* dst = reinterpret_cast<float2*>(&As[i * LDA + c4]);
                dst[0] = make_float2(xn[it].x, xn[it].y);
                dst[1] = make_float2(xn[it].z, xn[it].w);
                // row i, columns c4 .. c4+3 of the tile's rows = byte idx * 16
                if (stream_act) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, xn[it]), rsX, tid * 16, it * NT * 16, SBR_AUX_FWD_X);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, xn[it]), rsX, tid * 16, it * NT * 16, 0);
            }
        }
    };
    // One time step.  The body is instantiated twice — step 0 in straight-line code, steps >= 1 in the loop —
    // and the first PF k-blocks of the MFMA loop are peeled the same way: the compiler derives every
    // s_waitcnt from a static count of the memory operations issued since, and at a loop header it has to
    // assume the shortest history of all incoming paths (after the prologue: almost none), which would make
    // every iteration wait for the previous step's stores and gathers.
    auto step = [&](int t) {
        __syncthreads();  // x_t staged, h_{t-1} written by the previous epilogue
        PROF_MARK(1)
        __builtin_amdgcn_s_setprio(0);
        tid = thread_id();
        lane = tid & 63;
        j16 = lane & 15;
        kq = lane >> 4;
        const bool more = t + 1 < nsteps;
        f32x4 acc[RT][UPW][NG];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int p = 0; p < UPW; ++p)
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[rt][p][g] = (f32x4){bias[p][g], bias[p][g], bias[p][g], bias[p][g]};
        // the [x | h] operand of k-block S + 1 is read from LDS before the MFMAs of block S are issued (16-sequence tiles: the
        // registers are there; the 32-sequence form has none to spare and the other resident workgroup to cover the read)
        constexpr bool A_AHEAD = RT == 1;
        float av[RT][4];
        auto read_a = [&](int S) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float* arow = &As[(rt * 16 + j16) * LDA + 16 * (S < NS ? S : NS - 1) + kq];
                av[rt][0] = arow[0]; av[rt][1] = arow[4]; av[rt][2] = arow[8]; av[rt][3] = arow[12];
            }
        };
        if constexpr (A_AHEAD) read_a(0);
        auto mma_block = [&](int S, f32x4 (*bf)[NG]) {
            float ac[RT][4];
            if constexpr (A_AHEAD) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) ac[rt][sub] = av[rt][sub];
                read_a(S + 1);
            } else {
                read_a(S);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) ac[rt][sub] = av[rt][sub];
            }
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int p = 0; p < UPW; ++p)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[rt][p][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[rt][sub], bf[p][g][sub], acc[rt][p][g], 0, 0, 0);
        };
        // sched_barrier: the request that refills a ring slot is issued as soon as the slot has been consumed
        // (PF - 1 blocks of MFMAs ahead of its use), not sunk to the end of the loop body by the scheduler
        if constexpr (NS > PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                mma_block(j, ring[j]);
                __builtin_amdgcn_sched_barrier(0);
                load_b(ring[j], j + PF);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll 1
            for (int S0 = PF; S0 < NS - PF; S0 += PF) {
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    mma_block(S0 + j, ring[j]);
                    __builtin_amdgcn_sched_barrier(0);
                    load_b(ring[j], S0 + j + PF);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            mma_block(NS - PF + j, ring[j]);
            __builtin_amdgcn_sched_barrier(0);
            load_b(ring[j], j);  // step t+1's first blocks, ahead of the gather and of the epilogue's stores (unconditional:
                                 // a branch here would hide the request count from the wait-count analysis)
            __builtin_amdgcn_sched_barrier(0);
        }
        PROF_MARK(2)
        int row_begin_next = 0, nrows_next = 0;
        if (more) step_rows(t + 1, &row_begin_next, &nrows_next);
        auto request_next_x = [&]() {
            if (more) {
                tid = thread_id();
                prefetch_x();
                if (t + 2 < nsteps) prefetch_idx(t + 2);
            }
        };
        // 256-register form: x_{t+1} is gathered here and lands under the epilogue.  128-register form (32-sequence tiles):
        // after the epilogue (its registers would otherwise be spilled across it, and a spill of a just-requested row is a
        // wait for the gather); the other resident workgroup covers the latency.  16-sequence tiles hold ONE 16-byte piece
        // of x per thread and run where a CU has one or two workgroups and nobody to cover a gather: requested here too.
        constexpr bool EARLY_X = RT >= 4 || RT == 1;
        if constexpr (EARLY_X) request_next_x();
        PROF_MARK(3)
        __syncthreads();  // every wave is done reading As
        PROF_MARK(4)
        // the cell epilogue at a raised wave priority: it is the short, latency-sensitive phase between two MFMA phases, and
        // the other resident workgroup's MFMA stream otherwise wins the issue arbitration against it (forward -1.8 %; the
        // same in BPTT starves the key sort that runs beside it, so only here)
        __builtin_amdgcn_s_setprio(3);
        tid = thread_id();
        lane = tid & 63;
        j16 = lane & 15;
        kq = lane >> 4;
        {
            const __amdgpu_buffer_rsrc_t rsG = row_rsrc(w.G, 4 * D, row_begin + b0, nrows);
            const __amdgpu_buffer_rsrc_t rsC = row_rsrc(w.C, D, row_begin + b0, nrows);
            const __amdgpu_buffer_rsrc_t rsH = row_rsrc(H, D, row_begin + b0, nrows);
            const int vG = (kq * 4 * 4 * D + wv * UPW * 16 + j16) * 4;  // byte offsets of (row kq*4, unit wv*UPW*16 + j16)
            const int vC = (kq * 4 * D + wv * UPW * 16 + j16) * 4;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int p = 0; p < UPW; ++p) {
                    const int u = (wv * UPW + p) * 16 + j16;
#pragma unroll
                    for (int rp = 0; rp < 4; rp += 2) {  // two cells per pass (packed f32 arithmetic)
                        v2f zi2, zf2, zg2, zo2;
                        if (NG == 4) {
                            zi2 = (v2f){acc[rt][p][0][rp], acc[rt][p][0][rp + 1]};
                            zf2 = (v2f){acc[rt][p][1][rp], acc[rt][p][1][rp + 1]};
                            zg2 = (v2f){acc[rt][p][2][rp], acc[rt][p][2][rp + 1]};
                            zo2 = (v2f){acc[rt][p][NG - 1][rp], acc[rt][p][NG - 1][rp + 1]};
                        } else {
                            zi2 = pk_splat(0.0f);
                            zf2 = (v2f){acc[rt][p][0][rp], acc[rt][p][0][rp + 1]};
                            zg2 = (v2f){acc[rt][p][1][rp], acc[rt][p][1][rp + 1]};
                            zo2 = (v2f){acc[rt][p][2][rp], acc[rt][p][2][rp + 1]};
                        }
                        v2f gi2, gf2, gg2, go2, cc2, hh2;
                        lstm_cell_fwd_x2<NG == 3>(zi2, zf2, zg2, zo2, (v2f){cst[rt][p][rp], cst[rt][p][rp + 1]}, &gi2, &gf2, &gg2, &go2, &cc2, &hh2);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int reg = rp + e;
                            const int i = rt * 16 + kq * 4 + reg;
                            const float gi = gi2[e], gf = gf2[e], gg = gg2[e], go = go2[e], cc = cc2[e], hh = hh2[e];
                            cst[rt][p][reg] = cc;
                            // row i = rt*16 + kq*4 + reg: (rt, reg, p) go into the scalar offset, the gate into the immediate
                            const int sG = ((rt * 16 + reg) * 4 * D + p * 16) * 4, sC = ((rt * 16 + reg) * D + p * 16) * 4;
                            if (stream_act) {  // (wave-uniform: the launch's cache policy, sbr_kernels.h WorkView::stream_activations)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gi), rsG, vG, sG, SBR_AUX_FWD_G);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gf), rsG, vG + D * 4, sG, SBR_AUX_FWD_G);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gg), rsG, vG + 2 * D * 4, sG, SBR_AUX_FWD_G);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, go), rsG, vG + 3 * D * 4, sG, SBR_AUX_FWD_G);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cc), rsC, vC, sC, SBR_AUX_FWD_C);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gi), rsG, vG, sG, 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gf), rsG, vG + D * 4, sG, 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gg), rsG, vG + 2 * D * 4, sG, 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, go), rsG, vG + 3 * D * 4, sG, 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cc), rsC, vC, sC, 0);
                            }
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hh), rsH, vC, sC, 0);
                            As[i * LDA + D + u] = hh;
                        }
                        // 128-register form: one cell pair at a time (interleaving the pairs for instruction-level parallelism
                        // is what pushed this form into scratch)
                        if constexpr (RT < 4) __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
        PROF_MARK(5)
        if constexpr (!EARLY_X) request_next_x();
        row_begin = row_begin_next;
        nrows = nrows_next;
        if (more) stage();  // the x region of As is free since the barrier above
        PROF_MARK(0)
    };
    stage();
    step(0);
#pragma unroll 1
    for (int t = 1; t < nsteps; ++t) step(t);
#ifdef SBR_PROF_FWD
    if ((tile == 40 || tile == 700 || tile == 1400) && lane == 0 && (wv == 0 || wv == 5))
        printf("FWDPROF tile %d wave %d steps %d per-step cycles: stage(next) %lld bar1 %lld mfma %lld gather-issue %lld bar2 %lld epilogue %lld\n", tile, wv,
               nsteps, prof[0] / nsteps, prof[1] / nsteps, prof[2] / nsteps, prof[3] / nsteps, prof[4] / nsteps, prof[5] / nsteps);
#endif
}

// ------------------------------------------------------------------------------------------------
// K5: LSTM step, backward = two launches per step.
// (a) cell backward, elementwise over (row of step t, 4 units): dz_t and the dc carry.
// (b) dxh = dz_t * W^T on MFMA (accumulator from 0, j ascending): columns < D are dX of the row,
//     columns >= D the recurrent dh consumed by (a) of step t-1.
// ------------------------------------------------------------------------------------------------
template <int D, int NG>
__global__ __launch_bounds__(256) void lstm_bwd_cell_kernel(ModelView m, MbView mb, int t, BlockView blk, WorkView w) {
    constexpr int NGD = NG * D;
    constexpr int Q = D / 4;
    const int row_begin = mb.off[t];
    const int bt = mb.off[t + 1] - row_begin;
    const int bnext = t + 1 < mb.Tm ? mb.off[t + 2] - mb.off[t + 1] : 0;
    const int prev_begin = t > 0 ? mb.off[t - 1] : 0;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int b = idx / Q;
    if (b >= bt) return;
    const int u = (idx % Q) * 4;
    const size_t r = (size_t)(row_begin + b);
    const bool last = b >= bnext;
    const float4 dh0 = dh_loss4(m, blk, r, u, D);
    const float* G = w.G + r * 4 * D;
    const float4 gi = ld4(G + u), gf = ld4(G + D + u), gg = ld4(G + 2 * D + u), go = ld4(G + 3 * D + u);
    const float4 cc = ld4(w.C + r * D + u);
    float4 cp = make_float4(0.f, 0.f, 0.f, 0.f), dhr = cp, dcr = cp;
    if (t > 0) cp = ld4(w.C + (size_t)(prev_begin + b) * D + u);
    if (!last) {
        dhr = ld4(w.dHrec + (size_t)b * D + u);
        dcr = ld4(w.dCrec + (size_t)b * D + u);
    }
    const float dh_[4] = {dh0.x, dh0.y, dh0.z, dh0.w}, dhr_[4] = {dhr.x, dhr.y, dhr.z, dhr.w}, dcr_[4] = {dcr.x, dcr.y, dcr.z, dcr.w};
    const float gi_[4] = {gi.x, gi.y, gi.z, gi.w}, gf_[4] = {gf.x, gf.y, gf.z, gf.w}, gg_[4] = {gg.x, gg.y, gg.z, gg.w}, go_[4] = {go.x, go.y, go.z, go.w};
    const float cc_[4] = {cc.x, cc.y, cc.z, cc.w}, cp_[4] = {cp.x, cp.y, cp.z, cp.w};
    float dz[4][4], dco[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dh = dh_[j] + (last ? 0.0f : dhr_[j]);
        sbr_lstm_cell_bwd(dh, last ? 0.0f : dcr_[j], gi_[j], gf_[j], gg_[j], go_[j], cc_[j], cp_[j], NG == 3, &dz[0][j],
                          &dz[1][j], &dz[2][j], &dz[3][j], &dco[j]);
    }
    st4(w.dCrec + (size_t)b * D + u, make_float4(dco[0], dco[1], dco[2], dco[3]));
    float* dZ = w.dZ + r * NGD;
    if (NG == 4) {
        st4(dZ + u, make_float4(dz[0][0], dz[0][1], dz[0][2], dz[0][3]));
        st4(dZ + D + u, make_float4(dz[1][0], dz[1][1], dz[1][2], dz[1][3]));
        st4(dZ + 2 * D + u, make_float4(dz[2][0], dz[2][1], dz[2][2], dz[2][3]));
        st4(dZ + 3 * D + u, make_float4(dz[3][0], dz[3][1], dz[3][2], dz[3][3]));
    } else {
        st4(dZ + u, make_float4(dz[1][0], dz[1][1], dz[1][2], dz[1][3]));
        st4(dZ + D + u, make_float4(dz[2][0], dz[2][1], dz[2][2], dz[2][3]));
        st4(dZ + 2 * D + u, make_float4(dz[3][0], dz[3][1], dz[3][2], dz[3][3]));
    }
}

template <int D, int NG>
struct BwdCfg {
    static constexpr int NGD = NG * D;
    static constexpr int kparts() {
        for (int kp = 1; kp <= 8; ++kp)
            if (NGD % kp == 0 && NGD / kp <= 256 && (NGD / kp) % 16 == 0) return kp;
        return 1;
    }
    static constexpr int KP = kparts();  // the dz tile goes through LDS in KP column parts
    static constexpr int PW = NGD / KP;
    static constexpr int RT = 2;
};

template <int D, int NG>
__global__ __launch_bounds__(256) void lstm_bwd_gemm_kernel(ModelView m, MbView mb, int t, BlockView blk, WorkView w) {
    constexpr int K2 = 2 * D;
    constexpr int NGD = NG * D;
    constexpr int KP = BwdCfg<D, NG>::KP;
    constexpr int PW = BwdCfg<D, NG>::PW;
    constexpr int LDZ = PW + 2;
    constexpr int NSP = PW / 16;  // weight k-blocks per part
    constexpr int RT = BwdCfg<D, NG>::RT;
    constexpr int ROWS = 16 * RT;
    constexpr int CTW = K2 / 16 < 4 ? K2 / 16 : 4;  // column tiles (= busy waves) per workgroup
    constexpr int PF = NSP % 8 == 0 ? 8 : (NSP % 6 == 0 ? 6 : (NSP % 4 == 0 ? 4 : (NSP % 3 == 0 ? 3 : (NSP % 2 == 0 ? 2 : 1))));
    __shared__ float Zs[ROWS * LDZ];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15;
    const int kq = lane >> 4;
    const int row_begin = mb.off[t];
    const int bt = mb.off[t + 1] - row_begin;
    const int b0 = blockIdx.x * ROWS;
    const int nrows = bt - b0 < ROWS ? bt - b0 : ROWS;
    const int ct = blockIdx.y * CTW + wave;
    const bool busy = wave < CTW;
    const float* wp = m.WTp + ((size_t)(busy ? ct : 0) * (NGD / 16) * 64 + lane) * 4;
    float4 bf[PF];
#pragma unroll
    for (int S = 0; S < PF; ++S) bf[S] = ld4(wp + (size_t)S * 256);
    constexpr int NV = ROWS * (PW / 4);
    constexpr int ITER = (NV + 255) / 256;
    float4 zv[KP][ITER];
#pragma unroll
    for (int p = 0; p < KP; ++p)
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int idx = tid + it * 256;
            const int i = idx / (PW / 4);
            const int c4 = (idx % (PW / 4)) * 4;
            zv[p][it] = (idx < NV && i < nrows) ? ld4(w.dZ + (size_t)(row_begin + b0 + i) * NGD + p * PW + c4)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    f32x4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        if (p > 0) __syncthreads();
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int idx = tid + it * 256;
            if (idx < NV) {
                const int i = idx / (PW / 4);
                const int c4 = (idx % (PW / 4)) * 4;
                float2* dst = reinterpret_cast<float2*>(&Zs[i * LDZ + c4]);
                dst[0] = make_float2(zv[p][it].x, zv[p][it].y);
                dst[1] = make_float2(zv[p][it].z, zv[p][it].w);
            }
        }
        __syncthreads();
        if (busy) {
#pragma unroll
            for (int S0 = 0; S0 < NSP; S0 += PF) {
                float4 cur[PF];
#pragma unroll
                for (int S = 0; S < PF; ++S) cur[S] = bf[S];
                const int nextS = p * NSP + S0 + PF;
                if (nextS < NGD / 16) {
#pragma unroll
                    for (int S = 0; S < PF; ++S) bf[S] = ld4(wp + (size_t)(nextS + S) * 256);
                }
#pragma unroll
                for (int S = 0; S < PF; ++S) {
                    float av[RT][4];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const float* arow = &Zs[(rt * 16 + c16) * LDZ + 16 * (S0 + S) + kq];
                        av[rt][0] = arow[0]; av[rt][1] = arow[4]; av[rt][2] = arow[8]; av[rt][3] = arow[12];
                    }
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][0], cur[S].x, acc[rt], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][1], cur[S].y, acc[rt], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][2], cur[S].z, acc[rt], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][3], cur[S].w, acc[rt], 0, 0, 0);
                }
            }
        }
    }
    if (!busy) return;
    const int col = ct * 16 + c16;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = rt * 16 + kq * 4 + reg;
            if (i < nrows) {
                const int b = b0 + i;
                if (col < D) blk.dX[(size_t)(row_begin + b) * D + col] = acc[rt][reg];
                else w.dHrec[(size_t)b * D + (col - D)] = acc[rt][reg];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// K5, sequence-resident form: one workgroup owns a tile of 16*RT sequences for ALL its time steps, walking t
// downwards.  Per step: (a) cell backward for the tile — every thread owns RT (row, 4-unit) items, keeps
// their dc carry in registers, re-forms dloss/dh from (neg, coef) with two row gathers, reads the recurrent
// dh from LDS — writes dz into LDS (GEMM operand) and HBM (dense-gradient GEMM); (b) dxh = dz * W^T on
// v_mfma_f32_16x16x4_f32 (accumulators from 0, j ascending), wave w producing column tiles w (dX of the row,
// stored) and UT+w (recurrent dh for step t-1, written into columns [0, D) of the dz tile once every wave
// has finished reading it — the two never live at the same time, so they share LDS).
// Memory operations of a wave retire in order, so the order of issue is part of the design (as in the
// forward kernel): the weight-fragment ring runs across steps; the row gathers of step t-1 (item ids
// requested one step earlier) are issued behind the first ring blocks of step t's GEMM and land under it;
// the gate / cell-state rows of step t-1 are requested after the GEMM's last fragment request and land
// under the epilogue; loaded values are not touched before the phase that needs them; stores are
// branch-free (lanes of finished sequences fall outside the step's buffer ranges) and step ts = nsteps-1 and the first ring
// blocks are peeled, so that every wait count is static and exact.
// ------------------------------------------------------------------------------------------------
template <int D, int NG, int RT>
__global__ __launch_bounds__((D / 16) * 64, RT >= 4 ? 2 : 4) void lstm_bwd_seq_kernel(ModelView m, MbView mb, BlockView blk, WorkView w) {
    constexpr int NGD = NG * D;
    constexpr int LDZ = NGD + 2;
    constexpr int NSZ = NGD / 16;
    constexpr int UT = D / 16;
    constexpr int NT = UT * 64;
    constexpr int ROWS = 16 * RT;
    constexpr int Q = D / 4;
    constexpr int CITER = (ROWS * Q) / NT;  // = RT
    // ring depth (k-blocks in flight per column tile).  16-sequence tiles: a k-block is 8 MFMAs = 256 pipe cycles, so two blocks
    // ahead are less than a round trip to L2 and the GEMM phase ran at 60 % of the pipe (27 000 cycles for 256 MFMAs per wave,
    // two waves per SIMD; the forward kernel's 16-MFMA blocks: 86 %) — four there
#ifndef SBR_BWD_PF_RT1
#define SBR_BWD_PF_RT1 4
#endif
    constexpr int PF = RT >= 4 ? 4 : (RT == 1 && NSZ % SBR_BWD_PF_RT1 == 0 && NSZ >= 2 * SBR_BWD_PF_RT1 ? SBR_BWD_PF_RT1 : (NSZ % 2 == 0 ? 2 : 1));
    static_assert(NSZ >= 2 * PF && NSZ % PF == 0, "ring does not tile the GEMM");
    __shared__ float Zs[ROWS * LDZ];
    __shared__ int s_off[SBR_MAX_T + 2];
    const int tid0 = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    auto thread_id = [&]() { int t_ = tid0; asm volatile("" : "+v"(t_)); return t_; };
    // tiles are sorted by length; the list is folded as in the forward kernel, so that consecutive resident slots of a CU get
    // long/short/long/short ... (without it 512 tiles on 256 CUs gave CU 0 the two longest tiles of the two halves — 93 steps —
    // and CU 255 the two shortest — 35: BPTT 0.91 ms against the forward pass's 0.68 at 8 192 sequences per step)
    const int nslot = 256, ntiles = (int)gridDim.x;
    const int fq = (int)blockIdx.x / nslot, fc = (int)blockIdx.x % nslot;
    const int fold_w = ntiles - fq * nslot < nslot ? ntiles - fq * nslot : nslot;
    const int tile = (fq & 1) && ntiles <= w.fold_max_tiles ? fq * nslot + (fold_w - 1 - fc) : (int)blockIdx.x;
    const int b0 = tile * ROWS;
    const int nsteps = mb.steps[b0];
    for (int idx = tid0; idx <= nsteps; idx += NT) s_off[idx] = mb.off[idx];
    for (int idx = tid0; idx < ROWS * D; idx += NT) Zs[(idx / D) * LDZ + (idx % D)] = 0.0f;  // no recurrent dh yet
    __syncthreads();
    auto step_rows = [&](int t, int* row_begin, int* nrows) {  // rows of this tile alive at step t
        const int rb = __builtin_amdgcn_readfirstlane(s_off[t]);
        int nr = __builtin_amdgcn_readfirstlane(s_off[t + 1]) - rb - b0;
        *row_begin = rb;
        *nrows = nr < ROWS ? nr : ROWS;
    };
    float dc[CITER][4];
#pragma unroll
    for (int k = 0; k < CITER; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) dc[k][j] = 0.0f;
    // in flight between phases (raw: nothing below is touched before the cell phase that consumes it)
    float4 pg[CITER][4], pc[CITER], pcp[CITER], pen[CITER], pep[CITER];
    float pcoef[CITER];
    uint32_t nidx[CITER], oidx[CITER];
    // Row-addressed arrays are read and written through buffer resources that span the tile's live rows of the step
    // (base = first row, num_records = live rows x row bytes): reads of finished sequences' lanes return 0, their
    // stores are dropped, and the address is {resource, invariant per-lane byte offset, scalar / immediate offset} —
    // no per-access vector ALU work (see the forward kernel).
    auto row_rsrc = [&](const void* base, int row_bytes, int first_row, int live_rows) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(base) + (size_t)first_row * row_bytes), (short)0,
                                                 live_rows * row_bytes, SBR_BUFFER_RSRC_FLAGS);
    };
    // item k of this thread = (row i, units u .. u+3) with idx = tid + k*NT, i = idx / Q, u = (idx % Q) * 4
    static_assert(NT % Q == 0, "item k of a thread is NT / Q rows below item 0, same units");
    auto request_ids = [&](int t) {  // negatives / targets of step t (0 for finished sequences: a valid row of E)
        const int tid = thread_id();
        int rb, nr;
        step_rows(t, &rb, &nr);
        const __amdgpu_buffer_rsrc_t rsN = row_rsrc(blk.neg, 4, rb + b0, nr), rsO = row_rsrc(blk.out_idx, 4, rb + b0, nr);
        const int vo = (tid / Q) * 4;  // item 0's row; item k is NT / Q rows further
#pragma unroll
        for (int k = 0; k < CITER; ++k) {
            nidx[k] = __builtin_amdgcn_raw_buffer_load_b32(rsN, vo, k * (NT / Q) * 4, 0);
            oidx[k] = __builtin_amdgcn_raw_buffer_load_b32(rsO, vo, k * (NT / Q) * 4, 0);
        }
    };
    auto request_gathers = [&](int t) {  // E[neg], E[target], coef of step t (ids requested a step earlier)
        const int tid = thread_id();
        int rb, nr;
        step_rows(t, &rb, &nr);
        const float* E = launder(m.E);
        const __amdgpu_buffer_rsrc_t rsK = row_rsrc(blk.coef, 4, rb + b0, nr);
#pragma unroll
        for (int k = 0; k < CITER; ++k) {
            const int u = ((tid + k * NT) % Q) * 4;
            pen[k] = ld4(E + ((size_t)nidx[k] * D + u));
            pep[k] = ld4(E + ((size_t)oidx[k] * D + u));
            pcoef[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsK, (tid / Q) * 4, k * (NT / Q) * 4, 0));
        }
    };
    // cell state rows: the cell phase of step t needs c_t (pc) and c_{t-1} (pcp); c_{t-1} is step t-1's c_t, so one
    // row per step is requested and the pair rotates
    auto request_cstate = [&](float4* dst, int t) {  // c of every sequence of the tile alive at step t (t >= 0)
        const int tid = thread_id();
        int rb, nr_live;
        step_rows(t, &rb, &nr_live);
        const __amdgpu_buffer_rsrc_t rsC = row_rsrc(w.C, D * 4, rb + b0, nr_live);
#pragma unroll
        for (int k = 0; k < CITER; ++k)  // row i, units u..u+3 of a D-wide row = byte idx * 16
            dst[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsC, tid * 16, k * NT * 16, SBR_AUX_BWD_LD));
    };
    auto request_gates = [&](int t) {  // gate values of step t
        const int tid = thread_id();
        int rb, nr;
        step_rows(t, &rb, &nr);
        const __amdgpu_buffer_rsrc_t rsG = row_rsrc(w.G, 4 * D * 4, rb + b0, nr);
        const int vg = (tid / Q) * (4 * D * 4) + (tid % Q) * 16;  // item 0; item k is k * NT / Q rows further (NT is a multiple of Q)
#pragma unroll
        for (int k = 0; k < CITER; ++k) {
            const int so = k * (NT / Q) * (4 * D * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pg[k][g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsG, vg + g * D * 4, so, SBR_AUX_BWD_LD));
        }
    };

    f32x4 ring0[PF], ring1[PF];
    // buffer resource + invariant per-lane byte offset + scalar offset: no vector ALU work per request (see the forward kernel)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)m.WTp, (short)0, 2 * D * NGD * 4, SBR_BUFFER_RSRC_FLAGS);
    const uint32_t lane_byte = (uint32_t)(tid0 & 63) * 16u;
    auto load_ring = [&](int slot, int S) {
        ring0[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane_byte, (wv * NSZ + S) * 1024, 0));
        ring1[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane_byte, ((UT + wv) * NSZ + S) * 1024, 0));
    };
    // ---- prologue: everything step ts = nsteps-1 needs
    const int ts = nsteps - 1;
    request_ids(ts);
    request_gates(ts);
    request_cstate(pc, ts);
    request_cstate(pcp, ts > 0 ? ts - 1 : 0);
    request_gathers(ts);
    if (ts > 0) request_ids(ts - 1);
#pragma unroll
    for (int j = 0; j < PF; ++j) load_ring(j, j);

#ifdef SBR_PROF_BWD
    long long bprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define BPROF(k_) { const long long now_ = clock64(); bprof[k_] += now_ - bprof_t; bprof_t = now_; }
    long long bprof_t = clock64();
#else
#define BPROF(k_)
#endif
    auto step = [&](int t) {
        int row_begin, nrows, nrows_above = 0;
        step_rows(t, &row_begin, &nrows);
        if (t + 1 < nsteps) { int rb_; step_rows(t + 1, &rb_, &nrows_above); }
        BPROF(0)
        __syncthreads();  // A: the recurrent dh of step t+1 is in Zs[:, 0:D)
        BPROF(1)
        // (a) cell backward
        {
            const int tid = thread_id();
            const __amdgpu_buffer_rsrc_t rsZ = row_rsrc(w.dZ, NGD * 4, row_begin + b0, nrows);
            const int vz = (tid / Q) * (NGD * 4) + (tid % Q) * 16;
#pragma unroll
            for (int k = 0; k < CITER; ++k) {
                const int idx = tid + k * NT;
                const int i = idx / Q;
                const int u = (idx % Q) * 4;
                const bool carried = i < nrows_above;  // the sequence has a step t+1: recurrent dh and dc exist
                const float2 r01 = *reinterpret_cast<const float2*>(&Zs[i * LDZ + u]);
                const float2 r23 = *reinterpret_cast<const float2*>(&Zs[i * LDZ + u + 2]);
                const float recv[4] = {carried ? r01.x : 0.0f, carried ? r01.y : 0.0f, carried ? r23.x : 0.0f, carried ? r23.y : 0.0f};
                const float g_ = pcoef[k];
                const float dhl[4] = {g_ * pen[k].x - g_ * pep[k].x, g_ * pen[k].y - g_ * pep[k].y,
                                      g_ * pen[k].z - g_ * pep[k].z, g_ * pen[k].w - g_ * pep[k].w};
                const float gi[4] = {pg[k][0].x, pg[k][0].y, pg[k][0].z, pg[k][0].w};
                const float gf[4] = {pg[k][1].x, pg[k][1].y, pg[k][1].z, pg[k][1].w};
                const float gg[4] = {pg[k][2].x, pg[k][2].y, pg[k][2].z, pg[k][2].w};
                const float go[4] = {pg[k][3].x, pg[k][3].y, pg[k][3].z, pg[k][3].w};
                const float cc[4] = {pc[k].x, pc[k].y, pc[k].z, pc[k].w};
                const float cp[4] = {pcp[k].x, pcp[k].y, pcp[k].z, pcp[k].w};
                float dz[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dh = dhl[j] + recv[j];
                    float dco;
                    sbr_lstm_cell_bwd(dh, carried ? dc[k][j] : 0.0f, gi[j], gf[j], gg[j], go[j], cc[j], t > 0 ? cp[j] : 0.0f, NG == 3,
                                      &dz[0][j], &dz[1][j], &dz[2][j], &dz[3][j], &dco);
                    dc[k][j] = dco;
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int src = NG == 4 ? g : g + 1;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, make_float4(dz[src][0], dz[src][1], dz[src][2], dz[src][3])), rsZ,
                                                           vz + g * D * 4, k * (NT / Q) * (NGD * 4), SBR_AUX_BWD_Z);
                    float2* dst = reinterpret_cast<float2*>(&Zs[i * LDZ + g * D + u]);
                    dst[0] = make_float2(dz[src][0], dz[src][1]);
                    dst[1] = make_float2(dz[src][2], dz[src][3]);
                }
            }
        }
        BPROF(2)
        __syncthreads();  // B: dz tile complete
        BPROF(3)
        // (b) GEMM
        const int tid = thread_id();
        const int lane = tid & 63;
        const int c16 = lane & 15;
        const int kq = lane >> 4;
        f32x4 acc[2][RT];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[cc][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // The dz operand of k-block S + 1 is read from LDS BEFORE the MFMAs of block S are issued: a wave holds only two
        // accumulators here (the forward kernel: four), every second MFMA waits for its predecessor on the same accumulator, and
        // an LDS read queued behind them would expose its whole latency once per block (measured: 71 % of the pipe in this
        // phase against the forward kernel's 86 %)
        float av[RT][4];
        auto read_a = [&](int S) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float* arow = &Zs[(rt * 16 + c16) * LDZ + 16 * (S < NSZ ? S : NSZ - 1) + kq];
                av[rt][0] = arow[0]; av[rt][1] = arow[4]; av[rt][2] = arow[8]; av[rt][3] = arow[12];
            }
        };
        read_a(0);
        auto mma_block = [&](int S, int slot) {
            float ac[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) ac[rt][sub] = av[rt][sub];
            read_a(S + 1);
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    acc[0][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[rt][sub], ring0[slot][sub], acc[0][rt], 0, 0, 0);
                    acc[1][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[rt][sub], ring1[slot][sub], acc[1][rt], 0, 0, 0);
                }
        };
#pragma unroll
        for (int j = 0; j < PF; ++j) {  // head (peeled)
            mma_block(j, j);
            __builtin_amdgcn_sched_barrier(0);
            load_ring(j, j + PF);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t > 0) {  // behind the head's fragment requests: gathers of step t-1, ids of step t-2
            request_gathers(t - 1);
            request_ids(t > 1 ? t - 2 : 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int S0 = PF; S0 < NSZ - PF; S0 += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                mma_block(S0 + j, j);
                __builtin_amdgcn_sched_barrier(0);
                load_ring(j, S0 + j + PF);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int j = 0; j < PF; ++j) {  // tail
            mma_block(NSZ - PF + j, j);
            __builtin_amdgcn_sched_barrier(0);
        }
        // gate and cell-state rows of step t-1: requested now (the fragment ring is dead), they land under the
        // epilogue and the barrier.  At t = 0 nothing follows; the requests are kept (valid rows, values unused)
        // so that the operation counts stay static.
        {
            const int tn = t > 0 ? t - 1 : 0;
#pragma unroll
            for (int k = 0; k < CITER; ++k) pc[k] = pcp[k];
            request_gates(tn);
            request_cstate(pcp, tn > 0 ? tn - 1 : 0);
        }
        BPROF(4)
        __syncthreads();  // C: every wave is done reading the dz tile
        BPROF(5)
        // epilogue: dX to HBM (branch-free), recurrent dh into columns [0, D) of the tile
        {
            const __amdgpu_buffer_rsrc_t rsX = row_rsrc(blk.dX, D * 4, row_begin + b0, nrows);
            const int vx = (kq * 4 * D + wv * 16 + c16) * 4;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int i = rt * 16 + kq * 4 + reg;
                    const float dx = acc[0][rt][reg];  // (a bit_cast applied directly to the vector element reads element 0)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dx), rsX, vx, (rt * 16 + reg) * D * 4, SBR_AUX_BWD_ST);
                    Zs[i * LDZ + wv * 16 + c16] = acc[1][rt][reg];
                }
        }
        // first fragment blocks of the next step's GEMM (the weights do not change): requested behind the
        // epilogue's stores, which have the whole cell phase to retire
#pragma unroll
        for (int j = 0; j < PF; ++j) load_ring(j, j);
    };
    step(ts);
#pragma unroll 1
    for (int t = ts - 1; t >= 0; --t) step(t);
#ifdef SBR_PROF_BWD
    if ((tile == 20 || tile == 350) && (tid0 & 63) == 0 && (wv == 0 || wv == 5))
        printf("BWDPROF tile %d wave %d steps %d per-step cycles: epilogue+ringreq %lld barA %lld cell %lld barB %lld gemm+requests %lld barC %lld\n", tile, wv,
               nsteps, bprof[0] / nsteps, bprof[1] / nsteps, bprof[2] / nsteps, bprof[3] / nsteps, bprof[4] / nsteps, bprof[5] / nsteps);
#endif
}

// ------------------------------------------------------------------------------------------------
// Dense gradient dW[k][j] = sum_r xh[r][k] dz[r][j] : v_mfma_f32_32x32x2_f32.  One workgroup (4
// waves, 2x2) per 128x128 output tile and 1024-row chunk; every wave keeps a 64x64 sub-tile (2x2
// accumulators, from 0, rows ascending).  32-row slabs of xh (gathered: E rows / previous hidden
// rows) and dz are staged through LDS with 16-byte loads; the next slab's loads are in flight
// while the current one feeds the MFMAs.
// ------------------------------------------------------------------------------------------------
template <int D, int NG>
__global__ __launch_bounds__(256) void lstm_dw_kernel(ModelView m, MbView mb, BlockView blk, WorkView w) {
    constexpr int K2 = 2 * D;
    constexpr int NGD = NG * D;
    constexpr int TJ = (NGD + 127) / 128;
    constexpr int SLAB = 32;
    __shared__ float Xs[SLAB * 128];
    __shared__ float Zs[SLAB * 128];
    __shared__ int s_prev[SBR_DW_CHUNK_ROWS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, wj = wave & 1;
    // XCD-aware tile order: workgroups are handed to the 8 XCDs round-robin by linear id, and each XCD has
    // its own L2.  All tiles of one chunk therefore share an XCD (chunk = 8 * group + id % 8) and follow
    // each other in that XCD's dispatch order (tile = (id / 8) % tiles), so the chunk's X / H / dZ rows are
    // fetched from HBM once and re-read by the other tiles from that L2.
    constexpr int NTILE = ((K2 + 127) / 128) * TJ;
    const int wid = blockIdx.x;
    const int tile = (wid >> 3) % NTILE;
    const int c = (wid >> 3) / NTILE * 8 + (wid & 7);
    if (c * SBR_DW_CHUNK_ROWS >= mb.R) return;
    const int tk = tile / TJ, tj = tile % TJ;
    const int r0 = c * SBR_DW_CHUNK_ROWS;
    int r1 = r0 + SBR_DW_CHUNK_ROWS;
    if (r1 > mb.R) r1 = mb.R;
    const int nr = r1 - r0;
    for (int i = tid; i < nr; i += 256) {
        s_prev[i] = mb.prev_row[r0 + i];
    }
    __syncthreads();
    const int l31 = lane & 31;
    const int hh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;
    // staging map: thread owns column quad c4 of rows (tid/32) + 8*i, i = 0..3
    const int c4 = (tid & 31) * 4;
    const int srow = tid >> 5;
    const int kcol = tk * 128 + c4;
    const int jcol = tj * 128 + c4;
    float4 xr[4], zr[4];
    auto fetch = [&](int slab) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int lr = slab * SLAB + srow + 8 * i;
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), zv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lr < nr) {
                if (kcol < D) xv = ld4(w.X + (size_t)(r0 + lr) * D + kcol);
                else if (kcol < K2) {
                    const int pr = s_prev[lr];
                    if (pr >= 0) xv = ld4(blk.H + (size_t)pr * D + (kcol - D));
                }
                if (jcol < NGD) zv = ld4(w.dZ + (size_t)(r0 + lr) * NGD + jcol);
            }
            xr[i] = xv;
            zr[i] = zv;
        }
    };
    const int nslabs = (nr + SLAB - 1) / SLAB;
    // bias-gradient row (k = 2D): column sums of dz, a plain add chain over the rows of the chunk;
    // done by the tk == 0 workgroups from the dz slab they stage anyway (threads 0..127, one column each)
    float bias_acc = 0.0f;
    const bool do_bias = tk == 0 && tid < 128 && tj * 128 + tid < NGD;
    fetch(0);
    for (int slab = 0; slab < nslabs; ++slab) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            st4(&Xs[(srow + 8 * i) * 128 + c4], xr[i]);
            st4(&Zs[(srow + 8 * i) * 128 + c4], zr[i]);
        }
        __syncthreads();
        if (slab + 1 < nslabs) fetch(slab + 1);
        const float* xa = &Xs[hh * 128 + wk * 64 + l31];
        const float* zb = &Zs[hh * 128 + wj * 64 + l31];
#pragma unroll 4
        for (int s = 0; s < SLAB / 2; ++s) {
            const float a0 = xa[s * 256], a1 = xa[s * 256 + 32];
            const float b0 = zb[s * 256], b1 = zb[s * 256 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (do_bias) {
            const int rows_here = nr - slab * SLAB < SLAB ? nr - slab * SLAB : SLAB;
            for (int r = 0; r < rows_here; ++r) bias_acc = bias_acc + Zs[r * 128 + tid];
        }
        __syncthreads();
    }
    if (do_bias) w.partials[((size_t)c * (K2 + 1) + K2) * NGD + tj * 128 + tid] = bias_acc;
    float* part = w.partials + (size_t)c * (K2 + 1) * NGD;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = tj * 128 + wj * 64 + b * 32 + l31;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = tk * 128 + wk * 64 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh;
                if (k < K2 && j < NGD) part[(size_t)k * NGD + j] = acc[a][b][q];
            }
        }
}

// Same product for shapes whose 128x128 tiles are full (K2 and NG*D multiples of 128: d = 128, d = 256, d = 64
// with four gates), written on the recurrent kernels' recipe: branch-free slab fetches (rows past the end of the
// last chunk are clamped to a valid row for X / H and find ZEROS in dZ — the engine clears dZ's rows between R
// and the end of R's chunk — so they add +0 to every chain; first-step rows take their h_{t-1} from a row of
// zeros), so that every wait count is static, and the LDS operand reads of iteration s+1 are issued before the
// MFMAs of iteration s.
template <int D, int NG>
__global__ __launch_bounds__(256, SBR_DW_WPE) void lstm_dw_full_kernel(ModelView m, MbView mb, BlockView blk, WorkView w) {
    constexpr int K2 = 2 * D;
    constexpr int NGD = NG * D;
    constexpr int TJ = NGD / 128;
    constexpr int SLAB = SBR_DW_SLAB;  // rows staged through LDS per barrier pair
    constexpr int NI = SLAB / 8;       // 16-byte row pieces per thread and operand
    constexpr int NTILE = (K2 / 128) * TJ;
    static_assert(K2 % 128 == 0 && NGD % 128 == 0, "full tiles only");
    __shared__ float Xs[SLAB * 128];
    __shared__ float Zs[SLAB * 128];
    __shared__ int s_prev[SBR_DW_CHUNK_ROWS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, wj = wave & 1;
    const int wid = blockIdx.x;
    const int tile = (wid >> 3) % NTILE;
    const int c = (wid >> 3) / NTILE * 8 + (wid & 7);  // XCD-aware: all tiles of a chunk on one XCD (see lstm_dw_kernel)
    if (c * SBR_DW_CHUNK_ROWS >= mb.R) return;
    const int tk = tile / TJ, tj = tile % TJ;
    const int r0 = c * SBR_DW_CHUNK_ROWS;
    const int last = mb.R - 1;
    for (int i = tid; i < SBR_DW_CHUNK_ROWS; i += 256) s_prev[i] = mb.prev_row[r0 + i <= last ? r0 + i : last];
    __syncthreads();
    const int l31 = lane & 31;
    const int hh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;
    const int c4 = (tid & 31) * 4;
    const int srow = tid >> 5;
    const int kcol = tk * 128 + c4;
    const int jcol = tj * 128 + c4;
    const bool xpart = kcol < D;  // this thread's k columns are input (x) columns; otherwise previous-hidden columns
    float4 xr[NI], zr[NI];
    // Slab fetches through buffer resources (scalar base, invariant per-lane byte offset, scalar slab offset): dZ and X
    // rows of the chunk are contiguous; the previous-hidden rows are addressed by prev_row x row bytes, and a first-step
    // row (prev_row = -1) becomes an out-of-range offset, which reads as zeros.  Needs a 128-column tile to lie entirely
    // in the x or in the h half (D >= 128) and H below 2 GiB; otherwise the 64-bit per-lane addresses below.
    const bool small_h = !w.wide_addresses && (size_t)mb.R * D * 4 < ((size_t)1 << 31);  // byte offsets into H stay positive ints
    const int xrows = mb.R - r0 < SBR_DW_CHUNK_ROWS ? mb.R - r0 : SBR_DW_CHUNK_ROWS;
    // the resources of X and dZ end at the chunk's last LIVE row: rows past R are out of range and read as zeros (no clearing launch)
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void*)(w.dZ + (size_t)r0 * NGD), (short)0, xrows * NGD * 4, SBR_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsXc = __builtin_amdgcn_make_buffer_rsrc((void*)(w.X + (size_t)r0 * D), (short)0, xrows * D * 4, SBR_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)blk.H, (short)0, small_h ? (int)((size_t)mb.R * D * 4) : 0, SBR_BUFFER_RSRC_FLAGS);
    auto fetch = [&](int slab) {
        if (D >= 128 && small_h) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int lr = slab * SLAB + srow + 8 * i;
                if (tk * 128 < D) {  // x columns (uniform per workgroup); rows past R are out of range and read as zeros: they meet zeros in dZ
                    xr[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsXc, (srow * D + kcol) * 4, (slab * SLAB + 8 * i) * D * 4, SBR_AUX_DW_LD));
                } else {
                    const int pr = s_prev[lr];
                    xr[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsH, pr >= 0 ? (int)((uint32_t)pr * (uint32_t)(D * 4) + (uint32_t)((kcol - D) * 4)) : -16, 0, 0));
                }
                zr[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsZ, (srow * NGD + jcol) * 4, (slab * SLAB + 8 * i) * NGD * 4, SBR_AUX_DW_LD));
            }
            return;
        }
        const float* X = launder(w.X);
        const float* H = launder(blk.H);
        const float* dZ = launder(w.dZ);
        const float* zero = launder(w.zeros);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int lr = slab * SLAB + srow + 8 * i;
            const int row = r0 + lr <= last ? r0 + lr : last;
            const int pr = s_prev[lr];
            const float* src = xpart ? X + ((size_t)row * D + kcol) : (pr >= 0 ? H + ((size_t)pr * D + (kcol - D)) : zero + (kcol - D));
            xr[i] = ld4(src);
            zr[i] = ld4(dZ + ((size_t)(r0 + lr) * NGD + jcol));  // rows past R: cleared by the engine
        }
    };
    int nr = mb.R - r0;
    nr = nr < SBR_DW_CHUNK_ROWS ? nr : SBR_DW_CHUNK_ROWS;
    const int nslabs = (nr + SLAB - 1) / SLAB;
    float bias_acc = 0.0f;
    const bool do_bias = tk == 0 && tid < 128;
    fetch(0);
    for (int slab = 0; slab < nslabs; ++slab) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            st4(&Xs[(srow + 8 * i) * 128 + c4], xr[i]);
            st4(&Zs[(srow + 8 * i) * 128 + c4], zr[i]);
        }
        __syncthreads();
        fetch(slab + 1 < nslabs ? slab + 1 : slab);  // unconditional (static operation count); the last one is unused
        const float* xa = &Xs[hh * 128 + wk * 64 + l31];
        const float* zb = &Zs[hh * 128 + wj * 64 + l31];
        float a0 = xa[0], a1 = xa[32], b0 = zb[0], b1 = zb[32];
#pragma unroll
        for (int s = 0; s < SLAB / 2; ++s) {
            const int sn = s + 1 < SLAB / 2 ? s + 1 : s;
            const float na0 = xa[sn * 256], na1 = xa[sn * 256 + 32];
            const float nb0 = zb[sn * 256], nb1 = zb[sn * 256 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        if (do_bias) {  // bias-gradient row: plain add chain over the rows of the chunk (rows past R add +0)
            const int rows_here = nr - slab * SLAB < SLAB ? nr - slab * SLAB : SLAB;
            for (int r = 0; r < rows_here; ++r) bias_acc = bias_acc + Zs[r * 128 + tid];
        }
        __syncthreads();
    }
    if (do_bias) w.partials[((size_t)c * (K2 + 1) + K2) * NGD + tj * 128 + tid] = bias_acc;
    float* part = w.partials + (size_t)c * (K2 + 1) * NGD;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = tj * 128 + wj * 64 + b * 32 + l31;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = tk * 128 + wk * 64 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh;
                part[(size_t)k * NGD + j] = acc[a][b][q];
            }
        }
}

__global__ void dense_reduce_local_kernel(const float* partials, int nchunks, size_t n, float* dense) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = partials[i];
    int c = 1;
    for (; c + 8 <= nchunks; c += 8) { /* loads run ahead of the ordered add chain */
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = partials[(size_t)(c + j) * n + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = acc + v[j];
    }
    for (; c < nchunks; ++c) acc = acc + partials[(size_t)c * n + i];
    dense[i] = acc;
}

// ------------------------------------------------------------------------------------------------
// K6 dense: sum of the device blocks in device order + Adagrad; LSTM weights are re-emitted in the
// two MFMA-fragment packings (Wp for the forward GEMM, WTp for the BPTT GEMM).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t wp_index(int k, int jcol, int D, int NG) {
    const int K2 = 2 * D;
    const int g = jcol / D, u = jcol % D;
    const int ut = u >> 4, j16 = u & 15;
    const int S = k >> 4, sub = (k & 15) >> 2, kq = k & 3;
    const int lane = kq * 16 + j16;
    return ((((size_t)(ut * NG + g) * (K2 / 16) + S) * 64 + lane) * 4 + sub);
}
__device__ __forceinline__ size_t wtp_index(int k, int jcol, int D, int NG) {
    const int NGD = NG * D;
    const int ct = k >> 4, c16 = k & 15;
    const int S = jcol >> 4, sub = (jcol & 15) >> 2, kq = jcol & 3;
    const int lane = kq * 16 + c16;
    return ((((size_t)ct * (NGD / 16) + S) * 64 + lane) * 4 + sub);
}

__device__ __forceinline__ void dense_apply_element(const ModelView& m, size_t i, float g);
// partials != nullptr (one device): the gradient is still `nchunks` chunk partials — their ordered sum (dense_reduce_local_kernel's
// chain) is formed here, written to dense_out, and applied: one launch instead of two on the tail of the step
__global__ void dense_apply_kernel(ModelView m, const uint8_t* all_blocks, uint64_t block_bytes, uint64_t dense_off,
                                   int ndev, size_t n, const float* partials, int nchunks, float* dense_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float g;
    if (partials) {
        g = partials[i];
        int c = 1;
        for (; c + 8 <= nchunks; c += 8) { /* loads run ahead of the ordered add chain */
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = partials[(size_t)(c + j) * n + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) g = g + v[j];
        }
        for (; c < nchunks; ++c) g = g + partials[(size_t)c * n + i];
        dense_out[i] = g;
    } else {
        g = reinterpret_cast<const float*>(all_blocks + dense_off)[i];
        for (int q = 1; q < ndev; ++q) g = g + reinterpret_cast<const float*>(all_blocks + (size_t)q * block_bytes + dense_off)[i];
    }
    dense_apply_element(m, i, g);
}
// optimiser update of element i of the dense parameters (LSTM: [W | bW], re-emitting the packed weight copies; EWMA: alpha)
__device__ __forceinline__ void dense_apply_element(const ModelView& m, size_t i, float g) {
    if (m.ng) {
        const int NGD = m.ng * m.d;
        const size_t nw = (size_t)2 * m.d * NGD;
        if (i < nw) {
            const bool adam = m.optimizer == SBR_OPT_ADAM;
            float wv = m.W[i], G = m.Wacc[i], M = adam ? m.Wm[i] : 0.0f;
            opt_update(m, &wv, &G, &M, g);
            m.W[i] = wv;
            m.Wacc[i] = G;
            if (adam) m.Wm[i] = M;
            const int k = (int)(i / NGD), jcol = (int)(i % NGD);
            m.Wp[wp_index(k, jcol, m.d, m.ng)] = wv;
            m.WTp[wtp_index(k, jcol, m.d, m.ng)] = wv;
        } else {
            const size_t j = i - nw;
            const bool adam = m.optimizer == SBR_OPT_ADAM;
            float wv = m.bW[j], G = m.bWacc[j], M = adam ? m.bWm[j] : 0.0f;
            opt_update(m, &wv, &G, &M, g);
            m.bW[j] = wv;
            m.bWacc[j] = G;
            if (adam) m.bWm[j] = M;
        }
    } else {
        const bool adam = m.optimizer == SBR_OPT_ADAM;
        float wv = m.alpha[i], G = m.alpha_acc[i], M = adam ? m.alpha_m[i] : 0.0f;
        opt_update(m, &wv, &G, &M, g);
        m.alpha[i] = wv;
        m.alpha_acc[i] = G;
        if (adam) m.alpha_m[i] = M;
    }
}

__global__ void repack_lstm_kernel(ModelView m) {
    const int NGD = m.ng * m.d;
    const size_t nw = (size_t)2 * m.d * NGD;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw) return;
    const int k = (int)(i / NGD), jcol = (int)(i % NGD);
    const float wv = m.W[i];
    m.Wp[wp_index(k, jcol, m.d, m.ng)] = wv;
    m.WTp[wtp_index(k, jcol, m.d, m.ng)] = wv;
}

// one optimiser update of an item-embedding row (4 elements per lane of the row's group) and its bias
__device__ __forceinline__ void bias_update(const ModelView& m, uint64_t row, int lg, float gb) {
    if (lg != 0) return;
    const bool adam = m.optimizer == SBR_OPT_ADAM;
    float bv = m.b[row], ba = m.bacc[row], bmm = adam ? m.bm[row] : 0.0f;
    opt_update(m, &bv, &ba, &bmm, gb);
    m.b[row] = bv;
    m.bacc[row] = ba;
    if (adam) m.bm[row] = bmm;
}
template <int D>
__device__ __forceinline__ void row_update(const ModelView& m, uint64_t row, int lg, float4 g, bool has_b, float gb) {
    const bool adam = m.optimizer == SBR_OPT_ADAM;
    float* wrow = m.E + row * D + 4 * lg;
    float* arow = m.Eacc + row * D + 4 * lg;
    float4 wv = ld4(wrow), av = ld4(arow);
    float4 mv = adam ? ld4(m.Em + row * D + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
    opt_update(m, &wv.x, &av.x, &mv.x, g.x);
    opt_update(m, &wv.y, &av.y, &mv.y, g.y);
    opt_update(m, &wv.z, &av.z, &mv.z, g.z);
    opt_update(m, &wv.w, &av.w, &mv.w, g.w);
    st4(wrow, wv);
    st4(arow, av);
    if (adam) st4(m.Em + row * D + 4 * lg, mv);
    if (has_b) bias_update(m, row, lg, gb);
}

// ------------------------------------------------------------------------------------------------
// K6 sparse: (row, source) keys -> radix sort -> per-row reduction in the contract's chunked order
// (SBR_SEG_CHUNK, sbr_numerics.h) -> Emit.  One lane group (D/4 lanes) owns a row segment of the sorted
// keys.  Segments of at most SBR_SEG_CHUNK entries — all but the hot rows of a skewed catalogue — are
// reduced in place by `seg_short_kernel`; longer ones are only registered there, their chunks are
// reduced in parallel by `seg_chunk_kernel` (one lane group per chunk) and the chunk partials are
// added in order by `seg_finish_kernel`.  The three consumers differ only in what happens to a
// finished row (Emit): optimiser update, write into the owner's send chunk, or entry of the list.
// ------------------------------------------------------------------------------------------------

struct RowPrefetch {  // the row's parameter / optimiser-state quads, requested as soon as the row id is known
    float4 w, a, mo;
};
struct EmitApply {  // single device: one optimiser update per touched row
    ModelView m;
    template <int D>
    __device__ __forceinline__ RowPrefetch pre(uint32_t r, int lg) const {
        RowPrefetch q;
        q.w = ld4_upd(m.E + (size_t)r * D + 4 * lg);
        q.a = ld4_upd(m.Eacc + (size_t)r * D + 4 * lg);
        q.mo = m.optimizer == SBR_OPT_ADAM ? ld4(m.Em + (size_t)r * D + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
        return q;
    }
    template <int D>
    __device__ __forceinline__ void row(uint32_t r, uint64_t, int lg, float4 g, bool has_b, float gb, RowPrefetch q) const {
        const bool adam = m.optimizer == SBR_OPT_ADAM;
        opt_update(m, &q.w.x, &q.a.x, &q.mo.x, g.x);
        opt_update(m, &q.w.y, &q.a.y, &q.mo.y, g.y);
        opt_update(m, &q.w.z, &q.a.z, &q.mo.z, g.z);
        opt_update(m, &q.w.w, &q.a.w, &q.mo.w, g.w);
        st4_upd(m.E + (size_t)r * D + 4 * lg, q.w);
        st4_upd(m.Eacc + (size_t)r * D + 4 * lg, q.a);
        if (adam) st4(m.Em + (size_t)r * D + 4 * lg, q.mo);
        if (has_b) bias_update(m, r, lg, gb);
    }
};
struct EmitChunk {  // replicated multi-device: the row's sum goes into the owner's dense send chunk
    void* send;
    uint64_t S;
    template <int D>
    __device__ __forceinline__ RowPrefetch pre(uint32_t, int) const { return RowPrefetch{}; }
    template <int D>
    __device__ __forceinline__ void row(uint32_t r, uint64_t, int lg, float4 g, bool has_b, float gb, RowPrefetch) const {
        /* (S = ceil(num_items / num_devices) < 2^32: the owner and the row inside its slice by 32-bit division — the 64-bit one cost this
         * kernel a third of its registers) */
        const uint32_t S32 = (uint32_t)S, owner = r / S32;
        float* c = reinterpret_cast<float*>(send) + (uint64_t)owner * S * ((uint64_t)D + 2);
        const uint64_t lr = r - owner * S32;
        st4(c + lr * D + 4 * lg, g);
        if (lg == 0) {
            if (has_b) c[S * D + lr] = gb;
            reinterpret_cast<uint32_t*>(c + S * D + S)[lr] = 1u | (has_b ? 2u : 0u);
        }
    }
};
struct EmitList {  // partitioned table: list entry addressed by the position of the row's first key (fl is
    float* G;       // cleared beforehand: 0 at every position that is not a segment head)
    float* gbl;
    uint32_t* fl;
    template <int D>
    __device__ __forceinline__ RowPrefetch pre(uint32_t, int) const { return RowPrefetch{}; }
    template <int D>
    __device__ __forceinline__ void row(uint32_t, uint64_t p, int lg, float4 g, bool has_b, float gb, RowPrefetch) const {
        st4(G + p * D + 4 * lg, g);
        if (lg == 0) {
            gbl[p] = gb;
            fl[p] = 1u | (has_b ? 2u : 0u);
        }
    }
};

// in-order sum of the entries keys[begin, end) of one row (the first one initialises); the row loads of
// four consecutive entries are issued together, the adds stay in order
template <int D>
__device__ __forceinline__ void seg_accumulate(const BlockView& blk, const uint64_t* keys, uint64_t begin, uint64_t end, int lg,
                                               float4* g_out, float* gb_out, bool* has_b_out) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    float gb = 0.0f;
    bool has_b = false, first = true;
    for (uint64_t e = begin; e < end; e += 4) {
        float4 v[4];
        float sc[4];
        bool bias[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            sc[i] = 0.0f;
            bias[i] = false;
            if (e + i < end) { /* no traffic for the slots past the segment (most segments have 1-3 entries) */
                const uint32_t src = (uint32_t)keys[e + i];
                const uint32_t r = src / 3, kind = src % 3;
                v[i] = ld4((kind == 0 ? blk.dX : blk.H) + (size_t)r * D + 4 * lg);
                sc[i] = kind == 0 ? 1.0f : (kind == 1 ? -blk.coef[r] : blk.coef[r]);
                bias[i] = kind != 0;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (e + i < end) {
                if (first) {
                    g = make_float4(sc[i] * v[i].x, sc[i] * v[i].y, sc[i] * v[i].z, sc[i] * v[i].w);
                    first = false;
                } else {
                    g.x = g.x + sc[i] * v[i].x; g.y = g.y + sc[i] * v[i].y;
                    g.z = g.z + sc[i] * v[i].z; g.w = g.w + sc[i] * v[i].w;
                }
                if (bias[i]) {
                    gb = has_b ? gb + sc[i] : sc[i];
                    has_b = true;
                }
            }
        }
    }
    *g_out = g; *gb_out = gb; *has_b_out = has_b;
}

// The same in-order sum for the chunks of LONG segments (hot rows: up to SBR_SEG_CHUNK entries per call): NB entries per batch
// with their rows in flight together, and the NEXT batch's keys requested before this batch's rows are consumed — the short-segment
// form above pays two dependent memory round trips (key, then row + coefficient) per four entries, which made a 256-entry chunk
// 64 x ~2.5 us (Zipf(1) items at 8 192 sequences per step: 1 300 chunk units, seg_chunk_kernel 0.22 ms on the critical path).
// The chunk kernel takes NB = 4 at 80 registers (the short-segment kernel's budget): it is launched on every step, beside the
// dense-gradient GEMM whose four workgroups per CU leave 64-176 registers per SIMD lane free — at NB = 16 (256 registers) or NB = 8
// (140) its workgroups could not become resident until the GEMM's tail even when there was nothing for them to do (uniform items:
// 1.85 ms of elapsed time for an empty launch at 50 000 sequences per step), and 1 024 workgroups took four rounds.
template <int D, int NB>
__device__ __forceinline__ void seg_accumulate_long(const BlockView& blk, const uint64_t* keys, uint64_t begin, uint64_t end, int lg,
                                                    float4* g_out, float* gb_out, bool* has_b_out) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    float gb = 0.0f;
    bool has_b = false, first = true;
    const uint32_t* klo = reinterpret_cast<const uint32_t*>(keys);  // the low word of a key is its source (3 x packed row + kind)
    uint32_t src_n[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) src_n[i] = begin + i < end ? klo[2 * (begin + i)] : 0u;
#pragma unroll 1
    for (uint64_t e = begin; e < end; e += NB) {
        float4 v[NB];
        float sc[NB];
        uint32_t kind[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const uint32_t src = src_n[i];
            const uint32_t r = src / 3;
            kind[i] = src % 3;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            sc[i] = 0.0f;
            if (e + i < end) {
                v[i] = ld4((kind[i] == 0 ? blk.dX : blk.H) + (size_t)r * D + 4 * lg);
                sc[i] = blk.coef[r];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) src_n[i] = e + NB + i < end ? klo[2 * (e + NB + i)] : 0u;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (e + i < end) {
                const float s = kind[i] == 0 ? 1.0f : (kind[i] == 1 ? -sc[i] : sc[i]);
                if (first) {
                    g = make_float4(s * v[i].x, s * v[i].y, s * v[i].z, s * v[i].w);
                    first = false;
                } else {
                    g.x = g.x + s * v[i].x; g.y = g.y + s * v[i].y;
                    g.z = g.z + s * v[i].z; g.w = g.w + s * v[i].w;
                }
                if (kind[i] != 0) {
                    gb = has_b ? gb + s : s;
                    has_b = true;
                }
            }
        }
    }
    *g_out = g; *gb_out = gb; *has_b_out = has_b;
}

// first position in [lo, hi) whose row differs from `row` (keys are sorted by row)
__device__ __forceinline__ uint64_t seg_end(const uint64_t* keys, uint64_t lo, uint64_t hi, uint32_t row) {
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if ((uint32_t)(keys[mid] >> 32) == row) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One lane group per SEGMENT (the heads of the sorted keys were listed right after the sort): the group knows
// the segment's first key position p and its length from two consecutive heads, fetches the key window with
// one coalesced load, and then has every address it needs — the row's parameter / optimiser-state quads and
// the gradient source rows are requested together, so a segment costs two dependent memory round trips (keys,
// rows) instead of three, and no lane group spends a round trip finding out that its position is not a head.
// Positions are requested two segments ahead and key windows one segment ahead, so that in steady state only
// the row round trip is exposed.
// INLINE_LONG (small key counts: launch_seg_reduce): a segment of more than SBR_SEG_CHUNK entries is reduced right here by
// its lane group, chunk partial by chunk partial in the contract's order, instead of being registered for the three
// kernels of the chunked path — four launches fewer per step where a step is a handful of microseconds.
template <int D, class Emit, bool INLINE_LONG>
__device__ __forceinline__ void seg_short_rows(const BlockView& blk, const uint64_t* keys, uint64_t n, const SegScratch& sc, const Emit& emit,
                                               uint32_t wave, uint32_t nwaves) {
    constexpr int L = D / 4;
    constexpr int GPW = 64 / L;
    const int lane = threadIdx.x & 63, lg = lane % L, grp = lane / L;
    const int gbase = grp * L;
    const uint32_t nheads = *sc.nheads;
    const uint32_t stride = nwaves * GPW;
    uint32_t h = wave * GPW + grp;
    auto head_at = [&](uint32_t hh) { return sc.head_pos[hh < nheads ? hh : nheads]; };
    auto window_key = [&](uint32_t p0, uint32_t p1) {  // this lane's key of the segment [p0, p1) (clamped to valid memory)
        const uint32_t c = p1 - p0 < (uint32_t)L ? p1 - p0 : (uint32_t)L;
        return keys[(uint32_t)lg < c ? (uint64_t)p0 + lg : ((uint64_t)p0 < n ? (uint64_t)p0 : 0)];
    };
    // two segments ahead: (first key position, end); one segment ahead: its key window
    uint32_t p_cur = head_at(h), p_end = head_at(h + 1);
    uint32_t p_next = head_at(h + stride), p_next_end = head_at(h + stride + 1);
    uint64_t kmine = window_key(p_cur, p_end);
    for (uint32_t h0 = wave * GPW; h0 < nheads; h0 += stride, h += stride) {  // wave-uniform trip count
        const bool active = h < nheads;
        const uint32_t p_nn = head_at(h + 2 * stride), p_nn_end = head_at(h + 2 * stride + 1);
        const uint64_t p = p_cur;
        const uint32_t len = active ? p_end - p_cur : 0u;
        const uint32_t cnt = len < (uint32_t)L ? len : (uint32_t)L;  // entries of the segment inside the key window
        const uint32_t lo = (uint32_t)kmine;
        const uint32_t row = (uint32_t)__shfl((int)(uint32_t)(kmine >> 32), gbase, 64);
        // the next segment's key window (its position arrived an iteration ago) is requested here and consumed
        // in the next iteration
        kmine = window_key(p_next, p_next_end);
        p_cur = p_next;
        p_end = p_next_end;
        p_next = p_nn;
        p_next_end = p_nn_end;
        if (INLINE_LONG && len > SBR_SEG_CHUNK) { /* chunk partials in order: the first initialises (seg_chunk / seg_finish) */
            const RowPrefetch pre = emit.template pre<D>(row, lg);
            float4 g;
            float gb;
            bool has_b;
            seg_accumulate_long<D, 8>(blk, keys, p, p + SBR_SEG_CHUNK, lg, &g, &gb, &has_b);
            for (uint64_t q = p + SBR_SEG_CHUNK; q < p + len; q += SBR_SEG_CHUNK) {
                float4 v;
                float vb;
                bool vh;
                seg_accumulate_long<D, 8>(blk, keys, q, q + SBR_SEG_CHUNK < p + len ? q + SBR_SEG_CHUNK : p + len, lg, &v, &vb, &vh);
                g.x = g.x + v.x; g.y = g.y + v.y; g.z = g.z + v.z; g.w = g.w + v.w;
                if (vh) {
                    gb = has_b ? gb + vb : vb;
                    has_b = true;
                }
            }
            emit.template row<D>(row, p, lg, g, has_b, gb, pre);
            continue;
        }
        if (!INLINE_LONG && len > SBR_SEG_ROUTE) { /* long segment: registered for the chunked path (or listed already: seg_long_list_kernel) */
            if (lg == 0 && !sc.prelisted) {
                const uint32_t slot = atomicAdd(&sc.counters[0], 1u);
                if (slot < sc.cap) {
                    sc.long_start[slot] = (uint32_t)p;
                    sc.long_end[slot] = (uint32_t)p + len;
                }
            }
            continue;
        }
        if (!active) continue;
        const RowPrefetch pre = emit.template pre<D>(row, lg);
        float4 g;
        float gb;
        bool has_b;
        if (len <= (uint32_t)L) { /* the whole segment is in the window */
            g = make_float4(0.f, 0.f, 0.f, 0.f);
            gb = 0.0f;
            has_b = false;
            bool first = true;
            for (int e = 0; e < (int)cnt; e += 4) {
                float4 v[4];
                float scl[4];
                bool bias[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    scl[i] = 0.0f;
                    bias[i] = false;
                    if (e + i < (int)cnt) {
                        const uint32_t src = (uint32_t)__shfl((int)lo, gbase + e + i, 64);
                        const uint32_t r = src / 3, kind = src % 3;
                        v[i] = ld4((kind == 0 ? blk.dX : blk.H) + (size_t)r * D + 4 * lg);
                        scl[i] = kind == 0 ? 1.0f : (kind == 1 ? -blk.coef[r] : blk.coef[r]);
                        bias[i] = kind != 0;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (e + i < (int)cnt) {
                        if (first) {
                            g = make_float4(scl[i] * v[i].x, scl[i] * v[i].y, scl[i] * v[i].z, scl[i] * v[i].w);
                            first = false;
                        } else {
                            g.x = g.x + scl[i] * v[i].x; g.y = g.y + scl[i] * v[i].y;
                            g.z = g.z + scl[i] * v[i].z; g.w = g.w + scl[i] * v[i].w;
                        }
                        if (bias[i]) {
                            gb = has_b ? gb + scl[i] : scl[i];
                            has_b = true;
                        }
                    }
                }
            }
        } else {
            seg_accumulate<D>(blk, keys, p, p + len, lg, &g, &gb, &has_b);
        }
        emit.template row<D>(row, p, lg, g, has_b, gb, pre);
    }
}
template <int D, class Emit, bool INLINE_LONG = false>
__global__ __launch_bounds__(256) void seg_short_kernel(BlockView blk, const uint64_t* keys, uint64_t n, SegScratch sc, Emit emit) {
    seg_short_rows<D, Emit, INLINE_LONG>(blk, keys, n, sc, emit, (uint32_t)((blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6),
                                         (uint32_t)(((uint64_t)gridDim.x * blockDim.x) >> 6));
}

// ---- the optimiser half of a SMALL single-device LSTM step at d <= 32 in ONE launch (launch_small_back) ----
// Workgroups 0 .. dw_blocks-1: one thread per element of the dense gradient — its chain over the step's packed rows in row
// order (what a v_mfma_f32_32x32x2 accumulator of lstm_dw_block_kernel holds: fma per row from +0; the bias row is the add
// chain over dz) and, the gradient being elementwise, the element's optimiser update right behind it: dense-gradient launch
// and dense-update launch become none.  The remaining workgroups: the sparse update's segments (seg_short_rows), which do
// not depend on the dense half.  A single chunk only (<= SBR_DW_CHUNK_ROWS rows), so the chain IS the gradient.
template <int D>
__global__ __launch_bounds__(256) void small_back_kernel(ModelView m, MbView mb, BlockView blk, WorkView w, const uint64_t* keys, uint64_t nkeys,
                                                         SegScratch sc, int dw_blocks) {
    if ((int)blockIdx.x >= dw_blocks) {
        seg_short_rows<D, EmitApply, true>(blk, keys, nkeys, sc, EmitApply{m}, (uint32_t)((((uint64_t)blockIdx.x - dw_blocks) * 256 + threadIdx.x) >> 6),
                                           (uint32_t)((((uint64_t)gridDim.x - dw_blocks) * 256) >> 6));
        return;
    }
    if (m.ng == 0) { /* EWMA (one dense workgroup): dalpha = ordered sum of the sequences' partials (ewma_dab_final_kernel), then its update */
        const int k = threadIdx.x;
        if (k >= D) return;
        float pc = 0.0f;
        for (int b = 0; b < mb.B; ++b) pc = pc + w.dab[(size_t)b * D + k];
        const float a = sbr_sigmoidf(m.alpha[k]);
        const float g = pc * (a * (1.0f - a));
        blk.dense[k] = g;
        dense_apply_element(m, (size_t)k, g);
        return;
    }
    const int NGD = m.ng * D, K2 = 2 * D, R = mb.R;
    const size_t n = (size_t)(K2 + 1) * NGD;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    // operands through LDS: dZ of the whole step (<= 128 rows x <= 128 columns) and the xh columns of this workgroup's gradient rows,
    // requested by all threads at once — read straight from memory, eight rows at a time, every batch was a ~1 us round trip ahead
    // of its eight fmas (13 us per launch at ~105 rows)
    extern __shared__ __attribute__((aligned(16))) float sb_lds[];
    float* zs = sb_lds;                  // [R][NGD]
    float* xs = zs + (size_t)R * NGD;    // [nk][R]
    const int kmin = (int)(((size_t)blockIdx.x * 256) / NGD);
    const int klast = (int)(((size_t)blockIdx.x * 256 + 255) / NGD);
    const int nk = (klast < K2 ? klast : K2 - 1) - kmin + 1;  // (row K2 is the bias row: no xh operand)
    for (int idx = threadIdx.x; idx < R * NGD / 4; idx += 256) st4(zs + 4 * (size_t)idx, ld4(w.dZ + 4 * (size_t)idx));
    for (int idx = threadIdx.x; idx < nk * R; idx += 256) {
        const int kk = kmin + idx / R, r = idx % R;
        float v;
        if (kk < D) {
            v = w.X[(size_t)r * D + kk];
        } else {
            const int pr = mb.prev_row[r];
            v = pr >= 0 ? blk.H[(size_t)pr * D + (kk - D)] : 0.0f;
        }
        xs[idx] = v;
    }
    __syncthreads();
    if (i >= n) return;
    const int k = (int)(i / NGD), j = (int)(i % NGD);
    const float* zc = zs + j;
    float acc = 0.0f;
    if (k < K2) {
        const float* xc = xs + (size_t)(k - kmin) * R;
#pragma unroll 8
        for (int r = 0; r < R; ++r) acc = sbr_fma(xc[r], zc[(size_t)r * NGD], acc);
    } else {
#pragma unroll 8
        for (int r = 0; r < R; ++r) acc = acc + zc[(size_t)r * NGD];
    }
    blk.dense[i] = acc;
    dense_apply_element(m, i, acc);
}

// the long segments straight from the head list (launch_seg_prelist): what seg_short_kernel registers as it meets them, known
// before the update starts
__global__ __launch_bounds__(256) void seg_long_list_kernel(SegScratch sc) {
    const uint32_t nheads = *sc.nheads;
    for (uint32_t h = blockIdx.x * 256 + threadIdx.x; h < nheads; h += gridDim.x * 256) {
        const uint32_t p0 = sc.head_pos[h], p1 = sc.head_pos[h + 1];
        if (p1 - p0 > SBR_SEG_ROUTE) {
            const uint32_t slot = atomicAdd(&sc.counters[0], 1u);
            if (slot < sc.cap) {
                sc.long_start[slot] = p0;
                sc.long_end[slot] = p1;
            }
        }
    }
}

// prefix of the long segments' chunk counts (a few thousand entries at most: one workgroup)
__global__ void seg_units_kernel(SegScratch sc) {
    __shared__ uint32_t s_sum[256];
    const uint32_t nlong = sc.counters[0] < sc.cap ? sc.counters[0] : sc.cap;
    const uint32_t per = (nlong + 255) / 256;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < nlong ? lo + per : nlong;
    uint32_t local = 0;
    for (uint32_t i = lo; i < hi; ++i) local += (sc.long_end[i] - sc.long_start[i] + SBR_SEG_CHUNK - 1) / SBR_SEG_CHUNK;
    s_sum[threadIdx.x] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int t = 0; t < 256; ++t) { const uint32_t v = s_sum[t]; s_sum[t] = run; run += v; }
        sc.counters[1] = run;
        sc.unit_base[nlong] = run;
    }
    __syncthreads();
    uint32_t run = s_sum[threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) {
        sc.unit_base[i] = run;
        run += (sc.long_end[i] - sc.long_start[i] + SBR_SEG_CHUNK - 1) / SBR_SEG_CHUNK;
    }
}

template <int D>
__global__ __launch_bounds__(256, 6) void seg_chunk_kernel(BlockView blk, const uint64_t* keys, SegScratch sc) {
    constexpr int L = D / 4;
    constexpr int GPW = 64 / L;
    const int lane = threadIdx.x & 63, lg = lane % L, grp = lane / L;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t nlong = sc.counters[0] < sc.cap ? sc.counters[0] : sc.cap;
    const uint32_t units = sc.counters[1];
    for (uint64_t u = wave * GPW + grp; u < units; u += nwaves * GPW) {
        uint32_t lo = 0, hi = nlong; /* last segment whose unit_base <= u */
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (sc.unit_base[mid] <= u) lo = mid; else hi = mid;
        }
        const uint64_t begin = (uint64_t)sc.long_start[lo] + (u - sc.unit_base[lo]) * SBR_SEG_CHUNK;
        const uint64_t end = begin + SBR_SEG_CHUNK < sc.long_end[lo] ? begin + SBR_SEG_CHUNK : sc.long_end[lo];
        float4 g;
        float gb;
        bool has_b;
        seg_accumulate_long<D, 4>(blk, keys, begin, end, lg, &g, &gb, &has_b);
        st4(sc.P + u * D + 4 * lg, g);
        if (lg == 0) {
            sc.Pb[u] = gb;
            sc.Pf[u] = has_b ? 1u : 0u;
        }
    }
}

template <int D, class Emit>
__global__ __launch_bounds__(256) void seg_finish_kernel(const uint64_t* keys, SegScratch sc, Emit emit) {
    constexpr int L = D / 4;
    constexpr int GPW = 64 / L;
    const int lane = threadIdx.x & 63, lg = lane % L, grp = lane / L;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t nlong = sc.counters[0] < sc.cap ? sc.counters[0] : sc.cap;
    for (uint64_t sidx = wave * GPW + grp; sidx < nlong; sidx += nwaves * GPW) {
        const uint32_t u0 = sc.unit_base[sidx], u1 = sc.unit_base[sidx + 1];
        float4 g = ld4(sc.P + (uint64_t)u0 * D + 4 * lg);
        bool has_b = sc.Pf[u0] != 0;
        float gb = has_b ? sc.Pb[u0] : 0.0f;
        for (uint32_t u = u0 + 1; u < u1; u += 4) { /* chunk partials in order; four loads in flight */
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ld4(sc.P + (uint64_t)(u + i < u1 ? u + i : u1 - 1) * D + 4 * lg);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (u + i < u1) {
                    g.x = g.x + v[i].x; g.y = g.y + v[i].y; g.z = g.z + v[i].z; g.w = g.w + v[i].w;
                    if (sc.Pf[u + i]) {
                        gb = has_b ? gb + sc.Pb[u + i] : sc.Pb[u + i];
                        has_b = true;
                    }
                }
            }
        }
        const uint64_t p = sc.long_start[sidx];
        const uint32_t row = (uint32_t)(keys[p] >> 32);
        emit.template row<D>(row, p, lg, g, has_b, gb, emit.template pre<D>(row, lg));
    }
}

// ---- multi-device: owner-reduce protocol -----------------------------------------
// chunk = [G: S*d f32][gb: S f32][flags: S u32] for a slice of S = ceil(I / ndev) table rows
__device__ __forceinline__ float* chunk_ptr(void* base, uint64_t chunk, uint64_t S, int D) {
    return reinterpret_cast<float*>(base) + chunk * S * ((uint64_t)D + 2);
}
__device__ __forceinline__ const float* chunk_ptr(const void* base, uint64_t chunk, uint64_t S, int D) {
    return reinterpret_cast<const float*>(base) + chunk * S * ((uint64_t)D + 2);
}

__global__ void clear_chunk_flags_kernel(void* buf, int nchunks, uint64_t S, int D) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)nchunks * S) return;
    float* c = chunk_ptr(buf, i / S, S, D);
    reinterpret_cast<uint32_t*>(c + S * D + S)[i % S] = 0u;
}

// owner: contributions of the devices added in device order (first toucher initialises)
template <int D, int NQ>  // NQ: devices the row's requests are unrolled for (ndev <= NQ); flags first, then every touched contribution together
__global__ __launch_bounds__(256) void owner_reduce_kernel(ChunkPtrs recv, int ndev, uint64_t S, void* own) {
    constexpr int L = D / 4;
    constexpr int GPW = 64 / L;
    const int lane = threadIdx.x & 63, lg = lane % L, grp = lane / L;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    float* out = reinterpret_cast<float*>(own);
    uint32_t* ofl = reinterpret_cast<uint32_t*>(out + S * D + S);
    for (uint64_t i0 = wave * GPW; i0 < S; i0 += nwaves * GPW) {  // wave-uniform trip count
        const uint64_t i = i0 + grp;
        uint32_t f[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            f[q] = (q < ndev && i < S) ? reinterpret_cast<const uint32_t*>(reinterpret_cast<const float*>(recv.p[q]) + S * D + S)[i] : 0u;
        float4 v[NQ];
        float vb[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float* c = reinterpret_cast<const float*>(recv.p[q]); /* device q's contribution to this owner's rows */
            v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            vb[q] = 0.0f;
            if (f[q] & 1u) v[q] = ld4(c + i * D + 4 * lg);
            if (f[q] & 2u) vb[q] = c[S * D + i];
        }
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        float gb = 0.0f;
        uint32_t fl = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (f[q] & 1u) {
                if (fl & 1u) { g.x = g.x + v[q].x; g.y = g.y + v[q].y; g.z = g.z + v[q].z; g.w = g.w + v[q].w; }
                else g = v[q];
            }
            if (f[q] & 2u) gb = (fl & 2u) ? gb + vb[q] : vb[q];
            fl |= f[q];
        }
        if (i < S) {
            if (fl & 1u) st4(out + i * D + 4 * lg, g);
            if (lg == 0) {
                if (fl & 2u) out[S * D + i] = gb;
                ofl[i] = fl;
            }
        }
    }
}

// owner-APPLIED update (Parallelism::Synchronous, replicated table; ≙ the one shared HogwildParameter of lstm.rs:259-260, whose
// optimiser state exists once): the same device-order sum as owner_reduce_kernel, and right behind it the ONE optimiser update of
// the row — in place, on the owner's own rows [row0, row0 + nrows) of its replica; the updated PARAMETER slice (not the gradient
// sums) is what the devices then all-gather, straight into every replica's table.  Same sums, same update arithmetic as
// owner_reduce_kernel + table_apply_kernel: same bits; no device walks the whole table any more, and a row's optimiser state
// lives on its owner only.
template <int D, int NQ>  // NQ: devices the row's requests are unrolled for (ndev <= NQ)
__global__ __launch_bounds__(256) void owner_update_kernel(ModelView m, ChunkPtrs recv, int ndev, uint64_t S, uint64_t row0, uint64_t nrows) {
    constexpr int L = D / 4;
    constexpr int GPW = 64 / L;
    const int lane = threadIdx.x & 63, lg = lane % L, grp = lane / L;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const bool adam = m.optimizer == SBR_OPT_ADAM;
    /* A row is TWO dependent round trips: the devices' flags of the row, then — all requested together — the touched devices'
     * contributions, their bias words and the row's own parameter / optimiser-state quads; the adds stay in device order.  (The
     * first form requested a contribution only after the previous device's had been added: up to ten round trips per row.) */
    auto flags_of = [&](uint64_t i, uint32_t (&f)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            f[q] = (q < ndev && i < nrows) ? reinterpret_cast<const uint32_t*>(reinterpret_cast<const float*>(recv.p[q]) + S * D + S)[i] : 0u;
    };
    uint64_t i = wave * GPW + grp;
    uint32_t f[NQ];
    flags_of(i, f);
    for (uint64_t i0 = wave * GPW; i0 < nrows; i0 += nwaves * GPW, i += nwaves * GPW) {  // wave-uniform trip count
        uint32_t fl = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) fl |= f[q];
        float4 v[NQ];
        float vb[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {  /* (f[q] = 0 beyond ndev and past the slice: nothing is read there) */
            const float* c = reinterpret_cast<const float*>(recv.p[q]);
            v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            vb[q] = 0.0f;
            if (f[q] & 1u) v[q] = ld4(c + i * D + 4 * lg);
            if (f[q] & 2u) vb[q] = c[S * D + i];
        }
        const uint64_t row = row0 + i;
        float4 wv = make_float4(0.f, 0.f, 0.f, 0.f), av = wv, mv = wv;
        float bv = 0.0f, ba = 0.0f, bmm = 0.0f;
        if (fl & 1u) {
            wv = ld4(m.E + row * D + 4 * lg);
            av = ld4(m.Eacc + row * D + 4 * lg);
            if (adam) mv = ld4(m.Em + row * D + 4 * lg);
        }
        if ((fl & 2u) && lg == 0) {
            bv = m.b[row]; ba = m.bacc[row];
            if (adam) bmm = m.bm[row];
        }
        uint32_t fcur[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) fcur[q] = f[q];
        flags_of(i + nwaves * GPW, f);  /* the next row's flags travel underneath this row's contributions */
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        float gb = 0.0f;
        uint32_t seen = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (fcur[q] & 1u) {
                if (seen & 1u) { g.x = g.x + v[q].x; g.y = g.y + v[q].y; g.z = g.z + v[q].z; g.w = g.w + v[q].w; }
                else g = v[q];
            }
            if (fcur[q] & 2u) gb = (seen & 2u) ? gb + vb[q] : vb[q];
            seen |= fcur[q];
        }
        if (fl & 1u) {  /* row_update */
            opt_update(m, &wv.x, &av.x, &mv.x, g.x);
            opt_update(m, &wv.y, &av.y, &mv.y, g.y);
            opt_update(m, &wv.z, &av.z, &mv.z, g.z);
            opt_update(m, &wv.w, &av.w, &mv.w, g.w);
            st4(m.E + row * D + 4 * lg, wv);
            st4(m.Eacc + row * D + 4 * lg, av);
            if (adam) st4(m.Em + row * D + 4 * lg, mv);
        }
        if ((fl & 2u) && lg == 0) {  /* bias_update */
            opt_update(m, &bv, &ba, &bmm, gb);
            m.b[row] = bv;
            m.bacc[row] = ba;
            if (adam) m.bm[row] = bmm;
        }
    }
}

// every device: Adagrad on every touched row from the gathered global sums
template <int D>
__global__ __launch_bounds__(256) void table_apply_kernel(ModelView m, ChunkPtrs table, uint64_t S) {
    constexpr int L = D / 4;
    constexpr int GPW = 64 / L;
    const int lane = threadIdx.x & 63, lg = lane % L, grp = lane / L;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t row = wave * GPW + grp; row < m.num_items; row += nwaves * GPW) {
        const float* c = reinterpret_cast<const float*>(table.p[row / S]); /* the reduced chunk of the row's owner */
        const uint64_t lr = row % S;
        const uint32_t fl = reinterpret_cast<const uint32_t*>(c + S * D + S)[lr];
        if (fl & 1u) {
            row_update<D>(m, row, lg, ld4(c + lr * D + 4 * lg), false, 0.0f);
        }
        if (fl & 2u) bias_update(m, row, lg, c[S * D + lr]);
    }
}

// ---- partitioned item table: owner-computes over peer-readable gradient lists -----
// Every device reduces its own entries per table row exactly as sparse_scatter_kernel does, but into a
// LIST addressed by the position of the row's first key in the device's sorted key array
// (G[p][D], gb[p], fl[p]; fl = 0 at non-head positions).  The owner of a row range then reads the
// peers' lists directly (peer mappings over xGMI) and adds the devices' contributions in device
// order — the same association order as the replicated owner-reduce exchange.
// bounds[q] = first position of the sorted keys whose row is >= q * S (q = 0..ndev)
__global__ void owner_bounds_kernel(const uint64_t* keys, uint32_t n, int ndev, uint64_t S, uint32_t* bounds) {
    const int q = threadIdx.x;
    if (q > ndev) return;
    const uint64_t target = ((uint64_t)q * S) << 32;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    bounds[q] = q == ndev ? n : lo;
}

// where owner q's row range lies in every device's sorted keys, from the devices' owner bounds — on the device (one thread)
__global__ void merge_plan_kernel(PeerBounds pb, int ndev, int q, MergePlan* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t total = 0;
    for (int r = 0; r < 16; ++r) {
        if (r < ndev) {
            const uint32_t lo = pb.b[r][q], hi = pb.b[r][q + 1];
            out->lo[r] = lo;
            out->base[r] = total;
            total += hi - lo;
        } else {
            out->lo[r] = 0;
            out->base[r] = total;
        }
    }
    out->base[16] = total;
}

// A row of the merge has at most one entry per device (every device reduced its own entries per row first), so a row is TWO
// dependent round trips: its window of NQ merge keys (with the predecessor's, which says whether the position is a head), then —
// all requested together — the touched devices' list rows, their bias words and flags, and the row's own parameter /
// optimiser-state quads; the adds stay in device order.  (The first form walked key -> list row -> next key -> ... -> table row:
// up to 2 ndev + 2 dependent round trips per row, each of them an xGMI round trip for a peer's list.)
template <int D, int NQ>
__global__ __launch_bounds__(256) void owner_list_apply_kernel(ModelView m, PeerLists pl, const uint64_t* mkeys) {
    constexpr int L = D / 4;
    constexpr int GPW = 64 / L;
    const int lane = threadIdx.x & 63, lg = lane % L, grp = lane / L;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t n = pl.plan->base[16];  /* merge keys of this step (grid-stride: the launch does not depend on it) */
    const bool adam = m.optimizer == SBR_OPT_ADAM;
    for (uint64_t i0 = wave * GPW; i0 < n; i0 += nwaves * GPW) {  // wave-uniform trip count
        const uint64_t i = i0 + grp;
        uint64_t k[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) k[j] = i + j < n ? mkeys[i + j] : ~0ull;
        const uint64_t kprev = (i > 0 && i < n) ? mkeys[i - 1] : ~0ull;
        const uint32_t row = (uint32_t)(k[0] >> 32);
        const bool head = k[0] != ~0ull && (i == 0 || (uint32_t)(kprev >> 32) != row);  /* padding sorts to the end */
        if (!head) continue;
        float4 v[NQ];
        float vb[NQ];
        uint32_t vf[NQ];
        bool in[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {  /* devices in ascending order: the row's keys are consecutive */
            in[j] = k[j] != ~0ull && (uint32_t)(k[j] >> 32) == row;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            vb[j] = 0.0f;
            vf[j] = 0u;
            if (in[j]) {
                const int r = (int)((k[j] >> 28) & 15u);
                const uint32_t p = (uint32_t)(k[j] & 0x0FFFFFFFu);
                v[j] = ld4(pl.G[r] + (size_t)p * D + 4 * lg);
                vf[j] = pl.fl[r][p];
                vb[j] = pl.gb[r][p];
            }
        }
        float4 wv = ld4(m.E + (size_t)row * D + 4 * lg), av = ld4(m.Eacc + (size_t)row * D + 4 * lg);
        float4 mv = adam ? ld4(m.Em + (size_t)row * D + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
        float bv = 0.0f, ba = 0.0f, bmm = 0.0f;
        if (lg == 0) {  /* requested with the rest; used only if some device's entry carries a bias term */
            bv = m.b[row]; ba = m.bacc[row];
            if (adam) bmm = m.bm[row];
        }
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        float gb = 0.0f;
        bool has_b = false, first = true;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (in[j]) {
                if (first) { g = v[j]; first = false; }
                else { g.x = g.x + v[j].x; g.y = g.y + v[j].y; g.z = g.z + v[j].z; g.w = g.w + v[j].w; }
                if (vf[j] & 2u) {
                    gb = has_b ? gb + vb[j] : vb[j];
                    has_b = true;
                }
            }
        }
        /* row_update + bias_update */
        opt_update(m, &wv.x, &av.x, &mv.x, g.x);
        opt_update(m, &wv.y, &av.y, &mv.y, g.y);
        opt_update(m, &wv.z, &av.z, &mv.z, g.z);
        opt_update(m, &wv.w, &av.w, &mv.w, g.w);
        st4(m.E + (size_t)row * D + 4 * lg, wv);
        st4(m.Eacc + (size_t)row * D + 4 * lg, av);
        if (adam) st4(m.Em + (size_t)row * D + 4 * lg, mv);
        if (has_b && lg == 0) {
            opt_update(m, &bv, &ba, &bmm, gb);
            m.b[row] = bv;
            m.bacc[row] = ba;
            if (adam) m.bm[row] = bmm;
        }
    }
}

__global__ void accumulate_loss_kernel(const uint8_t* all_blocks, uint64_t block_bytes, int ndev, double* loss_acc,
                                       unsigned long long* ex_acc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    /* loss_acc: [0] all devices, [1 + q] device q;  ex_acc: [0] examples, [1] negatives scored, [2 + q] examples of device q */
    double l = loss_acc[0];
    unsigned long long e = ex_acc[0], tr = ex_acc[1];
    for (int q = 0; q < ndev; ++q) {
        const uint32_t* hdr = reinterpret_cast<const uint32_t*>(all_blocks + (size_t)q * block_bytes);
        const double lq = *reinterpret_cast<const double*>(hdr + 4);
        const unsigned long long eq = *reinterpret_cast<const unsigned long long*>(hdr + 6);
        l += lq;
        e += eq;
        tr += hdr[1];
        loss_acc[1 + q] += lq;
        ex_acc[2 + q] += eq;
    }
    loss_acc[0] = l;
    ex_acc[0] = e;
    ex_acc[1] = tr;
}

// ------------------------------------------------------------------------------------------------
// Prediction side: bias + chain-order dot (≙ an f32 MFMA accumulation over k)
// ------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ float chain_dot(const float* __restrict__ h, const float* __restrict__ e) {
    float acc = 0.0f;
#pragma unroll
    for (int k4 = 0; k4 < D; k4 += 4) {
        const float4 v = ld4(e + k4);
        acc = sbr_fma(h[k4 + 0], v.x, acc);
        acc = sbr_fma(h[k4 + 1], v.y, acc);
        acc = sbr_fma(h[k4 + 2], v.z, acc);
        acc = sbr_fma(h[k4 + 3], v.w, acc);
    }
    return acc;
}

template <int D>
__global__ void predict_kernel(ModelView m, const float* user, const uint32_t* items, uint64_t n, float* out) {
    __shared__ float hs[D];
    for (int k = threadIdx.x; k < D; k += blockDim.x) hs[k] = user[k];
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t it = items[i];
    out[i] = m.b[it] + chain_dot<D>(hs, m.E + (size_t)it * D);
}

// ------------------------------------------------------------------------------------------------
// K7: full-catalogue scoring + rank (mrr_score, evaluation.rs:27-43) as an f32 MFMA GEMM with a
// rank-count epilogue; the U x I score matrix is never materialised.
//   rank_test_score_kernel : ts[u] = MIN if the test item is in the history, else bias + chain dot
//   rank_gemm_kernel       : S[u][i] = b[i] + sum_k h[u][k] E[i][k] on v_mfma_f32_32x32x2_f32
//                            (k ascending from 0 = the oracle's chain order); counts S >= ts[u]
//   rank_history_kernel    : corrects the count for the (unique) history items, which the reference
//                            masks to f32::MIN (evaluation.rs:30-32)
// A workgroup owns 128 users (4 waves x 32) and a contiguous range of items: the users' states
// stay in registers as MFMA A fragments, 32-item tiles of E stream through LDS.
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ void rank_test_score_kernel(ModelView m, const float* reps, const int* rep_row, uint32_t num_users,
                                       const uint32_t* test_item, const uint32_t* test_in_hist, float* ts, uint32_t* ranks) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= num_users) return;
    const uint32_t ti = test_item[u];
    ts[u] = test_in_hist[u] ? SBR_F32_MIN : m.b[ti] + chain_dot<D>(reps + (size_t)rep_row[u] * D, m.E + (size_t)ti * D);
    ranks[u] = 0;
}

template <int D, int UW>
__global__ __launch_bounds__(256, UW == 1 ? (D <= 128 ? 4 : 2) : (D * UW <= 64 ? 4 : (D * UW <= 128 ? 3 : 2))) void rank_gemm_kernel(ModelView m, const float* reps, const int* rep_row, uint32_t num_users,
                                                                            const float* ts, uint32_t items_per_group, uint32_t* ranks,
                                                                            uint32_t* nonfinite_flag) {
    // UW = 32-user tiles per wave (a workgroup owns 128 * UW users): every staged 32-item tile of E feeds UW x 64 MFMAs per
    // wave between two barriers — UW = 2 halves the barriers, the LDS fills and the L2 traffic per flop at half the waves
    constexpr int LDE = D + 1;
    constexpr int KS = D / 2;  // MFMA k-steps
    constexpr int WGU = 128 * UW;  // users per workgroup
    __shared__ float Es[2][32 * LDE];
    __shared__ float Bs[2][32];
    // thresholds of the workgroup's users in the order the accumulator registers want them: Ts[wave][tile][hh][q] = threshold
    // of user (wave * UW + tile) * 32 + (q & 3) + 8 (q >> 2) + 4 hh — read back per item tile as 16-byte broadcasts instead of
    // living in 16 registers per user tile
    __shared__ float Ts[4][UW][2][16];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int hh = lane >> 5;
    const uint32_t u0 = blockIdx.x * WGU + wave * 32 * UW;
    // A fragments of this wave's users: a[t][s] = h[u0 + 32 t + l31][2 s + hh], held for the whole item range
    float a[UW][KS];
#pragma unroll
    for (int t = 0; t < UW; ++t) {
        const uint32_t u = u0 + 32 * t + l31;
        const float* h = reps + (size_t)rep_row[u < num_users ? u : num_users - 1] * D;
#pragma unroll
        for (int s = 0; s < KS; ++s) a[t][s] = u < num_users ? h[2 * s + hh] : 0.0f;
    }
    for (int e = tid; e < 4 * UW * 32; e += 256) {
        const int w2 = e / (UW * 32), t2 = (e / 32) % UW, h2 = (e >> 4) & 1, q = e & 15;
        const uint32_t u = blockIdx.x * WGU + (w2 * UW + t2) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h2;
        Ts[w2][t2][h2][q] = u < num_users ? ts[u] : 0.0f;
    }
    // per-lane counters of "score >= threshold", two 16-bit counters per register (a lane adds at most one per tile and
    // register: the launcher keeps an item range below 65 536 tiles)
    uint32_t cnt2[UW][8];
#pragma unroll
    for (int t = 0; t < UW; ++t)
#pragma unroll
        for (int q = 0; q < 8; ++q) cnt2[t][q] = 0;
    bool bad = false;
    const uint32_t i_begin = blockIdx.y * items_per_group;
    uint32_t i_end = i_begin + items_per_group;
    if (i_end > m.num_items) i_end = m.num_items;
    const int ntiles = i_begin < i_end ? (int)((i_end - i_begin + 31) / 32) : 0;
    // staging map: thread -> (item row tid/8, float4 columns (tid%8) + 8 j)
    constexpr int NV = 32 * (D / 4);
    constexpr int ITER = (NV + 255) / 256;
    float4 ev[ITER];
    float bv = 0.0f;
    auto fetch = [&](int tile) {
        const uint32_t ib = i_begin + (uint32_t)tile * 32;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int idx = tid + it * 256;
            const int r = idx / (D / 4);
            const int c4 = (idx % (D / 4)) * 4;
            ev[it] = (idx < NV && ib + r < i_end) ? ld4(m.E + (size_t)(ib + r) * D + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < 32) bv = ib + tid < i_end ? m.b[ib + tid] : 0.0f;
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int idx = tid + it * 256;
            if (idx < NV) {
                const int r = idx / (D / 4);
                const int c4 = (idx % (D / 4)) * 4;
                float* dst = &Es[buf][r * LDE + c4];
                dst[0] = ev[it].x; dst[1] = ev[it].y; dst[2] = ev[it].z; dst[3] = ev[it].w;
            }
        }
        if (tid < 32) Bs[buf][tid] = bv;
    };
    if (ntiles > 0) {
        fetch(0);
        stage(0);
    }
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) fetch(tile + 1);
        f32x16 acc[UW];
#pragma unroll
        for (int t = 0; t < UW; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
        const float* eb = &Es[buf][l31 * LDE + hh];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float bvs = eb[2 * s];
#pragma unroll
            for (int t = 0; t < UW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], bvs, acc[t], 0, 0, 0);
        }
        const float bias = Bs[buf][l31];
        const bool item_ok = i_begin + (uint32_t)tile * 32 + l31 < i_end;
#pragma unroll
        for (int t = 0; t < UW; ++t)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 t4 = ld4(&Ts[wave][t][hh][4 * q4]);
                const float tq[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = 4 * q4 + j;
                    const float sc = bias + acc[t][q];
                    if (item_ok) {
                        if (!(sc - sc == 0.0f)) bad = true;
                        if (sc >= tq[j]) cnt2[t][q >> 1] += (q & 1) ? 0x10000u : 1u;
                    }
                }
            }
        if (tile + 1 < ntiles) stage(buf ^ 1);
        __syncthreads();
    }
    // per-user totals: sum over the 32 item lanes of each half-wave
#pragma unroll
    for (int t = 0; t < UW; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int c = (int)((cnt2[t][q >> 1] >> ((q & 1) * 16)) & 0xFFFFu);
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
            const uint32_t u = u0 + 32 * t + (q & 3) + 8 * (q >> 2) + 4 * hh;
            if (l31 == 0 && u < num_users && c) atomicAdd(&ranks[u], (uint32_t)c);
        }
    if (__any(bad) && lane == 0) atomicOr(nonfinite_flag, 1u);
}

template <int D>
__global__ __launch_bounds__(64) void rank_history_kernel(ModelView m, const float* reps, const int* rep_row, const float* ts,
                                                          const uint64_t* hist_ptr, const uint32_t* hist_items, uint32_t* ranks) {
    const int u = blockIdx.x;
    const float* h = reps + (size_t)rep_row[u] * D;
    const float t = ts[u];
    int cnt = 0;
    for (uint64_t e = hist_ptr[u] + threadIdx.x; e < hist_ptr[u + 1]; e += 64) {
        const uint32_t i = hist_items[e];
        const float s = m.b[i] + chain_dot<D>(h, m.E + (size_t)i * D);
        if (s >= t) --cnt;               /* it was counted by the GEMM pass ...          */
        if (SBR_F32_MIN >= t) ++cnt;     /* ... but the masked value only counts against MIN */
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (threadIdx.x == 0 && cnt) atomicAdd(&ranks[u], (uint32_t)cnt); /* two's complement: adds a negative delta */
}

// ------------------------------------------------------------------------------------------------
// numerics self-tests
// ------------------------------------------------------------------------------------------------
__global__ void selftest_math_kernel(const float* x, float* e, float* s, float* t, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    {   /* one LSTM cell (shared-division gate activations): h of (zi, zf, zg, zo) = (x, x/2, -x, x/4 + 1), c_prev = 1/2 */
        float gi, gf, gg, go, cc, hh;
        sbr_lstm_cell_fwd(x[i], 0.5f * x[i], -x[i], sbr_fma(0.25f, x[i], 1.0f), 0.5f, 0, &gi, &gf, &gg, &go, &cc, &hh);
        e[i] = hh;
    }
    s[i] = sbr_sigmoidf(x[i]);
    t[i] = sbr_tanhf(x[i]);
}
template <int D>
__global__ void selftest_dot_tree_kernel(const float* x, const float* y, uint64_t nrows, float* out) {
    constexpr int L = D / 4;
    constexpr int GPW = 64 / L;
    const int lane = threadIdx.x & 63, lg = lane % L, grp = lane / L;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t r = wave * GPW + grp;
    const uint64_t rr = r < nrows ? r : nrows - 1;
    const float v = group_allreduce<L>(dot4(ld4(x + rr * D + 4 * lg), ld4(y + rr * D + 4 * lg)));
    if (r < nrows && lg == 0) out[r] = v;
}
// C[16][16] = c0 + A[16][K] B[K][16] by one wave on v_mfma_f32_16x16x4_f32
__global__ void selftest_mfma_chain_kernel(const float* a, const float* b, const float* c0, int K, float* out) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, kq = lane >> 4;
    f32x4 acc;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) acc[reg] = c0[(kq * 4 + reg) * 16 + j];
    for (int k0 = 0; k0 < K; k0 += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j * K + k0 + kq], b[(k0 + kq) * 16 + j], acc, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) out[(kq * 4 + reg) * 16 + j] = acc[reg];
}
// C[32][32] = A[32][K] B[K][32] by one wave on v_mfma_f32_32x32x2_f32 (rows of `out` follow the
// dense-gradient kernel's register map)
__global__ void selftest_mfma32_chain_kernel(const float* a, const float* b, int K, float* out) {
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, hh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[l31 * K + k0 + hh], b[(k0 + hh) * 32 + l31], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) out[((q & 3) + 8 * (q >> 2) + 4 * hh) * 32 + l31] = acc[q];
}

// ================================================================================================
// launchers
// ================================================================================================
#define DISPATCH_D(d, ...)                                         \
    switch (d) {                                                   \
        case 16: { constexpr int DD = 16; __VA_ARGS__; } break;    \
        case 32: { constexpr int DD = 32; __VA_ARGS__; } break;    \
        case 64: { constexpr int DD = 64; __VA_ARGS__; } break;    \
        case 128: { constexpr int DD = 128; __VA_ARGS__; } break;  \
        case 256: { constexpr int DD = 256; __VA_ARGS__; } break;  \
        default: break;                                            \
    }

/* Resident grids of the memory-bound grid-stride kernels: SBR_RESIDENT_WG_PER_CU workgroups of 256 threads per CU.  Eight
 * would fill every wave slot of the chip; seven leave one wave per SIMD for whatever runs beside them on another stream
 * (the key sort's short kernels could otherwise only advance at the main stream's kernel boundaries). */
#ifndef SBR_RESIDENT_WG_PER_CU
#define SBR_RESIDENT_WG_PER_CU 7
#endif
static inline int grid_for_groups(long long groups, int groups_per_block) {
    constexpr int per_cu = SBR_RESIDENT_WG_PER_CU;
    long long g = (groups + groups_per_block - 1) / groups_per_block;
    if (g < 1) g = 1;
    if (g > 256 * per_cu) g = 256 * per_cu;
    return (int)g;
}

/* The step's cache policy by size (profiles/r05_streaming_gathers.md): while the step's h rows fit the Infinity Cache beside the rows
 * the forward pass gathers (131 + 131 MB of 256 at the headline batch) the forward pass streams its other stores past the caches and
 * the score kernels read their rows nt; at 50 000 sequences per step (838 MB of h rows) both lose (forward +5 %, score +6 %). */
#ifndef SBR_STREAM_MAX_H_BYTES
#define SBR_STREAM_MAX_H_BYTES (192ull << 20)
#endif
static bool stream_policy(int rows, int d) {
    const char* e = std::getenv("SBR_STREAM"); /* test hook, read per call: "0" / "1" force either form (same bits: a cache hint) */
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
    return (unsigned long long)rows * (unsigned long long)d * 4ull <= SBR_STREAM_MAX_H_BYTES;
}
void launch_recurrent_forward(const ModelView& m, const MbView& mb, float* H, const WorkView& w_plan, int tm_host,
                              const int* off_host, hipStream_t s) {
    if (mb.R == 0) return;
    WorkView w = w_plan;
    w.stream_activations = stream_policy(mb.R, m.d) ? 1 : 0;
    if (m.ng == 0) {
        DISPATCH_D(m.d, {
            const int gpb = 4 * (64 / (DD / 4));
            hipLaunchKernelGGL((ewma_forward_kernel<DD>), dim3(grid_for_groups(mb.B, gpb)), dim3(256), 0, s, m, mb, H);
        });
        return;
    }
    if (launch_wave_forward(m, mb, H, w, tm_host, s)) return; /* small minibatch at d <= 32: one wave per sequence (sbr_wave.hip) */
    /* d = 256 keeps the per-step launches here: 16 unit tiles x 32-row tiles fill the chip at every step, and the
     * resident form would stream the 2 MiB of weights per 32 rows (measured 8.3 vs 7.9 ms at 20 000 sequences); the
     * BPTT kernel below does take the resident form at d = 256 (7.3 vs 9.3 ms) */
    if (m.d <= 128 && tm_host <= SBR_MAX_T) { /* sequence-resident kernel: one launch for all time steps */
        DISPATCH_D(m.d, {
            if constexpr (DD <= 128) {
                constexpr int UPW = 1;
                /* 32-sequence tiles (two workgroups per CU) or 64-sequence tiles (one): see SBR_FWD_RT4_MIN_TILES */
                const char* rt_str = std::getenv("SBR_SEQ_RT"); /* read per call: the tests force either form */
                const int rt_env = rt_str ? std::atoi(rt_str) : 0;
                const bool big = rt_env ? rt_env >= 4 : (mb.B + 63) / 64 >= SBR_FWD_RT4_MIN_TILES;
                /* 16-sequence tiles for small minibatches: below SBR_SEQ_RT1_MAX_TILES 32-sequence tiles the chip is not
                 * full and the kernel time is the longest tile's chain of dependent steps — half-size tiles halve the step */
                const bool tiny = rt_env ? rt_env == 1 : (mb.B + 31) / 32 < SBR_SEQ_RT1_MAX_TILES;
                if (tiny) {  /* every d <= 128: at d = 32 a 16-row step is 64 MFMAs per wave instead of 128 */
                    constexpr int RT = 1;
                    const int ntiles = (mb.B + 16 * RT - 1) / (16 * RT);
                    if (m.ng == 4)
                        hipLaunchKernelGGL((lstm_fwd_seq_kernel<DD, 4, RT, UPW>), dim3(ntiles), dim3((DD / 16 / UPW) * 64), 0, s, m, mb, H, w, ntiles);
                    else
                        hipLaunchKernelGGL((lstm_fwd_seq_kernel<DD, 3, RT, UPW>), dim3(ntiles), dim3((DD / 16 / UPW) * 64), 0, s, m, mb, H, w, ntiles);
                } else if (big && DD >= 64) {
                    constexpr int RT = 4;
                    const int ntiles = (mb.B + 16 * RT - 1) / (16 * RT);
                    if (m.ng == 4)
                        hipLaunchKernelGGL((lstm_fwd_seq_kernel<DD, 4, RT, UPW>), dim3(ntiles), dim3((DD / 16 / UPW) * 64), 0, s, m, mb, H, w, ntiles);
                    else
                        hipLaunchKernelGGL((lstm_fwd_seq_kernel<DD, 3, RT, UPW>), dim3(ntiles), dim3((DD / 16 / UPW) * 64), 0, s, m, mb, H, w, ntiles);
                } else {
                    constexpr int RT = 2;
                    const int ntiles = (mb.B + 16 * RT - 1) / (16 * RT);
                    if (m.ng == 4)
                        hipLaunchKernelGGL((lstm_fwd_seq_kernel<DD, 4, RT, UPW>), dim3(ntiles), dim3((DD / 16 / UPW) * 64), 0, s, m, mb, H, w, ntiles);
                    else
                        hipLaunchKernelGGL((lstm_fwd_seq_kernel<DD, 3, RT, UPW>), dim3(ntiles), dim3((DD / 16 / UPW) * 64), 0, s, m, mb, H, w, ntiles);
                }
            }
        });
        return;
    }
    for (int t = 0; t < tm_host; ++t) {
        const int nblk = (off_host[t + 1] - off_host[t] + 31) / 32; /* 32-row tiles alive at step t */
        DISPATCH_D(m.d, {
            if (m.ng == 4)
                hipLaunchKernelGGL((lstm_fwd_step_kernel<DD, 4>), dim3(nblk, DD / 16), dim3(256), 0, s, m, mb, t, H, w);
            else
                hipLaunchKernelGGL((lstm_fwd_step_kernel<DD, 3>), dim3(nblk, DD / 16), dim3(192), 0, s, m, mb, t, H, w);
        });
    }
}

/* WARP: rows per lane group in flight — two below this many packed rows per launch (latency-bound launches), one above
 * (bandwidth-bound; SBR_SCORE_U = 1 / 2 forces one form: the tests run both) */
#ifndef SBR_SCORE_U2_MAX_ROWS
#define SBR_SCORE_U2_MAX_ROWS 700000
#endif
static int score_warp_u(int rows) {
    const char* e = std::getenv("SBR_SCORE_U"); /* read per call */
    if (e && (e[0] == '1' || e[0] == '2')) return e[0] - '0';
    return rows < SBR_SCORE_U2_MAX_ROWS ? 2 : 1;
}
static int score_grid(int d, int rows, bool single_negative) {
    const int gpb = 4 * (64 / (d / 4)) * (single_negative ? SBR_SCORE_SINGLE_U : score_warp_u(rows));  // rows per workgroup and pass
    return grid_for_groups(rows, gpb);
}

bool small_tail_shape_ok(const ModelView& m, int sequences_host, int rows_host, bool wide) {
    return sequences_host == 1 && rows_host > 0 && rows_host <= SBR_SMALL_TAIL_MAX_ROWS && (m.d == 16 || m.d == 32 || wide);
}

void launch_score(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, uint64_t epoch_key,
                  int rows_host, hipStream_t s, const SmallTail* tail) {
    if (rows_host > 0) {
        DISPATCH_D(m.d, {
            if constexpr (DD <= 32) {
                if (tail) { /* one workgroup: its score pass, then the step's bookkeeping and key ordering (SmallTail) */
                    if (rows_host <= 64) hipLaunchKernelGGL((score_tail_kernel<DD, 256>), dim3(1), dim3(256), 0, s, m, mb, blk, w, epoch_key, *tail);
                    else hipLaunchKernelGGL((score_tail_kernel<DD, 1024>), dim3(1), dim3(1024), 0, s, m, mb, blk, w, epoch_key, *tail);
                    return;
                }
            }
            if (m.loss == SBR_LOSS_WARP) {
                const bool nt = stream_policy(rows_host, DD);
                if (score_warp_u(rows_host) == 2) {
                    if (nt) hipLaunchKernelGGL((score_kernel<DD, 2, false, true>), dim3(score_grid(DD, rows_host, false)), dim3(256), 0, s, m, mb, blk, w, epoch_key);
                    else hipLaunchKernelGGL((score_kernel<DD, 2, false, false>), dim3(score_grid(DD, rows_host, false)), dim3(256), 0, s, m, mb, blk, w, epoch_key);
                } else {
                    if (nt) hipLaunchKernelGGL((score_kernel<DD, 1, true, true>), dim3(score_grid(DD, rows_host, false)), dim3(256), 0, s, m, mb, blk, w, epoch_key);
                    else hipLaunchKernelGGL((score_kernel<DD, 1, true, false>), dim3(score_grid(DD, rows_host, false)), dim3(256), 0, s, m, mb, blk, w, epoch_key);
                }
            }
            else
                if (stream_policy(rows_host, DD))
                    hipLaunchKernelGGL((score_single_kernel<DD, SBR_SCORE_SINGLE_U, true>), dim3(score_grid(DD, rows_host, true)), dim3(256), 0, s, m, mb, blk, w,
                                       epoch_key);
                else
                    hipLaunchKernelGGL((score_single_kernel<DD, SBR_SCORE_SINGLE_U, false>), dim3(score_grid(DD, rows_host, true)), dim3(256), 0, s, m, mb, blk, w,
                                       epoch_key);
        });
    }
}

/* EWMA + single-negative loss: scan, scores and backward scan of a sequence in one pass (ewma_seq_kernel).  The grid is
 * launch_score's, so that launch_block_header finds the same number of loss partials.
 * A sequence's backward scan re-reads what its forward scan touched moments earlier — the target
 * and negative rows and its own h rows, ~3 x 4d bytes per step — and whether those re-reads come from the Infinity Cache or from
 * HBM is a matter of how many sequences are in flight between a row's two uses.  With every wave slot taken (7 workgroups per CU,
 * ~7 000 sequences at d = 256: ~1 GB between the two uses) they all come from HBM and the one-pass form only saves a launch (rounds 4-5 therefore ran the backward scan as its own launch); on TWO
 * workgroups per CU (SBR_EWMA_WHOLE_WG = 512: ~200 MB in flight) a good part of them hits: step 6.28 -> 5.73-5.95 ms at d = 256 /
 * 1e7 items / 50 000 sequences, 4.09 -> 3.59 ms at d = 128 / 1e6 items; one workgroup per CU has too few requests in flight (6.4-6.5 ms),
 * three (768) 5.8-5.9 (profiles/r06_ewma_bytes.md).  Same arithmetic, same bits. */
#ifndef SBR_EWMA_WHOLE_WG
#define SBR_EWMA_WHOLE_WG 512
#endif
void launch_ewma_sequences(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, uint64_t epoch_key,
                           int rows_host, hipStream_t s, const SmallTail* tail) {
    if (rows_host <= 0) return;
    const SmallTail none{};
    DISPATCH_D(m.d, {
        if constexpr (DD <= 32) {
            if (tail) { /* one sequence: scan, score, backward scan and the step's bookkeeping + key ordering in one workgroup's launch */
                hipLaunchKernelGGL((ewma_seq_kernel<DD, true>), dim3(1), dim3(256), 0, s, m, mb, blk, w, epoch_key, *tail, 1);
                return;
            }
        }
        const int grid = score_grid(DD, rows_host, true);
        hipLaunchKernelGGL((ewma_seq_kernel<DD, false>), dim3(grid), dim3(256), 0, s, m, mb, blk, w, epoch_key, none, grid < SBR_EWMA_WHOLE_WG ? grid : SBR_EWMA_WHOLE_WG);
    });
}

void launch_materialize_dh(const ModelView& m, const BlockView& blk, int rows_host, float* dH, hipStream_t s) {
    if (rows_host == 0) return;
    DISPATCH_D(m.d, {
        hipLaunchKernelGGL((materialize_dh_kernel<DD>), dim3((rows_host * (DD / 4) + 255) / 256), dim3(256), 0, s, m, blk, rows_host, dH);
    });
}

void launch_block_header(const ModelView& m, const BlockView& blk, const WorkView& w, const MbView& mb, int rows_host, double* loss_acc,
                         unsigned long long* ex_acc, float* lag_state, hipStream_t s) {
    launch_block_header_parts(blk.header, rows_host, w.part_loss, w.part_tries,
                              rows_host > 0 ? score_grid(m.d, rows_host, m.loss != SBR_LOSS_WARP) : 0, loss_acc, ex_acc, mb, w.loss, lag_state, s);
}

void launch_recurrent_backward(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w,
                               int tm_host, int rows_host, int b_host, const int* off_host, hipStream_t s) {
    if (rows_host == 0) {
        const size_t n = m.ng ? (size_t)(2 * m.d + 1) * m.ng * m.d : (size_t)m.d;
        (void)hipMemsetAsync(blk.dense, 0, n * sizeof(float), s);
        return;
    }
    if (m.ng == 0) {
        DISPATCH_D(m.d, {
            const int gpb = 4 * (64 / (DD / 4));
            hipLaunchKernelGGL((ewma_backward_kernel<DD>), dim3(grid_for_groups(b_host, gpb)), dim3(256), 0, s, m, mb, blk, w);
        });
        return; /* dalpha: launch_dense_gradient (the side stream, beside the sparse update) */
    }
    if (launch_wave_backward(m, mb, blk, w, tm_host, b_host, s)) return; /* small minibatch at d <= 32 (sbr_wave.hip) */
    bool stepwise = true;
    /* d = 256: 16 waves per workgroup leave 128 registers per wave, i.e. only the 32-sequence form, one workgroup per CU;
     * its time has a floor of ~4.8 ms (the longest tile's 63 dependent steps), so below ~320 tiles the per-step launches
     * are faster (ms resident / per-step at 8 192, 12 288, 16 384, 20 000 sequences: 4.84 / 4.32, 4.83 / 5.74, 5.27 / 6.92,
     * 7.0 / 9.3) */
    const char* min_tiles_env = std::getenv("SBR_BWD256_MIN_TILES"); /* tests force either form */
    const int min_tiles_256 = min_tiles_env ? std::atoi(min_tiles_env) : SBR_BWD256_MIN_TILES;
    const bool resident = m.d <= 128 || (m.d == 256 && (b_host + 31) / 32 >= min_tiles_256);
    if (resident && tm_host <= SBR_MAX_T) { /* sequence-resident BPTT: one launch for all time steps */
        DISPATCH_D(m.d, {
            if constexpr (DD <= 256) {
                const char* rt_str = std::getenv("SBR_SEQ_RT"); /* read per call: the tests force either form */
                const int rt_env = rt_str ? std::atoi(rt_str) : 0;
                const bool big = rt_env ? rt_env >= 4 : (b_host + 63) / 64 >= SBR_BWD_RT4_MIN_TILES;
                bool launched = false;
                if constexpr (DD >= 64 && DD <= 128) {  /* d = 256: 16 waves per workgroup leave 128 registers per wave */
                    if (big) { /* 64-sequence tiles, one workgroup per CU (see launch_recurrent_forward) */
                        const int ntiles = (b_host + 63) / 64;
                        if (m.ng == 4)
                            hipLaunchKernelGGL((lstm_bwd_seq_kernel<DD, 4, 4>), dim3(ntiles), dim3((DD / 16) * 64), 0, s, m, mb, blk, w);
                        else
                            hipLaunchKernelGGL((lstm_bwd_seq_kernel<DD, 3, 4>), dim3(ntiles), dim3((DD / 16) * 64), 0, s, m, mb, blk, w);
                        launched = true;
                    }
                }
                if constexpr (DD <= 128) {
                    const bool tiny = rt_env ? rt_env == 1 : (b_host + 31) / 32 < SBR_SEQ_RT1_MAX_TILES;
                    if (tiny && !launched) { /* 16-sequence tiles (see launch_recurrent_forward) */
                        const int ntiles = (b_host + 15) / 16;
                        if (m.ng == 4)
                            hipLaunchKernelGGL((lstm_bwd_seq_kernel<DD, 4, 1>), dim3(ntiles), dim3((DD / 16) * 64), 0, s, m, mb, blk, w);
                        else
                            hipLaunchKernelGGL((lstm_bwd_seq_kernel<DD, 3, 1>), dim3(ntiles), dim3((DD / 16) * 64), 0, s, m, mb, blk, w);
                        launched = true;
                    }
                }
                if (!launched) {
                    const int ntiles = (b_host + 31) / 32;
                    if (m.ng == 4)
                        hipLaunchKernelGGL((lstm_bwd_seq_kernel<DD, 4, 2>), dim3(ntiles), dim3((DD / 16) * 64), 0, s, m, mb, blk, w);
                    else
                        hipLaunchKernelGGL((lstm_bwd_seq_kernel<DD, 3, 2>), dim3(ntiles), dim3((DD / 16) * 64), 0, s, m, mb, blk, w);
                }
                stepwise = false;
            }
        });
    }
    for (int t = stepwise ? tm_host - 1 : -1; t >= 0; --t) {
        const int bt = off_host[t + 1] - off_host[t];
        DISPATCH_D(m.d, {
            const int cell_blocks = (bt * (DD / 4) + 255) / 256;
            constexpr int CTW = 2 * DD / 16 < 4 ? 2 * DD / 16 : 4;
            if (m.ng == 4) {
                constexpr int ROWS = 16 * BwdCfg<DD, 4>::RT;
                hipLaunchKernelGGL((lstm_bwd_cell_kernel<DD, 4>), dim3(cell_blocks), dim3(256), 0, s, m, mb, t, blk, w);
                hipLaunchKernelGGL((lstm_bwd_gemm_kernel<DD, 4>), dim3((bt + ROWS - 1) / ROWS, (2 * DD / 16) / CTW), dim3(256), 0, s, m, mb, t, blk, w);
            } else {
                constexpr int ROWS = 16 * BwdCfg<DD, 3>::RT;
                hipLaunchKernelGGL((lstm_bwd_cell_kernel<DD, 3>), dim3(cell_blocks), dim3(256), 0, s, m, mb, t, blk, w);
                hipLaunchKernelGGL((lstm_bwd_gemm_kernel<DD, 3>), dim3((bt + ROWS - 1) / ROWS, (2 * DD / 16) / CTW), dim3(256), 0, s, m, mb, t, blk, w);
            }
        });
    }
}

int launch_dense_gradient(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w_in, int rows_host,
                          int b_host, hipStream_t s, bool defer_reduce) {
    if (rows_host == 0) return 0; /* the empty case is handled by launch_recurrent_backward */
    if (m.ng == 0) { /* EWMA: dalpha from the per-sequence partials the backward scan left in w.dab */
        const int nch = (b_host + EWMA_CHUNK_SEQS - 1) / EWMA_CHUNK_SEQS;
        if (nch == 1) {
            hipLaunchKernelGGL(ewma_dab_final_kernel, dim3(1), dim3(m.d < 64 ? 64 : m.d), 0, s, w_in.dab, b_host, m.d, m.alpha, blk.dense);
            return 0;
        }
        hipLaunchKernelGGL(ewma_dab_chunk_kernel, dim3(nch), dim3(m.d < 64 ? 64 : m.d), 0, s, w_in.dab, b_host, m.d, w_in.partials);
        hipLaunchKernelGGL(ewma_dense_final_kernel, dim3(1), dim3(m.d < 64 ? 64 : m.d), 0, s, w_in.partials, nch, m.d, m.alpha, blk.dense);
        return 0;
    }
    const int nch = (rows_host + SBR_DW_CHUNK_ROWS - 1) / SBR_DW_CHUNK_ROWS;
    const int K2 = 2 * m.d, NGD = m.ng * m.d;
    const int tiles = ((K2 + 127) / 128) * ((NGD + 127) / 128);
    /* a single chunk: its partial IS the gradient — the GEMM writes it straight into the block and the reduction launch is skipped */
    WorkView w = w_in;
    if (nch == 1) w.partials = blk.dense;
    const bool block_form = launch_wave_dense_gradient(m, mb, blk, w, rows_host, s); /* small step at d <= 32 (sbr_wave.hip) */
    if (!block_form) DISPATCH_D(m.d, {
        const unsigned grid = (unsigned)(((nch + 7) / 8) * tiles * 8); /* chunk groups of 8 (one chunk per XCD) x tiles */
        constexpr bool full4 = (2 * DD) % 128 == 0 && (4 * DD) % 128 == 0, full3 = (2 * DD) % 128 == 0 && (3 * DD) % 128 == 0;
        /* the full-tile kernel reads dZ up to the end of the last chunk: through buffer resources (d >= 128, H below 2 GiB) those rows
         * are out of range and read as zeros; on its 64-bit address path they are cleared here */
        const bool buffer_path = DD >= 128 && !w.wide_addresses && (size_t)rows_host * DD * 4 < ((size_t)1 << 31);
        const size_t pad_rows = buffer_path ? 0 : (size_t)nch * SBR_DW_CHUNK_ROWS - (size_t)rows_host;
        constexpr int lds_pad = SBR_DW_LDS_PAD; /* (a residency cap through extra dynamic LDS — three workgroups per CU instead of four, room for
                                    * the sparse update's waves beside them — measured and dropped: profiles/r03_dw_experiments.md) */
        if (m.ng == 4) {
            if constexpr (full4) {
                if (pad_rows) (void)hipMemsetAsync(w.dZ + (size_t)rows_host * NGD, 0, pad_rows * NGD * sizeof(float), s);
                hipLaunchKernelGGL((lstm_dw_full_kernel<DD, 4>), dim3(grid), dim3(256), lds_pad, s, m, mb, blk, w);
            } else {
                hipLaunchKernelGGL((lstm_dw_kernel<DD, 4>), dim3(grid), dim3(256), 0, s, m, mb, blk, w);
            }
        } else {
            if constexpr (full3) {
                if (pad_rows) (void)hipMemsetAsync(w.dZ + (size_t)rows_host * NGD, 0, pad_rows * NGD * sizeof(float), s);
                hipLaunchKernelGGL((lstm_dw_full_kernel<DD, 3>), dim3(grid), dim3(256), 0, s, m, mb, blk, w);
            } else {
                hipLaunchKernelGGL((lstm_dw_kernel<DD, 3>), dim3(grid), dim3(256), 0, s, m, mb, blk, w);
            }
        }
    });
    if (nch == 1) return 0;
    if (defer_reduce) return nch; /* the consumer reduces: launch_dense_reduce, or launch_dense_reduce_apply in the optimiser step's launch */
    launch_dense_reduce(m, w, nch, blk, s);
    return 0;
}

/* ordered sum of `nchunks` chunk partials of the LSTM dense gradient into blk.dense */
void launch_dense_reduce(const ModelView& m, const WorkView& w, int nchunks, const BlockView& blk, hipStream_t s) {
    const size_t n = (size_t)(2 * m.d + 1) * m.ng * m.d;
    hipLaunchKernelGGL(dense_reduce_local_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w.partials, nchunks, n, blk.dense);
}

void launch_dense_apply(const ModelView& m, const uint8_t* all_blocks, uint64_t block_bytes, uint64_t dense_off,
                        int ndev, hipStream_t s) {
    const size_t n = m.ng ? (size_t)(2 * m.d + 1) * m.ng * m.d : (size_t)m.d;
    hipLaunchKernelGGL(dense_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, m, all_blocks, block_bytes, dense_off, ndev, n,
                       (const float*)nullptr, 0, (float*)nullptr);
}

/* one device: ordered reduction of the chunk partials + optimiser update of the dense parameters in one launch (blk.dense is
 * written as well: the debug fetch and the tests read it) */
void launch_dense_reduce_apply(const ModelView& m, const WorkView& w, int nchunks, const BlockView& blk, hipStream_t s) {
    const size_t n = (size_t)(2 * m.d + 1) * m.ng * m.d;
    hipLaunchKernelGGL(dense_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, m, (const uint8_t*)nullptr, (uint64_t)0, (uint64_t)0, 1, n,
                       (const float*)w.partials, nchunks, blk.dense);
}

void launch_repack_lstm(const ModelView& m, hipStream_t s) {
    if (!m.ng) return;
    const size_t nw = (size_t)2 * m.d * m.ng * m.d;
    hipLaunchKernelGGL(repack_lstm_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, m);
}

#ifndef SBR_SEG_INLINE_MAX_KEYS
#define SBR_SEG_INLINE_MAX_KEYS 4096 /* up to this many keys the sparse reduction is the single launch of seg_short_kernel<.., true> */
#endif
// sorted keys -> per-row reduction (chunked order) -> Emit: short segments, then the long ones
template <class Emit>
static void launch_seg_reduce(int d, const BlockView& blk, uint32_t rows_host, const uint64_t* keys_sorted, const SegScratch& sc,
                              const Emit& emit, hipStream_t s) {
    if (rows_host == 0) return;
    const uint64_t total = 3ull * rows_host;
    if (total <= SBR_SEG_INLINE_MAX_KEYS) { /* small step: one launch, long segments reduced in place */
        DISPATCH_D(d, {
            const int gpb = 4 * (64 / (DD / 4));
            hipLaunchKernelGGL((seg_short_kernel<DD, Emit, true>), dim3(grid_for_groups((long long)total / 2 + 1, gpb)), dim3(256), 0, s, blk,
                               keys_sorted, total, sc, emit);
        });
        return;
    }
    if (!sc.prelisted) (void)hipMemsetAsync(sc.counters, 0, 2 * sizeof(uint32_t), s);
    DISPATCH_D(d, {
        const int gpb = 4 * (64 / (DD / 4));
        /* 4 workgroups per CU: with 8 the update's waves fill the register file and the dense-gradient GEMM on the side
         * stream cannot become resident beside it (measured: 14.04 ms per step at 2048, 13.92 at 1024; the update alone
         * takes the same 1.13-1.2 ms either way) */
        constexpr int seg_grid_cap = 1024;
        int seg_grid = grid_for_groups((long long)total / 2 + 1, gpb);
        if (seg_grid > seg_grid_cap) seg_grid = seg_grid_cap;
        hipLaunchKernelGGL((seg_short_kernel<DD, Emit>), dim3(seg_grid), dim3(256), 0, s, blk, keys_sorted, total, sc, emit);
        if (sc.prelisted) return; /* the listed segments: launch_seg_hot_apply, on another stream */
        hipLaunchKernelGGL(seg_units_kernel, dim3(1), dim3(256), 0, s, sc);
        hipLaunchKernelGGL((seg_chunk_kernel<DD>), dim3(1024), dim3(256), 0, s, blk, keys_sorted, sc);
        hipLaunchKernelGGL((seg_finish_kernel<DD, Emit>), dim3(64), dim3(256), 0, s, keys_sorted, sc, emit);
    });
}

#ifndef SBR_SMALL_BACK_MAX_ROWS
#define SBR_SMALL_BACK_MAX_ROWS 256
#endif
bool small_back_shape_ok(const ModelView& m, int rows_host) {
    /* up to 128 rows: a row costs every element's thread one dependent fma and a pair of loads — ms per fit of the reference's
     * Criterion bench (2 352 / 588 / 147 / 37 steps of ~9 / 36 / 150 / 600 rows), this form against the three launches:
     * 74.5 / 90.1, 34.6 / 37.6, 14.6 / 13.7, 7.7 / 5.6 */
    if ((m.d != 16 && m.d != 32) || rows_host <= 0) return false;
    if (m.ng == 0) return 3ull * (uint64_t)rows_host <= SBR_SEG_INLINE_MAX_KEYS; /* EWMA: dalpha + its update + the sparse update; the caller checks
                                                                                    * that the sequences fit one dalpha chunk */
    return rows_host <= SBR_SMALL_BACK_MAX_ROWS;
}
void launch_small_back(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, uint32_t rows_host,
                       const uint64_t* keys_sorted, const SegScratch& sc, hipStream_t s) {
    const size_t n = m.ng ? (size_t)(2 * m.d + 1) * m.ng * m.d : (size_t)m.d;
    const int dw_blocks = (int)((n + 255) / 256);
    const uint64_t total = 3ull * rows_host;
    DISPATCH_D(m.d, {
        if constexpr (DD <= 32) {
            const int gpb = 4 * (64 / (DD / 4));
            const int seg_blocks = grid_for_groups((long long)total / 2 + 1, gpb);
            const size_t lds = m.ng ? ((size_t)rows_host * m.ng * DD + (size_t)(256 / (m.ng * DD) + 2) * rows_host) * 4 : 0;
            static std::atomic<size_t> granted[64]; /* dynamic LDS beyond 64 KB is granted per kernel and device, once */
            int dev = 0;
            (void)hipGetDevice(&dev);
            dev = dev >= 0 && dev < 64 ? dev : 0;
            if (lds > granted[dev]) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(small_back_kernel<DD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                granted[dev] = lds;
            }
            hipLaunchKernelGGL((small_back_kernel<DD>), dim3(dw_blocks + seg_blocks), dim3(256), lds, s, m, mb, blk, w, keys_sorted, total, sc, dw_blocks);
        }
    });
}

void launch_seg_prelist(const SegScratch& sc, hipStream_t s) {
    (void)hipMemsetAsync(sc.counters, 0, 2 * sizeof(uint32_t), s);
    hipLaunchKernelGGL(seg_long_list_kernel, dim3(256), dim3(256), 0, s, sc);
    hipLaunchKernelGGL(seg_units_kernel, dim3(1), dim3(256), 0, s, sc);
}
void launch_seg_hot_apply(const ModelView& m, const BlockView& blk, const uint64_t* keys_sorted, const SegScratch& sc, hipStream_t s) {
    DISPATCH_D(m.d, {
        hipLaunchKernelGGL((seg_chunk_kernel<DD>), dim3(1024), dim3(256), 0, s, blk, keys_sorted, sc);
        hipLaunchKernelGGL((seg_finish_kernel<DD, EmitApply>), dim3(64), dim3(256), 0, s, keys_sorted, sc, EmitApply{m});
    });
}

void launch_seg_apply(const ModelView& m, const BlockView& blk, uint32_t rows_host, const uint64_t* keys_sorted,
                      const SegScratch& sc, hipStream_t s) {
    launch_seg_reduce(m.d, blk, rows_host, keys_sorted, sc, EmitApply{m}, s);
}

void launch_accumulate_loss(const uint8_t* all_blocks, uint64_t block_bytes, int ndev, double* loss_acc,
                            unsigned long long* ex_acc, hipStream_t s) {
    hipLaunchKernelGGL(accumulate_loss_kernel, dim3(1), dim3(64), 0, s, all_blocks, block_bytes, ndev, loss_acc, ex_acc);
}

void launch_seg_scatter(const ModelView& m, const BlockView& blk, uint32_t rows_host, int ndev, uint64_t slice_rows,
                        void* send, const uint64_t* keys_sorted, const SegScratch& sc, hipStream_t s) {
    const uint64_t nflags = (uint64_t)ndev * slice_rows;
    hipLaunchKernelGGL(clear_chunk_flags_kernel, dim3((unsigned)((nflags + 255) / 256)), dim3(256), 0, s, send, ndev, slice_rows, m.d);
    launch_seg_reduce(m.d, blk, rows_host, keys_sorted, sc, EmitChunk{send, slice_rows}, s);
}

void launch_owner_reduce(const ModelView& m, const ChunkPtrs& recv, int ndev, uint64_t slice_rows, void* own, hipStream_t s) {
    DISPATCH_D(m.d, {
        const int gpb = 4 * (64 / (DD / 4));
        const dim3 grid(grid_for_groups((long long)slice_rows, gpb));
        if (ndev <= 4) hipLaunchKernelGGL((owner_reduce_kernel<DD, 4>), grid, dim3(256), 0, s, recv, ndev, slice_rows, own);
        else if (ndev <= 8) hipLaunchKernelGGL((owner_reduce_kernel<DD, 8>), grid, dim3(256), 0, s, recv, ndev, slice_rows, own);
        else hipLaunchKernelGGL((owner_reduce_kernel<DD, 16>), grid, dim3(256), 0, s, recv, ndev, slice_rows, own);
    });
}

void launch_owner_update(const ModelView& m, const ChunkPtrs& recv, int ndev, uint64_t slice_rows, uint64_t row0, uint64_t nrows, hipStream_t s) {
    if (nrows == 0) return;
    DISPATCH_D(m.d, {
        const int gpb = 4 * (64 / (DD / 4));
        const dim3 grid(grid_for_groups((long long)nrows, gpb));
        if (ndev <= 4) hipLaunchKernelGGL((owner_update_kernel<DD, 4>), grid, dim3(256), 0, s, m, recv, ndev, slice_rows, row0, nrows);
        else if (ndev <= 8) hipLaunchKernelGGL((owner_update_kernel<DD, 8>), grid, dim3(256), 0, s, m, recv, ndev, slice_rows, row0, nrows);
        else hipLaunchKernelGGL((owner_update_kernel<DD, 16>), grid, dim3(256), 0, s, m, recv, ndev, slice_rows, row0, nrows);
    });
}

void launch_table_apply(const ModelView& m, const ChunkPtrs& table, uint64_t slice_rows, hipStream_t s) {
    DISPATCH_D(m.d, {
        const int gpb = 4 * (64 / (DD / 4));
        hipLaunchKernelGGL((table_apply_kernel<DD>), dim3(grid_for_groups((long long)m.num_items, gpb)), dim3(256), 0, s, m, table, slice_rows);
    });
}

void launch_seg_list(const ModelView& m, const BlockView& blk, uint32_t rows_host, int ndev, uint64_t slice_rows,
                     const uint64_t* keys_sorted, float* G, float* gbl, uint32_t* fl, uint32_t* bounds, const SegScratch& sc,
                     hipStream_t s) {
    (void)hipMemsetAsync(fl, 0, (size_t)(3ull * rows_host) * sizeof(uint32_t), s); /* 0 at every position that is not a head */
    launch_seg_reduce(m.d, blk, rows_host, keys_sorted, sc, EmitList{G, gbl, fl}, s);
    hipLaunchKernelGGL(owner_bounds_kernel, dim3(1), dim3(64), 0, s, keys_sorted, (uint32_t)(3ull * rows_host), ndev, slice_rows, bounds);
}

void launch_owner_list_apply(const ModelView& m, const PeerLists& pl_in, const PeerBounds& pb, int ndev, int owner, MergePlan* plan,
                             uint32_t capacity, uint64_t* mkeys, uint64_t* mkeys_sorted, void* sort_temp, size_t sort_temp_bytes, hipStream_t s) {
    if (capacity == 0) return;
    PeerLists pl = pl_in;
    pl.plan = plan;
    hipLaunchKernelGGL(merge_plan_kernel, dim3(1), dim3(64), 0, s, pb, ndev, owner, plan);
    launch_merge_sort(pl, ndev, capacity, mkeys, mkeys_sorted, sort_temp, sort_temp_bytes, s);  /* (row, device, position) order */
    DISPATCH_D(m.d, {
        const int gpb = 4 * (64 / (DD / 4));
        /* grid-stride over the merge keys, whose number only the device knows: as many lane groups as the capacity would take, at
         * most 2 048 workgroups */
        int grid = grid_for_groups((long long)capacity, gpb);
        if (grid > 2048) grid = 2048;
        if (ndev <= 4) hipLaunchKernelGGL((owner_list_apply_kernel<DD, 4>), dim3(grid), dim3(256), 0, s, m, pl, mkeys_sorted);
        else if (ndev <= 8) hipLaunchKernelGGL((owner_list_apply_kernel<DD, 8>), dim3(grid), dim3(256), 0, s, m, pl, mkeys_sorted);
        else hipLaunchKernelGGL((owner_list_apply_kernel<DD, 16>), dim3(grid), dim3(256), 0, s, m, pl, mkeys_sorted);
    });
}

void launch_predict(const ModelView& m, const float* user, const uint32_t* items, uint64_t n, float* out, hipStream_t s) {
    if (n == 0) return;
    DISPATCH_D(m.d, { hipLaunchKernelGGL((predict_kernel<DD>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, m, user, items, n, out); });
}

void launch_rank(const ModelView& m, const float* reps, const int* rep_row, uint32_t num_users, const uint32_t* test_item,
                 const uint32_t* test_in_hist, const uint64_t* hist_ptr, const uint32_t* hist_items, float* ts_scratch,
                 uint32_t* ranks, uint32_t* nonfinite_flag, hipStream_t s) {
    if (num_users == 0) return;
    // 32 users per wave (128 per workgroup, four waves per SIMD).  (A 64-users-per-wave form — half the barriers and LDS fills per
    // flop at half the waves — measured 5 % slower at 8 192 users x 1e6 items, d = 128: 100 against 105 TFLOP/s; removed.)
    const uint32_t wgu = 128u;
    const uint32_t utiles = (num_users + wgu - 1) / wgu;
    // item groups: at least one 32-item tile each, and MANY more workgroups than the chip holds at once (a launch of
    // 1 024 workgroups on 768 resident slots ran one full round and a third of a second one)
    constexpr uint32_t target_wgs = 768u * 6u;
    uint32_t groups = (target_wgs + utiles - 1) / utiles;
    const uint32_t max_groups = (m.num_items + 31) / 32;
    if (groups > max_groups) groups = max_groups;
    if (groups < 1) groups = 1;
    uint32_t per = (m.num_items + groups - 1) / groups;
    per = ((per + 31) / 32) * 32;
    if (per > 65535u * 32u) per = 65535u * 32u; /* the kernel's per-lane counters are 16 bits wide: fewer than 65 536 tiles per range */
    groups = (m.num_items + per - 1) / per;
    DISPATCH_D(m.d, {
        hipLaunchKernelGGL((rank_test_score_kernel<DD>), dim3((num_users + 255) / 256), dim3(256), 0, s, m, reps, rep_row, num_users, test_item, test_in_hist, ts_scratch, ranks);
        hipLaunchKernelGGL((rank_gemm_kernel<DD, 1>), dim3(utiles, groups), dim3(256), 0, s, m, reps, rep_row, num_users, ts_scratch, per, ranks, nonfinite_flag);
        hipLaunchKernelGGL((rank_history_kernel<DD>), dim3(num_users), dim3(64), 0, s, m, reps, rep_row, ts_scratch, hist_ptr, hist_items, ranks);
    });
}

void launch_selftest_math(const float* x, float* out_cell_h, float* out_sig, float* out_tanh, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(selftest_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, out_cell_h, out_sig, out_tanh, n);
}
void launch_selftest_dot_tree(const float* x, const float* y, int d, uint64_t nrows, float* out, hipStream_t s) {
    DISPATCH_D(d, {
        const int gpb = 4 * (64 / (DD / 4));
        hipLaunchKernelGGL((selftest_dot_tree_kernel<DD>), dim3((unsigned)((nrows + gpb - 1) / gpb)), dim3(256), 0, s, x, y, nrows, out);
    });
}
void launch_selftest_mfma_chain(const float* a, const float* b, const float* c0, int k, float* out, hipStream_t s) {
    hipLaunchKernelGGL(selftest_mfma_chain_kernel, dim3(1), dim3(64), 0, s, a, b, c0, k, out);
}
void launch_selftest_mfma32_chain(const float* a, const float* b, int k, float* out, hipStream_t s) {
    hipLaunchKernelGGL(selftest_mfma32_chain_kernel, dim3(1), dim3(64), 0, s, a, b, k, out);
}

}  // namespace sbr
