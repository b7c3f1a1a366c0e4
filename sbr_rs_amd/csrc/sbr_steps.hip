// sbr_steps.hip — the reference's OWN schedule, one subsequence per optimiser step (/root/reference/src/models/sequence_model.rs:111-169),
// as runs of steps in one launch with each step's working set in LDS, and the reference-order negative stream:
//   ewma_steps_kernel        EWMA + single-negative loss + Adagrad, d <= 32 (ewma.rs:266-352 per step)
//   lstm_steps_kernel        LSTM Normal, d = 32, single-negative loss, Adagrad (lstm.rs:258-337 per step; benches/benchmark.rs:26-45)
//   score_refstream_kernel   the negatives of a step from the worker's sequential xorshift128 stream (sequence_model.rs:58-65, :137)
// Arithmetic: operation for operation that of the launches they replace (sbr_kernels.hip, sbr_wave.hip), so the bits are theirs.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see sbr_rs_amd/build.py).
#include "sbr_kernels.h"

#include <atomic>
#include <cstdlib>
#include <cstring>

#include "../../include/sbr_hip.h"
#include "sbr_device.h"
#include "sbr_numerics.h"
#include "sbr_wave_seq.h"

namespace sbr {

// ---- REFERENCE ORDER (sbr_model_set_reference_order; one sequence per step, one device): the negatives of a step come from the
// worker's own sequential generator, exactly as /root/reference/src/models/sequence_model.rs:58-65 / :137 draw them —
// `Uniform::new(0, num_items).sample(thread_rng)` (rand 0.5 as recalled: one xorshift128 u64 per attempt, widening multiply,
// rejection zone), one draw per try, WARP stopping at the first violating candidate — instead of the contract's counter-keyed
// draws.  The stream is sequential and its consumption data-dependent (WARP), so one wave walks it: windows of up to 64 draws
// are generated ahead (every lane runs the same scalar recurrence), their table rows gathered in ONE round trip, and the
// steps' tries then resolved in order out of LDS; the generator state after exactly the draws consumed goes back to memory
// (the same state shuffles the worker's partition at the next epoch, :109 — the host reads it back).  Scores, tests, loss and
// everything downstream are the contract's; the checker has the same mode (tests/test_parity_gpu.py compares the two bit for bit).
template <int D>
__global__ __launch_bounds__(256) void score_refstream_kernel(ModelView m, MbView mb, BlockView blk, WorkView w, uint32_t* rng_state, SmallTail tail) {
    constexpr int L = D / 4, NGRP = 256 / L, WIN = 64;
    extern __shared__ __attribute__((aligned(16))) float rl[];
    const int tid = threadIdx.x, lane = tid & 63, lg = lane % L, grp = tid / L;
    const int R = mb.R;
    float* Hs = rl;                              // [R][D]
    float* Cs = Hs + (size_t)R * D;              // [WIN][D] candidate rows of the window
    float* posv = Cs + WIN * D;                  // [R]
    float* negv = posv + R;                      // [R]
    uint32_t* njs = reinterpret_cast<uint32_t*>(negv + R);
    uint32_t* trs = njs + R;
    float* cb = reinterpret_cast<float*>(trs + R);  // [WIN]
    uint32_t* cand = reinterpret_cast<uint32_t*>(cb + WIN);
    uint32_t* sts = cand + WIN;                  // [WIN][4] generator state after draw i
    for (int r0 = 0; r0 < R; r0 += NGRP) {       // positives of every step: a lane group per row (workgroup-uniform trip count)
        const int r = r0 + grp, rr = r < R ? r : R - 1;
        const float4 h = ld4(blk.H + (size_t)rr * D + 4 * lg);
        const uint32_t pi = mb.out_idx[rr];
        const float pos = m.b[pi] + group_allreduce<L>(dot4(h, ld4(m.E + (size_t)pi * D + 4 * lg)));
        if (r < R) {
            st4(Hs + (size_t)r * D + 4 * lg, h);
            if (lg == 0) posv[r] = pos;
        }
    }
    __syncthreads();
    if (tid < 64) {
        uint32_t x = rng_state[0], y = rng_state[1], z = rng_state[2], ww = rng_state[3];
        const uint64_t range = (uint64_t)m.num_items;
        const uint64_t zone = ~0ull - ((0ull - range) % range);  // MAX - (MAX - range + 1) % range
        const int max_tries = m.loss == SBR_LOSS_WARP ? SBR_WARP_MAX_TRIES : 1;
        int tcur = 0, tries_cur = 0;
        while (tcur < R) {
            const int left = (R - tcur) * max_tries - tries_cur;
            const int want = left < WIN ? left : WIN;
            for (int i = 0; i < want; ++i) {  // the same scalar recurrence on every lane
                uint64_t v;
                for (;;) {
                    uint32_t t1 = x ^ (x << 11);
                    x = y; y = z; z = ww;
                    ww = ww ^ (ww >> 19) ^ (t1 ^ (t1 >> 8));
                    const uint32_t lo32 = ww;
                    t1 = x ^ (x << 11);
                    x = y; y = z; z = ww;
                    ww = ww ^ (ww >> 19) ^ (t1 ^ (t1 >> 8));
                    const uint64_t u = (uint64_t)lo32 | ((uint64_t)ww << 32);
                    const uint64_t plo = u * range;
                    if (plo <= zone) { v = __umul64hi(u, range); break; }
                }
                if (lane == 0) {
                    cand[i] = (uint32_t)v;
                    sts[4 * i] = x; sts[4 * i + 1] = y; sts[4 * i + 2] = z; sts[4 * i + 3] = ww;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int idx = lane; idx < want * L; idx += 64) {  // the window's rows: one round trip
                const int c = idx / L, l4 = idx - c * L;
                st4(Cs + (size_t)c * D + 4 * l4, ld4(m.E + (size_t)cand[c] * D + 4 * l4));
            }
            for (int c = lane; c < want; c += 64) cb[c] = m.b[cand[c]];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int pw = 0;
            while (pw < want && tcur < R) {  // (every lane group of the wave evaluates the same dot: uniform control flow)
                const float4 h = ld4(Hs + (size_t)tcur * D + 4 * lg);
                const float sc = cb[pw] + group_allreduce<L>(dot4(h, ld4(Cs + (size_t)pw * D + 4 * lg)));
                ++tries_cur;
                const uint32_t nj = cand[pw];
                ++pw;
                const bool stop = tries_cur == max_tries || (m.loss == SBR_LOSS_WARP && sbr_warp_violates(posv[tcur], sc));
                if (stop) {
                    if (lane == 0) { njs[tcur] = nj; negv[tcur] = sc; trs[tcur] = (uint32_t)tries_cur; }
                    ++tcur;
                    tries_cur = 0;
                }
            }
            if (pw > 0) { x = sts[4 * (pw - 1)]; y = sts[4 * (pw - 1) + 1]; z = sts[4 * (pw - 1) + 2]; ww = sts[4 * (pw - 1) + 3]; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) { rng_state[0] = x; rng_state[1] = y; rng_state[2] = z; rng_state[3] = ww; }
    }
    __syncthreads();
    double loss_part = 0.0;
    unsigned int tries_part = 0;
    for (int r = tid; r < R; r += 256) {
        float g, l;
        if (m.loss == SBR_LOSS_BPR) l = sbr_loss_bpr(posv[r], negv[r], &g);
        else l = sbr_loss_hinge(posv[r], negv[r], &g);
        blk.neg[r] = njs[r];
        blk.coef[r] = g;
        blk.in_idx[r] = mb.in_idx[r];
        blk.out_idx[r] = mb.out_idx[r];
        w.loss[r] = l;
        w.tries[r] = trs[r];
        loss_part += (double)l;
        tries_part += trs[r];
    }
    __shared__ double s_loss[4];
    __shared__ unsigned int s_tries[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        loss_part += __shfl_xor(loss_part, off, 64);
        tries_part += __shfl_xor(tries_part, off, 64);
    }
    if (lane == 0) {
        s_loss[tid >> 6] = loss_part;
        s_tries[tid >> 6] = tries_part;
    }
    __syncthreads();
    const double lsum = s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3];
    const unsigned int tsum = s_tries[0] + s_tries[1] + s_tries[2] + s_tries[3];
    if (tid == 0) {
        w.part_loss[0] = lsum;
        w.part_tries[0] = tsum;
    }
    small_tail<256>(mb, blk, w, tail, lsum, tsum);
}

// ---- a RUN of one-sequence optimiser steps in ONE launch (launch_epoch_steps) --------------------------------------------
// The reference's own schedule is one optimiser step per subsequence (sequence_model.rs:111-169).  At d <= 32 such a step is a
// few microseconds of arithmetic.  At one sequence per step nobody but the step's own workgroup touches the parameters, so a
// run of consecutive steps needs no grid-wide synchronisation: ONE workgroup walks them, step k + 1's gather reading the rows
// step k's update wrote from the same CU (workgroup-scope release / acquire).
// (Round 5 measured the LSTM in this form as the four launches' phases concatenated — lstm_fwd_wave_seq | score + tail |
// lstm_bwd_wave_seq | dense gradient + update | sparse update, barriers between — and it LOST to the four launches: 35.4 against
// 31 us per step on the reference's Criterion shape, 152 against 118 us on MovieLens-100K.  The launches' dispatch overhead was
// already hidden behind their predecessors; what a step waits for is its dependent memory round trips and the instruction count
// of lone waves (~9 cycles per instruction), and one CU's four waves are fewer than the launches' workgroups.  Removed; NOTES.md.)
// ---- EWMA + single-negative loss: a run of one-sequence steps with the step's whole working set in LDS --------------------
// (ewma.rs:266-352 for one subsequence; sequence_model.rs:111-169 around it.)  A one-sequence EWMA step is ~3 n rows of 4 d bytes
// and a few hundred flops; as launches — and as the concatenated phases of epoch_steps_kernel — it was a chain of ~10 dependent
// global-memory round trips (ids -> rows -> H -> coefficients -> rows again for the backward scan -> keys -> rows -> table), each
// ~1 us, around ~1 us of arithmetic.  Here a step makes ONE: the gather of its 3 n table rows (input, target, negative) together
// with the optimiser state of the rows it will update — the ids were requested during the previous step, the negative of a
// single-negative loss is a hash of the row counter, and the keys are ordered before the gather, so the touched rows are known.
// Everything between — scan, scores, loss, backward scan, dalpha, the per-row reduction and the Adagrad update — works on LDS;
// the updated rows are written back and the next step's gather waits for them (workgroup-scope release / acquire: one CU).
// alpha and its accumulator, the loss accumulators and the lagged-loss nodes live in LDS / registers for the whole run and are
// written back once.  The step's block arrays (debug fetch, sbr_fit_sparse_stats) are written for the LAST step of the run only.
// Arithmetic: operation for operation that of ewma_seq_body / ewma_backward_seq / small_tail / seg_short_rows / EmitApply /
// dense_apply_element, on the same lane layout (d/4 lanes x 4 elements per row), so the bits are theirs.
#define SBR_EWMA_STEPS_MAX_ROWS 128 /* rows per step: the gather keeps 3 x 128 d / 1 024 + 12 row pieces per thread in registers */
#define SBR_EWMA_STEPS_LDS_FLOATS(max_rows, d) \
    (4 * (size_t)(d) + 25 * (size_t)(((max_rows) + 3) & ~3) + 16 + 8 * (size_t)(max_rows) * (d) + 6 * (size_t)(max_rows) + (d) + 4)
template <int D>
__global__ __launch_bounds__(256) void ewma_steps_kernel(ModelView m, EpochView ev, BlockView blk, WorkView w, uint64_t epoch_key, SmallTail tail,
                                                         int step_begin, int step_end, int max_rows, unsigned long long* prof) {
    constexpr int L = D / 4, NGRP = 256 / L;
    constexpr int RQ = SBR_EWMA_STEPS_MAX_ROWS * L / 256;              // 16-byte pieces per thread and gathered array
    constexpr int HQ = (3 * SBR_EWMA_STEPS_MAX_ROWS + NGRP - 1) / NGRP;  // segment-head positions per lane group
    extern __shared__ __attribute__((aligned(16))) float el[];
    const int tid = threadIdx.x, lane = tid & 63, lg = lane % L, grp = tid / L;
    /* run-resident state */
    float* alphaL = el;                      // [D]
    float* alphaAcc = alphaL + D;            // [D]
    float* sigA = alphaAcc + D;              // [2 D] sigmoid(alpha) | 1 - sigmoid(alpha) of the step in flight
    float* lagN = sigA + 2 * D;              // [max_rows] loss nodes (sbr_report.hip)
    /* index work of a step — ids, keys, their order — is done a step AHEAD by waves 2-3, which have nothing else to do while wave 0
     * scans: two buffers of ids and ordered keys, one scratch of unordered keys */
    const int MR4 = (max_rows + 3) & ~3;
    uint32_t* idsBuf = reinterpret_cast<uint32_t*>(lagN + MR4);            // [2][3 MR4]
    uint64_t* kaS = reinterpret_cast<uint64_t*>(idsBuf + 6 * MR4);         // [3 MR4 + 8] (eight sentinels behind the keys)
    uint64_t* kbBuf = kaS + 3 * MR4 + 8;                                   // [2][3 MR4]
    float* stepL = reinterpret_cast<float*>(kbBuf + 6 * MR4);
    const int t2 = tid - 128;  // waves 2-3: row of the step they prepare
    for (int k = tid; k < D; k += 256) { alphaL[k] = m.alpha[k]; alphaAcc[k] = m.alpha_acc[k]; }
    for (int t = tid; t < max_rows; t += 256) lagN[t] = tail.lag_state[1 + 2 * t];
    float lag_acc = tail.lag_state[0];
    double la0 = 0.0, la1 = 0.0;             // thread 64 (wave 1 does the bookkeeping): loss_acc[0], loss_acc[1]
    unsigned long long ex0 = 0, ex1 = 0, ex2 = 0;
    if (tid == 64 && tail.loss_acc) { la0 = tail.loss_acc[0]; la1 = tail.loss_acc[1]; ex0 = tail.ex_acc[0]; ex1 = tail.ex_acc[1]; ex2 = tail.ex_acc[2]; }
    unsigned long long pc[5] = {0, 0, 0, 0, 0};
    unsigned long long tq = clock64();
#define SBR_PHASE_CLOCK(i) { const unsigned long long now = clock64(); pc[i] += now - tq; tq = now; }
    /* ids, keys and key order of a step (waves 2-3; the keys' stable order by row: distinct keys, so the rank of a key among all of
     * them is its place — eight keys per batch, their LDS reads independent of each other and of the count: a dependent read per
     * key cost a ~30-key step 1.6 us, a v_readlane broadcast of the keys 1.9 us) */
    auto index_ids = [&](const StepDesc& d, int buf) {
        if (t2 >= 0 && t2 < (int)d.rows) {
            const uint32_t vin = ev.in_idx[d.row_base + t2], vout = ev.out_idx[d.row_base + t2];
            const uint32_t ng = sbr_neg_draw(epoch_key, ev.ctr[d.row_base + t2], 0u, m.num_items);
            uint32_t* ib = idsBuf + (size_t)buf * 3 * MR4;
            ib[t2] = vin; ib[d.rows + t2] = vout; ib[2 * d.rows + t2] = ng;
            kaS[3 * t2] = ((uint64_t)vin << 32) | (uint32_t)(3 * t2);
            kaS[3 * t2 + 1] = ((uint64_t)vout << 32) | (uint32_t)(3 * t2 + 1);
            kaS[3 * t2 + 2] = ((uint64_t)ng << 32) | (uint32_t)(3 * t2 + 2);
        }
        if (t2 >= 0 && t2 < 8) kaS[3 * d.rows + t2] = ~0ull;  // sentinels: the ranking reads whole batches of eight keys
    };
    auto index_rank = [&](const StepDesc& d, int buf) {
        if (t2 < 0) return;
        const int m3 = 3 * (int)d.rows;
        uint64_t* kbn = kbBuf + (size_t)buf * 3 * MR4;
        /* a thread ranks up to three keys (t2, t2 + 128, t2 + 256) in ONE pass over the keys: eight keys per batch, their LDS reads
         * independent of each other and of the counts */
        const uint64_t key0 = t2 < m3 ? kaS[t2] : ~0ull;
        if (m3 <= 128) {
            int r0 = 0;
            for (int j0 = 0; j0 < m3; j0 += 8) {
                uint64_t kk[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) kk[q] = kaS[j0 + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) r0 += kk[q] < key0 ? 1 : 0;
            }
            if (t2 < m3) kbn[r0] = key0;
        } else {
            const uint64_t key1 = t2 + 128 < m3 ? kaS[t2 + 128] : ~0ull, key2 = t2 + 256 < m3 ? kaS[t2 + 256] : ~0ull;
            int r0 = 0, r1 = 0, r2 = 0;
            for (int j0 = 0; j0 < m3; j0 += 8) {
                uint64_t kk[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) kk[q] = kaS[j0 + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    r0 += kk[q] < key0 ? 1 : 0;
                    r1 += kk[q] < key1 ? 1 : 0;
                    r2 += kk[q] < key2 ? 1 : 0;
                }
            }
            if (t2 < m3) kbn[r0] = key0;
            if (t2 + 128 < m3) kbn[r1] = key1;
            if (t2 + 256 < m3) kbn[r2] = key2;
        }
    };
    StepDesc sd = ev.desc[step_begin];
    index_ids(sd, step_begin & 1);
    __syncthreads();
    index_rank(sd, step_begin & 1);
    __syncthreads();
    for (int st = step_begin; st < step_end; ++st) {
        const int n = (int)sd.rows, n3 = 3 * n;
        const bool last = st + 1 == step_end;
        float* X = stepL;                    // [n][D] E[in_t]
        float* P = X + (size_t)n * D;        // [n][D] E[out_t]
        float* N = P + (size_t)n * D;        // [n][D] E[neg_t]
        float* H = N + (size_t)n * D;        // [n][D] s_t
        float* DS = H + (size_t)n * D;       // [n][D] dloss/ds_t, then dX_t in place
        float* A = DS + (size_t)n * D;       // [3n][D] optimiser state (E_acc) of the row at segment head p
        const int cur = st & 1;
        const uint32_t* iin = idsBuf + (size_t)cur * 3 * MR4;
        const uint32_t* iout = iin + n;
        const uint32_t* ineg = iout + n;
        const uint64_t* kb = kbBuf + (size_t)cur * 3 * MR4;
        float* bp = A + (size_t)n3 * D;      // b[out_t], b[neg_t] and their optimiser state
        float* bn = bp + n;
        float* bpa = bn + n;
        float* bna = bpa + n;
        float* coef = bna + n;
        float* lossv = coef + n;
        float* dab = lossv + n;              // [D]
        StepDesc sdn = sd;
        if (!last) sdn = ev.desc[st + 1];
        /* ---- ONE batch of requests: the 3 n rows (16-byte pieces into registers, up to RQ per thread and array), the biases and
         * their optimiser state, and E_acc of every segment head's row (the order of this step's keys was made a step ago) */
        const int nq = n * L;  // pieces per array
        float4 vx[RQ], vp[RQ], vn[RQ];
#pragma unroll
        for (int u = 0; u < RQ; ++u) {
            const int idx = u * 256 + tid;
            if (idx < nq) {
                const int rr = idx / L, l4 = idx - rr * L;
                vx[u] = ld4(m.E + (size_t)iin[rr] * D + 4 * l4);
                vp[u] = ld4(m.E + (size_t)iout[rr] * D + 4 * l4);
                vn[u] = ld4(m.E + (size_t)ineg[rr] * D + 4 * l4);
            }
        }
        float b0 = 0.f, b1 = 0.f, a0 = 0.f, a1 = 0.f;
        if (tid < n) {
            const uint32_t po = iout[tid], pn = ineg[tid];
            b0 = m.b[po]; b1 = m.b[pn]; a0 = m.bacc[po]; a1 = m.bacc[pn];
        }
        /* (a lone wave pays ~9 cycles per instruction, so what a step costs is the longest instruction sequence any one wave runs
         * between two barriers: the waves specialise) — the last wave forms sigmoid(alpha) for the scans while the rows travel */
        const int wave = tid >> 6;
        if (wave == 3) {
            if (lane < D) {
                const float av = sbr_sigmoidf(alphaL[lane]);
                sigA[lane] = av;
                sigA[D + lane] = 1.0f - av;
            }
        }
        /* the optimiser state (E_acc) of every segment head's row: piece (p, lg) by the thread that will update it — it stays in
         * flight underneath the scan and lands in LDS before the update */
        const int hq = (n3 + NGRP - 1) / NGRP;  // head rounds this step needs (workgroup-uniform)
        float4 ha[HQ];
#pragma unroll
        for (int i = 0; i < HQ; ++i) {
            ha[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < hq) {
                const int p = i * NGRP + grp;
                if (p < n3) {
                    const uint32_t row = (uint32_t)(kb[p] >> 32);
                    if (p == 0 || (uint32_t)(kb[p - 1] >> 32) != row) ha[i] = ld4(m.Eacc + (size_t)row * D + 4 * lg);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RQ; ++u) {
            const int idx = u * 256 + tid;
            if (idx < nq) {
                st4(X + 4 * (size_t)idx, vx[u]);
                st4(P + 4 * (size_t)idx, vp[u]);
                st4(N + 4 * (size_t)idx, vn[u]);
            }
        }
        if (tid < n) { bp[tid] = b0; bn[tid] = b1; bpa[tid] = a0; bna[tid] = a1; }
        __syncthreads();
        SBR_PHASE_CLOCK(0)
        /* ---- scan (ewma.rs:302-313): ONE ELEMENT PER LANE of wave 0 (the d chains are independent: same operations per element
         * as the four-per-lane form, a quarter of the instructions on the wave that everybody waits for), s_t to LDS */
        if (tid < D) {
            const float av = sigA[tid], omav = sigA[D + tid];
            float sv = X[tid];
            H[tid] = sv;
            float xn = X[(size_t)(n > 1 ? 1 : 0) * D + tid];
            for (int t = 1; t < n; ++t) {
                const float x = xn;
                xn = X[(size_t)(t + 1 < n ? t + 1 : t) * D + tid];  // the next row's LDS read under this row's arithmetic
                sv = sbr_fma(av, sv, omav * x);
                H[(size_t)t * D + tid] = sv;
            }
        }
#pragma unroll
        for (int i = 0; i < HQ; ++i) {
            if (i < hq) {
                const int p = i * NGRP + grp;
                if (p < n3) st4(A + (size_t)p * D + 4 * lg, ha[i]);
            }
        }
        if (!last) index_ids(sdn, cur ^ 1);  /* waves 2-3, beside wave 0's scan */
        __syncthreads();
        /* ---- scores, loss, dloss/ds of every step (ewma.rs:315-335): a lane group per row */
        for (int r0 = 0; r0 < n; r0 += NGRP) {  // workgroup-uniform trip count (the DPP reductions want whole groups)
            const int r = r0 + grp;
            const int rr = r < n ? r : n - 1;
            const float4 sv = ld4(H + (size_t)rr * D + 4 * lg), ep = ld4(P + (size_t)rr * D + 4 * lg), ec = ld4(N + (size_t)rr * D + 4 * lg);
            const float pos = bp[rr] + group_allreduce<L>(dot4(sv, ep));
            const float neg = bn[rr] + group_allreduce<L>(dot4(sv, ec));
            float g, l;
            if (m.loss == SBR_LOSS_BPR) l = sbr_loss_bpr(pos, neg, &g);
            else l = sbr_loss_hinge(pos, neg, &g);
            if (r < n) {
                // dloss/ds: g*E[neg] - g*E[pos], two rounded products and one subtraction
                st4(DS + (size_t)r * D + 4 * lg, make_float4(g * ec.x - g * ep.x, g * ec.y - g * ep.y, g * ec.z - g * ep.z, g * ec.w - g * ep.w));
                if (lg == 0) { coef[r] = g; lossv[r] = l; }
            }
        }
        __syncthreads();
        SBR_PHASE_CLOCK(1)
        /* ---- backward scan (ewma_backward_seq): one element per lane of wave 0; dX over DS, dalpha partial to LDS.  Wave 1 meanwhile:
         * the loss bookkeeping */
        if (tid < D) {
            const float av = sigA[tid], omav = sigA[D + tid];
            float carry = 0.0f, da = 0.0f;
            float dsn = DS[(size_t)(n - 1) * D + tid], xnn = X[(size_t)(n - 1) * D + tid], spn = H[(size_t)(n > 1 ? n - 2 : 0) * D + tid];
            for (int t = n - 1; t >= 0; --t) {
                float ds = dsn;
                const float xv_ = xnn, spv_ = spn;
                {   // step t - 1's operands under step t's arithmetic
                    const int tp = t > 0 ? t - 1 : 0;
                    dsn = DS[(size_t)tp * D + tid];
                    xnn = X[(size_t)tp * D + tid];
                    spn = H[(size_t)(tp > 0 ? tp - 1 : 0) * D + tid];
                }
                ds = t != n - 1 ? ds + carry : ds + 0.0f;
                float dx;
                if (t > 0) {
                    dx = omav * ds;
                    carry = av * ds;
                    da = sbr_fma(ds, spv_ - xv_, da);
                } else {
                    dx = ds;
                }
                DS[(size_t)t * D + tid] = dx;
            }
            dab[tid] = da;
        } else if (tid >= 64 && tid < 128) { /* wave 1: the step's loss figures (block_header_kernel / small_tail) */
            double lp = 0.0;
            for (int r = lane; r < n; r += 64) lp += (double)lossv[r];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) lp += __shfl_xor(lp, off, 64);
            /* the lagged figure: the node of this length is read, then nodes 0 .. n-1 take the running sums (sbr_report.hip) */
            const float x = lagN[n - 1];
            float sum = 0.0f;
            for (int base = 0; base < n; base += 64) {
                const int tt = base + lane;
                const float v = tt < n ? lossv[tt] : 0.0f;
                const int cnt = n - base < 64 ? n - base : 64;
                float mine = 0.0f;
                for (int l = 0; l < cnt; ++l) {
                    sum = sum + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l));
                    mine = lane == l ? sum : mine;
                }
                if (tt < n) lagN[tt] = mine;
            }
            lag_acc = lag_acc + x;
            if (lane == 0) { /* (thread 64 keeps the accumulators of the run) */
                la0 += lp; la1 += lp;
                ex0 += (unsigned long long)n; ex1 += (unsigned long long)n; ex2 += (unsigned long long)n;
                if (last) {
                    tail.header[0] = (uint32_t)n;
                    tail.header[1] = (uint32_t)n;
                    tail.header[2] = tail.header[3] = 0;
                    *reinterpret_cast<double*>(tail.header + 4) = lp;
                    *reinterpret_cast<unsigned long long*>(tail.header + 6) = (unsigned long long)n;
                    w.part_loss[0] = lp;
                    w.part_tries[0] = (unsigned int)n;
                }
            }
        } else if (!last) {
            index_rank(sdn, cur ^ 1);  /* waves 2-3, beside the backward scan */
        }
        __syncthreads();
        SBR_PHASE_CLOCK(2)
        /* ---- dalpha and its update (ewma_dab_final_kernel with one sequence + dense_apply_element), in LDS */
        float galpha = 0.0f;
        if (wave == 3 && lane < D) { /* (the last wave: its lanes have no segments to reduce before the others do) */
            float pcv = 0.0f;
            pcv = pcv + dab[lane];
            const float av = sigA[lane];  // = sbr_sigmoidf(alpha): alpha does not move between the scans and here
            galpha = pcv * (av * (1.0f - av));
            float wv = alphaL[lane], G = alphaAcc[lane];
            sbr_adagrad(&wv, &G, galpha, m.lr, m.l2);
            alphaL[lane] = wv;
            alphaAcc[lane] = G;
            if (last) blk.dense[lane] = galpha;
        }
        /* ---- sparse update: a lane group per SEGMENT of the ordered keys (a position whose row differs from its predecessor's starts
         * one); entries in (packed row, kind) order, the first initialises, SBR_SEG_CHUNK-entry chunk partials added in order.  The
         * row's parameters are the gathered copy of any of its entries, its optimiser state came with the gather. */
        for (int p0 = 0; p0 < n3; p0 += NGRP) {
            const int p = p0 + grp;
            if (p >= n3) continue;
            const uint32_t row = (uint32_t)(kb[p] >> 32);
            if (p > 0 && (uint32_t)(kb[p - 1] >> 32) == row) continue;
            struct { float4 w, a, mo; } q;
            {
                const uint32_t src = (uint32_t)kb[p];
                const uint32_t r = src / 3, kind = src % 3;
                q.w = ld4((kind == 0 ? X : (kind == 1 ? P : N)) + (size_t)r * D + 4 * lg);
                q.a = ld4(A + (size_t)p * D + 4 * lg);
                q.mo = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float4 tot = make_float4(0.f, 0.f, 0.f, 0.f), g = tot;
            float totb = 0.0f, gb = 0.0f, bv = 0.0f, ba = 0.0f;
            bool tot_first = true, tot_hb = false, first = true, has_b = false, have_bias = false;
            int in_chunk = 0;
            for (int e = p; e < n3 && (uint32_t)(kb[e] >> 32) == row; ++e) {
                const uint32_t src = (uint32_t)kb[e];
                const uint32_t r = src / 3, kind = src % 3;
                const float4 v = ld4((kind == 0 ? DS : H) + (size_t)r * D + 4 * lg);
                const float scl = kind == 0 ? 1.0f : (kind == 1 ? -coef[r] : coef[r]);
                if (first) {
                    g = make_float4(scl * v.x, scl * v.y, scl * v.z, scl * v.w);
                    first = false;
                } else {
                    g.x = g.x + scl * v.x; g.y = g.y + scl * v.y; g.z = g.z + scl * v.z; g.w = g.w + scl * v.w;
                }
                if (kind != 0) {
                    gb = has_b ? gb + scl : scl;
                    has_b = true;
                    if (!have_bias) { bv = kind == 1 ? bp[r] : bn[r]; ba = kind == 1 ? bpa[r] : bna[r]; have_bias = true; }
                }
                if (++in_chunk == SBR_SEG_CHUNK) { /* a chunk is complete: the partials are added in order, the first initialises */
                    if (tot_first) { tot = g; tot_first = false; } else { tot.x = tot.x + g.x; tot.y = tot.y + g.y; tot.z = tot.z + g.z; tot.w = tot.w + g.w; }
                    if (has_b) { totb = tot_hb ? totb + gb : gb; tot_hb = true; }
                    first = true; has_b = false; in_chunk = 0; gb = 0.0f;
                }
            }
            if (in_chunk) {
                if (tot_first) { tot = g; tot_first = false; } else { tot.x = tot.x + g.x; tot.y = tot.y + g.y; tot.z = tot.z + g.z; tot.w = tot.w + g.w; }
                if (has_b) { totb = tot_hb ? totb + gb : gb; tot_hb = true; }
            }
            /* EmitApply::row + bias_update (Adagrad) */
            sbr_adagrad(&q.w.x, &q.a.x, tot.x, m.lr, m.l2);
            sbr_adagrad(&q.w.y, &q.a.y, tot.y, m.lr, m.l2);
            sbr_adagrad(&q.w.z, &q.a.z, tot.z, m.lr, m.l2);
            sbr_adagrad(&q.w.w, &q.a.w, tot.w, m.lr, m.l2);
            st4(m.E + (size_t)row * D + 4 * lg, q.w);
            st4(m.Eacc + (size_t)row * D + 4 * lg, q.a);
            if (tot_hb && lg == 0) {
                sbr_adagrad(&bv, &ba, totb, m.lr, m.l2);
                m.b[row] = bv;
                m.bacc[row] = ba;
            }
        }
        SBR_PHASE_CLOCK(3)
        if (last) { /* the block of the run's last step, for sbr_fit_debug_fetch / sbr_fit_sparse_stats */
            for (int idx = tid; idx < n * L; idx += 256) {
                const int rr = idx / L, l4 = idx - rr * L;
                st4(blk.H + (size_t)rr * D + 4 * l4, ld4(H + (size_t)rr * D + 4 * l4));
                st4(blk.dX + (size_t)rr * D + 4 * l4, ld4(DS + (size_t)rr * D + 4 * l4));
            }
            if (tid < n) {
                blk.in_idx[tid] = iin[tid]; blk.out_idx[tid] = iout[tid]; blk.neg[tid] = ineg[tid];
                blk.coef[tid] = coef[tid];
                w.loss[tid] = lossv[tid];
                w.tries[tid] = 1u;
            }
            for (int e = tid; e < n3; e += 256) tail.keys_sorted[e] = kb[e];
            if (tid == 0) {
                uint32_t nh = 0;
                for (int e = 0; e < n3; ++e)
                    if (e == 0 || (uint32_t)(kb[e] >> 32) != (uint32_t)(kb[e - 1] >> 32)) tail.head_pos[nh++] = (uint32_t)e;
                tail.head_pos[nh] = (uint32_t)n3;
                *tail.nheads = nh;
            }
        }
        sd = sdn;
        phase_sync(); /* the table rows written above are read by the next step's gather */
        SBR_PHASE_CLOCK(4)
    }
#undef SBR_PHASE_CLOCK
    for (int k = tid; k < D; k += 256) { m.alpha[k] = alphaL[k]; m.alpha_acc[k] = alphaAcc[k]; }
    for (int t = tid; t < max_rows; t += 256) tail.lag_state[1 + 2 * t] = lagN[t];
    if (tid == 64) {
        tail.lag_state[0] = lag_acc;
        if (tail.loss_acc) { tail.loss_acc[0] = la0; tail.loss_acc[1] = la1; tail.ex_acc[0] = ex0; tail.ex_acc[1] = ex1; tail.ex_acc[2] = ex2; }
    }
    if (prof && tid == 0) {
        for (int i = 0; i < 5; ++i) prof[i] += pc[i];
        prof[5] += (unsigned long long)(step_end - step_begin);
    }
}

// ---- LSTM (Normal, d = 32) + single-negative loss: a run of one-sequence steps with the step's working set in LDS -----------
// (lstm.rs:258-337 for one subsequence; sequence_model.rs:111-169 around it: the shape of the reference's own Criterion bench,
// benches/benchmark.rs:26-45.)  The four launches of such a step (lstm_fwd_wave | score + tail | lstm_bwd_wave | small_back) were
// each a chain of dependent global-memory round trips around the two recurrence chains; here ONE workgroup of eight waves walks a
// run of steps and a step makes one gather (3 n rows + the touched rows' optimiser state, ids a step ahead, keys ordered first) —
// the gathered rows, x-part pre-activations, gates, cell states, h, dloss/dh, dz and dX all live in LDS:
//   wave 0   forward recurrence (lstm_fwd_wave_seq's d = 32 / four-gate form: h_k by v_readlane, packed fma, rational tanh of the
//            lane's own two pre-activations, v_permlane32_swap across the half-waves), then the backward recurrence
//            (lstm_bwd_wave_seq's form: cell backward in lane u = l mod 32, column l of dz W^T with row l of W in registers)
//   all      x half of every pre-activation chain up front, scores / loss / dloss/dh / tanh(c) of every step, the dense gradient
//            (per-element chains over the step's rows, small_back_kernel's chain) with its Adagrad update, the per-row reduction
//            and Adagrad update of the touched table rows
// The LSTM weights are resident in LDS for the run (33 KB; every wave takes its share into registers where it uses it, the dense
// update rewrites LDS and global memory); the packed weight copies (Wp / WTp) are re-emitted once after the run.  Arithmetic: operation for operation that of the
// launches it replaces, so the bits are theirs (tests run both).
#define SBR_LSTM_STEPS_MAX_ROWS 48 /* rows per step: one 16-byte piece per thread and gathered array, 3 head rounds per lane group; LDS */
#define SBR_LSTM_STEPS_LDS_FLOATS(max_rows) ((size_t)(((max_rows) + 3) & ~3) + 128 + 65 * 128 + 24 * SBR_LSTM_STEPS_MAX_ROWS + 16 + (size_t)(max_rows) * (32 * 7 + 128 * 2 + 96) + 32 + 9 * (size_t)(max_rows) + 1 + 12 * (size_t)(max_rows))
__global__ __launch_bounds__(512) void lstm_steps_kernel(ModelView m, EpochView ev, BlockView blk, WorkView w, uint64_t epoch_key, SmallTail tail,
                                                         int step_begin, int step_end, int max_rows /* loss nodes: max_sequence_length - 1 */,
                                                         unsigned long long* prof) {
    constexpr int D = 32, NGD = 128, K2 = 64, L = 8, NT = 512, NGRP = NT / L, HQ = 3;  // four gates
    extern __shared__ __attribute__((aligned(16))) float ll[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane % L, grp = tid / L;
    const int u = lane & 31;
    const bool low = lane < 32;
    float* lagN = ll;                                   // [max_rows] loss nodes (sbr_report.hip)
    float* Zs = lagN + ((max_rows + 3) & ~3);           // [NGD] dz of the backward step in flight
    float* WL = Zs + NGD;                               // [2D][NGD] the LSTM weights, resident for the run (rows [x ; h])
    float* bWL = WL + (size_t)K2 * NGD;                 // [NGD]
    /* a step's index work — ids, the negatives' hash, the keys and their order — is done a step AHEAD (ids and keys beside the
     * x half, the ranking on waves 1-3 beside wave 0's forward recurrence): two buffers of ids and ordered keys, one scratch of
     * unordered keys with eight sentinels behind them */
    constexpr int MR = SBR_LSTM_STEPS_MAX_ROWS;
    uint32_t* idsBuf = reinterpret_cast<uint32_t*>(bWL + NGD);           // [2][3 MR]
    uint64_t* kaS = reinterpret_cast<uint64_t*>(idsBuf + 6 * MR);        // [3 MR + 8]
    uint64_t* kbBuf = kaS + 3 * MR + 8;                                  // [2][3 MR]
    float* stepL = reinterpret_cast<float*>(kbBuf + 6 * MR);
    for (int t = tid; t < max_rows; t += NT) lagN[t] = tail.lag_state[1 + 2 * t];
    for (int i = tid; i < K2 * NGD / 4; i += NT) st4(WL + 4 * (size_t)i, ld4(m.W + 4 * (size_t)i));
    if (tid < NGD) bWL[tid] = m.bW[tid];
    float lag_acc = tail.lag_state[0];
    double la0 = 0.0, la1 = 0.0;                        // thread 128 (wave 2 does the bookkeeping)
    unsigned long long ex0 = 0, ex1 = 0, ex2 = 0;
    if (tid == 128 && tail.loss_acc) { la0 = tail.loss_acc[0]; la1 = tail.loss_acc[1]; ex0 = tail.ex_acc[0]; ex1 = tail.ex_acc[1]; ex2 = tail.ex_acc[2]; }
    unsigned long long pc[5] = {0, 0, 0, 0, 0};
    unsigned long long tq = clock64();
#define SBR_PHASE_CLOCK(i) { const unsigned long long now = clock64(); pc[i] += now - tq; tq = now; }
    StepDesc sd = ev.desc[step_begin];
    uint32_t nin = 0, nout = 0, nctr = 0;
    if (tid < (int)sd.rows) { nin = ev.in_idx[sd.row_base + tid]; nout = ev.out_idx[sd.row_base + tid]; nctr = ev.ctr[sd.row_base + tid]; }
    auto index_ids = [&](const StepDesc& d, int buf) {  // threads < rows hold the step's ids in registers
        if (tid < (int)d.rows) {
            const uint32_t ng = sbr_neg_draw(epoch_key, nctr, 0u, m.num_items);
            uint32_t* ib = idsBuf + (size_t)buf * 3 * MR;
            ib[tid] = nin; ib[d.rows + tid] = nout; ib[2 * d.rows + tid] = ng;
            kaS[3 * tid] = ((uint64_t)nin << 32) | (uint32_t)(3 * tid);
            kaS[3 * tid + 1] = ((uint64_t)nout << 32) | (uint32_t)(3 * tid + 1);
            kaS[3 * tid + 2] = ((uint64_t)ng << 32) | (uint32_t)(3 * tid + 2);
        }
        if (tid >= 64 && tid < 72) kaS[3 * d.rows + (tid - 64)] = ~0ull;  // sentinels: the ranking reads whole batches of eight keys
    };
    auto index_rank = [&](const StepDesc& d, int buf) {  // waves 1-3: one key per thread (3 rows <= 144 keys)
        const int e = tid - 64, m3 = 3 * (int)d.rows;
        if (e < 0 || e >= 192) return;
        const uint64_t key = e < m3 ? kaS[e] : ~0ull;
        int rank = 0;
        for (int j0 = 0; j0 < m3; j0 += 8) {
            uint64_t kk[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) kk[q] = kaS[j0 + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) rank += kk[q] < key ? 1 : 0;
        }
        if (e < m3) (kbBuf + (size_t)buf * 3 * MR)[rank] = key;
    };
    index_ids(sd, step_begin & 1);
    __syncthreads();
    index_rank(sd, step_begin & 1);
    __syncthreads();
    for (int st = step_begin; st < step_end; ++st) {
        const int n = (int)sd.rows, n3 = 3 * n;
        const bool last = st + 1 == step_end;
        float* X = stepL;                                // [n][D]   E[in_t]
        float* P = X + (size_t)n * D;                    // [n][D]   E[out_t]
        float* N = P + (size_t)n * D;                    // [n][D]   E[neg_t]
        float* H = N + (size_t)n * D;                    // [n][D]   h_t
        float* Cc = H + (size_t)n * D;                   // [n+1][D] c_{t-1} at row t (row 0: zeros) .. c_t at row t+1
        float* TC = Cc + (size_t)(n + 1) * D;            // [n][D]   tanh(c_t)
        float* DH = TC + (size_t)n * D;                  // [n][D]   dloss/dh_t, then dX_t in place
        float* G = DH + (size_t)n * D;                   // [n][4D]  gate values
        float* PZ = G + (size_t)n * NGD;                 // [n][4D]  bias + x part of the pre-activations, then dz_t
        float* A = PZ + (size_t)n * NGD;                 // [3n][D]  optimiser state (E_acc) of the row at segment head p
        const int cur = st & 1;
        const uint32_t* iin = idsBuf + (size_t)cur * 3 * MR;
        const uint32_t* iout = iin + n;
        const uint32_t* ineg = iout + n;
        const uint64_t* kb = kbBuf + (size_t)cur * 3 * MR;
        float* bp = A + (size_t)n3 * D;
        float* bn = bp + n;
        float* bpa = bn + n;
        float* bna = bpa + n;
        float* coef = bna + n;
        float* lossv = coef + n;
        /* the next step's ids are requested now (registers of threads < rows; they become LDS ids and keys beside the x half) */
        StepDesc sdn = sd;
        if (!last) {
            sdn = ev.desc[st + 1];
            if (tid < (int)sdn.rows) { nin = ev.in_idx[sdn.row_base + tid]; nout = ev.out_idx[sdn.row_base + tid]; nctr = ev.ctr[sdn.row_base + tid]; }
        }
        /* ---- ONE batch of requests: the 3 n rows (one 16-byte piece per thread and array), the biases and their optimiser state,
         * E_acc of every segment head's row (the order of this step's keys was made a step ago) */
        const int nq = n * L;
        float4 vx = make_float4(0.f, 0.f, 0.f, 0.f), vp = vx, vn = vx;
        if (tid < nq) {
            const int rr = tid / L, l4 = tid - rr * L;
            vx = ld4(m.E + (size_t)iin[rr] * D + 4 * l4);
            vp = ld4(m.E + (size_t)iout[rr] * D + 4 * l4);
            vn = ld4(m.E + (size_t)ineg[rr] * D + 4 * l4);
        }
        float b0 = 0.f, b1 = 0.f, a0 = 0.f, a1 = 0.f;
        if (tid < n) {
            const uint32_t po = iout[tid], pn = ineg[tid];
            b0 = m.b[po]; b1 = m.b[pn]; a0 = m.bacc[po]; a1 = m.bacc[pn];
        }
        const int stream = tid / NGD, jx = tid % NGD;
        const int hq = (n3 + NGRP - 1) / NGRP;
        float4 ha[HQ];
#pragma unroll
        for (int i = 0; i < HQ; ++i) {
            ha[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < hq) {
                const int p = i * NGRP + grp;
                if (p < n3) {
                    const uint32_t row = (uint32_t)(kb[p] >> 32);
                    if (p == 0 || (uint32_t)(kb[p - 1] >> 32) != row) ha[i] = ld4(m.Eacc + (size_t)row * D + 4 * lg);
                }
            }
        }
        if (tid < nq) {
            st4(X + 4 * (size_t)tid, vx);
            st4(P + 4 * (size_t)tid, vp);
            st4(N + 4 * (size_t)tid, vn);
        }
        if (tid < n) { bp[tid] = b0; bn[tid] = b1; bpa[tid] = a0; bna[tid] = a1; }
        if (tid < D) Cc[tid] = 0.0f;  // c_{-1}
        __syncthreads();
        SBR_PHASE_CLOCK(0)
        /* ---- x half: P_t = bW + x_t Wx (the k order puts it first, so it does not depend on the recurrence): thread (stream, jx)
         * forms column jx of rows stream, stream + 4, ... with its column of Wx out of the run-resident copy in LDS */
        if (stream < n) {
            float wx[D];
#pragma unroll
            for (int k = 0; k < D; ++k) wx[k] = WL[(size_t)k * NGD + jx];
            const float bj = bWL[jx];
        for (int i = stream; i < n; i += NT / NGD) {
            float acc = bj;
#pragma unroll
            for (int k = 0; k < D; k += 4) {
                const float4 x = ld4(X + (size_t)i * D + k);
                acc = sbr_fma(x.x, wx[k], acc);
                acc = sbr_fma(x.y, wx[k + 1], acc);
                acc = sbr_fma(x.z, wx[k + 2], acc);
                acc = sbr_fma(x.w, wx[k + 3], acc);
            }
            PZ[(size_t)i * NGD + jx] = acc;
        }
        }
#pragma unroll
        for (int i = 0; i < HQ; ++i) {
            if (i < hq) {
                const int p = i * NGRP + grp;
                if (p < n3) st4(A + (size_t)p * D + 4 * lg, ha[i]);
            }
        }
        if (!last) index_ids(sdn, cur ^ 1);
        __syncthreads();
        /* ---- forward recurrence: wave 0 alone, no barriers (lstm_fwd_wave_seq, d = 32 with four gates); waves 1-3 meanwhile order
         * the next step's keys */
        if (wave == 0) {
            v2f w2[D];  // the h rows of gate columns lane and 64 + lane
#pragma unroll
            for (int k = 0; k < D; ++k) w2[k] = (v2f){WL[(size_t)(D + k) * NGD + lane], WL[(size_t)(D + k) * NGD + 64 + lane]};
            float c_prev = 0.0f, h_prev = 0.0f;
            for (int i = 0; i < n; ++i) {
                v2f z2 = (v2f){PZ[(size_t)i * NGD + lane], PZ[(size_t)i * NGD + 64 + lane]};
                float hk[D];
#pragma unroll
                for (int k = 0; k < D; ++k) hk[k] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(h_prev), k));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < D; ++k) z2 = pk_fma(pk_splat(hk[k]), w2[k], z2);
                const float half1 = 0.5f * z2.y;
                v2f p2, q2;
                tanh_pq_x2((v2f){0.5f * z2.x, low ? z2.y : half1}, &p2, &q2);
                const v2u sp0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p2.x), __float_as_uint(p2.x), false, false);
                const v2u sq0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(q2.x), __float_as_uint(q2.x), false, false);
                const v2u sp1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p2.y), __float_as_uint(p2.y), false, false);
                const v2u sq1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(q2.y), __float_as_uint(q2.y), false, false);
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
                float* Grow = G + (size_t)i * NGD;
                Grow[lane] = low ? gi : gf;
                Grow[64 + lane] = low ? gg : go;
                h_prev = go * sbr_tanhf(cc);
                if (low) {
                    Cc[(size_t)(i + 1) * D + u] = cc;
                    H[(size_t)i * D + u] = h_prev;
                }
            }
        } else if (!last) {
            index_rank(sdn, cur ^ 1);
        }
        __syncthreads();
        SBR_PHASE_CLOCK(1)
        /* ---- scores, loss, dloss/dh and tanh(c) of every step: a lane group per row (n <= 64 groups: one round) */
        {
            const int r = grp, rr = r < n ? r : n - 1;
            const float4 hv = ld4(H + (size_t)rr * D + 4 * lg), ep = ld4(P + (size_t)rr * D + 4 * lg), ec = ld4(N + (size_t)rr * D + 4 * lg);
            const float pos = bp[rr] + group_allreduce<L>(dot4(hv, ep));
            const float neg = bn[rr] + group_allreduce<L>(dot4(hv, ec));
            float g, l;
            if (m.loss == SBR_LOSS_BPR) l = sbr_loss_bpr(pos, neg, &g);
            else l = sbr_loss_hinge(pos, neg, &g);
            if (r < n) {
                // dloss/dh: g*E[neg] - g*E[pos], two rounded products and one subtraction (the oracle's order)
                st4(DH + (size_t)r * D + 4 * lg, make_float4(g * ec.x - g * ep.x, g * ec.y - g * ep.y, g * ec.z - g * ep.z, g * ec.w - g * ep.w));
                const float4 cv = ld4(Cc + (size_t)(r + 1) * D + 4 * lg);
                st4(TC + (size_t)r * D + 4 * lg, make_float4(sbr_tanhf(cv.x), sbr_tanhf(cv.y), sbr_tanhf(cv.z), sbr_tanhf(cv.w)));
                if (lg == 0) { coef[r] = g; lossv[r] = l; }
            }
        }
        __syncthreads();
        /* ---- backward recurrence (lstm_bwd_wave_seq: both lane groups of the wave run the cell of unit u; lane c owns column c of
         * dz W^T); dz_t over the pre-activation buffer, dX_t over dloss/dh_t.  Wave 2 meanwhile: the step's loss figures */
        if (wave == 0) {
            float wt[NGD];  // row `lane` of W: column lane of dz W^T (< D -> dX, >= D -> the recurrent dh)
#pragma unroll
            for (int j = 0; j < NGD; j += 4) {
                const float4 v = ld4(WL + (size_t)lane * NGD + j);
                wt[j] = v.x; wt[j + 1] = v.y; wt[j + 2] = v.z; wt[j + 3] = v.w;
            }
            float dc = 0.0f, rec = 0.0f;
            for (int i = n - 1; i >= 0; --i) {
                const bool carried = i + 1 < n;
                const float dh = DH[(size_t)i * D + u] + (carried ? rec : 0.0f);
                const float gi = G[(size_t)i * NGD + u], gf = G[(size_t)i * NGD + D + u], gg = G[(size_t)i * NGD + 2 * D + u], go = G[(size_t)i * NGD + 3 * D + u];
                const float tc = TC[(size_t)i * D + u];
                const float cp = Cc[(size_t)i * D + u];  // c_{t-1}
                // sbr_lstm_cell_bwd, operation for operation
                const float d_o = dh * tc;
                const float dcv = sbr_fma(dh * go, 1.0f - tc * tc, carried ? dc : 0.0f);
                const float di = dcv * gg, dg = dcv * gi;
                const float df = dcv * cp;
                dc = dcv * gf;
                float dz[4];
                dz[0] = di * (gi * (1.0f - gi));
                dz[1] = df * (gf * (1.0f - gf));
                dz[2] = dg * (1.0f - gg * gg);
                dz[3] = d_o * (go * (1.0f - go));
                float* dZrow = PZ + (size_t)i * NGD;
                // the two lane groups share the four gate rows: group 0 writes gates 0 and 2, group 1 gates 1 and 3
                {
                    const int g0 = low ? 0 : 1;
                    const float v0 = low ? dz[0] : dz[1], v1 = low ? dz[2] : dz[3];
                    dZrow[g0 * D + u] = v0; Zs[g0 * D + u] = v0;
                    dZrow[(g0 + 2) * D + u] = v1; Zs[(g0 + 2) * D + u] = v1;
                }
                wave_seq::wave_lds_fence();
                float acc = 0.0f;  // column `col` of dz W^T, j ascending from 0
#pragma unroll
                for (int j = 0; j < NGD; j += 4) {
                    const float4 zv = ld4(&Zs[j]);
                    acc = sbr_fma(zv.x, wt[j], acc);
                    acc = sbr_fma(zv.y, wt[j + 1], acc);
                    acc = sbr_fma(zv.z, wt[j + 2], acc);
                    acc = sbr_fma(zv.w, wt[j + 3], acc);
                }
                if (lane < D) DH[(size_t)i * D + lane] = acc;  // dX_t (dloss/dh_t of this step was read above)
                const v2u sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
                rec = __uint_as_float(sw.y);  // the recurrent dh of unit u = column D + u, held by lane 32 + u
                wave_seq::wave_lds_fence();
            }
        } else if (wave == 2) {
            double lp = 0.0;
            for (int r = lane; r < n; r += 64) lp += (double)lossv[r];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) lp += __shfl_xor(lp, off, 64);
            const float x = lagN[n - 1];
            float sum = 0.0f;
            for (int base = 0; base < n; base += 64) {
                const int tt = base + lane;
                const float v = tt < n ? lossv[tt] : 0.0f;
                const int cnt = n - base < 64 ? n - base : 64;
                float mine = 0.0f;
                for (int l = 0; l < cnt; ++l) {
                    sum = sum + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l));
                    mine = lane == l ? sum : mine;
                }
                if (tt < n) lagN[tt] = mine;
            }
            lag_acc = lag_acc + x;
            if (lane == 0) {
                la0 += lp; la1 += lp;
                ex0 += (unsigned long long)n; ex1 += (unsigned long long)n; ex2 += (unsigned long long)n;
                if (last) {
                    tail.header[0] = (uint32_t)n;
                    tail.header[1] = (uint32_t)n;
                    tail.header[2] = tail.header[3] = 0;
                    *reinterpret_cast<double*>(tail.header + 4) = lp;
                    *reinterpret_cast<unsigned long long*>(tail.header + 6) = (unsigned long long)n;
                    w.part_loss[0] = lp;
                    w.part_tries[0] = (unsigned int)n;
                }
            }
        }
        __syncthreads();
        SBR_PHASE_CLOCK(2)
        /* ---- sparse update: a lane group per segment of the ordered keys (as in ewma_steps_kernel) */
        for (int p0 = 0; p0 < n3; p0 += NGRP) {
            const int p = p0 + grp;
            if (p >= n3) continue;
            const uint32_t row = (uint32_t)(kb[p] >> 32);
            if (p > 0 && (uint32_t)(kb[p - 1] >> 32) == row) continue;
            float4 qw, qa;
            {
                const uint32_t src = (uint32_t)kb[p];
                const uint32_t r = src / 3, kind = src % 3;
                qw = ld4((kind == 0 ? X : (kind == 1 ? P : N)) + (size_t)r * D + 4 * lg);
                qa = ld4(A + (size_t)p * D + 4 * lg);
            }
            float4 tot = make_float4(0.f, 0.f, 0.f, 0.f), g = tot;
            float totb = 0.0f, gb = 0.0f, bv = 0.0f, ba = 0.0f;
            bool tot_first = true, tot_hb = false, first = true, has_b = false, have_bias = false;
            int in_chunk = 0;
            for (int e = p; e < n3 && (uint32_t)(kb[e] >> 32) == row; ++e) {
                const uint32_t src = (uint32_t)kb[e];
                const uint32_t r = src / 3, kind = src % 3;
                const float4 v = ld4((kind == 0 ? DH : H) + (size_t)r * D + 4 * lg);
                const float scl = kind == 0 ? 1.0f : (kind == 1 ? -coef[r] : coef[r]);
                if (first) {
                    g = make_float4(scl * v.x, scl * v.y, scl * v.z, scl * v.w);
                    first = false;
                } else {
                    g.x = g.x + scl * v.x; g.y = g.y + scl * v.y; g.z = g.z + scl * v.z; g.w = g.w + scl * v.w;
                }
                if (kind != 0) {
                    gb = has_b ? gb + scl : scl;
                    has_b = true;
                    if (!have_bias) { bv = kind == 1 ? bp[r] : bn[r]; ba = kind == 1 ? bpa[r] : bna[r]; have_bias = true; }
                }
                if (++in_chunk == SBR_SEG_CHUNK) {
                    if (tot_first) { tot = g; tot_first = false; } else { tot.x = tot.x + g.x; tot.y = tot.y + g.y; tot.z = tot.z + g.z; tot.w = tot.w + g.w; }
                    if (has_b) { totb = tot_hb ? totb + gb : gb; tot_hb = true; }
                    first = true; has_b = false; in_chunk = 0; gb = 0.0f;
                }
            }
            if (in_chunk) {
                if (tot_first) { tot = g; tot_first = false; } else { tot.x = tot.x + g.x; tot.y = tot.y + g.y; tot.z = tot.z + g.z; tot.w = tot.w + g.w; }
                if (has_b) { totb = tot_hb ? totb + gb : gb; tot_hb = true; }
            }
            sbr_adagrad(&qw.x, &qa.x, tot.x, m.lr, m.l2);
            sbr_adagrad(&qw.y, &qa.y, tot.y, m.lr, m.l2);
            sbr_adagrad(&qw.z, &qa.z, tot.z, m.lr, m.l2);
            sbr_adagrad(&qw.w, &qa.w, tot.w, m.lr, m.l2);
            st4(m.E + (size_t)row * D + 4 * lg, qw);
            st4(m.Eacc + (size_t)row * D + 4 * lg, qa);
            if (tot_hb && lg == 0) {
                sbr_adagrad(&bv, &ba, totb, m.lr, m.l2);
                m.b[row] = bv;
                m.bacc[row] = ba;
            }
        }
        SBR_PHASE_CLOCK(3)
        /* ---- dense gradient + dense update: thread (ksub, j) owns column j of the rows k = ksub + 4 e of [x ; h]: per packed row ONE
         * dz value and sixteen broadcast reads of its xh values feed sixteen independent chains (row-ascending from +0: small_back_kernel's
         * chain; xh = x_t for k < D, h_{t-1} — zero at t = 0 — beyond); the bias row is the add chain over dz.  The elements' Adagrad
         * updates follow with all their parameter / accumulator loads in flight together. */
        {
            constexpr int KS = NT / NGD, NE = K2 / KS;  // 4 phases of 16 rows each
            const int j = tid % NGD, ksub = tid / NGD;
            float acc[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) acc[e] = 0.0f;
            float accb = 0.0f;
            for (int r = 0; r < n; ++r) {
                const float z = PZ[(size_t)r * NGD + j];
                float xv[NE];
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int k = ksub + KS * e;
                    xv[e] = k < D ? X[(size_t)r * D + k] : (r > 0 ? H[(size_t)(r - 1) * D + (k - D)] : 0.0f);
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) acc[e] = sbr_fma(xv[e], z, acc[e]);
                accb = accb + z;
            }
            float wv[NE], Gv[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const size_t i = (size_t)(ksub + KS * e) * NGD + j;
                wv[e] = WL[i];
                Gv[e] = m.Wacc[i];
            }
            float bvv = 0.0f, bGv = 0.0f;
            if (tid < NGD) { bvv = bWL[tid]; bGv = m.bWacc[tid]; }
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const size_t i = (size_t)(ksub + KS * e) * NGD + j;
                sbr_adagrad(&wv[e], &Gv[e], acc[e], m.lr, m.l2);
                WL[i] = wv[e];
                m.W[i] = wv[e];
                m.Wacc[i] = Gv[e];
                if (last) blk.dense[i] = acc[e];
            }
            if (tid < NGD) {
                sbr_adagrad(&bvv, &bGv, accb, m.lr, m.l2);
                bWL[tid] = bvv;
                m.bW[tid] = bvv;
                m.bWacc[tid] = bGv;
                if (last) blk.dense[(size_t)K2 * NGD + tid] = accb;
            }
        }
        if (last) { /* the block and the work arrays of the run's last step (debug fetch, sparse stats) */
            for (int idx = tid; idx < n * L; idx += NT) {
                const int rr = idx / L, l4 = idx - rr * L;
                st4(blk.H + (size_t)rr * D + 4 * l4, ld4(H + (size_t)rr * D + 4 * l4));
                st4(blk.dX + (size_t)rr * D + 4 * l4, ld4(DH + (size_t)rr * D + 4 * l4));
                st4(w.X + (size_t)rr * D + 4 * l4, ld4(X + (size_t)rr * D + 4 * l4));
                st4(w.C + (size_t)rr * D + 4 * l4, ld4(Cc + (size_t)(rr + 1) * D + 4 * l4));
            }
            for (int idx = tid; idx < n * NGD / 4; idx += NT) {
                st4(w.G + 4 * (size_t)idx, ld4(G + 4 * (size_t)idx));
                st4(w.dZ + 4 * (size_t)idx, ld4(PZ + 4 * (size_t)idx));
            }
            if (tid < n) {
                blk.in_idx[tid] = iin[tid]; blk.out_idx[tid] = iout[tid]; blk.neg[tid] = ineg[tid];
                blk.coef[tid] = coef[tid];
                w.loss[tid] = lossv[tid];
                w.tries[tid] = 1u;
            }
            for (int e = tid; e < n3; e += NT) tail.keys_sorted[e] = kb[e];
            if (tid == 0) {
                uint32_t nh = 0;
                for (int e = 0; e < n3; ++e)
                    if (e == 0 || (uint32_t)(kb[e] >> 32) != (uint32_t)(kb[e - 1] >> 32)) tail.head_pos[nh++] = (uint32_t)e;
                tail.head_pos[nh] = (uint32_t)n3;
                *tail.nheads = nh;
            }
        }
        sd = sdn;
        phase_sync(); /* table rows and weights written above are read by the next step */
        SBR_PHASE_CLOCK(4)
    }
#undef SBR_PHASE_CLOCK
    for (int t = tid; t < max_rows; t += NT) tail.lag_state[1 + 2 * t] = lagN[t];
    if (tid == 128) {
        tail.lag_state[0] = lag_acc;
        if (tail.loss_acc) { tail.loss_acc[0] = la0; tail.loss_acc[1] = la1; tail.ex_acc[0] = ex0; tail.ex_acc[1] = ex1; tail.ex_acc[2] = ex2; }
    }
    if (prof && tid == 0) {
        for (int i = 0; i < 5; ++i) prof[i] += pc[i];
        prof[5] += (unsigned long long)(step_end - step_begin);
    }
}

/* reference-order scoring of a one-sequence step (score_refstream_kernel): every kernel width; what bounds a step is the workgroup's
 * LDS — the step's h rows, a 64-draw candidate window, four words per row beside 12 KB of static arrays (the step's key ordering):
 * 255 rows at d <= 64, 220 at d = 128, 80 at d = 256 */
#define SBR_REFSTREAM_MAX_LDS (147 * 1024)
static size_t score_refstream_lds(int d, int rows) { return ((size_t)rows * d + 64 * (size_t)d + 4 * (size_t)rows + 64 * 6) * 4; }
bool reference_order_shape_ok(int d, int max_rows) {
    return (d == 16 || d == 32 || d == 64 || d == 128 || d == 256) && max_rows > 0 && max_rows <= SBR_SMALL_TAIL_MAX_ROWS &&
           score_refstream_lds(d, max_rows) <= SBR_REFSTREAM_MAX_LDS;
}
/* false: the shape has no such form */
bool launch_score_reference_order(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, uint32_t* rng_state,
                                  int rows_host, hipStream_t s, const SmallTail& tail) {
    if (!small_tail_shape_ok(m, mb.B, rows_host, /*wide=*/true) || !reference_order_shape_ok(m.d, rows_host)) return false;
    const size_t lds = score_refstream_lds(m.d, rows_host);
#define SBR_REFSTREAM(DD)                                                                                                                    \
    {                                                                                                                                        \
        static std::atomic<size_t> granted[64]; /* dynamic LDS beyond 64 KB is granted per kernel and device, once */                        \
        int dev = 0;                                                                                                                         \
        (void)hipGetDevice(&dev);                                                                                                            \
        dev = dev >= 0 && dev < 64 ? dev : 0;                                                                                                \
        if (lds > 65536 && lds > granted[dev]) {                                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(score_refstream_kernel<DD>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)SBR_REFSTREAM_MAX_LDS);                                                                           \
            granted[dev] = SBR_REFSTREAM_MAX_LDS;                                                                                            \
        }                                                                                                                                    \
        hipLaunchKernelGGL((score_refstream_kernel<DD>), dim3(1), dim3(256), lds, s, m, mb, blk, w, rng_state, tail);                        \
    }
    switch (m.d) {
        case 16: SBR_REFSTREAM(16) break;
        case 32: SBR_REFSTREAM(32) break;
        case 64: SBR_REFSTREAM(64) break;
        case 128: SBR_REFSTREAM(128) break;
        default: SBR_REFSTREAM(256) break;
    }
#undef SBR_REFSTREAM
    return true;
}

/* dynamic LDS of ewma_steps_kernel for steps of at most max_rows rows (0: the shape cannot take the one-launch form) */
static size_t epoch_steps_lds(const ModelView& m, int max_rows) {
    if ((m.d != 16 && m.d != 32) || max_rows <= 0 || max_rows > SBR_EWMA_STEPS_MAX_ROWS) return 0;
    if (m.optimizer != SBR_OPT_ADAGRAD || m.ng != 0 || m.loss == SBR_LOSS_WARP) return 0;
    /* [alpha | alpha_acc | lag nodes] + the step's rows, optimiser state, ids, coefficients and keys */
    const size_t fl = SBR_EWMA_STEPS_LDS_FLOATS(max_rows, m.d);
    return fl * 4 <= SBR_EPOCH_STEPS_MAX_LDS ? fl * 4 : 0;
}
bool epoch_steps_shape_ok(const ModelView& m, int max_rows) { return epoch_steps_lds(m, max_rows) != 0; }
void launch_epoch_steps(const ModelView& m, const EpochView& ev, const BlockView& blk, const WorkView& w, uint64_t epoch_key,
                        const SmallTail& tail, int step_begin, int step_end, int max_rows, unsigned long long* prof, hipStream_t s) {
    const size_t lds = epoch_steps_lds(m, max_rows);
    if (!lds || step_end <= step_begin) return;
#define SBR_EWMA_STEPS(DD)                                                                                                          \
    {                                                                                                                               \
        static std::atomic<size_t> granted[64]; /* dynamic LDS beyond 64 KB is granted per kernel and device, once */                     \
        int dev = 0;                                                                                                                \
        (void)hipGetDevice(&dev);                                                                                                   \
        dev = dev >= 0 && dev < 64 ? dev : 0;                                                                                       \
        if (lds > granted[dev]) {                                                                                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ewma_steps_kernel<DD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            granted[dev] = lds;                                                                                                     \
        }                                                                                                                           \
        hipLaunchKernelGGL((ewma_steps_kernel<DD>), dim3(1), dim3(256), lds, s, m, ev, blk, w, epoch_key, tail, step_begin, step_end, max_rows, prof); \
    }
    if (m.d == 32) SBR_EWMA_STEPS(32) else SBR_EWMA_STEPS(16)
#undef SBR_EWMA_STEPS
}

/* LSTM step runs (lstm_steps_kernel): which models, and how many rows a step of the run may have */
bool lstm_steps_shape_ok(const ModelView& m, int lag_rows) {
    return m.d == 32 && m.ng == 4 && m.optimizer == SBR_OPT_ADAGRAD && m.loss != SBR_LOSS_WARP && lag_rows > 0 &&
           lag_rows <= SBR_SMALL_TAIL_MAX_ROWS;
}
int lstm_steps_max_rows() { return SBR_LSTM_STEPS_MAX_ROWS; }
void launch_lstm_steps(const ModelView& m, const EpochView& ev, const BlockView& blk, const WorkView& w, uint64_t epoch_key,
                       const SmallTail& tail, int step_begin, int step_end, int lag_rows, int run_max_rows, unsigned long long* prof,
                       hipStream_t s) {
    if (step_end <= step_begin || run_max_rows <= 0 || run_max_rows > SBR_LSTM_STEPS_MAX_ROWS) return;
    /* [loss nodes | dz | the largest step's arrays] */
    const size_t lds = (SBR_LSTM_STEPS_LDS_FLOATS(run_max_rows) - (size_t)((run_max_rows + 3) & ~3) + (size_t)((lag_rows + 3) & ~3)) * 4;
    static std::atomic<size_t> granted[64]; /* dynamic LDS beyond 64 KB is granted per kernel and device, once */
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev >= 0 && dev < 64 ? dev : 0;
    if (lds > granted[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_steps_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        granted[dev] = lds;
    }
    hipLaunchKernelGGL(lstm_steps_kernel, dim3(1), dim3(512), lds, s, m, ev, blk, w, epoch_key, tail, step_begin, step_end, lag_rows, prof);
    launch_repack_lstm(m, s); /* the packed weight copies follow the run's last dense update */
}

}  // namespace sbr
