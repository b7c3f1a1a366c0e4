// sbr_sort.hip — key ordering of the sparse update on gfx950: the (row, entry) keys of a minibatch's
// gradient entries in (row, entry) order plus the list of row-segment heads, hand-written.
//
// What is ordered.  Entry e = 3 r + kind of packed row r (kind 0 = input row -> dX, 1 = target row, 2 = sampled
// negative) has the key (table row << 32) | e.  The optimiser visits every touched table row once and adds its
// entries in entry order (sbr_numerics.h, SBR_SEG_CHUNK); that is sequence_model.rs:163-169's optimizer.step over the
// sparse gradients of the graph (/root/reference/src/models/sequence_model.rs:160-169, lstm.rs:272-291).
//
// How.  The entries are GENERATED in entry order, so a STABLE sort on the row bits alone yields (row, entry) order:
// an LSD radix sort over ceil(log2 num_items) bits — one pass of <= 11 bits for a MovieLens-size catalogue, two
// 10-bit passes for 1e6 items, three for 1e7 — instead of a general 64-bit sort over all 52 key bits.  The first
// pass reads the index arrays directly (no key array is materialised first).  Per pass:
//   radix_hist     one wave per 4096-key tile: digit histogram in LDS (ds_add), written to counts[bin][tile]
//   radix_binscan  one wave per bin: exclusive scan over the tiles (contiguous), bin totals
//   radix_scatter  one wave per tile: bin bases (a 64-lane scan of the totals) + the tile's offsets into LDS, then the
//                  keys in order, 64 at a time: lanes with equal digits find each other with one v_cmp ballot per
//                  digit bit (no LDS atomics with a return value: the rank inside the wave is a population count, so the
//                  order is the entry order by construction), the highest lane of a digit group advances its offset
// and afterwards the segment heads (positions whose row differs from the previous key's) are listed in ascending
// order by a count / scan / write triple over the same tiles.  Nothing here depends on dispatch order or on
// forward progress of other workgroups, so the kernels can run beside the MFMA-bound kernels on another stream.
// Everything is integer work: results are exact and identical to any other stable sort.

#include "sbr_kernels.h"

#include <cstdlib>

#include "../../include/sbr_hip.h"
#include "sbr_numerics.h"

namespace sbr {

namespace {

constexpr int SORT_ITEMS = 64;                 // keys per lane of a wave-tile: 4 096-key tiles ...
constexpr int SORT_ITEMS_FINE = 8;             // ... or 512-key tiles while the input is small (SORT_FINE_MAX_N): a step of a few
                                               // hundred sequences is 10^4-10^5 keys — six 4 096-key tiles would leave the
                                               // ordering to six waves (150 us per step at 25 000 keys, under a 100 us BPTT)
#ifndef SBR_SORT_FINE_MAX_LOG2
#define SBR_SORT_FINE_MAX_LOG2 18
#endif
constexpr uint32_t SORT_FINE_MAX_N = 1u << SBR_SORT_FINE_MAX_LOG2;
constexpr int SORT_BATCH = 8;                  // keys per lane requested together
constexpr int SORT_MAX_DIGIT_BITS = 11;
/* wave priority of the ordering's kernels (s_setprio 0..3): they run on their own stream underneath BPTT, whose MFMA waves are
 * older and win the issue arbitration of every SIMD (MI355X_MICROARCH.md "VALU issue is arbitrated ... by priority, then age") —
 * a ~0.1 ms job then takes the whole backward pass.  A/B: profiles/r04_tail_experiments.md */
#ifndef SBR_SORT_SETPRIO
#define SBR_SORT_SETPRIO 1
#endif
#define SORT_PRIO() do { if (SBR_SORT_SETPRIO) __builtin_amdgcn_s_setprio(SBR_SORT_SETPRIO); } while (0)
constexpr int SORT_WAVES = 4;                  // wave-tiles per workgroup

struct PassPlan {
    int passes;
    int digit_bits;  // same width for every pass (>= 6: the scatter kernel scans the bin totals with 64 lanes)
};
inline PassPlan plan_passes(int row_bits) {
    if (row_bits < 1) row_bits = 1;
    if (row_bits > 32) row_bits = 32;
    PassPlan p;
    p.passes = (row_bits + SORT_MAX_DIGIT_BITS - 1) / SORT_MAX_DIGIT_BITS;
    p.digit_bits = (row_bits + p.passes - 1) / p.passes;
    if (p.digit_bits < 6) p.digit_bits = 6;
    return p;
}
inline int items_of(uint64_t n) {  // keys per lane of a tile for an input of n keys (SBR_SORT_ITEMS = 64 / 8 forces one: tests)
    const char* e = std::getenv("SBR_SORT_ITEMS");
    if (e && std::atoi(e) == SORT_ITEMS) return SORT_ITEMS;
    if (e && std::atoi(e) == SORT_ITEMS_FINE) return n <= SORT_FINE_MAX_N ? SORT_ITEMS_FINE : SORT_ITEMS;  // (the scratch is sized for that)
    return n <= SORT_FINE_MAX_N ? SORT_ITEMS_FINE : SORT_ITEMS;
}
inline uint32_t tiles_of(uint64_t n, int items) { return (uint32_t)((n + 64 * (uint64_t)items - 1) / (64 * (uint64_t)items)); }

// ---- key sources -------------------------------------------------------------------------------
struct SrcBlock {  // a device's own entries after the score kernel has chosen the negatives
    const uint32_t *in_idx, *out_idx, *neg;
    __device__ __forceinline__ uint64_t operator()(uint32_t e) const {
        const uint32_t r = e / 3u, kind = e - 3u * r;
        const uint32_t* a = kind == 0 ? in_idx : (kind == 1 ? out_idx : neg);
        return ((uint64_t)a[r] << 32) | e;
    }
};
struct SrcEarly {  // single-negative losses: the negative is a hash of the row counter, known before the forward pass
    const uint32_t *in_idx, *out_idx, *ctr;
    uint64_t epoch_key;
    uint32_t num_items;
    __device__ __forceinline__ uint64_t operator()(uint32_t e) const {
        const uint32_t r = e / 3u, kind = e - 3u * r;
        const uint32_t row = kind == 0 ? in_idx[r] : (kind == 1 ? out_idx[r] : sbr_neg_draw(epoch_key, ctr[r], 0u, num_items));
        return ((uint64_t)row << 32) | e;
    }
};
struct SrcKeys {  // a key array (passes after the first)
    const uint64_t* k;
    __device__ __forceinline__ uint64_t operator()(uint32_t e) const { return k[e]; }
};
struct SrcRows {  // self-test: entry e of a plain row array
    const uint32_t* rows;
    __device__ __forceinline__ uint64_t operator()(uint32_t e) const { return ((uint64_t)rows[e] << 32) | e; }
};
struct SrcMerge {  // partitioned table: (row, device, position) of every list head in the owner's row range; where that range
    PeerLists pl;   // lies in every device's sorted keys is DEVICE-resident (merge_plan_kernel): the host never learns it
    int ndev;
    __device__ __forceinline__ uint64_t operator()(uint32_t i) const {
        const MergePlan& mp = *pl.plan;
        int r = 0;
        while (r + 1 < ndev && i >= mp.base[r + 1]) ++r;
        const uint32_t p = mp.lo[r] + (i - mp.base[r]);
        return (pl.fl[r][p] & 1u) ? ((pl.keys[r][p] >> 32) << 32) | ((uint64_t)r << 28) | (uint64_t)p : ~0ull;
    }
};

// LDS traffic between the lanes of ONE wave (a bin's running offset: every lane of a digit group reads it, the group's highest
// lane updates it, the next round reads it again): the hardware executes a wave's LDS instructions in order, so no barrier is
// needed — but the compiler must keep the program order and may not cache the value in a register across the round
__device__ __forceinline__ void wave_lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// ---- one radix pass ----------------------------------------------------------------------------
template <class Src, int ITEMS>
__global__ __launch_bounds__(SORT_WAVES * 64) void radix_hist_kernel(Src src, uint32_t n, int shift, int digit_bits, uint32_t ntiles,
                                                                     uint32_t* __restrict__ counts, const uint32_t* __restrict__ n_dev) {
    SORT_PRIO();
    if (n_dev) {  // the key count lives on the device (the launch is sized for the capacity): surplus workgroups leave at once
        n = *n_dev;
        ntiles = (uint32_t)(((uint64_t)n + 64 * ITEMS - 1) / (64 * ITEMS));
        if (blockIdx.x * SORT_WAVES >= ntiles) return;
    }
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nb = 1u << digit_bits, mask = nb - 1u;
    uint32_t* h = lds + (size_t)wave * nb;
    for (uint32_t b = lane; b < nb; b += 64) h[b] = 0u;
    __syncthreads();
    const uint32_t tile = blockIdx.x * SORT_WAVES + wave;
    const uint64_t base = (uint64_t)tile * (64 * ITEMS);
    for (int i0 = 0; i0 < ITEMS; i0 += SORT_BATCH) {
        uint64_t k[SORT_BATCH];
#pragma unroll
        for (int j = 0; j < SORT_BATCH; ++j) {
            const uint64_t e = base + (uint64_t)(i0 + j) * 64 + lane;
            k[j] = e < n ? src((uint32_t)e) : 0ull;
        }
#pragma unroll
        for (int j = 0; j < SORT_BATCH; ++j) {
            const uint64_t e = base + (uint64_t)(i0 + j) * 64 + lane;
            if (e < n) atomicAdd(&h[((uint32_t)(k[j] >> 32) >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    if (tile < ntiles)
        for (uint32_t b = lane; b < nb; b += 64) counts[(size_t)b * ntiles + tile] = h[b];
}

// one wave per bin: exclusive scan of the bin's tile counts (in place) and the bin total
__global__ __launch_bounds__(256) void radix_binscan_kernel(uint32_t* __restrict__ counts, uint32_t ntiles, uint32_t nb,
                                                            uint32_t* __restrict__ bintotal, const uint32_t* __restrict__ n_dev, int items) {
    SORT_PRIO();
    if (n_dev) ntiles = (uint32_t)(((uint64_t)*n_dev + 64 * (uint64_t)items - 1) / (64 * (uint64_t)items));
    const int lane = threadIdx.x & 63;
    const uint32_t bin = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (bin >= nb) return;
    uint32_t* row = counts + (size_t)bin * ntiles;
    uint32_t running = 0;
    constexpr int U = 8;  // 512 tiles' counts requested together
    for (uint32_t t0 = 0; t0 < ntiles; t0 += 64 * U) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t t = t0 + u * 64 + lane;
            v[u] = t < ntiles ? row[t] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t t = t0 + u * 64 + lane;
            const uint32_t incl = wave_inclusive_scan(v[u], lane);
            if (t < ntiles) row[t] = running + incl - v[u];
            running += (uint32_t)__shfl((int)incl, 63, 64);
        }
    }
    if (lane == 0) bintotal[bin] = running;
}

template <class Src, int ITEMS>
__global__ __launch_bounds__(SORT_WAVES * 64) void radix_scatter_kernel(Src src, uint32_t n, int shift, int digit_bits, uint32_t ntiles,
                                                                        const uint32_t* __restrict__ counts,
                                                                        const uint32_t* __restrict__ bintotal, uint64_t* __restrict__ out,
                                                                        const uint32_t* __restrict__ n_dev) {
    SORT_PRIO();
    if (n_dev) {
        n = *n_dev;
        ntiles = (uint32_t)(((uint64_t)n + 64 * ITEMS - 1) / (64 * ITEMS));
        if (blockIdx.x * SORT_WAVES >= ntiles) return;
    }
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nb = 1u << digit_bits, mask = nb - 1u;
    uint32_t* h = lds + (size_t)wave * nb;
    const uint32_t tile = blockIdx.x * SORT_WAVES + wave;
    const uint32_t tt = tile < ntiles ? tile : ntiles - 1;  // surplus waves compute on a valid tile and store nothing
    {   // h[bin] = (keys of lower bins) + (keys of this bin in earlier tiles): lane l owns bins l*per .. l*per + per - 1
        const uint32_t per = nb >> 6;
        uint32_t sum = 0;
        for (uint32_t j = 0; j < per; ++j) sum += bintotal[lane * per + j];
        uint32_t run = wave_inclusive_scan(sum, lane) - sum;
        for (uint32_t j = 0; j < per; ++j) {
            const uint32_t b = lane * per + j;
            h[b] = run + counts[(size_t)b * ntiles + tt];
            run += bintotal[b];
        }
    }
    __syncthreads();
    const uint64_t base = (uint64_t)tile * (64 * ITEMS);
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int i0 = 0; i0 < ITEMS; i0 += SORT_BATCH) {
        uint64_t k[SORT_BATCH];
#pragma unroll
        for (int j = 0; j < SORT_BATCH; ++j) {
            const uint64_t e = base + (uint64_t)(i0 + j) * 64 + lane;
            k[j] = (tile < ntiles && e < n) ? src((uint32_t)e) : 0ull;
        }
#pragma unroll
        for (int j = 0; j < SORT_BATCH; ++j) {
            const uint64_t e = base + (uint64_t)(i0 + j) * 64 + lane;
            const bool valid = tile < ntiles && e < n;
            const uint32_t d = ((uint32_t)(k[j] >> 32) >> shift) & mask;
            uint64_t same = __ballot(valid);  // lanes of this round that hold the same digit
            for (int b = 0; b < digit_bits; ++b) {
                const bool bit = (d >> b) & 1u;
                const uint64_t vote = __ballot(bit);
                same &= bit ? vote : ~vote;
            }
            if (valid) {
                const uint32_t pos = h[d] + (uint32_t)__popcll(same & lt);
                out[pos] = k[j];
                if ((same >> lane) == 1ull) h[d] += (uint32_t)__popcll(same);  // the group's highest lane, after every lane's read
            }
            wave_lds_order();  // the next round's h[d] reads must see this write: one wave, LDS in program order, compiler fenced
        }
    }
}

// ---- segment heads -----------------------------------------------------------------------------
template <bool WRITE, int ITEMS>
__global__ __launch_bounds__(SORT_WAVES * 64) void head_tiles_kernel(const uint64_t* __restrict__ keys, uint32_t n, uint32_t ntiles,
                                                                     uint32_t* __restrict__ tile_heads, uint32_t* __restrict__ head_pos) {
    SORT_PRIO();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * SORT_WAVES + wave;
    if (tile >= ntiles) return;
    const uint64_t base = (uint64_t)tile * (64 * ITEMS);
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t off = WRITE ? tile_heads[tile] : 0u;
    for (int i0 = 0; i0 < ITEMS; i0 += SORT_BATCH) {
        uint32_t row[SORT_BATCH], prev[SORT_BATCH];
#pragma unroll
        for (int j = 0; j < SORT_BATCH; ++j) {
            const uint64_t p = base + (uint64_t)(i0 + j) * 64 + lane;
            row[j] = p < n ? (uint32_t)(keys[p] >> 32) : 0u;
            prev[j] = (p < n && p > 0) ? (uint32_t)(keys[p - 1] >> 32) : 0u;
        }
#pragma unroll
        for (int j = 0; j < SORT_BATCH; ++j) {
            const uint64_t p = base + (uint64_t)(i0 + j) * 64 + lane;
            const bool head = p < n && (p == 0 || row[j] != prev[j]);
            const uint64_t m = __ballot(head);
            if (WRITE && head) head_pos[off + (uint32_t)__popcll(m & lt)] = (uint32_t)p;
            off += (uint32_t)__popcll(m);
        }
    }
    if (!WRITE && lane == 0) tile_heads[tile] = off;
}

// exclusive scan of the tiles' head counts (in place), the head count and the sentinel head_pos[nheads] = n
__global__ __launch_bounds__(1024) void head_scan_kernel(uint32_t* __restrict__ tile_heads, uint32_t ntiles, uint32_t n,
                                                         uint32_t* __restrict__ nheads, uint32_t* __restrict__ head_pos) {
    SORT_PRIO();
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < ntiles; t0 += 1024) {
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < ntiles ? tile_heads[t] : 0u;
        const uint32_t incl = wave_inclusive_scan(v, lane);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (int w2 = 0; w2 < wave; ++w2) before += s_wave[w2];
        if (t < ntiles) tile_heads[t] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *nheads = s_carry;
        head_pos[s_carry] = n;
    }
}

// ---- small inputs: everything in ONE launch -------------------------------------------------------
// Up to SMALL_N keys (a minibatch of one or a few sequences — the reference's own schedule is one subsequence per
// optimiser step, sequence_model.rs:111-169, i.e. ~10^2 keys) fit the LDS of one workgroup twice over: the workgroup
// generates the keys into LDS and runs every radix pass there — each of its eight waves owns a contiguous eighth of the
// keys (its own histogram, then, after a scan over (bin, wave), the same ballot-ranked stable scatter as
// radix_scatter_kernel over its slice; ping-pong between two LDS arrays) — writes the ordered keys out and lists the
// segment heads: nine launches of the general path become one, which is what a step of ~10^2 rows is made of.  (As ONE
// wave the 1 620 keys of a 16-sequence step took 48 us, as long as the step's backward pass.)
constexpr int SMALL_N = 4096;
constexpr int SMALL_WAVES = 8;  // wave w orders the w-th eighth of the keys; the bins' offsets are exchanged through LDS
template <class Src>
__global__ __launch_bounds__(SMALL_WAVES * 64) void small_sort_kernel(Src src, uint32_t n, int passes, int digit_bits, uint64_t* __restrict__ out,
                                                                      uint32_t* __restrict__ head_pos, uint32_t* __restrict__ nheads) {
    constexpr int NT = SMALL_WAVES * 64;
    __shared__ uint64_t ka[SMALL_N], kb[SMALL_N];
    __shared__ uint32_t h[SMALL_WAVES][1 << SORT_MAX_DIGIT_BITS];  // h[w][bin]: count, then first output position, of wave w's keys of the bin
    __shared__ uint32_t s_part[SMALL_WAVES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t nb = 1u << digit_bits, mask = nb - 1u;
    const uint64_t lt = (1ull << lane) - 1ull;
    // contiguous slices in entry order (multiples of 64 keys): stability = lower waves' keys of a bin come first
    const uint32_t slice = ((n + SMALL_WAVES * 64 - 1) / (SMALL_WAVES * 64)) * 64;
    const uint32_t lo = wave * slice < n ? wave * slice : n, hi = lo + slice < n ? lo + slice : n;
    for (uint32_t e = tid; e < n; e += NT) ka[e] = src(e);
    uint64_t* from = ka;
    uint64_t* to = kb;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * digit_bits;
        for (uint32_t i = tid; i < SMALL_WAVES * nb; i += NT) h[i / nb][i % nb] = 0u;
        __syncthreads();  // (also: the keys of `from` are in place)
        for (uint32_t e = lo + lane; e < hi; e += 64) atomicAdd(&h[wave][((uint32_t)(from[e] >> 32) >> shift) & mask], 1u);
        __syncthreads();
        {   // h[w][b] <- (keys of lower bins) + (keys of bin b in lower waves): thread t owns bins t*per .. t*per + per - 1
            const uint32_t per = (nb + NT - 1) / NT;
            const uint32_t b0 = tid * per, b1 = b0 + per < nb ? b0 + per : nb;
            uint32_t sum = 0;
            for (uint32_t b = b0; b < b1; ++b)
#pragma unroll
                for (int w2 = 0; w2 < SMALL_WAVES; ++w2) sum += h[w2][b];
            const uint32_t incl = wave_inclusive_scan(sum, lane);
            if (lane == 63) s_part[wave] = incl;
            __syncthreads();
            uint32_t run = incl - sum;
            for (int w2 = 0; w2 < wave; ++w2) run += s_part[w2];
            for (uint32_t b = b0; b < b1; ++b) {
#pragma unroll
                for (int w2 = 0; w2 < SMALL_WAVES; ++w2) {
                    const uint32_t c = h[w2][b];
                    h[w2][b] = run;
                    run += c;
                }
            }
        }
        __syncthreads();
        for (uint32_t e0 = lo; e0 < hi; e0 += 64) {  // the wave's slice in order, 64 keys at a time (radix_scatter_kernel's ranking)
            const uint32_t e = e0 + lane;
            const bool valid = e < hi;
            const uint64_t k = valid ? from[e] : 0ull;
            const uint32_t d = ((uint32_t)(k >> 32) >> shift) & mask;
            uint64_t same = __ballot(valid);
            for (int b = 0; b < digit_bits; ++b) {
                const bool bit = (d >> b) & 1u;
                const uint64_t vote = __ballot(bit);
                same &= bit ? vote : ~vote;
            }
            if (valid) {
                to[h[wave][d] + (uint32_t)__popcll(same & lt)] = k;
                if ((same >> lane) == 1ull) h[wave][d] += (uint32_t)__popcll(same);
            }
            wave_lds_order();
        }
        __syncthreads();
        uint64_t* t = from; from = to; to = t;
    }
    // ordered keys out; segment heads in ascending order: heads per slice, offsets across the waves, then the positions
    uint32_t heads = 0;
    for (uint32_t e0 = lo; e0 < hi; e0 += 64) {
        const uint32_t e = e0 + lane;
        const bool valid = e < hi;
        const bool head = valid && (e == 0 || (uint32_t)(from[e] >> 32) != (uint32_t)(from[e - 1] >> 32));
        if (valid) out[e] = from[e];
        heads += (uint32_t)__popcll(__ballot(head));
    }
    if (lane == 0) s_part[wave] = heads;
    __syncthreads();
    uint32_t off = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < SMALL_WAVES; ++w2) {
        if (w2 < wave) off += s_part[w2];
        total += s_part[w2];
    }
    for (uint32_t e0 = lo; e0 < hi; e0 += 64) {
        const uint32_t e = e0 + lane;
        const bool valid = e < hi;
        const bool head = valid && (e == 0 || (uint32_t)(from[e] >> 32) != (uint32_t)(from[e - 1] >> 32));
        const uint64_t mm = __ballot(head);
        if (head) head_pos[off + (uint32_t)__popcll(mm & lt)] = e;
        off += (uint32_t)__popcll(mm);
    }
    if (tid == 0) {
        *nheads = total;
        head_pos[total] = n;
    }
}

struct Scratch {
    uint32_t *counts, *bintotal, *tile_heads;
};
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline Scratch carve(void* temp, size_t max_entries, int row_bits) {
    const PassPlan pp = plan_passes(row_bits);
    const size_t nb = (size_t)1 << pp.digit_bits, nt = tiles_of(max_entries, items_of(max_entries)) + 1;
    uint8_t* p = reinterpret_cast<uint8_t*>(temp);
    Scratch s;
    s.counts = reinterpret_cast<uint32_t*>(p);
    p += align256(nb * nt * 4);
    s.bintotal = reinterpret_cast<uint32_t*>(p);
    p += align256(nb * 4);
    s.tile_heads = reinterpret_cast<uint32_t*>(p);
    return s;
}

// stable sort of n generated keys by their row bits: first pass from `first`, later passes between the two buffers,
// the last pass always writes `out`
// n_dev != null: `n` is the CAPACITY the launches are sized for and the key count is read from device memory by every kernel
template <class Src, int ITEMS>
void radix_sort_tiles(const Src& first, uint32_t n, int row_bits, uint64_t* tmp, uint64_t* out, const Scratch& sc, hipStream_t s,
                      const uint32_t* n_dev = nullptr) {
    const PassPlan pp = plan_passes(row_bits);
    const uint32_t ntiles = tiles_of(n, ITEMS), nb = 1u << pp.digit_bits;
    const unsigned grid = (ntiles + SORT_WAVES - 1) / SORT_WAVES;
    const size_t lds = (size_t)SORT_WAVES * nb * 4;
    for (int p = 0; p < pp.passes; ++p) {
        const int shift = p * pp.digit_bits;
        uint64_t* dst = ((pp.passes - 1 - p) & 1) ? tmp : out;
        const uint64_t* from = dst == out ? tmp : out;
        if (p == 0) {
            hipLaunchKernelGGL((radix_hist_kernel<Src, ITEMS>), dim3(grid), dim3(SORT_WAVES * 64), lds, s, first, n, shift, pp.digit_bits, ntiles, sc.counts, n_dev);
        } else {
            hipLaunchKernelGGL((radix_hist_kernel<SrcKeys, ITEMS>), dim3(grid), dim3(SORT_WAVES * 64), lds, s, SrcKeys{from}, n, shift, pp.digit_bits, ntiles,
                               sc.counts, n_dev);
        }
        hipLaunchKernelGGL(radix_binscan_kernel, dim3((nb * 64 + 255) / 256), dim3(256), 0, s, sc.counts, ntiles, nb, sc.bintotal, n_dev, ITEMS);
        if (p == 0) {
            hipLaunchKernelGGL((radix_scatter_kernel<Src, ITEMS>), dim3(grid), dim3(SORT_WAVES * 64), lds, s, first, n, shift, pp.digit_bits, ntiles, sc.counts,
                               sc.bintotal, dst, n_dev);
        } else {
            hipLaunchKernelGGL((radix_scatter_kernel<SrcKeys, ITEMS>), dim3(grid), dim3(SORT_WAVES * 64), lds, s, SrcKeys{from}, n, shift, pp.digit_bits, ntiles,
                               sc.counts, sc.bintotal, dst, n_dev);
        }
    }
}
// stable sort of n generated keys by their row bits: first pass from `first`, later passes between the two buffers,
// the last pass always writes `out`
template <class Src>
void radix_sort(const Src& first, uint32_t n, int row_bits, uint64_t* tmp, uint64_t* out, const Scratch& sc, hipStream_t s,
                const uint32_t* n_dev = nullptr) {
    if (items_of(n) == SORT_ITEMS_FINE) radix_sort_tiles<Src, SORT_ITEMS_FINE>(first, n, row_bits, tmp, out, sc, s, n_dev);
    else radix_sort_tiles<Src, SORT_ITEMS>(first, n, row_bits, tmp, out, sc, s, n_dev);
}

// keys in (row, entry) order + segment heads: one launch for small inputs, the tiled passes otherwise
template <class Src>
void sort_and_list(const Src& first, uint32_t n, int row_bits, uint64_t* tmp, uint64_t* out, const Scratch& sc, uint32_t* head_pos,
                   uint32_t* nheads, hipStream_t s);

template <int ITEMS>
void list_heads_tiles(const uint64_t* keys_sorted, uint32_t n, const Scratch& sc, uint32_t* head_pos, uint32_t* nheads, hipStream_t s) {
    const uint32_t ntiles = tiles_of(n, ITEMS);
    const unsigned grid = (ntiles + SORT_WAVES - 1) / SORT_WAVES;
    hipLaunchKernelGGL((head_tiles_kernel<false, ITEMS>), dim3(grid), dim3(SORT_WAVES * 64), 0, s, keys_sorted, n, ntiles, sc.tile_heads, head_pos);
    hipLaunchKernelGGL(head_scan_kernel, dim3(1), dim3(1024), 0, s, sc.tile_heads, ntiles, n, nheads, head_pos);
    hipLaunchKernelGGL((head_tiles_kernel<true, ITEMS>), dim3(grid), dim3(SORT_WAVES * 64), 0, s, keys_sorted, n, ntiles, sc.tile_heads, head_pos);
}
void list_heads(const uint64_t* keys_sorted, uint32_t n, const Scratch& sc, uint32_t* head_pos, uint32_t* nheads, hipStream_t s) {
    if (items_of(n) == SORT_ITEMS_FINE) list_heads_tiles<SORT_ITEMS_FINE>(keys_sorted, n, sc, head_pos, nheads, s);
    else list_heads_tiles<SORT_ITEMS>(keys_sorted, n, sc, head_pos, nheads, s);
}

template <class Src>
void sort_and_list(const Src& first, uint32_t n, int row_bits, uint64_t* tmp, uint64_t* out, const Scratch& sc, uint32_t* head_pos,
                   uint32_t* nheads, hipStream_t s) {
    const char* small_env = std::getenv("SBR_SORT_SMALL"); /* read per call: the tests run the tiled path on small inputs too */
    const bool small_on = !(small_env && small_env[0] == '0');
    if (n <= (uint32_t)SMALL_N && small_on) {
        const PassPlan pp = plan_passes(row_bits);
        hipLaunchKernelGGL((small_sort_kernel<Src>), dim3(1), dim3(SMALL_WAVES * 64), 0, s, first, n, pp.passes, pp.digit_bits, out, head_pos, nheads);
        return;
    }
    radix_sort(first, n, row_bits, tmp, out, sc, s);
    list_heads(out, n, sc, head_pos, nheads, s);
}

}  // namespace

size_t sparse_sort_temp_bytes(size_t max_entries, int key_bits) {
    const int row_bits = key_bits - 32;
    const PassPlan pp = plan_passes(row_bits);
    const size_t nb = (size_t)1 << pp.digit_bits;
    /* the larger of the two tile granularities: 4 096-key tiles for max_entries, 512-key tiles for what of it is ordered that way */
    const size_t fine_n = max_entries < SORT_FINE_MAX_N ? max_entries : SORT_FINE_MAX_N;
    const size_t nt_c = tiles_of(max_entries, SORT_ITEMS) + 1, nt_f = tiles_of(fine_n, SORT_ITEMS_FINE) + 1;
    const size_t nt = nt_c > nt_f ? nt_c : nt_f;
    return align256(nb * nt * 4) + align256(nb * 4) + align256((nt + 1) * 4);
}

void launch_own_sort(const BlockView& blk, uint32_t rows_host, uint64_t* keys, uint64_t* keys_sorted, void* sort_temp,
                     size_t sort_temp_bytes, int key_bits, const SegScratch& sc, hipStream_t s, const MbView* early_mb,
                     uint64_t epoch_key, uint32_t num_items) {
    (void)sort_temp_bytes;
    if (rows_host == 0) return;
    const uint32_t total = 3u * rows_host;
    const int row_bits = key_bits - 32;
    const Scratch scr = carve(sort_temp, total, row_bits);
    if (early_mb)
        sort_and_list(SrcEarly{early_mb->in_idx, early_mb->out_idx, early_mb->ctr, epoch_key, num_items}, total, row_bits, keys, keys_sorted, scr,
                      sc.head_pos, sc.nheads, s);
    else
        sort_and_list(SrcBlock{blk.in_idx, blk.out_idx, blk.neg}, total, row_bits, keys, keys_sorted, scr, sc.head_pos, sc.nheads, s);
}

// `capacity` keys at most; the count itself is pl.plan->base[16] on the device (merge_plan_kernel): launches sized for the
// capacity, the tile granularity chosen by it as well (the scratch is carved for it)
void launch_merge_sort(const PeerLists& pl, int ndev, uint32_t capacity, uint64_t* mkeys, uint64_t* mkeys_sorted, void* sort_temp,
                       size_t sort_temp_bytes, hipStream_t s) {
    (void)sort_temp_bytes;
    if (capacity == 0) return;
    radix_sort(SrcMerge{pl, ndev}, capacity, 32, mkeys, mkeys_sorted, carve(sort_temp, capacity, 32), s, &pl.plan->base[16]);
}

void launch_selftest_sort(const uint32_t* rows, uint32_t n, int row_bits, uint64_t* tmp, uint64_t* out, void* temp, uint32_t* head_pos,
                          uint32_t* nheads, hipStream_t s) {
    if (n == 0) return;
    const Scratch scr = carve(temp, n, row_bits);
    sort_and_list(SrcRows{rows}, n, row_bits, tmp, out, scr, head_pos, nheads, s);
}

}  // namespace sbr
