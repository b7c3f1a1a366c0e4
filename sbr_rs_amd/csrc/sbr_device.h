/* sbr_device.h — device-side helpers shared by the kernel translation units (sbr_kernels.hip, sbr_steps.hip): 16-byte loads and
 * stores, the group all-reduce that realises the contract's dot order, the optimiser element update, the packed-f32 forms of the
 * rational tanh, and the SmallTail of a one-sequence step (header + loss accumulators + lagged loss figure + key ordering).
 * Everything here is __device__ __forceinline__: no symbol leaves a translation unit. */
#ifndef SBR_DEVICE_H
#define SBR_DEVICE_H

#include <hip/hip_runtime.h>

#include "../../include/sbr_hip.h"
#include "sbr_kernels.h"
#include "sbr_numerics.h"

namespace sbr {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// All-reduce over the L lanes of a group in the contract's tree order: p += p[lane ^ off] for off = L/2 ... 1
// (the dot order of sbr_numerics.h).  Every step is a cross-lane move inside the VALU — v_permlane32_swap /
// v_permlane16_swap (gfx950) for off = 32 / 16, DPP row rotate / shifts / quad permutes below — instead of a
// ds_bpermute round trip through the LDS crossbar per step (five dependent ones per dot product made the
// score kernel issue-bound).  Float addition is commutative, so "mine + theirs" has the same bits in both lanes.
typedef unsigned v2u __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ float dpp_read(float p) {  // every lane has an in-row source for the controls used here
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(p), CTRL, 0xF, 0xF, true));
}
template <int L>
__device__ __forceinline__ float group_allreduce(float p) {
    if constexpr (L >= 64) {
        const v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);
        p = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    if constexpr (L >= 32) {
        const v2u r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(p), false, false);
        p = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    if constexpr (L >= 16) p = p + dpp_read<0x128>(p);  // row_ror:8 = lane ^ 8 inside a row of 16
    if constexpr (L >= 8) {  // lane ^ 4: banks 0, 2 of a row read 4 lanes up, banks 1, 3 read 4 lanes down
        unsigned t = __builtin_amdgcn_update_dpp(0u, __float_as_uint(p), 0x104, 0xF, 0x5, false);  // row_shl:4
        t = __builtin_amdgcn_update_dpp(t, __float_as_uint(p), 0x114, 0xF, 0xA, false);            // row_shr:4
        p = p + __uint_as_float(t);
    }
    if constexpr (L >= 4) p = p + dpp_read<0x4E>(p);  // quad_perm [2,3,0,1] = lane ^ 2
    if constexpr (L >= 2) p = p + dpp_read<0xB1>(p);  // quad_perm [1,0,3,2] = lane ^ 1
    return p;
}
// optimiser element update: Adagrad (acc = sum of squares) or Adam (acc = second moment, mom = first)
__device__ __forceinline__ void opt_update(const ModelView& m, float* w, float* acc, float* mom, float g) {
    if (m.optimizer == SBR_OPT_ADAM) sbr_adam(w, mom, acc, g, m.lr, m.l2, m.c1, m.c2);
    else sbr_adagrad(w, acc, g, m.lr, m.l2);
}

// A kernel-argument pointer passed through an empty asm: the compiler can no longer hoist "pointer + per-lane
// offset" out of the time loop as a 64-bit VGPR pair that lives across it (those pairs were being spilled to
// scratch, and a scratch reload is a vector-memory operation that waits for EVERYTHING outstanding).
template <class T>
__device__ __forceinline__ T* launder(T* p) {
    typedef T __attribute__((address_space(1))) * global_ptr;  // keep the address space: a generic pointer would
    global_ptr g = (global_ptr)p;                              // turn every access into a flat_ operation
    asm volatile("" : "+s"(g));
    return (T*)g;
}

__device__ __forceinline__ float dot4(float4 x, float4 y) {
    float p = x.x * y.x;
    p = sbr_fma(x.y, y.y, p);
    p = sbr_fma(x.z, y.z, p);
    p = sbr_fma(x.w, y.w, p);
    return p;
}

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_splat(float x) { return (v2f){x, x}; }
__device__ __forceinline__ void tanh_pq_x2(v2f x, v2f* p, v2f* q) {
    // sbr_tanh_pq's clamp (two comparison + select pairs; a NaN stays NaN) as one v_med3_f32 — which alone would turn a NaN
    // into -C — and one unordered comparison + select that puts the NaN back: the same bits for every input, fewer vector
    // instructions (which f32 MFMAs do not hide)
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

// workgroup-scope release / acquire around a barrier: the phases of a one-workgroup step run hand their results to each other
// through this CU's vector cache / L2 and through LDS
__device__ __forceinline__ void phase_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- the tail of a ONE-sequence step (SmallTail, sbr_kernels.h): one workgroup of 256 threads, after its score pass ----
// header + accumulators (block_header_kernel), the lagged loss figure of the one sequence (lagged_chain with B = 1: the node of
// its length is read, then takes the sequence's t-ascending loss sum), and the step's 3 R keys in (row, entry) order with the
// list of segment heads.  The keys are distinct (the entry number is their low word), so ranking every key among all of them
// IS the stable order by row that small_sort_kernel produces: integer work, identical output.
template <int NT>
__device__ __forceinline__ void small_tail(const MbView& mb, const BlockView& blk, const WorkView& w, const SmallTail& t, double lsum,
                                           unsigned int tsum) {
    constexpr int NMAX = 3 * SBR_SMALL_TAIL_MAX_ROWS, NW = NT / 64;
    __shared__ uint64_t ka[NMAX], kb[NMAX];
    __shared__ uint32_t s_cnt[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = mb.R;
    const uint32_t n = 3u * (uint32_t)R;
    /* the rows this workgroup's lanes have just written (ids, negatives, losses) are read by other lanes from here on */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (wave == 0) {
        const int ns = mb.steps[0];
        /* the lagged loss figure (sbr_report.hip): the node of this length is read, then nodes 0 .. ns-1 take the sequence's
         * running sums — lane l holds the sum after term base + l */
        const float x = t.lag_state[1 + 2 * (ns - 1)];
        float sum = 0.0f;
        for (int base = 0; base < ns; base += 64) {
            const int tt = base + lane;
            const float v = tt < ns ? w.loss[mb.off[tt]] : 0.0f;
            const int cnt = ns - base < 64 ? ns - base : 64;
            float mine = 0.0f;
            for (int l = 0; l < cnt; ++l) {
                sum = sum + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l));
                mine = lane == l ? sum : mine;
            }
            if (tt < ns) t.lag_state[1 + 2 * tt] = mine;
        }
        if (lane == 0) {
            t.header[0] = (uint32_t)R;
            t.header[1] = tsum;
            t.header[2] = t.header[3] = 0;
            *reinterpret_cast<double*>(t.header + 4) = lsum;
            *reinterpret_cast<unsigned long long*>(t.header + 6) = (unsigned long long)R;
            if (t.loss_acc) {
                t.loss_acc[0] += lsum;
                t.loss_acc[1] += lsum;
                t.ex_acc[0] += (unsigned long long)R;
                t.ex_acc[1] += tsum;
                t.ex_acc[2] += (unsigned long long)R;
            }
            t.lag_state[0] = t.lag_state[0] + x;
        }
    }
    for (uint32_t e = tid; e < n; e += NT) {
        const uint32_t r = e / 3u, kind = e - 3u * r;
        const uint32_t* a = kind == 0 ? blk.in_idx : (kind == 1 ? blk.out_idx : blk.neg);
        ka[e] = ((uint64_t)a[r] << 32) | e;
    }
    __syncthreads();
    for (uint32_t e = tid; e < n; e += NT) {
        const uint64_t k = ka[e];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += ka[j] < k ? 1u : 0u;
        kb[rank] = k;
    }
    __syncthreads();
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t base_heads = 0;
    for (uint32_t p0 = 0; p0 < n; p0 += NT) {  // workgroup-uniform trip count
        const uint32_t p = p0 + tid;
        const bool valid = p < n;
        const bool head = valid && (p == 0 || (uint32_t)(kb[p] >> 32) != (uint32_t)(kb[p - 1] >> 32));
        if (valid) t.keys_sorted[p] = kb[p];
        const uint64_t mm = __ballot(head);
        if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(mm);
        __syncthreads();
        uint32_t off = base_heads, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) {
            if (w2 < wave) off += s_cnt[w2];
            total += s_cnt[w2];
        }
        if (head) t.head_pos[off + (uint32_t)__popcll(mm & lt)] = p;
        base_heads += total;
        __syncthreads();
    }
    if (tid == 0) {
        *t.nheads = base_heads;
        t.head_pos[base_heads] = n;
    }
}

}  // namespace sbr
#endif
