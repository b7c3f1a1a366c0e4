// sbr_engine.hip — host side of libsbr_hip.so: the C-ABI of include/sbr_hip.h.
//
// Owns device memory (parameters resident in HBM for the life of the model), restates the
// reference's training driver (fit_sequence_model, /root/reference/src/models/sequence_model.rs:
// 70-178) as host-side index work (chunking, shuffles, partitioning, packing) and drives the
// gfx950 kernels of sbr_kernels.hip on one HIP stream.  There is no CPU compute fallback: without
// a HIP device every entry point returns SBR_ERR_NO_DEVICE.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/sbr_hip.h"
#include "sbr_kernels.h"
#include "sbr_numerics.h"

namespace {

#define HIPCHK(expr)                                                                  \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            std::fprintf(stderr, "[sbr_hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? SBR_ERR_OUT_OF_MEMORY                  \
                   : (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? SBR_ERR_NO_DEVICE : SBR_ERR_HIP; \
        }                                                                             \
    } while (0)

#define SBRCHK(expr)                     \
    do {                                 \
        sbr_status _s = (expr);          \
        if (_s != SBR_OK) return _s;     \
    } while (0)

/* Scratch of a fit call recycled across calls.  The reference's own bench re-fits one model in a loop (benches/benchmark.rs:40-42:
 * 10 000 interactions, three epochs per call): ~40 hipMalloc + 10 hipHostMalloc at sbr_fit_begin and as many frees at the end
 * cost 4-18 ms per call there — more than the call's kernels.  Blocks up to 64 MiB go back to this process-wide cache (768 MiB at
 * most, per kind) instead of the driver and are handed out again to requests of at most twice... the same size class; anything
 * larger takes the driver's path as before.  A block is cached only after the streams that used it were synchronised
 * (sbr_fit_plan_destroy, sbr_model_destroy), and nothing relies on fresh memory being zero.  sbr_release_cached_memory() empties it. */
struct ScratchCache {
    static constexpr size_t kMaxBlock = 64ull << 20, kMaxTotal = 768ull << 20;
    struct Blk { void* p; size_t bytes; int device; bool host; };
    std::mutex mu;
    std::vector<Blk> idle;
    std::unordered_map<void*, Blk> live;
    size_t cached[2] = {0, 0}; /* device, pinned host */
    static bool enabled() { return true; }
    hipError_t get(void** out, size_t bytes, bool host) {
        *out = nullptr;
        bytes = (bytes + 255) & ~(size_t)255;
        int device = 0;
        if (!host) (void)hipGetDevice(&device);
        if (enabled() && bytes <= kMaxBlock) {
            std::lock_guard<std::mutex> g(mu);
            size_t best = idle.size();
            for (size_t i = 0; i < idle.size(); ++i) {
                const Blk& b = idle[i];
                if (b.host != host || (!host && b.device != device) || b.bytes < bytes || b.bytes > 2 * bytes + 4096) continue;
                if (best == idle.size() || b.bytes < idle[best].bytes) best = i;
            }
            if (best != idle.size()) {
                const Blk b = idle[best];
                idle[best] = idle.back();
                idle.pop_back();
                cached[host] -= b.bytes;
                live[b.p] = b;
                *out = b.p;
                return hipSuccess;
            }
        }
        const hipError_t e = host ? hipHostMalloc(out, bytes, hipHostMallocDefault) : hipMalloc(out, bytes);
        if (e != hipSuccess) {
            bool have_cached;
            {
                std::lock_guard<std::mutex> g(mu);
                have_cached = cached[0] + cached[1] != 0;
            }
            if (e == hipErrorOutOfMemory && have_cached) { /* give the cache back and try once more */
                trim();
                return host ? hipHostMalloc(out, bytes, hipHostMallocDefault) : hipMalloc(out, bytes);
            }
            return e;
        }
        if (enabled() && bytes <= kMaxBlock) {
            std::lock_guard<std::mutex> g(mu);
            live[*out] = Blk{*out, bytes, device, host};
        }
        return hipSuccess;
    }
    void put(void* p, bool host) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = live.find(p);
            if (it != live.end()) {
                const Blk b = it->second;
                live.erase(it);
                if (b.host == host && cached[host] + b.bytes <= kMaxTotal) {
                    idle.push_back(b);
                    cached[host] += b.bytes;
                    return;
                }
            }
        }
        if (host) (void)hipHostFree(p); else (void)hipFree(p);
    }
    void trim() {
        std::vector<Blk> drop;
        {
            std::lock_guard<std::mutex> g(mu);
            drop.swap(idle);
            cached[0] = cached[1] = 0;
        }
        for (const Blk& b : drop) {
            if (b.host) (void)hipHostFree(b.p); else (void)hipFree(b.p);
        }
    }
};
ScratchCache& scratch_cache() {
    static ScratchCache* c = new ScratchCache; /* never destroyed: the HIP runtime may be gone before static destructors run */
    return *c;
}

template <typename T>
sbr_status dmalloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIPCHK(scratch_cache().get(reinterpret_cast<void**>(p), count * sizeof(T), false));
    return SBR_OK;
}
inline void dfree(void* p) { scratch_cache().put(p, false); }
template <typename T>
hipError_t hmalloc(T** p, size_t bytes) { return scratch_cache().get(reinterpret_cast<void**>(p), bytes, true); }
inline void hfree(void* p) { scratch_cache().put(p, true); }

int dim_ok(uint32_t d) { return d == 16 || d == 32 || d == 64 || d == 128 || d == 256; }

/* rows a replica's item-table arrays are allocated for: num_devices x S, S = ceil(num_items / num_devices) (one device: num_items) */
uint64_t table_rows_allocated(const sbr_hparams* hp) {
    const uint64_t n = hp->num_devices ? hp->num_devices : 1, I = hp->num_items;
    return n > 1 ? n * ((I + n - 1) / n) : I;
}

/* Storage width of an embedding_dim: the kernels exist for 16 / 32 / 64 / 128 / 256 columns; any other
 * embedding_dim <= 256 (the reference's builder takes any usize, lstm.rs:86-89) is stored in the next width up
 * with the extra columns — of the embeddings, of every weight row and column, of alpha — ZERO.  That is a fixed
 * point of the training step: a zero input/hidden unit with zero weights has z = 0, c = 0, h = 0 in the forward
 * pass, receives a zero gradient from the loss (its embedding columns are zero) and from the recurrence (its
 * weights are zero), so Adagrad/Adam leave it at zero — the padded model IS the embedding_dim-wide model, with
 * +0 terms in its sums.  0 = unsupported. */
int storage_dim(uint32_t e) {
    for (uint32_t p = 16; p <= 256; p *= 2)
        if (e >= 1 && e <= p) return (int)p;
    return 0;
}

/* ≙ wyrm nn::lstm::Parameters::new as recalled: four [(hidden+input) x hidden]
 * xavier_normal matrices (std 1/sqrt(rows)) drawn one after the other — forget, update gate, update
 * value, output gate — rows = hidden part first; a coupled layer draws all four and ignores the update
 * gate's.  Engine layout: W[2d][ng*d], rows [x ; h], column blocks i,f,g,o (coupled: f,g,o). */
void draw_lstm_weights(sbr_xorshift* rng, int dl, int d, int ng, std::vector<float>* w) { /* dl = embedding_dim, d = storage width */
    const double std_w = 1.0 / std::sqrt((double)(2 * dl));
    const int nz = ng * d;
    w->assign((size_t)2 * d * nz, 0.0f);
    for (int gate = 0; gate < 4; ++gate) {
        const int block = ng == 4 ? (gate == 0 ? 1 : gate == 1 ? 0 : gate) : (gate == 0 ? 0 : gate == 1 ? -1 : gate - 1);
        for (int row = 0; row < 2 * dl; ++row) {
            const int k = row < dl ? d + row : row - dl;
            for (int u = 0; u < dl; ++u) {
                const float v = sbr_rand_normal_f32(rng, 0.0, std_w);
                if (block >= 0) (*w)[(size_t)k * nz + block * d + u] = v;
            }
        }
    }
}

#ifndef SBR_PACK_THREADS
#define SBR_PACK_THREADS 8 /* host threads that fill a large minibatch's packed index arrays (build_epoch) */
#endif
struct TimingPair { hipEvent_t a, b; int family; uint64_t launches; };

}  // namespace

/* hipMemImportFromShareableHandle takes the POSIX file descriptor in its `void* osHandle` argument, and
 * the two HIP runtimes this library meets disagree on how: the runtime PyTorch bundles (ROCm 7.0) reads an
 * int THROUGH the pointer, the system runtime (ROCm 7.2) takes the descriptor VALUE cast to a pointer.
 * Passing a pointer is safe on both (7.2 rejects it with hipErrorInvalidValue), passing the value crashes
 * 7.0 — so: pointer first, value only after a clean rejection. */
static hipError_t import_shareable_fd(hipMemGenericAllocationHandle_t* handle, int fd) {
    int fd_copy = fd;
    hipError_t e = hipMemImportFromShareableHandle(handle, &fd_copy, hipMemHandleTypePosixFileDescriptor);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    return hipMemImportFromShareableHandle(handle, reinterpret_cast<void*>(static_cast<uintptr_t>(fd)), hipMemHandleTypePosixFileDescriptor);
}

/* A device buffer made of one HIP virtual-memory allocation, so that it can be handed to another process
 * as a file descriptor (hipMemExportToShareableHandle) and mapped there. */
struct VmmBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
    hipMemGenericAllocationHandle_t h{};
    bool have = false;

    static size_t granularity(int device) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        prop.requestedHandleTypes = hipMemHandleTypePosixFileDescriptor;
        size_t g = 0;
        if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return 0;
        return g;
    }
    sbr_status map_access(int device) {
        if (hipMemAddressReserve(&ptr, bytes, 0, nullptr, 0) != hipSuccess) { ptr = nullptr; return SBR_ERR_OUT_OF_MEMORY; }
        if (hipMemMap(ptr, bytes, 0, h, 0) != hipSuccess) return SBR_ERR_HIP;
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(ptr, bytes, &acc, 1) != hipSuccess) return SBR_ERR_UNSUPPORTED;
        return SBR_OK;
    }
    sbr_status alloc(size_t want, int device) {
        const size_t g = granularity(device);
        if (!g) return SBR_ERR_UNSUPPORTED;
        bytes = (want + g - 1) / g * g;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        prop.requestedHandleTypes = hipMemHandleTypePosixFileDescriptor;
        if (hipMemCreate(&h, bytes, &prop, 0) != hipSuccess) return SBR_ERR_OUT_OF_MEMORY;
        have = true;
        return map_access(device);
    }
    sbr_status export_fd(int* fd) const {
        if (!have) return SBR_ERR_INVALID_ARGUMENT;
        if (hipMemExportToShareableHandle(fd, h, hipMemHandleTypePosixFileDescriptor, 0) != hipSuccess) return SBR_ERR_HIP;
        return SBR_OK;
    }
    sbr_status import_fd(int fd, size_t nbytes, int device) {
        if (import_shareable_fd(&h, fd) != hipSuccess) return SBR_ERR_HIP;
        have = true;
        bytes = nbytes;
        return map_access(device);
    }
    void release() {
        if (ptr) { (void)hipMemUnmap(ptr, bytes); (void)hipMemAddressFree(ptr, bytes); ptr = nullptr; }
        if (have) { (void)hipMemRelease(h); have = false; }
    }
};

/* The item table of a partitioned model: ONE virtual address range per array (hipMemAddressReserve)
 * whose physical pages live on the owners' devices (hipMemCreate per run of pages, hipMemMap), readable
 * and writable from every device (hipMemSetAccess; remote pages travel over xGMI).  All replicas therefore
 * use the same table layout and the same global row ids — the kernels of the hot path are unchanged — while
 * each row is stored once and updated only by its owner.  Two deployments: a single-process group creates
 * every part itself (local_rank < 0); under one process per GPU each process creates the parts homed on its
 * rank and maps the others from file descriptors its peers exported. */
struct SharedTable {
    struct Part { hipMemGenericAllocationHandle_t h{}; size_t off = 0, bytes = 0; int home = 0; bool have = false, mapped = false; };
    void* base = nullptr;
    size_t total = 0;
    std::vector<Part> parts;
    float *E = nullptr, *Eacc = nullptr, *Em = nullptr, *b = nullptr, *bacc = nullptr, *bm = nullptr;
    int local_rank = -1;
    uint64_t slice = 0;

    ~SharedTable() {
        for (auto& pt : parts) {
            if (pt.mapped) (void)hipMemUnmap(reinterpret_cast<char*>(base) + pt.off, pt.bytes);
            if (pt.have) (void)hipMemRelease(pt.h);
        }
        if (base) (void)hipMemAddressFree(base, total);
    }

    /* layout: identical on every rank.  A page goes to the rank that owns its first row (rows [r*S, (r+1)*S)
     * belong to rank r); runs of pages with the same home are one part. */
    sbr_status plan(uint64_t num_items, uint64_t d, bool adam, int nranks, uint64_t S, int device) {
        slice = S;
        const size_t gran = VmmBuf::granularity(device);
        if (!gran) return SBR_ERR_UNSUPPORTED;
        struct Arr { float** ptr; uint64_t row_bytes; bool on; };
        Arr arrs[6] = {{&E, d * 4, true}, {&Eacc, d * 4, true}, {&Em, d * 4, adam}, {&b, 4, true}, {&bacc, 4, true}, {&bm, 4, adam}};
        size_t off[6], bytes[6];
        total = 0;
        for (int a = 0; a < 6; ++a) {
            off[a] = total;
            bytes[a] = arrs[a].on ? ((size_t)(num_items * arrs[a].row_bytes) + gran - 1) / gran * gran : 0;
            total += bytes[a];
        }
        if (hipMemAddressReserve(&base, total, gran, nullptr, 0) != hipSuccess) { base = nullptr; return SBR_ERR_OUT_OF_MEMORY; }
        for (int a = 0; a < 6; ++a) {
            if (!arrs[a].on) continue;
            *arrs[a].ptr = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off[a]);
            const size_t npages = bytes[a] / gran;
            auto home = [&](size_t page) {
                const uint64_t row = (uint64_t)(page * gran) / arrs[a].row_bytes;
                uint64_t owner = row / S;
                if (owner >= (uint64_t)nranks) owner = (uint64_t)nranks - 1;
                return (int)owner;
            };
            size_t run_begin = 0;
            while (run_begin < npages) {
                const int hm = home(run_begin);
                size_t run_end = run_begin + 1;
                while (run_end < npages && home(run_end) == hm) ++run_end;
                Part pt;
                pt.off = off[a] + run_begin * gran;
                pt.bytes = (run_end - run_begin) * gran;
                pt.home = hm;
                parts.push_back(pt);
                run_begin = run_end;
            }
        }
        return SBR_OK;
    }

    /* physical allocation + mapping of the parts this process is responsible for */
    sbr_status create_parts(const std::vector<int>& device_of_rank) {
        for (auto& pt : parts) {
            if (local_rank >= 0 && pt.home != local_rank) continue;
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = device_of_rank[pt.home];
            prop.requestedHandleTypes = hipMemHandleTypePosixFileDescriptor;
            if (hipMemCreate(&pt.h, pt.bytes, &prop, 0) != hipSuccess) return SBR_ERR_OUT_OF_MEMORY;
            pt.have = true;
            if (hipMemMap(reinterpret_cast<char*>(base) + pt.off, pt.bytes, 0, pt.h, 0) != hipSuccess) return SBR_ERR_HIP;
            pt.mapped = true;
        }
        return SBR_OK;
    }

    sbr_status import_part(uint32_t i, int fd) {
        if (i >= parts.size() || parts[i].mapped) return SBR_ERR_INVALID_ARGUMENT;
        Part& pt = parts[i];
        if (import_shareable_fd(&pt.h, fd) != hipSuccess) return SBR_ERR_HIP;
        pt.have = true;
        if (hipMemMap(reinterpret_cast<char*>(base) + pt.off, pt.bytes, 0, pt.h, 0) != hipSuccess) return SBR_ERR_HIP;
        pt.mapped = true;
        return SBR_OK;
    }

    sbr_status set_access(const std::vector<int>& devices) {
        for (const auto& pt : parts)
            if (!pt.mapped) return SBR_ERR_INVALID_ARGUMENT; /* a peer's part has not been imported yet */
        std::vector<int> uniq(devices);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        std::vector<hipMemAccessDesc> acc(uniq.size());
        for (size_t i = 0; i < uniq.size(); ++i) {
            acc[i].location.type = hipMemLocationTypeDevice;
            acc[i].location.id = uniq[i];
            acc[i].flags = hipMemAccessFlagsProtReadWrite;
        }
        if (hipMemSetAccess(base, total, acc.data(), acc.size()) != hipSuccess) return SBR_ERR_UNSUPPORTED;
        return SBR_OK;
    }

    /* single-process group: devices[r] = HIP device of replica r */
    sbr_status create(uint64_t num_items, uint64_t d, bool adam, const std::vector<int>& devices, uint64_t S) {
        SBRCHK(plan(num_items, d, adam, (int)devices.size(), S, devices[0]));
        SBRCHK(create_parts(devices));
        return set_access(devices);
    }
};

/* Scratch of the prediction side (user_representation / mrr_score), kept for the life of the model: one device allocation that
 * only grows, carved per call — a call used to make and free ~25 device allocations (a quarter of mrr_score's wall time at
 * 8 192 users x 1e6 items went to the allocator and its implicit synchronisations). */
struct DeviceArena {
    uint8_t* base = nullptr;
    size_t cap = 0, used = 0;
    sbr_status reserve(size_t need) {
        used = 0;
        if (need <= cap) return SBR_OK;
        if (base) (void)hipFree(base);
        base = nullptr; cap = 0;
        const size_t want = need + need / 8 + 4096;
        if (hipMalloc(reinterpret_cast<void**>(&base), want) != hipSuccess) { base = nullptr; return SBR_ERR_OUT_OF_MEMORY; }
        cap = want;
        return SBR_OK;
    }
    static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    template <typename T>
    T* take(size_t count) {
        T* p = reinterpret_cast<T*>(base + used);
        used += padded((count ? count : 1) * sizeof(T));
        return p;
    }
    void release() { if (base) (void)hipFree(base); base = nullptr; cap = used = 0; }
};

struct sbr_model {
    sbr_hparams hp;
    std::shared_ptr<SharedTable> shared; /* partitioned group: the table arrays belong to this object */
    int d = 0, ng = 0; /* d = storage width (storage_dim of hp.embedding_dim) */
    int dl = 0;        /* embedding_dim as the caller sees it */
    int device = 0;
    sbr::ModelView mv;
    sbr_xorshift rng;
    sbr_xorshift rng_after_table; /* RNG state right after the item-table initialisation */
    std::vector<float> pending_E;  /* process-per-GPU partitioned table: this rank's rows until sbr_partition_finalize */
    bool partition_finalized = true;
    uint64_t global_epoch = 0;
    uint64_t opt_steps = 0; /* optimiser steps taken (Adam bias correction) */
    float last_lagged_loss = 0.0f; /* what the reference's fit would have returned for the last sbr_model_fit / sbr_group_fit */
    bool opt_state_partial = false; /* owner-applied steps (sbr_fit_step_owner_update) have run since the item table's optimiser state was last
                                     * complete on this replica: only the rows this rank owns are current (sbr_model_table_slice /
                                     * sbr_model_optimizer_state_gathered; the library's own drivers gather when a fit ends) */
    bool reference_order = false; /* sbr_model_set_reference_order: negatives from the worker's sequential stream (one sequence per step) */
    int step_fusion = 2; /* one-sequence steps at d <= 32 (sbr_model_set_step_fusion): 0 separate launches, 1 fused launches
                          * (SmallTail + small_back: four per step), 2 runs of steps in one launch where the shape allows */
    DeviceArena eval_arena;        /* prediction-side scratch (guarded by mu) */
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t side = nullptr;          /* second stream: dense-gradient GEMM runs beside the sparse update */
    hipStream_t sorter = nullptr;        /* third stream: key sort of the sparse update, underneath the backward pass */
    hipStream_t copier = nullptr;        /* uploads of the packed epochs; created at the first fit, kept (creating and destroying a
                                          * stream costs 1.5 + 1.3 ms — as much as a whole small fit call's kernels) */
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_scored = nullptr, ev_sorted = nullptr;
    std::mutex mu;
    bool timing = false;
    uint32_t timing_mask = 0xffffffffu; /* kernel families whose launches are bracketed by events while `timing` is on */
    bool overlap = true; /* false: the side-stream work is queued on the main stream (standalone kernel timing) */
    std::vector<TimingPair> pending;
    double ms[SBR_K_FAMILIES] = {0};
    uint64_t launches[SBR_K_FAMILIES] = {0};
};

namespace {

struct ScopedTimer {
    sbr_model* m;
    TimingPair tp;
    bool on;
    hipStream_t st;
    ScopedTimer(sbr_model* model, int family, uint64_t launches, hipStream_t stream = nullptr)
        : m(model), on(model->timing && ((model->timing_mask >> family) & 1u)), st(stream ? stream : model->stream) {
        if (!on) return;
        tp.family = family;
        tp.launches = launches;
        hipEventCreate(&tp.a);
        hipEventCreate(&tp.b);
        hipEventRecord(tp.a, st);
    }
    ~ScopedTimer() {
        if (!on) return;
        hipEventRecord(tp.b, st);
        m->pending.push_back(tp);
    }
};

uint64_t dense_count(const sbr_model* m) {
    return m->ng ? (uint64_t)(2 * m->d + 1) * m->ng * m->d : (uint64_t)m->d;
}

/* exchange block geometry (identical to the oracle's) */
uint64_t block_bytes_for(const sbr_model* m, uint64_t rmax) {
    const uint64_t words = 8 + 4 * rmax + 2 * rmax * (uint64_t)m->d + dense_count(m);
    return ((words * 4 + 15) / 16) * 16;
}
uint64_t dense_offset_bytes(const sbr_model* m, uint64_t rmax) { return (8 + 4 * rmax + 2 * rmax * (uint64_t)m->d) * 4; }

sbr::BlockView block_view(const sbr_model* m, void* block, uint64_t rmax) {
    sbr::BlockView v;
    uint32_t* w = reinterpret_cast<uint32_t*>(block);
    v.header = w;
    v.in_idx = w + 8;
    v.out_idx = v.in_idx + rmax;
    v.neg = v.out_idx + rmax;
    v.coef = reinterpret_cast<float*>(v.neg + rmax);
    v.H = reinterpret_cast<float*>(w + 8 + 4 * rmax);
    v.dX = v.H + rmax * (uint64_t)m->d;
    v.dense = v.dX + rmax * (uint64_t)m->d;
    return v;
}

/* host-side packed description of a set of sequences (training minibatch or evaluation batch) */
struct Packed {
    int R = 0, B = 0, Tm = 0;
    std::vector<int> off, steps, prev_row, order; /* order[b] = index of the b-th sequence in the input */
    std::vector<uint32_t> in_idx, out_idx, ctr;
};

/* Sequences are given as (pointer to first item, number of steps, has_target).  Training:
 * steps = len - 1, in_t = item[t], out_t = item[t+1] (sequence_model.rs:115-122).  Evaluation:
 * steps = len, no targets (sequence_model.rs:192-194).  Ordered by steps descending (stable). */
void pack_sequences(const std::vector<const uint32_t*>& first, const std::vector<int>& nsteps, bool with_targets,
                    const std::vector<uint64_t>* ctr_base, int maxT, Packed* out) {
    const int nb = (int)first.size();
    out->B = nb;
    out->order.resize(nb);
    std::vector<int> cnt(maxT + 2, 0), pos(maxT + 2, 0);
    for (int i = 0; i < nb; ++i) cnt[nsteps[i]]++;
    int acc = 0;
    for (int l = maxT; l >= 0; --l) { pos[l] = acc; acc += cnt[l]; }
    for (int i = 0; i < nb; ++i) out->order[pos[nsteps[i]]++] = i;
    const int Tm = nb ? nsteps[out->order[0]] : 0;
    out->Tm = Tm;
    out->off.assign(Tm + 1, 0);
    out->steps.resize(nb);
    for (int b = 0; b < nb; ++b) out->steps[b] = nsteps[out->order[b]];
    {
        int alive = nb;
        for (int t = 0; t < Tm; ++t) {
            while (alive > 0 && out->steps[alive - 1] <= t) --alive;
            out->off[t + 1] = out->off[t] + alive;
        }
    }
    const int R = out->off[Tm];
    out->R = R;
    out->in_idx.assign(R, 0);
    out->out_idx.assign(R, 0);
    out->ctr.assign(R, 0);
    out->prev_row.assign(R, -1);
    for (int b = 0; b < nb; ++b) {
        const int src = out->order[b];
        const uint32_t* it = first[src];
        for (int t = 0; t < out->steps[b]; ++t) {
            const int r = out->off[t] + b;
            out->in_idx[r] = it[t];
            if (with_targets) out->out_idx[r] = it[t + 1];
            if (ctr_base) out->ctr[r] = (uint32_t)((*ctr_base)[src] + (uint64_t)t);
            out->prev_row[r] = t ? out->off[t - 1] + b : -1;
        }
    }
}

struct DevicePacked { /* device image of one or more Packed, concatenated */
    int* off = nullptr;
    int* steps = nullptr;
    int* prev_row = nullptr;
    uint32_t *in_idx = nullptr, *out_idx = nullptr, *ctr = nullptr;
    void release() {
        dfree(off); dfree(steps); dfree(prev_row); dfree(in_idx); dfree(out_idx); dfree(ctr);
        off = steps = prev_row = nullptr;
        in_idx = out_idx = ctr = nullptr;
    }
};

struct WorkBuffers {
    sbr::WorkView v{};
    void release() {
        dfree(v.C); dfree(v.G); dfree(v.dH); dfree(v.dZ); dfree(v.X); dfree(v.zeros); dfree(v.dHrec); dfree(v.dCrec); dfree(v.dab);
        dfree(v.partials); dfree(v.loss); dfree(v.tries); dfree(v.part_loss); dfree(v.part_tries);
        v = sbr::WorkView{};
    }
};

sbr_status alloc_work(const sbr_model* m, uint64_t rmax, uint64_t bmax, bool training, WorkBuffers* wb) {
    const uint64_t d = (uint64_t)m->d;
    sbr::WorkView& v = wb->v;
    v.fold_max_tiles = SBR_FOLD_MAX_TILES_DEFAULT;
    if (m->ng) {
        SBRCHK(dmalloc(&v.C, rmax * d));
        SBRCHK(dmalloc(&v.G, rmax * d * 4));
        SBRCHK(dmalloc(&v.X, rmax * d));
    }
    if (training) {
        SBRCHK(dmalloc(&v.dH, rmax * d));
        SBRCHK(dmalloc(&v.loss, rmax));
        SBRCHK(dmalloc(&v.tries, rmax));
        SBRCHK(dmalloc(&v.part_loss, 2048));
        SBRCHK(dmalloc(&v.part_tries, 2048));
        if (m->ng) {
            const uint64_t rchunk = (rmax + SBR_DW_CHUNK_ROWS - 1) / SBR_DW_CHUNK_ROWS * SBR_DW_CHUNK_ROWS;
            SBRCHK(dmalloc(&v.dZ, rchunk * d * (uint64_t)m->ng)); /* the dense-gradient GEMM reads dZ to the end of the last chunk */
            v.wide_addresses = std::getenv("SBR_DW_WIDE_ADDRESSES") ? 1 : 0; /* read when the plan's buffers are made */
            SBRCHK(dmalloc(&v.zeros, 256));
            HIPCHK(hipMemset(v.zeros, 0, 256 * sizeof(float)));
            SBRCHK(dmalloc(&v.dHrec, bmax * d));
            SBRCHK(dmalloc(&v.dCrec, bmax * d));
            const uint64_t nch = (rmax + SBR_DW_CHUNK_ROWS - 1) / SBR_DW_CHUNK_ROWS;
            SBRCHK(dmalloc(&v.partials, nch * dense_count(m)));
        } else {
            SBRCHK(dmalloc(&v.dab, bmax * d));
            SBRCHK(dmalloc(&v.partials, ((bmax + 255) / 256) * d));
        }
    }
    return SBR_OK;
}

}  // namespace

struct sbr_fit_plan {
    sbr_model* m = nullptr;
    int ndev = 1, rank = 0, T = 0;
    uint64_t nseq_total = 0, part_len = 0;
    std::vector<uint64_t> seq_start; /* [ndev * part_len], offsets into items */
    std::vector<uint32_t> seq_len;
    std::vector<sbr_xorshift> part_rng;
    std::vector<uint64_t> fit_seed;
    std::vector<uint32_t> items; /* host copy of the CSR item ids */
    uint64_t rmax = 0, bmax = 0;
    /* epoch data is double-buffered: the GPU consumes ep[cur] while a host thread may already
     * shuffle, pack and upload the next epoch into ep[cur ^ 1] (sbr_fit_epoch_prefetch) */
    struct Mb { int R, B, Tm; uint64_t row_base, off_base, seq_base; };
    struct Epoch {
        uint64_t epoch_key_epoch = 0;
        uint64_t num_mb = 0;
        std::vector<Mb> mbs;
        std::vector<int> off_host;         /* concatenated off tables of this rank */
        std::vector<uint32_t> rows_of_dev; /* [num_mb][ndev] */
        DevicePacked dp;
        sbr::StepDesc* d_desc = nullptr;   /* one sequence per step (batch_sequences = 1): the steps of the epoch for the one-launch form */
        uint64_t desc_cap = 0;
        std::vector<sbr::StepDesc> desc_host;
        uint64_t rows_cap = 0, off_cap = 0, seq_cap = 0;
        /* pinned host staging of the packed index arrays (written directly by the packer, one DMA each) */
        uint32_t *h_in = nullptr, *h_out = nullptr, *h_ctr = nullptr;
        int *h_prev = nullptr, *h_steps = nullptr;
        uint64_t h_rows_cap = 0, h_seq_cap = 0;
        hipEvent_t free_event = nullptr;   /* recorded on the compute stream when the GPU is done with dp */
        bool free_recorded = false;
    } ep[2];
    int cur = 0;
    std::thread worker;
    bool pending = false;
    sbr_status pending_status = SBR_OK;
    hipStream_t copy_stream = nullptr;
    /* work */
    WorkBuffers wb;
    uint8_t* block = nullptr;
    uint64_t block_bytes = 0;
    uint64_t *keys = nullptr, *keys_sorted = nullptr;
    void* sort_temp = nullptr;
    size_t sort_temp_bytes = 0;
    int key_bits = 64;
    double* loss_acc = nullptr;
    unsigned long long* ex_acc = nullptr;
    uint32_t* ref_rng = nullptr;   /* reference order: the worker's xorshift128 state on the device (advanced by the score launch) */
    bool ref_rng_live = false;     /* ... and it has been advanced since the host last held it */
    unsigned long long* phase_clocks = nullptr; /* [6] epoch_steps_kernel's per-phase s_memtime ticks + steps (sbr_fit_debug_phase_clocks) */
    sbr::SegScratch seg{}; /* long-segment path of the sparse reduction (hot rows) */
    bool dense_pending = false; /* the side stream still owes blk.dense */
    bool hot_prelisted = false; /* this step's long segments (hot rows) were listed behind the ordering: their chunks are reduced on the
                                 * ordering's stream beside the short-segment pass (sbr_fit_step_apply) */
    hipEvent_t ev_hot = nullptr;
    bool fuse_back = false;     /* sbr_fit_step: the step's optimiser half may take the single-launch form (sbr::launch_small_back) */
    bool dw_deferred = false;   /* ... and step_local left the dense gradient to it */
    int dense_unreduced_chunks = 0; /* > 0: blk.dense is still that many chunk partials in wb.v.partials (one device: reduced by its consumer) */
    bool sort_off_stream = false; /* the step's key ordering ran on another stream than the main one: ev_sorted joins it */
    bool sorted_event_live = false; /* ev_sorted has been recorded at least once: the multi-device consumers wait on it whatever the
                                     * last step's placement was (a completed event costs nothing; the flag above belongs to ONE step) */
    bool header_accumulated = false; /* single device: block_header_kernel already added this step to loss_acc / ex_acc */
    /* the loss figure the reference's fit returns (sbr_report.hip): [accumulator | loss-node value per sequence length], the
     * per-sequence sums of the current step, and the events that order the one-wave chain kernel (sorter stream) against the
     * main stream */
    float *lag_state = nullptr, *lag_seqsum = nullptr;
    hipEvent_t ev_seqsum = nullptr, ev_lagged = nullptr;
    bool lag_busy = false; /* ev_lagged is pending on another stream than the main one */
    /* partitioned item table: this device's gradient list (addressed by sorted-key position), the owner
     * bounds, and the owner-side merge buffers */
    float *glist = nullptr, *gblist = nullptr;
    uint32_t *gfl = nullptr, *bounds_dev = nullptr;
    /* one process per GPU: the list (and the sorted keys) live in exportable allocations, the peers' lists
     * are mapped from the file descriptors they exported: [0] keys_sorted [1] glist [2] gblist [3] gfl */
    /* peer transport of the replicated exchange: [0] this rank's send buffer (ndev chunks), [1] its reduced
     * own chunk — exportable; the peers' pair is mapped here and read in place by the kernels */
    VmmBuf xchg_own[2];
    std::vector<std::array<VmmBuf, 2>> xchg_peer;
    bool exportable_lists = false;
    VmmBuf own_x[4];
    std::vector<std::array<VmmBuf, 4>> peer_x;
    uint64_t *mkeys = nullptr, *mkeys_sorted = nullptr;
    void* msort_temp = nullptr;
    size_t msort_temp_bytes = 0;
    uint64_t mcap = 0;
    sbr::MergePlan* mplan = nullptr;  /* where this owner's row range lies in every device's sorted keys: on the DEVICE (merge_plan_kernel) */
    uint32_t* all_bounds_dev = nullptr; /* one process per GPU: the ranks' owner bounds, gathered on the device by the host's collective */
    /* last step (debug) */
    int last_R = 0;
    const void* last_block = nullptr;
};

namespace {

sbr::MbView mb_view(const sbr_fit_plan* p, uint64_t i) {
    const sbr_fit_plan::Epoch& e = p->ep[p->cur];
    const sbr_fit_plan::Mb& mb = e.mbs[i];
    sbr::MbView v;
    v.R = mb.R; v.B = mb.B; v.Tm = mb.Tm;
    v.off = e.dp.off + mb.off_base;
    v.steps = e.dp.steps + mb.seq_base;
    v.prev_row = e.dp.prev_row + mb.row_base;
    v.in_idx = e.dp.in_idx + mb.row_base;
    v.out_idx = e.dp.out_idx + mb.row_base;
    v.ctr = e.dp.ctr + mb.row_base;
    return v;
}

void shuffle_pairs(uint64_t* start, uint32_t* len, uint64_t n, sbr_xorshift* r) {
    for (uint64_t i = n; i >= 2;) { /* rand 0.5 Rng::shuffle: Fisher-Yates from the end over gen_range(0, i + 1) */
        i -= 1;
        const uint64_t j = sbr_rand_gen_range(r, i + 1);
        std::swap(start[i], start[j]);
        std::swap(len[i], len[j]);
    }
}

/* one optimiser step begins: advance the step count and refresh Adam's bias corrections */
void begin_optimizer_step(sbr_model* m) {
    m->opt_steps += 1;
    if (m->hp.optimizer == SBR_OPT_ADAM) sbr_adam_corrections(m->opt_steps, &m->mv.c1, &m->mv.c2);
}

sbr_status ensure_device(const sbr_model* m) {
    HIPCHK(hipSetDevice(m->device));
    return SBR_OK;
}

}  // namespace

extern "C" {

uint32_t sbr_abi_version(void) { return 10; }

void sbr_release_cached_memory(void) { scratch_cache().trim(); }

const char* sbr_status_string(sbr_status s) {
    switch (s) {
        case SBR_OK: return "ok";
        case SBR_ERR_NO_INTERACTIONS: return "No interactions were supplied.";
        case SBR_ERR_INVALID_PREDICTION: return "Invalid prediction value: non-finite or not a number.";
        case SBR_ERR_INVALID_ARGUMENT: return "invalid argument";
        case SBR_ERR_UNSUPPORTED: return "unsupported configuration";
        case SBR_ERR_NO_DEVICE: return "no HIP device (the engine has no CPU fallback)";
        case SBR_ERR_HIP: return "HIP runtime error";
        case SBR_ERR_OUT_OF_MEMORY: return "out of device memory";
    }
    return "unknown";
}

sbr_status sbr_device_info(char* device_name, uint64_t name_bytes, uint32_t* out_cus, uint64_t* out_hbm_bytes) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return SBR_ERR_NO_DEVICE;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    if (device_name && name_bytes) {
        std::snprintf(device_name, (size_t)name_bytes, "%s", prop.gcnArchName);
    }
    if (out_cus) *out_cus = (uint32_t)prop.multiProcessorCount;
    if (out_hbm_bytes) *out_hbm_bytes = (uint64_t)prop.totalGlobalMem;
    return SBR_OK;
}

/* shared != null: the table arrays are the group's SharedTable; only the replica with write_table
 * initialises them, the others take the RNG state that follows the table initialisation */
enum TableInit { TABLE_WRITE_ALL = 0, TABLE_WRITE_NONE = 1, TABLE_OWN_ROWS_DEFERRED = 2 };
static sbr_status model_create_impl(const sbr_hparams* hp, std::shared_ptr<SharedTable> shared, int table_init,
                                    const sbr_xorshift* rng_after_table, sbr_model** out) {
    if (!hp || !out) return SBR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!storage_dim(hp->embedding_dim) || hp->num_items == 0 || hp->max_sequence_length < 3 || hp->num_devices == 0 ||
        hp->num_devices > 16 || hp->device_rank >= hp->num_devices || hp->batch_sequences == 0 ||
        hp->model < 0 || hp->model > 2 || hp->loss < 0 || hp->loss > 2)
        return SBR_ERR_INVALID_ARGUMENT;
    if (hp->optimizer != SBR_OPT_ADAGRAD && hp->optimizer != SBR_OPT_ADAM) return SBR_ERR_INVALID_ARGUMENT;
    int ndevices = 0;
    if (hipGetDeviceCount(&ndevices) != hipSuccess || ndevices == 0) return SBR_ERR_NO_DEVICE;
    sbr_model* m = new (std::nothrow) sbr_model();
    if (!m) return SBR_ERR_OUT_OF_MEMORY;
    m->hp = *hp;
    m->dl = (int)hp->embedding_dim;
    m->d = storage_dim(hp->embedding_dim);
    m->ng = hp->model == SBR_MODEL_LSTM_NORMAL ? 4 : hp->model == SBR_MODEL_LSTM_COUPLED ? 3 : 0;
    if (hipGetDevice(&m->device) != hipSuccess) { delete m; return SBR_ERR_NO_DEVICE; }
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { delete m; return SBR_ERR_HIP; }
    m->own_stream = true;
    /* the side stream carries the MFMA-bound dense-gradient GEMM beside the HBM-bound sparse update:
     * lowest priority, so the update's workgroups are placed first whenever a slot frees up */
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    /* the key sort gets its own stream: on the GEMM's stream its tail — the passes that did not fit underneath the
     * backward pass, whose workgroups fill every CU, so that the sort's kernels only run in slots that retiring
     * workgroups free — delayed the GEMM, which does not need the sorted keys, by 0.36 ms per step.  Priority: normal
     * (13.42-13.50 ms per step; high 13.62-13.74: the sort then wins every freed slot and the backward pass loses more
     * than the update gains; low 13.58-13.63). */
    (void)prio_greatest;
    if (hipStreamCreateWithPriority(&m->sorter, hipStreamNonBlocking, 0) != hipSuccess) { delete m; return SBR_ERR_HIP; }
    if (hipStreamCreateWithPriority(&m->side, hipStreamNonBlocking, prio_least) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_scored, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_sorted, hipEventDisableTiming) != hipSuccess) { delete m; return SBR_ERR_HIP; }
    sbr::ModelView& v = m->mv;
    std::memset(&v, 0, sizeof(v));
    v.d = m->d; v.ng = m->ng; v.coupled = m->ng == 3;
    v.num_items = hp->num_items; v.loss = hp->loss; v.lr = hp->learning_rate; v.l2 = hp->l2_penalty;
    v.optimizer = hp->optimizer; v.c1 = 1.0f; v.c2 = 1.0f;
    const bool adam = hp->optimizer == SBR_OPT_ADAM;
    const uint64_t I = hp->num_items, d = (uint64_t)m->d;
    /* a replica of a multi-device model stores num_devices x ceil(I / num_devices) rows: the owners' slices are then equally long, so
     * the all-gather of the updated parameter slices (sbr_fit_step_owner_update) can write every replica's table IN PLACE */
    const uint64_t Ia = table_rows_allocated(hp);
    sbr_status st = SBR_OK;
    auto fail = [&](sbr_status s) { sbr_model_destroy(m); return s; };
    m->shared = shared;
    if (shared) {
        v.E = shared->E; v.Eacc = shared->Eacc; v.b = shared->b; v.bacc = shared->bacc;
        v.Em = shared->Em; v.bm = shared->bm;
    } else {
        if ((st = dmalloc(&v.E, Ia * d)) != SBR_OK) return fail(st);
        if ((st = dmalloc(&v.Eacc, Ia * d)) != SBR_OK) return fail(st);
        if ((st = dmalloc(&v.b, Ia)) != SBR_OK) return fail(st);
        if ((st = dmalloc(&v.bacc, Ia)) != SBR_OK) return fail(st);
        if (adam) {
            if ((st = dmalloc(&v.Em, Ia * d)) != SBR_OK) return fail(st);
            if ((st = dmalloc(&v.bm, Ia)) != SBR_OK) return fail(st);
        }
        if (Ia > I) { /* the padding rows: never addressed by a row id, but they travel with the last owner's slice */
            hipMemsetAsync(v.E + I * d, 0, (Ia - I) * d * 4, m->stream);
            hipMemsetAsync(v.Eacc + I * d, 0, (Ia - I) * d * 4, m->stream);
            hipMemsetAsync(v.b + I, 0, (Ia - I) * 4, m->stream);
            hipMemsetAsync(v.bacc + I, 0, (Ia - I) * 4, m->stream);
            if (adam) {
                hipMemsetAsync(v.Em + I * d, 0, (Ia - I) * d * 4, m->stream);
                hipMemsetAsync(v.bm + I, 0, (Ia - I) * 4, m->stream);
            }
        }
    }
    /* ≙ build_params (lstm.rs:174-194): embeddings first, then the recurrent weights, same RNG */
    sbr_xs_seed(&m->rng, hp->seed);
    if (!shared || table_init == TABLE_WRITE_ALL) {
        if (adam) {
            hipMemsetAsync(v.Em, 0, I * d * 4, m->stream);
            hipMemsetAsync(v.bm, 0, I * 4, m->stream);
        }
        hipMemsetAsync(v.Eacc, 0, I * d * 4, m->stream);
        hipMemsetAsync(v.b, 0, I * 4, m->stream);
        hipMemsetAsync(v.bacc, 0, I * 4, m->stream);
        std::vector<float> host(I * d, 0.0f);
        const uint64_t dl = (uint64_t)m->dl;
        const double std_e = 1.0 / (double)dl; /* embedding_init, lstm.rs:22-25: row-major, embedding_dim values per row */
        for (uint64_t row = 0; row < I; ++row)
            for (uint64_t c = 0; c < dl; ++c) host[row * d + c] = sbr_rand_normal_f32(&m->rng, 0.0, std_e);
        if (hipMemcpy(v.E, host.data(), host.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail(SBR_ERR_HIP);
    } else if (table_init == TABLE_OWN_ROWS_DEFERRED) {
        /* every rank walks the whole initialisation stream (the RNG has no skip-ahead through the normal
         * sampler's rejections) and keeps the rows it owns; they are written once the peers' parts are mapped */
        const uint64_t S = shared->slice, r0 = std::min<uint64_t>(I, (uint64_t)hp->device_rank * S), r1 = std::min<uint64_t>(I, r0 + S);
        m->pending_E.assign((size_t)((r1 - r0) * d), 0.0f);
        const uint64_t dl = (uint64_t)m->dl;
        const double std_e = 1.0 / (double)dl;
        for (uint64_t i = 0; i < I * dl; ++i) {
            const float val = sbr_rand_normal_f32(&m->rng, 0.0, std_e);
            const uint64_t row = i / dl;
            if (row >= r0 && row < r1) m->pending_E[(size_t)((row - r0) * d + i % dl)] = val;
        }
        m->partition_finalized = false;
    } else {
        m->rng = *rng_after_table;
    }
    m->rng_after_table = m->rng;
    if (m->ng) {
        const uint64_t nw = 2 * d * (uint64_t)m->ng * d, nb = (uint64_t)m->ng * d;
        if ((st = dmalloc(&v.W, nw)) != SBR_OK) return fail(st);
        if ((st = dmalloc(&v.Wacc, nw)) != SBR_OK) return fail(st);
        if ((st = dmalloc(&v.Wp, nw)) != SBR_OK) return fail(st);
        if ((st = dmalloc(&v.WTp, nw)) != SBR_OK) return fail(st);
        if ((st = dmalloc(&v.bW, nb)) != SBR_OK) return fail(st);
        if ((st = dmalloc(&v.bWacc, nb)) != SBR_OK) return fail(st);
        if (adam) {
            if ((st = dmalloc(&v.Wm, nw)) != SBR_OK) return fail(st);
            if ((st = dmalloc(&v.bWm, nb)) != SBR_OK) return fail(st);
            hipMemsetAsync(v.Wm, 0, nw * 4, m->stream);
            hipMemsetAsync(v.bWm, 0, nb * 4, m->stream);
        }
        hipMemsetAsync(v.Wacc, 0, nw * 4, m->stream);
        hipMemsetAsync(v.bW, 0, nb * 4, m->stream);
        hipMemsetAsync(v.bWacc, 0, nb * 4, m->stream);
        std::vector<float> host;
        draw_lstm_weights(&m->rng, m->dl, m->d, m->ng, &host);
        if (hipMemcpy(v.W, host.data(), nw * 4, hipMemcpyHostToDevice) != hipSuccess) return fail(SBR_ERR_HIP);
        sbr::launch_repack_lstm(v, m->stream);
    } else {
        if ((st = dmalloc(&v.alpha, d)) != SBR_OK) return fail(st);
        if ((st = dmalloc(&v.alpha_acc, d)) != SBR_OK) return fail(st);
        if (adam) {
            if ((st = dmalloc(&v.alpha_m, d)) != SBR_OK) return fail(st);
            hipMemsetAsync(v.alpha_m, 0, d * 4, m->stream);
        }
        hipMemsetAsync(v.alpha, 0, d * 4, m->stream);
        hipMemsetAsync(v.alpha_acc, 0, d * 4, m->stream);
        /* the reference also draws the two unused d x d dense_init matrices fc1, fc2 from this RNG
         * (ewma.rs:179-188): drawn and dropped, so the driver's shuffles start where the reference's do */
        const uint64_t dl = (uint64_t)m->dl;
        const double std_fc = std::sqrt(2.0 / (double)(dl + dl));
        for (uint64_t i = 0; i < 2 * dl * dl; ++i) (void)sbr_rand_normal_f32(&m->rng, 0.0, std_fc);
    }
    if (hipStreamSynchronize(m->stream) != hipSuccess) return fail(SBR_ERR_HIP);
    *out = m;
    return SBR_OK;
}

sbr_status sbr_model_create(const sbr_hparams* hp, sbr_model** out) {
    return model_create_impl(hp, nullptr, TABLE_WRITE_ALL, nullptr, out);
}

void sbr_model_destroy(sbr_model* m) {
    if (!m) return;
    hipSetDevice(m->device);
    hipStreamSynchronize(m->stream); /* every stream that may still touch the arrays: they go back to the scratch cache, not the driver */
    if (m->side) hipStreamSynchronize(m->side);
    if (m->sorter) hipStreamSynchronize(m->sorter);
    sbr::ModelView& v = m->mv;
    if (!m->shared) { dfree(v.E); dfree(v.Eacc); dfree(v.b); dfree(v.bacc); dfree(v.Em); dfree(v.bm); }
    dfree(v.W); dfree(v.Wacc); dfree(v.bW); dfree(v.bWacc); dfree(v.Wp); dfree(v.WTp);
    dfree(v.alpha); dfree(v.alpha_acc);
    dfree(v.Wm); dfree(v.bWm); dfree(v.alpha_m);
    for (auto& tp : m->pending) { hipEventDestroy(tp.a); hipEventDestroy(tp.b); }
    m->eval_arena.release();
    if (m->own_stream && m->stream) hipStreamDestroy(m->stream);
    if (m->side) { hipStreamSynchronize(m->side); hipStreamDestroy(m->side); }
    if (m->sorter) { hipStreamSynchronize(m->sorter); hipStreamDestroy(m->sorter); }
    if (m->copier) { hipStreamSynchronize(m->copier); hipStreamDestroy(m->copier); }
    if (m->ev_fork) hipEventDestroy(m->ev_fork);
    if (m->ev_join) hipEventDestroy(m->ev_join);
    if (m->ev_scored) hipEventDestroy(m->ev_scored);
    if (m->ev_sorted) hipEventDestroy(m->ev_sorted);
    delete m;
}

sbr_status sbr_model_set_stream(sbr_model* m, void* hip_stream) {
    if (!m) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->own_stream && m->stream) hipStreamDestroy(m->stream);
    m->stream = reinterpret_cast<hipStream_t>(hip_stream);
    m->own_stream = false;
    return SBR_OK;
}

sbr_status sbr_model_synchronize(sbr_model* m) {
    if (!m) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    return SBR_OK;
}

/* Parameter arrays as the caller sees them: shapes in embedding_dim (dl) — E [I][dl], W [2 dl][ng dl] (rows [x ; h],
 * column blocks per gate), bW [ng dl], alpha [dl].  Storage is in the padded width d (storage_dim): `stored` is the
 * device array's element count, `shape` says how logical elements map into it. */
enum ParamShape { SHAPE_FLAT = 0, SHAPE_ROWS = 1, SHAPE_LSTM_W = 2, SHAPE_LSTM_B = 3 };
static float* param_ptr(sbr_model* m, int32_t which, uint64_t* count, uint64_t* stored = nullptr, int* shape = nullptr) {
    const uint64_t I = m->hp.num_items, d = (uint64_t)m->d, dl = (uint64_t)m->dl, ng = (uint64_t)m->ng;
    sbr::ModelView& v = m->mv;
    float* p = nullptr;
    uint64_t n = 0, ns = 0;
    int sh = SHAPE_FLAT;
    switch (which) {
        case SBR_PARAM_ITEM_EMBEDDING: p = v.E; n = I * dl; ns = I * d; sh = SHAPE_ROWS; break;
        case SBR_PARAM_ITEM_EMBEDDING_ACC: p = v.Eacc; n = I * dl; ns = I * d; sh = SHAPE_ROWS; break;
        case SBR_PARAM_ITEM_BIAS: p = v.b; n = ns = I; break;
        case SBR_PARAM_ITEM_BIAS_ACC: p = v.bacc; n = ns = I; break;
        case SBR_PARAM_LSTM_W: p = v.W; n = 2 * dl * ng * dl; ns = 2 * d * ng * d; sh = SHAPE_LSTM_W; break;
        case SBR_PARAM_LSTM_W_ACC: p = v.Wacc; n = 2 * dl * ng * dl; ns = 2 * d * ng * d; sh = SHAPE_LSTM_W; break;
        case SBR_PARAM_LSTM_B: p = v.bW; n = ng * dl; ns = ng * d; sh = SHAPE_LSTM_B; break;
        case SBR_PARAM_LSTM_B_ACC: p = v.bWacc; n = ng * dl; ns = ng * d; sh = SHAPE_LSTM_B; break;
        case SBR_PARAM_EWMA_ALPHA: p = v.alpha; n = ng ? 0 : dl; ns = ng ? 0 : d; break;
        case SBR_PARAM_EWMA_ALPHA_ACC: p = v.alpha_acc; n = ng ? 0 : dl; ns = ng ? 0 : d; break;
        case SBR_PARAM_ITEM_EMBEDDING_M: p = v.Em; n = v.Em ? I * dl : 0; ns = v.Em ? I * d : 0; sh = SHAPE_ROWS; break;
        case SBR_PARAM_ITEM_BIAS_M: p = v.bm; n = ns = v.bm ? I : 0; break;
        case SBR_PARAM_LSTM_W_M: p = v.Wm; n = v.Wm ? 2 * dl * ng * dl : 0; ns = v.Wm ? 2 * d * ng * d : 0; sh = SHAPE_LSTM_W; break;
        case SBR_PARAM_LSTM_B_M: p = v.bWm; n = v.bWm ? ng * dl : 0; ns = v.bWm ? ng * d : 0; sh = SHAPE_LSTM_B; break;
        case SBR_PARAM_EWMA_ALPHA_M: p = v.alpha_m; n = v.alpha_m ? dl : 0; ns = v.alpha_m ? d : 0; break;
    }
    *count = n;
    if (stored) *stored = ns;
    if (shape) *shape = sh;
    return p;
}

static bool is_table_optimizer_state(int32_t which) {
    return which == SBR_PARAM_ITEM_EMBEDDING_ACC || which == SBR_PARAM_ITEM_BIAS_ACC || which == SBR_PARAM_ITEM_EMBEDDING_M || which == SBR_PARAM_ITEM_BIAS_M;
}

/* copy between the caller's logical array and the stored (padded) array, host side; to_stored fills the padding with 0 */
static void param_repack(const sbr_model* m, int shape, const float* src, float* dst, bool to_stored) {
    const uint64_t I = m->hp.num_items, d = (uint64_t)m->d, dl = (uint64_t)m->dl, ng = (uint64_t)m->ng;
    auto move = [&](uint64_t logical, uint64_t stored, uint64_t n) {
        if (to_stored) std::memcpy(dst + stored, src + logical, n * 4);
        else std::memcpy(dst + logical, src + stored, n * 4);
    };
    switch (shape) {
        case SHAPE_ROWS:
            for (uint64_t r = 0; r < I; ++r) move(r * dl, r * d, dl);
            break;
        case SHAPE_LSTM_W:
            for (uint64_t kl = 0; kl < 2 * dl; ++kl) {
                const uint64_t k = kl < dl ? kl : d + (kl - dl);
                for (uint64_t g = 0; g < ng; ++g) move(kl * ng * dl + g * dl, k * ng * d + g * d, dl);
            }
            break;
        case SHAPE_LSTM_B:
            for (uint64_t g = 0; g < ng; ++g) move(g * dl, g * d, dl);
            break;
        default:
            move(0, 0, dl);  /* alpha */
            break;
    }
}

sbr_status sbr_model_param_count(const sbr_model* m, int32_t which, uint64_t* out_count) {
    if (!m || !out_count) return SBR_ERR_INVALID_ARGUMENT;
    param_ptr(const_cast<sbr_model*>(m), which, out_count);
    return SBR_OK;
}

sbr_status sbr_model_get_param(sbr_model* m, int32_t which, float* host_out, uint64_t count) {
    if (!m || !host_out) return SBR_ERR_INVALID_ARGUMENT;
    uint64_t n = 0, stored = 0;
    int shape = SHAPE_FLAT;
    float* p = param_ptr(m, which, &n, &stored, &shape);
    if (!p || n != count) return SBR_ERR_INVALID_ARGUMENT;
    if (m->opt_state_partial && is_table_optimizer_state(which)) return SBR_ERR_INVALID_ARGUMENT; /* gather the owners' slices first */
    SBRCHK(ensure_device(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (stored == n) {
        HIPCHK(hipMemcpy(host_out, p, n * 4, hipMemcpyDeviceToHost));
    } else {
        std::vector<float> tmp(stored);
        HIPCHK(hipMemcpy(tmp.data(), p, stored * 4, hipMemcpyDeviceToHost));
        param_repack(m, shape, tmp.data(), host_out, false);
    }
    return SBR_OK;
}

/* selected rows of an item-table block (embeddings and their optimiser state: [n][embedding_dim]; biases: [n]) */
sbr_status sbr_model_get_param_rows(sbr_model* m, int32_t which, const uint32_t* rows, uint64_t n, float* host_out) {
    if (!m || !host_out || (n && !rows)) return SBR_ERR_INVALID_ARGUMENT;
    uint64_t count = 0, stored = 0;
    int shape = SHAPE_FLAT;
    float* p = param_ptr(m, which, &count, &stored, &shape);
    const uint64_t I = m->hp.num_items;
    const bool table = shape == SHAPE_ROWS, bias = shape == SHAPE_FLAT && stored == I && (which == SBR_PARAM_ITEM_BIAS || which == SBR_PARAM_ITEM_BIAS_ACC || which == SBR_PARAM_ITEM_BIAS_M);
    if (!p || !count || (!table && !bias)) return SBR_ERR_INVALID_ARGUMENT;
    if (m->opt_state_partial && is_table_optimizer_state(which)) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    const uint64_t w = table ? (uint64_t)m->dl : 1, ws = table ? (uint64_t)m->d : 1;
    for (uint64_t i = 0; i < n; ++i) {
        if (rows[i] >= I) return SBR_ERR_INVALID_ARGUMENT;
        HIPCHK(hipMemcpy(host_out + i * w, p + (uint64_t)rows[i] * ws, w * 4, hipMemcpyDeviceToHost));
    }
    return SBR_OK;
}

sbr_status sbr_model_set_param(sbr_model* m, int32_t which, const float* host_in, uint64_t count) {
    if (!m || !host_in) return SBR_ERR_INVALID_ARGUMENT;
    uint64_t n = 0, stored = 0;
    int shape = SHAPE_FLAT;
    float* p = param_ptr(m, which, &n, &stored, &shape);
    if (!p || n != count) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (stored == n) {
        HIPCHK(hipMemcpy(p, host_in, n * 4, hipMemcpyHostToDevice));
    } else {
        std::vector<float> tmp(stored, 0.0f);
        param_repack(m, shape, host_in, tmp.data(), true);
        HIPCHK(hipMemcpy(p, tmp.data(), stored * 4, hipMemcpyHostToDevice));
    }
    if (which == SBR_PARAM_LSTM_W) {
        sbr::launch_repack_lstm(m->mv, m->stream);
        HIPCHK(hipStreamSynchronize(m->stream));
    }
    return SBR_OK;
}

sbr_status sbr_model_get_epoch(const sbr_model* m, uint64_t* out_global_epoch) {
    if (!m || !out_global_epoch) return SBR_ERR_INVALID_ARGUMENT;
    *out_global_epoch = m->global_epoch;
    return SBR_OK;
}

sbr_status sbr_model_get_counters(const sbr_model* m, uint64_t* out_global_epoch, uint64_t* out_optimizer_steps) {
    if (!m) return SBR_ERR_INVALID_ARGUMENT;
    if (out_global_epoch) *out_global_epoch = m->global_epoch;
    if (out_optimizer_steps) *out_optimizer_steps = m->opt_steps;
    return SBR_OK;
}

sbr_status sbr_model_set_counters(sbr_model* m, uint64_t global_epoch, uint64_t optimizer_steps) {
    if (!m) return SBR_ERR_INVALID_ARGUMENT;
    m->global_epoch = global_epoch;
    m->opt_steps = optimizer_steps;
    return SBR_OK;
}

sbr_status sbr_model_get_rng(const sbr_model* m, uint8_t out_state[16]) {
    if (!m || !out_state) return SBR_ERR_INVALID_ARGUMENT;
    const uint32_t w[4] = {m->rng.x, m->rng.y, m->rng.z, m->rng.w};
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 4; ++b) out_state[4 * i + b] = (uint8_t)(w[i] >> (8 * b));
    return SBR_OK;
}

sbr_status sbr_model_set_rng(sbr_model* m, const uint8_t state[16]) {
    if (!m || !state) return SBR_ERR_INVALID_ARGUMENT;
    sbr_xs_seed(&m->rng, state);
    return SBR_OK;
}

sbr_status sbr_model_set_overlap(sbr_model* m, int32_t enable) {
    if (!m) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipStreamSynchronize(m->side));
    HIPCHK(hipStreamSynchronize(m->sorter));
    m->overlap = enable != 0;
    return SBR_OK;
}

sbr_status sbr_model_timing_enable(sbr_model* m, int32_t enable) {
    if (!m) return SBR_ERR_INVALID_ARGUMENT;
    m->timing = enable != 0;
    return SBR_OK;
}

sbr_status sbr_model_timing_select(sbr_model* m, uint32_t family_mask) {
    if (!m) return SBR_ERR_INVALID_ARGUMENT;
    m->timing_mask = family_mask;
    return SBR_OK;
}

sbr_status sbr_model_timing_read(sbr_model* m, double* out_ms, uint64_t* out_launches) {
    if (!m) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    for (auto& tp : m->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, tp.a, tp.b) == hipSuccess) {
            m->ms[tp.family] += (double)ms;
            m->launches[tp.family] += tp.launches;
        }
        hipEventDestroy(tp.a);
        hipEventDestroy(tp.b);
    }
    m->pending.clear();
    for (int f = 0; f < SBR_K_FAMILIES; ++f) {
        if (out_ms) out_ms[f] = m->ms[f];
        if (out_launches) out_launches[f] = m->launches[f];
        m->ms[f] = 0.0;
        m->launches[f] = 0;
    }
    return SBR_OK;
}

/* ---------------------------------------------------------------------------------------------
 * fit
 * ------------------------------------------------------------------------------------------- */
sbr_status sbr_fit_begin(sbr_model* m, const uint64_t* user_ptr, const uint32_t* item_ids, uint64_t num_users,
                         sbr_fit_plan** out) {
    if (!m || !user_ptr || !out) return SBR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    SBRCHK(ensure_device(m));
    const uint64_t T = m->hp.max_sequence_length;
    for (uint64_t u = 0; u < num_users; ++u)
        if (user_ptr[u + 1] < user_ptr[u]) return SBR_ERR_INVALID_ARGUMENT; /* pointers must be non-decreasing */
    const uint64_t nnz = user_ptr[num_users];
    if (nnz && !item_ids) return SBR_ERR_INVALID_ARGUMENT;
    for (uint64_t i = 0; i < nnz; ++i)
        if (item_ids[i] >= m->hp.num_items) return SBR_ERR_INVALID_ARGUMENT;
    /* subsequences = chunks (first chunk short, data.rs:406-431) with len > 2 (sequence_model.rs:76-83) */
    std::vector<uint64_t> start;
    std::vector<uint32_t> len;
    for (uint64_t u = 0; u < num_users; ++u) {
        const uint64_t n = user_ptr[u + 1] - user_ptr[u];
        uint64_t idx = 0;
        while (idx < n) {
            const uint64_t mod = (n - idx) % T, cs = mod == 0 ? T : mod;
            if (cs > 2) { start.push_back(user_ptr[u] + idx); len.push_back((uint32_t)cs); }
            idx += cs;
        }
    }
    const uint64_t nseq = start.size();
    if (nseq == 0) return SBR_ERR_NO_INTERACTIONS; /* :86-88 */
    shuffle_pairs(start.data(), len.data(), nseq, &m->rng); /* :84, model RNG */
    const int ndev = (int)m->hp.num_devices;
    const uint64_t part = nseq / ndev; /* :91; the zip at :94-98 drops the remainder */
    if (part == 0) return SBR_ERR_INVALID_ARGUMENT; /* the reference panics here (chunks_mut(0)) */
    sbr_fit_plan* p = new (std::nothrow) sbr_fit_plan();
    if (!p) return SBR_ERR_OUT_OF_MEMORY;
    p->m = m; p->ndev = ndev; p->rank = (int)m->hp.device_rank; p->T = (int)T;
    p->nseq_total = nseq; p->part_len = part;
    start.resize((size_t)ndev * part);
    len.resize((size_t)ndev * part);
    p->seq_start.swap(start);
    p->seq_len.swap(len);
    p->part_rng.resize(ndev);
    p->fit_seed.resize(ndev);
    for (int q = 0; q < ndev; ++q) { /* :97 — XorShiftRng::from_seed(parameters.rng().gen()) */
        uint8_t seed[16];
        sbr_rand_gen_seed16(&m->rng, seed);
        sbr_xs_seed(&p->part_rng[q], seed);
        /* (reference order: the worker's stream is not touched here — its first use is the first epoch's shuffle, :109) */
        p->fit_seed[q] = m->reference_order ? 0 : sbr_xs_u64(&p->part_rng[q]);
    }
    p->items.assign(item_ids, item_ids + nnz);
    p->bmax = m->hp.batch_sequences;
    p->rmax = p->bmax * (T - 1);
    if (3 * p->rmax >= (1ull << 32) || part * T >= (1ull << 32)) { delete p; return SBR_ERR_INVALID_ARGUMENT; }
    sbr_status st = alloc_work(m, p->rmax, p->bmax, true, &p->wb);
    if (st == SBR_OK) { p->block_bytes = block_bytes_for(m, p->rmax); st = dmalloc(&p->block, p->block_bytes); }
    const uint64_t max_entries = 3 * p->rmax; /* only a device's own entries are ever sorted */
    int item_bits = 1;
    while ((1ull << item_bits) < (uint64_t)m->hp.num_items) ++item_bits;
    p->key_bits = 32 + item_bits;
    if (st == SBR_OK) st = dmalloc(&p->keys, max_entries);
    p->exportable_lists = m->shared && m->shared->local_rank >= 0;
    if (st == SBR_OK) {
        if (p->exportable_lists) {
            st = p->own_x[0].alloc(max_entries * sizeof(uint64_t), m->device);
            p->keys_sorted = reinterpret_cast<uint64_t*>(p->own_x[0].ptr);
        } else {
            st = dmalloc(&p->keys_sorted, max_entries);
        }
    }
    if (st == SBR_OK) {
        p->sort_temp_bytes = sbr::sparse_sort_temp_bytes(max_entries, p->key_bits);
        uint8_t* tmp = nullptr;
        st = dmalloc(&tmp, p->sort_temp_bytes);
        p->sort_temp = tmp;
    }
    {   /* a long segment has more than SBR_SEG_CHUNK entries: at most max_entries / chunk of them, and
         * their chunks number at most max_entries / chunk + (number of long segments) */
        const uint64_t cap = max_entries / SBR_SEG_ROUTE + 2, units = cap + max_entries / SBR_SEG_CHUNK + 2; /* routed segments; their chunks */
        p->seg.cap = (uint32_t)cap;
        if (st == SBR_OK) st = dmalloc(&p->seg.counters, 4);
        if (st == SBR_OK) st = dmalloc(&p->seg.long_start, cap);
        if (st == SBR_OK) st = dmalloc(&p->seg.long_end, cap);
        if (st == SBR_OK) st = dmalloc(&p->seg.unit_base, cap + 1);
        if (st == SBR_OK) st = dmalloc(&p->seg.P, units * (uint64_t)m->d);
        if (st == SBR_OK) st = dmalloc(&p->seg.Pb, units);
        if (st == SBR_OK) st = dmalloc(&p->seg.Pf, units);
        if (st == SBR_OK) st = dmalloc(&p->seg.head_pos, max_entries + 1);
        if (st == SBR_OK) st = dmalloc(&p->seg.nheads, 1);
    }
    if (st == SBR_OK && !m->copier && hipStreamCreateWithFlags(&m->copier, hipStreamNonBlocking) != hipSuccess) st = SBR_ERR_HIP;
    p->copy_stream = m->copier;
    for (int i = 0; i < 2 && st == SBR_OK; ++i)
        if (hipEventCreateWithFlags(&p->ep[i].free_event, hipEventDisableTiming) != hipSuccess) st = SBR_ERR_HIP;
    if (st == SBR_OK) st = dmalloc(&p->lag_state, 1 + 2 * T); /* accumulator + (node, staged value) per step: sbr_report.hip */
    if (st == SBR_OK) st = dmalloc(&p->lag_seqsum, p->bmax);
    if (st == SBR_OK && (hipEventCreateWithFlags(&p->ev_seqsum, hipEventDisableTiming) != hipSuccess ||
                         hipEventCreateWithFlags(&p->ev_lagged, hipEventDisableTiming) != hipSuccess ||
                         hipEventCreateWithFlags(&p->ev_hot, hipEventDisableTiming) != hipSuccess)) st = SBR_ERR_HIP;
    if (st == SBR_OK) st = dmalloc(&p->loss_acc, 17);  /* [0] all devices, [1 + q] device q */
    if (st == SBR_OK && m->reference_order) st = dmalloc(&p->ref_rng, 4);
    if (st == SBR_OK) st = dmalloc(&p->phase_clocks, 6);
    if (st == SBR_OK) hipMemsetAsync(p->phase_clocks, 0, 6 * sizeof(unsigned long long), m->stream);
    if (st == SBR_OK) st = dmalloc(&p->ex_acc, 18);    /* [0] examples, [1] negatives scored, [2 + q] examples of device q */
    if (st != SBR_OK) { sbr_fit_plan_destroy(p); return st; }
    hipMemsetAsync(p->lag_state, 0, (1 + 2 * T) * sizeof(float), m->stream); /* the reference builds its loss nodes per fit call */
    hipMemsetAsync(p->loss_acc, 0, 17 * sizeof(double), m->stream);
    hipMemsetAsync(p->ex_acc, 0, 18 * sizeof(unsigned long long), m->stream);
    hipMemsetAsync(p->block, 0, p->block_bytes, m->stream);
    *out = p;
    return SBR_OK;
}

void sbr_fit_plan_destroy(sbr_fit_plan* p) {
    if (!p) return;
    hipSetDevice(p->m->device);
    if (p->pending) { p->worker.join(); p->pending = false; }
    hipStreamSynchronize(p->m->side);
    hipStreamSynchronize(p->m->sorter);
    hipStreamSynchronize(p->m->stream);
    for (int i = 0; i < 2; ++i) {
        p->ep[i].dp.release();
        dfree(p->ep[i].d_desc);
        if (p->ep[i].free_event) hipEventDestroy(p->ep[i].free_event);
        hfree(p->ep[i].h_in); hfree(p->ep[i].h_out); hfree(p->ep[i].h_ctr);
        hfree(p->ep[i].h_prev); hfree(p->ep[i].h_steps);
    }
    if (p->copy_stream) hipStreamSynchronize(p->copy_stream); /* the model's: stays */
    p->wb.release();
    for (auto& px : p->xchg_peer) for (auto& b : px) b.release();
    for (auto& b : p->xchg_own) b.release();
    if (p->exportable_lists) { /* the sorted keys and the list are virtual-memory allocations, not hipMalloc'ed */
        for (auto& px : p->peer_x) for (auto& b : px) b.release();
        for (auto& b : p->own_x) b.release();
        p->keys_sorted = nullptr; p->glist = p->gblist = nullptr; p->gfl = nullptr;
    }
    dfree(p->block); dfree(p->keys); dfree(p->keys_sorted); dfree(p->sort_temp);
    dfree(p->loss_acc); dfree(p->ex_acc); dfree(p->phase_clocks); dfree(p->ref_rng);
    dfree(p->lag_state); dfree(p->lag_seqsum);
    if (p->ev_seqsum) hipEventDestroy(p->ev_seqsum);
    if (p->ev_lagged) hipEventDestroy(p->ev_lagged);
    if (p->ev_hot) hipEventDestroy(p->ev_hot);
    dfree(p->seg.counters); dfree(p->seg.long_start); dfree(p->seg.long_end); dfree(p->seg.unit_base);
    dfree(p->seg.P); dfree(p->seg.Pb); dfree(p->seg.Pf);
    dfree(p->seg.head_pos); dfree(p->seg.nheads);
    dfree(p->glist); dfree(p->gblist); dfree(p->gfl); dfree(p->bounds_dev);
    dfree(p->mkeys); dfree(p->mkeys_sorted); dfree(p->msort_temp); dfree(p->mplan); dfree(p->all_bounds_dev);
    delete p;
}

/* Host side of one epoch: ≙ thread_rng.shuffle(partition) (sequence_model.rs:109) for every
 * partition (so that each device also knows the row counts its peers contribute to a step), then
 * packing of this rank's minibatches and upload of the packed index arrays on the copy stream. */
static sbr_status build_epoch(sbr_fit_plan* p, sbr_fit_plan::Epoch& e) {
    sbr_model* m = p->m;
    HIPCHK(hipSetDevice(m->device));
    if (p->ref_rng && p->ref_rng_live) { /* reference order: the stream the steps drew from shuffles the next epoch (:109) */
        HIPCHK(hipStreamSynchronize(m->stream));
        uint32_t st4w[4];
        HIPCHK(hipMemcpy(st4w, p->ref_rng, sizeof(st4w), hipMemcpyDeviceToHost));
        sbr_xorshift& r = p->part_rng[p->rank];
        r.x = st4w[0]; r.y = st4w[1]; r.z = st4w[2]; r.w = st4w[3];
        p->ref_rng_live = false;
    }
    for (int q = 0; q < p->ndev; ++q)
        shuffle_pairs(p->seq_start.data() + (size_t)q * p->part_len, p->seq_len.data() + (size_t)q * p->part_len,
                      p->part_len, &p->part_rng[q]);
    if (p->ref_rng) {
        const sbr_xorshift& r = p->part_rng[p->rank];
        const uint32_t st4w[4] = {r.x, r.y, r.z, r.w};
        HIPCHK(hipMemcpy(p->ref_rng, st4w, sizeof(st4w), hipMemcpyHostToDevice));
    }
    e.epoch_key_epoch = m->global_epoch;
    m->global_epoch += 1;
    const uint64_t B = p->bmax;
    const uint64_t nmb = (p->part_len + B - 1) / B;
    e.num_mb = nmb;
    e.rows_of_dev.assign(nmb * p->ndev, 0);
    for (int q = 0; q < p->ndev; ++q) {
        const uint32_t* ln = p->seq_len.data() + (size_t)q * p->part_len;
        for (uint64_t mb = 0; mb < nmb; ++mb) {
            const uint64_t p0 = mb * B, p1 = std::min(p0 + B, p->part_len);
            uint32_t r = 0;
            for (uint64_t i = p0; i < p1; ++i) r += ln[i] - 1;
            e.rows_of_dev[mb * p->ndev + q] = r;
        }
    }
    const uint64_t* st = p->seq_start.data() + (size_t)p->rank * p->part_len;
    const uint32_t* ln = p->seq_len.data() + (size_t)p->rank * p->part_len;
    e.mbs.resize(nmb);
    e.off_host.clear();
    uint64_t total_rows = 0;
    for (uint64_t i = 0; i < p->part_len; ++i) total_rows += ln[i] - 1;
    if (total_rows > e.h_rows_cap || p->part_len > e.h_seq_cap) {
        hfree(e.h_in); hfree(e.h_out); hfree(e.h_ctr); hfree(e.h_prev); hfree(e.h_steps);
        e.h_in = e.h_out = e.h_ctr = nullptr; e.h_prev = e.h_steps = nullptr;
        e.h_rows_cap = e.h_seq_cap = 0;
        const size_t rb = (size_t)(total_rows ? total_rows : 1) * 4, sb = (size_t)(p->part_len ? p->part_len : 1) * 4;
        if (hmalloc(&e.h_in, rb) != hipSuccess || hmalloc(&e.h_out, rb) != hipSuccess || hmalloc(&e.h_ctr, rb) != hipSuccess ||
            hmalloc(&e.h_prev, rb) != hipSuccess || hmalloc(&e.h_steps, sb) != hipSuccess)
            return SBR_ERR_OUT_OF_MEMORY;
        e.h_rows_cap = total_rows; e.h_seq_cap = p->part_len;
    }
    /* The staging buffers are read by the DMA of two epochs ago at the latest, which was waited for below
     * before that epoch's build returned: they are free.  Packing (time-major, sequences by length
     * descending, stable): counting sort of the minibatch, then the rows are written in row order — step t
     * outer, sequence inner — so every output array is filled sequentially; the t range is cut into pieces
     * of about equal row count for a few worker threads. */
    uint64_t row_base = 0, seq_base = 0;
    const int T = p->T;
    std::vector<int> cnt(T + 2), pos(T + 2), order;
    for (uint64_t mb = 0; mb < nmb; ++mb) {
        const uint64_t p0 = mb * B, p1 = std::min(p0 + B, p->part_len);
        const int nb = (int)(p1 - p0);
        std::fill(cnt.begin(), cnt.end(), 0);
        for (int i = 0; i < nb; ++i) cnt[ln[p0 + i] - 1]++;
        int acc = 0;
        for (int l = T; l >= 0; --l) { pos[l] = acc; acc += cnt[l]; }
        order.resize(nb);
        for (int i = 0; i < nb; ++i) order[pos[ln[p0 + i] - 1]++] = i;
        int* steps = e.h_steps + seq_base;
        for (int b = 0; b < nb; ++b) steps[b] = (int)ln[p0 + order[b]] - 1;
        const int Tm = nb ? steps[0] : 0;
        const size_t off_base = e.off_host.size();
        e.off_host.resize(off_base + Tm + 1);
        int* off = e.off_host.data() + off_base;
        off[0] = 0;
        {
            int alive = nb;
            for (int t = 0; t < Tm; ++t) {
                while (alive > 0 && steps[alive - 1] <= t) --alive;
                off[t + 1] = off[t] + alive;
            }
        }
        const int R = off[Tm];
        uint32_t *in_idx = e.h_in + row_base, *out_idx = e.h_out + row_base, *ctr = e.h_ctr + row_base;
        int* prev_row = e.h_prev + row_base;
        auto fill = [&](int t0, int t1) {
            for (int t = t0; t < t1; ++t) {
                const int alive = off[t + 1] - off[t];
                const int r0 = off[t], rp = t ? off[t - 1] : 0;
                for (int b = 0; b < alive; ++b) {
                    const uint64_t src = p0 + (uint64_t)order[b];
                    const uint32_t* it = p->items.data() + st[src];
                    in_idx[r0 + b] = it[t];
                    out_idx[r0 + b] = it[t + 1];
                    ctr[r0 + b] = (uint32_t)(src * (uint64_t)T + (uint64_t)t);
                    prev_row[r0 + b] = t ? rp + b : -1;
                }
            }
        };
        static const int pack_threads = [] {  /* at most half of the host's cores */
            const unsigned hc = std::thread::hardware_concurrency();
            const int half = hc >= 2 ? (int)(hc / 2) : 1;
            return half < SBR_PACK_THREADS ? half : SBR_PACK_THREADS;
        }();
        const int nthreads = R > (1 << 18) ? pack_threads : 1;
        if (nthreads == 1) fill(0, Tm);
        else {
            std::vector<std::thread> workers;
            int t0 = 0;
            for (int k = 0; k < nthreads; ++k) {
                int t1 = t0;
                const long long target = (long long)R * (k + 1) / nthreads;
                while (t1 < Tm && off[t1] < target) ++t1;
                if (k == nthreads - 1) t1 = Tm;
                if (t1 > t0) workers.emplace_back(fill, t0, t1);
                t0 = t1;
            }
            for (auto& w : workers) w.join();
        }
        sbr_fit_plan::Mb& d = e.mbs[mb];
        d.R = R; d.B = nb; d.Tm = Tm;
        d.row_base = row_base; d.seq_base = seq_base; d.off_base = off_base;
        row_base += (uint64_t)R;
        seq_base += (uint64_t)nb;
    }
    /* the GPU may still be reading this buffer's previous contents (two epochs ago) */
    if (e.free_recorded) {
        HIPCHK(hipEventSynchronize(e.free_event));
        e.free_recorded = false;
    }
    if (row_base > e.rows_cap || e.off_host.size() > e.off_cap || seq_base > e.seq_cap) {
        e.dp.release();
        SBRCHK(dmalloc(&e.dp.in_idx, row_base));
        SBRCHK(dmalloc(&e.dp.out_idx, row_base));
        SBRCHK(dmalloc(&e.dp.ctr, row_base));
        SBRCHK(dmalloc(&e.dp.prev_row, row_base));
        SBRCHK(dmalloc(&e.dp.off, e.off_host.size()));
        SBRCHK(dmalloc(&e.dp.steps, seq_base));
        e.rows_cap = row_base; e.off_cap = e.off_host.size(); e.seq_cap = seq_base;
    }
    hipStream_t cs = p->copy_stream;
    e.desc_host.clear();
    if (B == 1 && p->ndev == 1) {
        e.desc_host.resize(nmb);
        for (uint64_t mb = 0; mb < nmb; ++mb) {
            const sbr_fit_plan::Mb& d = e.mbs[mb];
            e.desc_host[mb] = sbr::StepDesc{(uint32_t)d.R, (uint32_t)d.row_base, (uint32_t)d.off_base, (uint32_t)d.seq_base};
        }
        if (nmb > e.desc_cap) {
            dfree(e.d_desc);
            e.d_desc = nullptr; e.desc_cap = 0;
            SBRCHK(dmalloc(&e.d_desc, nmb));
            e.desc_cap = nmb;
        }
        HIPCHK(hipMemcpyAsync(e.d_desc, e.desc_host.data(), nmb * sizeof(sbr::StepDesc), hipMemcpyHostToDevice, cs));
    }
    HIPCHK(hipMemcpyAsync(e.dp.in_idx, e.h_in, row_base * 4, hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(e.dp.out_idx, e.h_out, row_base * 4, hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(e.dp.ctr, e.h_ctr, row_base * 4, hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(e.dp.prev_row, e.h_prev, row_base * 4, hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(e.dp.off, e.off_host.data(), e.off_host.size() * 4, hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(e.dp.steps, e.h_steps, seq_base * 4, hipMemcpyHostToDevice, cs));
    HIPCHK(hipStreamSynchronize(cs));
    return SBR_OK;
}

sbr_status sbr_fit_epoch_prepare(sbr_fit_plan* p, uint64_t* out_num_minibatches) {
    if (!p) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    if (p->pending) { /* the next epoch was prefetched: hand the current buffers back and switch */
        p->worker.join();
        p->pending = false;
        SBRCHK(p->pending_status);
        sbr_fit_plan::Epoch& old = p->ep[p->cur];
        HIPCHK(hipEventRecord(old.free_event, m->stream));
        old.free_recorded = true;
        p->cur ^= 1;
    } else {
        sbr_fit_plan::Epoch& e = p->ep[p->cur];
        HIPCHK(hipEventRecord(e.free_event, m->stream));
        e.free_recorded = true;
        SBRCHK(build_epoch(p, e));
    }
    if (out_num_minibatches) *out_num_minibatches = p->ep[p->cur].num_mb;
    return SBR_OK;
}

sbr_status sbr_fit_epoch_prefetch(sbr_fit_plan* p) {
    if (!p || p->pending) return SBR_ERR_INVALID_ARGUMENT;
    if (p->ref_rng) return SBR_OK; /* reference order: the next epoch's shuffle continues the stream this epoch's steps draw from */
    p->pending = true;
    p->pending_status = SBR_OK;
    p->worker = std::thread([p]() { p->pending_status = build_epoch(p, p->ep[p->cur ^ 1]); });
    return SBR_OK;
}

sbr_status sbr_fit_minibatch_rows(const sbr_fit_plan* p, uint64_t minibatch, uint64_t* out_rows) {
    if (!p || !out_rows || minibatch >= p->ep[p->cur].num_mb) return SBR_ERR_INVALID_ARGUMENT;
    *out_rows = (uint64_t)p->ep[p->cur].mbs[minibatch].R;
    return SBR_OK;
}

static sbr_status join_dense(sbr_fit_plan* p) {
    if (p->dense_pending) {
        HIPCHK(hipStreamWaitEvent(p->m->stream, p->m->ev_join, 0));
        p->dense_pending = false;
    }
    return SBR_OK;
}

/* blk.dense complete on the main stream: joins the GEMM and, if its chunk partials are still unreduced, reduces them */
static sbr_status ensure_dense_reduced(sbr_fit_plan* p) {
    SBRCHK(join_dense(p));
    if (p->dense_unreduced_chunks > 0) {
        sbr::launch_dense_reduce(p->m->mv, p->wb.v, p->dense_unreduced_chunks, block_view(p->m, p->block, p->rmax), p->m->stream);
        p->dense_unreduced_chunks = 0;
    }
    return SBR_OK;
}

sbr_status sbr_fit_step_local(sbr_fit_plan* p, uint64_t minibatch) {
    if (!p || minibatch >= p->ep[p->cur].num_mb) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    void* block = p->block;
    const sbr_fit_plan::Epoch& ep = p->ep[p->cur];
    const sbr_fit_plan::Mb& mb = ep.mbs[minibatch];
    const sbr::MbView mv = mb_view(p, minibatch);
    const sbr::BlockView bv = block_view(m, block, p->rmax);
    const int* off_host = ep.off_host.data() + mb.off_base;
    /* The sort of the sparse-update keys needs only the index arrays and the negatives: it runs on its own stream,
     * underneath the backward pass (WARP: the negatives come out of the score kernel) or, for the single-negative losses
     * whose negatives are a hash of the row counter, from the very start of the step (joined by step_apply / step_scatter). */
    /* A small step (its sparse update is the single-launch form: <= 4 096 keys) is a chain of launches of a few microseconds
     * each; the event hand-offs between three streams then cost more than the overlap buys (MovieLens-100K at one sequence per
     * step: 1.67 s with the side streams, 1.57 s on one), so everything is queued on the main stream. */
    constexpr int small_rows = 1365;
    const bool overlap = m->overlap && mb.R > small_rows;
    hipStream_t side = overlap ? m->side : m->stream;
    /* Where the key ordering of a WARP step runs (its negatives come out of the score kernel): on its own stream from the end of the
     * score kernel on — it takes whatever slots the backward pass's retiring workgroups free, and is finished by the time the update
     * needs the keys; a small step (everything on one stream) orders between score and backward pass.  (Measured and dropped: the
     * main stream before or right after the backward pass for large steps — its 0.1-0.3 ms are then on the critical path;
     * profiles/r04_tail_experiments.md (c).) */
    enum { SORT_OWN_STREAM, SORT_PRE };
    const int place = overlap ? SORT_OWN_STREAM : SORT_PRE;
    hipStream_t sorter = place == SORT_OWN_STREAM ? m->sorter : m->stream;
    const bool early_sort = m->hp.loss != SBR_LOSS_WARP;
    const uint64_t epoch_key = sbr_epoch_key(p->fit_seed[p->rank], ep.epoch_key_epoch);
    /* WARP step with the ordering on its own stream: the per-sequence loss sums and the block header (row count, loss sum; the
     * single-device accumulators) are queued on THAT stream, behind the score kernel's event and ahead of the ordering — between
     * score and BPTT they were two launches and their gaps (~35 us of a 2.5 ms step) on the critical path for nobody's benefit.
     * Their consumers (update / scatter / dense / fit_end) all join the ordering's stream first. */
    const bool side_header = overlap && !early_sort && place == SORT_OWN_STREAM;
    /* ONE subsequence per step at d <= 32 (the reference's own schedule): header, lagged loss figure and the ordering of the step's
     * keys run at the end of the score launch (sbr::SmallTail) — three launches of ~5 us fewer in a step of ~40-100 us */
    const bool small_tail = (m->step_fusion >= 1 || m->reference_order) && !overlap && (p->ndev == 1 || m->reference_order) &&
                            sbr::small_tail_shape_ok(m->mv, (int)mb.B, (int)mb.R, m->reference_order);
    auto launch_sort = [&](hipStream_t on) -> sbr_status {
        if (on != m->stream) {
            /* everything before: the previous step's readers of the keys, this step's score (a WARP step records the event
             * itself, before it queues the backward pass) */
            if (early_sort) HIPCHK(hipEventRecord(m->ev_scored, m->stream));
            HIPCHK(hipStreamWaitEvent(on, m->ev_scored, 0));
        }
        if (side_header) { /* the step's loss bookkeeping rides on the ordering's stream: nothing on the main stream waits for it */
            sbr::launch_seq_loss(mv, p->wb.v.loss, p->lag_seqsum, p->lag_state, mb.B, on);
            sbr::launch_block_header(m->mv, bv, p->wb.v, mv, mb.R, p->header_accumulated ? p->loss_acc : nullptr,
                                     p->header_accumulated ? p->ex_acc : nullptr, nullptr, on);
        }
        {
            ScopedTimer t(m, SBR_K_SPARSE_SORT, 1, on);
            sbr::launch_own_sort(bv, (uint32_t)mb.R, p->keys, p->keys_sorted, p->sort_temp, p->sort_temp_bytes, p->key_bits, p->seg, on,
                                 early_sort ? &mv : nullptr, epoch_key, m->hp.num_items);
        }
        p->sort_off_stream = on != m->stream; /* on the main stream the update is ordered behind the sort anyway: no event */
        if (p->sort_off_stream) {
            HIPCHK(hipEventRecord(m->ev_sorted, on));
            p->sorted_event_live = true;
            /* single device: the hot rows (segments of more than SBR_SEG_CHUNK entries: a skewed catalogue) are listed and their
             * chunk units counted HERE, still underneath BPTT — registering them during the short-segment pass put their three
             * launches behind it on the update's critical path (Zipf(1) items at 8 192 sequences per step: 0.13 of 2.66 ms).
             * (after ev_sorted: the short-segment pass does not wait for the list)
             * Only under sbr_fit_step (fuse_back): the list's only consumer is sbr_fit_step_apply; a caller that drives ONE device
             * through the exchange halves (scatter / reduce_own register long segments themselves, on the main stream) must not
             * find the list kernels running beside them on this stream. */
            if (p->ndev == 1 && overlap && p->fuse_back && 3ull * (uint64_t)mb.R > 4096) {
                sbr::launch_seg_prelist(p->seg, on);
                p->hot_prelisted = true;
            }
        }
        return SBR_OK;
    };
    p->hot_prelisted = false;
    if (early_sort && !small_tail) SBRCHK(launch_sort(overlap ? m->sorter : m->stream));
    if (small_tail) {
        if (p->lag_busy) { /* an earlier step's chain on the ordering's stream still owns lag_state */
            HIPCHK(hipStreamWaitEvent(m->stream, p->ev_lagged, 0));
            p->lag_busy = false;
        }
        p->sort_off_stream = false;
    }
    /* EWMA + single-negative loss (BASELINE configs[4]): scan, scores and backward scan of a sequence in ONE pass (ewma_seq_kernel; same
     * bits as scan | score | backward scan, which EWMA + WARP and the reference-order mode still take: their negatives depend on the
     * scores / on a sequential stream) */
    const bool ewma_fused = !m->ng && m->hp.loss != SBR_LOSS_WARP && mb.R > 0 && !m->reference_order;
    if (ewma_fused) {
        ScopedTimer t(m, SBR_K_SCORE, 1);
        p->header_accumulated = p->ndev == 1;
        const sbr::SmallTail tail{bv.header, p->header_accumulated ? p->loss_acc : nullptr, p->header_accumulated ? p->ex_acc : nullptr,
                                  p->lag_state, p->keys_sorted, p->seg.head_pos, p->seg.nheads};
        sbr::launch_ewma_sequences(m->mv, mv, bv, p->wb.v, epoch_key, mb.R, m->stream, small_tail ? &tail : nullptr);
    } else {
        {
            ScopedTimer t(m, SBR_K_RECURRENT_FWD, m->ng && m->d > 128 ? (uint64_t)mb.Tm : 1); /* d <= 128: one sequence-resident launch */
            sbr::launch_recurrent_forward(m->mv, mv, bv.H, p->wb.v, mb.Tm, off_host, m->stream);
        }
        {
            ScopedTimer t(m, SBR_K_SCORE, 1);
            p->header_accumulated = p->ndev == 1;
            const sbr::SmallTail tail{bv.header, p->header_accumulated ? p->loss_acc : nullptr, p->header_accumulated ? p->ex_acc : nullptr,
                                      p->lag_state, p->keys_sorted, p->seg.head_pos, p->seg.nheads};
            if (m->reference_order) {
                if (!p->ref_rng || !small_tail || !sbr::launch_score_reference_order(m->mv, mv, bv, p->wb.v, p->ref_rng, mb.R, m->stream, tail))
                    return SBR_ERR_UNSUPPORTED;
                p->ref_rng_live = true;
            } else {
                sbr::launch_score(m->mv, mv, bv, p->wb.v, epoch_key, mb.R, m->stream, small_tail ? &tail : nullptr);
            }
        }
    }
    /* the figure the reference's fit returns (sbr_report.hip): a small step folds it into the header launch; otherwise the
     * per-sequence sums come from a parallel kernel here and the sequential chain over the sequences runs as one wave on the
     * sorter stream (queued at the end of this call), off the critical path */
    /* single device: the loss accumulators take the block's header in the header kernel itself (one launch fewer per step) */
    p->header_accumulated = p->ndev == 1;
    const bool fuse_lag = !overlap && mb.B <= SBR_HEADER_LAG_MAX_B;
    if (small_tail) {
        /* done by the score launch */
    } else if (side_header) {
        HIPCHK(hipEventRecord(m->ev_scored, m->stream)); /* right behind the score kernel */
    } else {
        if (p->lag_busy) { /* the previous step's chain may still be running on the sorter stream: it owns lag_state / lag_seqsum */
            HIPCHK(hipStreamWaitEvent(m->stream, p->ev_lagged, 0));
            p->lag_busy = false;
        }
        if (!fuse_lag) {
            sbr::launch_seq_loss(mv, p->wb.v.loss, p->lag_seqsum, p->lag_state, mb.B, m->stream);
            if (overlap) HIPCHK(hipEventRecord(p->ev_seqsum, m->stream));
        }
        sbr::launch_block_header(m->mv, bv, p->wb.v, mv, mb.R, p->header_accumulated ? p->loss_acc : nullptr,
                                 p->header_accumulated ? p->ex_acc : nullptr, fuse_lag ? p->lag_state : nullptr, m->stream);
    }
    /* host order: with the ordering on its own stream the backward pass is queued FIRST — the ordering's up to nine short
     * launches would otherwise sit in the host's queue ahead of it (50 us at a few hundred sequences per step, as long as
     * the pass itself); ev_scored, recorded here, is what the ordering waits for either way */
    const bool sort_first = place == SORT_PRE;
    if (!early_sort && sort_first && !small_tail) SBRCHK(launch_sort(sorter));
    if (!early_sort && place == SORT_OWN_STREAM && !side_header) HIPCHK(hipEventRecord(m->ev_scored, m->stream));
    if (!ewma_fused) {
        ScopedTimer t(m, SBR_K_RECURRENT_BWD, m->ng && m->d > 128 ? 2 * (uint64_t)mb.Tm : 1);
        sbr::launch_recurrent_backward(m->mv, mv, bv, p->wb.v, mb.Tm, mb.R, mb.B, off_host, m->stream);
    }
    if (!early_sort && place == SORT_OWN_STREAM) SBRCHK(launch_sort(sorter));
    /* the dense-gradient GEMM (MFMA-bound, reads dZ / X / H only) goes to the side stream so that the
     * HBM-bound sparse update that follows on the main stream overlaps it; joined in step_apply /
     * step_dense before anything reads blk.dense */
    if (side != m->stream) {
        HIPCHK(hipEventRecord(m->ev_fork, m->stream));
        HIPCHK(hipStreamWaitEvent(side, m->ev_fork, 0));
    }
    p->dw_deferred = m->step_fusion >= 1 && p->fuse_back && !overlap && p->ndev == 1 && sbr::small_back_shape_ok(m->mv, (int)mb.R) && (m->ng || mb.B <= 256);
    if (p->dw_deferred) p->dense_unreduced_chunks = 0;
    if (!p->dw_deferred) {
        /* one device: the ordered reduction of the chunk partials is left to the consumer — the optimiser step folds it into
         * the dense update's launch (sbr_fit_step_apply); the exchange halves and the debug fetch reduce on demand */
        ScopedTimer t(m, SBR_K_DENSE_GRAD, 1, side);
        p->dense_unreduced_chunks = sbr::launch_dense_gradient(m->mv, mv, bv, p->wb.v, mb.R, mb.B, side, /*defer_reduce=*/p->ndev == 1);
    }
    if (side != m->stream) HIPCHK(hipEventRecord(m->ev_join, side));
    p->dense_pending = side != m->stream;
    if (!fuse_lag) {
        hipStream_t ls = overlap ? m->sorter : m->stream;
        if (ls != m->stream && !side_header) HIPCHK(hipStreamWaitEvent(ls, p->ev_seqsum, 0)); /* (side_header: same stream as seq_loss) */
        sbr::launch_lagged_chain(mv, p->lag_seqsum, mb.B, p->lag_state, ls);
        if (ls != m->stream) {
            HIPCHK(hipEventRecord(p->ev_lagged, ls));
            p->lag_busy = true;
        }
    }
    p->last_R = mb.R;
    p->last_block = block;
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

/* single device: optimiser step straight from the local block */
sbr_status sbr_fit_step_apply(sbr_fit_plan* p, uint64_t minibatch) {
    if (!p || minibatch >= p->ep[p->cur].num_mb || p->ndev != 1) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    const uint8_t* all = p->block;
    begin_optimizer_step(m);
    if (!p->header_accumulated) sbr::launch_accumulate_loss(all, p->block_bytes, 1, p->loss_acc, p->ex_acc, m->stream);
    p->header_accumulated = false;
    if (p->dw_deferred) { /* small LSTM step: dense gradient + dense update + sparse update in one launch */
        p->dw_deferred = false;
        ScopedTimer t(m, SBR_K_SPARSE_UPDATE, 1);
        if (p->sort_off_stream) HIPCHK(hipStreamWaitEvent(m->stream, m->ev_sorted, 0));
        sbr::launch_small_back(m->mv, mb_view(p, minibatch), block_view(m, p->block, p->rmax), p->wb.v, p->ep[p->cur].rows_of_dev[minibatch],
                               p->keys_sorted, p->seg, m->stream);
        HIPCHK(hipGetLastError());
        return SBR_OK;
    }
    {
        ScopedTimer t(m, SBR_K_SPARSE_UPDATE, 1);
        if (p->sort_off_stream) HIPCHK(hipStreamWaitEvent(m->stream, m->ev_sorted, 0));
        sbr::SegScratch sc = p->seg;
        sc.prelisted = p->hot_prelisted ? 1u : 0u;
        const sbr::BlockView bv = block_view(m, p->block, p->rmax);
        sbr::launch_seg_apply(m->mv, bv, p->ep[p->cur].rows_of_dev[minibatch], p->keys_sorted, sc, m->stream);
        if (p->hot_prelisted) { /* the listed hot rows: chunk partials + ordered finish on the ordering's stream (behind the list, and
                                 * behind BPTT: ev_fork), beside the short segments' pass — disjoint table rows */
            p->hot_prelisted = false;
            HIPCHK(hipStreamWaitEvent(m->sorter, m->ev_fork, 0));
            sbr::launch_seg_hot_apply(m->mv, bv, p->keys_sorted, sc, m->sorter);
            HIPCHK(hipEventRecord(p->ev_hot, m->sorter));
            HIPCHK(hipStreamWaitEvent(m->stream, p->ev_hot, 0));
        }
    }
    SBRCHK(join_dense(p));
    {
        ScopedTimer t(m, SBR_K_DENSE_UPDATE, 1);
        if (p->dense_unreduced_chunks > 0) { /* ordered reduction of the chunk partials + dense update in ONE launch */
            sbr::launch_dense_reduce_apply(m->mv, p->wb.v, p->dense_unreduced_chunks, block_view(m, p->block, p->rmax), m->stream);
            p->dense_unreduced_chunks = 0;
        } else {
            sbr::launch_dense_apply(m->mv, all, p->block_bytes, dense_offset_bytes(m, p->rmax), 1, m->stream);
        }
    }
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

sbr_status sbr_fit_step(sbr_fit_plan* p, uint64_t minibatch) {
    if (!p) return SBR_ERR_INVALID_ARGUMENT;
    if (p->ndev != 1) return SBR_ERR_INVALID_ARGUMENT; /* multi-device: the owner-reduce halves below, driven by the host */
    p->fuse_back = true; /* nobody looks at the block between the two halves */
    const sbr_status st = sbr_fit_step_local(p, minibatch);
    p->fuse_back = false;
    SBRCHK(st);
    return sbr_fit_step_apply(p, minibatch);
}

/* `count` consecutive optimiser steps starting at `first`.  One sequence per step at d <= 32 (the reference's own schedule,
 * sequence_model.rs:111-169) with EWMA, a single-negative loss and Adagrad: runs of up to SBR_EPOCH_STEPS_PER_LAUNCH steps are ONE
 * launch each (sbr::launch_epoch_steps) — same bits as `count` calls of sbr_fit_step, which is what every other shape gets. */
#ifndef SBR_EPOCH_STEPS_PER_LAUNCH
#define SBR_EPOCH_STEPS_PER_LAUNCH 8192 /* a launch stays well under a second (a one-sequence step is 5-100 us) */
#endif
sbr_status sbr_fit_steps(sbr_fit_plan* p, uint64_t first, uint64_t count) {
    if (!p || p->ndev != 1) return SBR_ERR_INVALID_ARGUMENT;
    sbr_fit_plan::Epoch& ep = p->ep[p->cur];
    if (first > ep.num_mb || count > ep.num_mb - first) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    const bool runs_ok = m->step_fusion >= 2 && !m->timing && !m->reference_order && p->bmax == 1 && ep.d_desc && ep.desc_host.size() == ep.num_mb;
    const bool ewma_runs = runs_ok && sbr::epoch_steps_shape_ok(m->mv, p->T - 1);
    const bool lstm_runs = runs_ok && !ewma_runs && sbr::lstm_steps_shape_ok(m->mv, p->T - 1);
    if (!ewma_runs && !lstm_runs) {
        for (uint64_t mb = first; mb < first + count; ++mb) SBRCHK(sbr_fit_step(p, mb));
        return SBR_OK;
    }
    if (!count) return SBR_OK;
    SBRCHK(ensure_device(m));
    const sbr::BlockView bv = block_view(m, p->block, p->rmax);
    const sbr::EpochView ev{ep.dp.off, ep.dp.steps, ep.dp.prev_row, ep.dp.in_idx, ep.dp.out_idx, ep.dp.ctr, ep.d_desc};
    const sbr::SmallTail tail{bv.header, p->loss_acc, p->ex_acc, p->lag_state, p->keys_sorted, p->seg.head_pos, p->seg.nheads};
    const uint64_t epoch_key = sbr_epoch_key(p->fit_seed[p->rank], ep.epoch_key_epoch);
    const uint32_t row_cap = ewma_runs ? (uint32_t)(p->T - 1) : (uint32_t)sbr::lstm_steps_max_rows();
    /* maximal runs of consecutive steps the one-launch form takes (the LSTM form: steps of at most row_cap rows); a longer step goes
     * through the separate launches and the next run starts behind it */
    uint64_t b = first;
    const uint64_t end = first + count;
    while (b < end) {
        if (ep.desc_host[b].rows > row_cap) {
            SBRCHK(sbr_fit_step(p, b));
            ++b;
            continue;
        }
        uint64_t e = b;
        uint32_t run_max = 0;
        while (e < end && e - b < SBR_EPOCH_STEPS_PER_LAUNCH && ep.desc_host[e].rows <= row_cap) {
            run_max = std::max(run_max, ep.desc_host[e].rows);
            ++e;
        }
        if (p->lag_busy) { /* an earlier step's chain on the ordering's stream still owns lag_state */
            HIPCHK(hipStreamWaitEvent(m->stream, p->ev_lagged, 0));
            p->lag_busy = false;
        }
        if (p->sorted_event_live) HIPCHK(hipStreamWaitEvent(m->stream, m->ev_sorted, 0)); /* (a larger step before: its ordering owned the keys) */
        SBRCHK(join_dense(p));
        if (ewma_runs) sbr::launch_epoch_steps(m->mv, ev, bv, p->wb.v, epoch_key, tail, (int)b, (int)e, p->T - 1, p->phase_clocks, m->stream);
        else sbr::launch_lstm_steps(m->mv, ev, bv, p->wb.v, epoch_key, tail, (int)b, (int)e, p->T - 1, (int)run_max, p->phase_clocks, m->stream);
        m->opt_steps += e - b; /* Adagrad: no per-step host-side corrections */
        p->hot_prelisted = p->sort_off_stream = p->dw_deferred = p->header_accumulated = false;
        p->dense_unreduced_chunks = 0;
        p->last_R = ep.mbs[e - 1].R;
        p->last_block = p->block;
        b = e;
    }
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

sbr_status sbr_fit_debug_phase_clocks(sbr_fit_plan* p, uint64_t out[6]) {
    if (!p || !out) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(p->m));
    HIPCHK(hipStreamSynchronize(p->m->stream));
    HIPCHK(hipMemcpy(out, p->phase_clocks, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return SBR_OK;
}

/* Reference order across devices (sequence_model.rs:163-166; wyrm's SynchronizedOptimizer as recalled): the workers rendezvous, then
 * every worker's gradient goes in as its OWN optimiser step, one after the other in worker order — n Adagrad applications per step
 * (G += g_q^2 each) where the contract adds the devices' gradients and applies one.  device_blocks: the n devices' local blocks of
 * this step, gathered (block q at q * block bytes); every replica applies the same sequence and stays bit-identical. */
sbr_status sbr_fit_step_apply_blocks_in_order(sbr_fit_plan* p, uint64_t minibatch, const void* device_blocks) {
    if (!p || !device_blocks || minibatch >= p->ep[p->cur].num_mb || !p->m->reference_order) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    const uint8_t* all = reinterpret_cast<const uint8_t*>(device_blocks);
    SBRCHK(join_dense(p));
    sbr::launch_accumulate_loss(all, p->block_bytes, p->ndev, p->loss_acc, p->ex_acc, m->stream);
    p->header_accumulated = false;
    for (int q = 0; q < p->ndev; ++q) {
        const uint32_t R = p->ep[p->cur].rows_of_dev[minibatch * p->ndev + q];
        uint8_t* base = const_cast<uint8_t*>(all) + (size_t)q * p->block_bytes;
        const sbr::BlockView bv = block_view(m, base, p->rmax);
        begin_optimizer_step(m);
        sbr::launch_own_sort(bv, R, p->keys, p->keys_sorted, p->sort_temp, p->sort_temp_bytes, p->key_bits, p->seg, m->stream, nullptr, 0, 0);
        sbr::SegScratch sc = p->seg;
        sc.prelisted = 0u;
        sbr::launch_seg_apply(m->mv, bv, R, p->keys_sorted, sc, m->stream);
        sbr::launch_dense_apply(m->mv, base, p->block_bytes, dense_offset_bytes(m, p->rmax), 1, m->stream);
    }
    p->sort_off_stream = false;
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

sbr_status sbr_fit_block_bytes(const sbr_fit_plan* p, uint64_t* out_bytes) {
    if (!p || !out_bytes) return SBR_ERR_INVALID_ARGUMENT;
    *out_bytes = p->block_bytes;
    return SBR_OK;
}

sbr_status sbr_model_set_reference_order(sbr_model* m, int32_t on) {
    if (!m) return SBR_ERR_INVALID_ARGUMENT;
    if (on && (m->hp.batch_sequences != 1 || !sbr::reference_order_shape_ok(m->d, (int)m->hp.max_sequence_length - 1) ||
               (m->hp.num_devices != 1 && (m->hp.parallelism != SBR_PAR_SYNCHRONOUS || m->shared))))
        return SBR_ERR_UNSUPPORTED; /* one sequence per step whose h rows + candidate window fit one workgroup's LDS (255 rows at d <= 64, 220 at
                                     * d = 128, 80 at d = 256); several workers: Synchronous, replicated */
    m->reference_order = on != 0;
    return SBR_OK;
}

sbr_status sbr_model_set_step_fusion(sbr_model* m, int32_t level) {
    if (!m || level < 0 || level > 2) return SBR_ERR_INVALID_ARGUMENT;
    m->step_fusion = level;
    return SBR_OK;
}

/* ---- multi-device owner-reduce protocol ------------------------------------------ */
static uint64_t slice_rows(const sbr_fit_plan* p) { return ((uint64_t)p->m->hp.num_items + p->ndev - 1) / p->ndev; }

sbr_status sbr_fit_chunk_bytes(const sbr_fit_plan* p, uint64_t* out_bytes) {
    if (!p || !out_bytes) return SBR_ERR_INVALID_ARGUMENT;
    *out_bytes = slice_rows(p) * ((uint64_t)p->m->d + 2) * 4;
    return SBR_OK;
}

sbr_status sbr_fit_dense_bytes(const sbr_fit_plan* p, uint64_t* out_bytes) {
    if (!p || !out_bytes) return SBR_ERR_INVALID_ARGUMENT;
    *out_bytes = (8 + dense_count(p->m)) * 4;
    return SBR_OK;
}

sbr_status sbr_fit_step_scatter(sbr_fit_plan* p, uint64_t minibatch, void* device_send) {
    if (!p || !device_send || minibatch >= p->ep[p->cur].num_mb) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    /* reference order with several workers is n optimiser applications per step in worker order plus an exchange of the workers'
     * generator states per epoch: only the single-process group driver sequences that (sbr_fit_step_apply_blocks_in_order).  The
     * summed-gradient exchange would silently give neither order. */
    if (m->reference_order && p->ndev > 1) return SBR_ERR_UNSUPPORTED;
    SBRCHK(ensure_device(m));
    const sbr::BlockView bv = block_view(m, p->block, p->rmax);
    const uint32_t R = p->ep[p->cur].rows_of_dev[minibatch * p->ndev + p->rank];
    {
        ScopedTimer t(m, SBR_K_SPARSE_UPDATE, 1);
        if (p->sorted_event_live) HIPCHK(hipStreamWaitEvent(m->stream, m->ev_sorted, 0));
        sbr::launch_seg_scatter(m->mv, bv, R, p->ndev, slice_rows(p), device_send, p->keys_sorted, p->seg, m->stream);
    }
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

/* small dense block = [8-word header | dense grads]; joins the side stream's dense-gradient GEMM, so
 * calling it late (after the chunk all-to-all has been queued) lets that GEMM overlap the transfer */
sbr_status sbr_fit_step_dense(sbr_fit_plan* p, void* device_dense_out) {
    if (!p || !device_dense_out) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    const sbr::BlockView bv = block_view(m, p->block, p->rmax);
    SBRCHK(ensure_dense_reduced(p));
    /* a WARP step with the ordering on its own stream writes the block header THERE (side_header in sbr_fit_step_local): join
     * it here rather than rely on the caller having run sbr_fit_step_scatter first */
    if (p->sorted_event_live && p->sort_off_stream) HIPCHK(hipStreamWaitEvent(m->stream, m->ev_sorted, 0));
    HIPCHK(hipMemcpyAsync(device_dense_out, bv.header, 32, hipMemcpyDeviceToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(reinterpret_cast<uint8_t*>(device_dense_out) + 32, bv.dense, dense_count(m) * 4,
                          hipMemcpyDeviceToDevice, m->stream));
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

static sbr::ChunkPtrs contiguous_chunks(const sbr_fit_plan* p, const void* base) {
    sbr::ChunkPtrs c;
    const uint64_t chunk = slice_rows(p) * ((uint64_t)p->m->d + 2) * 4;
    for (int q = 0; q < 16; ++q) c.p[q] = q < p->ndev ? reinterpret_cast<const uint8_t*>(base) + (size_t)q * chunk : nullptr;
    return c;
}

sbr_status sbr_fit_step_owner_reduce(sbr_fit_plan* p, const void* device_recv, void* device_own_chunk) {
    if (!p || !device_recv || !device_own_chunk) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    {
        ScopedTimer t(m, SBR_K_SPARSE_UPDATE, 1);
        sbr::launch_owner_reduce(m->mv, contiguous_chunks(p, device_recv), p->ndev, slice_rows(p), device_own_chunk, m->stream);
    }
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

sbr_status sbr_fit_step_owner_reduce_on(sbr_fit_plan* p, const void* device_recv, void* device_own_chunk, void* hip_stream) {
    if (!p || !device_recv || !device_own_chunk) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    {
        ScopedTimer t(m, SBR_K_SPARSE_UPDATE, 1, st);
        sbr::launch_owner_reduce(m->mv, contiguous_chunks(p, device_recv), p->ndev, slice_rows(p), device_own_chunk, st);
    }
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

/* the part of an exchanged step that every device applies identically: step counter, loss header,
 * dense parameters (device-order sum of the gathered dense blocks) */
static sbr_status apply_dense_blocks(sbr_fit_plan* p, const void* device_dense_all, bool begins_step = true) {
    sbr_model* m = p->m;
    const uint64_t db = (8 + dense_count(m)) * 4;
    const uint8_t* dall = reinterpret_cast<const uint8_t*>(device_dense_all);
    if (begins_step) begin_optimizer_step(m);
    /* one device driven through the exchange halves (bench.py --force-exchange): step_local's header launch has already added
     * this step to the plan's loss accumulators */
    if (!(p->ndev == 1 && p->header_accumulated)) sbr::launch_accumulate_loss(dall, db, p->ndev, p->loss_acc, p->ex_acc, m->stream);
    p->header_accumulated = false;
    {
        ScopedTimer t(m, SBR_K_DENSE_UPDATE, 1);
        sbr::launch_dense_apply(m->mv, dall, db, 32, p->ndev, m->stream);
    }
    return SBR_OK;
}

sbr_status sbr_fit_step_apply_table(sbr_fit_plan* p, const void* device_table, const void* device_dense_all) {
    if (!p || !device_table || !device_dense_all) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    if (m->shared) return SBR_ERR_INVALID_ARGUMENT; /* a partitioned table is updated by its owners (sbr_group_fit) */
    if (m->opt_state_partial) return SBR_ERR_INVALID_ARGUMENT; /* this replica holds its own rows' optimiser state only: gather it first */
    SBRCHK(ensure_device(m));
    SBRCHK(apply_dense_blocks(p, device_dense_all));
    {
        ScopedTimer t(m, SBR_K_SPARSE_UPDATE, 1);
        sbr::launch_table_apply(m->mv, contiguous_chunks(p, device_table), slice_rows(p), m->stream);
    }
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

/* sbr_fit_step_apply_table in two halves, so that the item-table half does not have to wait for the dense-gradient
 * GEMM: rows first (it opens the optimiser step), dense second, once each per step. */
sbr_status sbr_fit_step_apply_rows(sbr_fit_plan* p, const void* device_table) {
    if (!p || !device_table) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    if (m->shared || m->opt_state_partial) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(m));
    begin_optimizer_step(m);
    {
        ScopedTimer t(m, SBR_K_SPARSE_UPDATE, 1);
        sbr::launch_table_apply(m->mv, contiguous_chunks(p, device_table), slice_rows(p), m->stream);
    }
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

sbr_status sbr_fit_step_apply_dense(sbr_fit_plan* p, const void* device_dense_all) {
    if (!p || !device_dense_all) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    if (m->shared) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(m));
    SBRCHK(apply_dense_blocks(p, device_dense_all, /*begins_step=*/false));
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

/* ---- owner-APPLIED update of the replicated exchange (Parallelism::Synchronous) ------------------------------------------
 * ≙ the one shared parameter + optimiser state of lstm.rs:259-260 / ewma.rs:267-269 under the synchronised step of
 * sequence_model.rs:163-169.  After the all-to-all the owner of a slice adds the devices' contributions in device order AND applies
 * the one optimiser update of every touched row of its slice, in place in its own replica; what the devices then all-gather is the
 * updated PARAMETER slices (E rows and biases), written straight into every replica's table (sbr_model_table_slice: the arrays are
 * allocated for num_devices equal slices).  Same sums in the same order, same update arithmetic, same bytes on the links as the
 * gradient all-gather of sbr_fit_step_owner_reduce + sbr_fit_step_apply_rows — same bits — but no replica walks the whole table
 * any more and the optimiser state of a row is maintained by its owner alone. */
sbr_status sbr_fit_step_owner_update(sbr_fit_plan* p, const void* device_recv) {
    if (!p || !device_recv) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    if (m->shared) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(m));
    begin_optimizer_step(m);
    const uint64_t S = slice_rows(p), I = m->hp.num_items;
    const uint64_t row0 = std::min<uint64_t>(I, (uint64_t)p->rank * S), nrows = std::min<uint64_t>(I, row0 + S) - row0;
    {
        ScopedTimer t(m, SBR_K_SPARSE_UPDATE, 1);
        sbr::launch_owner_update(m->mv, contiguous_chunks(p, device_recv), p->ndev, S, row0, nrows, m->stream);
    }
    if (p->ndev > 1) m->opt_state_partial = true;
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

/* the owner update reading chunk `rank` of every peer's send buffer in place (peer transport) is sbr_fit_step_owner_reduce_peers'
 * sibling; not built: the peers' parameter slices would have to be exportable allocations as well */

sbr_status sbr_model_table_slice(sbr_model* m, int32_t which, void** out_base, uint64_t* out_slice_bytes) {
    if (!m || !out_base || !out_slice_bytes || m->shared) return SBR_ERR_INVALID_ARGUMENT;
    const uint64_t n = m->hp.num_devices, S = ((uint64_t)m->hp.num_items + n - 1) / n, d = (uint64_t)m->d;
    const sbr::ModelView& v = m->mv;
    float* base = nullptr;
    uint64_t row_bytes = 0;
    switch (which) {
        case SBR_PARAM_ITEM_EMBEDDING: base = v.E; row_bytes = d * 4; break;
        case SBR_PARAM_ITEM_EMBEDDING_ACC: base = v.Eacc; row_bytes = d * 4; break;
        case SBR_PARAM_ITEM_EMBEDDING_M: base = v.Em; row_bytes = d * 4; break;
        case SBR_PARAM_ITEM_BIAS: base = v.b; row_bytes = 4; break;
        case SBR_PARAM_ITEM_BIAS_ACC: base = v.bacc; row_bytes = 4; break;
        case SBR_PARAM_ITEM_BIAS_M: base = v.bm; row_bytes = 4; break;
        default: return SBR_ERR_INVALID_ARGUMENT;
    }
    *out_base = base; /* null: the block does not exist (Adam moments under Adagrad) */
    *out_slice_bytes = base ? S * row_bytes : 0;
    return SBR_OK;
}

sbr_status sbr_model_optimizer_state_gathered(sbr_model* m) {
    if (!m) return SBR_ERR_INVALID_ARGUMENT;
    m->opt_state_partial = false;
    return SBR_OK;
}

sbr_status sbr_model_optimizer_state_is_partial(const sbr_model* m, int32_t* out) {
    if (!m || !out) return SBR_ERR_INVALID_ARGUMENT;
    *out = m->opt_state_partial ? 1 : 0;
    return SBR_OK;
}

/* ---- partitioned item table (sbr_group_create flag SBR_GROUP_PARTITION_ITEM_TABLE) ------------------ */
static sbr_status partition_buffers(sbr_fit_plan* p) {
    if (p->glist) return SBR_OK;
    const uint64_t cap = 3 * p->rmax;
    if (cap >= (1ull << 28)) return SBR_ERR_UNSUPPORTED; /* list positions are 28-bit in the merge keys */
    if (p->exportable_lists) {
        const int dev = p->m->device;
        SBRCHK(p->own_x[1].alloc(cap * (uint64_t)p->m->d * 4, dev));
        SBRCHK(p->own_x[2].alloc(cap * 4, dev));
        SBRCHK(p->own_x[3].alloc(cap * 4, dev));
        p->glist = reinterpret_cast<float*>(p->own_x[1].ptr);
        p->gblist = reinterpret_cast<float*>(p->own_x[2].ptr);
        p->gfl = reinterpret_cast<uint32_t*>(p->own_x[3].ptr);
    } else {
        SBRCHK(dmalloc(&p->glist, cap * (uint64_t)p->m->d));
        SBRCHK(dmalloc(&p->gblist, cap));
        SBRCHK(dmalloc(&p->gfl, cap));
    }
    SBRCHK(dmalloc(&p->bounds_dev, 17));
    return SBR_OK;
}

/* The owner's merge buffers.  Since round 6 the number of merge keys of a step — the key positions of all devices inside this
 * owner's row range — is known on the device only (the host no longer reads the owner bounds back: that read drained every queue
 * every step), so the buffers hold the worst case, every entry of every device: ndev x 3 x rmax keys (2 x 8 B each + the radix
 * counters: 0.2 GB per owner at configs[4]'s 8 x 8 192 sequences of <= 63 steps).  Allocated at the first partitioned step. */
static sbr_status merge_capacity(sbr_fit_plan* p) {
    if (p->mcap) return SBR_OK;
    const uint64_t cap = (uint64_t)p->ndev * 3ull * p->rmax;
    if (cap >= (1ull << 32)) return SBR_ERR_UNSUPPORTED;
    SBRCHK(dmalloc(&p->mkeys, cap));
    SBRCHK(dmalloc(&p->mkeys_sorted, cap));
    p->msort_temp_bytes = sbr::sparse_sort_temp_bytes(cap, 64);
    uint8_t* tmp = nullptr;
    SBRCHK(dmalloc(&tmp, p->msort_temp_bytes));
    p->msort_temp = tmp;
    SBRCHK(dmalloc(&p->mplan, 1));
    p->mcap = cap;
    return SBR_OK;
}

/* this device's own entries reduced per row into its list + the owner bounds (after sbr_fit_step_local) */
static sbr_status partition_reduce_own(sbr_fit_plan* p, uint64_t minibatch) {
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    SBRCHK(partition_buffers(p));
    const sbr::BlockView bv = block_view(m, p->block, p->rmax);
    const uint32_t R = p->ep[p->cur].rows_of_dev[minibatch * p->ndev + p->rank];
    if (p->sorted_event_live) HIPCHK(hipStreamWaitEvent(m->stream, m->ev_sorted, 0));
    {
        ScopedTimer t(m, SBR_K_SPARSE_UPDATE, 1);
        sbr::launch_seg_list(m->mv, bv, R, p->ndev, slice_rows(p), p->keys_sorted, p->glist, p->gblist, p->gfl, p->bounds_dev, p->seg,
                             m->stream);
    }
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

/* owner side: merge the peers' lists over this device's row range (device order) and update its rows; the range's position in the
 * peers' lists comes from their owner bounds ON THE DEVICE (pb: peer-readable arrays, or one gathered array) */
static sbr_status partition_owner_apply(sbr_fit_plan* p, const sbr::PeerLists& pl, const sbr::PeerBounds& pb) {
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    SBRCHK(merge_capacity(p));
    {
        ScopedTimer t(m, SBR_K_SPARSE_UPDATE, 1);
        sbr::launch_owner_list_apply(m->mv, pl, pb, p->ndev, p->rank, p->mplan, (uint32_t)p->mcap, p->mkeys, p->mkeys_sorted, p->msort_temp,
                                     p->msort_temp_bytes, m->stream);
    }
    HIPCHK(hipGetLastError());
    return SBR_OK;
}

sbr_status sbr_fit_end(sbr_fit_plan* p, float* out_loss, uint64_t* out_examples) {
    if (!p) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipStreamSynchronize(m->sorter)); /* a step's header launch may ride on the ordering's stream */
    double loss[17];
    unsigned long long ex[18];
    HIPCHK(hipMemcpy(loss, p->loss_acc, sizeof(loss), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ex, p->ex_acc, sizeof(ex), hipMemcpyDeviceToHost));
    /* ≙ the sum over the partitions' workers of loss_value / (1.0 + examples) (sequence_model.rs:173-177); the
     * true loss sums, not the stale node values the reference reads */
    double total = 0.0;
    for (int q = 0; q < p->ndev; ++q) total += loss[1 + q] / (1.0 + (double)ex[2 + q]);
    if (out_loss) *out_loss = (float)total;
    if (out_examples) *out_examples = (uint64_t)ex[0];
    return SBR_OK;
}

/* ≙ the value `fit` returns in the reference: sum over the workers of (stale loss-node values) / (1 + examples)
 * (sequence_model.rs:157 before :160, :173-177).  This device's term; single device: the whole figure. */
sbr_status sbr_fit_end_lagged(sbr_fit_plan* p, float* out_term) {
    if (!p || !out_term) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipStreamSynchronize(m->sorter));
    float lagged = 0.0f;
    unsigned long long ex[18];
    HIPCHK(hipMemcpy(&lagged, p->lag_state, sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ex, p->ex_acc, sizeof(ex), hipMemcpyDeviceToHost));
    *out_term = lagged / (1.0f + (float)ex[2 + p->rank]);
    return SBR_OK;
}

sbr_status sbr_model_last_fit_lagged_loss(const sbr_model* m, float* out_loss) {
    if (!m || !out_loss) return SBR_ERR_INVALID_ARGUMENT;
    *out_loss = m->last_lagged_loss;
    return SBR_OK;
}

sbr_status sbr_fit_counters(sbr_fit_plan* p, uint64_t* out_examples, uint64_t* out_negatives_scored) {
    if (!p) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(p->m));
    HIPCHK(hipStreamSynchronize(p->m->stream));
    HIPCHK(hipStreamSynchronize(p->m->sorter));
    unsigned long long v[2] = {0, 0};
    HIPCHK(hipMemcpy(v, p->ex_acc, sizeof(v), hipMemcpyDeviceToHost));
    if (out_examples) *out_examples = v[0];
    if (out_negatives_scored) *out_negatives_scored = v[1];
    return SBR_OK;
}

sbr_status sbr_fit_sparse_stats(sbr_fit_plan* p, uint64_t* out_entries, uint64_t* out_unique_rows) {
    if (!p) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(p->m));
    HIPCHK(hipStreamSynchronize(p->m->side));
    HIPCHK(hipStreamSynchronize(p->m->sorter));
    HIPCHK(hipStreamSynchronize(p->m->stream));
    uint32_t nheads = 0;
    HIPCHK(hipMemcpy(&nheads, p->seg.nheads, sizeof(nheads), hipMemcpyDeviceToHost));
    if (out_entries) *out_entries = 3ull * (uint64_t)p->last_R;
    if (out_unique_rows) *out_unique_rows = nheads;
    return SBR_OK;
}

sbr_status sbr_model_fit(sbr_model* m, const uint64_t* user_ptr, const uint32_t* item_ids, uint64_t num_users,
                         float* out_loss) {
    if (!m) return SBR_ERR_INVALID_ARGUMENT;
    if (m->hp.num_devices != 1) return SBR_ERR_INVALID_ARGUMENT; /* multi-device fits are driven by the host (distributed.py) */
    sbr_fit_plan* p = nullptr;
    SBRCHK(sbr_fit_begin(m, user_ptr, item_ids, num_users, &p));
    sbr_status st = SBR_OK;
    for (uint32_t e = 0; e < m->hp.num_epochs && st == SBR_OK; ++e) {
        uint64_t nmb = 0;
        st = sbr_fit_epoch_prepare(p, &nmb);
        if (st == SBR_OK && e + 1 < m->hp.num_epochs) st = sbr_fit_epoch_prefetch(p); /* host packs epoch e+1 while the GPU runs e */
        if (st == SBR_OK) st = sbr_fit_steps(p, 0, nmb);
    }
    if (st == SBR_OK) st = sbr_fit_end(p, out_loss, nullptr);
    if (st == SBR_OK) st = sbr_fit_end_lagged(p, &m->last_lagged_loss);
    sbr_fit_plan_destroy(p);
    return st;
}

/* ---- the rendezvous of a step through RCCL INSIDE the library (one process per GPU; xGMI within a node) --------------------
 * ≙ the synchronised optimiser step of sequence_model.rs:92, 163-166 across processes.  A C / Rust host that runs one process per
 * GPU needs no collective library of its own: rank 0 makes an id (sbr_comm_unique_id), the host hands its 128 bytes to the other
 * ranks through whatever channel it has (a file, a socket, MPI), every rank calls sbr_comm_create on its device, and a step is
 * sbr_fit_step_local + sbr_fit_step_exchange.  librccl is opened at run time (dlopen): the engine has no link-time dependency on
 * it and hosts that bring their own transport (torch.distributed, MPI) never load it. */
namespace {
struct RcclId128 { char b[128]; }; /* ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank */
struct RcclApi {
    typedef RcclId128 Id128;
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    bool ok = false;
};
RcclApi* rccl_api() {
    static RcclApi* api = [] {
        RcclApi* a = new RcclApi;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a->lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a->lib) break;
        }
        if (!a->lib) return a;
        auto sym = [&](const char* n) { return dlsym(a->lib, n); };
        a->GetUniqueId = reinterpret_cast<decltype(a->GetUniqueId)>(sym("ncclGetUniqueId"));
        a->CommInitRank = reinterpret_cast<decltype(a->CommInitRank)>(sym("ncclCommInitRank"));
        a->CommDestroy = reinterpret_cast<decltype(a->CommDestroy)>(sym("ncclCommDestroy"));
        a->GroupStart = reinterpret_cast<decltype(a->GroupStart)>(sym("ncclGroupStart"));
        a->GroupEnd = reinterpret_cast<decltype(a->GroupEnd)>(sym("ncclGroupEnd"));
        a->Send = reinterpret_cast<decltype(a->Send)>(sym("ncclSend"));
        a->Recv = reinterpret_cast<decltype(a->Recv)>(sym("ncclRecv"));
        a->AllGather = reinterpret_cast<decltype(a->AllGather)>(sym("ncclAllGather"));
        a->ok = a->GetUniqueId && a->CommInitRank && a->CommDestroy && a->GroupStart && a->GroupEnd && a->Send && a->Recv && a->AllGather;
        return a;
    }();
    return api;
}
constexpr int kNcclUint8 = 1; /* ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1) */
}  // namespace

struct sbr_comm {
    void* comm = nullptr;
    uint32_t world = 0, rank = 0;
    int device = 0;
    uint8_t *send = nullptr, *recv = nullptr, *dense = nullptr, *dense_all = nullptr;
    uint64_t chunk = 0, db = 0; /* sizes the buffers were made for */
};

sbr_status sbr_comm_unique_id(uint8_t out_id[128]) {
    if (!out_id) return SBR_ERR_INVALID_ARGUMENT;
    RcclApi& a = *rccl_api();
    if (!a.ok) return SBR_ERR_UNSUPPORTED; /* no librccl on this host */
    return a.GetUniqueId(out_id) == 0 ? SBR_OK : SBR_ERR_HIP;
}

sbr_status sbr_comm_create(const uint8_t id[128], uint32_t world, uint32_t rank, sbr_comm** out) {
    if (!id || !out || world == 0 || world > 16 || rank >= world) return SBR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    RcclApi& a = *rccl_api();
    if (!a.ok) return SBR_ERR_UNSUPPORTED;
    sbr_comm* c = new (std::nothrow) sbr_comm;
    if (!c) return SBR_ERR_OUT_OF_MEMORY;
    c->world = world; c->rank = rank;
    (void)hipGetDevice(&c->device);
    RcclApi::Id128 uid;
    std::memcpy(uid.b, id, 128);
    if (a.CommInitRank(&c->comm, (int)world, uid, (int)rank) != 0) { delete c; return SBR_ERR_HIP; }
    *out = c;
    return SBR_OK;
}

void sbr_comm_destroy(sbr_comm* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    if (c->comm) rccl_api()->CommDestroy(c->comm);
    dfree(c->send); dfree(c->recv); dfree(c->dense); dfree(c->dense_all);
    delete c;
}

/* the exchange and the update of one optimiser step (after sbr_fit_step_local): scatter -> all-to-all -> owner update (device-order
 * sum + the one optimiser update of the owner's rows, in place) -> all-gather of the updated parameter slices straight into this
 * replica's table; the dense block joins the dense-gradient GEMM late -> all-gather -> dense update.  Everything is queued on the
 * model's stream; the call does not block the host. */
static sbr_status comm_buffers(sbr_fit_plan* p, sbr_comm* c, uint64_t chunk, uint64_t db) {
    if (c->chunk == chunk && c->db == db && c->send) return SBR_OK;
    sbr_model* m = p->m;
    const uint32_t n = c->world;
    HIPCHK(hipStreamSynchronize(m->stream));
    dfree(c->send); dfree(c->recv); dfree(c->dense); dfree(c->dense_all);
    c->send = c->recv = c->dense = c->dense_all = nullptr;
    c->chunk = c->db = 0;
    /* all or nothing, BEFORE any collective of the step is queued: a rank that failed half-way through would leave its peers
     * waiting inside theirs */
    sbr_status st = dmalloc(&c->send, n * chunk);
    if (st == SBR_OK) st = dmalloc(&c->recv, n * chunk);
    if (st == SBR_OK) st = dmalloc(&c->dense, db);
    if (st == SBR_OK) st = dmalloc(&c->dense_all, n * db);
    if (st != SBR_OK) {
        dfree(c->send); dfree(c->recv); dfree(c->dense); dfree(c->dense_all);
        c->send = c->recv = c->dense = c->dense_all = nullptr;
        return st;
    }
    c->chunk = chunk; c->db = db;
    return SBR_OK;
}

/* in-place all-gather of one item-table block's owner slices (slice r of every replica <- rank r) */
static sbr_status comm_gather_block(sbr_model* m, sbr_comm* c, int32_t which) {
    RcclApi& a = *rccl_api();
    void* base = nullptr;
    uint64_t sb = 0;
    SBRCHK(sbr_model_table_slice(m, which, &base, &sb));
    if (!base) return SBR_OK;
    if (a.AllGather(reinterpret_cast<uint8_t*>(base) + (size_t)c->rank * sb, base, sb, kNcclUint8, c->comm, m->stream) != 0) return SBR_ERR_HIP;
    return SBR_OK;
}

sbr_status sbr_fit_step_exchange(sbr_fit_plan* p, uint64_t minibatch, sbr_comm* c) {
    if (!p || !c || (int)c->world != p->ndev || (int)c->rank != p->rank || minibatch >= p->ep[p->cur].num_mb || p->m->shared)
        return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    if (c->device != m->device) return SBR_ERR_INVALID_ARGUMENT; /* the communicator belongs to the device that was current at sbr_comm_create */
    SBRCHK(ensure_device(m));
    RcclApi& a = *rccl_api();
    uint64_t chunk = 0, db = 0;
    SBRCHK(sbr_fit_chunk_bytes(p, &chunk));
    SBRCHK(sbr_fit_dense_bytes(p, &db));
    const uint32_t n = c->world;
    SBRCHK(comm_buffers(p, c, chunk, db));
    hipStream_t st = m->stream;
    SBRCHK(sbr_fit_step_scatter(p, minibatch, c->send));
    if (a.GroupStart() != 0) return SBR_ERR_HIP; /* all-to-all: chunk q of this rank -> rank q */
    int bad = 0; /* the group is closed whatever happens inside it: an open group would swallow every later collective of this thread */
    for (uint32_t q = 0; q < n; ++q) {
        bad |= a.Send(c->send + (size_t)q * chunk, chunk, kNcclUint8, (int)q, c->comm, st);
        bad |= a.Recv(c->recv + (size_t)q * chunk, chunk, kNcclUint8, (int)q, c->comm, st);
    }
    bad |= a.GroupEnd();
    if (bad) return SBR_ERR_HIP;
    SBRCHK(sbr_fit_step_owner_update(p, c->recv));
    SBRCHK(comm_gather_block(m, c, SBR_PARAM_ITEM_EMBEDDING));
    SBRCHK(comm_gather_block(m, c, SBR_PARAM_ITEM_BIAS));
    SBRCHK(sbr_fit_step_dense(p, c->dense));
    if (a.AllGather(c->dense, c->dense_all, db, kNcclUint8, c->comm, st) != 0) return SBR_ERR_HIP;
    SBRCHK(sbr_fit_step_apply_dense(p, c->dense_all));
    return SBR_OK;
}

/* every replica's copy of the item table's optimiser state made complete again from the owners' slices (a fit through
 * sbr_fit_step_exchange leaves a row's state on its owner only) */
sbr_status sbr_comm_gather_optimizer_state(sbr_model* m, sbr_comm* c) {
    if (!m || !c || m->hp.num_devices != c->world || m->hp.device_rank != c->rank || c->device != m->device) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(m));
    for (int32_t which : {SBR_PARAM_ITEM_EMBEDDING_ACC, SBR_PARAM_ITEM_BIAS_ACC, SBR_PARAM_ITEM_EMBEDDING_M, SBR_PARAM_ITEM_BIAS_M})
        SBRCHK(comm_gather_block(m, c, which));
    HIPCHK(hipStreamSynchronize(m->stream));
    m->opt_state_partial = false;
    return SBR_OK;
}

/* the whole fit of THIS rank through the library's own transport: ≙ fit with num_threads(world) across processes */
sbr_status sbr_model_fit_comm(sbr_model* m, sbr_comm* c, const uint64_t* user_ptr, const uint32_t* item_ids, uint64_t num_users, float* out_loss) {
    if (!m || !c || m->hp.num_devices != c->world || m->hp.device_rank != c->rank) return SBR_ERR_INVALID_ARGUMENT;
    if (m->reference_order && c->world > 1) return SBR_ERR_UNSUPPORTED; /* several workers in reference order: sbr_group_fit only */
    sbr_fit_plan* p = nullptr;
    SBRCHK(sbr_fit_begin(m, user_ptr, item_ids, num_users, &p));
    sbr_status st = SBR_OK;
    for (uint32_t e = 0; e < m->hp.num_epochs && st == SBR_OK; ++e) {
        uint64_t nmb = 0;
        st = sbr_fit_epoch_prepare(p, &nmb);
        if (st == SBR_OK && e + 1 < m->hp.num_epochs) st = sbr_fit_epoch_prefetch(p);
        for (uint64_t mb = 0; mb < nmb && st == SBR_OK; ++mb) {
            st = sbr_fit_step_local(p, mb);
            if (st == SBR_OK) st = sbr_fit_step_exchange(p, mb, c);
        }
    }
    if (st == SBR_OK) st = sbr_comm_gather_optimizer_state(m, c);
    if (st == SBR_OK) st = sbr_fit_end(p, out_loss, nullptr);
    if (st == SBR_OK) st = sbr_fit_end_lagged(p, &m->last_lagged_loss); /* this rank's term; hosts add the ranks' terms in rank order */
    sbr_fit_plan_destroy(p);
    return st;
}

/* ---- peer transport of the replicated owner-reduce exchange (one process per GPU) ----------------------
 * Instead of moving the chunks with collectives, every rank exports its send buffer and its reduced own
 * chunk once (file descriptors, like the partitioned table); the owner-reduce kernel then reads chunk q of
 * every peer's send buffer IN PLACE and the table update reads every owner's reduced chunk in place — the
 * bytes cross xGMI exactly once, inside the kernels that consume them, and no bulk collective is involved.
 * The host supplies the ordering: [scatter] barrier [owner reduce] barrier + all-gather of the small dense
 * blocks [apply]; the next step's scatter is safe because every rank passed the following barrier. */
sbr_status sbr_fit_exchange_export(sbr_fit_plan* p, int32_t out_fds[2], uint64_t out_bytes[2]) {
    if (!p || !out_fds || !out_bytes || p->ndev < 1 || p->m->shared) return SBR_ERR_INVALID_ARGUMENT;
    /* the peer transport runs the synchronous step whatever hp.parallelism says (the staleness-one pipeline belongs to the
     * collective transport): Asynchronous here IS Synchronous — deterministic, and a valid outcome of Hogwild */
    SBRCHK(ensure_device(p->m));
    const uint64_t chunk = slice_rows(p) * ((uint64_t)p->m->d + 2) * 4;
    if (!p->xchg_own[0].ptr) {
        SBRCHK(p->xchg_own[0].alloc((size_t)p->ndev * chunk, p->m->device));
        SBRCHK(p->xchg_own[1].alloc((size_t)chunk, p->m->device));
    }
    for (int i = 0; i < 2; ++i) {
        int fd = -1;
        SBRCHK(p->xchg_own[i].export_fd(&fd));
        out_fds[i] = fd;
        out_bytes[i] = p->xchg_own[i].bytes;
    }
    return SBR_OK;
}

sbr_status sbr_fit_exchange_import(sbr_fit_plan* p, uint32_t peer_rank, const int32_t fds[2], const uint64_t bytes[2]) {
    if (!p || !fds || !bytes || (int)peer_rank >= p->ndev || (int)peer_rank == p->rank) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(p->m));
    if (p->xchg_peer.empty()) p->xchg_peer.resize(p->ndev);
    for (int i = 0; i < 2; ++i) SBRCHK(p->xchg_peer[peer_rank][i].import_fd(fds[i], bytes[i], p->m->device));
    return SBR_OK;
}

static sbr_status peer_chunks(const sbr_fit_plan* p, int which, sbr::ChunkPtrs* out) {
    const uint64_t chunk = slice_rows(p) * ((uint64_t)p->m->d + 2) * 4;
    for (int r = 0; r < 16; ++r) out->p[r] = nullptr;
    for (int r = 0; r < p->ndev; ++r) {
        const VmmBuf* b = r == p->rank ? &p->xchg_own[which] : (p->xchg_peer.empty() ? nullptr : &p->xchg_peer[r][which]);
        if (!b || !b->ptr) return SBR_ERR_INVALID_ARGUMENT; /* exchange buffers of rank r not mapped */
        /* which 0: chunk `rank` of r's send buffer (what r contributes to my rows); which 1: r's reduced own chunk */
        out->p[r] = reinterpret_cast<const uint8_t*>(b->ptr) + (which == 0 ? (size_t)p->rank * chunk : 0);
    }
    return SBR_OK;
}

sbr_status sbr_fit_step_scatter_shared(sbr_fit_plan* p, uint64_t minibatch) {
    if (!p || !p->xchg_own[0].ptr) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(sbr_fit_step_scatter(p, minibatch, p->xchg_own[0].ptr));
    HIPCHK(hipStreamSynchronize(p->m->stream));
    return SBR_OK;
}

sbr_status sbr_fit_step_owner_reduce_peers(sbr_fit_plan* p) {
    if (!p || !p->xchg_own[1].ptr) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    sbr::ChunkPtrs src;
    SBRCHK(peer_chunks(p, 0, &src));
    {
        ScopedTimer t(m, SBR_K_SPARSE_UPDATE, 1);
        sbr::launch_owner_reduce(m->mv, src, p->ndev, slice_rows(p), p->xchg_own[1].ptr, m->stream);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(m->stream));
    return SBR_OK;
}

sbr_status sbr_fit_step_apply_table_peers(sbr_fit_plan* p, const void* device_dense_all) {
    if (!p || !device_dense_all || !p->xchg_own[1].ptr) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    sbr::ChunkPtrs src;
    SBRCHK(peer_chunks(p, 1, &src));
    SBRCHK(apply_dense_blocks(p, device_dense_all));
    {
        ScopedTimer t(m, SBR_K_SPARSE_UPDATE, 1);
        sbr::launch_table_apply(m->mv, src, slice_rows(p), m->stream);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(m->stream));
    return SBR_OK;
}

/* ---- one process per GPU over a partitioned table: the step halves the host sequences -----------------
 *   sbr_fit_step_local; sbr_fit_step_reduce_own (list + dense block; returns the owner bounds on the host,
 *   stream drained: this process has finished READING the table)  -> host: all-gather of the bounds and of
 *   the dense blocks (a rendezvous: every process has finished reading)  -> sbr_fit_step_owner_apply (merge
 *   of the peers' lists over this rank's rows, in-place update, stream drained)  -> host: barrier. */
sbr_status sbr_fit_lists_export(sbr_fit_plan* p, int32_t out_fds[4], uint64_t out_bytes[4]) {
    if (!p || !out_fds || !out_bytes || !p->exportable_lists) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(p->m));
    SBRCHK(partition_buffers(p));
    for (int i = 0; i < 4; ++i) {
        int fd = -1;
        SBRCHK(p->own_x[i].export_fd(&fd));
        out_fds[i] = fd;
        out_bytes[i] = p->own_x[i].bytes;
    }
    return SBR_OK;
}

sbr_status sbr_fit_lists_import(sbr_fit_plan* p, uint32_t peer_rank, const int32_t fds[4], const uint64_t bytes[4]) {
    if (!p || !fds || !bytes || !p->exportable_lists || (int)peer_rank >= p->ndev || (int)peer_rank == p->rank)
        return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(p->m));
    if (p->peer_x.empty()) p->peer_x.resize(p->ndev);
    for (int i = 0; i < 4; ++i) SBRCHK(p->peer_x[peer_rank][i].import_fd(fds[i], bytes[i], p->m->device));
    return SBR_OK;
}

sbr_status sbr_fit_step_reduce_own(sbr_fit_plan* p, uint64_t minibatch, uint32_t* host_bounds, void* device_dense_out) {
    if (!p || !host_bounds || !device_dense_out || !p->m->shared || minibatch >= p->ep[p->cur].num_mb) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(partition_reduce_own(p, minibatch));
    SBRCHK(sbr_fit_step_dense(p, device_dense_out));
    HIPCHK(hipStreamSynchronize(p->m->stream));
    HIPCHK(hipMemcpy(host_bounds, p->bounds_dev, (p->ndev + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SBR_OK;
}

/* the same half WITHOUT the host in the loop: nothing is drained, the owner bounds stay on the device (*out_device_bounds:
 * num_devices + 1 u32 of this rank, for the host's DEVICE all-gather) */
sbr_status sbr_fit_step_reduce_own_queued(sbr_fit_plan* p, uint64_t minibatch, void** out_device_bounds, void* device_dense_out) {
    if (!p || !out_device_bounds || !device_dense_out || !p->m->shared || minibatch >= p->ep[p->cur].num_mb) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(partition_reduce_own(p, minibatch));
    SBRCHK(sbr_fit_step_dense(p, device_dense_out));
    *out_device_bounds = p->bounds_dev;
    return SBR_OK;
}

static sbr_status owner_apply_lists(sbr_fit_plan* p, sbr::PeerLists* pl) {
    const int n = p->ndev, q = p->rank;
    std::memset(pl, 0, sizeof(*pl));
    for (int r = 0; r < n; ++r) {
        if (r == q) {
            pl->keys[r] = p->keys_sorted; pl->G[r] = p->glist; pl->gb[r] = p->gblist; pl->fl[r] = p->gfl;
        } else {
            if (p->peer_x.empty() || !p->peer_x[r][0].ptr) return SBR_ERR_INVALID_ARGUMENT; /* lists of rank r not imported */
            pl->keys[r] = reinterpret_cast<const uint64_t*>(p->peer_x[r][0].ptr);
            pl->G[r] = reinterpret_cast<const float*>(p->peer_x[r][1].ptr);
            pl->gb[r] = reinterpret_cast<const float*>(p->peer_x[r][2].ptr);
            pl->fl[r] = reinterpret_cast<const uint32_t*>(p->peer_x[r][3].ptr);
        }
    }
    return SBR_OK;
}

/* device_all_bounds: the ranks' owner bounds gathered ON THE DEVICE, rank r's num_devices + 1 words at r * (num_devices + 1);
 * queued, nothing drained */
sbr_status sbr_fit_step_owner_apply_queued(sbr_fit_plan* p, const void* device_all_bounds, const void* device_dense_all) {
    if (!p || !device_all_bounds || !device_dense_all || !p->exportable_lists) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    sbr::PeerLists pl;
    SBRCHK(owner_apply_lists(p, &pl));
    sbr::PeerBounds pb;
    for (int r = 0; r < 16; ++r)
        pb.b[r] = r < p->ndev ? reinterpret_cast<const uint32_t*>(device_all_bounds) + (size_t)r * (p->ndev + 1) : nullptr;
    SBRCHK(apply_dense_blocks(p, device_dense_all));
    return partition_owner_apply(p, pl, pb);
}

sbr_status sbr_fit_step_owner_apply(sbr_fit_plan* p, const uint32_t* all_bounds, const void* device_dense_all) {
    if (!p || !all_bounds || !device_dense_all || !p->exportable_lists) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    const size_t words = (size_t)p->ndev * (p->ndev + 1);
    if (!p->all_bounds_dev) SBRCHK(dmalloc(&p->all_bounds_dev, 17 * 16));
    HIPCHK(hipMemcpyAsync(p->all_bounds_dev, all_bounds, words * sizeof(uint32_t), hipMemcpyHostToDevice, m->stream));
    SBRCHK(sbr_fit_step_owner_apply_queued(p, p->all_bounds_dev, device_dense_all));
    HIPCHK(hipStreamSynchronize(m->stream));
    return SBR_OK;
}

/* ---- single-process multi-device fit (≙ fit with num_threads(n) on one host, sequence_model.rs:90-102, 163-169) --------
 * n replicas, each on its own HIP device and stream; the owner-reduce exchange of sbr_fit_step_scatter / _owner_reduce /
 * _apply_table is carried by peer copies (xGMI when the devices are peers) ordered with events:
 *   scattered[r]  send_r of this step is complete
 *   reduced[p]    own_p and dense_p of this step are complete
 *   applied[q]    q has finished reading its peers' buffers and has applied the step
 * A step is a short list of PHASES; inside a phase device r's work depends on nothing another device queues in the same phase,
 * so the phases either run as loops of one host thread or — sbr_group_plan_set_host_threads — on one host thread per device
 * (the reference runs one rayon worker per partition, sequence_model.rs:100-102): a device's ~20 launches per step are then
 * queued beside the other devices' instead of behind them.  An event is recorded in the phase BEFORE the one that waits for
 * it, and phases are separated by a host barrier, so hipStreamWaitEvent never sees an event that has not been recorded yet. */
namespace {

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    std::this_thread::yield();
#endif
}

/* n - 1 helper threads (the caller is worker 0).  run(f) = f(r) on every worker, then a barrier.  The helpers spin for the next
 * phase for a short while (a step's phases follow each other within microseconds) and then sleep on a condition variable. */
struct PhaseWorkers {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<uint64_t> generation{0};
    std::atomic<uint32_t> pending{0};
    std::atomic<bool> quit{false};
    const std::function<sbr_status(uint32_t)>* job = nullptr;
    std::vector<sbr_status> status;

    explicit PhaseWorkers(uint32_t n) : status(n, SBR_OK) {
        for (uint32_t r = 1; r < n; ++r) threads.emplace_back([this, r] { loop(r); });
    }
    ~PhaseWorkers() {
        {
            std::lock_guard<std::mutex> g(mu);
            quit.store(true, std::memory_order_release);
            generation.fetch_add(1, std::memory_order_release);
        }
        cv.notify_all();
        for (auto& t : threads) t.join();
    }
    void loop(uint32_t r) {
        uint64_t seen = 0;
        for (;;) {
            int spins = 0;
            while (generation.load(std::memory_order_acquire) == seen) {
                if (++spins < 20000) { cpu_relax(); continue; }
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return generation.load(std::memory_order_acquire) != seen; });
            }
            seen = generation.load(std::memory_order_acquire);
            if (quit.load(std::memory_order_acquire)) return;
            status[r] = (*job)(r);
            pending.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    sbr_status run(const std::function<sbr_status(uint32_t)>& f) {
        const uint32_t n = (uint32_t)status.size();
        job = &f;
        pending.store(n - 1, std::memory_order_release);
        {
            std::lock_guard<std::mutex> g(mu); /* a helper between its last spin and its wait must not miss the new generation */
            generation.fetch_add(1, std::memory_order_release);
        }
        cv.notify_all();
        status[0] = f(0);
        while (pending.load(std::memory_order_acquire) != 0) cpu_relax();
        for (uint32_t r = 0; r < n; ++r)
            if (status[r] != SBR_OK) return status[r];
        return SBR_OK;
    }
};

}  // namespace

struct sbr_group_plan {
    struct Dev {
        sbr_fit_plan* plan = nullptr;
        uint8_t *send = nullptr, *dense = nullptr, *recv = nullptr, *own = nullptr, *table = nullptr, *dense_all = nullptr;
        hipEvent_t scattered = nullptr, reduced = nullptr, applied = nullptr, gathered = nullptr;
        hipStream_t xs = nullptr; /* exchange stream (Asynchronous) */
    };
    std::vector<sbr_model*> models;
    std::vector<Dev> dev;
    uint32_t n = 0;
    uint64_t chunk = 0, db = 0, block_bytes = 0;
    bool partitioned = false, async = false, reforder = false;
    bool owner_applied = true;       /* Synchronous, replicated: the owners apply the update and the parameter slices travel (sbr_group_plan_set_exchange) */
    bool first = true;               /* no step has been applied yet: nothing to wait for */
    uint64_t nmb = 0;                /* minibatches of the prepared epoch */
    uint32_t epochs_prepared = 0;
    int64_t local_done = -1;         /* minibatch whose local half sbr_group_step_local has queued (parity access), or -1 */
    int64_t async_local_done = -1;   /* Asynchronous: minibatch whose local half the pipeline has queued ahead */
    std::unique_ptr<PhaseWorkers> workers;
    double enqueue_ms = 0.0;         /* host time spent inside sbr_group_step (queueing; a partitioned step includes its rendezvous) */
    uint64_t steps = 0;

    sbr_status phase(const std::function<sbr_status(uint32_t)>& f) {
        if (workers) return workers->run(f);
        for (uint32_t r = 0; r < n; ++r) SBRCHK(f(r));
        return SBR_OK;
    }
    sbr_status wait_applied(uint32_t r, hipStream_t s) { /* peers must be done with the previous step's buffers / table rows */
        if (first) return SBR_OK;
        for (uint32_t q = 0; q < n; ++q)
            if (q != r) HIPCHK(hipStreamWaitEvent(s, dev[q].applied, 0));
        return SBR_OK;
    }
    /* local half of minibatch mb on device r (a partitioned table must not still be written by the previous step's owners) */
    sbr_status local(uint32_t r, uint64_t mb) {
        SBRCHK(ensure_device(models[r]));
        /* (reference order: the peers copy this plan's block on their own streams) */
        if (partitioned || reforder) SBRCHK(wait_applied(r, models[r]->stream));
        return sbr_fit_step_local(dev[r].plan, mb);
    }
    /* reference order (sbr_model_set_reference_order on every replica): the devices' blocks are gathered and applied as n optimiser
     * steps in device order on every replica (sbr_fit_step_apply_blocks_in_order) */
    sbr_status reforder_step(uint64_t mb) {
        const bool have_local = local_done == (int64_t)mb;
        SBRCHK(phase([&](uint32_t r) -> sbr_status {
            SBRCHK(ensure_device(models[r]));
            SBRCHK(wait_applied(r, models[r]->stream)); /* the peers have copied the previous step's block out of this plan */
            if (!have_local) SBRCHK(sbr_fit_step_local(dev[r].plan, mb));
            SBRCHK(ensure_dense_reduced(dev[r].plan));
            HIPCHK(hipEventRecord(dev[r].scattered, models[r]->stream));
            return SBR_OK;
        }));
        SBRCHK(phase([&](uint32_t q) -> sbr_status {
            SBRCHK(ensure_device(models[q]));
            for (uint32_t r = 0; r < n; ++r) {
                if (r != q) HIPCHK(hipStreamWaitEvent(models[q]->stream, dev[r].scattered, 0));
                HIPCHK(hipMemcpyAsync(dev[q].recv + (size_t)r * block_bytes, dev[r].plan->block, block_bytes, hipMemcpyDefault, models[q]->stream));
            }
            SBRCHK(sbr_fit_step_apply_blocks_in_order(dev[q].plan, mb, dev[q].recv));
            HIPCHK(hipEventRecord(dev[q].applied, models[q]->stream));
            return SBR_OK;
        }));
        first = false;
        return SBR_OK;
    }
    sbr_status single_step(uint64_t mb) {
        if (local_done == (int64_t)mb) return sbr_fit_step_apply(dev[0].plan, mb);
        return sbr_fit_step(dev[0].plan, mb);
    }
    /* Parallelism::Synchronous: compute, exchange, apply — every device sees every update before its next minibatch.
     * owner_applied (default): the owner of a slice adds the devices' contributions AND updates its rows in place; the updated
     * parameter slices are all-gathered straight into every replica's table (peer copies) — no replica walks the whole table, a row's
     * optimiser state lives on its owner (gathered to all replicas when the fit ends).  Otherwise (sbr_group_plan_set_exchange 1;
     * what the staleness-one pipeline runs as well): the gradient chunks are all-gathered and every replica applies every update. */
    sbr_status sync_step(uint64_t mb) {
        const bool have_local = local_done == (int64_t)mb;
        SBRCHK(phase([&](uint32_t r) -> sbr_status {
            if (!have_local) SBRCHK(local(r, mb));
            SBRCHK(ensure_device(models[r]));
            SBRCHK(wait_applied(r, models[r]->stream));
            SBRCHK(sbr_fit_step_scatter(dev[r].plan, mb, dev[r].send));
            HIPCHK(hipEventRecord(dev[r].scattered, models[r]->stream));
            return SBR_OK;
        }));
        SBRCHK(phase([&](uint32_t p) -> sbr_status { /* all-to-all: chunk p of every device -> device p */
            SBRCHK(ensure_device(models[p]));
            for (uint32_t r = 0; r < n; ++r) {
                if (r != p) HIPCHK(hipStreamWaitEvent(models[p]->stream, dev[r].scattered, 0));
                HIPCHK(hipMemcpyAsync(dev[p].recv + r * chunk, dev[r].send + p * chunk, chunk, hipMemcpyDefault, models[p]->stream));
            }
            if (owner_applied) SBRCHK(sbr_fit_step_owner_update(dev[p].plan, dev[p].recv));
            else SBRCHK(sbr_fit_step_owner_reduce(dev[p].plan, dev[p].recv, dev[p].own));
            SBRCHK(sbr_fit_step_dense(dev[p].plan, dev[p].dense));
            HIPCHK(hipEventRecord(dev[p].reduced, models[p]->stream));
            return SBR_OK;
        }));
        SBRCHK(phase([&](uint32_t q) -> sbr_status { /* all-gather of the owners' slices / chunks and of the dense blocks */
            SBRCHK(ensure_device(models[q]));
            for (uint32_t p = 0; p < n; ++p) {
                if (p != q) HIPCHK(hipStreamWaitEvent(models[q]->stream, dev[p].reduced, 0));
                if (owner_applied) {
                    if (p != q) {
                        SBRCHK(copy_slice(q, p, SBR_PARAM_ITEM_EMBEDDING));
                        SBRCHK(copy_slice(q, p, SBR_PARAM_ITEM_BIAS));
                    }
                } else {
                    HIPCHK(hipMemcpyAsync(dev[q].table + p * chunk, dev[p].own, chunk, hipMemcpyDefault, models[q]->stream));
                }
                HIPCHK(hipMemcpyAsync(dev[q].dense_all + p * db, dev[p].dense, db, hipMemcpyDefault, models[q]->stream));
            }
            if (owner_applied) SBRCHK(sbr_fit_step_apply_dense(dev[q].plan, dev[q].dense_all));
            else SBRCHK(sbr_fit_step_apply_table(dev[q].plan, dev[q].table, dev[q].dense_all));
            HIPCHK(hipEventRecord(dev[q].applied, models[q]->stream));
            return SBR_OK;
        }));
        first = false;
        return SBR_OK;
    }
    /* the owner's reduced chunk and the gathered chunks of the gradient all-gather (the pipeline; sbr_group_plan_set_exchange 1) */
    sbr_status gradient_gather_buffers(uint32_t r) {
        Dev& v = dev[r];
        if (v.own) return SBR_OK;
        SBRCHK(ensure_device(models[r]));
        SBRCHK(dmalloc(&v.own, chunk));
        SBRCHK(dmalloc(&v.table, n * chunk));
        return SBR_OK;
    }
    /* replica q's copy of owner p's slice of one item-table block <- replica p's (queued on q's stream) */
    sbr_status copy_slice(uint32_t q, uint32_t p, int32_t which) {
        void *src = nullptr, *dst = nullptr;
        uint64_t sb = 0, sb2 = 0;
        SBRCHK(sbr_model_table_slice(models[p], which, &src, &sb));
        SBRCHK(sbr_model_table_slice(models[q], which, &dst, &sb2));
        if (!src || !dst || sb != sb2) return src || dst ? SBR_ERR_INVALID_ARGUMENT : SBR_OK;
        HIPCHK(hipMemcpyAsync(reinterpret_cast<uint8_t*>(dst) + (size_t)p * sb, reinterpret_cast<const uint8_t*>(src) + (size_t)p * sb, sb,
                              hipMemcpyDefault, models[q]->stream));
        return SBR_OK;
    }
    /* owner-applied steps leave a row's optimiser state on its owner only: every replica's copy made complete again (fit end) */
    sbr_status gather_optimizer_state() {
        bool any = false;
        for (uint32_t r = 0; r < n; ++r) any = any || models[r]->opt_state_partial;
        if (!any || partitioned) return SBR_OK;
        drain(); /* every owner's last update has landed */
        for (uint32_t q = 0; q < n; ++q) {
            SBRCHK(ensure_device(models[q]));
            for (uint32_t p = 0; p < n; ++p) {
                if (p == q) continue;
                for (int32_t which : {SBR_PARAM_ITEM_EMBEDDING_ACC, SBR_PARAM_ITEM_BIAS_ACC, SBR_PARAM_ITEM_EMBEDDING_M, SBR_PARAM_ITEM_BIAS_M})
                    SBRCHK(copy_slice(q, p, which));
            }
        }
        drain();
        for (uint32_t r = 0; r < n; ++r) models[r]->opt_state_partial = false;
        return SBR_OK;
    }
    /* Partitioned item table: every row is stored once (on its owner) and read by everybody through the shared mapping.  A step:
     * all devices compute on the current table; each reduces its own entries into a list and binary-searches its owner bounds;
     * every owner then waits — on its stream, through the devices' events — until nobody is READING the table any more, builds
     * its merge plan from the peers' bounds on the device, merges the peers' lists over its rows in device order and updates them in
     * place.  Bitwise the replicated Synchronous exchange.  Nothing of a step waits for the host. */
    sbr_status partitioned_step(uint64_t mb) {
        const bool have_local = local_done == (int64_t)mb;
        SBRCHK(phase([&](uint32_t r) -> sbr_status {
            if (!have_local) SBRCHK(local(r, mb));
            SBRCHK(ensure_device(models[r]));
            SBRCHK(partition_reduce_own(dev[r].plan, mb));
            SBRCHK(sbr_fit_step_dense(dev[r].plan, dev[r].dense));
            /* `scattered`: device r has finished READING the table, its list, owner bounds and dense block are complete — what the
             * owners wait for, on their own streams; the host reads nothing back and drains nothing (until round 5: a stream
             * synchronise + a blocking copy of the bounds per device and step, 0.96 of the step's time on the host) */
            HIPCHK(hipEventRecord(dev[r].scattered, models[r]->stream));
            return SBR_OK;
        }));
        SBRCHK(phase([&](uint32_t q) -> sbr_status {
            SBRCHK(ensure_device(models[q]));
            sbr::PeerLists pl;
            sbr::PeerBounds pb;
            std::memset(&pl, 0, sizeof(pl));
            for (uint32_t r = 0; r < 16; ++r) pb.b[r] = nullptr;
            for (uint32_t r = 0; r < n; ++r) {
                const sbr_fit_plan* pr = dev[r].plan;
                pl.keys[r] = pr->keys_sorted; pl.G[r] = pr->glist; pl.gb[r] = pr->gblist; pl.fl[r] = pr->gfl;
                pb.b[r] = pr->bounds_dev; /* read in place by merge_plan_kernel (peer-readable like the lists) */
                if (r != q) HIPCHK(hipStreamWaitEvent(models[q]->stream, dev[r].scattered, 0));
            }
            for (uint32_t r = 0; r < n; ++r)
                HIPCHK(hipMemcpyAsync(dev[q].dense_all + r * db, dev[r].dense, db, hipMemcpyDefault, models[q]->stream));
            SBRCHK(apply_dense_blocks(dev[q].plan, dev[q].dense_all));
            SBRCHK(partition_owner_apply(dev[q].plan, pl, pb));
            HIPCHK(hipEventRecord(dev[q].applied, models[q]->stream));
            return SBR_OK;
        }));
        first = false;
        return SBR_OK;
    }
    /* Parallelism::Asynchronous (mod.rs:36-38), the deterministic analogue of Hogwild: staleness is fixed at one step.  Device
     * r computes minibatch mb + 1 on its compute stream while the exchange of step mb (peer copies + owner reduce) runs on its
     * exchange stream; update mb is applied after that computation has read the parameters. */
    sbr_status async_step(uint64_t mb) {
        if (async_local_done < (int64_t)mb) {
            SBRCHK(phase([&](uint32_t r) -> sbr_status { return local(r, mb); }));
            async_local_done = (int64_t)mb;
        }
        const bool ahead = mb + 1 < nmb;
        SBRCHK(phase([&](uint32_t r) -> sbr_status {
            SBRCHK(ensure_device(models[r]));
            SBRCHK(wait_applied(r, models[r]->stream));
            SBRCHK(sbr_fit_step_scatter(dev[r].plan, mb, dev[r].send));
            SBRCHK(sbr_fit_step_dense(dev[r].plan, dev[r].dense));
            HIPCHK(hipEventRecord(dev[r].scattered, models[r]->stream));
            if (ahead) SBRCHK(sbr_fit_step_local(dev[r].plan, mb + 1));
            return SBR_OK;
        }));
        if (ahead) async_local_done = (int64_t)mb + 1;
        SBRCHK(phase([&](uint32_t p) -> sbr_status {
            SBRCHK(ensure_device(models[p]));
            for (uint32_t r = 0; r < n; ++r) {
                HIPCHK(hipStreamWaitEvent(dev[p].xs, dev[r].scattered, 0));
                HIPCHK(hipMemcpyAsync(dev[p].recv + r * chunk, dev[r].send + p * chunk, chunk, hipMemcpyDefault, dev[p].xs));
            }
            SBRCHK(sbr_fit_step_owner_reduce_on(dev[p].plan, dev[p].recv, dev[p].own, dev[p].xs)); /* belongs to the exchange stream */
            HIPCHK(hipEventRecord(dev[p].reduced, dev[p].xs));
            return SBR_OK;
        }));
        SBRCHK(phase([&](uint32_t q) -> sbr_status {
            SBRCHK(ensure_device(models[q]));
            for (uint32_t p = 0; p < n; ++p) {
                if (p != q) HIPCHK(hipStreamWaitEvent(dev[q].xs, dev[p].reduced, 0));
                HIPCHK(hipMemcpyAsync(dev[q].table + p * chunk, dev[p].own, chunk, hipMemcpyDefault, dev[q].xs));
                HIPCHK(hipMemcpyAsync(dev[q].dense_all + p * db, dev[p].dense, db, hipMemcpyDefault, dev[q].xs));
            }
            HIPCHK(hipEventRecord(dev[q].gathered, dev[q].xs));
            HIPCHK(hipStreamWaitEvent(models[q]->stream, dev[q].gathered, 0));
            SBRCHK(sbr_fit_step_apply_table(dev[q].plan, dev[q].table, dev[q].dense_all));
            HIPCHK(hipEventRecord(dev[q].applied, models[q]->stream));
            return SBR_OK;
        }));
        first = false;
        return SBR_OK;
    }
    sbr_status begin(sbr_model* const* ms, uint32_t count, const uint64_t* user_ptr, const uint32_t* item_ids, uint64_t num_users) {
        n = count;
        models.assign(ms, ms + count);
        dev.resize(n);
        partitioned = models[0]->shared != nullptr;
        /* a partitioned table is updated in place by its owners after a rendezvous, so there is no staleness-one pipeline for
         * it: Parallelism::Asynchronous runs the synchronous step there (same everywhere a partitioned table is driven) */
        async = models[0]->hp.parallelism == SBR_PAR_ASYNCHRONOUS && !partitioned;
        reforder = models[0]->reference_order;
        for (uint32_t r = 0; r < n; ++r)
            if (models[r]->reference_order != reforder) return SBR_ERR_INVALID_ARGUMENT;
        for (uint32_t r = 0; r < n; ++r) SBRCHK(sbr_fit_begin(models[r], user_ptr, item_ids, num_users, &dev[r].plan));
        block_bytes = dev[0].plan->block_bytes;
        SBRCHK(sbr_fit_chunk_bytes(dev[0].plan, &chunk));
        SBRCHK(sbr_fit_dense_bytes(dev[0].plan, &db));
        for (uint32_t r = 0; r < n; ++r) {
            Dev& v = dev[r];
            SBRCHK(ensure_device(models[r]));
            for (uint32_t q = 0; q < n; ++q)
                if (models[q]->device != models[r]->device) {
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, models[r]->device, models[q]->device) == hipSuccess && can)
                        (void)hipDeviceEnablePeerAccess(models[q]->device, 0); /* already-enabled is fine */
                }
            (void)hipGetLastError();
            if (n == 1) continue; /* single_step: no exchange buffers */
            if (reforder) { /* the devices' whole blocks travel (one sequence each: tens of KB) */
                SBRCHK(dmalloc(&v.recv, n * block_bytes));
                HIPCHK(hipEventCreateWithFlags(&v.scattered, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&v.applied, hipEventDisableTiming));
                continue;
            }
            SBRCHK(dmalloc(&v.dense, db)); SBRCHK(dmalloc(&v.dense_all, n * db));
            if (!partitioned) { /* the replicated exchange moves table-sized chunks; the partitioned one needs none */
                SBRCHK(dmalloc(&v.send, n * chunk)); SBRCHK(dmalloc(&v.recv, n * chunk));
                if (async) SBRCHK(gradient_gather_buffers(r)); /* (Synchronous: the parameter slices are gathered in place) */
            }
            HIPCHK(hipEventCreateWithFlags(&v.scattered, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&v.reduced, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&v.applied, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&v.gathered, hipEventDisableTiming));
            if (async) HIPCHK(hipStreamCreateWithFlags(&v.xs, hipStreamNonBlocking));
        }
        return SBR_OK;
    }
    sbr_status epoch_prepare(uint64_t* out_nmb, bool prefetch_next) {
        uint64_t k0 = 0;
        if (reforder && n > 1) { /* every replica shuffles EVERY worker's partition (it needs the workers' step sizes): each worker's
                                  * stream, advanced on its own device during the epoch, goes to all of them first */
            for (uint32_t r = 0; r < n; ++r) {
                sbr_fit_plan* pr = dev[r].plan;
                if (!pr->ref_rng || !pr->ref_rng_live) continue;
                SBRCHK(ensure_device(models[r]));
                HIPCHK(hipStreamSynchronize(models[r]->stream));
                uint32_t st4w[4];
                HIPCHK(hipMemcpy(st4w, pr->ref_rng, sizeof(st4w), hipMemcpyDeviceToHost));
                for (uint32_t q = 0; q < n; ++q) {
                    sbr_xorshift& x = dev[q].plan->part_rng[r];
                    x.x = st4w[0]; x.y = st4w[1]; x.z = st4w[2]; x.w = st4w[3];
                }
                pr->ref_rng_live = false;
            }
        }
        for (uint32_t r = 0; r < n; ++r) {
            uint64_t k = 0;
            SBRCHK(sbr_fit_epoch_prepare(dev[r].plan, &k));
            if (r && k != k0) return SBR_ERR_INVALID_ARGUMENT;
            k0 = k;
            if (prefetch_next) SBRCHK(sbr_fit_epoch_prefetch(dev[r].plan));
        }
        nmb = k0;
        local_done = async_local_done = -1;
        ++epochs_prepared;
        if (out_nmb) *out_nmb = k0;
        return SBR_OK;
    }
    sbr_status step(uint64_t mb) {
        if (mb >= nmb) return SBR_ERR_INVALID_ARGUMENT;
        const auto t0 = std::chrono::steady_clock::now();
        /* a group of one has nobody to exchange with: the single-device step (its table may still be a mapped range) */
        const sbr_status st = n == 1 ? single_step(mb) : reforder ? reforder_step(mb) : partitioned ? partitioned_step(mb) : async ? async_step(mb) : sync_step(mb);
        local_done = -1;
        enqueue_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ++steps;
        return st;
    }
    /* every stream of the group drained (also on an error path: a peer copy still in flight must not find its buffers recycled) */
    void drain() {
        for (uint32_t r = 0; r < n; ++r) {
            hipSetDevice(models[r]->device);
            hipStreamSynchronize(models[r]->stream);
            if (models[r]->side) hipStreamSynchronize(models[r]->side);
            if (models[r]->sorter) hipStreamSynchronize(models[r]->sorter);
            if (dev[r].xs) hipStreamSynchronize(dev[r].xs);
        }
    }
    ~sbr_group_plan() {
        workers.reset();
        (void)gather_optimizer_state(); /* (an abandoned plan as well: the models outlive it) */
        drain();
        for (uint32_t r = 0; r < n; ++r) {
            Dev& v = dev[r];
            hipSetDevice(models[r]->device);
            dfree(v.send); dfree(v.dense); dfree(v.recv); dfree(v.own); dfree(v.table); dfree(v.dense_all);
            if (v.scattered) hipEventDestroy(v.scattered);
            if (v.reduced) hipEventDestroy(v.reduced);
            if (v.applied) hipEventDestroy(v.applied);
            if (v.gathered) hipEventDestroy(v.gathered);
            if (v.xs) hipStreamDestroy(v.xs);
            if (v.plan) sbr_fit_plan_destroy(v.plan);
        }
    }
};

sbr_status sbr_group_fit_begin(sbr_model* const* models, uint32_t n, const uint64_t* user_ptr, const uint32_t* item_ids,
                               uint64_t num_users, sbr_group_plan** out) {
    if (!models || !out || n == 0 || n > 16) return SBR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    for (uint32_t r = 0; r < n; ++r) {
        const sbr_model* m = models[r];
        if (!m || m->hp.num_devices != n || m->hp.device_rank != r || m->hp.num_epochs != models[0]->hp.num_epochs ||
            m->shared != models[0]->shared)
            return SBR_ERR_INVALID_ARGUMENT;
    }
    sbr_group_plan* g = new (std::nothrow) sbr_group_plan;
    if (!g) return SBR_ERR_OUT_OF_MEMORY;
    const sbr_status st = g->begin(models, n, user_ptr, item_ids, num_users);
    if (st != SBR_OK) { delete g; return st; }
    /* default: one host thread per device from four devices on (measured: profiles/r05_group_driver.jsonl) */
    if (n >= 4) g->workers.reset(new PhaseWorkers(n));
    *out = g;
    return SBR_OK;
}

sbr_status sbr_group_plan_set_host_threads(sbr_group_plan* g, int32_t enable) {
    if (!g) return SBR_ERR_INVALID_ARGUMENT;
    if (enable && !g->workers && g->n > 1) g->workers.reset(new PhaseWorkers(g->n));
    if (!enable) g->workers.reset();
    return SBR_OK;
}

sbr_status sbr_group_plan_set_exchange(sbr_group_plan* g, int32_t gradient_all_gather) {
    if (!g) return SBR_ERR_INVALID_ARGUMENT;
    for (uint32_t r = 0; r < g->n; ++r)
        if (g->models[r]->opt_state_partial && gradient_all_gather) return SBR_ERR_INVALID_ARGUMENT; /* owner-applied steps have run: finish the fit first */
    if (gradient_all_gather && g->n > 1 && !g->partitioned && !g->reforder)
        for (uint32_t r = 0; r < g->n; ++r) SBRCHK(g->gradient_gather_buffers(r));
    g->owner_applied = !gradient_all_gather;
    return SBR_OK;
}

sbr_status sbr_group_gather_optimizer_state(sbr_group_plan* g) {
    if (!g) return SBR_ERR_INVALID_ARGUMENT;
    return g->gather_optimizer_state();
}

sbr_status sbr_group_epoch_prepare(sbr_group_plan* g, uint64_t* out_num_minibatches, int32_t prefetch_next) {
    if (!g) return SBR_ERR_INVALID_ARGUMENT;
    return g->epoch_prepare(out_num_minibatches, prefetch_next != 0);
}

sbr_status sbr_group_step(sbr_group_plan* g, uint64_t minibatch) {
    if (!g) return SBR_ERR_INVALID_ARGUMENT;
    return g->step(minibatch);
}

sbr_status sbr_group_step_local(sbr_group_plan* g, uint64_t minibatch) {
    if (!g || minibatch >= g->nmb || g->async) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(g->phase([&](uint32_t r) -> sbr_status { return g->local(r, minibatch); }));
    g->local_done = (int64_t)minibatch;
    return SBR_OK;
}

sbr_status sbr_group_member_plan(sbr_group_plan* g, uint32_t replica, sbr_fit_plan** out) {
    if (!g || !out || replica >= g->n) return SBR_ERR_INVALID_ARGUMENT;
    *out = g->dev[replica].plan;
    return SBR_OK;
}

sbr_status sbr_group_synchronize(sbr_group_plan* g) {
    if (!g) return SBR_ERR_INVALID_ARGUMENT;
    g->drain();
    return hipGetLastError() == hipSuccess ? SBR_OK : SBR_ERR_HIP;
}

sbr_status sbr_group_plan_stats(const sbr_group_plan* g, double* out_host_enqueue_ms, uint64_t* out_steps, int32_t* out_host_threads) {
    if (!g) return SBR_ERR_INVALID_ARGUMENT;
    if (out_host_enqueue_ms) *out_host_enqueue_ms = g->enqueue_ms;
    if (out_steps) *out_steps = g->steps;
    if (out_host_threads) *out_host_threads = g->workers ? (int32_t)g->n : 1;
    return SBR_OK;
}

sbr_status sbr_group_fit_end(sbr_group_plan* g, float* out_loss) {
    if (!g) return SBR_ERR_INVALID_ARGUMENT;
    auto finish = [&]() -> sbr_status {
        const uint32_t n = g->n;
        SBRCHK(g->gather_optimizer_state());
        for (uint32_t r = 1; r < n; ++r) SBRCHK(sbr_fit_end(g->dev[r].plan, nullptr, nullptr));
        float lagged = 0.0f; /* the workers' terms added in worker order, f32 (sequence_model.rs:173-177) */
        for (uint32_t r = 0; r < n; ++r) {
            float term = 0.0f;
            SBRCHK(sbr_fit_end_lagged(g->dev[r].plan, &term));
            lagged = lagged + term;
        }
        for (uint32_t r = 0; r < n; ++r) g->models[r]->last_lagged_loss = lagged;
        return sbr_fit_end(g->dev[0].plan, out_loss, nullptr);
    };
    const sbr_status st = finish();
    delete g;
    return st;
}

void sbr_group_plan_destroy(sbr_group_plan* g) { delete g; }

sbr_status sbr_group_fit(sbr_model* const* models, uint32_t n, const uint64_t* user_ptr, const uint32_t* item_ids,
                         uint64_t num_users, float* out_loss) {
    if (!models || n == 0) return SBR_ERR_INVALID_ARGUMENT;
    if (n == 1) {
        if (!models[0] || models[0]->hp.num_devices != 1 || models[0]->hp.device_rank != 0) return SBR_ERR_INVALID_ARGUMENT;
        return sbr_model_fit(models[0], user_ptr, item_ids, num_users, out_loss);
    }
    sbr_group_plan* g = nullptr;
    SBRCHK(sbr_group_fit_begin(models, n, user_ptr, item_ids, num_users, &g));
    const uint32_t epochs = models[0]->hp.num_epochs;
    sbr_status st = SBR_OK;
    for (uint32_t e = 0; e < epochs && st == SBR_OK; ++e) {
        uint64_t nmb = 0;
        st = sbr_group_epoch_prepare(g, &nmb, e + 1 < epochs);
        for (uint64_t mb = 0; mb < nmb && st == SBR_OK; ++mb) st = sbr_group_step(g, mb);
    }
    if (st != SBR_OK) { sbr_group_plan_destroy(g); return st; }
    return sbr_group_fit_end(g, out_loss);
}


/* n replicas of one model (num_devices = n, device_rank = r, replica r on HIP device r mod device count).
 * SBR_GROUP_PARTITION_ITEM_TABLE: instead of n full copies, the item table (embeddings, biases and their
 * optimiser state) exists once, rows [r*S, (r+1)*S), S = ceil(num_items / n), on replica r's device, mapped
 * into every replica's address space. */
sbr_status sbr_group_create(const sbr_hparams* hp, uint32_t n, uint32_t flags, sbr_model** out_models) {
    if (!hp || !out_models || n == 0 || n > 16) return SBR_ERR_INVALID_ARGUMENT;
    if (!storage_dim(hp->embedding_dim) || hp->num_items == 0) return SBR_ERR_INVALID_ARGUMENT;
    int ndevices = 0;
    if (hipGetDeviceCount(&ndevices) != hipSuccess || ndevices == 0) return SBR_ERR_NO_DEVICE;
    for (uint32_t r = 0; r < n; ++r) out_models[r] = nullptr;
    std::vector<int> devices(n);
    for (uint32_t r = 0; r < n; ++r) devices[r] = (int)(r % (uint32_t)ndevices);
    std::shared_ptr<SharedTable> shared;
    sbr_status st = SBR_OK;
    if (flags & SBR_GROUP_PARTITION_ITEM_TABLE) {
        shared = std::make_shared<SharedTable>();
        const uint64_t S = ((uint64_t)hp->num_items + n - 1) / n;
        st = shared->create(hp->num_items, (uint32_t)storage_dim(hp->embedding_dim), hp->optimizer == SBR_OPT_ADAM, devices, S);
    }
    for (uint32_t r = 0; r < n && st == SBR_OK; ++r) {
        sbr_hparams h = *hp;
        h.num_devices = n;
        h.device_rank = r;
        if (hipSetDevice(devices[r]) != hipSuccess) { st = SBR_ERR_HIP; break; }
        st = model_create_impl(&h, shared, r == 0 ? TABLE_WRITE_ALL : TABLE_WRITE_NONE, r ? &out_models[0]->rng_after_table : nullptr,
                               &out_models[r]);
    }
    (void)hipSetDevice(devices[0]);
    if (st != SBR_OK)
        for (uint32_t r = 0; r < n; ++r) { sbr_model_destroy(out_models[r]); out_models[r] = nullptr; }
    return st;
}

sbr_status sbr_model_is_partitioned(const sbr_model* m, int32_t* out) {
    if (!m || !out) return SBR_ERR_INVALID_ARGUMENT;
    *out = m->shared ? 1 : 0;
    return SBR_OK;
}

/* ---- partitioned item table under ONE PROCESS PER GPU --------------------------------------------------
 * Every process calls sbr_model_create_partitioned with the same hyper-parameters (num_devices = world,
 * device_rank = its rank) on its own device: the layout of the shared virtual range is planned identically
 * everywhere, the parts homed on this rank are allocated and mapped.  The host then moves file descriptors:
 * sbr_partition_export_part for own parts -> the peers' sbr_partition_import_part (Unix-socket SCM_RIGHTS;
 * sbr_rs_amd/partitioned.py), and sbr_partition_finalize writes this rank's rows. */
sbr_status sbr_model_create_partitioned(const sbr_hparams* hp, sbr_model** out) {
    if (!hp || !out) return SBR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!storage_dim(hp->embedding_dim) || hp->num_items == 0 || hp->num_devices == 0 || hp->num_devices > 16 ||
        hp->device_rank >= hp->num_devices)
        return SBR_ERR_INVALID_ARGUMENT;
    /* a partitioned table is updated in place by its owners after a rendezvous: Parallelism::Asynchronous runs the
     * synchronous step (no staleness-one pipeline), here as in sbr_group_fit */
    int ndevices = 0, device = 0;
    if (hipGetDeviceCount(&ndevices) != hipSuccess || ndevices == 0) return SBR_ERR_NO_DEVICE;
    HIPCHK(hipGetDevice(&device));
    auto shared = std::make_shared<SharedTable>();
    shared->local_rank = (int)hp->device_rank;
    const uint64_t S = ((uint64_t)hp->num_items + hp->num_devices - 1) / hp->num_devices;
    SBRCHK(shared->plan(hp->num_items, (uint32_t)storage_dim(hp->embedding_dim), hp->optimizer == SBR_OPT_ADAM, (int)hp->num_devices, S, device));
    std::vector<int> device_of_rank(hp->num_devices, device); /* only this rank's entry is used */
    SBRCHK(shared->create_parts(device_of_rank));
    return model_create_impl(hp, shared, TABLE_OWN_ROWS_DEFERRED, nullptr, out);
}

sbr_status sbr_partition_num_parts(const sbr_model* m, uint32_t* out) {
    if (!m || !m->shared || !out) return SBR_ERR_INVALID_ARGUMENT;
    *out = (uint32_t)m->shared->parts.size();
    return SBR_OK;
}

sbr_status sbr_partition_part_info(const sbr_model* m, uint32_t part, uint32_t* out_home_rank, uint64_t* out_bytes) {
    if (!m || !m->shared || part >= m->shared->parts.size()) return SBR_ERR_INVALID_ARGUMENT;
    if (out_home_rank) *out_home_rank = (uint32_t)m->shared->parts[part].home;
    if (out_bytes) *out_bytes = m->shared->parts[part].bytes;
    return SBR_OK;
}

sbr_status sbr_partition_export_part(sbr_model* m, uint32_t part, int32_t* out_fd) {
    if (!m || !m->shared || !out_fd || part >= m->shared->parts.size()) return SBR_ERR_INVALID_ARGUMENT;
    SharedTable::Part& pt = m->shared->parts[part];
    if (!pt.have || pt.home != m->shared->local_rank) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(m));
    int fd = -1;
    HIPCHK(hipMemExportToShareableHandle(&fd, pt.h, hipMemHandleTypePosixFileDescriptor, 0));
    *out_fd = fd;
    return SBR_OK;
}

sbr_status sbr_partition_import_part(sbr_model* m, uint32_t part, int32_t fd) {
    if (!m || !m->shared || m->shared->local_rank < 0) return SBR_ERR_INVALID_ARGUMENT;
    SBRCHK(ensure_device(m));
    return m->shared->import_part(part, fd);
}

sbr_status sbr_partition_finalize(sbr_model* m) {
    if (!m || !m->shared || m->shared->local_rank < 0) return SBR_ERR_INVALID_ARGUMENT;
    if (m->partition_finalized) return SBR_OK;
    SBRCHK(ensure_device(m));
    SBRCHK(m->shared->set_access(std::vector<int>{m->device}));
    const sbr::ModelView& v = m->mv;
    const uint64_t I = m->hp.num_items, d = (uint64_t)m->d, S = m->shared->slice;
    const uint64_t r0 = std::min<uint64_t>(I, (uint64_t)m->hp.device_rank * S), r1 = std::min<uint64_t>(I, r0 + S), nr = r1 - r0;
    if (nr) {
        HIPCHK(hipMemcpy(v.E + r0 * d, m->pending_E.data(), nr * d * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(v.Eacc + r0 * d, 0, nr * d * 4));
        HIPCHK(hipMemset(v.b + r0, 0, nr * 4));
        HIPCHK(hipMemset(v.bacc + r0, 0, nr * 4));
        if (m->hp.optimizer == SBR_OPT_ADAM) {
            HIPCHK(hipMemset(v.Em + r0 * d, 0, nr * d * 4));
            HIPCHK(hipMemset(v.bm + r0, 0, nr * 4));
        }
    }
    HIPCHK(hipDeviceSynchronize());
    std::vector<float>().swap(m->pending_E);
    m->partition_finalized = true;
    return SBR_OK;
}

sbr_status sbr_device_count(int32_t* out_count) {
    if (!out_count) return SBR_ERR_INVALID_ARGUMENT;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) return SBR_ERR_NO_DEVICE;
    *out_count = nd;
    return SBR_OK;
}

sbr_status sbr_fit_debug_fetch(sbr_fit_plan* p, int32_t which, void* host_out, uint64_t bytes) {
    if (!p || !host_out || !p->last_block) return SBR_ERR_INVALID_ARGUMENT;
    sbr_model* m = p->m;
    SBRCHK(ensure_device(m));
    if (which == SBR_DBG_DENSE_GRAD) SBRCHK(ensure_dense_reduced(p));
    else SBRCHK(join_dense(p));
    HIPCHK(hipStreamSynchronize(m->stream));
    const sbr::BlockView bv = block_view(m, const_cast<void*>(p->last_block), p->rmax);
    const uint64_t R = (uint64_t)p->last_R, d = (uint64_t)m->d;
    const void* src = nullptr;
    uint64_t n = 0;
    switch (which) {
        case SBR_DBG_HIDDEN: src = bv.H; n = R * d * 4; break;
        case SBR_DBG_NEGATIVES: src = bv.neg; n = R * 4; break;
        case SBR_DBG_COEF: src = bv.coef; n = R * 4; break;
        case SBR_DBG_LOSS: src = p->wb.v.loss; n = R * 4; break;
        case SBR_DBG_DHIDDEN:
            /* only defined before the optimiser step of the same minibatch has been applied */
            sbr::launch_materialize_dh(m->mv, bv, p->last_R, p->wb.v.dH, m->stream);
            HIPCHK(hipStreamSynchronize(m->stream));
            src = p->wb.v.dH; n = R * d * 4; break;
        case SBR_DBG_DINPUT: src = bv.dX; n = R * d * 4; break;
        case SBR_DBG_DENSE_GRAD: src = bv.dense; n = dense_count(m) * 4; break;
        case SBR_DBG_IN_IDX: src = bv.in_idx; n = R * 4; break;
        case SBR_DBG_OUT_IDX: src = bv.out_idx; n = R * 4; break;
        case SBR_DBG_TRIES: src = p->wb.v.tries; n = R * 4; break;
        case SBR_DBG_DZ: if (!m->ng) return SBR_ERR_INVALID_ARGUMENT; src = p->wb.v.dZ; n = R * (uint64_t)m->ng * d * 4; break;
        default: return SBR_ERR_INVALID_ARGUMENT;
    }
    if (bytes < n) return SBR_ERR_INVALID_ARGUMENT;
    if (n) HIPCHK(hipMemcpy(host_out, src, n, hipMemcpyDeviceToHost));
    return SBR_OK;
}

/* ---------------------------------------------------------------------------------------------
 * prediction side
 * ------------------------------------------------------------------------------------------- */
namespace {

/* Runs the recurrent forward for a batch of histories; the hidden states are left in *H_out — memory of the model's eval arena,
 * valid until the next arena.reserve — and rep_row[i] = packed row of the final state of history i.  `extra_bytes`: what the
 * caller will carve from the arena afterwards (the rank kernels' arrays), reserved in the same allocation. */
size_t arena_bytes_forward(const sbr_model* m, uint64_t R, uint64_t B, uint64_t noff) {
    const uint64_t d = (uint64_t)m->d;
    size_t n = 4 * DeviceArena::padded(R * 4) + DeviceArena::padded(noff * 4) + DeviceArena::padded(B * 4) + DeviceArena::padded(R * d * 4);
    if (m->ng) n += 2 * DeviceArena::padded(R * d * 4) + DeviceArena::padded(R * d * 16);
    return n + 4096;
}

sbr_status forward_histories(sbr_model* m, const std::vector<const uint32_t*>& first, const std::vector<int>& nsteps,
                             float** H_out, std::vector<int>* rep_row, size_t extra_bytes) {
    Packed pk;
    pack_sequences(first, nsteps, false, nullptr, (int)m->hp.max_sequence_length, &pk);
    DeviceArena& ar = m->eval_arena;
    HIPCHK(hipStreamSynchronize(m->stream)); /* nothing of an earlier call may still read the arena */
    SBRCHK(ar.reserve(arena_bytes_forward(m, (uint64_t)pk.R, (uint64_t)pk.B, pk.off.size()) + extra_bytes));
    const uint64_t R = (uint64_t)pk.R, d = (uint64_t)m->d;
    DevicePacked dp;
    dp.in_idx = ar.take<uint32_t>(R); dp.out_idx = ar.take<uint32_t>(R); dp.ctr = ar.take<uint32_t>(R);
    dp.prev_row = ar.take<int>(R); dp.off = ar.take<int>(pk.off.size()); dp.steps = ar.take<int>((size_t)pk.B);
    float* H = ar.take<float>(R * d);
    WorkBuffers wb;
    wb.v.fold_max_tiles = SBR_FOLD_MAX_TILES_DEFAULT;
    if (m->ng) {
        wb.v.C = ar.take<float>(R * d);
        wb.v.G = ar.take<float>(R * d * 4);
        wb.v.X = ar.take<float>(R * d);
    }
    HIPCHK(hipMemcpyAsync(dp.in_idx, pk.in_idx.data(), (size_t)pk.R * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(dp.prev_row, pk.prev_row.data(), (size_t)pk.R * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(dp.off, pk.off.data(), pk.off.size() * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(dp.steps, pk.steps.data(), (size_t)pk.B * 4, hipMemcpyHostToDevice, m->stream));
    sbr::MbView mv;
    mv.R = pk.R; mv.B = pk.B; mv.Tm = pk.Tm;
    mv.off = dp.off; mv.steps = dp.steps; mv.prev_row = dp.prev_row;
    mv.in_idx = dp.in_idx; mv.out_idx = dp.out_idx; mv.ctr = dp.ctr;
    {
        ScopedTimer t(m, SBR_K_RECURRENT_FWD, m->ng && m->d > 128 ? (uint64_t)pk.Tm : 1);
        sbr::launch_recurrent_forward(m->mv, mv, H, wb.v, pk.Tm, pk.off.data(), m->stream);
    }
    /* the host vectors of `pk` are read by the asynchronous copies above: drain them before pk goes out of scope */
    HIPCHK(hipStreamSynchronize(m->stream));
    rep_row->assign(first.size(), 0);
    for (int b = 0; b < pk.B; ++b) (*rep_row)[pk.order[b]] = pk.off[pk.steps[b] - 1] + b;
    *H_out = H;
    return SBR_OK;
}

}  // namespace

sbr_status sbr_user_representation(sbr_model* m, const uint32_t* item_ids, uint64_t n, float* out_dim) {
    if (!m || !out_dim || (n && !item_ids)) return SBR_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(m->mu);
    SBRCHK(ensure_device(m));
    const uint64_t T = m->hp.max_sequence_length;
    static const uint32_t zero = 0;
    if (n > T) { item_ids += n - T; n = T; } /* sequence_model.rs:188 */
    if (n == 0) { item_ids = &zero; n = 1; } /* default index 0 (lstm.rs:262-264) */
    for (uint64_t t = 0; t < n; ++t)
        if (item_ids[t] >= m->hp.num_items) return SBR_ERR_INVALID_ARGUMENT;
    std::vector<const uint32_t*> first{item_ids};
    std::vector<int> nsteps{(int)n};
    float* H = nullptr;
    std::vector<int> rep_row;
    SBRCHK(forward_histories(m, first, nsteps, &H, &rep_row, 0));
    hipError_t e = hipMemcpy(out_dim, H + (size_t)rep_row[0] * m->d, (size_t)m->dl * 4, hipMemcpyDeviceToHost);
    return e == hipSuccess ? SBR_OK : SBR_ERR_HIP;
}

sbr_status sbr_predict(sbr_model* m, const float* user_dim, const uint32_t* item_ids, uint64_t n, float* out) {
    if (!m || !user_dim || (n && (!item_ids || !out))) return SBR_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(m->mu);
    SBRCHK(ensure_device(m));
    for (uint64_t i = 0; i < n; ++i)
        if (item_ids[i] >= m->hp.num_items) return SBR_ERR_INVALID_ARGUMENT;
    if (n == 0) return SBR_OK;
    float *d_user = nullptr, *d_out = nullptr;
    uint32_t* d_items = nullptr;
    sbr_status st = SBR_OK;
    if ((st = dmalloc(&d_user, m->d)) == SBR_OK && (st = dmalloc(&d_items, n)) == SBR_OK && (st = dmalloc(&d_out, n)) == SBR_OK) {
        hipMemset(d_user, 0, (size_t)m->d * 4);
        hipMemcpy(d_user, user_dim, (size_t)m->dl * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_items, item_ids, n * 4, hipMemcpyHostToDevice);
        sbr::launch_predict(m->mv, d_user, d_items, n, d_out, m->stream);
        if (hipStreamSynchronize(m->stream) != hipSuccess || hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost) != hipSuccess)
            st = SBR_ERR_HIP;
    }
    dfree(d_user); dfree(d_items); dfree(d_out);
    if (st != SBR_OK) return st;
    for (uint64_t i = 0; i < n; ++i)
        if (!std::isfinite(out[i])) return SBR_ERR_INVALID_PREDICTION; /* sequence_model.rs:225-229 */
    return SBR_OK;
}

sbr_status sbr_mrr_score(sbr_model* m, const uint64_t* user_ptr, const uint32_t* item_ids, uint64_t num_users,
                         float* out_mrr, uint32_t* out_ranks, uint64_t* out_num_ranked) {
    if (!m || !user_ptr || !out_mrr) return SBR_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(m->mu);
    SBRCHK(ensure_device(m));
    const uint64_t T = m->hp.max_sequence_length;
    for (uint64_t u = 0; u < num_users; ++u)
        if (user_ptr[u + 1] < user_ptr[u]) return SBR_ERR_INVALID_ARGUMENT; /* pointers must be non-decreasing */
    const uint64_t nnz = user_ptr[num_users];
    if (nnz && !item_ids) return SBR_ERR_INVALID_ARGUMENT;
    for (uint64_t i = 0; i < nnz; ++i)
        if (item_ids[i] >= m->hp.num_items) return SBR_ERR_INVALID_ARGUMENT;
    std::vector<uint64_t> users; /* users with >= 2 interactions (evaluation.rs:20) */
    for (uint64_t u = 0; u < num_users; ++u)
        if (user_ptr[u + 1] - user_ptr[u] >= 2) users.push_back(u);
    if (out_num_ranked) *out_num_ranked = users.size();
    std::vector<uint32_t> ranks(users.size(), 0);
    /* users per scoring launch: the rank GEMM streams the item table once per 128 users whatever the launch size, but every
     * launch has its ramp and tail — 8 192 users x 1e6 items at d = 128: 104 / 108 / 109 / 112 TFLOP/s at 1 024 / 2 048 /
     * 4 096 / 8 192 users per launch */
    constexpr size_t EVAL_B = 8192;
    /* the forward pass's scratch is (users x history steps) rows of 6d floats: bound a launch by rows as well */
    const size_t eval_rows_cap = (size_t)1 << 22;
    for (size_t c0 = 0, c1 = 0; c0 < users.size(); c0 = c1) {
        size_t rows = 0;
        for (c1 = c0; c1 < users.size() && c1 - c0 < EVAL_B; ++c1) {
            const uint64_t nh = user_ptr[users[c1] + 1] - user_ptr[users[c1]] - 1;
            rows += (size_t)std::min<uint64_t>(nh, T);
            if (rows > eval_rows_cap && c1 > c0) break;
        }
        const size_t nu = c1 - c0;
        std::vector<const uint32_t*> first(nu);
        std::vector<int> nsteps(nu);
        std::vector<uint32_t> test_item(nu), test_in_hist(nu, 0), hist_items;
        std::vector<uint64_t> hist_ptr(nu + 1, 0), raw_ptr(nu + 1, 0);
        for (size_t i = 0; i < nu; ++i) raw_ptr[i + 1] = raw_ptr[i] + (user_ptr[users[c0 + i] + 1] - user_ptr[users[c0 + i]] - 1);
        hist_items.resize(raw_ptr[nu]);
        std::vector<uint32_t> uniq_count(nu, 0);
        /* ALL history items of a user are masked (evaluation.rs:30-32): sorted + de-duplicated in place inside one buffer (a vector
         * per user was a third of mrr_score's host time at 8 192 users), a few host threads over user ranges */
        auto prepare = [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; ++i) {
                const uint64_t u = users[c0 + i];
                const uint32_t* it = item_ids + user_ptr[u];
                const uint64_t n = user_ptr[u + 1] - user_ptr[u];
                const uint64_t nh = n - 1; /* train_items = all but last (evaluation.rs:24) */
                test_item[i] = it[n - 1];
                const uint64_t keep = std::min(nh, T); /* last T items feed the state (sequence_model.rs:188) */
                first[i] = it + (nh - keep);
                nsteps[i] = (int)keep;
                uint32_t* h = hist_items.data() + raw_ptr[i];
                std::memcpy(h, it, nh * 4);
                std::sort(h, h + nh);
                uint32_t* e = std::unique(h, h + nh);
                test_in_hist[i] = std::binary_search(h, e, test_item[i]) ? 1u : 0u;
                uniq_count[i] = (uint32_t)(e - h);
            }
        };
        {
            const size_t nthreads = nu >= 2048 ? 8 : 1;
            if (nthreads == 1) prepare(0, nu);
            else {
                std::vector<std::thread> workers;
                for (size_t k = 0; k < nthreads; ++k) workers.emplace_back(prepare, nu * k / nthreads, nu * (k + 1) / nthreads);
                for (auto& w : workers) w.join();
            }
        }
        for (size_t i = 0; i < nu; ++i) { /* close the gaps the de-duplication left */
            if (hist_ptr[i] != raw_ptr[i]) std::memmove(hist_items.data() + hist_ptr[i], hist_items.data() + raw_ptr[i], (size_t)uniq_count[i] * 4);
            hist_ptr[i + 1] = hist_ptr[i] + uniq_count[i];
        }
        hist_items.resize(hist_ptr[nu]);
        float* H = nullptr;
        std::vector<int> rep_row;
        const size_t rank_bytes = 6 * DeviceArena::padded(nu * 4) + DeviceArena::padded(hist_items.size() * 4 + 4) + DeviceArena::padded(4) +
                                  DeviceArena::padded((nu + 1) * 8);
        SBRCHK(forward_histories(m, first, nsteps, &H, &rep_row, rank_bytes));
        DeviceArena& ar = m->eval_arena;
        int* d_rep = ar.take<int>(nu);
        uint32_t *d_test = ar.take<uint32_t>(nu), *d_tih = ar.take<uint32_t>(nu), *d_hist = ar.take<uint32_t>(hist_items.size()),
                 *d_ranks = ar.take<uint32_t>(nu), *d_flag = ar.take<uint32_t>(1);
        float* d_ts = ar.take<float>(nu);
        uint64_t* d_hptr = ar.take<uint64_t>(nu + 1);
        sbr_status st = SBR_OK;
        {
            hipMemcpyAsync(d_rep, rep_row.data(), nu * 4, hipMemcpyHostToDevice, m->stream);
            hipMemcpyAsync(d_test, test_item.data(), nu * 4, hipMemcpyHostToDevice, m->stream);
            hipMemcpyAsync(d_tih, test_in_hist.data(), nu * 4, hipMemcpyHostToDevice, m->stream);
            if (!hist_items.empty()) hipMemcpyAsync(d_hist, hist_items.data(), hist_items.size() * 4, hipMemcpyHostToDevice, m->stream);
            hipMemcpyAsync(d_hptr, hist_ptr.data(), (nu + 1) * 8, hipMemcpyHostToDevice, m->stream);
            hipMemsetAsync(d_flag, 0, 4, m->stream);
            {
                ScopedTimer t(m, SBR_K_RANK, 1);
                sbr::launch_rank(m->mv, H, d_rep, (uint32_t)nu, d_test, d_tih, d_hptr, d_hist, d_ts, d_ranks, d_flag, m->stream);
            }
            uint32_t flag = 0;
            if (hipStreamSynchronize(m->stream) != hipSuccess ||
                hipMemcpy(ranks.data() + c0, d_ranks, nu * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(&flag, d_flag, 4, hipMemcpyDeviceToHost) != hipSuccess)
                st = SBR_ERR_HIP;
            else if (flag)
                st = SBR_ERR_INVALID_PREDICTION; /* predict fails the call on a non-finite score */
        }
        if (st != SBR_OK) return st;
    }
    float sum = 0.0f; /* evaluation.rs:47 — sequential f32 sum in user order */
    for (size_t i = 0; i < ranks.size(); ++i) sum += 1.0f / (float)ranks[i];
    *out_mrr = sum / (float)ranks.size();
    if (out_ranks) std::memcpy(out_ranks, ranks.data(), ranks.size() * 4);
    return SBR_OK;
}

/* ---------------------------------------------------------------------------------------------
 * numerics self-tests (tests/test_numerics_gpu.py): run the contract's primitives on the device
 * ------------------------------------------------------------------------------------------- */
sbr_status sbr_selftest_math(const float* x, uint64_t n, float* out_cell_h, float* out_sig, float* out_tanh) {
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) return SBR_ERR_NO_DEVICE;
    float *dx, *de, *ds, *dt;
    SBRCHK(dmalloc(&dx, n)); SBRCHK(dmalloc(&de, n)); SBRCHK(dmalloc(&ds, n)); SBRCHK(dmalloc(&dt, n));
    HIPCHK(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
    sbr::launch_selftest_math(dx, de, ds, dt, n, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_cell_h, de, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_sig, ds, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_tanh, dt, n * 4, hipMemcpyDeviceToHost));
    dfree(dx); dfree(de); dfree(ds); dfree(dt);
    return SBR_OK;
}

sbr_status sbr_selftest_dot_tree(const float* x, const float* y, uint32_t d, uint64_t nrows, float* out) {
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) return SBR_ERR_NO_DEVICE;
    if (!dim_ok(d) || nrows == 0) return SBR_ERR_INVALID_ARGUMENT;
    float *dx, *dy, *dout;
    SBRCHK(dmalloc(&dx, nrows * d)); SBRCHK(dmalloc(&dy, nrows * d)); SBRCHK(dmalloc(&dout, nrows));
    HIPCHK(hipMemcpy(dx, x, nrows * d * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dy, y, nrows * d * 4, hipMemcpyHostToDevice));
    sbr::launch_selftest_dot_tree(dx, dy, (int)d, nrows, dout, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout, nrows * 4, hipMemcpyDeviceToHost));
    dfree(dx); dfree(dy); dfree(dout);
    return SBR_OK;
}

/* out[16][16] = c0 + a[16][k] b[k][16] on v_mfma_f32_16x16x4_f32; out32[32][32] = a32[32][k] b32[k][32]
 * on v_mfma_f32_32x32x2_f32 (k multiple of 4) */
sbr_status sbr_selftest_mfma(const float* a, const float* b, const float* c0, uint32_t k, float* out,
                             const float* a32, const float* b32, float* out32) {
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) return SBR_ERR_NO_DEVICE;
    if (k == 0 || k % 4) return SBR_ERR_INVALID_ARGUMENT;
    float *da, *db, *dc, *dout, *da32, *db32, *dout32;
    SBRCHK(dmalloc(&da, 16 * k)); SBRCHK(dmalloc(&db, 16 * k)); SBRCHK(dmalloc(&dc, 256)); SBRCHK(dmalloc(&dout, 256));
    SBRCHK(dmalloc(&da32, 32 * k)); SBRCHK(dmalloc(&db32, 32 * k)); SBRCHK(dmalloc(&dout32, 1024));
    HIPCHK(hipMemcpy(da, a, 16 * k * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, b, 16 * k * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dc, c0, 256 * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(da32, a32, 32 * k * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db32, b32, 32 * k * 4, hipMemcpyHostToDevice));
    sbr::launch_selftest_mfma_chain(da, db, dc, (int)k, dout, nullptr);
    sbr::launch_selftest_mfma32_chain(da32, db32, (int)k, dout32, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout, 256 * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out32, dout32, 1024 * 4, hipMemcpyDeviceToHost));
    dfree(da); dfree(db); dfree(dc); dfree(dout); dfree(da32); dfree(db32); dfree(dout32);
    return SBR_OK;
}

/* keys (rows[e] << 32 | e), e = 0..n-1, in (row, e) order on the engine's own key ordering (sbr_sort.hip), and the
 * positions at which a new row starts */
sbr_status sbr_selftest_sort(const uint32_t* rows, uint64_t n, uint32_t row_bits, uint64_t* out_keys, uint32_t* out_head_pos,
                             uint32_t* out_nheads) {
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) return SBR_ERR_NO_DEVICE;
    if (!rows || !out_keys || !out_head_pos || !out_nheads || n == 0 || n >= (1ull << 32) || row_bits == 0 || row_bits > 32)
        return SBR_ERR_INVALID_ARGUMENT;
    uint32_t *drows, *dheads, *dn;
    uint64_t *dtmp, *dout;
    uint8_t* temp;
    SBRCHK(dmalloc(&drows, n)); SBRCHK(dmalloc(&dheads, n + 1)); SBRCHK(dmalloc(&dn, 1));
    SBRCHK(dmalloc(&dtmp, n)); SBRCHK(dmalloc(&dout, n));
    SBRCHK(dmalloc(&temp, sbr::sparse_sort_temp_bytes(n, 32 + (int)row_bits)));
    HIPCHK(hipMemcpy(drows, rows, n * 4, hipMemcpyHostToDevice));
    sbr::launch_selftest_sort(drows, (uint32_t)n, (int)row_bits, dtmp, dout, temp, dheads, dn, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_keys, dout, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_nheads, dn, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_head_pos, dheads, ((uint64_t)*out_nheads + 1) * 4, hipMemcpyDeviceToHost));
    dfree(drows); dfree(dheads); dfree(dn); dfree(dtmp); dfree(dout); dfree(temp);
    return SBR_OK;
}

sbr_status sbr_set_device(int32_t ordinal) {
    HIPCHK(hipSetDevice(ordinal));
    return SBR_OK;
}

}  // extern "C"
