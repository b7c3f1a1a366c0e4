// hbm_ceiling.hip — what this MI355X's memory system sustains for the access shapes of the hot path,
// measured with nothing else in the kernels: sequential float4 read / copy, and random ROW gathers and
// read-modify-writes of 512 B and 1 KiB rows out of tables far larger than the 256 MiB Infinity Cache
// (the item-table accesses of score_kernel / seg_short_kernel).  The roofline fractions in DESIGN.md are
// quoted against the 8 TB/s spec; this prints the attainable figures beside it.
//
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_ceiling.hip -o tools/bin/hbm_ceiling && tools/bin/hbm_ceiling
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ void fill_idx(uint32_t* idx, size_t n, uint32_t rows, uint64_t seed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)(((mix64(seed + i * 0x9E3779B97F4A7C15ULL) >> 32) * (uint64_t)rows) >> 32);
}

__global__ __launch_bounds__(256) void seq_read(const float4* __restrict__ src, size_t n4, float* out) {
    float acc = 0.0f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[i + u * stride < n4 ? i + u * stride : i];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void seq_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

// one L-lane group per row (16 B per lane), U independent rows per group and pass
template <int L, int U, bool RMW>
__global__ __launch_bounds__(256) void row_access(float4* tab, const uint32_t* __restrict__ idx, size_t n, float* out) {
    const int lane = threadIdx.x & 63;
    const int lg = lane % L, grp = lane / L;
    constexpr int GPW = 64 / L;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    float acc = 0.0f;
    for (size_t base = wave * GPW * U; base < n; base += nwaves * GPW * U) {
        uint32_t r[U];
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t p = base + u * GPW + grp;
            r[u] = idx[p < n ? p : n - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = tab[(size_t)r[u] * L + lg];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (RMW) {
                v[u].x += 1.0f;
                if (base + u * GPW + grp < n) tab[(size_t)r[u] * L + lg] = v[u];
            } else {
                acc += v[u].x + v[u].y + v[u].z + v[u].w;
            }
        }
    }
    if (!RMW && acc == 12345.678f) out[0] = acc;
}

template <class F>
static double time_ms(F&& launch, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <int L, int U, bool RMW>
static void run_rows(const char* what, float4* tab, size_t table_rows, const uint32_t* idx, size_t n, float* out) {
    const double ms = time_ms([&] { hipLaunchKernelGGL((row_access<L, U, RMW>), dim3(256 * 8), dim3(256), 0, 0, tab, idx, n, out); }, 5);
    const double bytes = (double)n * L * 16 * (RMW ? 2 : 1);
    printf("{\"access\": \"%s\", \"row_bytes\": %d, \"rows_in_flight_per_group\": %d, \"table_GiB\": %.2f, \"accesses\": %zu, \"ms\": %.3f, \"GBps\": %.0f}\n",
           what, L * 16, U, (double)table_rows * L * 16 / (1 << 30), n, ms, bytes / ms / 1e6);
}

int main() {
    const size_t table_bytes = (size_t)8 << 30;  // 8 GiB: 32x the Infinity Cache
    const size_t n_access = (size_t)8 << 20;     // 8 Mi row accesses per launch
    float4 *tab, *dst;
    uint32_t* idx;
    float* out;
    CHECK(hipMalloc(&tab, table_bytes));
    CHECK(hipMalloc(&dst, table_bytes));
    CHECK(hipMalloc(&idx, n_access * 4));
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMemset(tab, 0, table_bytes));
    CHECK(hipMemset(dst, 0, table_bytes));
    const size_t n4 = table_bytes / 16;
    double ms = time_ms([&] { hipLaunchKernelGGL(seq_read, dim3(256 * 16), dim3(256), 0, 0, tab, n4, out); }, 5);
    printf("{\"access\": \"sequential read\", \"GiB\": 8, \"ms\": %.3f, \"GBps\": %.0f}\n", ms, table_bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(seq_copy, dim3(256 * 16), dim3(256), 0, 0, tab, dst, n4); }, 5);
    printf("{\"access\": \"sequential copy (read + write)\", \"GiB\": 8, \"ms\": %.3f, \"GBps\": %.0f}\n", ms, 2.0 * table_bytes / ms / 1e6);
    {  // 512-byte rows (d = 128)
        const size_t rows = table_bytes / 512;
        hipLaunchKernelGGL(fill_idx, dim3((n_access + 255) / 256), dim3(256), 0, 0, idx, n_access, (uint32_t)rows, 1ull);
        run_rows<32, 1, false>("random row gather", tab, rows, idx, n_access, out);
        run_rows<32, 4, false>("random row gather", tab, rows, idx, n_access, out);
        run_rows<32, 8, false>("random row gather", tab, rows, idx, n_access, out);
        run_rows<32, 4, true>("random row read-modify-write", tab, rows, idx, n_access, out);
    }
    {  // 1 KiB rows (d = 256)
        const size_t rows = table_bytes / 1024;
        hipLaunchKernelGGL(fill_idx, dim3((n_access + 255) / 256), dim3(256), 0, 0, idx, n_access, (uint32_t)rows, 2ull);
        run_rows<64, 1, false>("random row gather", tab, rows, idx, n_access / 2, out);
        run_rows<64, 4, false>("random row gather", tab, rows, idx, n_access / 2, out);
        run_rows<64, 8, false>("random row gather", tab, rows, idx, n_access / 2, out);
        run_rows<64, 4, true>("random row read-modify-write", tab, rows, idx, n_access / 2, out);
    }
    {  // 512-byte rows out of a table that half fits the cache (the bench's 1 M x 128 table: 488 MiB)
        const size_t rows = 1000000;
        hipLaunchKernelGGL(fill_idx, dim3((n_access + 255) / 256), dim3(256), 0, 0, idx, n_access, (uint32_t)rows, 3ull);
        run_rows<32, 4, false>("random row gather", tab, rows, idx, n_access, out);
    }
    return 0;
}
