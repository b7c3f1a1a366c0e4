// counter_calibration.hip — what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for KNOWN byte counts, per access
// width and shape (MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts half the bytes of 16-byte-per-lane streaming reads;
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel touches a 4 GiB buffer (16x the Infinity Cache) exactly once per launch, so the bytes that must come from
// / go to memory are known: run under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE -d A -o run -- tools/bin/counter_calibration
//     rocprofv3 --kernel-trace --pmc WRITE_SIZE -d B -o run -- tools/bin/counter_calibration
// and divide (tools/pmc_dispatches.py).  The shapes are the score kernel's: 16-byte-per-lane row reads (sequential and random
// 512-byte rows), 4-byte-per-lane sequential reads (index arrays), random 4-byte reads (bias gathers: one 4-byte word per
// 64-byte line), and 16- / 4-byte-per-lane sequential writes (its result arrays).
//   hipcc --offload-arch=gfx950 -O3 tools/counter_calibration.hip -o tools/bin/counter_calibration
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// names carry the known bytes per launch: B = buffer bytes
__global__ __launch_bounds__(256) void cal_read_seq_16B(const float4* __restrict__ src, size_t n, float* out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void cal_read_seq_8B(const float2* __restrict__ src, size_t n, float* out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float2 v = src[i]; acc += v.x + v.y; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void cal_read_seq_4B(const float* __restrict__ src, size_t n, float* out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc == 12345.678f) out[0] = acc;
}
// random 512-byte rows, 16 B per lane (32 lanes per row): `rows` distinct-ish rows of a table with `table_rows` rows
__global__ __launch_bounds__(256) void cal_read_rows_512B(const float4* __restrict__ tab, size_t table_rows, size_t accesses, float* out) {
    float acc = 0.0f;
    const size_t grp = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 5, ngrp = ((size_t)gridDim.x * 256) >> 5;
    const int lg = threadIdx.x & 31;
    for (size_t a = grp; a < accesses; a += ngrp) {
        const size_t r = (size_t)(((mix64(a * 0x9E3779B97F4A7C15ULL + 1) >> 32) * table_rows) >> 32);
        const float4 v = tab[r * 32 + lg];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
// random 4-byte words, one per lane, every lane its own random 64-byte line (the bias gathers)
__global__ __launch_bounds__(256) void cal_read_random_4B(const float* __restrict__ tab, size_t words, size_t accesses, float* out) {
    float acc = 0.0f;
    for (size_t a = (size_t)blockIdx.x * 256 + threadIdx.x; a < accesses; a += (size_t)gridDim.x * 256) {
        const size_t w = (size_t)(((mix64(a * 0x9E3779B97F4A7C15ULL + 7) >> 32) * words) >> 32);
        acc += tab[w];
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void cal_write_seq_16B(float4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void cal_write_seq_4B(float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = 1.0f;
}
// one 4-byte word written per 64-byte line (a lane group's single result word per row)
__global__ __launch_bounds__(256) void cal_write_strided_4B_per_64B(float* __restrict__ dst, size_t lines) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < lines; i += (size_t)gridDim.x * 256) dst[i * 16] = 1.0f;
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    float* buf;
    float* out;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 8), blk(256);
    const size_t accesses = (size_t)4 << 20;  // random accesses per launch
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(cal_read_seq_16B, grid, blk, 0, 0, (const float4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(cal_read_seq_8B, grid, blk, 0, 0, (const float2*)buf, bytes / 8, out);
        hipLaunchKernelGGL(cal_read_seq_4B, grid, blk, 0, 0, (const float*)buf, bytes / 4, out);
        hipLaunchKernelGGL(cal_read_rows_512B, grid, blk, 0, 0, (const float4*)buf, bytes / 512, accesses, out);
        hipLaunchKernelGGL(cal_read_random_4B, grid, blk, 0, 0, (const float*)buf, bytes / 4, accesses, out);
        hipLaunchKernelGGL(cal_write_seq_16B, grid, blk, 0, 0, (float4*)buf, bytes / 16);
        hipLaunchKernelGGL(cal_write_seq_4B, grid, blk, 0, 0, buf, bytes / 4);
        hipLaunchKernelGGL(cal_write_strided_4B_per_64B, grid, blk, 0, 0, buf, bytes / 64);
        CHECK(hipDeviceSynchronize());
    }
    printf("{\"buffer_bytes\": %zu, \"random_accesses\": %zu, \"known_bytes\": {\"cal_read_seq_16B\": %zu, \"cal_read_seq_8B\": %zu, \"cal_read_seq_4B\": %zu, "
           "\"cal_read_rows_512B\": %zu, \"cal_read_random_4B_useful\": %zu, \"cal_read_random_4B_lines64\": %zu, \"cal_write_seq_16B\": %zu, "
           "\"cal_write_seq_4B\": %zu, \"cal_write_strided_4B_per_64B_useful\": %zu, \"cal_write_strided_4B_per_64B_lines64\": %zu}}\n",
           bytes, accesses, bytes, bytes, bytes, accesses * 512, accesses * 4, accesses * 64, bytes, bytes, bytes / 64 * 4, bytes);
    return 0;
}
