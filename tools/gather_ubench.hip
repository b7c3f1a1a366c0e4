// Micro-benchmark: what does MI355X deliver for random 4d-byte embedding-row gathers
// (d = 128 -> 512 B rows, table 488 MiB > Infinity Cache)?  Gives the practical ceiling for the
// gather + WARP-score kernel next to the streaming peak.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int D, int ROWS_PER_GROUP, bool STREAM>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ E, const uint32_t* __restrict__ idx,
                                                     size_t n, const float* __restrict__ Hs, float* __restrict__ out, float* __restrict__ dHs) {
    constexpr int L = D / 4, GPW = 64 / L;
    const int lane = threadIdx.x & 63, lg = lane % L, grp = lane / L;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    float acc = 0.f;
    for (size_t base = wave * GPW * ROWS_PER_GROUP; base < n; base += nwaves * GPW * ROWS_PER_GROUP) {
        float4 v[ROWS_PER_GROUP];
#pragma unroll
        for (int j = 0; j < ROWS_PER_GROUP; ++j) {
            size_t r = base + (size_t)j * GPW + grp;
            if (r >= n) r = n - 1;
            v[j] = *reinterpret_cast<const float4*>(E + (size_t)idx[r] * D + 4 * lg);
        }
#pragma unroll
        for (int j = 0; j < ROWS_PER_GROUP; ++j) {
            size_t r = base + (size_t)j * GPW + grp;
            if (r >= n) r = n - 1;
            if (STREAM) {
                float4 h = *reinterpret_cast<const float4*>(Hs + r * D + 4 * lg);
                float4 o = make_float4(v[j].x * h.x, v[j].y * h.y, v[j].z * h.z, v[j].w * h.w);
                *reinterpret_cast<float4*>(dHs + r * D + 4 * lg) = o;
            }
            acc += v[j].x + v[j].y + v[j].z + v[j].w;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int D, int RPG, bool STREAM>
void run(const char* name, const float* E, const uint32_t* idx, size_t n, const float* H, float* out, float* dH, int blocks) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((gather_kernel<D, RPG, STREAM>), dim3(blocks), dim3(256), 0, 0, E, idx, n, H, out, dH);
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((gather_kernel<D, RPG, STREAM>), dim3(blocks), dim3(256), 0, 0, E, idx, n, H, out, dH);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    double gathered = (double)n * D * 4;
    double total = gathered + (STREAM ? 2.0 * n * D * 4 : 0.0) + 4.0 * n;
    printf("%-44s blocks %5d: %8.1f us  gather %7.1f GB/s  total %7.1f GB/s\n", name, blocks, ms * 1e3, gathered / ms / 1e6, total / ms / 1e6);
}

int main() {
    const size_t I = 1000000, D = 128, n = 4u << 20;  // 4M gathered rows = 2 GiB of row reads per launch
    float *E, *H, *dH, *out;
    uint32_t* idx;
    CK(hipMalloc(&E, I * D * 4)); CK(hipMalloc(&H, n * D * 4)); CK(hipMalloc(&dH, n * D * 4)); CK(hipMalloc(&out, 64));
    CK(hipMalloc(&idx, n * 4));
    CK(hipMemset(E, 0, I * D * 4)); CK(hipMemset(H, 0, n * D * 4));
    std::vector<uint32_t> h(n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)((s >> 32) * I >> 32); }
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
    for (int blocks : {1024, 2048, 4096, 8192}) {
        run<128, 1, false>("random 512B rows, 1 row/group in flight", E, idx, n, H, out, dH, blocks);
        run<128, 2, false>("random 512B rows, 2 rows/group in flight", E, idx, n, H, out, dH, blocks);
        run<128, 4, false>("random 512B rows, 4 rows/group in flight", E, idx, n, H, out, dH, blocks);
        run<128, 8, false>("random 512B rows, 8 rows/group in flight", E, idx, n, H, out, dH, blocks);
        run<128, 2, true>("random rows + stream read h + stream write", E, idx, n, H, out, dH, blocks);
        run<128, 4, true>("random rows(4) + stream read h + write", E, idx, n, H, out, dH, blocks);
    }
    // sequential rows for reference (idx = identity mod I)
    for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)(i % I);
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
    run<128, 4, false>("sequential rows (streaming reference)", E, idx, n, H, out, dH, 4096);
    return 0;
}
