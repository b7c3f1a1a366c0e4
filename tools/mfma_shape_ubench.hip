// Would a v_mfma_f32_32x32x2_f32 form of the recurrent kernels hide more of the cell mathematics than v_mfma_f32_16x16x4_f32?
//   Per loop iteration: 512 pipe cycles of f32 MFMA in one of the two shapes — 16 x 16x16x4 (32 cycles
// each, four accumulators of 4 registers) or 8 x 32x32x2 (64 cycles each, four accumulators of 16) — and NV independent v_fma_f32 on
// other registers, in ONE basic block (the compiler interleaves them), at 1 / 2 / 4 waves per SIMD.  Reports ms for 20 000
// iterations on every CU and the cycles per iteration at the delivered clock (s_memtime deltas of wave 0).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_ubench.hip -o tools/bin/mfma_shape_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SHAPE, int NV>  // SHAPE 0: 16x16x4, 1: 32x32x2, 2: no MFMA
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float a, float b) {
    f32x4 acc4[4];
    f32x16 acc16[4];
    for (int i = 0; i < 4; ++i) {
        acc4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < 16; ++q) acc16[i][q] = 0.f;
    }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
    float av = a + threadIdx.x, bv = b;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc4[i], 0, 0, 0);
        } else if (SHAPE == 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc16[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc16[i], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NV / 8; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) {
        s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
        for (int q = 0; q < 16; ++q) s += acc16[i][q];
    }
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int SHAPE, int NV>
static void run(float* out, long long* cyc, int cus, int wps, int iters, const char* label) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, NV>), dim3(cus * wps), dim3(256), 0, 0, out, cyc, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("| %d | %s | %d | %.2f | %.0f | %.2f |\n", wps, label, NV, ms, (double)c / iters, (double)c / 1e6 / ms);
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, iters = 20000;
    float* out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    long long* cyc; CK(hipMalloc(&cyc, 8));
    printf("| waves / SIMD | MFMA shape (512 pipe cycles per iteration) | independent v_fma_f32 per iteration | ms (20 000 iterations, every CU) | clock64 ticks per iteration (wave 0) | ticks per us |\n|---|---|---|---|---|---|\n");
    for (int wps = 1; wps <= 4; wps *= 2) {
        run<0, 0>(out, cyc, cus, wps, iters, "16x16x4");   run<1, 0>(out, cyc, cus, wps, iters, "32x32x2");
        run<2, 64>(out, cyc, cus, wps, iters, "none");
        run<0, 64>(out, cyc, cus, wps, iters, "16x16x4");  run<1, 64>(out, cyc, cus, wps, iters, "32x32x2");
        run<2, 128>(out, cyc, cus, wps, iters, "none");
        run<0, 128>(out, cyc, cus, wps, iters, "16x16x4"); run<1, 128>(out, cyc, cus, wps, iters, "32x32x2");
        run<2, 256>(out, cyc, cus, wps, iters, "none");
        run<0, 256>(out, cyc, cus, wps, iters, "16x16x4"); run<1, 256>(out, cyc, cus, wps, iters, "32x32x2");
    }
    return 0;
}
