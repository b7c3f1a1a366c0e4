// Measured f32 MFMA ceiling of the device: register-only loops of v_mfma_f32_16x16x4_f32 and
// v_mfma_f32_32x32x2_f32 (8 / 4 independent accumulators per wave), W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_ubench.hip -o tools/bin/mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k16(float* out, int iters, float a, float b) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k32(float* out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %.0f MHz\n", p.gcnArchName, cus, p.clockRate / 1000.0);
    float* out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        for (int kind = 0; kind < 2; ++kind) {
            dim3 grid(cus * wps), block(256);  // 4 waves per block = one per SIMD; wps blocks per CU
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (kind == 0) hipLaunchKernelGGL(k16, grid, block, 0, 0, out, iters, 1.0f, 0.5f);
                else hipLaunchKernelGGL(k32, grid, block, 0, 0, out, iters, 1.0f, 0.5f);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = (double)cus * wps * 4 * iters * (kind == 0 ? 8 * 2048.0 : 4 * 4096.0);
            printf("%s  %d wave(s)/SIMD: %.2f ms  %.1f TFLOP/s\n", kind == 0 ? "16x16x4f32" : "32x32x2f32", wps, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
