// Does VALU work of the SAME wave hide in the shadow of its MFMAs?  Per loop iteration: 16 independent
// v_mfma_f32_16x16x4_f32 (512 cycles of matrix pipe) and/or NV dependent-free fma chains on other
// registers.  Modes: 0 = MFMA only, 1 = VALU only, 2 = both in one basic block (compiler interleaves).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int j = 0; j < NV / 8; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int NV>
static float run(float* out, int cus, int wps, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NV>), dim3(cus * wps), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, iters = 20000;
    float* out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    for (int wps = 1; wps <= 4; wps *= 2) {
        printf("%d wave(s)/SIMD, per iteration 16 MFMA (512 pipe cycles) + NV fma:\n", wps);
        printf("  MFMA only            %.2f ms\n", run<0, 0>(out, cus, wps, iters));
        printf("  NV=48   VALU only %.2f  both %.2f ms\n", run<1, 48>(out, cus, wps, iters), run<2, 48>(out, cus, wps, iters));
        printf("  NV=96   VALU only %.2f  both %.2f ms\n", run<1, 96>(out, cus, wps, iters), run<2, 96>(out, cus, wps, iters));
        printf("  NV=192  VALU only %.2f  both %.2f ms\n", run<1, 192>(out, cus, wps, iters), run<2, 192>(out, cus, wps, iters));
    }
    return 0;
}
