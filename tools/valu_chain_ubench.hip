// Dependent-issue latency of the vector ALU for a LONE wave on a SIMD (the regime of sbr_wave.hip): cycles per instruction
// of (a) one chain of dependent v_fmac_f32, (b) two interleaved chains, (c) one chain of dependent v_pk_fma_f32,
// (d) v_readlane_b32 + v_fmac_f32 with the scalar operand (distinct scalar registers, reads 16 ahead).
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_chain_ubench tools/valu_chain_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 64
__global__ void k(float* out, long long* cyc, float a, float b) {
    float x = a + threadIdx.x, y = b + threadIdx.x;
    long long t0, t1;
    t0 = clock64();
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(x) : "v"(a));
    }
    t1 = clock64();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    t0 = clock64();
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(x) : "v"(a)); asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(y) : "v"(a)); }
    }
    t1 = clock64();
    if (threadIdx.x == 0) cyc[1] = t1 - t0;
    v2f p = {x, y}, q = {a, b};
    t0 = clock64();
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(p) : "v"(q));
    }
    t1 = clock64();
    if (threadIdx.x == 0) cyc[2] = t1 - t0;
    t0 = clock64();
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            asm volatile("v_readlane_b32 s[%1], %2, %3\n" :: "i"(0), "i"(40 + (j % 16)), "v"(y), "i"(j) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55");
            asm volatile("v_fmac_f32 %0, s[%1], %0" : "+v"(x) : "i"(40 + ((j + 1) % 16)) : "s40");
        }
    }
    t1 = clock64();
    if (threadIdx.x == 0) cyc[3] = t1 - t0;
    out[threadIdx.x] = x + y + p.x + p.y;
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256); hipMalloc(&cyc, 64);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 0.999f, 1.001f);
    long long h[4];
    hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    const double n = 32.0 * REP;
    printf("cycles per instruction, lone wave: dependent v_fmac_f32 %.2f | two interleaved chains %.2f | dependent v_pk_fma_f32 %.2f | readlane+fmac pair %.2f (per pair)\n",
           h[0] / n, h[1] / n, h[2] / n, h[3] / n);
    return 0;
}
