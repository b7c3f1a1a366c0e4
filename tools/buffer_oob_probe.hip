// buffer_oob_probe.hip — how gfx950 range-checks raw buffer accesses (stride 0): which of the instruction offset,
// the per-lane VGPR offset and the scalar offset take part in the comparison against num_records.  The forward
// kernel's branch-free stores rely on "offset >= num_records => the store is dropped".
//   hipcc --offload-arch=gfx950 -O3 tools/buffer_oob_probe.hip -o tools/bin/buffer_oob_probe && tools/bin/buffer_oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void probe(unsigned* buf, int num_records, int mode) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, (short)0, num_records, 0x00020000);
    const unsigned lane = threadIdx.x;
    if (mode == 0) __builtin_amdgcn_raw_buffer_store_b32(0xA0000000u | lane, rs, lane * 32u, 0, 0);          // VGPR offset crosses num_records
    if (mode == 1) __builtin_amdgcn_raw_buffer_store_b32(0xB0000000u | lane, rs, lane * 4u, 2048, 0);        // scalar offset beyond num_records
    if (mode == 2) __builtin_amdgcn_raw_buffer_store_b32(0xC0000000u | lane, rs, lane * 16u + 512u, 0, 0);  // VGPR offset, constant part folded by the compiler
    if (mode == 3) {                                                                                      // loads: out of range reads 0
        unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 32u, 0, 0);
        buf[2048 + lane] = v;
    }
}

int main() {
    unsigned* d;
    const int words = 4096;
    CHECK(hipMalloc(&d, words * 4));
    std::vector<unsigned> h(words);
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<unsigned> init(words, 0x11111111u);
        CHECK(hipMemcpy(d, init.data(), words * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 1024, mode);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d, words * 4, hipMemcpyDeviceToHost));
        int in_range = 0, beyond = 0;
        for (int i = 0; i < 2048; ++i) {
            if (h[i] != 0x11111111u) { if (i * 4 < 1024) ++in_range; else ++beyond; }
        }
        if (mode < 3) printf("mode %d: words written inside num_records %d, beyond num_records %d\n", mode, in_range, beyond);
        else {
            int zeros = 0, data = 0;
            for (int l = 0; l < 64; ++l) { if (h[2048 + l] == 0) ++zeros; else if (h[2048 + l] == 0x11111111u) ++data; }
            printf("mode 3: loads returning data %d, returning 0 %d\n", data, zeros);
        }
    }
    return 0;
}
