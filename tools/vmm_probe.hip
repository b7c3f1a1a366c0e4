// Probe: HIP virtual-memory management on the target (hipMemAddressReserve / hipMemCreate / hipMemMap /
// hipMemSetAccess) — one contiguous virtual range backed by several physical allocations, the mechanism
// behind the partitioned item table.  hipcc --offload-arch=gfx950 tools/vmm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(float* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (float)i; }
int main() {
    int nd = 0; CK(hipGetDeviceCount(&nd)); printf("devices %d\n", nd);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    size_t grec = 0; CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity min %zu recommended %zu\n", gran, grec);
    const int parts = 3; size_t part = gran * 2, total = part * parts;
    void* base = nullptr; CK(hipMemAddressReserve(&base, total, gran, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(parts);
    for (int i = 0; i < parts; ++i) {
        prop.location.id = i % nd;
        CK(hipMemCreate(&h[i], part, &prop, 0));
        CK(hipMemMap((char*)base + i * part, part, 0, h[i], 0));
    }
    std::vector<hipMemAccessDesc> acc(nd);
    for (int d = 0; d < nd; ++d) { acc[d].location.type = hipMemLocationTypeDevice; acc[d].location.id = d; acc[d].flags = hipMemAccessFlagsProtReadWrite; }
    CK(hipMemSetAccess(base, total, acc.data(), nd));
    size_t n = total / 4;
    fill<<<(unsigned)((n + 255) / 256), 256>>>((float*)base, n); CK(hipDeviceSynchronize());
    std::vector<float> host(n); CK(hipMemcpy(host.data(), base, total, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += host[i] != (float)i;
    printf("mismatches %zu of %zu\n", bad, n);
    CK(hipMemset(base, 0, total)); CK(hipMemcpy(base, host.data(), total, hipMemcpyHostToDevice));
    for (int i = 0; i < parts; ++i) { CK(hipMemUnmap((char*)base + i * part, part)); CK(hipMemRelease(h[i])); }
    CK(hipMemAddressFree(base, total));
    printf("VMM OK\n"); return 0;
}
