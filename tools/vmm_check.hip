// tools/vmm_check.hip -- does this box's HIP runtime do virtual memory management (hipMemAddressReserve / hipMemCreate / hipMemMap), at what granularity,
// how long do map / unmap of 1 GB pieces take, and does a kernel read the mapped range at full speed?  (Segments' blocks freed piece by piece
// while a group is built from them need it: DESIGN 8.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_sum(const uint4* p, size_t n, unsigned long long* out)
{
    unsigned long long a = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; a += v.x + v.y + v.z + v.w; }
    atomicAdd(out, a);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    int dev = 0, vmm = 0;
    CK(hipSetDevice(dev));
    CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    printf("virtual memory management supported: %d\n", vmm);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity (recommended): %zu\n", gran);
    const size_t piece = (size_t)1 << 30, npieces = 16, total = piece * npieces;
    void* va = nullptr;
    size_t fa, fb, tt; CK(hipMemGetInfo(&fa, &tt));
    double t = now();
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(npieces);
    for (size_t i = 0; i < npieces; ++i) { CK(hipMemCreate(&h[i], piece, &prop, 0)); CK(hipMemMap((char*)va + i * piece, piece, 0, h[i], 0)); }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    printf("reserve + create + map + access of %zu x 1 GB: %.3f s\n", npieces, now() - t);
    CK(hipMemGetInfo(&fb, &tt)); printf("hipMemGetInfo: free before %.1f GB, after mapping 16 GB %.1f GB\n", fa / 1e9, fb / 1e9);
    CK(hipMemset(va, 1, total));
    unsigned long long* out; CK(hipMalloc(&out, 8)); CK(hipMemset(out, 0, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_sum, dim3(4096), dim3(256), 0, 0, (const uint4*)va, total / 16, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("streaming read of the mapped range: %.0f GB/s\n", total / (ms * 1e-3) / 1e9);
    size_t f0, f1, tot; CK(hipMemGetInfo(&f0, &tot));
    t = now();
    for (size_t i = 0; i < npieces / 2; ++i) { CK(hipMemUnmap((char*)va + i * piece, piece)); CK(hipMemRelease(h[i])); }
    CK(hipMemGetInfo(&f1, &tot));
    printf("unmap + release of the first %zu pieces: %.3f s, free memory %.1f -> %.1f GB\n", npieces / 2, now() - t, f0 / 1e9, f1 / 1e9);
    CK(hipDeviceSynchronize()); CK(hipMemGetInfo(&f1, &tot)); printf("after a device synchronise: %.1f GB\n", f1 / 1e9);
    CK(hipMemset(out, 0, 8));
    hipLaunchKernelGGL(k_sum, dim3(4096), dim3(256), 0, 0, (const uint4*)((char*)va + total / 2), total / 32, out);
    CK(hipDeviceSynchronize());
    printf("the second half is still readable\n");
    void* q = nullptr; t = now(); CK(hipMalloc(&q, total / 2)); printf("hipMalloc of the freed 8 GB: %.3f s\n", now() - t); CK(hipFree(q));
    for (size_t i = npieces / 2; i < npieces; ++i) { CK(hipMemUnmap((char*)va + i * piece, piece)); CK(hipMemRelease(h[i])); }
    CK(hipMemAddressFree(va, total));
    printf("ok\n");
    return 0;
}
