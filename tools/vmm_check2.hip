// tools/vmm_check2.hip -- is memory that hipMemUnmap + hipMemRelease give back REALLY free again (hipMemGetInfo does not say so on this runtime)?
// 270 GB are mapped in 1-GB pieces, half of them released, and 130 GB created again (and, separately, hipMalloc'ed): both must succeed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    CK(hipSetDevice(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t piece = (size_t)1 << 30, n = 270;
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, piece * n, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    size_t got = 0;
    for (size_t i = 0; i < n; ++i) {
        if (hipMemCreate(&h[i], piece, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        CK(hipMemMap((char*)va + i * piece, piece, 0, h[i], 0)); CK(hipMemSetAccess((char*)va + i * piece, piece, &acc, 1));
        got = i + 1;
    }
    size_t f, t; CK(hipMemGetInfo(&f, &t));
    printf("mapped %zu GiB; hipMemGetInfo free %.1f GB\n", got, f / 1e9);
    const size_t half = got / 2;
    for (size_t i = 0; i < half; ++i) { CK(hipMemUnmap((char*)va + i * piece, piece)); CK(hipMemRelease(h[i])); }
    CK(hipMemGetInfo(&f, &t));
    printf("released %zu GiB; hipMemGetInfo free %.1f GB\n", half, f / 1e9);
    size_t again = 0;
    for (size_t i = 0; i < half; ++i) {
        if (hipMemCreate(&h[i], piece, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        CK(hipMemMap((char*)va + i * piece, piece, 0, h[i], 0)); CK(hipMemSetAccess((char*)va + i * piece, piece, &acc, 1));
        again = i + 1;
    }
    printf("created and mapped again: %zu of %zu GiB\n", again, half);
    for (size_t i = 0; i < again; ++i) { CK(hipMemUnmap((char*)va + i * piece, piece)); CK(hipMemRelease(h[i])); }
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, (half - 2) * piece);
    printf("hipMalloc of %zu GiB after the release: %s\n", half - 2, hipGetErrorString(e));
    if (e == hipSuccess) CK(hipFree(q));
    CK(hipMemGetInfo(&f, &t)); printf("hipMemGetInfo free at the end %.1f GB\n", f / 1e9);
    for (size_t i = half; i < got; ++i) { CK(hipMemUnmap((char*)va + i * piece, piece)); CK(hipMemRelease(h[i])); }
    CK(hipMemAddressFree(va, piece * n));
    CK(hipDeviceSynchronize());
    CK(hipMemGetInfo(&f, &t)); printf("everything released, the address range freed: hipMemGetInfo free %.1f GB\n", f / 1e9);
    e = hipMalloc(&q, (size_t)200 << 30);
    printf("hipMalloc of 200 GiB now: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) { CK(hipMemGetInfo(&f, &t)); printf("  free with it %.1f GB\n", f / 1e9); CK(hipFree(q)); CK(hipMemGetInfo(&f, &t)); printf("  free after hipFree %.1f GB\n", f / 1e9); }
    return 0;
}
