// tools/atomic_rate.hip -- how fast does ONE address take returning global atomics on this GPU?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_rate tools/atomic_rate.hip && /tmp/atomic_rate
// 8192 waves x 256 atomicAdd (lane 0 of each wave, result broadcast) on `stride` distinct cache lines.  MI355X: one line
// takes an atomic every ~12 ns (2.1 M atomics: 25 ms on 1 line, 3.2 ms on 8, 0.55 ms on 64, 0.11 ms on 1024) -- the reason
// the hit buffer is reserved once per >= 512-record flush, the deferred pass counts before it writes, and the probe
// kernel's statistics go to 64 copies of their slots.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* ctr, unsigned long long* out, int iters, int stride)
{
    const int lane = threadIdx.x & 63;
    unsigned long long acc = 0;
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int i = 0; i < iters; ++i) {
        unsigned long long g = 0;
        if (lane == 0) g = atomicAdd(&ctr[(size_t)(w % stride) * 16], 100ull);
        g = __shfl(g, 0);
        acc += g;
    }
    if (lane == 0) out[w] = acc;
}
int main()
{
    unsigned long long *ctr, *out;
    (void)hipMalloc(&ctr, 1 << 20); (void)hipMalloc(&out, 8 << 20); (void)hipMemset(ctr, 0, 1 << 20);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int stride : {1, 2, 8, 64, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(1024), dim3(512), 0, 0, ctr, out, 256, stride);   // 8192 waves x 256 atomics = 2.1 M
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("stride %d: %.3f ms for 2097152 atomics = %.1f ns each\n", stride, ms, ms * 1e6 / 2097152.0);
        }
    }
    return 0;
}
