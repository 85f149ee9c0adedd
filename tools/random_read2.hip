// tools/random_read2.hip -- how fast can the chip read RANDOM blocks of 64..1024 bytes when the kernel keeps many loads in
// flight?  (fpx_measure_bandwidth's random kernel issues one dependent-free load per step but only ~8 waves x 1 load deep;
// here every lane group streams UNROLL independent blocks per iteration.)  Decides whether fetching only PART of a 512-B
// index block could pay.  Build: hipcc --offload-arch=gfx950 -O3 tools/random_read2.hip -o /tmp/rr2 ; run: /tmp/rr2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int LANES, int UNROLL>      // LANES x 16 B per block
__global__ __launch_bounds__(256) void k_rr(const uint8_t* __restrict__ src, uint64_t nblocks, uint32_t stride, uint32_t iters, unsigned long long* sink)
{
    const uint32_t lane = threadIdx.x % LANES;
    const uint64_t grp = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    uint64_t x = grp * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
            const uint64_t b = x % nblocks;
            v[u] = *reinterpret_cast<const uint4*>(src + b * stride + lane * 16u);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x9e3779b9u) atomicAdd(sink, 1ull);
}

template <int LANES, int UNROLL>
static void run(const uint8_t* buf, size_t bytes, uint32_t stride, unsigned long long* sink)
{
    const uint32_t iters = 32, grid = 256 * 32;
    const uint64_t nblocks = bytes / stride;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_rr<LANES, UNROLL>), dim3(grid), dim3(256), 0, 0, buf, nblocks, stride, iters, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double nb = (double)grid * 256 / LANES * iters * UNROLL;
    printf("{\"read_bytes\": %d, \"stride\": %u, \"unroll\": %d, \"Gblocks_per_s\": %.2f, \"GBs\": %.1f}\n", LANES * 16, stride, UNROLL,
           nb / (ms * 1e-3) / 1e9, nb * LANES * 16 / (ms * 1e-3) / 1e9);
}

int main()
{
    const size_t bytes = 64ull << 30;
    uint8_t* buf; unsigned long long* sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemset(buf, 0x5a, bytes)); CHECK(hipMemset(sink, 0, 8));
    // blocks of 512 B (the index's), of which the first 64 .. 512 bytes are read
    run<4, 8>(buf, bytes, 512, sink);
    run<8, 8>(buf, bytes, 512, sink);
    run<12, 4>(buf, bytes, 512, sink);    // 192 B: 12 lanes (groups do not tile a wave evenly: 5 groups of 12 + 4 idle lanes)
    run<16, 8>(buf, bytes, 512, sink);
    run<32, 4>(buf, bytes, 512, sink);
    run<32, 8>(buf, bytes, 512, sink);
    // dense small blocks
    run<4, 8>(buf, bytes, 64, sink);
    run<8, 8>(buf, bytes, 128, sink);
    run<16, 8>(buf, bytes, 256, sink);
    run<64, 4>(buf, bytes, 1024, sink);
    return 0;
}
