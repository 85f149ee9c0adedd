// tools/random_lines.hip -- the rate at which this chip serves RANDOM 128-byte lines, one line per LANE (16 bytes of it: what a probe's
// first load is), as a function of the FOOTPRINT the lines are spread over.  k_search_query reads its queries' lines anywhere in the packed
// group (137 GB of lines + 9 GB of `ext` for the 100 M index); round 2's 47 G lines/s (profiles/r02_random_read_rates.txt) was measured
// over 8 GB.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/random_lines.bin tools/random_lines.hip;  run: tools/random_lines.bin [GB ...]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_random_lines(const uint4* __restrict__ buf, uint64_t nlines, uint32_t steps, uint64_t seed, uint32_t* sink)
{
    uint64_t x = seed ^ ((uint64_t)(blockIdx.x * 256u + threadIdx.x) * 0x9E3779B97F4A7C15ull);
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
            const uint64_t line = ((x * 0x2545F4914F6CDD1Dull) >> 11) % nlines;
            v[u] = buf[line * 8u];                       // the first 16 bytes of a 128-byte line
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].w;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char** argv)
{
    std::vector<double> gbs;
    for (int i = 1; i < argc; ++i) gbs.push_back(atof(argv[i]));
    if (gbs.empty()) gbs = {1, 8, 32, 137, 146};
    double maxg = 0;
    for (double g : gbs) maxg = g > maxg ? g : maxg;
    uint4* buf = nullptr; uint32_t* sink = nullptr;
    const size_t bytes = (size_t)(maxg * (1ull << 30));
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (double g : gbs) {
        const uint64_t nlines = (uint64_t)(g * (1ull << 30)) / 128u;
        for (int U : {1, 4, 8}) {
            for (int wgs_per_cu : {4, 8}) {
                const uint32_t grid = 256u * wgs_per_cu, steps = 2048 / U;
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0, 0));
                    if (U == 1) hipLaunchKernelGGL(k_random_lines<1>, dim3(grid), dim3(256), 0, 0, buf, nlines, steps, 77ull + rep, sink);
                    else if (U == 4) hipLaunchKernelGGL(k_random_lines<4>, dim3(grid), dim3(256), 0, 0, buf, nlines, steps, 77ull + rep, sink);
                    else hipLaunchKernelGGL(k_random_lines<8>, dim3(grid), dim3(256), 0, 0, buf, nlines, steps, 77ull + rep, sink);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                }
                const double lines = (double)grid * 256.0 * steps * U;
                printf("{\"footprint_GB\": %.0f, \"loads_in_flight_per_lane\": %d, \"waves_per_simd\": %d, \"G_lines_per_s\": %.2f, \"GBs_at_128B\": %.0f}\n",
                       g, U, wgs_per_cu, lines / (ms * 1e-3) / 1e9, lines * 128.0 / (ms * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
