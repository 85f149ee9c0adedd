// tools/lds_dma_check.hip -- what global_load_lds_dword / _dwordx4 do on this chip: lane l's piece lands at M0 + l * size (4 / 16 bytes),
// inactive lanes write nothing, the data is there after s_waitcnt vmcnt(0).  k_probe_pgroup's prefetch of the next round's keys and line
// heads rests on exactly this.  hipcc --offload-arch=gfx950 -O2 tools/lds_dma_check.hip -o tools/lds_dma_check.bin && tools/lds_dma_check.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
__global__ void k(const uint8_t* src, const uint32_t* idx, uint32_t* out4, uint32_t* out1)
{
    extern __shared__ __align__(16) uint8_t dyn[];
    const uint32_t tid = threadIdx.x, wave0 = tid & ~63u;
    for (uint32_t i = tid; i < 256u * 5u; i += 256u) reinterpret_cast<uint32_t*>(dyn)[i] = 0xDEAD0000u + i;
    __syncthreads();
    const uint32_t m4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(dyn + wave0 * 16u));
    const uint32_t m1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(dyn + 4096u + wave0 * 4u));
    const __attribute__((address_space(1))) uint8_t* p = (const __attribute__((address_space(1))) uint8_t*)(src + (size_t)idx[tid] * 128u);
    if ((tid % 5u) != 3u) {                               // (every fifth lane stays out)
        uint32_t sv;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                     "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %4, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(sv) : "s"(m4), "s"(m1), "v"(p), "v"(p + 20) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint4 v = *reinterpret_cast<const uint4*>(dyn + tid * 16u);
    out4[tid * 4 + 0] = v.x; out4[tid * 4 + 1] = v.y; out4[tid * 4 + 2] = v.z; out4[tid * 4 + 3] = v.w;
    out1[tid] = reinterpret_cast<const uint32_t*>(dyn + 4096u)[tid];
}
int main()
{
    const uint32_t nlines = 4096;
    std::vector<uint32_t> h(nlines * 32), idx(256);
    for (uint32_t i = 0; i < h.size(); ++i) h[i] = i * 2654435761u;
    for (uint32_t i = 0; i < 256; ++i) idx[i] = (i * 977u + 13u) % nlines;
    uint8_t* d; uint32_t *di, *o4, *o1;
    hipMalloc(&d, h.size() * 4); hipMalloc(&di, 1024); hipMalloc(&o4, 4096); hipMalloc(&o1, 1024);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(di, idx.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, d, di, o4, o1);
    std::vector<uint32_t> r4(1024), r1(256);
    if (hipMemcpy(r4.data(), o4, 4096, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(r1.data(), o1, 1024, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("hip error\n"); return 2; }
    int bad = 0;
    for (uint32_t t = 0; t < 256; ++t) {
        const bool in = (t % 5u) != 3u;
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t want = in ? h[idx[t] * 32 + j] : 0xDEAD0000u + t * 4 + j;
            if (r4[t * 4 + j] != want) { if (bad < 8) std::printf("x4 lane %u word %u: %08x, want %08x\n", t, j, r4[t * 4 + j], want); ++bad; }
        }
        const uint32_t want1 = in ? h[idx[t] * 32 + 5] : 0xDEAD0000u + 1024 + t;
        if (r1[t] != want1) { if (bad < 8) std::printf("x1 lane %u: %08x, want %08x\n", t, r1[t], want1); ++bad; }
    }
    std::printf(bad ? "lds dma check: %d MISMATCHES\n" : "lds dma check ok: a gather into LDS lands at M0 + lane x size, inactive lanes write nothing\n", bad);
    return bad ? 1 : 0;
}
