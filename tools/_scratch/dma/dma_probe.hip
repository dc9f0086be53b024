// LDS-DMA throughput probe (tools/_scratch, not product): how many GB/s per CU can global -> LDS DMA sustain from L2?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
template <int DEPTH>
__device__ __forceinline__ void waitv() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory"); }

// each wave: iters x [issue one 1 KiB piece], keeping DEPTH in flight.  src rows of 128 B at `stride` bytes.
template <int DEPTH, bool BUF>
__global__ __launch_bounds__(1024) void probe(const unsigned char* src, size_t region, int stride, int iters, int lds_per_wave, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const unsigned char* base = src + (size_t)blockIdx.x * region;
    unsigned char* dst0 = smem + wave * lds_per_wave;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)region, 0x00020000);
    const int row = lane >> 3, pos = lane & 7;
    const unsigned rows_total = (unsigned)(region / stride);
    unsigned rowbase = wave * 8;
    int slot = 0;
    const int nslots = lds_per_wave / 1024;
    for (int it = 0; it < iters; ++it) {
        const unsigned rr = (rowbase + row) % rows_total;
        const unsigned off = rr * (unsigned)stride + ((pos ^ row) << 4);
        unsigned char* d = dst0 + slot * 1024;
        if (BUF) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)d, 16, off, 0, 0, 0);
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off), (__attribute__((address_space(3))) void*)d, 16, 0, 0);
        waitv<DEPTH>();
        rowbase += nw * 8;
        slot = slot + 1 == nslots ? 0 : slot + 1;
    }
    waitv<0>();
    __syncthreads();
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = ((unsigned*)smem)[lane];
}
// register staging: UNROLL loads in flight per wave (global_load_dwordx4), each then stored with ds_write_b128
template <int UNROLL>
__global__ __launch_bounds__(1024) void probe_reg(const unsigned char* src, size_t region, int stride, int iters, int lds_per_wave, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const unsigned char* base = src + (size_t)blockIdx.x * region;
    unsigned char* dst0 = smem + wave * lds_per_wave;
    const int row = lane >> 3, pos = lane & 7;
    const unsigned rows_total = (unsigned)(region / stride);
    unsigned rowbase = wave * 8;
    const int nslots = lds_per_wave / 1024;
    int slot = 0;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    for (int it = 0; it < iters; it += UNROLL) {
        u4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const unsigned rr = (rowbase + u * nw * 8 + row) % rows_total;
            v[u] = *(const u4*)(base + rr * (unsigned)stride + ((pos ^ row) << 4));
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            *(u4*)(dst0 + slot * 1024 + lane * 16) = v[u];
            slot = slot + 1 == nslots ? 0 : slot + 1;
        }
        rowbase += UNROLL * nw * 8;
    }
    __syncthreads();
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = ((unsigned*)smem)[lane];
}
template <int UNROLL>
float run_reg(const unsigned char* src, size_t region, int stride, int iters, int waves, int grid, unsigned* sink) {
    const int lds_per_wave = (160 * 1024 / waves) / 1024 * 1024;
    hipFuncSetAttribute((const void*)&probe_reg<UNROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe_reg<UNROLL>), dim3(grid), dim3(waves * 64), lds_per_wave * waves, 0, src, region, stride, iters, lds_per_wave, sink);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((probe_reg<UNROLL>), dim3(grid), dim3(waves * 64), lds_per_wave * waves, 0, src, region, stride, iters, lds_per_wave, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
template <int DEPTH, bool BUF>
float run(const unsigned char* src, size_t region, int stride, int iters, int waves, int grid, unsigned* sink) {
    const int lds_per_wave = (160 * 1024 / waves) / 1024 * 1024;
    hipFuncSetAttribute((const void*)&probe<DEPTH, BUF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<DEPTH, BUF>), dim3(grid), dim3(waves * 64), lds_per_wave * waves, 0, src, region, stride, iters, lds_per_wave, sink);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((probe<DEPTH, BUF>), dim3(grid), dim3(waves * 64), lds_per_wave * waves, 0, src, region, stride, iters, lds_per_wave, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
int main() {
    const int grid = 256;
    unsigned char* src; unsigned* sink;
    const size_t total = 256ull << 20;
    hipMalloc(&src, total); hipMemset(src, 1, total); hipMalloc(&sink, 4096 * 4);
    const int iters = 2000;
    printf("# LDS-DMA probe: 256 WGs, each wave issues %d x 1 KiB pieces (8 rows x 128 B); GB/s per CU and TB/s aggregate\n", iters);
    for (size_t region : {128u << 10}) {
        for (int stride : {640}) {
            for (int waves : {4, 8, 16}) {
                double bytes = (double)grid * waves * iters * 1024.0;
                float t2 = run<2, false>(src, region, stride, iters, waves, grid, sink);
                float t6 = run<6, false>(src, region, stride, iters, waves, grid, sink);
                float t12 = run<12, false>(src, region, stride, iters, waves, grid, sink);
                float t24 = run<24, false>(src, region, stride, iters, waves, grid, sink);
                float b12 = run<12, true>(src, region, stride, iters, waves, grid, sink);
                float r4 = run_reg<4>(src, region, stride, iters, waves, grid, sink);
                float r8 = run_reg<8>(src, region, stride, iters, waves, grid, sink);
                printf("   register staging (global_load_dwordx4 + ds_write_b128): unroll4 %6.1f  unroll8 %6.1f GB/s/CU\n", bytes / r4 / 1e6 / 256, bytes / r8 / 1e6 / 256);
                printf("region %4zu KB stride %4d waves %2d: depth2 %6.1f  depth6 %6.1f  depth12 %6.1f  depth24 %6.1f  buf12 %6.1f GB/s/CU  (max %.1f TB/s)\n",
                       region >> 10, stride, waves, bytes / t2 / 1e6 / 256, bytes / t6 / 1e6 / 256, bytes / t12 / 1e6 / 256, bytes / t24 / 1e6 / 256,
                       bytes / b12 / 1e6 / 256, bytes / (t24 < t12 ? t24 : t12) / 1e9);
                fflush(stdout);
            }
        }
    }
    return 0;
}
