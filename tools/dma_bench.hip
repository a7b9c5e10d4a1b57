// Developer micro-benchmark: round-trip time of a batch of n 16-byte-per-lane loads issued by every wave of a
// workgroup and awaited together -- LDS-DMA (global_load_lds_dwordx4, the stage loader of gemm.h) against plain
// global_load_dwordx4 into registers -- from an L2-resident source.  time(n) ~ latency + n * per-instruction cost.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dma_bench.hip -o tools/dma_bench && ./dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_wave_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_wave_addr)
        : "memory");
}

// mode 0: LDS-DMA ; mode 1: global_load_dwordx4 to registers.  N loads per wave and round; row pitch `pitch` floats.
template <int N, int MODE>
__global__ __launch_bounds__(256) void dma_kernel(const float* __restrict__ src, int64_t region_floats, int pitch, int iters, unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    // a wave's load n covers 8 rows x 128 bytes (the MMAJOR stage pattern): lane -> row = lane / 8, chunk = lane % 8
    const float* base = src + ((int64_t)blockIdx.x * 4096) % region_floats;
    const unsigned l0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(lds + wave * N * 256));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned long long t0 = 0, total = 0;
    for (int it = 0; it < iters + 1; ++it) {
        if (it == 1) t0 = __builtin_amdgcn_s_memtime();
        const float* p = base + (int64_t)((it * 7 + wave * N) % 64) * 8 * pitch;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int n = 0; n < N; ++n) glds16(p + (int64_t)(n * 8 + lane / 8) * pitch + (lane % 8) * 4, l0 + n * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            float4 v[N];
#pragma unroll
            for (int n = 0; n < N; ++n) v[n] = *reinterpret_cast<const float4*>(p + (int64_t)(n * 8 + lane / 8) * pitch + (lane % 8) * 4);
#pragma unroll
            for (int n = 0; n < N; ++n) { acc.x += v[n].x; acc.y += v[n].y; acc.z += v[n].z; acc.w += v[n].w; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    total = __builtin_amdgcn_s_memtime() - t0;
    if (t == 0) cyc[blockIdx.x] = total;
    if (acc.x + acc.y + acc.z + acc.w + lds[t] == 12345.678f) sink[t] = acc.x;
}

template <int N, int MODE>
static void run(const float* src, int64_t region, int pitch, int blocks, unsigned long long* cyc_d, float* sink) {
    const int iters = 2000;
    hipLaunchKernelGGL((dma_kernel<N, MODE>), dim3(blocks), dim3(256), 4 * N * 1024, 0, src, region, pitch, iters, cyc_d, sink);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((dma_kernel<N, MODE>), dim3(blocks), dim3(256), 4 * N * 1024, 0, src, region, pitch, iters, cyc_d, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    CK(hipMemcpy(h.data(), cyc_d, blocks * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= blocks;
    const double ns = ms * 1e6 / iters;
    printf("%s  n=%2d loads/wave  %4d workgroups : %7.0f ns per round (%5.0f shader cycles) -> %6.1f ns per load ; %6.1f GB/s per CU ; %5.2f TB/s chip\n",
           MODE == 0 ? "lds-dma " : "register", N, blocks, ns, mean / iters, ns / N, 4.0 * N * 1024 / ns * (blocks > 256 ? 2 : 1), (double)blocks * 4 * N * 1024 / ns / 1e3);
}

int main() {
    const int64_t region = 16 << 20;   // 64 MiB of floats' worth of addresses is too much for L2: keep the footprint at 4 MiB
    float* src;
    CK(hipMalloc(&src, (size_t)region * 4 + (1 << 22)));
    CK(hipMemset(src, 0, (size_t)region * 4 + (1 << 22)));
    unsigned long long* cyc;
    float* sink;
    CK(hipMalloc(&cyc, 8192 * 8));
    CK(hipMalloc(&sink, 4096));
    const int64_t small = 1 << 20;   // 4 MiB footprint: L2 / MALL resident
    for (int blocks : {256, 512}) {
        run<1, 0>(src, small, 512, blocks, cyc, sink);
        run<2, 0>(src, small, 512, blocks, cyc, sink);
        run<4, 0>(src, small, 512, blocks, cyc, sink);
        run<6, 0>(src, small, 512, blocks, cyc, sink);
        run<8, 0>(src, small, 512, blocks, cyc, sink);
        run<12, 0>(src, small, 512, blocks, cyc, sink);
        run<1, 1>(src, small, 512, blocks, cyc, sink);
        run<2, 1>(src, small, 512, blocks, cyc, sink);
        run<4, 1>(src, small, 512, blocks, cyc, sink);
        run<8, 1>(src, small, 512, blocks, cyc, sink);
        run<12, 1>(src, small, 512, blocks, cyc, sink);
    }
    return 0;
}
