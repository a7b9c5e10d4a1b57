// Developer micro-benchmark: what the matrix pipe and the vector ALU of a gfx950 SIMD deliver together, from registers
// only (no LDS, no memory) -- the ceiling the split-product kernels are measured against.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_peak.hip -o tools/mfma_peak && ./mfma_peak
// Per wave and iteration: 24 v_mfma_f32_32x32x16_bf16 on 4 accumulator tiles (one 16-deep step of a 64 x 64 wave
// tile, six plane products) plus VALU independent v_fma_f32 instructions (the split arithmetic's stand-in).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int VALU, bool F32>
__global__ __launch_bounds__(256) void peak_kernel(float* out, int iters, float seed) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[i][e] = 0x3f803f80u + threadIdx.x + i;
            b[i][e] = 0x3f803f80u + threadIdx.x * 3 + i;
        }
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = seed + e + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 6; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (F32) {
                        // eight fp32-input MFMAs carry the same 32 x 32 x 16 block; 6 plane products ~ 8 of these: issue 4 per (p, i, j) of 3 p
                        if (p < 3) {
#pragma unroll
                            for (int s = 0; s < 4; ++s)
                                acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[i][s]), __uint_as_float(b[j][s]), acc[i * 2 + j], 0, 0, 0);
                        }
                    } else {
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i * 2 + j], 0, 0, 0);
                    }
                }
            // VALU / 6 independent vector-ALU instructions between the product groups
#pragma unroll
            for (int q = 0; q < VALU / 6; ++q) {
                const int e = (p * (VALU / 6) + q) % 16;
                v[e] = __builtin_fmaf(v[e], 1.0000001f, 0.5f);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int e = 0; e < 16; ++e) s += v[e];
    if (s == 12345.f) out[threadIdx.x] = s;
}

// The same flops per iteration pair with v_mfma_f32_16x16x32_bf16 (MI355X_MICROARCH.md, DVFS give-back item 7: under the
// chip's power limit the 16x16x32 shape delivered ~1.15 x the FLOP/s of 32x32x16 at equal cycles per flop): a 64 x 64 wave
// tile is 16 accumulator tiles of 4 registers, one 32-deep step of the six plane products = 96 MFMAs of 16 cycles, beside
// VALU independent vector instructions.  RANDOM operand bits in both kernels of this comparison (hash of lane and slot).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
template <int VALU, int SHAPE>   // SHAPE 0: 32x32x16, two 16-deep steps (48 MFMAs of 32 cycles); 1: 16x16x32, one 32-deep step (96 of 16)
__global__ __launch_bounds__(256) void shape_kernel(float* out, int iters, float seed) {
    u32x4 a[3][4], b[3][4];   // three planes x four 16-row (or two 32-row) fragments
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // bf16 pairs with random mantissas and exponents near 1 (finite, no denormals)
                a[p][i][e] = (hash32(threadIdx.x * 977u + p * 131u + i * 17u + e) & 0x007f007fu) | 0x3f003f00u;
                b[p][i][e] = (hash32(threadIdx.x * 613u + p * 257u + i * 29u + e + 99u) & 0x007f007fu) | 0x3f003f00u;
            }
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = seed + e + threadIdx.x;
    f32x16 acc32[4];
    f32x4 acc16[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};   // a3b1 a1b3 a2b2 a2b1 a1b2 a1b1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            if constexpr (SHAPE == 0) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc32[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[pa[p]][2 * s + i]),
                                                                                       __builtin_bit_cast(bf16x8, b[pb[p]][2 * s + j]), acc32[i * 2 + j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc16[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[pa[p]][i]), __builtin_bit_cast(bf16x8, b[pb[p]][j]),
                                                                                   acc16[i * 4 + j], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < VALU / 6; ++q) {
                const int e = (p * (VALU / 6) + q) % 16;
                v[e] = __builtin_fmaf(v[e], 1.0000001f, 0.5f);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc32[i][e];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3];
#pragma unroll
    for (int e = 0; e < 16; ++e) s += v[e];
    if (s == 12345.f) out[threadIdx.x] = s;
}
template <int VALU, int SHAPE>
static void run_shape(int blocks_per_cu, int iters) {
    float* out;
    CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL((shape_kernel<VALU, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);   // clock settles under load
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((shape_kernel<VALU, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 48.0 * 2.0 * 32 * 32 * 16;   // per wave-iteration, both shapes
    const double tf = (double)blocks * 4 * iters * flop / ms / 1e9;
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * blocks_per_cu);
    printf("%s random operands  valu/iter %3d  waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s raw (%5.1f algorithmic fp32 = raw / 6)  %6.0f cycles@2.4GHz per wave-iteration (matrix pipe alone: 1536)\n",
           SHAPE == 0 ? "32x32x16 x48" : "16x16x32 x96", VALU, blocks_per_cu, ms, tf, tf / 6.0, cyc);
    CK(hipFree(out));
}

template <int VALU, bool F32>
static void run(int blocks_per_cu, int iters) {
    float* out;
    CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL((peak_kernel<VALU, F32>), dim3(blocks), dim3(256), 0, 0, out, iters / 10, 1.f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((peak_kernel<VALU, F32>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = F32 ? 48.0 : 24.0;
    const double flop_per = F32 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
    const double tf = (double)blocks * 4 * iters * mfma * flop_per / ms / 1e9;
    // cycles a SIMD spends per wave-iteration at 2.4 GHz (waves per SIMD = blocks_per_cu)
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * blocks_per_cu);
    printf("%s  valu/iter %3d  waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s raw   %6.0f cycles@2.4GHz per wave-iteration (matrix pipe alone: %d)\n",
           F32 ? "fp32 x48" : "bf16 x24", VALU, blocks_per_cu, ms, tf, cyc, F32 ? 48 * 64 : 24 * 32);
    CK(hipFree(out));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    run<0, false>(1, iters);
    run<0, false>(2, iters);
    run<96, false>(1, iters);
    run<96, false>(2, iters);
    run<180, false>(1, iters);
    run<180, false>(2, iters);
    run<0, true>(1, iters / 4);
    run<0, true>(2, iters / 4);
    // MFMA shape under the power limit, random operand bits, long launches (the clock settles)
    for (int w = 1; w <= 2; ++w) {
        run_shape<0, 0>(w, 4 * iters);
        run_shape<0, 1>(w, 4 * iters);
        run_shape<360, 0>(w, 4 * iters);
        run_shape<360, 1>(w, 4 * iters);
    }
    return 0;
}
