// Developer micro-benchmark: FP32-accurate products on the BF16 matrix pipe.  x = x1 + x2 + x3 (three bf16
// pieces, truncation split: 8 + 8 + 8 mantissa bits), x*y ~ x1y1 + x1y2 + x2y1 + x2y2 + x1y3 + x3y1
// (dropped terms ~2^-24 |xy|), six v_mfma_f32_32x32x16_bf16 per 32x32x16 block versus eight
// v_mfma_f32_32x32x2_f32.  Each wave: 64 x 64 tile, fragments re-read from LDS every step, split in registers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/split_bench.hip -o tools/split_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// three bf16 planes of 8 floats: plane p as 4 dwords (2 bf16 each)
__device__ __forceinline__ void split3(const float (&x)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned a = __float_as_uint(x[2 * i]), b = __float_as_uint(x[2 * i + 1]);
        p1[i] = __builtin_amdgcn_perm(b, a, 0x07060302);           // [hi16(b) : hi16(a)]
        const float ra = x[2 * i] - __uint_as_float(a & 0xFFFF0000u);
        const float rb = x[2 * i + 1] - __uint_as_float(b & 0xFFFF0000u);
        const unsigned a2 = __float_as_uint(ra), b2 = __float_as_uint(rb);
        p2[i] = __builtin_amdgcn_perm(b2, a2, 0x07060302);
        const float sa = ra - __uint_as_float(a2 & 0xFFFF0000u);
        const float sb = rb - __uint_as_float(b2 & 0xFFFF0000u);
        p3[i] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302);
    }
}
__device__ __forceinline__ bf16x8 as_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

template <int MODE>   // 0: fp32 MFMA, 1: bf16 x 6
__global__ __launch_bounds__(256) void k(const float* __restrict__ g, float* __restrict__ out, int iters) {
    __shared__ float lds[2 * 128 * 32];   // A [128][32], B [128][32] fp32 stage image (contents arbitrary)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 2 * 128 * 32; i += 256) lds[i] = g[i];
    __syncthreads();
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const float* la = lds;
    const float* lb = lds + 128 * 32;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {   // two 16-deep steps of a 32-deep stage
            float a[2][8], b[2][8];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm + i * 32 + (lane & 31);
                const int c0 = 4 * s + 2 * (lane >> 5);
                const v4f u = *reinterpret_cast<const v4f*>(la + row * 32 + ((c0 ^ ((row >> 1) & 7)) << 2));
                const v4f v = *reinterpret_cast<const v4f*>(la + row * 32 + (((c0 + 1) ^ ((row >> 1) & 7)) << 2));
                a[i][0] = u.x; a[i][1] = u.y; a[i][2] = u.z; a[i][3] = u.w; a[i][4] = v.x; a[i][5] = v.y; a[i][6] = v.z; a[i][7] = v.w;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn + j * 32 + (lane & 31);
                const int c0 = 4 * s + 2 * (lane >> 5);
                const v4f u = *reinterpret_cast<const v4f*>(lb + row * 32 + ((c0 ^ ((row >> 1) & 7)) << 2));
                const v4f v = *reinterpret_cast<const v4f*>(lb + row * 32 + (((c0 + 1) ^ ((row >> 1) & 7)) << 2));
                b[j][0] = u.x; b[j][1] = u.y; b[j][2] = u.z; b[j][3] = u.w; b[j][4] = v.x; b[j][5] = v.y; b[j][6] = v.z; b[j][7] = v.w;
            }
            if constexpr (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            } else {
                u32x4 a1[2], a2[2], a3[2], b1[2], b2[2], b3[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) split3(a[i], a1[i], a2[i], a3[i]);
#pragma unroll
                for (int j = 0; j < 2; ++j) split3(b[j], b1[j], b2[j], b3[j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x16 c = acc[i][j];
                        // small terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a3[i]), as_bf(b1[j]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a1[i]), as_bf(b3[j]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a2[i]), as_bf(b2[j]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a2[i]), as_bf(b1[j]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a1[i]), as_bf(b2[j]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a1[i]), as_bf(b1[j]), c, 0, 0, 0);
                        acc[i][j] = c;
                    }
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    out[(size_t)blockIdx.x * 256 + t] = sum;
    // tile (0,0) of wave 0 of block 0 for the accuracy check: D[row][col], col = lane & 31, row = (e&3) + 8(e>>2) + 4(lane>>5)
    if (blockIdx.x == 0 && wave == 0)
#pragma unroll
        for (int e = 0; e < 16; ++e) out[(size_t)gridDim.x * 256 + ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[0][0][e];
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int grid = 512;
    std::vector<float> h(2 * 128 * 32);
    srand(1);
    const bool positive = argc > 2;
    const int acc_iters = argc > 3 ? atoi(argv[3]) : 1;
    for (auto& v : h) v = positive ? (float)rand() / RAND_MAX * 2.f + 0.1f : ((float)rand() / RAND_MAX - 0.5f) * 4.f;
    float *g, *out;
    CK(hipMalloc(&g, h.size() * 4));
    CK(hipMalloc(&out, ((size_t)grid * 256 + 1024) * 4));
    CK(hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // reference for one iteration of tile (0,0): sum over k of A[row][k] * B[col][k] with the swizzled image
    std::vector<double> ref(1024, 0.0);
    for (int r = 0; r < 32; ++r)
        for (int c = 0; c < 32; ++c) {
            double s = 0;
            for (int kk = 0; kk < 32; ++kk) {   // logical k sits at chunk (k/4) ^ ((row/2)&7) of its row
                const int pa = (((kk >> 2) ^ ((r >> 1) & 7)) << 2) + (kk & 3), pb = (((kk >> 2) ^ ((c >> 1) & 7)) << 2) + (kk & 3);
                s += (double)h[r * 32 + pa] * (double)h[128 * 32 + c * 32 + pb];
            }
            ref[r * 32 + c] = s;
        }
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, g, out, rep == 0 ? acc_iters : iters);
            else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, g, out, rep == 0 ? acc_iters : iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            if (rep == 0) {   // accuracy of one stage (same k set for every lane: the row / column sums over all 32 k)
                std::vector<float> o(1024);
                CK(hipMemcpy(o.data(), out + (size_t)grid * 256, 4096, hipMemcpyDeviceToHost));
                double mx = 0, sc = 0;
                double bias = 0;
                for (int i = 0; i < 1024; ++i) { const double r = ref[i] * acc_iters; mx = fmax(mx, fabs(o[i] - r)); sc = fmax(sc, fabs(r)); bias += (o[i] - r) / r / 1024; }
                printf("   mean signed rel err %.3e (K = %d)\n", bias, 32 * acc_iters);
                printf("mode %d: max abs err vs float64 %.3e (scale %.3e, rel %.3e)\n", mode, mx, sc, mx / sc);
            } else {
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double flop = 2.0 * 128 * 128 * 32 * (double)iters * grid;
                printf("mode %d (%s): %.3f ms, %.1f TFLOP/s algorithmic\n", mode, mode ? "bf16 x 6" : "fp32 mfma", ms, flop / ms / 1e9);
            }
        }
    }
    return 0;
}
