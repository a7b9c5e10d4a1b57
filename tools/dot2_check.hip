// Developer check (not part of the product build): is v_dot2c_f32_bf16 an exact "x - bf16_hi(x)" and what does it cost?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dot2_check.hip -o tools/dot2_check
// The split of gemm.h (split3) forms the residual of a truncated bf16 piece with v_and_b32 + v_sub_f32; the dot product
// D = D + A.lo * B.lo + A.hi * B.hi with B = (-1, 0) or (0, -1) takes the piece straight out of the packed plane dword.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ float res_lo(unsigned packed, float x) { return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, packed), __builtin_bit_cast(bf2, 0x0000BF80u), x, false); }
__device__ __forceinline__ float res_hi(unsigned packed, float x) { return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, packed), __builtin_bit_cast(bf2, 0xBF800000u), x, false); }

__global__ void check(const float* x, int n, unsigned* bad, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    const unsigned p1 = __builtin_amdgcn_perm(ub, ua, 0x07060302);
    const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
    const float da = res_lo(p1, a), db = res_hi(p1, b);
    if (__float_as_uint(ra) != __float_as_uint(da) || __float_as_uint(rb) != __float_as_uint(db)) {
        const unsigned k = atomicAdd(bad, 1u);
        if (k < 8) { out[4 * k] = a; out[4 * k + 1] = ra; out[4 * k + 2] = da; out[4 * k + 3] = b; }
    }
}
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void rate(float* o, int iters) {
    float a = threadIdx.x * 1.0001f + 1.f, b = a * 1.37f, c = a * 0.77f, d = a * 1.91f;
    unsigned p = __float_as_uint(a);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {   // and + sub
                a = a - __uint_as_float(__float_as_uint(a) & 0xFFFF0000u) + 1.f;
                b = b - __uint_as_float(__float_as_uint(b) & 0xFFFF0000u) + 1.f;
                c = c - __uint_as_float(__float_as_uint(c) & 0xFFFF0000u) + 1.f;
                d = d - __uint_as_float(__float_as_uint(d) & 0xFFFF0000u) + 1.f;
            } else if (MODE == 2) {   // and + packed sub (+ packed add)
                f2 ab = {a, b}, cd = {c, d};
                f2 tab = {__uint_as_float(__float_as_uint(a) & 0xFFFF0000u), __uint_as_float(__float_as_uint(b) & 0xFFFF0000u)};
                f2 tcd = {__uint_as_float(__float_as_uint(c) & 0xFFFF0000u), __uint_as_float(__float_as_uint(d) & 0xFFFF0000u)};
                ab = ab - tab + 1.f;
                cd = cd - tcd + 1.f;
                a = ab.x; b = ab.y; c = cd.x; d = cd.y;
            } else {           // dot2c
                a = res_lo(__float_as_uint(a) >> 16, a) + 1.f;
                b = res_lo(__float_as_uint(b) >> 16, b) + 1.f;
                c = res_lo(__float_as_uint(c) >> 16, c) + 1.f;
                d = res_lo(__float_as_uint(d) >> 16, d) + 1.f;
            }
        }
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + __uint_as_float(p);
}
int main() {
    const int n = 1 << 24;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
        const unsigned e = (u >> 23) & 0xFF;
        if (e == 0xFF) u &= ~(1u << 30);   // no inf / nan
        if (i % 4 == 0) u = (u & 0x807FFFFFu) | ((unsigned)(100 + rand() % 56) << 23);   // ordinary magnitudes
        memcpy(&h[i], &u, 4);
    }
    float *x, *out; unsigned* bad;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&out, 64 * 4)); CK(hipMalloc(&bad, 4));
    CK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(check, dim3(n / 2 / 256), dim3(256), 0, 0, x, n, bad, out);
    unsigned nb; float ho[32];
    CK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho, out, 32 * 4, hipMemcpyDeviceToHost));
    printf("residual via v_dot2c_f32_bf16 vs and+sub over %d values (all exponents): %u pairs differ\n", n, nb);
    for (unsigned k = 0; k < nb && k < 8; ++k) printf("   x=%a  and+sub=%a  dot2=%a (other=%a)\n", ho[4 * k], ho[4 * k + 1], ho[4 * k + 2], ho[4 * k + 3]);
    float* o; CK(hipMalloc(&o, 1024 * 256 * 4));
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, o, 2000);
            else if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(1024), dim3(256), 0, 0, o, 2000);
            else hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, o, 2000);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.3f ms for 2000 x 64 residuals per thread\n", mode == 0 ? "and + sub (+ add)" : (mode == 2 ? "and + packed sub (+ packed add)" : "dot2c (+ shift, add)"), ms);
    }
    return 0;
}
