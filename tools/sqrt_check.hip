// Is the device's float64 square root correctly rounded?  (numpy's is: np.argmin over np.linalg.norm decides ties on the
// ROUNDED roots, so the centroid search must produce the same doubles.)  Compares sqrt(), __dsqrt_rn() and an fma-corrected
// root against the host's sqrt on random inputs; prints the mismatch counts.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ double sqrt_cr(double x) {
    double s = sqrt(x);
    if (x > 0.0 && x < INFINITY) {
        const double r = fma(-s, s, x);                                   // x - s^2 (exact up to the rounding of a tiny number)
        const double up = __longlong_as_double(__double_as_longlong(s) + 1);
        const double dn = __longlong_as_double(__double_as_longlong(s) - 1);
        // s is the correctly rounded root iff (s + dn) / 2 < sqrt(x) < (s + up) / 2, i.e. |x - s^2| below s * ulp(s)
        if (r > s * (up - s)) s = up;
        else if (-r > s * (s - dn)) s = dn;
    }
    return s;
}
__global__ void k(const double* x, double* a, double* b, double* c, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    a[i] = sqrt(x[i]);
    b[i] = __dsqrt_rn(x[i]);
    c[i] = sqrt_cr(x[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n);
    srand48(7);
    for (int i = 0; i < n; ++i) x[i] = (i & 1) ? drand48() * 4.0 : ldexp(drand48(), (int)(lrand48() % 60) - 30);
    double *dx, *da, *db, *dc;
    hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dc, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, dc, n);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
    long ma = 0, mb = 0, mc = 0;
    for (int i = 0; i < n; ++i) {
        const double r = sqrt(x[i]);
        ma += a[i] != r; mb += b[i] != r; mc += c[i] != r;
    }
    printf("n=%d mismatches vs host sqrt: sqrt() %ld, __dsqrt_rn() %ld, fma-corrected %ld\n", n, ma, mb, mc);
    return 0;
}
