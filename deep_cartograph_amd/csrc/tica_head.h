// The d x d algebra of the Deep-TICA loss head spread over the lanes of one wave, shared by the statistics launch of the
// layer-by-layer step (mlp.hip) and the fused small-network forward (snet_dt.hip).
#pragma once
#include "common.h"
#include <math.h>

namespace dcv {

// tica_grad_body spread over the lanes of one wave (D <= 4: lane l < D*D owns matrix element (l / D, l % D)): the
// single-thread form is a chain of ~2000 dependent float64 instructions (8 us); here every matrix product is one step
// of D multiply-adds per lane and only the Cholesky factorisation (D columns) and the two triangular solves (one
// column of the inverse per lane) stay sequential.  Matrices live in LDS; the wave is its own barrier.
template <int D>
struct TicaWaveLds {
    double mu[D], ml[D], invL[D];
    double C0[D * D], Ct[D * D], L[D * D], A[D * D], K[D * D], T[D * D];
};
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int D>
__device__ __forceinline__ void tica_grad_wave(TicaWaveLds<D>& w, const double* __restrict__ stats, double Bg, double reg,
                                               double* __restrict__ gradp, double* __restrict__ log, int* __restrict__ log_count,
                                               int log_cap, int log_width, int lane) {
    const int i = lane / D, j = lane % D;
    const bool el = lane < D * D;
    const double invB = 1.0 / Bg;
    const double* Stt = stats + 2 * D;
    const double* Stl = stats + 2 * D + D * D;
    if (lane < D) {
        w.mu[lane] = stats[lane] * invB;
        w.ml[lane] = stats[D + lane] * invB;
    }
    wave_sync_lds();
    if (el) {
        w.C0[lane] = 0.5 * (Stt[i * D + j] + Stt[j * D + i]) * invB - w.mu[i] * w.mu[j];
        const double cij = Stl[i * D + j] * invB - w.mu[i] * w.ml[j];
        const double cji = Stl[j * D + i] * invB - w.mu[j] * w.ml[i];
        w.Ct[lane] = 0.5 * (cij + cji);
        w.L[lane] = 0.0;
    }
    wave_sync_lds();
    // Cholesky of C0 + reg I, column by column
    bool ok = true;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        if (lane == 0) {
            double s = w.C0[k * D + k] + reg;
#pragma unroll
            for (int m = 0; m < D; ++m)
                if (m < k) s -= w.L[k * D + m] * w.L[k * D + m];
            const double lkk = sqrt(s);
            w.L[k * D + k] = lkk;
            w.invL[k] = s > 0.0 ? 1.0 / lkk : NAN;
        }
        wave_sync_lds();
        if (lane < D && lane > k) {   // L[lane][k]
            double s = w.C0[lane * D + k];
#pragma unroll
            for (int m = 0; m < D; ++m)
                if (m < k) s -= w.L[lane * D + m] * w.L[k * D + m];
            w.L[lane * D + k] = s * w.invL[k];
        }
        wave_sync_lds();
    }
#pragma unroll
    for (int k = 0; k < D; ++k) ok = ok && !(w.invL[k] != w.invL[k]);
    // A = (L L^T)^-1: lane c < D solves L y = e_c, L^T a = y (column c of A)
    if (lane < D) {
        const int c = lane;
        double y[D], a[D];
#pragma unroll
        for (int r = 0; r < D; ++r) {
            double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < D; ++m)
                if (m < r) s -= w.L[r * D + m] * y[m];
            y[r] = s * w.invL[r];
        }
#pragma unroll
        for (int rr = 0; rr < D; ++rr) {
            const int r = D - 1 - rr;
            double s = y[r];
#pragma unroll
            for (int m = 0; m < D; ++m)
                if (m > r) s -= w.L[m * D + r] * a[m];
            a[r] = s * w.invL[r];
        }
#pragma unroll
        for (int r = 0; r < D; ++r) w.A[r * D + c] = a[r];
    }
    wave_sync_lds();
    if (el) {   // K = A Ct
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < D; ++m) s += w.A[i * D + m] * w.Ct[m * D + j];
        w.K[lane] = s;
    }
    wave_sync_lds();
    double Tij = 0.0;
    if (el) {   // T = K A (= A Ct A)
#pragma unroll
        for (int m = 0; m < D; ++m) Tij += w.K[i * D + m] * w.A[m * D + j];
        w.T[lane] = Tij;
    }
    wave_sync_lds();
    if (gradp && el) {
        double g0 = 0.0;
#pragma unroll
        for (int m = 0; m < D; ++m) g0 += w.K[i * D + m] * w.T[m * D + j];
        const double Gt = -(w.T[i * D + j] + w.T[j * D + i]);
        gradp[D + lane] = 4.0 * g0 * invB;               // (2/B) G0, G0 = 2 K T
        gradp[D + D * D + lane] = Gt * invB;             // (1/B) Gtau
    }
    if (gradp && lane < D) {
        gradp[lane] = w.mu[lane];
        double cs = 0.0;
#pragma unroll
        for (int m = 0; m < D; ++m) cs += -(w.T[lane * D + m] + w.T[m * D + lane]) * (w.ml[m] - w.mu[m]);
        gradp[D + 2 * D * D + lane] = -cs * invB;
    }
    if (log_count == nullptr) return;   // no record asked for (the workgroups of a fused backward other than the first)
    const int slot = *log_count;
    if (slot < log_cap) {
        double* rec = log + (int64_t)slot * log_width;
        if (lane == 0) {
            double loss = 0.0;   // -tr(K K), summed in the order of the single-thread form
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int b = 0; b < D; ++b) loss -= w.K[a * D + b] * w.K[b * D + a];
            rec[0] = ok ? loss : NAN;
            rec[1] = Bg;
        }
        if (el) {
            rec[2 + lane] = w.C0[lane];
            rec[2 + D * D + lane] = w.Ct[lane];
        }
        if (lane < D) rec[2 + 2 * D * D + lane] = w.mu[lane];
    }
    wave_sync_lds();
    if (lane == 0) *log_count = slot + 1;
}

// what the last arriver of a ticketed statistics reduction goes on to do (one-GPU steps: the loss head in the same launch)
struct FusedHead {
    int on;            // run tica_grad_body in the last block
    double Bg, reg;
    double* gradp;     // null: evaluation only
    double* log;
    int* log_count;
    int log_cap, log_width;
};

}  // namespace dcv
