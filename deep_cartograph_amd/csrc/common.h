// Shared helpers for libdcv.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dcv.h"

namespace dcv {

void set_error(const char* fmt, ...);

#define DCV_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            dcv::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return DCV_EHIP;                                                             \
        }                                                                                \
    } while (0)

#define DCV_REQUIRE(cond, ...)                                                           \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            dcv::set_error(__VA_ARGS__);                                                 \
            return DCV_EINVAL;                                                           \
        }                                                                                \
    } while (0)

#define DCV_CHECK_LAUNCH() DCV_CHECK_HIP(hipGetLastError())

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kWave = 64;  // gfx950 wavefront

// Kernel-exact timing of ONE launch (dcv_mlp_profile_*): the caller parks a pair of events here, the block engine's launcher
// hands them to hipExtLaunchKernel, which stamps them with the kernel's own begin / end (the interval rocprofv3 reports:
// events recorded around a launch also bracket the command processor's work between launches, 3-4 us at the contract batch),
// and clears the slot.  A slot nobody consumed (a launch that does not go through the block engine) is noticed by the
// caller, which then falls back to recording its events around the launch.
struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};
extern thread_local LaunchEvents g_launch_ev;
extern thread_local hipEvent_t g_launch_taken;   // start event of the pair the launcher consumed last
// Number of CUs of the current device (cached).
int num_cus();
// arithmetic of the matrix products: false = FP32-input MFMA, true = FP32-accurate split products on the BF16
// matrix pipe (default; dcv_set_gemm_mode / environment DCV_GEMM_MODE=native|split)
bool gemm_split();
void set_gemm_split(bool on);

}  // namespace dcv
