#include "common.h"
#include <string.h>
#include <stdlib.h>

namespace dcv {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int num_cus() {
    static int cached = 0;
    if (cached) return cached;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cached = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return cached;
}

thread_local LaunchEvents g_launch_ev;
thread_local hipEvent_t g_launch_taken = nullptr;
static int g_gemm_split = -1;   // -1: not decided yet (environment, else the default)
bool gemm_split() {
    if (g_gemm_split < 0) {
        const char* e = getenv("DCV_GEMM_MODE");
        g_gemm_split = (e && strcmp(e, "native") == 0) ? 0 : 1;
    }
    return g_gemm_split == 1;
}
void set_gemm_split(bool on) { g_gemm_split = on ? 1 : 0; }
}  // namespace dcv

extern "C" int dcv_set_gemm_mode(int mode) {
    DCV_REQUIRE(mode == DCV_GEMM_NATIVE_F32 || mode == DCV_GEMM_SPLIT_BF16X6, "dcv_set_gemm_mode: unknown mode %d", mode);
    dcv::set_gemm_split(mode == DCV_GEMM_SPLIT_BF16X6);
    return DCV_OK;
}
extern "C" int dcv_get_gemm_mode(void) { return dcv::gemm_split() ? DCV_GEMM_SPLIT_BF16X6 : DCV_GEMM_NATIVE_F32; }

extern "C" int dcv_abi_version(void) { return DCV_ABI_VERSION; }
extern "C" const char* dcv_last_error(void) { return dcv::g_err; }

extern "C" int dcv_device_info(int device, int* n_cu, int64_t* hbm_bytes, char* name, size_t name_len) {
    hipDeviceProp_t prop;
    DCV_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (name && name_len) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    return DCV_OK;
}
