// RCCL communicator behind the C-ABI (dcv_comm_*): the collectives of a frame-sharded fit -- batch statistics and
// gradient buffers of the MLP engine, covariance blocks, k-means sums -- issued from the library itself, in stream
// order with its kernels, with no host language in the loop.  One process per GPU; the 128-byte unique id is
// created on rank 0 (dcv_comm_unique_id) and handed to the other ranks by whatever bootstrap the caller has (the
// Python host broadcasts it with torch.distributed / a file / MPI).  librccl.so is opened at run time: libdcv.so
// keeps loading on hosts without RCCL, where dcv_comm_create reports the reason.
//
// EXPERIMENTAL (include/dcv.h says so too).  State of the evidence: built and exercised with world = 1 on the one-GPU box
// (tests/test_mlp_gpu.py); no multi-GPU node has been available to the builder, so a multi-rank communicator has not
// executed -- bench.py therefore keeps torch.distributed (the same RCCL underneath) as the default transport and takes
// this path with --native-rccl.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>
#include <new>

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};

RcclApi* rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return &api;
    tried = true;
    // A process that already carries an RCCL -- torch bundles one and loads it with `import torch` -- must not get a second
    // copy beside it (two RCCLs in one process each keep their own bootstrap threads and device state): first ask for the
    // copy that is ALREADY mapped (RTLD_NOLOAD), only then load one.
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (api.lib) break;
    }
    if (!api.lib) {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
    }
    if (!api.lib) {
        snprintf(api.why, sizeof(api.why), "librccl.so not found (%s)", dlerror());
        return &api;
    }
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
        snprintf(api.why, sizeof(api.why), "librccl.so lacks an expected symbol");
        api.lib = nullptr;
    }
    return &api;
}

}  // namespace

struct dcv_comm {
    ncclComm_t comm;
    int world, rank;
    hipStream_t main;        // the launch stream the collectives are ordered with (dcv_comm_bind_stream)
    hipStream_t side;        // stream of reductions started with DCV_DP_UPPER_START
    hipEvent_t fork, join;
    bool pending;
};

#define DCV_CHECK_RCCL(expr)                                                                         \
    do {                                                                                             \
        ncclResult_t _r = (expr);                                                                    \
        if (_r != ncclSuccess) {                                                                     \
            dcv::set_error("%s failed: %s", #expr, rccl()->GetErrorString(_r));                      \
            return DCV_EHIP;                                                                         \
        }                                                                                            \
    } while (0)

extern "C" int dcv_comm_unique_id(void* id_out_128) {
    DCV_REQUIRE(id_out_128, "dcv_comm_unique_id: null");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
    RcclApi* a = rccl();
    DCV_REQUIRE(a->lib, "dcv_comm_unique_id: %s", a->why);
    DCV_CHECK_RCCL(a->GetUniqueId(static_cast<ncclUniqueId*>(id_out_128)));
    return DCV_OK;
}

extern "C" int dcv_comm_create(int32_t world, int32_t rank, const void* id_128, dcv_comm** out) {
    DCV_REQUIRE(out && id_128 && world >= 1 && rank >= 0 && rank < world, "dcv_comm_create: bad arguments");
    *out = nullptr;
    RcclApi* a = rccl();
    DCV_REQUIRE(a->lib, "dcv_comm_create: %s", a->why);
    dcv_comm* c = new (std::nothrow) dcv_comm();
    DCV_REQUIRE(c, "dcv_comm_create: out of host memory");
    c->world = world;
    c->rank = rank;
    c->main = nullptr;
    c->pending = false;
    ncclUniqueId id;
    memcpy(&id, id_128, sizeof(id));
    ncclResult_t r = a->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        dcv::set_error("ncclCommInitRank(world=%d, rank=%d) failed: %s", world, rank, a->GetErrorString(r));
        delete c;
        return DCV_EHIP;
    }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->join, hipEventDisableTiming) != hipSuccess) {
        dcv::set_error("dcv_comm_create: stream / event creation failed");
        (void)a->CommDestroy(c->comm);
        delete c;
        return DCV_EHIP;
    }
    *out = c;
    return DCV_OK;
}

extern "C" void dcv_comm_destroy(dcv_comm* c) {
    if (!c) return;
    (void)hipStreamSynchronize(c->side);
    (void)rccl()->CommDestroy(c->comm);
    (void)hipEventDestroy(c->fork);
    (void)hipEventDestroy(c->join);
    (void)hipStreamDestroy(c->side);
    delete c;
}

extern "C" int dcv_comm_bind_stream(dcv_comm* c, void* stream) {
    DCV_REQUIRE(c, "dcv_comm_bind_stream: null");
    c->main = dcv::as_stream(stream);
    return DCV_OK;
}

extern "C" int dcv_comm_allreduce(dcv_comm* c, void* buf_d, int64_t count, int32_t dtype, int32_t op, void* stream) {
    DCV_REQUIRE(c && buf_d && count >= 0 && (dtype == DCV_DTYPE_F32 || dtype == DCV_DTYPE_F64) && op >= 0 && op <= 2, "dcv_comm_allreduce: bad arguments");
    const ncclRedOp_t ops[3] = {ncclSum, ncclMin, ncclMax};
    DCV_CHECK_RCCL(rccl()->AllReduce(buf_d, buf_d, (size_t)count, dtype == DCV_DTYPE_F64 ? ncclFloat64 : ncclFloat32, ops[op], c->comm,
                                     dcv::as_stream(stream)));
    return DCV_OK;
}

// The all-reduce callback of dcv_mlp_dp_step over a communicator: user = the dcv_comm (bound to the launch stream).
static int comm_dp_allreduce(void* user, void* buf_d, int64_t count, int32_t dtype, int32_t phase) {
    dcv_comm* c = static_cast<dcv_comm*>(user);
    if (phase == DCV_DP_WAIT) {
        if (c->pending) {   // the launch stream waits for the reductions started on the side stream
            DCV_CHECK_HIP(hipEventRecord(c->join, c->side));
            DCV_CHECK_HIP(hipStreamWaitEvent(c->main, c->join, 0));
            c->pending = false;
        }
        return DCV_OK;
    }
    if (phase == DCV_DP_UPPER_START) {   // behind what the launch stream has enqueued so far, beside what it enqueues next
        DCV_CHECK_HIP(hipEventRecord(c->fork, c->main));
        DCV_CHECK_HIP(hipStreamWaitEvent(c->side, c->fork, 0));
        c->pending = true;
        return dcv_comm_allreduce(c, buf_d, count, dtype, 0, c->side);
    }
    return dcv_comm_allreduce(c, buf_d, count, dtype, 0, c->main);
}

extern "C" dcv_allreduce_fn dcv_comm_dp_allreduce_fn(void) { return comm_dp_allreduce; }
extern "C" int32_t dcv_comm_world(const dcv_comm* c) { return c ? c->world : 0; }
extern "C" int32_t dcv_comm_rank(const dcv_comm* c) { return c ? c->rank : -1; }
