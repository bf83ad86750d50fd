// Gradient all-reduce of the data-parallel trainer behind the C ABI (`rvlm_allreduce_grads(comm, buf,
// count, dtype)`), one process per GPU, RCCL over xGMI.  Replaces what the reference gets from single-process
// torch.nn.DataParallel (train/adversarial_training_clip.py:184-191: per-forward parameter broadcast + gather + reduce_add in
// backward) with ONE in-place sum per gradient bucket on the caller's stream.
//
// The Python host of this package reduces through torch.distributed (the same RCCL underneath: it already owns the process
// group, the rendezvous and the comm stream); these entry points are for a host WITHOUT torch - with them the C ABI alone runs
// the data-parallel step (rvlm_vit_backward_params_stages -> rvlm_allreduce_grads per bucket -> rvlm_adamw_step with
// 1 / world folded in).
//
// librccl is resolved with dlopen at the first rvlm_comm_* call, not at load time: librvlm.so keeps no hard dependency on it
// (the attack path needs no collective at all) - neither at run time nor at BUILD time: the handful of
// RCCL types and enumerators the five entry points use are declared here (NCCL's stable public ABI: ncclUniqueId = 128
// opaque bytes, ncclResult_t 0 = success, ncclSum = 0, ncclFloat32 = 7, ncclBfloat16 = 9), so the library builds on a box
// without the RCCL headers.
#include "kernels.h"

#include <dlfcn.h>
#include <string.h>
#include <mutex>

extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclFloat32 = 7, ncclBfloat16 = 9 } ncclDataType_t;
}

struct rvlm_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

namespace rvlm {
namespace {
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl& rccl() {
    std::call_once(g_rccl_once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g_rccl.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.so) break;
        }
        if (!g_rccl.so) { g_rccl.error = std::string("librccl not found: ") + dlerror(); return; }
        auto sym = [&](const char* n) {
            void* p = dlsym(g_rccl.so, n);
            if (!p && g_rccl.error.empty()) g_rccl.error = std::string("librccl lacks ") + n;
            return p;
        };
        g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
        g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
        g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    });
    return g_rccl;
}
int rccl_fail(const char* what, ncclResult_t r) {
    const Rccl& R = rccl();
    return fail(RVLM_ERR_HIP, std::string(what) + ": " + (R.GetErrorString ? R.GetErrorString(r) : "RCCL error"));
}
}  // namespace
}  // namespace rvlm

using namespace rvlm;

static_assert(RVLM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the rendezvous token is RCCL's ncclUniqueId");

extern "C" int rvlm_comm_unique_id(uint8_t* out_id) {
    RVLM_REQUIRE(out_id, "rvlm_comm_unique_id: null output");
    const Rccl& R = rccl();
    if (!R.error.empty()) return fail(RVLM_ERR_UNSUPPORTED, "rvlm_comm_unique_id: " + R.error);
    ncclUniqueId id;
    const ncclResult_t r = R.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    memcpy(out_id, id.internal, NCCL_UNIQUE_ID_BYTES);
    return RVLM_OK;
}

extern "C" int rvlm_comm_create(const uint8_t* id, int rank, int world, rvlm_comm** out) {
    RVLM_REQUIRE(id && out, "rvlm_comm_create: null argument");
    RVLM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rvlm_comm_create: need 0 <= rank < world");
    const Rccl& R = rccl();
    if (!R.error.empty()) return fail(RVLM_ERR_UNSUPPORTED, "rvlm_comm_create: " + R.error);
    ncclUniqueId nid;
    memcpy(nid.internal, id, NCCL_UNIQUE_ID_BYTES);
    rvlm_comm* c = new rvlm_comm();
    c->rank = rank; c->world = world;
    (void)hipGetDevice(&c->device);                 // the communicator lives on the calling thread's current device
    const ncclResult_t r = R.CommInitRank(&c->comm, world, nid, rank);
    if (r != ncclSuccess) { delete c; return rccl_fail("ncclCommInitRank", r); }
    *out = c;
    return RVLM_OK;
}

extern "C" int rvlm_comm_destroy(rvlm_comm* c) {
    if (!c) return RVLM_OK;
    const Rccl& R = rccl();
    ncclResult_t r = ncclSuccess;
    if (c->comm && R.CommDestroy) r = R.CommDestroy(c->comm);
    delete c;
    return r == ncclSuccess ? RVLM_OK : rccl_fail("ncclCommDestroy", r);
}

extern "C" int rvlm_comm_info(const rvlm_comm* c, int* rank, int* world) {
    RVLM_REQUIRE(c, "rvlm_comm_info: null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return RVLM_OK;
}

// In-place SUM over the ranks of buf[0 .. count) on `stream` (asynchronous: ordered behind the kernels that produced the
// bucket, ahead of whatever the caller enqueues next on that stream; to overlap the reduction with the remaining backward
// stages, give it its own stream and order it with events).  dtype: RVLM_DTYPE_F32 (the trainer's flat gradient buffer)
// or RVLM_DTYPE_BF16.  The 1 / world factor is NOT applied here: rvlm_adamw_step takes it (grad_scale).
extern "C" int rvlm_allreduce_grads(rvlm_comm* c, void* buf, size_t count, int dtype, rvlm_stream_t stream) {
    RVLM_REQUIRE(c && c->comm, "rvlm_allreduce_grads: null communicator");
    RVLM_REQUIRE(buf || count == 0, "rvlm_allreduce_grads: null buffer");
    RVLM_REQUIRE(dtype == RVLM_DTYPE_F32 || dtype == RVLM_DTYPE_BF16, "rvlm_allreduce_grads: dtype must be RVLM_DTYPE_F32 / _BF16");
    if (count == 0) return RVLM_OK;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != c->device)       // RCCL would enqueue on the communicator's device whatever `stream` belongs to
        return fail(RVLM_ERR_STATE, "rvlm_allreduce_grads: the calling thread's current device (" + std::to_string(cur) +
                                    ") is not the device the communicator was created on (" + std::to_string(c->device) + ")");
    const Rccl& R = rccl();
    const ncclResult_t r = R.AllReduce(buf, buf, count, dtype == RVLM_DTYPE_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, c->comm,
                                       (hipStream_t)stream);
    return r == ncclSuccess ? RVLM_OK : rccl_fail("ncclAllReduce", r);
}
