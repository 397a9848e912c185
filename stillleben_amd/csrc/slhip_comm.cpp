// slhip_comm.cpp -- the one exchange step of the path: all-gather of rendered batches across the
// ranks of a node (one process per GPU) with RCCL over xGMI.  The reference has no counterpart in
// code: it runs one process per GPU (python/src/py_context.cpp:34-52) and leaves the exchange to
// the training framework; SURVEY.md 8b/8e put it behind the C-ABI.
//
// RCCL is bound at run time (dlopen): a process that already carries an RCCL (PyTorch ships one and
// maps it with libtorch_hip.so) must not get a second copy, and a single-GPU user needs none at all.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include <rccl/rccl.h>

#include "slhip_common.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char where[256] = "";
};

RcclApi g_rccl;
std::once_flag g_rccl_once;
char g_rccl_error[512] = "";

void load_rccl()
{
    // 1. an RCCL the process already mapped (PyTorch's), 2. the loader's search path, 3. the ROCm install
    const char* cands[] = {nullptr, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    const char* via = "already mapped librccl.so.1";
    if (const char* e = getenv("SLHIP_RCCL_LIB")) {
        h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
        via = e;
    }
    for (int i = 1; !h && i < 4; ++i) {
        h = dlopen(cands[i], RTLD_NOW | RTLD_GLOBAL);
        via = cands[i];
    }
    if (!h) {
        snprintf(g_rccl_error, sizeof(g_rccl_error), "RCCL not found (librccl.so.1): %s", dlerror());
        return;
    }
    RcclApi a;
    a.handle = h;
#define SLHIP_SYM(field, name)                                                                     \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                                  \
    if (!a.field) {                                                                                 \
        snprintf(g_rccl_error, sizeof(g_rccl_error), "RCCL (%s) lacks %s", via, name);              \
        return;                                                                                     \
    }
    SLHIP_SYM(GetUniqueId, "ncclGetUniqueId")
    SLHIP_SYM(CommInitRank, "ncclCommInitRank")
    SLHIP_SYM(CommDestroy, "ncclCommDestroy")
    SLHIP_SYM(AllGather, "ncclAllGather")
    SLHIP_SYM(GroupStart, "ncclGroupStart")
    SLHIP_SYM(GroupEnd, "ncclGroupEnd")
    SLHIP_SYM(GetErrorString, "ncclGetErrorString")
#undef SLHIP_SYM
    snprintf(a.where, sizeof(a.where), "%s", via);
    g_rccl = a;
}

const RcclApi* rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.handle) {
        slhip::set_error("%s", g_rccl_error);
        return nullptr;
    }
    return &g_rccl;
}

#define SLHIP_NCCL(api, expr)                                                                      \
    do {                                                                                            \
        ncclResult_t _r = (expr);                                                                   \
        if (_r != ncclSuccess) {                                                                    \
            ::slhip::set_error("%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(_r), __FILE__, __LINE__); \
            return -5;                                                                              \
        }                                                                                           \
    } while (0)

}  // namespace

struct slhip_comm {
    ncclComm_t comm;
    int n_ranks, rank, device;
};

static_assert(SLHIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "slhip_comm id size must equal ncclUniqueId");

extern "C" int slhip_comm_unique_id(uint8_t id_out[SLHIP_COMM_ID_BYTES])
{
    const RcclApi* R = rccl();
    if (!R) return -5;
    if (!id_out) {
        slhip::set_error("slhip_comm_unique_id: null output");
        return -1;
    }
    ncclUniqueId id;
    SLHIP_NCCL(R, R->GetUniqueId(&id));
    memcpy(id_out, id.internal, SLHIP_COMM_ID_BYTES);
    return 0;
}

extern "C" int slhip_comm_create(const uint8_t id[SLHIP_COMM_ID_BYTES], int n_ranks, int rank, slhip_comm** comm_out)
{
    const RcclApi* R = rccl();
    if (!R) return -5;
    if (!id || !comm_out || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        slhip::set_error("slhip_comm_create: bad argument (n_ranks %d, rank %d)", n_ranks, rank);
        return -1;
    }
    ncclUniqueId uid;
    memcpy(uid.internal, id, SLHIP_COMM_ID_BYTES);
    slhip_comm* c = new slhip_comm();
    c->n_ranks = n_ranks;
    c->rank = rank;
    hipError_t he = hipGetDevice(&c->device);
    if (he != hipSuccess) {
        delete c;
        slhip::set_error("slhip_comm_create: hipGetDevice failed: %s", hipGetErrorString(he));
        return -2;
    }
    ncclResult_t r = R->CommInitRank(&c->comm, n_ranks, uid, rank);   // collective over the ranks of the node
    if (r != ncclSuccess) {
        delete c;
        slhip::set_error("ncclCommInitRank failed: %s", R->GetErrorString(r));
        return -5;
    }
    *comm_out = c;
    return 0;
}

extern "C" int slhip_comm_destroy(slhip_comm* comm)
{
    if (!comm) return 0;
    const RcclApi* R = rccl();
    if (!R) return -5;
    ncclResult_t r = R->CommDestroy(comm->comm);
    delete comm;
    if (r != ncclSuccess) {
        slhip::set_error("ncclCommDestroy failed: %s", R->GetErrorString(r));
        return -5;
    }
    return 0;
}

extern "C" int slhip_comm_info(const slhip_comm* comm, int* n_ranks, int* rank)
{
    if (!comm) {
        slhip::set_error("slhip_comm_info: null communicator");
        return -1;
    }
    if (n_ranks) *n_ranks = comm->n_ranks;
    if (rank) *rank = comm->rank;
    return 0;
}

extern "C" int slhip_allgather_group(slhip_comm* comm, uint32_t n_buffers, const void* const* d_send, void* const* d_recv,
                                     const uint64_t* bytes, void* stream_)
{
    const RcclApi* R = rccl();
    if (!R) return -5;
    if (!comm || (n_buffers && (!d_send || !d_recv || !bytes))) {
        slhip::set_error("slhip_allgather_group: null argument");
        return -1;
    }
    hipStream_t stream = (hipStream_t)stream_;
    for (uint32_t i = 0; i < n_buffers; ++i)
        if (!d_send[i] || !d_recv[i]) {
            slhip::set_error("slhip_allgather_group: buffer %u is null", i);
            return -1;
        }
    // one fused launch for all buffers of a rendered chunk (rgb, coord, class, instance, normals):
    // an all-gather is type-agnostic, bytes are moved as ncclUint8
    SLHIP_NCCL(R, R->GroupStart());
    for (uint32_t i = 0; i < n_buffers; ++i) {
        ncclResult_t r = R->AllGather(d_send[i], d_recv[i], (size_t)bytes[i], ncclUint8, comm->comm, stream);
        if (r != ncclSuccess) {
            (void)R->GroupEnd();
            slhip::set_error("ncclAllGather (buffer %u, %llu bytes) failed: %s", i, (unsigned long long)bytes[i],
                             R->GetErrorString(r));
            return -5;
        }
    }
    SLHIP_NCCL(R, R->GroupEnd());
    return 0;
}

extern "C" int slhip_allgather(slhip_comm* comm, const void* d_send, void* d_recv, uint64_t bytes, void* stream)
{
    return slhip_allgather_group(comm, 1, &d_send, &d_recv, &bytes, stream);
}
