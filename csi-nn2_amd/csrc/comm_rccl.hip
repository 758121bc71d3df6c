// comm_rccl.hip -- the one collective of the conv path behind the C-ABI: RCCL broadcast of the plans'
// constant blocks (packed weights + zero-point fold + scale / bias tables) from the rank that packed them
// to its peers, over xGMI (BASELINE north_star: "RCCL broadcast of weights over xGMI only").
//
// The data path itself has no collective: the batch is cut into contiguous per-rank slices and every rank
// runs its slice alone.  So RCCL is needed once, at setup, and only by multi-GPU callers: librccl.so is
// opened on first use (dlopen) instead of being a load-time dependency of every single-GPU process.
//
// Bootstrap: rank `root` draws a 128-byte ncclUniqueId (shl_mi355x_comm_unique_id) and ships it to its
// peers through whatever channel the host program already has (a file, MPI, torch.distributed's store ...);
// every rank then calls shl_mi355x_comm_create -- a collective -- with the same bytes.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "common.h"

namespace shl {

struct RcclApi {
    void *lib;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*CommCount)(const ncclComm_t, int *);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *);
    const char *(*GetErrorString)(ncclResult_t);
};

static RcclApi *rccl()
{
    static RcclApi api;
    static int state;  // 0 untried, 1 ready, -1 unavailable
    if (state == 0) {
        state = -1;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!api.lib) {
            set_error("RCCL is not available: %s", dlerror());
            return nullptr;
        }
#define SHL_SYM(field, name)                                                  \
    *reinterpret_cast<void **>(&api.field) = dlsym(api.lib, name);             \
    if (!api.field) {                                                         \
        set_error("librccl.so lacks %s", name);                               \
        return nullptr;                                                       \
    }
        SHL_SYM(GetUniqueId, "ncclGetUniqueId")
        SHL_SYM(CommInitRank, "ncclCommInitRank")
        SHL_SYM(CommDestroy, "ncclCommDestroy")
        SHL_SYM(Broadcast, "ncclBroadcast")
        SHL_SYM(GroupStart, "ncclGroupStart")
        SHL_SYM(GroupEnd, "ncclGroupEnd")
        SHL_SYM(CommCount, "ncclCommCount")
        SHL_SYM(CommUserRank, "ncclCommUserRank")
        SHL_SYM(CommCuDevice, "ncclCommCuDevice")
        SHL_SYM(GetErrorString, "ncclGetErrorString")
#undef SHL_SYM
        state = 1;
    }
    return state == 1 ? &api : nullptr;
}

static int rccl_fail(RcclApi *r, ncclResult_t e, const char *what)
{
    set_error("RCCL error %d (%s) in %s", (int)e, r->GetErrorString(e), what);
    return SHL_MI355X_EHIP;
}

}  // namespace shl

using namespace shl;

extern "C" {

int shl_mi355x_comm_available(void) { return rccl() ? 1 : 0; }

int shl_mi355x_comm_unique_id(void *id128)
{
    RcclApi *r = rccl();
    if (!r) return SHL_MI355X_ENOTSUP;
    if (!id128) return SHL_MI355X_EINVAL;
    ncclUniqueId id;
    ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return rccl_fail(r, e, "ncclGetUniqueId");
    memcpy(id128, &id, sizeof(id));
    return SHL_MI355X_OK;
}

int shl_mi355x_comm_create(const void *id128, int32_t rank, int32_t world, void **comm_out)
{
    RcclApi *r = rccl();
    if (!r) return SHL_MI355X_ENOTSUP;
    if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) {
        set_error("comm_create: bad arguments (rank %d of %d)", rank, world);
        return SHL_MI355X_EINVAL;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        set_error("comm_create: no gfx950 device visible");
        return SHL_MI355X_ENODEV;
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t e = r->CommInitRank(&comm, world, id, rank);
    if (e != ncclSuccess) return rccl_fail(r, e, "ncclCommInitRank");
    *comm_out = comm;
    return SHL_MI355X_OK;
}

int shl_mi355x_comm_destroy(void *comm)
{
    RcclApi *r = rccl();
    if (!r || !comm) return SHL_MI355X_OK;
    ncclResult_t e = r->CommDestroy((ncclComm_t)comm);
    return e == ncclSuccess ? SHL_MI355X_OK : rccl_fail(r, e, "ncclCommDestroy");
}

int shl_mi355x_comm_info(void *comm, int32_t *nranks, int32_t *rank, int32_t *device)
{
    RcclApi *r = rccl();
    if (!r) return SHL_MI355X_ENOTSUP;
    if (!comm) {
        set_error("comm_info: NULL communicator");
        return SHL_MI355X_EINVAL;
    }
    int n = 0, me = 0, dev = 0;
    ncclResult_t e = r->CommCount((ncclComm_t)comm, &n);
    if (e == ncclSuccess) e = r->CommUserRank((ncclComm_t)comm, &me);
    if (e == ncclSuccess) e = r->CommCuDevice((ncclComm_t)comm, &dev);
    if (e != ncclSuccess) return rccl_fail(r, e, "ncclCommCount / UserRank / CuDevice");
    if (nranks) *nranks = n;
    if (rank) *rank = me;
    if (device) *device = dev;
    return SHL_MI355X_OK;
}

int shl_mi355x_comm_bcast(void *comm, void *const *blocks_dev, const size_t *bytes, int32_t count, int32_t root,
                          void *stream)
{
    RcclApi *r = rccl();
    if (!r) return SHL_MI355X_ENOTSUP;
    if (!comm || (count > 0 && (!blocks_dev || !bytes)) || count < 0) {
        set_error("comm_bcast: bad arguments");
        return SHL_MI355X_EINVAL;
    }
    // one group = one fused launch: the blocks travel as a handful of large messages instead of one small
    // collective per layer (xGMI links are point-to-point: message count is what costs at 4-25 MB total)
    ncclResult_t e = r->GroupStart();
    if (e != ncclSuccess) return rccl_fail(r, e, "ncclGroupStart");
    for (int i = 0; i < count; ++i) {
        if (bytes[i] == 0) continue;
        e = r->Broadcast(blocks_dev[i], blocks_dev[i], bytes[i], ncclUint8, root, (ncclComm_t)comm, (hipStream_t)stream);
        if (e != ncclSuccess) {
            (void)r->GroupEnd();
            return rccl_fail(r, e, "ncclBroadcast");
        }
    }
    e = r->GroupEnd();
    if (e != ncclSuccess) return rccl_fail(r, e, "ncclGroupEnd");
    return SHL_MI355X_OK;
}

}  // extern "C"
