// gsage_comm.hip -- the data-parallel step's collectives, issued by the library itself (include/gsage.h,
// "The step's collectives").
//
// The reference has no counterpart: it is one process.  Seed batches shard over the GPUs of a
// node and ONE exchange per step averages the gradients: a flat fp32 bucket through
// ncclAllReduce, plus -- for a trainable embedding table (nn_modules.py:126-155) -- every rank's touched row ids and
// fp32 gradient rows through ncclAllGather (the dense 418 MB table gradient never travels).  Until round 3 the
// all-reduce was a torch.distributed call made from Python between three command lists; here it is a node of the
// step's list (on the list's side stream when it overlaps the next batch's gathers), so a data-parallel step is one
// C call like a single-GPU step.
//
// RCCL is loaded with dlopen: libgsage_hip.so has no link-time dependency on it and a single-GPU process never
// maps it.  xGMI is point-to-point (7 links per GPU): the bucket is 0.9-2.8 MB, latency-bound, hence exactly one
// collective per step for it and never per-parameter reductions.
#include "gsage_common.h"

#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

namespace gsage {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;

struct Comm {
    ncclComm_t c = nullptr;
    int rank = 0, world = 1;
};

template <typename F> static bool sym(void *h, const char *name, F &out)
{
    out = reinterpret_cast<F>(dlsym(h, name));
    return out != nullptr;
}

static int nccl_fail(const char *what, ncclResult_t r)
{
    set_error("%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return GSAGE_ELAUNCH;
}

// a collective as a list node: failures at replay surface through gsage_cmdlist_replay's return code
template <typename Fn> static int issue_or_record(const char *what, hipStream_t stream, Fn fn)
{
    if (t_recording) {
        t_recording->target().emplace_back([what, fn](hipStream_t s) {
            const ncclResult_t r = fn(s);
            if (r != ncclSuccess) {
                nccl_fail(what, r);
                t_node_error = 1;
            }
        });
        t_recording->n_marks += 1;
        return GSAGE_OK;
    }
    const ncclResult_t r = fn(stream);
    return r == ncclSuccess ? GSAGE_OK : nccl_fail(what, r);
}

}  // namespace gsage

using namespace gsage;

extern "C" {

int gsage_comm_load(const char *path)
{
    if (g_rccl.handle) return GSAGE_OK;
    const char *names[] = {path, "librccl.so", "librccl.so.1", nullptr};
    void *h = nullptr;
    for (int i = path ? 0 : 1; names[i] && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        set_error("comm_load: cannot load librccl (%s)", dlerror());
        return GSAGE_ENODEV;
    }
    Rccl r;
    r.handle = h;
    const bool ok = sym(h, "ncclGetUniqueId", r.GetUniqueId) && sym(h, "ncclCommInitRank", r.CommInitRank) &&
                    sym(h, "ncclCommDestroy", r.CommDestroy) && sym(h, "ncclAllReduce", r.AllReduce) &&
                    sym(h, "ncclAllGather", r.AllGather) && sym(h, "ncclGroupStart", r.GroupStart) &&
                    sym(h, "ncclGroupEnd", r.GroupEnd) && sym(h, "ncclGetErrorString", r.GetErrorString);
    if (!ok) {
        set_error("comm_load: librccl lacks an expected entry point (%s)", dlerror());
        dlclose(h);
        return GSAGE_ENODEV;
    }
    g_rccl = r;
    return GSAGE_OK;
}

int gsage_comm_unique_id(void *id128)
{
    GSAGE_REQUIRE(g_rccl.handle, "comm_unique_id: call gsage_comm_load first");
    GSAGE_REQUIRE(id128, "comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail("comm_unique_id", r);
    memcpy(id128, &id, sizeof(id));
    return GSAGE_OK;
}

int gsage_comm_create(const void *id128, int32_t rank, int32_t world, void **comm)
{
    GSAGE_REQUIRE(g_rccl.handle, "comm_create: call gsage_comm_load first");
    GSAGE_REQUIRE(id128 && comm && world >= 1 && rank >= 0 && rank < world, "comm_create: bad arguments");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm *c = new Comm();
    c->rank = rank;
    c->world = world;
    const ncclResult_t r = g_rccl.CommInitRank(&c->c, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return nccl_fail("comm_create (ncclCommInitRank)", r);
    }
    *comm = c;
    return GSAGE_OK;
}

int gsage_comm_destroy(void *comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return GSAGE_OK;
    if (c->c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->c);
    delete c;
    return GSAGE_OK;
}

int gsage_comm_all_reduce_f32(void *comm, float *buf, int64_t n, int32_t average, void *stream)
{
    GSAGE_REQUIRE(comm && buf && n > 0, "comm_all_reduce_f32: bad arguments");
    const ncclComm_t c = ((Comm *)comm)->c;
    const ncclRedOp_t op = average ? ncclAvg : ncclSum;
    const size_t count = (size_t)n;
    return issue_or_record("comm_all_reduce_f32 (ncclAllReduce)", (hipStream_t)stream, [=](hipStream_t s) {
        return g_rccl.AllReduce(buf, buf, count, ncclFloat32, op, c, s);
    });
}

int gsage_comm_all_gather(void *comm, const void *send, void *recv, int64_t bytes, void *stream)
{
    GSAGE_REQUIRE(comm && send && recv && bytes > 0 && bytes % 4 == 0, "comm_all_gather: bad arguments (bytes %% 4 == 0)");
    const ncclComm_t c = ((Comm *)comm)->c;
    const size_t count = (size_t)(bytes / 4);
    return issue_or_record("comm_all_gather (ncclAllGather)", (hipStream_t)stream, [=](hipStream_t s) {
        return g_rccl.AllGather(send, recv, count, ncclUint32, c, s);
    });
}

int gsage_comm_group(void *comm, int32_t begin, void *stream)
{
    GSAGE_REQUIRE(comm && g_rccl.handle, "comm_group: no communicator");
    return issue_or_record(begin ? "comm_group (ncclGroupStart)" : "comm_group (ncclGroupEnd)", (hipStream_t)stream,
                           [=](hipStream_t) { return begin ? g_rccl.GroupStart() : g_rccl.GroupEnd(); });
}

}  // extern "C"
