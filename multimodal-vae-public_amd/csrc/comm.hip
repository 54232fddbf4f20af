// comm.hip -- the data-parallel gradient exchange behind the C ABI: RCCL over xGMI.
//
// The reference is single-device (no torch.distributed, no NCCL call site anywhere: SURVEY 2.1); this is the new
// component north_star asks for, as a host that is NOT PyTorch would drive it: one process per GPU, one
// communicator per process, in-place sum all-reduces of contiguous fp32 ranges of the gradient arena (the 1/N is
// folded into mvae_adam_*'s grad_scale), a broadcast to make the replicas start equal.
//
// Streams: the communicator owns ONE communication stream.  mvae_comm_allreduce_async(.., stream) orders the
// collective after everything already enqueued on the caller's `stream` (event edge), runs it on the
// communication stream -- so it overlaps with whatever the caller enqueues next (the rest of the backward) -- and
// hands back a ticket; mvae_comm_wait(ticket, stream) makes `stream` wait for that collective.  Nothing blocks
// the host.  Both are plain stream/event operations + one RCCL enqueue, so a whole train step -- forward,
// backward, bucket all-reduces, per-bucket Adam -- can be captured into ONE hipGraph (RCCL supports stream
// capture): the communication stream forks from and joins back into the capturing stream.
//
// RCCL is bound at run time (dlopen): libmvae_hip.so carries no link-time dependency on it, a single-GPU user
// never loads it, and a PyTorch host can point MVAE_RCCL_LIB (or mvae_comm_use_library) at the librccl.so its
// torch already loaded so the process holds one copy.  No global mutable state other than that binding.
#include <dlfcn.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>

#include <rccl/rccl.h>      // types and enums only; every call goes through the table below

#include "common.h"

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;     // optional (mvae_comm_async_error)
    // optional (MVAE_COMM_ALGO=rs_ag): the all-reduce as a direct reduce-scatter + all-gather
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    char path[512] = {0};
};

RcclApi g_rccl;
std::mutex g_rccl_mutex;
char g_rccl_override[512] = {0};

bool bind_all(RcclApi &a, void *h) {
#define MVAE_BIND(field, name)                                           \
    *(void **)(&a.field) = dlsym(h, name);                               \
    if (!a.field) return false;
    MVAE_BIND(GetUniqueId, "ncclGetUniqueId")
    MVAE_BIND(CommInitRank, "ncclCommInitRank")
    MVAE_BIND(CommDestroy, "ncclCommDestroy")
    MVAE_BIND(AllReduce, "ncclAllReduce")
    MVAE_BIND(Broadcast, "ncclBroadcast")
    MVAE_BIND(GetVersion, "ncclGetVersion")
    MVAE_BIND(GetErrorString, "ncclGetErrorString")
#undef MVAE_BIND
    *(void **)(&a.CommGetAsyncError) = dlsym(h, "ncclCommGetAsyncError");      // absent: async errors are not observable
    *(void **)(&a.ReduceScatter) = dlsym(h, "ncclReduceScatter");               // absent: MVAE_COMM_ALGO=rs_ag falls back to ncclAllReduce
    *(void **)(&a.AllGather) = dlsym(h, "ncclAllGather");
    return true;
}

// first use: explicit path (mvae_comm_use_library) > $MVAE_RCCL_LIB > the loader's search path
const RcclApi *rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    const char *cands[4] = {nullptr, nullptr, "librccl.so.1", "librccl.so"};
    if (g_rccl_override[0]) cands[0] = g_rccl_override;
    const char *env = getenv("MVAE_RCCL_LIB");
    if (env && env[0]) cands[1] = env;
    for (const char *c : cands) {
        if (!c) continue;
        void *h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        RcclApi a;
        if (bind_all(a, h)) {
            a.handle = h;
            snprintf(a.path, sizeof(a.path), "%s", c);
            g_rccl = a;
            return &g_rccl;
        }
        dlclose(h);
    }
    return nullptr;
}

constexpr int COMM_SLOTS = 16;      // tickets in flight between two waits (a step uses 2-3 buckets)

}  // namespace

struct mvae_comm {
    ncclComm_t nccl;
    int rank, world, device;
    hipStream_t stream;                        // the communication stream
    hipEvent_t ready[COMM_SLOTS], done[COMM_SLOTS];
    long issued;                               // tickets handed out so far
    int rs_ag;                                 // MVAE_COMM_ALGO=rs_ag at mvae_comm_init (see mvae_comm_allreduce_async)
    char last_error[256];
};

namespace {

int comm_fail(mvae_comm *c, const char *what, const char *detail) {
    if (c) snprintf(c->last_error, sizeof(c->last_error), "%s: %s", what, detail ? detail : "?");
    return MVAE_ERR_COMM;
}

int nccl_check(mvae_comm *c, const RcclApi *api, ncclResult_t r, const char *what) {
    if (r == ncclSuccess) return MVAE_OK;
    return comm_fail(c, what, api->GetErrorString ? api->GetErrorString(r) : "rccl error");
}

int hip_check(mvae_comm *c, hipError_t e, const char *what) {
    if (e == hipSuccess) return MVAE_OK;
    return comm_fail(c, what, hipGetErrorString(e));
}

}  // namespace

MVAE_EXPORT int mvae_comm_use_library(const char *path) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return (path && strcmp(path, g_rccl.path) == 0) ? MVAE_OK : MVAE_ERR_ARG;   // already bound
    if (!path || strlen(path) >= sizeof(g_rccl_override)) return MVAE_ERR_ARG;
    snprintf(g_rccl_override, sizeof(g_rccl_override), "%s", path);
    return MVAE_OK;
}

MVAE_EXPORT int mvae_comm_rccl_version(void) {
    const RcclApi *api = rccl();
    int v = 0;
    if (!api || api->GetVersion(&v) != ncclSuccess) return MVAE_ERR_COMM;
    return v;
}

MVAE_EXPORT int mvae_comm_unique_id(void *id_out, size_t id_bytes) {
    if (!id_out || id_bytes < MVAE_COMM_ID_BYTES) return MVAE_ERR_ARG;
    static_assert(MVAE_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size follows RCCL's");
    const RcclApi *api = rccl();
    if (!api) return MVAE_ERR_COMM;
    ncclUniqueId id;
    if (api->GetUniqueId(&id) != ncclSuccess) return MVAE_ERR_COMM;
    memcpy(id_out, &id, sizeof(id));
    return MVAE_OK;
}

MVAE_EXPORT int mvae_comm_init(mvae_comm_t **comm_out, const void *id, size_t id_bytes, int rank, int world, int device) {
    if (!comm_out || !id || id_bytes < MVAE_COMM_ID_BYTES || world < 1 || rank < 0 || rank >= world || device < 0)
        return MVAE_ERR_ARG;
    *comm_out = nullptr;
    const RcclApi *api = rccl();
    if (!api) return MVAE_ERR_COMM;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) return MVAE_ERR_ARG;
    mvae_comm *c = (mvae_comm *)calloc(1, sizeof(mvae_comm));
    if (!c) return MVAE_ERR_COMM;
    c->rank = rank; c->world = world; c->device = device; c->issued = 0;
    {
        const char *algo = getenv("MVAE_COMM_ALGO");
        c->rs_ag = (algo && strcmp(algo, "rs_ag") == 0 && api->ReduceScatter && api->AllGather) ? 1 : 0;
    }
    int prev_device = -1;
    (void)hipGetDevice(&prev_device);
    int n_events = 0;               // events created so far (ready / done alternate): what a failure has to give back
    bool have_stream = false;
    int rc = hip_check(c, hipSetDevice(device), "hipSetDevice");
    if (rc == MVAE_OK) {
        rc = hip_check(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate");
        have_stream = rc == MVAE_OK;
    }
    for (int i = 0; rc == MVAE_OK && i < COMM_SLOTS; ++i) {
        rc = hip_check(c, hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming), "hipEventCreate");
        if (rc == MVAE_OK) {
            ++n_events;
            rc = hip_check(c, hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming), "hipEventCreate");
            if (rc == MVAE_OK) ++n_events;
        }
    }
    if (rc == MVAE_OK) {
        ncclUniqueId uid;
        memcpy(&uid, id, sizeof(uid));
        rc = nccl_check(c, api, api->CommInitRank(&c->nccl, world, uid, rank), "ncclCommInitRank");
    }
    if (rc != MVAE_OK) {
        fprintf(stderr, "[mvae_comm_init] rank %d/%d on device %d failed: %s\n", rank, world, device, c->last_error);
        // give back what was created, and the caller's device
        for (int i = 0; i < n_events; ++i) (void)hipEventDestroy((i & 1) ? c->done[i >> 1] : c->ready[i >> 1]);
        if (have_stream) (void)hipStreamDestroy(c->stream);
        if (prev_device >= 0 && prev_device != device) (void)hipSetDevice(prev_device);
        free(c);
        return rc;
    }
    *comm_out = c;
    return MVAE_OK;
}

MVAE_EXPORT int mvae_comm_rank(const mvae_comm_t *c) { return c ? c->rank : MVAE_ERR_ARG; }
MVAE_EXPORT int mvae_comm_world(const mvae_comm_t *c) { return c ? c->world : MVAE_ERR_ARG; }
MVAE_EXPORT const char *mvae_comm_last_error(const mvae_comm_t *c) { return c ? c->last_error : "null communicator"; }

// buf[0..count) <- sum over ranks, in place, after everything enqueued on `stream` so far.
MVAE_EXPORT int mvae_comm_allreduce_async(mvae_comm_t *c, float *buf, size_t count, mvae_stream_t stream, int *ticket) {
    if (!c || !buf || count == 0 || !ticket) return MVAE_ERR_ARG;
    const RcclApi *api = rccl();
    if (!api) return MVAE_ERR_COMM;
    const int slot = (int)(c->issued % COMM_SLOTS);
    int rc = hip_check(c, hipEventRecord(c->ready[slot], (hipStream_t)stream), "hipEventRecord(ready)");
    bool forked = false;            // the communication stream now depends on `stream` (under capture: it is part of the capture)
    if (rc == MVAE_OK) {
        rc = hip_check(c, hipStreamWaitEvent(c->stream, c->ready[slot], 0), "hipStreamWaitEvent(comm)");
        forked = rc == MVAE_OK;
    }
    // Algorithm.  Default: ONE ncclAllReduce per bucket -- RCCL picks ring / tree and the channel count from the message size.
    // MVAE_COMM_ALGO=rs_ag (opt-in, read at mvae_comm_init): the same sum as an in-place ncclReduceScatter + ncclAllGather over
    // the bucket's whole multiples of the world size (the < world-size remainder goes through ncclAllReduce).  On a full xGMI
    // mesh a direct reduce-scatter / all-gather moves S / world bytes over EVERY link at once, where one ring pushes
    // 2 (N - 1) / N x S through a single link per rank ("prefer direct over ring"; DESIGN section 6's
    // table prices both).  It is a switch, not a default: no multi-GPU node was available to measure it on.
    const size_t per = c->rs_ag ? count / (size_t)c->world : 0;
    if (rc == MVAE_OK && per > 0) {
        float *mine = buf + (size_t)c->rank * per;
        rc = nccl_check(c, api, api->ReduceScatter(buf, mine, per, ncclFloat32, ncclSum, c->nccl, c->stream), "ncclReduceScatter");
        if (rc == MVAE_OK)
            rc = nccl_check(c, api, api->AllGather(mine, buf, per, ncclFloat32, c->nccl, c->stream), "ncclAllGather");
        const size_t done_n = per * (size_t)c->world;
        if (rc == MVAE_OK && done_n < count)
            rc = nccl_check(c, api, api->AllReduce(buf + done_n, buf + done_n, count - done_n, ncclFloat32, ncclSum, c->nccl, c->stream),
                            "ncclAllReduce(tail)");
    } else if (rc == MVAE_OK) {
        rc = nccl_check(c, api, api->AllReduce(buf, buf, count, ncclFloat32, ncclSum, c->nccl, c->stream), "ncclAllReduce");
    }
    if (rc == MVAE_OK) rc = hip_check(c, hipEventRecord(c->done[slot], c->stream), "hipEventRecord(done)");
    if (rc != MVAE_OK) {
        // a partial enqueue must not leave the communication stream forked into a capture with no way back: join it
        // into the caller's stream (best effort; last_error keeps the first failure), so EndCapture sees one tail
        if (forked && hipEventRecord(c->done[slot], c->stream) == hipSuccess)
            (void)hipStreamWaitEvent((hipStream_t)stream, c->done[slot], 0);
        return rc;
    }
    *ticket = (int)c->issued;          // 2^31 collectives per process: ~10^8 steps at the deepest bucket plan
    c->issued++;
    return MVAE_OK;
}

// `stream` waits for the collective behind `ticket` (ticket < 0: the most recent one -- the communication
// stream is in order, so that is all of them).
MVAE_EXPORT int mvae_comm_wait(mvae_comm_t *c, int ticket, mvae_stream_t stream) {
    if (!c) return MVAE_ERR_ARG;
    if (c->issued == 0) return MVAE_OK;
    const long t = ticket < 0 ? c->issued - 1 : (long)ticket;
    if (t >= c->issued || c->issued - t > COMM_SLOTS) return MVAE_ERR_ARG;       // never issued / slot reused since
    return hip_check(c, hipStreamWaitEvent((hipStream_t)stream, c->done[t % COMM_SLOTS], 0), "hipStreamWaitEvent(consumer)");
}

// buf <- rank `root`'s buf, ordered after `stream`'s work; `stream` waits for it (parameters / buffers at start-up).
MVAE_EXPORT int mvae_comm_broadcast(mvae_comm_t *c, void *buf, size_t bytes, int root, mvae_stream_t stream) {
    if (!c || !buf || bytes == 0 || root < 0 || root >= c->world) return MVAE_ERR_ARG;
    const RcclApi *api = rccl();
    if (!api) return MVAE_ERR_COMM;
    const int slot = (int)(c->issued % COMM_SLOTS);
    int rc = hip_check(c, hipEventRecord(c->ready[slot], (hipStream_t)stream), "hipEventRecord(ready)");
    if (rc == MVAE_OK) rc = hip_check(c, hipStreamWaitEvent(c->stream, c->ready[slot], 0), "hipStreamWaitEvent(comm)");
    if (rc == MVAE_OK)
        rc = nccl_check(c, api, api->Broadcast(buf, buf, bytes, ncclUint8, root, c->nccl, c->stream), "ncclBroadcast");
    if (rc == MVAE_OK) rc = hip_check(c, hipEventRecord(c->done[slot], c->stream), "hipEventRecord(done)");
    if (rc == MVAE_OK) rc = hip_check(c, hipStreamWaitEvent((hipStream_t)stream, c->done[slot], 0), "hipStreamWaitEvent(consumer)");
    if (rc == MVAE_OK) c->issued++;
    return rc;
}

// Has a collective of this communicator failed asynchronously (a peer died, a link error)?  MVAE_OK / MVAE_ERR_COMM.
MVAE_EXPORT int mvae_comm_async_error(mvae_comm_t *c) {
    if (!c) return MVAE_ERR_ARG;
    const RcclApi *api = rccl();
    if (!api) return MVAE_ERR_COMM;
    if (!api->CommGetAsyncError) return MVAE_OK;
    ncclResult_t async = ncclSuccess;
    const ncclResult_t r = api->CommGetAsyncError(c->nccl, &async);
    if (r != ncclSuccess) return nccl_check(c, api, r, "ncclCommGetAsyncError");
    if (async == ncclSuccess || async == ncclInProgress) return MVAE_OK;
    return nccl_check(c, api, async, "asynchronous communicator error");
}

// The watchdog: block the HOST until everything enqueued on `stream` so far has finished -- which, after
// mvae_comm_wait, includes the collectives -- but for at most `timeout_ms`, polling the communicator's asynchronous
// error state meanwhile.  A collective whose peer is gone never completes; hipStreamSynchronize would hang for good.
// MVAE_OK: finished, no error.  MVAE_ERR_COMM: a peer failed or the budget ran out (text: mvae_comm_last_error);
// the communicator must then be abandoned (the process cannot cancel the work still enqueued).  Not capturable.
MVAE_EXPORT int mvae_comm_synchronize(mvae_comm_t *c, mvae_stream_t stream, int timeout_ms) {
    if (!c || timeout_ms < 0) return MVAE_ERR_ARG;
    hipEvent_t ev;
    int rc = hip_check(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate(watchdog)");
    if (rc != MVAE_OK) return rc;
    rc = hip_check(c, hipEventRecord(ev, (hipStream_t)stream), "hipEventRecord(watchdog)");
    const auto t0 = std::chrono::steady_clock::now();
    unsigned polls = 0;
    while (rc == MVAE_OK) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) { rc = mvae_comm_async_error(c); break; }      // done: but did it finish by giving up on a peer?
        if (q != hipErrorNotReady) { rc = hip_check(c, q, "hipEventQuery(watchdog)"); break; }
        if ((++polls & 63u) == 0) {
            rc = mvae_comm_async_error(c);
            if (rc != MVAE_OK) break;
            const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (ms > timeout_ms) {
                char msg[96];
                snprintf(msg, sizeof(msg), "stream work incl. collectives not finished within %d ms", timeout_ms);
                rc = comm_fail(c, "watchdog", msg);
                break;
            }
        }
        sched_yield();
    }
    // an event with work still pending behind it is released by the runtime when that work retires
    (void)hipEventDestroy(ev);
    return rc;
}

MVAE_EXPORT int mvae_comm_destroy(mvae_comm_t *c) {
    if (!c) return MVAE_ERR_ARG;
    const RcclApi *api = rccl();
    (void)hipStreamSynchronize(c->stream);
    int rc = MVAE_OK;
    if (api) rc = nccl_check(c, api, api->CommDestroy(c->nccl), "ncclCommDestroy");
    for (int i = 0; i < COMM_SLOTS; ++i) { (void)hipEventDestroy(c->ready[i]); (void)hipEventDestroy(c->done[i]); }
    (void)hipStreamDestroy(c->stream);
    free(c);
    return rc;
}
