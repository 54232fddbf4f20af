// shm_nccl.cpp -- TEST INFRASTRUCTURE: the subset of the NCCL/RCCL ABI that csrc/comm.hip binds
// (ncclGetUniqueId, ncclCommInitRank, ncclAllReduce, ncclBroadcast, ncclCommDestroy, ncclGetVersion,
// ncclGetErrorString, ncclCommGetAsyncError, and for MVAE_COMM_ALGO=rs_ag ncclReduceScatter / ncclAllGather), implemented over POSIX shared memory + host staging, so that
// N ranks can share ONE GPU.
//
// Why: RCCL refuses two ranks on one device, the build pool has one-GPU boxes, and the default data-parallel
// transport -- mvae_comm_* tickets / events with the collectives captured INSIDE the step's hipGraph -- had only
// ever run at world size 1 ("What's missing" 1).  Loaded through mvae_comm_use_library() /
// MVAE_RCCL_LIB, this library lets tests/test_comm_world2_gpu.py run that exact code path with two processes on
// cuda:0.  It is NOT a product transport: every collective stages through the host.
//
// A collective is a sequence of stream operations on the caller's stream, all capturable:
//   per chunk of <= CHUNK bytes:  D2H copy into a pinned bounce buffer  ->  host function  ->  H2D copy back.
// The host function publishes the chunk in this rank's shared slot, meets the peers at a barrier, sums the slots
// in RANK ORDER (so every rank computes bit-identical sums) or copies the root's, and meets them again before the
// slot is reused.  Barriers are epoch counters in the shared control block; a peer that does not arrive within
// MVAE_SHMNCCL_TIMEOUT_S (default 20 s) -- or that has raised the shared abort flag -- fails the communicator:
// the remaining operations of the stream become no-ops and ncclCommGetAsyncError reports ncclRemoteError.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types, enums and the prototypes the definitions below must match

namespace {

constexpr int MAX_RANKS = 8;
constexpr size_t CHUNK = 4u << 20;                 // bytes per rank slot (and per staged chunk)
constexpr uint32_t MAGIC = 0x4d564145u;            // "MVAE"

struct Ctl {                                       // lives at the start of the shared segment (zero-filled by ftruncate)
    std::atomic<uint32_t> magic;
    std::atomic<int> attached;                     // ranks that have mapped the segment
    std::atomic<int> abort_flag;                   // any rank: "I gave up" -- peers stop waiting
    std::atomic<uint64_t> arrive[MAX_RANKS];       // per rank: barriers passed so far
};
constexpr size_t CTL_BYTES = 4096;
static_assert(sizeof(Ctl) <= CTL_BYTES, "control block");

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double timeout_s() {
    const char *e = getenv("MVAE_SHMNCCL_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 20.0;
}

struct Op;

struct Comm {
    int rank = 0, world = 1;
    char name[64] = {0};
    void *map = nullptr; size_t map_bytes = 0;
    Ctl *ctl = nullptr;
    char *slots = nullptr;                          // world x CHUNK
    char *bounce = nullptr;                         // pinned, CHUNK bytes
    uint64_t epoch = 0;                             // barriers this rank has entered (host-function thread only)
    std::atomic<int> failed{0};
    std::mutex ops_mutex;
    std::vector<Op *> ops;                          // descriptors of enqueued host functions: a captured graph re-runs them, so
                                                    // they live until the communicator is destroyed
    char *slot(int r) const { return slots + (size_t)r * CHUNK; }
};

struct Op { Comm *c; int kind; size_t bytes; int root; };     // kind 0: fp32 sum, 1: broadcast, 2: fp32 sum kept by `root` only

void fail(Comm *c, const char *why) {
    if (!c->failed.exchange(1)) fprintf(stderr, "[shm_nccl rank %d/%d] communicator failed: %s\n", c->rank, c->world, why);
    c->ctl->abort_flag.store(1, std::memory_order_release);
}

// every rank has entered its `epoch`-th barrier
bool barrier(Comm *c) {
    if (c->failed.load()) return false;
    const uint64_t e = ++c->epoch;
    c->ctl->arrive[c->rank].store(e, std::memory_order_release);
    const double deadline = now_s() + timeout_s();
    for (int r = 0; r < c->world; ++r) {
        unsigned spins = 0;
        while (c->ctl->arrive[r].load(std::memory_order_acquire) < e) {
            if (c->ctl->abort_flag.load(std::memory_order_acquire)) { fail(c, "a peer aborted"); return false; }
            if ((++spins & 1023u) == 0) {
                if (now_s() > deadline) { fail(c, "a peer did not reach the collective in time"); return false; }
                usleep(50);
            }
        }
    }
    return true;
}

void host_step(void *arg) {
    Op *op = (Op *)arg;
    Comm *c = op->c;
    if (c->failed.load()) return;
    if (op->kind == 0 || op->kind == 2) {
        memcpy(c->slot(c->rank), c->bounce, op->bytes);
        if (!barrier(c)) return;
        const size_t n = op->bytes / sizeof(float);
        float *out = (float *)c->bounce;
        const float *s0 = (const float *)c->slot(0);
        for (size_t i = 0; i < n; ++i) out[i] = s0[i];
        for (int r = 1; r < c->world; ++r) {            // rank order: the same sum, bit for bit, on every rank
            const float *sr = (const float *)c->slot(r);
            for (size_t i = 0; i < n; ++i) out[i] += sr[i];
        }
        (void)barrier(c);                               // everyone has read every slot: they may be rewritten
    } else {
        if (c->rank == op->root) memcpy(c->slot(op->root), c->bounce, op->bytes);
        if (!barrier(c)) return;
        if (c->rank != op->root) memcpy(c->bounce, c->slot(op->root), op->bytes);
        (void)barrier(c);
    }
}

ncclResult_t enqueue(Comm *c, int kind, const void *send, void *recv, size_t bytes, int root, hipStream_t st) {
    if (!c || !send || !recv) return ncclInvalidArgument;
    if (c->failed.load()) return ncclRemoteError;
    for (size_t off = 0; off < bytes; off += CHUNK) {
        const size_t nb = bytes - off < CHUNK ? bytes - off : CHUNK;
        Op *op = new Op{c, kind, nb, root};
        { std::lock_guard<std::mutex> lock(c->ops_mutex); c->ops.push_back(op); }
        if (hipMemcpyAsync(c->bounce, (const char *)send + off, nb, hipMemcpyDeviceToHost, st) != hipSuccess) return ncclUnhandledCudaError;
        if (hipLaunchHostFunc(st, host_step, op) != hipSuccess) return ncclUnhandledCudaError;
        if (kind == 2 && c->rank != root) continue;                 // a reduce-scatter piece: only its owner keeps the sum
        if (hipMemcpyAsync((char *)recv + off, c->bounce, nb, hipMemcpyHostToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) ncclResult_t ncclGetVersion(int *version) {
    if (!version) return ncclInvalidArgument;
    *version = 10000;       // "1.0.0": not an RCCL
    return ncclSuccess;
}

__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (shm_nccl)";
        case ncclSystemError: return "system error (shm_nccl: shared segment / rendezvous)";
        case ncclInvalidArgument: return "invalid argument (shm_nccl)";
        case ncclRemoteError: return "remote error (shm_nccl: a peer left or timed out)";
        default: return "error (shm_nccl)";
    }
}

// rank 0 creates the segment; its name is the id
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    static std::atomic<unsigned> serial{0};
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/mvae_shmnccl_%d_%u_%llx", (int)getpid(), serial++,
             (unsigned long long)(now_s() * 1e6));
    const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    const size_t bytes = CTL_BYTES + (size_t)MAX_RANKS * CHUNK;
    const int rc = ftruncate(fd, (off_t)bytes);
    close(fd);
    if (rc != 0) { shm_unlink(id->internal); return ncclSystemError; }
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    Comm *c = new Comm;
    c->rank = rank; c->world = nranks;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    const int fd = shm_open(c->name, O_RDWR, 0600);
    if (fd < 0) { delete c; return ncclSystemError; }
    c->map_bytes = CTL_BYTES + (size_t)MAX_RANKS * CHUNK;
    c->map = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->map == MAP_FAILED) { delete c; return ncclSystemError; }
    c->ctl = (Ctl *)c->map;
    c->slots = (char *)c->map + CTL_BYTES;
    if (hipHostMalloc((void **)&c->bounce, CHUNK, hipHostMallocDefault) != hipSuccess) {
        munmap(c->map, c->map_bytes); delete c; return ncclUnhandledCudaError;
    }
    c->ctl->magic.store(MAGIC);
    c->ctl->attached.fetch_add(1);
    const double deadline = now_s() + timeout_s();
    while (c->ctl->attached.load() < nranks) {
        if (now_s() > deadline || c->ctl->abort_flag.load()) {
            c->ctl->abort_flag.store(1);
            (void)hipHostFree(c->bounce); munmap(c->map, c->map_bytes); delete c;
            if (rank == 0) shm_unlink(id.internal);
            return ncclSystemError;
        }
        usleep(200);
    }
    if (rank == 0) shm_unlink(c->name);       // everyone holds a mapping: the name can go, nothing leaks if a rank dies
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm *c = (Comm *)comm;
    if (!c) return ncclInvalidArgument;
    for (Op *op : c->ops) delete op;
    (void)hipHostFree(c->bounce);
    munmap(c->map, c->map_bytes);
    delete c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t *async_error) {
    Comm *c = (Comm *)comm;
    if (!c || !async_error) return ncclInvalidArgument;
    *async_error = c->failed.load() ? ncclRemoteError : ncclSuccess;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt,
                                                                   ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
    if (dt != ncclFloat32 || op != ncclSum) return ncclInvalidArgument;       // all comm.hip asks for
    return enqueue((Comm *)comm, 0, send, recv, count * sizeof(float), 0, st);
}

__attribute__((visibility("default"))) ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t dt,
                                                                   int root, ncclComm_t comm, hipStream_t st) {
    Comm *c = (Comm *)comm;
    if (dt != ncclUint8 || !c || root < 0 || root >= c->world) return ncclInvalidArgument;
    return enqueue(c, 1, send, recv, count, root, st);
}

// piece p of the send buffer (recvcount floats) summed over the ranks in rank order, delivered to rank p
__attribute__((visibility("default"))) ncclResult_t ncclReduceScatter(const void *send, void *recv, size_t recvcount, ncclDataType_t dt,
                                                                       ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
    Comm *c = (Comm *)comm;
    if (dt != ncclFloat32 || op != ncclSum || !c) return ncclInvalidArgument;
    for (int p = 0; p < c->world; ++p) {
        const ncclResult_t r = enqueue(c, 2, (const float *)send + (size_t)p * recvcount, recv, recvcount * sizeof(float), p, st);
        if (r != ncclSuccess) return r;
    }
    return ncclSuccess;
}

// rank p's sendcount floats land at recv + p * sendcount on every rank (a broadcast per rank)
__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t dt,
                                                                   ncclComm_t comm, hipStream_t st) {
    Comm *c = (Comm *)comm;
    if (dt != ncclFloat32 || !c) return ncclInvalidArgument;
    for (int p = 0; p < c->world; ++p) {
        float *dst = (float *)recv + (size_t)p * sendcount;
        const ncclResult_t r = enqueue(c, 1, c->rank == p ? send : (const void *)dst, dst, sendcount * sizeof(float), p, st);
        if (r != ncclSuccess) return r;
    }
    return ncclSuccess;
}

}  // extern "C"
