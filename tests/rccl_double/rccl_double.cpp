// rccl_double.cpp -- TEST DOUBLE of the RCCL entry points libbpmf_hip.so resolves at run time
// (bpmf_amd/csrc/capi.hip: rccl()), for ranks that SHARE ONE GPU.
//
// Why: RCCL refuses two ranks on one device ("Duplicate GPU detected"), and the GPU boxes of this
// project have one MI355X.  With BPMF_HIP_RCCL_LIBRARY=<this .so> the library's multi-rank code --
// the mesh of grouped ncclSend / ncclRecv (launch_impl.h: exchange), the per-part exchange on a
// second stream, the second communicator of the statistics streams, the grouped ncclReduce of the
// BPMF_REDUCE formulation, the packed connectivity-aware lists -- runs with nranks >= 2, the ranks
// being threads of one process (`bpmf -g 2`) or separate processes (bench.py / tests/_mr_worker.py).
// This file is test infrastructure: nothing under bpmf_amd/ links or loads it unless that variable
// names it, and it is never the thing measured.
//
// How: a communicator is a POSIX shared-memory segment (named by the 128-byte unique id) with
//   * one mailbox per ordered pair of ranks: a message travels in chunks -- the sender waits for the
//     slot to be free, copies a chunk device -> slot, posts it; the receiver copies slot -> device and
//     frees it.  The first chunk carries the message size: a count mismatch between a send and its
//     receive is an error (ncclInvalidArgument), not a silent truncation;
//   * one slot per rank for reductions: every rank deposits its piece, a barrier, every rank (or the
//     root) adds the pieces IN RANK ORDER on the host -- the same bits on every rank, like a ring
//     all-reduce -- and copies the sum back.
// Semantics are those of NCCL with every operation completed before the call (or ncclGroupEnd)
// returns: the stream of an operation is synchronised first, so what was enqueued before it is
// visible, and what is enqueued after it sees the result.  Stronger than the real library (no
// overlap), never weaker.  Every wait is bounded (BPMF_RCCL_DOUBLE_TIMEOUT_S, default 60): a
// mismatched call order between ranks -- the deadlock a real run would hang in -- ends as
// ncclSystemError with a message on stderr.
//
// Ranks of one communicator must issue their operations in the same program order (NCCL's own rule).
//
// ASYNCHRONOUS MODE (BPMF_RCCL_DOUBLE_ASYNC=1; round 4).  The mode above cannot see what the real library would
// punish: with it every operation is over when the call returns, so a caller that reads a result early, reuses a
// buffer, or relies on two communicators / streams progressing in a particular relative order is never caught.  In
// asynchronous mode a call (or ncclGroupEnd) only ENQUEUES, like NCCL:
//   * a one-lane kernel (k_gate) goes onto the operation's stream: it tells the host "the stream has reached the
//     operation" (everything enqueued before it is complete) and then holds the stream until the host says "done" --
//     what an NCCL kernel does to its stream while it waits for its peers;
//   * a helper thread per communicator takes the enqueued groups in order, waits for the stream(s) to arrive, sleeps a
//     random, rank-dependent time (BPMF_RCCL_DOUBLE_ASYNC_DELAY_US, default up to 300 us: the two communicators and the
//     exchange stream of a rank interleave differently from its peers' and from run to run), moves the data with copies
//     on a stream of its own, and releases the kernel.
// The caller returns at once with ncclSuccess; a failure makes the communicator's error sticky (every later call fails)
// and releases the streams.  ncclCommAbort fails everything in flight on every rank of the communicator.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int MAXR = 8;
constexpr size_t CHUNK = 2u << 20;           // bytes per mailbox slot / reduction slot
constexpr size_t RCHUNK = CHUNK - 64;        // payload of a reduction slot (its last word carries the element count)
constexpr uint64_t MAGIC = 0x62706d6672636331ull;

struct Mailbox {
    std::atomic<uint64_t> posted;            // chunks written by the sender
    std::atomic<uint64_t> consumed;          // chunks read by the receiver
    std::atomic<uint64_t> total_bytes;       // of the message whose first chunk is in the slot
    std::atomic<uint64_t> chunk_bytes;
    char pad[32];
};

struct Header {
    std::atomic<uint64_t> magic;
    std::atomic<int> nranks;
    std::atomic<int> attached;               // ranks that have mapped the segment
    std::atomic<int> detached;
    std::atomic<int> bar_count;
    std::atomic<int> bar_sense;
    std::atomic<int> nsplit;                 // communicators split off so far (names the child segment)
    std::atomic<int> error;                  // sticky: some rank gave up
    char pad[64];
    Mailbox box[MAXR][MAXR];                 // [src][dst]
};

size_t seg_bytes(int n) { return sizeof(Header) + (size_t)n * n * CHUNK + (size_t)n * CHUNK; }

double timeout_s()
{
    static const double v = [] { const char *e = getenv("BPMF_RCCL_DOUBLE_TIMEOUT_S"); return (e && *e) ? atof(e) : 60.0; }();
    return v;
}

bool async_mode()
{
    static const bool v = [] { const char *e = getenv("BPMF_RCCL_DOUBLE_ASYNC"); return e && *e && atoi(e) != 0; }();
    return v;
}

struct Group;

struct Comm {
    std::string name;
    Header *h = nullptr;
    size_t bytes = 0;
    int nranks = 0, rank = 0;
    int local_sense = 0;
    uint64_t ops = 0;                        // operations completed (diagnostics)
    // asynchronous mode: the helper thread of this communicator, its queue of enqueued groups, its copy stream, and the
    // (reached, done) flag pairs the gate kernels and the helper talk through (pinned host memory)
    int device = 0;
    std::thread helper;
    std::mutex m;
    std::condition_variable cv;
    std::deque<Group *> q;
    bool busy = false, stop = false;
    hipStream_t hs = nullptr;
    unsigned *flags = nullptr, *flags_dev = nullptr;
    size_t next_flag = 0;
    char *slot(int src, int dst) const { return reinterpret_cast<char *>(h) + sizeof(Header) + ((size_t)src * nranks + dst) * CHUNK; }
    char *red_slot(int r) const { return reinterpret_cast<char *>(h) + sizeof(Header) + (size_t)nranks * nranks * CHUNK + (size_t)r * CHUNK; }
};

constexpr size_t NFLAG = 8192;              // flag pairs in flight per communicator (a ring)

struct Op {
    enum Kind { SEND, RECV, ALLREDUCE, REDUCE, COPY } kind;
    Comm *comm;
    const void *src;
    void *dst;
    size_t bytes;
    int peer;                                // SEND / RECV: the other rank; REDUCE: the root
    ncclDataType_t type;
    hipStream_t stream;
};

struct Group {
    std::vector<Op> ops;
    std::vector<size_t> flag;                // one pair per distinct stream of the group
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
// set on a communicator's helper thread: copies go to its own stream (the operation's stream is held by the gate kernel)
thread_local hipStream_t t_copy_stream = nullptr;
thread_local bool t_async = false;

int complain(Comm *c, const char *what)
{
    fprintf(stderr, "[rccl_double] rank %d of %d (%s, op %llu): %s\n", c ? c->rank : -1, c ? c->nranks : -1, c ? c->name.c_str() : "-",
            (unsigned long long)(c ? c->ops : 0), what);
    if (c && c->h) c->h->error.store(1);
    return 1;
}

struct Deadline {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    // false: keep waiting; true: give up
    bool expired(Comm *c)
    {
        if ((++spins & 63u) == 0) {
            sched_yield();
            if (c->h->error.load(std::memory_order_relaxed)) return true;
            if ((spins & 0xFFFu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) return true;
        }
        return false;
    }
};

ncclResult_t barrier(Comm *c)
{
    Header *h = c->h;
    const int sense = c->local_sense ^= 1;
    if (h->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {
        h->bar_count.store(0, std::memory_order_relaxed);
        h->bar_sense.store(sense, std::memory_order_release);
        return ncclSuccess;
    }
    Deadline d;
    while (h->bar_sense.load(std::memory_order_acquire) != sense)
        if (d.expired(c)) { complain(c, "barrier: the other ranks never arrived (different call order between ranks?)"); return ncclSystemError; }
    return ncclSuccess;
}

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

#define HIP_OK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "[rccl_double] %s: %s\n", #expr, hipGetErrorString(e_));           \
            return ncclUnhandledCudaError;                                                     \
        }                                                                                      \
    } while (0)

ncclResult_t copy_d2h(void *dst, const void *src, size_t n)
{
    if (!t_async) { HIP_OK(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost)); return ncclSuccess; }
    HIP_OK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, t_copy_stream));
    HIP_OK(hipStreamSynchronize(t_copy_stream));
    return ncclSuccess;
}
ncclResult_t copy_to_dev(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t op_stream)
{
    hipStream_t st = t_async ? t_copy_stream : op_stream;
    HIP_OK(hipMemcpyAsync(dst, src, n, kind, st));
    HIP_OK(hipStreamSynchronize(st));
    return ncclSuccess;
}
#define NCCL_OK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return r_; } while (0)

// all point-to-point operations of one group, progressed together (a rank that posts its sends first
// and a peer that does the same must not wait for each other: chunks move whenever a slot allows it)
ncclResult_t run_p2p(std::vector<Op> &ops)
{
    struct Xfer { Op *op; size_t done = 0; bool started = false; };
    std::vector<std::vector<Xfer>> sendq(MAXR), recvq(MAXR);       // per peer, in issue order (one communicator per group)
    std::vector<size_t> shead(MAXR, 0), rhead(MAXR, 0);
    Comm *c = nullptr;
    size_t pending = 0;
    for (Op &o : ops) {
        if (o.kind != Op::SEND && o.kind != Op::RECV) continue;
        if (c && o.comm != c) { complain(o.comm, "a group with point-to-point operations of two communicators is not supported"); return ncclInvalidUsage; }
        c = o.comm;
        if (o.peer < 0 || o.peer >= c->nranks || o.peer == c->rank) { complain(c, "send / recv: bad peer"); return ncclInvalidArgument; }
        (o.kind == Op::SEND ? sendq : recvq)[(size_t)o.peer].push_back(Xfer{&o});
        ++pending;
    }
    if (!c) return ncclSuccess;
    Deadline d;
    while (pending) {
        bool moved = false;
        for (int p = 0; p < c->nranks; ++p) {
            if (shead[(size_t)p] < sendq[(size_t)p].size()) {
                Xfer &x = sendq[(size_t)p][shead[(size_t)p]];
                Mailbox &b = c->h->box[c->rank][p];
                if (b.posted.load(std::memory_order_acquire) == b.consumed.load(std::memory_order_acquire)) {      // slot free
                    const size_t n = std::min(CHUNK, x.op->bytes - x.done);
                    if (n) NCCL_OK(copy_d2h(c->slot(c->rank, p), static_cast<const char *>(x.op->src) + x.done, n));
                    b.total_bytes.store(x.op->bytes, std::memory_order_relaxed);
                    b.chunk_bytes.store(n, std::memory_order_relaxed);
                    b.posted.fetch_add(1, std::memory_order_release);
                    x.done += n; x.started = true; moved = true;
                    if (x.done == x.op->bytes) { ++shead[(size_t)p]; --pending; }
                }
            }
            if (rhead[(size_t)p] < recvq[(size_t)p].size()) {
                Xfer &x = recvq[(size_t)p][rhead[(size_t)p]];
                Mailbox &b = c->h->box[p][c->rank];
                if (b.posted.load(std::memory_order_acquire) > b.consumed.load(std::memory_order_acquire)) {       // a chunk waits
                    const size_t n = b.chunk_bytes.load(std::memory_order_relaxed);
                    if (b.total_bytes.load(std::memory_order_relaxed) != x.op->bytes || x.done + n > x.op->bytes) {
                        char msg[256];
                        snprintf(msg, sizeof msg, "recv from rank %d expects %zu bytes, the matching send carries %llu: the ranks disagree about a message size",
                                 p, x.op->bytes, (unsigned long long)b.total_bytes.load());
                        complain(c, msg);
                        return ncclInvalidArgument;
                    }
                    if (n) NCCL_OK(copy_to_dev(static_cast<char *>(x.op->dst) + x.done, c->slot(p, c->rank), n, hipMemcpyHostToDevice, x.op->stream));
                    b.consumed.fetch_add(1, std::memory_order_release);
                    x.done += n; moved = true;
                    if (x.done == x.op->bytes) { ++rhead[(size_t)p]; --pending; }
                }
            }
        }
        if (moved) { d = Deadline(); continue; }
        if (d.expired(c)) { complain(c, "send / recv: the peer never posted the matching operation"); return ncclSystemError; }
    }
    // a sender may only return once its last chunks were taken (the slot is reused, and the caller may overwrite the buffer -- that
    // part is already safe: the chunk left the device -- but a later barrier-free operation must find the mailbox empty)
    for (int p = 0; p < c->nranks; ++p) {
        if (sendq[(size_t)p].empty()) continue;
        Mailbox &b = c->h->box[c->rank][p];
        Deadline dd;
        while (b.posted.load(std::memory_order_acquire) != b.consumed.load(std::memory_order_acquire))
            if (dd.expired(c)) { complain(c, "send: the peer never took the last chunk"); return ncclSystemError; }
    }
    return ncclSuccess;
}

template <typename T>
void add_into(T *acc, const T *x, size_t n) { for (size_t i = 0; i < n; ++i) acc[i] += x[i]; }

// sum over the ranks, pieces of CHUNK bytes through the reduction slots; root < 0: every rank gets the result
ncclResult_t run_reduce(const Op &o)
{
    Comm *c = o.comm;
    const size_t esz = type_bytes(o.type);
    if (!(o.type == ncclFloat64 || o.type == ncclFloat32 || o.type == ncclInt64 || o.type == ncclInt32 || o.type == ncclUint64)) {
        complain(c, "reduction: data type not supported by the test double");
        return ncclInvalidArgument;
    }
    const int root = o.kind == Op::REDUCE ? o.peer : -1;
    std::vector<char> acc(CHUNK);
    for (size_t off = 0; off < o.bytes || off == 0; off += RCHUNK) {
        const size_t n = std::min(RCHUNK, o.bytes - off);
        if (n) NCCL_OK(copy_d2h(c->red_slot(c->rank), static_cast<const char *>(o.src) + off, n));
        *reinterpret_cast<volatile uint64_t *>(c->red_slot(c->rank) + CHUNK - 8) = (uint64_t)o.bytes;      // (size check)
        ncclResult_t r = barrier(c);
        if (r != ncclSuccess) return r;
        for (int q = 0; q < c->nranks; ++q)
            if (*reinterpret_cast<volatile uint64_t *>(c->red_slot(q) + CHUNK - 8) != (uint64_t)o.bytes) {
                complain(c, "reduction: the ranks disagree about the element count");
                return ncclInvalidArgument;
            }
        if ((root < 0 || root == c->rank) && n) {
            memcpy(acc.data(), c->red_slot(0), n);
            for (int q = 1; q < c->nranks; ++q) {
                const size_t cnt = n / esz;
                switch (o.type) {
                case ncclFloat64: add_into(reinterpret_cast<double *>(acc.data()), reinterpret_cast<const double *>(c->red_slot(q)), cnt); break;
                case ncclFloat32: add_into(reinterpret_cast<float *>(acc.data()), reinterpret_cast<const float *>(c->red_slot(q)), cnt); break;
                case ncclInt64: case ncclUint64: add_into(reinterpret_cast<uint64_t *>(acc.data()), reinterpret_cast<const uint64_t *>(c->red_slot(q)), cnt); break;
                default: add_into(reinterpret_cast<uint32_t *>(acc.data()), reinterpret_cast<const uint32_t *>(c->red_slot(q)), cnt); break;
                }
            }
            NCCL_OK(copy_to_dev(static_cast<char *>(o.dst) + off, acc.data(), n, hipMemcpyHostToDevice, o.stream));
        }
        r = barrier(c);                                             // (the slots are rewritten by the next piece / operation)
        if (r != ncclSuccess) return r;
        if (o.bytes == 0) break;
    }
    return ncclSuccess;
}

ncclResult_t run_group(std::vector<Op> &ops)
{
    // everything enqueued before the operations must be visible to the copies below
    // (asynchronous mode: the gate kernels have reported that their streams arrived; the streams themselves are held)
    std::vector<hipStream_t> seen;
    for (const Op &o : ops) {
        bool dup = false;
        for (hipStream_t s : seen) dup = dup || s == o.stream;
        if (!dup && !t_async) { seen.push_back(o.stream); HIP_OK(hipStreamSynchronize(o.stream)); }
    }
    for (Op &o : ops)
        if (o.kind == Op::COPY && o.bytes && o.src != o.dst) NCCL_OK(copy_to_dev(o.dst, o.src, o.bytes, hipMemcpyDeviceToDevice, o.stream));
    ncclResult_t r = run_p2p(ops);
    if (r != ncclSuccess) return r;
    for (Op &o : ops)
        if (o.kind == Op::ALLREDUCE || o.kind == Op::REDUCE) {
            r = run_reduce(o);
            if (r != ncclSuccess) return r;
        }
    for (Op &o : ops) o.comm->ops++;
    return ncclSuccess;
}

// ---- asynchronous mode ---------------------------------------------------------------------------------
// reached = 1: everything ahead of the operation on this stream is complete; then the stream is held until done != 0
// (bounded: a helper that died must not leave a wave spinning for ever)
__global__ void k_gate(unsigned *reached, unsigned *done, unsigned long long max_ticks)
{
    __hip_atomic_store(reached, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();                    // 100 MHz
    while (__hip_atomic_load(done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > max_ticks) break;
    }
}

void helper_main(Comm *c)
{
    (void)hipSetDevice(c->device);
    t_async = true; t_copy_stream = c->hs;
    static const int max_delay_us = [] { const char *e = getenv("BPMF_RCCL_DOUBLE_ASYNC_DELAY_US"); return (e && *e) ? atoi(e) : 300; }();
    std::mt19937 rng((unsigned)(c->rank * 7919 + std::hash<std::string>()(c->name) + (unsigned)getpid()));
    for (;;) {
        Group *g = nullptr;
        {
            std::unique_lock<std::mutex> lk(c->m);
            c->cv.wait(lk, [c] { return c->stop || !c->q.empty(); });
            if (c->q.empty()) return;
            g = c->q.front(); c->q.pop_front(); c->busy = true;
        }
        ncclResult_t r = ncclSuccess;
        for (size_t f : g->flag) {                                   // the streams of the group arrive at the operation
            Deadline d;
            while (__atomic_load_n(&c->flags[2 * f], __ATOMIC_ACQUIRE) == 0u)
                if (d.expired(c)) { complain(c, "asynchronous mode: a stream never reached its operation (held by another operation that cannot complete?)"); r = ncclSystemError; break; }
            if (r != ncclSuccess) break;
        }
        if (r == ncclSuccess && max_delay_us > 0) usleep((useconds_t)(rng() % (unsigned)max_delay_us));
        if (r == ncclSuccess && c->h->error.load()) r = ncclSystemError;
        if (r == ncclSuccess) r = run_group(g->ops);
        if (r != ncclSuccess) c->h->error.store(1);
        for (size_t f : g->flag) __atomic_store_n(&c->flags[2 * f + 1], 1u, __ATOMIC_RELEASE);     // release the streams (also after a failure)
        delete g;
        { std::lock_guard<std::mutex> lk(c->m); c->busy = false; }
        c->cv.notify_all();
    }
}

ncclResult_t enqueue_async(std::vector<Op> &ops)
{
    Comm *c = ops[0].comm;
    for (const Op &o : ops)
        if (o.comm != c) { complain(c, "asynchronous mode: one communicator per group"); return ncclInvalidUsage; }
    if (c->h->error.load()) return ncclSystemError;                   // sticky: something already failed (or the communicator was aborted)
    Group *g = new Group();
    g->ops = ops;
    std::vector<hipStream_t> seen;
    const unsigned long long max_ticks = (unsigned long long)((timeout_s() + 5.0) * 1e8);
    for (const Op &o : ops) {
        bool dup = false;
        for (hipStream_t s : seen) dup = dup || s == o.stream;
        if (dup) continue;
        seen.push_back(o.stream);
        const size_t f = c->next_flag++ % NFLAG;
        __atomic_store_n(&c->flags[2 * f], 0u, __ATOMIC_RELAXED);
        __atomic_store_n(&c->flags[2 * f + 1], 0u, __ATOMIC_RELEASE);
        g->flag.push_back(f);
        hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, o.stream, c->flags_dev + 2 * f, c->flags_dev + 2 * f + 1, max_ticks);
        if (hipGetLastError() != hipSuccess) { delete g; complain(c, "asynchronous mode: the gate kernel could not be launched"); return ncclUnhandledCudaError; }
    }
    { std::lock_guard<std::mutex> lk(c->m); c->q.push_back(g); }
    c->cv.notify_all();
    return ncclSuccess;
}

// everything enqueued on this communicator is over (asynchronous mode)
void drain(Comm *c)
{
    if (!c->helper.joinable()) return;
    std::unique_lock<std::mutex> lk(c->m);
    c->cv.wait(lk, [c] { return c->q.empty() && !c->busy; });
}

ncclResult_t start_helper(Comm *c)
{
    if (!async_mode()) return ncclSuccess;
    HIP_OK(hipGetDevice(&c->device));
    HIP_OK(hipStreamCreateWithFlags(&c->hs, hipStreamNonBlocking));
    HIP_OK(hipHostMalloc((void **)&c->flags, 2 * NFLAG * sizeof(unsigned), hipHostMallocMapped));
    HIP_OK(hipHostGetDevicePointer((void **)&c->flags_dev, c->flags, 0));
    memset(c->flags, 0, 2 * NFLAG * sizeof(unsigned));
    c->helper = std::thread(helper_main, c);
    return ncclSuccess;
}

void stop_helper(Comm *c)
{
    if (!c->helper.joinable()) return;
    drain(c);
    { std::lock_guard<std::mutex> lk(c->m); c->stop = true; }
    c->cv.notify_all();
    c->helper.join();
    if (c->hs) { (void)hipStreamSynchronize(c->hs); (void)hipStreamDestroy(c->hs); c->hs = nullptr; }
    if (c->flags) { (void)hipHostFree(c->flags); c->flags = nullptr; }
}

ncclResult_t dispatch(std::vector<Op> &ops)
{
    if (ops.empty()) return ncclSuccess;
    if (async_mode() && ops[0].comm->helper.joinable()) return enqueue_async(ops);
    if (ops[0].comm->h->error.load()) return ncclSystemError;
    return run_group(ops);
}

ncclResult_t submit(const Op &o)
{
    g_ops.push_back(o);
    if (g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return dispatch(ops);
}

ncclResult_t attach(const std::string &name, int nranks, int rank, Comm **out)
{
    if (nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->name = name; c->nranks = nranks; c->rank = rank; c->bytes = seg_bytes(nranks);
    int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    const bool creator = fd >= 0;
    if (creator) {
        if (ftruncate(fd, (off_t)c->bytes) != 0) { close(fd); shm_unlink(name.c_str()); delete c; return ncclSystemError; }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            fd = shm_open(name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c->bytes) break;
            if (fd >= 0) close(fd);
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) { delete c; return ncclSystemError; }
            usleep(200);
        }
    }
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->h = static_cast<Header *>(p);
    if (creator) {
        c->h->nranks.store(nranks);                                 // (a fresh segment is all zero)
        c->h->magic.store(MAGIC, std::memory_order_release);
    } else {
        Deadline d;
        while (c->h->magic.load(std::memory_order_acquire) != MAGIC)
            if (d.expired(c)) { complain(c, "attach: the segment was never initialised"); munmap(p, c->bytes); delete c; return ncclSystemError; }
        if (c->h->nranks.load() != nranks) { complain(c, "attach: the ranks disagree about the communicator size"); munmap(p, c->bytes); delete c; return ncclInvalidArgument; }
    }
    c->h->attached.fetch_add(1);
    {   // nobody runs ahead of a rank that has not attached yet (it would miss a barrier generation)
        Deadline d;
        while (c->h->attached.load(std::memory_order_acquire) < nranks)
            if (d.expired(c)) { complain(c, "attach: not every rank joined the communicator"); return ncclSystemError; }
    }
    *out = c;
    return ncclSuccess;
}

std::atomic<unsigned> g_id_counter{0};

}  // namespace

extern "C" {

__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    const unsigned long long t = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof id->internal, "/bpmf_rccl_double_%d_%u_%llx", (int)getpid(), g_id_counter.fetch_add(1), t);
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    if (strncmp(id.internal, "/bpmf_rccl_double_", 18) != 0) {
        fprintf(stderr, "[rccl_double] ncclCommInitRank: this unique id was not made by the test double\n");
        return ncclInvalidArgument;
    }
    Comm *c = nullptr;
    ncclResult_t r = attach(id.internal, nranks, rank, &c);
    if (r != ncclSuccess) return r;
    if ((r = start_helper(c)) != ncclSuccess) return r;
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

// every rank with the same colour (what capi.hip asks for: a duplicate of the communicator); key = the new rank
__attribute__((visibility("default"))) ncclResult_t ncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t *newcomm, ncclConfig_t *)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !newcomm) return ncclInvalidArgument;
    if (color != 0 || key != c->rank) { complain(c, "ncclCommSplit: the test double only duplicates a communicator (colour 0, key = rank)"); return ncclInvalidUsage; }
    drain(c);                                                       // (the barriers below are this thread's: the helper must be idle)
    ncclResult_t r = barrier(c);
    if (r != ncclSuccess) return r;
    const int gen = c->h->nsplit.load();
    r = barrier(c);                                                 // (everybody has read the generation)
    if (r != ncclSuccess) return r;
    if (c->rank == 0) c->h->nsplit.fetch_add(1);
    Comm *n = nullptr;
    r = attach(c->name + "_s" + std::to_string(gen), c->nranks, c->rank, &n);
    if (r != ncclSuccess) return r;
    if ((r = start_helper(n)) != ncclSuccess) return r;
    *newcomm = reinterpret_cast<ncclComm_t>(n);
    return barrier(c);
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    stop_helper(c);
    const bool last = c->h->detached.fetch_add(1) + 1 == c->nranks;
    munmap(c->h, c->bytes);
    if (last) shm_unlink(c->name.c_str());
    delete c;
    return ncclSuccess;
}

// ncclCommAbort: everything in flight on the communicator fails, on every rank (the sticky error word lives in the shared
// segment: a peer waiting for this rank gives up at once instead of after its time-out), the held streams are released.
__attribute__((visibility("default"))) ncclResult_t ncclCommAbort(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    c->h->error.store(1);
    return ncclCommDestroy(comm);
}

__attribute__((visibility("default"))) ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t *asyncError)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !asyncError) return ncclInvalidArgument;
    *asyncError = c->h->error.load() ? ncclSystemError : ncclSuccess;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    if (!comm || !count) return ncclInvalidArgument;
    *count = reinterpret_cast<Comm *>(comm)->nranks;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank)
{
    if (!comm || !rank) return ncclInvalidArgument;
    *rank = reinterpret_cast<Comm *>(comm)->rank;
    return ncclSuccess;
}

__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "rccl_double: HIP error (see stderr)";
    case ncclSystemError: return "rccl_double: a rank waited for its peers in vain (see stderr)";
    case ncclInvalidArgument: return "rccl_double: invalid argument (see stderr)";
    case ncclInvalidUsage: return "rccl_double: invalid usage (see stderr)";
    default: return "rccl_double: error";
    }
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return dispatch(ops);
}

__attribute__((visibility("default"))) ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !type_bytes(datatype)) return ncclInvalidArgument;
    return submit(Op{Op::SEND, c, sendbuff, nullptr, count * type_bytes(datatype), peer, datatype, stream});
}

__attribute__((visibility("default"))) ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !type_bytes(datatype)) return ncclInvalidArgument;
    return submit(Op{Op::RECV, c, nullptr, recvbuff, count * type_bytes(datatype), peer, datatype, stream});
}

__attribute__((visibility("default"))) ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                                                                 hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !type_bytes(datatype) || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    const size_t bytes = count * type_bytes(datatype);
    ncclGroupStart();
    ncclResult_t r = ncclSuccess;
    if (c->rank == root) {
        r = submit(Op{Op::COPY, c, sendbuff, recvbuff, bytes, root, datatype, stream});
        for (int p = 0; p < c->nranks && r == ncclSuccess; ++p)
            if (p != root) r = submit(Op{Op::SEND, c, sendbuff, nullptr, bytes, p, datatype, stream});
    } else {
        r = submit(Op{Op::RECV, c, nullptr, recvbuff, bytes, root, datatype, stream});
    }
    const ncclResult_t e = ncclGroupEnd();
    return r != ncclSuccess ? r : e;
}

__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                                                                 hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !type_bytes(datatype)) return ncclInvalidArgument;
    const size_t bytes = sendcount * type_bytes(datatype);
    ncclGroupStart();
    ncclResult_t r = submit(Op{Op::COPY, c, sendbuff, static_cast<char *>(recvbuff) + (size_t)c->rank * bytes, bytes, c->rank, datatype, stream});
    for (int p = 0; p < c->nranks && r == ncclSuccess; ++p) {
        if (p == c->rank) continue;
        r = submit(Op{Op::SEND, c, sendbuff, nullptr, bytes, p, datatype, stream});
        if (r == ncclSuccess) r = submit(Op{Op::RECV, c, nullptr, static_cast<char *>(recvbuff) + (size_t)p * bytes, bytes, p, datatype, stream});
    }
    const ncclResult_t e = ncclGroupEnd();
    return r != ncclSuccess ? r : e;
}

__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                                                                 hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !type_bytes(datatype)) return ncclInvalidArgument;
    if (op != ncclSum) { complain(c, "ncclAllReduce: the test double only sums"); return ncclInvalidArgument; }
    return submit(Op{Op::ALLREDUCE, c, sendbuff, recvbuff, count * type_bytes(datatype), -1, datatype, stream});
}

__attribute__((visibility("default"))) ncclResult_t ncclReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root,
                                                              ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !type_bytes(datatype) || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    if (op != ncclSum) { complain(c, "ncclReduce: the test double only sums"); return ncclInvalidArgument; }
    return submit(Op{Op::REDUCE, c, sendbuff, recvbuff, count * type_bytes(datatype), root, datatype, stream});
}

}  // extern "C"
