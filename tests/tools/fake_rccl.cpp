// fake_rccl.cpp -- a CHECKING stand-in for librccl.so.1 (test infrastructure; never shipped, never linked by the product).
//
// csrc/multi.cpp resolves RCCL at run time (dlopen of "librccl.so.1").  The GPU boxes the tests run on have ONE GPU, so the RCCL branch of the
// fan-out (grouped ncclSend / ncclRecv, in-place ncclAllGather, ncclCommInitAll / ncclCommInitRank, ncclCommAbort) can never run there with
// more than one rank on the real library.  This file builds a librccl.so.1 whose ranks may SHARE a device and which moves the bytes with
// stream-ordered device copies -- with RCCL's matching rules enforced, not the looser (src, dst, seq) mail boxes of VP_MULTI_PEER_COPY:
//
//   * operations of one communicator execute strictly in issue order (they are kernels on the rank's stream): a rank's operation i + 1 cannot
//     match anything before its operation i has completed;
//   * a send matches the FIRST unmatched receive of the peer's CURRENT operation that names this rank (FIFO per ordered pair), and only if the
//     peer's current operation is a point-to-point group; the byte counts must be equal (RCCL: undefined behaviour / hang) -> ncclInvalidUsage;
//   * an all-gather matches only all-gathers: every rank's current operation must be one, with equal count and datatype; sendbuff inside
//     recvbuff must sit exactly at recvbuff + rank * count (the in-place rule) -> ncclInvalidUsage otherwise;
//   * if every rank of the communicator is inside an operation and nothing can be matched, the issue orders of the ranks differ: on the real
//     library that is a hang; here every rank returns ncclInvalidUsage with the operations spelled out;
//   * a rank that waits longer than FAKE_RCCL_TIMEOUT_MS (default 20000) for its peers returns ncclSystemError ("time-out") instead of hanging.
//
// Host-side semantics are STRICTER than RCCL's: a call returns only once its transfers have been matched (RCCL returns at once and matches on
// the device).  Whatever completes under rendezvous matching completes under RCCL's asynchronous matching; the converse is not true, which is
// the point of a checker.  Data: at the match both ranks are inside their calls, each having recorded an "entry" event on its stream; the
// receiver's stream waits for the sender's entry event, copies, records a "done" event; the sender's stream waits for that before anything
// issued after the operation runs (its send buffer may be overwritten then).
//
// Fault injection (environment, read at communicator creation):
//   FAKE_RCCL_ASYNC_ERROR_RANK=r FAKE_RCCL_ASYNC_ERROR_AFTER_OPS=n [FAKE_RCCL_STALL_MS=ms]
//       after rank r has issued n operations its stream is stalled for ms (default 1500) milliseconds by a host function and
//       ncclCommGetAsyncError(rank r) reports ncclSystemError from then on (drives multi.cpp's asynchronous-error branch).
// Introspection: fake_rccl_stats(out[8]) = {communicators created, p2p groups, sends, recvs, all-gathers, bytes moved, errors reported, aborts}.
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define FAKE_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

struct Xfer { bool send; int peer; void* ptr; size_t bytes; bool matched; hipEvent_t done; };

struct Op {
    enum Kind { NONE, P2P, ALLGATHER } kind = NONE;
    std::vector<Xfer> x;                         // P2P
    const void* sendbuff = nullptr; void* recvbuff = nullptr; size_t count = 0; ncclDataType_t dt = ncclUint8;   // ALLGATHER
    bool executed = false;                       // ALLGATHER: copies enqueued
    hipStream_t stream = nullptr;
    hipEvent_t entry = nullptr;                  // recorded on the rank's stream when the operation was issued
    std::vector<hipEvent_t> wait_after;          // events the rank's stream has to wait for before the operation counts as complete on it
};

struct Clique;
struct Comm {
    uint32_t magic = 0xfa4ecc1u;
    std::shared_ptr<Clique> q;
    int rank = 0, dev = 0;
    bool in_op = false, aborted = false;
    Op cur;
    uint64_t ops_issued = 0;
};

struct Clique {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    std::vector<Comm*> comms;                    // by rank; nullptr until the rank has joined
    std::string error;                           // first usage error: every waiting rank returns ncclInvalidUsage
    bool aborted = false;
    int async_rank = -1; long async_after = -1; int stall_ms = 1500;
};

std::mutex g_m;
std::map<std::string, std::shared_ptr<Clique>> g_by_id;
std::atomic<long long> g_stats[8];
std::atomic<uint64_t> g_id_counter{1};
thread_local std::string tl_detail;
thread_local int tl_group_depth = 0;
struct Pending { Comm* c; Xfer x; hipStream_t stream; };
thread_local std::vector<Pending> tl_pending;

size_t dt_size(ncclDataType_t dt)
{
    switch (dt) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
#if defined(RCCL_BFLOAT16)
    case ncclBfloat16: return 2;
#endif
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

long env_long(const char* name, long dflt) { const char* e = getenv(name); return e && *e ? strtol(e, nullptr, 10) : dflt; }

ncclResult_t fail(ncclResult_t code, const std::string& why)
{
    tl_detail = why;
    g_stats[6]++;
    fprintf(stderr, "[fake_rccl] %s\n", why.c_str());
    return code;
}

std::string describe(const Comm* c)
{
    if (!c) return "not joined";
    if (!c->in_op) return "idle (operation " + std::to_string(c->ops_issued) + " done)";
    const Op& o = c->cur;
    std::string s = "op " + std::to_string(c->ops_issued) + ": ";
    if (o.kind == Op::ALLGATHER) return s + "all-gather of " + std::to_string(o.count) + " x " + std::to_string(dt_size(o.dt)) + " B";
    s += "p2p group [";
    for (const Xfer& x : o.x) s += std::string(x.send ? " send->" : " recv<-") + std::to_string(x.peer) + ":" + std::to_string(x.bytes) + (x.matched ? "*" : "");
    return s + " ]";
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// one device copy src rank -> dst rank, ordered after both ranks' entry events, on the receiver's stream; `done` lands in both ranks' wait lists
bool copy_between(Comm* src, Comm* dst, const void* from, void* to, size_t bytes, std::string& err)
{
    DeviceGuard dg(dst->dev);
    hipEvent_t done = nullptr;
    hipError_t e = hipStreamWaitEvent(dst->cur.stream, src->cur.entry, 0);
    if (e == hipSuccess && bytes && from != to) e = hipMemcpyAsync(to, from, bytes, hipMemcpyDefault, dst->cur.stream);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(done, dst->cur.stream);
    if (e != hipSuccess) { err = std::string("HIP error while moving data: ") + hipGetErrorString(e); return false; }
    src->cur.wait_after.push_back(done);         // the sender may not reuse its buffer before the copy has read it
    g_stats[5] += (long long)bytes;
    return true;
}

bool op_complete(const Comm* c)
{
    if (c->cur.kind == Op::ALLGATHER) return c->cur.executed;
    for (const Xfer& x : c->cur.x) if (!x.matched) return false;
    return true;
}

// Match whatever can be matched (clique mutex held).  Sets q.error on a usage error or a dead end.
void progress(Clique& q)
{
    if (!q.error.empty() || q.aborted) return;
    bool again = true;
    while (again) {
        again = false;
        // ---- point to point: FIFO per ordered pair inside the two CURRENT operations
        for (Comm* a : q.comms) {
            if (!a || !a->in_op || a->cur.kind != Op::P2P) continue;
            std::vector<char> pair_blocked(q.n, 0);          // an earlier send of a to p is still unmatched: later ones to p must wait (FIFO)
            for (Xfer& s : a->cur.x) {
                if (!s.send || s.matched) { continue; }
                if (pair_blocked[s.peer]) continue;
                Comm* b = q.comms[s.peer];
                Xfer* r = nullptr;
                if (b && b->in_op && b->cur.kind == Op::P2P)
                    for (Xfer& y : b->cur.x) if (!y.send && !y.matched && y.peer == a->rank) { r = &y; break; }
                if (!r) { pair_blocked[s.peer] = 1; continue; }
                if (r->bytes != s.bytes) {
                    q.error = "size mismatch: rank " + std::to_string(a->rank) + " sends " + std::to_string(s.bytes) + " B to rank " + std::to_string(b->rank) +
                              ", whose matching receive (issue order) takes " + std::to_string(r->bytes) + " B";
                    return;
                }
                std::string err;
                if (!copy_between(a, b, s.ptr, r->ptr, s.bytes, err)) { q.error = err; return; }
                s.matched = r->matched = true;
                again = true;
            }
        }
        // ---- all-gather: every rank's current operation must be one
        bool all_ag = true, any_ag = false;
        for (Comm* c : q.comms) {
            const bool ag = c && c->in_op && c->cur.kind == Op::ALLGATHER && !c->cur.executed;
            all_ag = all_ag && ag; any_ag = any_ag || ag;
        }
        if (any_ag && all_ag) {
            const Op& o0 = q.comms[0]->cur;
            for (Comm* c : q.comms)
                if (c->cur.count != o0.count || c->cur.dt != o0.dt) {
                    q.error = "all-gather mismatch: rank 0 gathers " + std::to_string(o0.count) + " x " + std::to_string(dt_size(o0.dt)) + " B, rank " +
                              std::to_string(c->rank) + " " + std::to_string(c->cur.count) + " x " + std::to_string(dt_size(c->cur.dt)) + " B";
                    return;
                }
            const size_t blk = o0.count * dt_size(o0.dt);
            for (Comm* d : q.comms)
                for (Comm* s : q.comms) {
                    std::string err;
                    if (!copy_between(s, d, s->cur.sendbuff, (char*)d->cur.recvbuff + (size_t)s->rank * blk, blk, err)) { q.error = err; return; }
                }
            for (Comm* c : q.comms) c->cur.executed = true;
            again = true;
        }
    }
    // ---- dead end?  Every rank is inside an operation and none of them is complete-able: the ranks' issue orders differ.
    bool all_in = true, any_incomplete = false;
    for (Comm* c : q.comms) { all_in = all_in && c && c->in_op; if (c && c->in_op && !op_complete(c)) any_incomplete = true; }
    if (all_in && any_incomplete) {
        bool all_stuck = true;
        for (Comm* c : q.comms) all_stuck = all_stuck && !op_complete(c);
        // (ranks whose operation is complete are about to leave it and may issue the operation the others wait for)
        if (all_stuck) {
            std::string s = "dead end: every rank is inside an operation and nothing matches (issue orders differ between ranks):";
            for (Comm* c : q.comms) s += "\n    rank " + std::to_string(c->rank) + ": " + describe(c);
            q.error = s;
        }
    }
}

// Issue `op` on communicator c and wait (host) until it has been matched.
ncclResult_t run_op(Comm* c, Op&& op)
{
    Clique& q = *c->q;
    hipError_t he;
    {
        DeviceGuard dg(c->dev);
        he = hipEventCreateWithFlags(&op.entry, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventRecord(op.entry, op.stream);
    }
    if (he != hipSuccess) return fail(ncclUnhandledCudaError, std::string("hipEventRecord: ") + hipGetErrorString(he));
    std::unique_lock<std::mutex> lk(q.m);
    if (c->aborted || q.aborted) return fail(ncclInternalError, "communicator aborted");
    if (!q.error.empty()) return fail(ncclInvalidUsage, q.error);
    if (c->in_op) return fail(ncclInvalidUsage, "rank " + std::to_string(c->rank) + " issued an operation from two threads at once");
    c->cur = std::move(op);
    c->in_op = true;
    ++c->ops_issued;
    progress(q);
    q.cv.notify_all();
    const long timeout_ms = env_long("FAKE_RCCL_TIMEOUT_MS", 20000);
    const bool ok = q.cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return op_complete(c) || !q.error.empty() || q.aborted || c->aborted; });
    ncclResult_t res = ncclSuccess;
    if (op_complete(c) && q.error.empty() && !q.aborted && !c->aborted) {
        DeviceGuard dg(c->dev);
        for (hipEvent_t ev : c->cur.wait_after) (void)hipStreamWaitEvent(c->cur.stream, ev, 0);
        // fault injection: stall this rank's stream and start reporting an asynchronous error
        if (q.async_rank == c->rank && q.async_after >= 0 && (long)c->ops_issued == q.async_after) {
            static std::atomic<int> stall_ms_arg;
            stall_ms_arg = q.stall_ms;
            fprintf(stderr, "[fake_rccl] fault injection: rank %d stalls its stream for %d ms after operation %llu; ncclCommGetAsyncError reports an error from now on\n",
                    c->rank, q.stall_ms, (unsigned long long)c->ops_issued);
            (void)hipLaunchHostFunc(c->cur.stream, [](void* p) { std::this_thread::sleep_for(std::chrono::milliseconds(((std::atomic<int>*)p)->load())); }, &stall_ms_arg);
        }
    } else if (!q.error.empty()) res = fail(ncclInvalidUsage, q.error);
    else if (q.aborted || c->aborted) res = fail(ncclInternalError, "communicator aborted while rank " + std::to_string(c->rank) + " waited in " + describe(c));
    else if (!ok) {
        std::string s = "time-out: rank " + std::to_string(c->rank) + " waited " + std::to_string(timeout_ms) + " ms in " + describe(c) + "; peers:";
        for (Comm* p : q.comms) if (p != c) s += "\n    rank " + std::to_string(p ? p->rank : -1) + ": " + describe(p);
        res = fail(ncclSystemError, s);
    }
    // (events are leaked on purpose: a checker for a handful of frames, and destroying an event another stream still waits on is a race)
    c->in_op = false;
    c->cur = Op{};
    progress(q);                                  // this rank leaving may turn the others' state into a detectable dead end -- or not: they wait for its next operation
    q.cv.notify_all();
    return res;
}

Comm* as_comm(ncclComm_t h)
{
    Comm* c = reinterpret_cast<Comm*>(h);
    return (c && c->magic == 0xfa4ecc1u) ? c : nullptr;
}

void read_injection(Clique& q)
{
    q.async_rank = (int)env_long("FAKE_RCCL_ASYNC_ERROR_RANK", -1);
    q.async_after = env_long("FAKE_RCCL_ASYNC_ERROR_AFTER_OPS", -1);
    q.stall_ms = (int)env_long("FAKE_RCCL_STALL_MS", 1500);
    if (q.async_rank >= 0) fprintf(stderr, "[fake_rccl] fault injection armed: rank %d after %ld operations\n", q.async_rank, q.async_after);
}

ncclResult_t queue_p2p(bool send, void* buff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = as_comm(comm);
    if (!c) return fail(ncclInvalidArgument, "not a communicator (freed by ncclCommAbort / ncclCommDestroy?)");
    if (peer < 0 || peer >= c->q->n) return fail(ncclInvalidArgument, "peer " + std::to_string(peer) + " outside the communicator");
    if (peer == c->rank) return fail(ncclInvalidUsage, "send / recv to self is not used by libvpfx");
    if (!dt_size(dt)) return fail(ncclInvalidArgument, "datatype");
    if (count && !buff) return fail(ncclInvalidArgument, "null buffer");
    g_stats[send ? 2 : 3]++;
    tl_pending.push_back(Pending{c, Xfer{send, peer, buff, count * dt_size(dt), false, nullptr}, stream});
    if (tl_group_depth == 0) return ncclGroupEnd();      // an ungrouped call is a group of one
    return ncclSuccess;
}

}  // namespace

FAKE_EXPORT void fake_rccl_stats(long long out[8]) { for (int i = 0; i < 8; ++i) out[i] = g_stats[i].load(); }

// FAKE_RCCL_STATS_FILE=path: the counters as JSON when the process ends (the parent test asserts that the RCCL branch really ran on this library)
__attribute__((destructor)) static void write_stats_file()
{
    const char* path = getenv("FAKE_RCCL_STATS_FILE");
    if (!path || !*path) return;
    if (FILE* f = fopen(path, "w")) {
        fprintf(f, "{\"communicators\": %lld, \"p2p_groups\": %lld, \"sends\": %lld, \"recvs\": %lld, \"all_gathers\": %lld, \"bytes\": %lld, \"errors\": %lld, \"aborts\": %lld}\n",
                g_stats[0].load(), g_stats[1].load(), g_stats[2].load(), g_stats[3].load(), g_stats[4].load(), g_stats[5].load(), g_stats[6].load(), g_stats[7].load());
        fclose(f);
    }
}

FAKE_EXPORT const char* ncclGetErrorString(ncclResult_t r)
{
    static thread_local std::string s;
    static const char* names[] = {"no error", "unhandled cuda error", "unhandled system error", "internal error", "invalid argument", "invalid usage",
                                  "remote error", "in progress"};
    s = std::string("fake_rccl: ") + ((int)r >= 0 && (int)r < 8 ? names[(int)r] : "?") + (r != ncclSuccess && !tl_detail.empty() ? " -- " + tl_detail : "");
    return s.c_str();
}

FAKE_EXPORT ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return fail(ncclInvalidArgument, "null id");
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "fake-rccl-%llu-%d", (unsigned long long)g_id_counter++, (int)getpid());
    return ncclSuccess;
}

FAKE_EXPORT ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(ncclInvalidArgument, "ncclCommInitRank arguments");
    std::shared_ptr<Clique> q;
    {
        std::lock_guard<std::mutex> lk(g_m);
        auto& slot = g_by_id[std::string(id.internal, sizeof id.internal)];
        if (!slot) { slot = std::make_shared<Clique>(); slot->n = nranks; slot->comms.assign(nranks, nullptr); read_injection(*slot); }
        q = slot;
    }
    std::lock_guard<std::mutex> lk(q->m);
    if (q->n != nranks) return fail(ncclInvalidArgument, "ranks of one unique id disagree on the communicator size");
    if (q->comms[rank]) return fail(ncclInvalidUsage, "rank " + std::to_string(rank) + " joined twice");
    Comm* c = new Comm();
    c->q = q; c->rank = rank;
    if (hipGetDevice(&c->dev) != hipSuccess) { delete c; return fail(ncclUnhandledCudaError, "hipGetDevice"); }
    q->comms[rank] = c;
    *comm = reinterpret_cast<ncclComm_t>(c);
    g_stats[0]++;
    q->cv.notify_all();
    return ncclSuccess;                                   // (non-blocking: the first operation waits for the ranks that have not joined yet)
}

FAKE_EXPORT ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist)
{
    if (!comms || ndev < 1) return fail(ncclInvalidArgument, "ncclCommInitAll arguments");
    auto q = std::make_shared<Clique>();
    q->n = ndev; q->comms.assign(ndev, nullptr); read_injection(*q);
    for (int i = 0; i < ndev; ++i) {
        Comm* c = new Comm();
        c->q = q; c->rank = i; c->dev = devlist ? devlist[i] : i;     // (unlike RCCL: ranks may share a device -- the reason this library exists)
        q->comms[i] = c;
        comms[i] = reinterpret_cast<ncclComm_t>(c);
        g_stats[0]++;
    }
    return ncclSuccess;
}

FAKE_EXPORT ncclResult_t ncclCommCount(const ncclComm_t comm, int* count)
{
    Comm* c = as_comm(comm);
    if (!c || !count) return fail(ncclInvalidArgument, "ncclCommCount arguments");
    *count = c->q->n;
    return ncclSuccess;
}

static ncclResult_t retire(ncclComm_t comm, bool abort)
{
    Comm* c = as_comm(comm);
    if (!c) return fail(ncclInvalidArgument, "not a communicator");
    std::shared_ptr<Clique> q = c->q;
    {
        std::lock_guard<std::mutex> lk(q->m);
        c->aborted = true;
        if (abort) { q->aborted = true; g_stats[7]++; }   // an aborted rank takes the communicator down for everybody (its peers would wait for ever)
        // the handle stays allocated with a dead magic word: a use after abort / destroy is REPORTED, not a crash
        if (!c->in_op) c->magic = 0xdeadc0deu;
        q->cv.notify_all();
    }
    return ncclSuccess;
}
FAKE_EXPORT ncclResult_t ncclCommDestroy(ncclComm_t comm) { return retire(comm, false); }
FAKE_EXPORT ncclResult_t ncclCommAbort(ncclComm_t comm) { return retire(comm, true); }

FAKE_EXPORT ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* res)
{
    Comm* c = as_comm(comm);
    if (!c || !res) return fail(ncclInvalidArgument, "ncclCommGetAsyncError arguments");
    Clique& q = *c->q;
    std::lock_guard<std::mutex> lk(q.m);
    *res = (q.async_rank == c->rank && q.async_after >= 0 && (long)c->ops_issued >= q.async_after) ? ncclSystemError : ncclSuccess;
    if (*res != ncclSuccess) tl_detail = "injected asynchronous error on rank " + std::to_string(c->rank);
    return ncclSuccess;
}

FAKE_EXPORT ncclResult_t ncclGroupStart() { ++tl_group_depth; return ncclSuccess; }

FAKE_EXPORT ncclResult_t ncclGroupEnd()
{
    if (tl_group_depth > 0 && --tl_group_depth > 0) return ncclSuccess;
    std::vector<Pending> pend;
    pend.swap(tl_pending);
    if (pend.empty()) return ncclSuccess;                 // (the ncclCommInitRank group of multi_create lands here)
    // libvpfx drives one rank per host thread: a group holds the transfers of ONE communicator on ONE stream
    Comm* c = pend[0].c;
    Op op;
    op.kind = Op::P2P; op.stream = pend[0].stream;
    for (const Pending& p : pend) {
        if (p.c != c || p.stream != op.stream) return fail(ncclInvalidUsage, "a group that mixes communicators or streams is outside what this stand-in models");
        op.x.push_back(p.x);
    }
    g_stats[1]++;
    return run_op(c, std::move(op));
}

FAKE_EXPORT ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    return queue_p2p(true, const_cast<void*>(sendbuff), count, dt, peer, comm, stream);
}

FAKE_EXPORT ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    return queue_p2p(false, recvbuff, count, dt, peer, comm, stream);
}

FAKE_EXPORT ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = as_comm(comm);
    if (!c) return fail(ncclInvalidArgument, "not a communicator (freed by ncclCommAbort / ncclCommDestroy?)");
    if (tl_group_depth > 0) return fail(ncclInvalidUsage, "an all-gather inside a group is outside what this stand-in models");
    const size_t es = dt_size(dt);
    if (!es || (sendcount && (!sendbuff || !recvbuff))) return fail(ncclInvalidArgument, "ncclAllGather arguments");
    const char* s = (const char*)sendbuff; const char* r = (const char*)recvbuff;
    const size_t blk = sendcount * es, total = blk * (size_t)c->q->n;
    if (s + blk > r && s < r + total && s != r + (size_t)c->rank * blk)
        return fail(ncclInvalidUsage, "in-place all-gather: sendbuff of rank " + std::to_string(c->rank) + " overlaps recvbuff but is not recvbuff + rank * count");
    g_stats[4]++;
    Op op;
    op.kind = Op::ALLGATHER; op.stream = stream; op.sendbuff = sendbuff; op.recvbuff = recvbuff; op.count = sendcount; op.dt = dt;
    return run_op(c, std::move(op));
}
