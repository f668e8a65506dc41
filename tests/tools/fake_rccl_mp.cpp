// fake_rccl_mp.cpp -- a MULTI-PROCESS stand-in for librccl (test infrastructure; never shipped, never linked by the product).
//
// The driver measures scaling with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`: one PROCESS per rank, each with its
// own vp_context (num_devices 1, world_size N, first_rank r) whose RCCL rank comes from ncclCommInitRank on a shared ncclUniqueId.  The GPU boxes
// the tests run on have ONE GPU and the real library refuses two ranks on one device, so that command line could never run to its JSON line
// there.  This library lets it: ranks in DIFFERENT processes (or threads) that share a device, the bytes staged through a file-backed shared
// segment named by the unique id (device -> segment on the sender, segment -> device on the receiver).  tests/test_gpu_bench_cli.py points
// libvpfx at it with VPFX_RCCL_LIBRARY (csrc/multi.cpp: the one place that names the RCCL library).
//
// It is the plumbing twin of tests/tools/fake_rccl.cpp, not a second checker: that file enforces RCCL's matching rules with stream-ordered copies
// inside one process and is what the fan-out's schedule is verified against.  Here every call is host-synchronous -- the stream is drained, the
// payload staged, the call returns once the peer has taken it -- but the matching is still RCCL's: a rank is inside ONE operation at a time,
// operations run in issue order, a send pairs with the first unmatched receive of the peer's current operation (FIFO per ordered pair), byte
// counts must agree, all-gathers pair only with all-gathers of the same size and must be in place.  A rank that waits longer than
// FAKE_RCCL_TIMEOUT_MS (default 120000) returns ncclSystemError instead of hanging; a mismatch returns ncclInvalidUsage on every rank.
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define FAKE_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

constexpr int MAX_RANKS = 8, MAX_XFERS = 96;
constexpr uint64_t OUTBOX = 192ull << 20;        // per rank: everything a rank sends in ONE operation (sparse file: pages exist once touched)
constexpr uint32_t MAGIC = 0x6d70fa4eu;

struct ShXfer { int send, peer, state /* 0 posted, 1 matched, 2 taken */, partner; uint64_t bytes, off /* send: in the sender's outbox; recv: in the peer's */; };
struct ShRank {
    int joined, in_op, kind /* 1 p2p group, 2 all-gather */, nx;
    uint64_t ops;
    ShXfer x[MAX_XFERS];
};
struct ShHdr {
    uint32_t magic;
    int n;                                       // communicator size (first joiner sets it)
    pthread_mutex_t m;
    pthread_cond_t cv;
    int aborted;
    char error[1024];
    // all-gather number g uses slot g & 1: ranks posted, ranks that have taken every block, block size
    uint64_t ag_gen[2], ag_bytes[2];
    int ag_posted[2], ag_done[2];
    ShRank r[MAX_RANKS];
};
constexpr uint64_t HDR_BYTES = (sizeof(ShHdr) + 4095) & ~4095ull;

struct Comm {
    uint32_t magic = MAGIC;
    ShHdr* h = nullptr;
    char* base = nullptr;
    int rank = 0, dev = 0;
    uint64_t ag_count = 0;                       // all-gathers this rank has finished
    std::string path;
};

std::atomic<unsigned> g_id_counter{1};
thread_local std::string tl_detail;
thread_local int tl_group_depth = 0;
struct Pending { Comm* c; bool send; int peer; void* ptr; uint64_t bytes; hipStream_t stream; };
thread_local std::vector<Pending> tl_pending;

size_t dt_size(ncclDataType_t dt)
{
    switch (dt) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

long env_long(const char* name, long dflt) { const char* e = getenv(name); return e && *e ? strtol(e, nullptr, 10) : dflt; }

ncclResult_t fail(ncclResult_t code, const std::string& why)
{
    tl_detail = why;
    fprintf(stderr, "[fake_rccl_mp pid %d] %s\n", (int)getpid(), why.c_str());
    return code;
}

char* outbox(Comm* c, int rank) { return c->base + HDR_BYTES + (uint64_t)rank * OUTBOX; }

struct Lock {
    ShHdr* h;
    explicit Lock(ShHdr* hh) : h(hh)
    {
        const int e = pthread_mutex_lock(&h->m);
        if (e == EOWNERDEAD) {                    // a rank died inside the segment: the communicator is gone for everybody
            pthread_mutex_consistent(&h->m);
            h->aborted = 1;
        }
    }
    ~Lock() { pthread_mutex_unlock(&h->m); }
    // false = timed out
    bool wait_until(const timespec& deadline)
    {
        const int e = pthread_cond_timedwait(&h->cv, &h->m, &deadline);
        if (e == EOWNERDEAD) { pthread_mutex_consistent(&h->m); h->aborted = 1; }
        return e != ETIMEDOUT;
    }
    void wake() { pthread_cond_broadcast(&h->cv); }
};

// re-acquire the segment mutex after a copy made outside it (same owner-death handling as Lock)
void relock(ShHdr* h)
{
    if (pthread_mutex_lock(&h->m) == EOWNERDEAD) { pthread_mutex_consistent(&h->m); h->aborted = 1; }
}

timespec deadline_in_ms(long ms)
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    t.tv_sec += ms / 1000; t.tv_nsec += (ms % 1000) * 1000000L;
    if (t.tv_nsec >= 1000000000L) { t.tv_sec += 1; t.tv_nsec -= 1000000000L; }
    return t;
}

std::string describe(const ShHdr* h, int r)
{
    const ShRank& k = h->r[r];
    if (!k.joined) return "not joined";
    if (!k.in_op) return "idle (" + std::to_string(k.ops) + " operations done)";
    if (k.kind == 2) return "all-gather";
    std::string s = "p2p group [";
    for (int i = 0; i < k.nx; ++i)
        s += std::string(k.x[i].send ? " send->" : " recv<-") + std::to_string(k.x[i].peer) + ":" + std::to_string(k.x[i].bytes) + (k.x[i].state ? "*" : "");
    return s + " ]";
}

// pair sends with receives (segment mutex held): FIFO per ordered pair, inside the two CURRENT operations only
void match(ShHdr* h)
{
    if (h->error[0] || h->aborted) return;
    for (int a = 0; a < h->n; ++a) {
        ShRank& A = h->r[a];
        if (!A.in_op || A.kind != 1) continue;
        bool blocked[MAX_RANKS] = {};
        for (int i = 0; i < A.nx; ++i) {
            ShXfer& s = A.x[i];
            if (!s.send || s.state || blocked[s.peer]) { if (s.send && !s.state) blocked[s.peer] = true; continue; }
            ShRank& B = h->r[s.peer];
            int j = -1;
            if (B.in_op && B.kind == 1)
                for (int q = 0; q < B.nx; ++q) if (!B.x[q].send && !B.x[q].state && B.x[q].peer == a) { j = q; break; }
            if (j < 0) { blocked[s.peer] = true; continue; }
            if (B.x[j].bytes != s.bytes) {
                snprintf(h->error, sizeof h->error, "size mismatch: rank %d sends %llu B to rank %d, whose matching receive (issue order) takes %llu B",
                         a, (unsigned long long)s.bytes, s.peer, (unsigned long long)B.x[j].bytes);
                return;
            }
            s.state = 1; s.partner = j;
            B.x[j].state = 1; B.x[j].partner = i; B.x[j].off = s.off;
        }
    }
}

Comm* as_comm(ncclComm_t hnd)
{
    Comm* c = reinterpret_cast<Comm*>(hnd);
    return (c && c->magic == MAGIC) ? c : nullptr;
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

ncclResult_t leave_with(Comm* c, Lock& lk, ncclResult_t res)
{
    c->h->r[c->rank].in_op = 0;
    c->h->r[c->rank].nx = 0;
    lk.wake();
    return res;
}

ncclResult_t status_of(Comm* c, bool timed_out, long timeout_ms)
{
    ShHdr* h = c->h;
    if (h->error[0] && !timed_out) return fail(ncclInvalidUsage, h->error);
    if (h->aborted) return fail(ncclInternalError, "communicator aborted while rank " + std::to_string(c->rank) + " waited in " + describe(h, c->rank));
    if (timed_out) {
        snprintf(h->error, sizeof h->error, "rank %d timed out after %ld ms: the communicator is unusable from here on", c->rank, timeout_ms);
        std::string s = "time-out: rank " + std::to_string(c->rank) + " waited " + std::to_string(timeout_ms) + " ms in " + describe(h, c->rank) + "; peers:";
        for (int p = 0; p < h->n; ++p) if (p != c->rank) s += "\n    rank " + std::to_string(p) + ": " + describe(h, p);
        return fail(ncclSystemError, s);
    }
    return ncclSuccess;
}

ncclResult_t run_p2p(Comm* c, const std::vector<Pending>& pend)
{
    ShHdr* h = c->h;
    if ((int)pend.size() > MAX_XFERS) return fail(ncclInternalError, "stand-in: more than " + std::to_string(MAX_XFERS) + " transfers in one group");
    DeviceGuard dg(c->dev);
    // the payload leaves the device now: everything queued in front of the operation has to be over
    if (hipStreamSynchronize(pend[0].stream) != hipSuccess) return fail(ncclUnhandledCudaError, "hipStreamSynchronize before a p2p group");
    std::vector<ShXfer> xs(pend.size());
    uint64_t off = 0;
    for (size_t i = 0; i < pend.size(); ++i) {
        xs[i] = ShXfer{pend[i].send ? 1 : 0, pend[i].peer, 0, -1, pend[i].bytes, 0};
        if (!pend[i].send) continue;
        if (off + pend[i].bytes > OUTBOX) return fail(ncclInternalError, "stand-in: a rank sends more than its outbox holds in one group");
        if (pend[i].bytes && hipMemcpy(outbox(c, c->rank) + off, pend[i].ptr, pend[i].bytes, hipMemcpyDeviceToHost) != hipSuccess)
            return fail(ncclUnhandledCudaError, "staging a send");
        xs[i].off = off;
        off += (pend[i].bytes + 255) & ~255ull;
    }
    const long timeout_ms = env_long("FAKE_RCCL_TIMEOUT_MS", 120000);
    const timespec deadline = deadline_in_ms(timeout_ms);
    Lock lk(h);
    if (h->error[0] || h->aborted) return status_of(c, false, timeout_ms);
    ShRank& R = h->r[c->rank];
    if (R.in_op) return fail(ncclInvalidUsage, "rank " + std::to_string(c->rank) + " issued an operation while inside another one");
    R.kind = 1; R.nx = (int)xs.size();
    for (size_t i = 0; i < xs.size(); ++i) R.x[i] = xs[i];
    R.in_op = 1; ++R.ops;
    match(h);
    lk.wake();
    for (;;) {
        if (h->error[0] || h->aborted) return leave_with(c, lk, status_of(c, false, timeout_ms));
        // take what has been matched to my receives (outside the lock: the sender's slot cannot change before I mark it taken)
        bool took = false;
        for (int i = 0; i < R.nx; ++i) {
            if (R.x[i].send || R.x[i].state != 1) continue;
            const ShXfer y = R.x[i];
            pthread_mutex_unlock(&h->m);
            const hipError_t e = y.bytes ? hipMemcpy(pend[i].ptr, outbox(c, y.peer) + y.off, y.bytes, hipMemcpyHostToDevice) : hipSuccess;
            relock(h);
            if (e != hipSuccess) { snprintf(h->error, sizeof h->error, "rank %d: HIP error while taking a message: %s", c->rank, hipGetErrorString(e)); break; }
            R.x[i].state = 2;
            h->r[y.peer].x[y.partner].state = 2;
            took = true;
        }
        if (took) { lk.wake(); continue; }
        bool complete = true;
        for (int i = 0; i < R.nx; ++i) complete = complete && R.x[i].state == 2;
        if (complete) return leave_with(c, lk, ncclSuccess);
        if (!lk.wait_until(deadline)) return leave_with(c, lk, status_of(c, true, timeout_ms));
        match(h);
    }
}

ncclResult_t run_allgather(Comm* c, const void* sendbuff, void* recvbuff, uint64_t blk, hipStream_t stream)
{
    ShHdr* h = c->h;
    if (blk > OUTBOX) return fail(ncclInternalError, "stand-in: all-gather block larger than the outbox");
    DeviceGuard dg(c->dev);
    if (hipStreamSynchronize(stream) != hipSuccess) return fail(ncclUnhandledCudaError, "hipStreamSynchronize before an all-gather");
    if (blk && hipMemcpy(outbox(c, c->rank), sendbuff, blk, hipMemcpyDeviceToHost) != hipSuccess) return fail(ncclUnhandledCudaError, "staging an all-gather block");
    const long timeout_ms = env_long("FAKE_RCCL_TIMEOUT_MS", 120000);
    const timespec deadline = deadline_in_ms(timeout_ms);
    const uint64_t g = c->ag_count;
    const int slot = (int)(g & 1);
    Lock lk(h);
    if (h->error[0] || h->aborted) return status_of(c, false, timeout_ms);
    ShRank& R = h->r[c->rank];
    if (R.in_op) return fail(ncclInvalidUsage, "rank " + std::to_string(c->rank) + " issued an operation while inside another one");
    R.kind = 2; R.nx = 0; R.in_op = 1; ++R.ops;
    if (h->ag_gen[slot] != g + 1) { h->ag_gen[slot] = g + 1; h->ag_posted[slot] = h->ag_done[slot] = 0; h->ag_bytes[slot] = blk; }   // first rank of all-gather g (everyone left g - 2 long ago)
    if (h->ag_bytes[slot] != blk)
        snprintf(h->error, sizeof h->error, "all-gather mismatch: rank %d gathers blocks of %llu B, an earlier rank %llu B", c->rank, (unsigned long long)blk,
                 (unsigned long long)h->ag_bytes[slot]);
    ++h->ag_posted[slot];
    lk.wake();
    while (h->ag_posted[slot] < h->n && !h->error[0] && !h->aborted)
        if (!lk.wait_until(deadline)) return leave_with(c, lk, status_of(c, true, timeout_ms));
    if (h->error[0] || h->aborted) return leave_with(c, lk, status_of(c, false, timeout_ms));
    pthread_mutex_unlock(&h->m);
    hipError_t e = hipSuccess;
    for (int p = 0; p < h->n && e == hipSuccess; ++p)
        if (p != c->rank && blk) e = hipMemcpy((char*)recvbuff + (uint64_t)p * blk, outbox(c, p), blk, hipMemcpyHostToDevice);
    relock(h);
    if (e != hipSuccess) snprintf(h->error, sizeof h->error, "rank %d: HIP error while taking all-gather blocks: %s", c->rank, hipGetErrorString(e));
    ++h->ag_done[slot];
    lk.wake();
    // nobody may overwrite its outbox (its next operation) before every rank has taken every block
    while (h->ag_done[slot] < h->n && !h->error[0] && !h->aborted)
        if (!lk.wait_until(deadline)) return leave_with(c, lk, status_of(c, true, timeout_ms));
    ++c->ag_count;
    return leave_with(c, lk, status_of(c, false, timeout_ms));
}

}  // namespace

FAKE_EXPORT int fake_rccl_mp_marker(void) { return 1; }        // (the test asserts that THIS library is the one libvpfx resolved)

FAKE_EXPORT const char* ncclGetErrorString(ncclResult_t r)
{
    static thread_local std::string s;
    static const char* names[] = {"no error", "unhandled cuda error", "unhandled system error", "internal error", "invalid argument", "invalid usage",
                                  "remote error", "in progress"};
    s = std::string("fake_rccl_mp: ") + ((int)r >= 0 && (int)r < 8 ? names[(int)r] : "?") + (r != ncclSuccess && !tl_detail.empty() ? " -- " + tl_detail : "");
    return s.c_str();
}

// The id IS the path of the shared segment; the caller (rank 0 of the job) creates and initialises it here, before any other rank has the id.
FAKE_EXPORT ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return fail(ncclInvalidArgument, "null id");
    memset(id, 0, sizeof *id);
    struct stat sb;
    const char* dir = (stat("/dev/shm", &sb) == 0 && S_ISDIR(sb.st_mode)) ? "/dev/shm" : "/tmp";
    snprintf(id->internal, sizeof id->internal, "%s/fake-rccl-mp-%d-%u", dir, (int)getpid(), g_id_counter++);
    const int fd = open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return fail(ncclSystemError, std::string("cannot create ") + id->internal + ": " + strerror(errno));
    const uint64_t total = HDR_BYTES + (uint64_t)MAX_RANKS * OUTBOX;
    if (ftruncate(fd, (off_t)total) != 0) { close(fd); unlink(id->internal); return fail(ncclSystemError, "ftruncate of the shared segment"); }
    void* p = mmap(nullptr, HDR_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { unlink(id->internal); return fail(ncclSystemError, "mmap of the shared segment"); }
    ShHdr* h = static_cast<ShHdr*>(p);
    memset(h, 0, sizeof *h);
    pthread_mutexattr_t ma; pthread_mutexattr_init(&ma);
    pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
    pthread_mutexattr_setrobust(&ma, PTHREAD_MUTEX_ROBUST);
    pthread_mutex_init(&h->m, &ma);
    pthread_condattr_t ca; pthread_condattr_init(&ca);
    pthread_condattr_setpshared(&ca, PTHREAD_PROCESS_SHARED);
    pthread_condattr_setclock(&ca, CLOCK_MONOTONIC);
    pthread_cond_init(&h->cv, &ca);
    h->magic = MAGIC;
    munmap(p, HDR_BYTES);
    return ncclSuccess;
}

FAKE_EXPORT ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return fail(ncclInvalidArgument, "ncclCommInitRank arguments (the stand-in holds up to 8 ranks)");
    char path[sizeof id.internal + 1];
    memcpy(path, id.internal, sizeof id.internal); path[sizeof id.internal] = 0;
    const int fd = open(path, O_RDWR);
    if (fd < 0) return fail(ncclSystemError, std::string("unique id names no shared segment (") + path + "): was it made by this library's ncclGetUniqueId?");
    const uint64_t total = HDR_BYTES + (uint64_t)MAX_RANKS * OUTBOX;
    void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_NORESERVE, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail(ncclSystemError, "mmap of the shared segment");
    ShHdr* h = static_cast<ShHdr*>(p);
    if (h->magic != MAGIC) { munmap(p, total); return fail(ncclSystemError, "shared segment not initialised"); }
    Comm* c = new Comm();
    c->h = h; c->base = static_cast<char*>(p); c->rank = rank; c->path = path;
    if (hipGetDevice(&c->dev) != hipSuccess) { delete c; munmap(p, total); return fail(ncclUnhandledCudaError, "hipGetDevice"); }
    int bad = 0;
    {
        Lock lk(h);
        if (h->n == 0) h->n = nranks;
        if (h->n != nranks) bad = 1;
        else if (h->r[rank].joined) bad = 2;
        else { h->r[rank].joined = 1; lk.wake(); }
    }
    if (bad) {                                   // (unmapped only after the lock on the segment has been released)
        munmap(p, total); delete c;
        return bad == 1 ? fail(ncclInvalidArgument, "ranks of one unique id disagree on the communicator size")
                        : fail(ncclInvalidUsage, "rank " + std::to_string(rank) + " joined twice");
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;                          // (non-blocking: the first operation waits for the ranks that have not joined yet)
}

FAKE_EXPORT ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist)
{
    if (!comms || ndev < 1 || ndev > MAX_RANKS) return fail(ncclInvalidArgument, "ncclCommInitAll arguments");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int i = 0; i < ndev && r == ncclSuccess; ++i) {
        (void)hipSetDevice(devlist ? devlist[i] : i);
        r = ncclCommInitRank(&comms[i], ndev, id, i);
    }
    (void)hipSetDevice(prev);
    return r;
}

FAKE_EXPORT ncclResult_t ncclCommCount(const ncclComm_t comm, int* count)
{
    Comm* c = as_comm(comm);
    if (!c || !count) return fail(ncclInvalidArgument, "ncclCommCount arguments");
    *count = c->h->n;
    return ncclSuccess;
}

static ncclResult_t retire(ncclComm_t comm, bool abort)
{
    Comm* c = as_comm(comm);
    if (!c) return fail(ncclInvalidArgument, "not a communicator");
    bool last = true;
    {
        Lock lk(c->h);
        if (abort) c->h->aborted = 1;
        c->h->r[c->rank].joined = 0;
        for (int p = 0; p < c->h->n; ++p) last = last && !c->h->r[p].joined;
        lk.wake();
    }
    if (last) unlink(c->path.c_str());           // the segment disappears with its last rank
    c->magic = 0xdeadc0deu;
    munmap(c->base, HDR_BYTES + (uint64_t)MAX_RANKS * OUTBOX);
    delete c;
    return ncclSuccess;
}
FAKE_EXPORT ncclResult_t ncclCommDestroy(ncclComm_t comm) { return retire(comm, false); }
FAKE_EXPORT ncclResult_t ncclCommAbort(ncclComm_t comm) { return retire(comm, true); }

FAKE_EXPORT ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* res)
{
    Comm* c = as_comm(comm);
    if (!c || !res) return fail(ncclInvalidArgument, "ncclCommGetAsyncError arguments");
    *res = c->h->aborted ? ncclInternalError : ncclSuccess;
    return ncclSuccess;
}

FAKE_EXPORT ncclResult_t ncclGroupStart() { ++tl_group_depth; return ncclSuccess; }

FAKE_EXPORT ncclResult_t ncclGroupEnd()
{
    if (tl_group_depth > 0 && --tl_group_depth > 0) return ncclSuccess;
    std::vector<Pending> pend;
    pend.swap(tl_pending);
    if (pend.empty()) return ncclSuccess;        // (the ncclCommInitRank group of multi_create lands here)
    for (const Pending& p : pend)
        if (p.c != pend[0].c || p.stream != pend[0].stream) return fail(ncclInvalidUsage, "stand-in: a group holds the transfers of ONE communicator on ONE stream");
    return run_p2p(pend[0].c, pend);
}

static ncclResult_t queue_p2p(bool send, void* buff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = as_comm(comm);
    if (!c) return fail(ncclInvalidArgument, "not a communicator (freed by ncclCommAbort / ncclCommDestroy?)");
    if (peer < 0 || peer >= c->h->n || peer == c->rank) return fail(ncclInvalidArgument, "peer " + std::to_string(peer));
    if (!dt_size(dt)) return fail(ncclInvalidArgument, "datatype");
    if (count && !buff) return fail(ncclInvalidArgument, "null buffer");
    tl_pending.push_back(Pending{c, send, peer, buff, (uint64_t)count * dt_size(dt), stream});
    if (tl_group_depth == 0) return ncclGroupEnd();
    return ncclSuccess;
}

FAKE_EXPORT ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    return queue_p2p(true, const_cast<void*>(sendbuff), count, dt, peer, comm, stream);
}

FAKE_EXPORT ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    return queue_p2p(false, recvbuff, count, dt, peer, comm, stream);
}

FAKE_EXPORT ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = as_comm(comm);
    if (!c) return fail(ncclInvalidArgument, "not a communicator (freed by ncclCommAbort / ncclCommDestroy?)");
    if (!dt_size(dt) || (count && (!sendbuff || !recvbuff))) return fail(ncclInvalidArgument, "ncclAllGather arguments");
    if (tl_group_depth > 0) return fail(ncclInvalidUsage, "stand-in: all-gather inside a group is not used by libvpfx");
    const uint64_t blk = (uint64_t)count * dt_size(dt);
    if (sendbuff != (const char*)recvbuff + (uint64_t)c->rank * blk)
        return fail(ncclInvalidUsage, "all-gather: libvpfx gathers in place -- sendbuff must be recvbuff + rank * count");
    return run_allgather(c, sendbuff, recvbuff, blk, stream);
}
