// multi.cpp -- the multi-GPU fan-out BEHIND the C ABI ("device list ... multi-GPU fan-out is internal"; 8(e): z-slabs).
//
// The caller the boundary serves is VolumetricParticleRenderer.OnPostRender (VPR.cs:181-220): one C# main-thread loop that calls
// BinParticlesToMetavoxels / FillMetavoxels / RenderMetavoxels once per frame.  It cannot run a process per GPU, so the fan-out lives
// here: a context created with a device list (vp_config.num_devices / devices[]) -- or as one rank of a multi-process job (world_size,
// first_rank, rccl_unique_id) -- is a FAN-OUT context that owns, per local GPU,
//     one slab context (an ordinary single-device vp_ctx restricted to a contiguous zz range: api.cpp),
//     one host worker thread (kernel launches of different GPUs are issued concurrently: ~50 launches per GPU and frame would otherwise
//     serialise into more host time than the 8-GPU frame has), one HIP stream, one RCCL rank,
// and vp_set_frame / vp_bin / vp_fill / vp_raymarch stay the only calls the host makes.  Every worker runs the same SPMD sequence a
// rank of a multi-process job runs; the two launch styles differ only in how the communicator is created (ncclCommInitAll vs.
// ncclCommInitRank).  The path has exactly two cross-slab dependencies:
//   fill       per-column transmitted light (Fill.shader:224,250), linear in the incoming light: every slab fills with T_in = 1 and publishes
//              its transmittance map tau; ONE all-gather of tau (in place, into the buffer the finish kernel multiplies straight out of).
//   ray-march  (a) inter-metavoxel blending (VPR.cs:652-711): zz-major draw order, so a slab's metavoxels are contiguous in it; each slab
//              composites a premultiplied partial image; the ordered OVER/UNDER blend is per pixel, hence sharded: all-to-all of screen
//              pieces, blend, gather on the display rank (or, VP_MULTI_EXCHANGE_ALL_GATHER, one all-gather of whole images).
//              (b) saturation: the reference's single render target lets a ray stop once it is opaque; a slab alone only knows its own
//              metavoxels and would march everything hidden behind the slabs in front (709 M instead of 321 M samples at 8 slabs of the
//              benchmark scene).  The slabs, front to back, form rm_groups groups; a group marches concurrently, then sends each slab's
//              transmittance map (1 - alpha, 4 B/pixel) to the slabs of the later groups, whose rays stop once the product says nothing
//              visible is left (RmHandoff, raymarch.hip).  Groups, not a fully serial chain: every hop costs a kernel boundary + a
//              point-to-point message (~40 us), which a chain of 8 would pay 7 times.
// RCCL is resolved at run time (dlopen of librccl.so.1: the same copy torch loads when the host is a Python process, /opt/rocm's
// otherwise), so libvpfx.so itself has no link-time dependency on it and single-GPU hosts never touch it.
// VP_MULTI_PEER_COPY (test hook) replaces every RCCL call by device-to-device copies between the local contexts, which lets one GPU
// stand in for N (devices = {0,0,..}): the tests drive the whole fan-out -- threads, slab cut, hand-off, exchange, blend -- on one GPU.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <tuple>
#include <vector>

#include <rccl/rccl.h>          // types and prototypes only: the functions are resolved with dlsym (no link-time dependency)

#include "vpfx_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------------------
// RCCL, resolved at run time
// ---------------------------------------------------------------------------------------------------------------------------------
struct Rccl {
    void* so = nullptr;
    bool tried = false;
    std::string err;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;                 // optional (failure handling): absent -> no abort, time-outs still report
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;

    bool load()
    {
        static std::mutex m;
        std::lock_guard<std::mutex> lk(m);
        if (tried) return so != nullptr;
        tried = true;
        // VPFX_RCCL_LIBRARY=<path>: the RCCL build to use instead of the one the loader finds by name (a site's own build; the tests' multi-process
        // stand-in, which a process that has torch -- and with it the real librccl.so.1 -- mapped can only reach by path).  No fallback when set.
        const char* forced = getenv("VPFX_RCCL_LIBRARY");
        if (forced && *forced) {
            so = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!so) {                                                  // dlerror() clears its state: ONE call per failure
                const char* de = dlerror();
                err = std::string("VPFX_RCCL_LIBRARY=") + forced + " cannot be loaded (" + (de ? de : "?") + ")";
                return false;
            }
        } else {
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            std::string why;
            for (const char* n : names) {
                so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (so) break;
                const char* de = dlerror();                                // once, straight after the failing dlopen
                why = de ? de : "?";
            }
            if (!so) { err = "librccl.so.1 not found (" + why + ")"; return false; }
        }
        bool ok = true;
        auto sym = [&](auto& fn, const char* name) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(so, name));
            if (!fn) { ok = false; err = std::string("librccl: missing symbol ") + name; }
        };
        sym(GetUniqueId, "ncclGetUniqueId"); sym(CommInitRank, "ncclCommInitRank"); sym(CommInitAll, "ncclCommInitAll");
        sym(CommDestroy, "ncclCommDestroy"); sym(CommCount, "ncclCommCount"); sym(AllGather, "ncclAllGather");
        sym(Send, "ncclSend"); sym(Recv, "ncclRecv"); sym(GroupStart, "ncclGroupStart"); sym(GroupEnd, "ncclGroupEnd");
        sym(GetErrorString, "ncclGetErrorString");
        if (ok) {
            CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(so, "ncclCommAbort"));
            CommGetAsyncError = reinterpret_cast<decltype(CommGetAsyncError)>(dlsym(so, "ncclCommGetAsyncError"));
        }
        if (!ok) { dlclose(so); so = nullptr; }
        return ok;
    }
};
Rccl& rccl() { static Rccl r; return r; }

#define VP_NCCL(call)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r_ = (call);                                                                                  \
        if (r_ != ncclSuccess) return vp_fail(c, VP_ERR_RCCL, "%s failed: %s (%s:%d)", #call, rccl().GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------------------
// One local rank
// ---------------------------------------------------------------------------------------------------------------------------------
struct Kid {
    vp_ctx* c = nullptr;              // the slab context
    int rank = 0, device = 0, index = 0;
    hipStream_t stream = nullptr;
    // Exchange stream (round 5): the all-gather of the slab transmittance maps runs HERE, beside the rank's compute stream.  Only the finish pass
    // of a rank > 0 waits for it; rank 0 -- whose fused fill needs nobody's light and whose slab usually holds most of the ray-march -- goes
    // straight from its fill into its march.  One communicator, two streams: the rank's compute stream waits for ev_tau before it issues its
    // next operation on the communicator (comm_fence), so operations of the communicator never run side by side and every rank still issues them
    // in the same order (tau all-gather, then hand-off, then image exchange).
    hipStream_t xstream = nullptr;
    hipEvent_t ev_local = nullptr, ev_tau = nullptr;      // local fill pass done (compute stream) / transmittance maps gathered (exchange stream)
    bool tau_pending = false;                             // ev_tau recorded and not yet waited for by the compute stream
    ncclComm_t comm = nullptr;
    // `comm` is touched by two threads: its owner (this rank's thread: ncclSend / ncclRecv / ncclAllGather / ncclCommGetAsyncError) and whoever
    // aborts the context (multi_abort, any thread: takes the communicator away and ncclCommAbort()s it, which frees it).  The gate's mutex is
    // held only to hand the pointer over -- NEVER across an RCCL call, which can block for ever (lazy connection set-up waiting for a dead peer,
    // an in-process collective waiting for a peer thread that already failed).  The owner marks itself "inside a call" and works on
    // a local copy of the handle; the aborting thread waits up to 200 ms for the call to return (the clean case: calls only enqueue work) and
    // otherwise aborts UNDER the blocked call -- what ncclCommAbort exists for (it makes the blocked call return); the owner drops its copy when
    // it comes back and never destroys it.  Every communicator of an aborted context is therefore aborted, by the aborter or (free_kid) its owner.
    struct CommGate { std::mutex m; std::condition_variable cv; bool in_call = false; };
    std::unique_ptr<CommGate> gate{new CommGate()};
    float* d_tau_all = nullptr;       // [world][LH][LW]: all-gathered slab transmittance maps (own slot written by the local fill pass)
    float* d_img[2] = {nullptr, nullptr};   // partial images of the slab (phase-A composite, phase-B composite), padded to whole pieces
    uint8_t* d_tmaps = nullptr;       // [world][H][W] received hand-off maps of the slabs in front (one byte per pixel, RmHandoff)
    uint8_t* d_tout[2] = {nullptr, nullptr};  // own maps: 1 - alpha(A), (1 - alpha(A)) (1 - alpha(B))
    float* d_pieces = nullptr;        // tiles: [world + 1][piece][4] received pieces; all-gather: [world + 1][pixpad][4] whole images
    float* d_piece_out = nullptr;     // [piece][4] this rank's blended piece
    float* d_final = nullptr;         // display rank: [pixpad][4]
    float* d_xfer = nullptr;          // [world][Nz + 8] rebalance payload
    hipEvent_t ev[3][2] = {};         // stream time of the exchanges: tau all-gather, saturation hand-off, image exchange + blend
    bool ev_valid[3] = {false, false, false};
    int rc = VP_OK;
    bool voted = false;               // the failure this rank returned was agreed on by a vote: every local rank left at the same point
    bool drop_next_send = false;      // VP_MULTI_TEST_DROP_SEND (test hook)
    std::vector<uint64_t> seq_to, seq_from;   // loopback message counters per peer
    std::vector<float> h_xfer;
};

struct Mail { void* dst = nullptr; size_t bytes = 0; bool posted = false, done = false; hipEvent_t ev = nullptr; };

struct P2P { bool send; int peer; void* ptr; size_t bytes; };

}  // namespace

struct vp_multi {
    vp_ctx* parent = nullptr;
    int world = 1, nlocal = 1, first_rank = 0, flags = 0, groups = 1;
    bool use_rccl = false;
    int rccl_ranks = 0;
    std::vector<Kid> kids;
    std::vector<int> cuts;                    // [world + 1]
    bool need_plan = true, have_profile = false, want_profile = false;
    int chain[VP_MAX_RANKS] = {}, group_of[VP_MAX_RANKS] = {};
    size_t npix = 0, piece = 0, pixpad = 0, lm = 0;
    // worker pool: one persistent thread per local rank (none when there is only one)
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    std::function<int(Kid&)> job;
    bool job_exchanges = true;                // the current job contains inter-rank exchanges: an un-voted failure of one rank aborts the context
    uint64_t gen = 0;
    int pending = 0;
    bool quit = false;
    // host barrier of the local ranks (error votes before collectives)
    std::mutex bm;
    std::condition_variable bcv;
    int b_count = 0, b_rc = 0, b_result = 0;
    uint64_t b_gen = 0;
    // loopback mail boxes (VP_MULTI_PEER_COPY)
    std::mutex mm;
    std::condition_variable mcv;
    std::map<std::tuple<int, int, uint64_t>, Mail> box;
    // failure handling (round 4).  The reference logs and carries on (VPR.cs:352,790); a fan-out cannot carry on past a rank that left an
    // exchange, but it must not hang either: whoever notices -- a rank that fails between exchanges, a wait that exceeds the time-out, an
    // asynchronous RCCL error -- ABORTS the context: the flag wakes every host-side wait (mail boxes, votes), ncclCommAbort ends the
    // collectives stuck on the devices, every local rank returns VP_ERR_RCCL, and every later call on the context fails fast until it is
    // destroyed.
    std::atomic<int> aborted{0};
    std::mutex am;
    std::string abort_msg;
    int timeout_ms = 120000;                  // vp_config.exchange_timeout_ms
};

namespace {

Kid* local_kid(vp_multi* M, int rank) { return (rank >= M->first_rank && rank < M->first_rank + M->nlocal) ? &M->kids[rank - M->first_rank] : nullptr; }

// Owner side of Kid::gate: take a local copy of the communicator for ONE RCCL call sequence (nullptr: aborted / taken away) ...
ncclComm_t comm_enter(vp_multi* M, Kid& k)
{
    std::lock_guard<std::mutex> lk(k.gate->m);
    if (!k.comm || M->aborted.load()) return nullptr;
    k.gate->in_call = true;
    return k.comm;
}
// ... and give the gate back when the calls have returned (RAII: every exit path of the caller)
struct CommLeave {
    Kid& k;
    ~CommLeave() { { std::lock_guard<std::mutex> lk(k.gate->m); k.gate->in_call = false; } k.gate->cv.notify_all(); }
};

void multi_abort(vp_multi* M, int rank, const std::string& why)
{
    int expected = 0;
    if (!M->aborted.compare_exchange_strong(expected, 1)) return;           // first reporter wins
    { std::lock_guard<std::mutex> lk(M->am); M->abort_msg = "rank " + std::to_string(rank) + ": " + why; }
    fprintf(stderr, "[libvpfx] fan-out aborted by rank %d: %s\n", rank, why.c_str());
    if (M->use_rccl && rccl().CommAbort)
        for (Kid& k : M->kids) {
            ncclComm_t cm = nullptr;
            {
                std::unique_lock<std::mutex> lk(k.gate->m);
                // the owner is inside an RCCL call: give it 200 ms to come back (calls only enqueue work); if it does not, it is blocked -- on a
                // dead peer, on us -- and the abort below is what makes it return.  Either way the communicator is taken and aborted: none is left.
                k.gate->cv.wait_for(lk, std::chrono::milliseconds(200), [&] { return !k.gate->in_call; });
                cm = k.comm; k.comm = nullptr;
            }
            if (cm) (void)rccl().CommAbort(cm);              // ends the kernels of stuck collectives and frees the communicator
        }
    { std::lock_guard<std::mutex> lk(M->mm); M->mcv.notify_all(); }
    { std::lock_guard<std::mutex> lk(M->bm); M->bcv.notify_all(); }
}

int aborted_fail(vp_multi* M, vp_ctx* c)
{
    std::lock_guard<std::mutex> lk(M->am);
    return vp_fail(c, VP_ERR_RCCL, "fan-out context aborted (%s); destroy it and create a new one", M->abort_msg.c_str());
}

// hipStreamSynchronize with a time-out: an exchange whose peer never arrives (a rank of another process died, a link went down) leaves its
// kernel spinning on the device for ever.  Polls the stream, the communicator's asynchronous error state and the clock.
int kid_wait(vp_multi* M, Kid& k, const char* what)
{
    vp_ctx* c = k.c;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        hipError_t e = hipStreamQuery(k.stream);
        if (e == hipSuccess && k.xstream) e = hipStreamQuery(k.xstream);       // (an all-gather still running beside an idle compute stream)
        if (e == hipSuccess) return M->aborted.load() ? aborted_fail(M, c) : VP_OK;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); multi_abort(M, k.rank, std::string(what) + ": " + hipGetErrorString(e)); return aborted_fail(M, c); }
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (M->aborted.load() && ms > 2000) return aborted_fail(M, c);           // aborted elsewhere and the stream does not drain: give up on it
        if (M->use_rccl && rccl().CommGetAsyncError && (spin & 63) == 63) {
            ncclResult_t ar = ncclSuccess;
            bool have = false;
            if (ncclComm_t cm = comm_enter(M, k)) { CommLeave leave{k}; have = rccl().CommGetAsyncError(cm, &ar) == ncclSuccess; }
            if (have && ar != ncclSuccess && ar != ncclInProgress) {
                multi_abort(M, k.rank, std::string(what) + ": asynchronous RCCL error: " + rccl().GetErrorString(ar));
                return aborted_fail(M, c);
            }
        }
        if (!M->aborted.load() && ms > M->timeout_ms) {
            multi_abort(M, k.rank, std::string(what) + ": no completion after " + std::to_string(M->timeout_ms) + " ms (a peer left the exchange?)");
            return aborted_fail(M, c);
        }
        if (ms < 10) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(100));     // a frame's wait is a few ms: no sleep granularity on it
    }
}

void worker_main(vp_multi* M, int i)
{
    (void)hipSetDevice(M->kids[i].device);
    uint64_t seen = 0;
    for (;;) {
        std::function<int(Kid&)> job;
        bool exchanges = true;
        {
            std::unique_lock<std::mutex> lk(M->m);
            M->cv_go.wait(lk, [&] { return M->quit || M->gen != seen; });
            if (M->quit) return;
            seen = M->gen;
            job = M->job;
            exchanges = M->job_exchanges;
        }
        M->kids[i].voted = false;
        const int rc = job(M->kids[i]);
        if (rc && !M->kids[i].voted && exchanges) multi_abort(M, M->kids[i].rank, M->kids[i].c->err);      // its peers may be waiting for it in an exchange
        {
            std::lock_guard<std::mutex> lk(M->m);
            M->kids[i].rc = rc;
            if (--M->pending == 0) M->cv_done.notify_all();
        }
    }
}

// Run fn on every local rank (SPMD).  One local rank: inline on the calling thread.  Returns the first failure, its message copied to the
// fan-out context.  exchanges: the job contains inter-rank exchanges, so a rank that fails WITHOUT a vote may leave its peers inside one: the
// context is aborted.  A job without exchanges (frame / particle upload / occluders / sync: plain argument checks and copies per rank) only
// reports: a VP_ERR_BAD_ARG every rank returns identically must stay recoverable.
int run_all(vp_multi* M, const std::function<int(Kid&)>& fn, bool exchanges = true)
{
    if (M->aborted.load()) return aborted_fail(M, M->parent);
    if (M->nlocal == 1) {
        (void)hipSetDevice(M->kids[0].device);
        M->kids[0].voted = false;
        M->kids[0].rc = fn(M->kids[0]);
        if (M->kids[0].rc && !M->kids[0].voted && M->world > 1 && exchanges) multi_abort(M, M->kids[0].rank, M->kids[0].c->err);   // the other processes find out by their time-outs
    } else {
        std::unique_lock<std::mutex> lk(M->m);
        M->job = fn;
        M->job_exchanges = exchanges;
        M->pending = M->nlocal;
        ++M->gen;
        M->cv_go.notify_all();
        M->cv_done.wait(lk, [&] { return M->pending == 0; });
    }
    for (Kid& k : M->kids)
        if (k.rc) { M->parent->err = "rank " + std::to_string(k.rank) + ": " + k.c->err; return k.rc; }
    return VP_OK;
}

// Barrier of the local ranks that also agrees on an error: a rank that failed BEFORE a collective must not leave the others blocked in it.
int vote(vp_multi* M, int rc)
{
    if (M->nlocal == 1) return rc;
    std::unique_lock<std::mutex> lk(M->bm);
    if (rc && !M->b_rc) M->b_rc = rc;
    const uint64_t g = M->b_gen;
    if (++M->b_count == M->nlocal) {
        M->b_result = M->b_rc; M->b_rc = 0; M->b_count = 0; ++M->b_gen;
        M->bcv.notify_all();
    } else {
        M->bcv.wait(lk, [&] { return M->b_gen != g || M->aborted.load(); });
        if (M->b_gen == g) return rc ? rc : VP_ERR_RCCL;           // aborted while waiting: the vote never completed
    }
    return rc ? rc : M->b_result;
}
#define VP_VOTE(expr)                                                                                                          \
    do {                                                                                                                       \
        const int own_ = (expr);                                                                                               \
        const int all_ = vote(M, own_);                                                                                        \
        if (all_) { k.voted = !M->aborted.load(); return own_ ? own_ : vp_fail(c, all_, "another local rank failed before the collective"); } \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------------------
// Exchanges.  RCCL: grouped ncclSend / ncclRecv and ncclAllGather on the rank's communicator and stream.  Loopback (test hook):
// receiver posts its buffer, sender copies device to device on ITS stream and records an event, receiver's stream waits for the event.
// ---------------------------------------------------------------------------------------------------------------------------------
// Before the compute stream issues an operation on the communicator: the all-gather the exchange stream may still be running has to be over
// (one operation of a communicator at a time, in the order every rank issues them).
int comm_fence(Kid& k)
{
    vp_ctx* c = k.c;
    if (k.tau_pending) { VP_HIP(hipStreamWaitEvent(k.stream, k.ev_tau, 0)); k.tau_pending = false; }
    return VP_OK;
}

int p2p_batch(vp_multi* M, Kid& k, const std::vector<P2P>& ops, hipStream_t stream = nullptr)
{
    vp_ctx* c = k.c;
    if (ops.empty()) return VP_OK;
    if (!stream) { stream = k.stream; const int rf = comm_fence(k); if (rf) return rf; }
    if (M->use_rccl) {
        Rccl& R = rccl();
        const ncclComm_t cm = comm_enter(M, k);
        if (!cm) return aborted_fail(M, c);                 // aborted, or taken away by multi_abort
        CommLeave leave{k};
        VP_NCCL(R.GroupStart());
        for (const P2P& o : ops) {
            const ncclResult_t r = o.send ? R.Send(o.ptr, o.bytes, ncclUint8, o.peer, cm, stream) : R.Recv(o.ptr, o.bytes, ncclUint8, o.peer, cm, stream);
            if (r != ncclSuccess) { (void)R.GroupEnd(); return vp_fail(c, VP_ERR_RCCL, "ncclSend/ncclRecv failed: %s", R.GetErrorString(r)); }
        }
        VP_NCCL(R.GroupEnd());
        return M->aborted.load() ? aborted_fail(M, c) : VP_OK;   // aborted under the call: `cm` is gone, never touched again
    }
    // loopback: 1. post every receive buffer (never blocks), 2. do the sends (each waits for the peer's post), 3. wait for the receives
    std::vector<std::tuple<int, int, uint64_t>> mine;
    {
        std::lock_guard<std::mutex> lk(M->mm);
        for (const P2P& o : ops)
            if (!o.send) {
                const auto key = std::make_tuple(o.peer, k.rank, k.seq_from[o.peer]++);
                Mail& ml = M->box[key];
                ml.dst = o.ptr; ml.bytes = o.bytes; ml.posted = true;
                mine.push_back(key);
            }
        M->mcv.notify_all();
    }
    for (const P2P& o : ops)
        if (o.send) {
            const auto key = std::make_tuple(k.rank, o.peer, k.seq_to[o.peer]++);
            void* dst = nullptr;
            {
                std::unique_lock<std::mutex> lk(M->mm);
                const bool ok = M->mcv.wait_for(lk, std::chrono::milliseconds(M->timeout_ms),
                                                [&] { auto it = M->box.find(key); return (it != M->box.end() && it->second.posted) || M->aborted.load(); });
                if (!ok || M->aborted.load()) {
                    lk.unlock();
                    if (!ok) multi_abort(M, k.rank, "exchange: rank " + std::to_string(o.peer) + " never posted its receive buffer");
                    return aborted_fail(M, c);
                }
                Mail& ml = M->box[key];
                if (ml.bytes != o.bytes) return vp_fail(c, VP_ERR_STATE, "loopback exchange: size mismatch between ranks %d and %d", k.rank, o.peer);
                dst = ml.dst;
            }
            if (k.drop_next_send) { k.drop_next_send = false; continue; }      // TEST HOOK: this rank silently leaves the exchange
            hipEvent_t ev = nullptr;
            VP_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            VP_HIP(hipMemcpyAsync(dst, o.ptr, o.bytes, hipMemcpyDefault, stream));
            VP_HIP(hipEventRecord(ev, stream));
            {
                std::lock_guard<std::mutex> lk(M->mm);
                Mail& ml = M->box[key];
                ml.ev = ev; ml.done = true;
                M->mcv.notify_all();
            }
        }
    for (const auto& key : mine) {
        hipEvent_t ev = nullptr;
        {
            std::unique_lock<std::mutex> lk(M->mm);
            const bool ok = M->mcv.wait_for(lk, std::chrono::milliseconds(M->timeout_ms), [&] { return M->box[key].done || M->aborted.load(); });
            if (!ok || !M->box[key].done) {
                const int peer = std::get<0>(key);
                lk.unlock();
                if (!ok) multi_abort(M, k.rank, "exchange: nothing arrived from rank " + std::to_string(peer) + " within " + std::to_string(M->timeout_ms) + " ms");
                return aborted_fail(M, c);
            }
            ev = M->box[key].ev;
            M->box.erase(key);
        }
        VP_HIP(hipStreamWaitEvent(stream, ev, 0));
        VP_HIP(hipEventDestroy(ev));                     // released once the wait has consumed it
    }
    return VP_OK;
}

// all-gather IN PLACE: rank r's block already sits at buf + r * count
int all_gather_inplace(vp_multi* M, Kid& k, float* buf, size_t count, hipStream_t stream = nullptr)
{
    vp_ctx* c = k.c;
    if (!stream) { stream = k.stream; const int rf = comm_fence(k); if (rf) return rf; }
    if (M->use_rccl) {
        const ncclComm_t cm = comm_enter(M, k);
        if (!cm) return aborted_fail(M, c);                 // aborted, or taken away by multi_abort
        CommLeave leave{k};
        VP_NCCL(rccl().AllGather(buf + (size_t)k.rank * count, buf, count, ncclFloat, cm, stream));
        return M->aborted.load() ? aborted_fail(M, c) : VP_OK;
    }
    std::vector<P2P> ops;
    for (int r = 0; r < M->world; ++r)
        if (r != k.rank) {
            Kid* peer = local_kid(M, r);
            if (!peer) return vp_fail(c, VP_ERR_STATE, "loopback exchange needs every rank in this process");
            ops.push_back(P2P{false, r, buf + (size_t)r * count, count * sizeof(float)});
            ops.push_back(P2P{true, r, buf + (size_t)k.rank * count, count * sizeof(float)});
        }
    return p2p_batch(M, k, ops, stream);
}

int copy_on_stream(Kid& k, void* dst, const void* src, size_t bytes)
{
    vp_ctx* c = k.c;
    if (dst != src && bytes) VP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, k.stream));
    return VP_OK;
}

void free_kid(Kid& k, bool aborted = false)
{
    if (!k.c && !k.stream) return;
    (void)hipSetDevice(k.device);
    if (k.stream) {
        // bounded: a stream still stuck in an aborted exchange must not hang vp_destroy
        const auto t0 = std::chrono::steady_clock::now();
        while (hipStreamQuery(k.stream) == hipErrorNotReady &&
               std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() < 5000)
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        (void)hipGetLastError();
    }
    if (k.xstream) {
        const auto t1 = std::chrono::steady_clock::now();
        while (hipStreamQuery(k.xstream) == hipErrorNotReady &&
               std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t1).count() < 5000)
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        (void)hipGetLastError();
    }
    if (k.comm && rccl().so) {
        // a communicator of an ABORTED context that nobody took (no ncclCommAbort at abort time, or a kid created after it) may hold stuck
        // operations: ncclCommDestroy would wait for them.  Abort it; destroy only communicators of a healthy context.
        if (aborted && rccl().CommAbort) (void)rccl().CommAbort(k.comm); else if (!aborted) (void)rccl().CommDestroy(k.comm);
    }
    void* bufs[] = {k.d_tau_all, k.d_img[0], k.d_img[1], k.d_tmaps, k.d_tout[0], k.d_tout[1], k.d_pieces, k.d_piece_out, k.d_final, k.d_xfer};
    for (void* p : bufs) if (p) (void)hipFree(p);
    for (auto& e : k.ev) for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x);
    if (k.c) { k.c->stream = nullptr; vp_destroy_single(k.c); }
    if (k.stream) (void)hipStreamDestroy(k.stream);
    if (k.xstream) (void)hipStreamDestroy(k.xstream);
    if (k.ev_local) (void)hipEventDestroy(k.ev_local);
    if (k.ev_tau) (void)hipEventDestroy(k.ev_tau);
    k = Kid{};
}

bool is_a_only(const vp_multi* M, int rank, int zb) { return M->cuts[rank + 1] - 1 <= zb; }     // every slice of the slab is drawn in phase A

// ---------------------------------------------------------------------------------------------------------------------------------
// Slab cut.  Collective (every rank computes the same cut from the same gathered numbers).
// ---------------------------------------------------------------------------------------------------------------------------------
int plan_slabs(vp_multi* M, Kid& k)
{
    vp_ctx* c = k.c;
    const int nz = c->g.Nz, world = M->world;
    std::vector<int> cuts(world + 1);
    if ((M->flags & VP_MULTI_UNIFORM_SLABS) || world == 1) {
        for (int i = 0; i <= world; ++i) cuts[i] = (int)(((long long)i * nz) / world);
    } else {
        // fill work: (particle, metavoxel) pairs per light-axis slice over the WHOLE grid (every rank has all particles: same histogram)
        int rc = launch_z_histogram(c, c->d_cursor); if (rc) return rc;
        std::vector<int> pairs(nz);
        VP_HIP(hipMemcpyAsync(pairs.data(), c->d_cursor, (size_t)nz * sizeof(int), hipMemcpyDeviceToHost, k.stream));
        // last frame's measurements of every rank: samples executed per slice, kernel times
        const int stride = nz + 8;
        std::vector<float> all((size_t)world * stride, 0.f);
        if (M->have_profile) {
            k.h_xfer.assign(stride, 0.f);
            std::vector<long long> zs(nz);
            rc = api_read_zsamples(c, zs.data(), false); if (rc) return rc;
            for (int z = 0; z < nz; ++z) k.h_xfer[z] = (float)zs[z];
            float ms = 0.f;
            if (c->ev_valid[1] && hipEventElapsedTime(&ms, c->ev[1][0], c->ev[1][1]) == hipSuccess) k.h_xfer[nz] = ms;   // the local (rank 0: fused) pass; the planner adds the finish pass as a fixed share
            k.h_xfer[nz + 1] = (float)c->h_meta.pairs;
            if (c->ev_valid[2] && hipEventElapsedTime(&ms, c->ev[2][0], c->ev[2][1]) == hipSuccess) k.h_xfer[nz + 2] = ms;
            double s = 0.0;
            for (int z = 0; z < nz; ++z) s += zs[z];
            k.h_xfer[nz + 3] = (float)s;
            VP_HIP(hipMemcpyAsync(k.d_xfer + (size_t)k.rank * stride, k.h_xfer.data(), stride * sizeof(float), hipMemcpyHostToDevice, k.stream));
            rc = all_gather_inplace(M, k, k.d_xfer, stride); if (rc) return rc;
            VP_HIP(hipMemcpyAsync(all.data(), k.d_xfer, all.size() * sizeof(float), hipMemcpyDeviceToHost, k.stream));
        }
        { const int rw = kid_wait(M, k, "slab re-cut (all-gather of the work profile)"); if (rw) return rw; }
        c->binned = c->filled = c->local_done = false;                 // the cursor scratch was reused
        // ms per pair / per sample: measured when a frame has run (sum of kernel ms over ranks / sum of units), else the MI355X C3 figures
        double fill_ms_sum = 0, pairs_sum = 0, rm_ms_sum = 0, samp_sum = 0;
        for (int r = 0; r < world; ++r) {
            const float* p = all.data() + (size_t)r * stride;
            fill_ms_sum += p[nz]; pairs_sum += p[nz + 1]; rm_ms_sum += p[nz + 2]; samp_sum += p[nz + 3];
        }
        const double ms_per_pair = (fill_ms_sum > 0 && pairs_sum > 0) ? fill_ms_sum / pairs_sum : 6.9e-6;
        const double ms_per_sample = (rm_ms_sum > 0 && samp_sum > 0) ? rm_ms_sum / samp_sum : 3.0e-9;
        std::vector<double> fill_ms(nz), rm_ms(nz, 0.0);
        for (int z = 0; z < nz; ++z) fill_ms[z] = pairs[z] * ms_per_pair;
        bool have_rm = false;
        for (int r = 0; r < world; ++r)
            for (int z = 0; z < nz; ++z) { rm_ms[z] += all[(size_t)r * stride + z] * ms_per_sample; have_rm = have_rm || all[(size_t)r * stride + z] > 0.f; }
        // every local rank holds the same numbers: ONE of them runs the planner (its exact search is ~world nz^4 / 2 steps), the others take
        // the published cut after the barrier
        if (k.index == 0) { hl_plan_slabs(nz, world, fill_ms.data(), have_rm ? rm_ms.data() : nullptr, M->groups, cuts.data()); M->cuts = cuts; }
        VP_VOTE(VP_OK);
        cuts = M->cuts;
        return api_set_slab(c, cuts[k.rank], cuts[k.rank + 1]);
    }
    if (k.index == 0) M->cuts = cuts;
    return api_set_slab(c, cuts[k.rank], cuts[k.rank + 1]);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// Lifetime
// ---------------------------------------------------------------------------------------------------------------------------------
int multi_create(const vp_config* cfg, vp_ctx** out)
{
    const int nlocal = cfg->num_devices > 0 ? cfg->num_devices : 1;
    const int world = cfg->world_size > 0 ? cfg->world_size : nlocal;
    const int first = cfg->world_size > 0 ? cfg->first_rank : 0;
    const bool loopback = (cfg->multi_flags & VP_MULTI_PEER_COPY) != 0;
    // TEST HOOK: ranks that share a GPU on the RCCL path -- only a stand-in librccl supports that (tests/tools/fake_rccl.cpp); the real
    // library refuses a duplicate device in ncclCommInitAll
    const bool shared_dev_hook = (cfg->multi_flags & VP_MULTI_TEST_HOOKS) && (cfg->multi_flags & VP_MULTI_TEST_SHARED_DEVICE);
    if (first < 0 || first + nlocal > world) return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: ranks [%d, %d) outside world_size %d", first, first + nlocal, world);
    if (world > cfg->num_mv[2]) return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: %d slabs for %d light-axis slices (at most one rank per slice)", world, cfg->num_mv[2]);
    if (loopback && nlocal != world) return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: VP_MULTI_PEER_COPY needs every rank in this process");
    if (cfg->slab_z0 != 0 || cfg->slab_z1 != 0) return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: a fan-out context cuts its own slabs (slab_z0 = slab_z1 = 0)");
    if (cfg->rm_groups < 0) return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: rm_groups %d", cfg->rm_groups);
    if (cfg->reserved[2] < 0 || cfg->reserved[2] > 3600000) return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: exchange time-out %d ms (vp_config.reserved[2]: 0 = 120 s, at most an hour)", cfg->reserved[2]);
    if ((cfg->multi_flags & VP_MULTI_TEST_DROP_SEND) && !((cfg->multi_flags & VP_MULTI_TEST_HOOKS) && loopback))
        return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: VP_MULTI_TEST_DROP_SEND needs VP_MULTI_TEST_HOOKS and VP_MULTI_PEER_COPY");
    int devs[VP_MAX_LOCAL_DEVICES];
    for (int i = 0; i < nlocal; ++i) devs[i] = cfg->num_devices > 0 ? cfg->devices[i] : cfg->device;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return vp_fail(nullptr, VP_ERR_NO_DEVICE, "vp_create: no HIP device; libvpfx has no CPU fallback");
    for (int i = 0; i < nlocal; ++i) {
        if (devs[i] < 0) { if (hipGetDevice(&devs[i]) != hipSuccess) devs[i] = 0; }
        if (devs[i] >= ndev) return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: device %d of %d", devs[i], ndev);
        for (int j = 0; j < i && !loopback && !shared_dev_hook; ++j)
            if (devs[j] == devs[i]) return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: device %d listed twice (RCCL needs one GPU per rank; VP_MULTI_PEER_COPY is the one-GPU test hook)", devs[i]);
    }
    if (!loopback && !rccl().load()) return vp_fail(nullptr, VP_ERR_RCCL, "vp_create: %s", rccl().err.c_str());

    vp_ctx* P = new (std::nothrow) vp_ctx();
    vp_multi* M = new (std::nothrow) vp_multi();
    if (!P || !M) { delete P; delete M; return vp_fail(nullptr, VP_ERR_OOM, "vp_create: host allocation failed"); }
    P->cfg = *cfg;
    P->multi = M;
    P->device = devs[0];
    M->parent = P;
    M->world = world; M->nlocal = nlocal; M->first_rank = first; M->flags = cfg->multi_flags;
    M->groups = cfg->rm_groups > 0 ? std::min(cfg->rm_groups, world) : 1;
    M->use_rccl = !loopback;
    M->timeout_ms = cfg->reserved[2] > 0 ? cfg->reserved[2] : 120000;   // (generous: the first exchanges also pay for RCCL's lazy peer-connection set-up)         // vp_config.reserved[2]: exchange time-out in ms (fan-out contexts)
    M->npix = (size_t)cfg->width * cfg->height;
    M->piece = (M->npix + world - 1) / world;
    M->pixpad = M->piece * world;
    M->lm = (size_t)cfg->num_mv[0] * cfg->num_voxels * cfg->num_mv[1] * cfg->num_voxels;
    M->cuts.resize(world + 1);
    for (int i = 0; i <= world; ++i) M->cuts[i] = (int)(((long long)i * cfg->num_mv[2]) / world);
    M->kids.resize(nlocal);
    const bool gather_all = (cfg->multi_flags & VP_MULTI_EXCHANGE_ALL_GATHER) != 0;
    auto fail = [&](int code, const char* what) {
        std::string msg = what;
        for (Kid& k : M->kids) if (k.c && !k.c->err.empty()) msg += ": " + k.c->err;
        if (msg == what && !g_vp_create_error.empty()) msg += ": " + g_vp_create_error;
        for (Kid& k : M->kids) free_kid(k, M->aborted.load() != 0);
        delete M; delete P;
        return vp_fail(nullptr, code, "vp_create (fan-out): %s", msg.c_str());
    };
    for (int i = 0; i < nlocal; ++i) {
        Kid& k = M->kids[i];
        k.index = i; k.rank = first + i; k.device = devs[i];
        k.seq_to.assign(world, 0); k.seq_from.assign(world, 0);
        k.drop_next_send = (cfg->multi_flags & VP_MULTI_TEST_DROP_SEND) && k.rank == world - 1;
        vp_config one = *cfg;
        one.num_devices = 0; one.world_size = 0; one.multi_flags = cfg->multi_flags & VP_MULTI_TEST_HOOKS; one.first_rank = 0;
        one.reserved[2] = 0;
        one.device = devs[i];
        one.slab_z0 = M->cuts[k.rank]; one.slab_z1 = M->cuts[k.rank + 1];
        int rc = vp_create_single(&one, &k.c);
        if (rc) return fail(rc, "slab context");
        // The exchange stream gets the device's HIGHEST priority: the all-gather of the transmittance maps is launched while the first slab's
        // ray-march fills every CU (the planner counts on the two running side by side: host_logic.cpp, two_maxima_partition); at equal priority
        // its kernel could queue behind the march's remaining workgroups and hold back every other rank's finish pass.  With a higher
        // priority the dispatcher hands it the first CUs that free up.  (No priorities on the device: lo == hi, a plain stream.)
        int prio_lo = 0, prio_hi = 0;
        if (hipSetDevice(k.device) != hipSuccess) return fail(VP_ERR_HIP, "hipSetDevice");
        if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) { (void)hipGetLastError(); prio_lo = prio_hi = 0; }
        if (hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithPriority(&k.xstream, hipStreamNonBlocking, prio_hi) != hipSuccess || hipEventCreateWithFlags(&k.ev_local, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&k.ev_tau, hipEventDisableTiming) != hipSuccess) return fail(VP_ERR_HIP, "hipStreamCreate");
        k.c->stream = k.stream;
        const size_t img = M->pixpad * 4, plane = M->npix;
        const size_t pieces = gather_all ? (size_t)(world + 1) * M->pixpad * 4 : (size_t)(world + 1) * M->piece * 4;
        struct { void** p; size_t n; } allocs[] = {
            {(void**)&k.d_tau_all, (size_t)world * M->lm * 4}, {(void**)&k.d_img[0], img * 4}, {(void**)&k.d_img[1], img * 4},
            {(void**)&k.d_tmaps, (size_t)world * plane}, {(void**)&k.d_tout[0], plane}, {(void**)&k.d_tout[1], plane},
            {(void**)&k.d_pieces, pieces * 4}, {(void**)&k.d_piece_out, M->piece * 16}, {(void**)&k.d_final, (k.rank == 0 ? img : 4) * 4},
            {(void**)&k.d_xfer, (size_t)world * (cfg->num_mv[2] + 8) * 4}};
        for (auto& a : allocs)
            if (hipMalloc(a.p, a.n) != hipSuccess) { k.c->err = "hipMalloc of the exchange buffers failed"; return fail(VP_ERR_OOM, "exchange buffers"); }
        // the padding pixels of the partial images are exchanged and blended like any others: keep them defined
        if (hipMemsetAsync(k.d_img[0], 0, img * sizeof(float), k.stream) != hipSuccess || hipMemsetAsync(k.d_img[1], 0, img * sizeof(float), k.stream) != hipSuccess)
            return fail(VP_ERR_HIP, "hipMemset");
        for (auto& e : k.ev) for (hipEvent_t& x : e) if (hipEventCreate(&x) != hipSuccess) return fail(VP_ERR_HIP, "hipEventCreate");
        if (loopback)
            for (int j = 0; j < i; ++j)
                if (devs[j] != devs[i]) {                      // best effort: without peer access the copies are staged by the runtime
                    (void)hipSetDevice(devs[i]); (void)hipDeviceEnablePeerAccess(devs[j], 0);
                    (void)hipSetDevice(devs[j]); (void)hipDeviceEnablePeerAccess(devs[i], 0);
                    (void)hipGetLastError();
                }
        if (hipStreamSynchronize(k.stream) != hipSuccess) return fail(VP_ERR_HIP, "hipStreamSynchronize");
    }
    if (M->use_rccl) {
        Rccl& R = rccl();
        ncclComm_t comms[VP_MAX_LOCAL_DEVICES] = {};
        ncclResult_t r = ncclSuccess;
        if (cfg->world_size == 0) {
            r = R.CommInitAll(comms, nlocal, devs);            // one process drives every GPU of the job
        } else {
            ncclUniqueId id;
            static_assert(sizeof(id) == 128, "rccl_unique_id is 128 bytes");
            memcpy(&id, cfg->rccl_unique_id, sizeof id);
            r = R.GroupStart();
            for (int i = 0; i < nlocal && r == ncclSuccess; ++i) {
                if (hipSetDevice(devs[i]) != hipSuccess) { r = ncclUnhandledCudaError; break; }
                r = R.CommInitRank(&comms[i], world, id, first + i);
            }
            const ncclResult_t r2 = R.GroupEnd();
            if (r == ncclSuccess) r = r2;
        }
        if (r != ncclSuccess) { M->kids[0].c->err = std::string("RCCL communicator: ") + R.GetErrorString(r); return fail(VP_ERR_RCCL, "ncclCommInit"); }
        for (int i = 0; i < nlocal; ++i) M->kids[i].comm = comms[i];
        (void)R.CommCount(comms[0], &M->rccl_ranks);
        if (M->rccl_ranks != world) { M->kids[0].c->err = "communicator size differs from world_size"; return fail(VP_ERR_RCCL, "ncclCommCount"); }
    }
    if (nlocal > 1)
        for (int i = 0; i < nlocal; ++i) M->threads.emplace_back(worker_main, M, i);
    *out = P;
    return VP_OK;
}

void multi_destroy(vp_ctx* P)
{
    vp_multi* M = P->multi;
    if (M) {
        {
            std::lock_guard<std::mutex> lk(M->m);
            M->quit = true;
            M->cv_go.notify_all();
        }
        for (std::thread& t : M->threads) t.join();
        for (Kid& k : M->kids) free_kid(k, M->aborted.load() != 0);
        delete M;
    }
    delete P;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Per-frame entry points (each: one SPMD pass over the local ranks)
// ---------------------------------------------------------------------------------------------------------------------------------
int multi_set_frame(vp_ctx* P, const float* l2w, const float* gc)
{
    vp_multi* M = P->multi;
    // (every job starts with a bounded wait for what the previous frame left on the rank's stream -- ranks other than the display rank return
    //  from vp_raymarch with their exchanges still queued --: the copies and host polls below would otherwise block without a time-out behind an
    //  exchange whose peer is gone.  One hipStreamQuery when the stream is idle.)
    const int rc = run_all(M, [&](Kid& k) -> int { const int rw = kid_wait(M, k, "vp_set_frame (previous frame's exchanges)"); return rw ? rw : vp_set_frame(k.c, l2w, gc); }, false);
    if (!rc) { P->have_frame = true; P->binned = P->filled = false; }
    return rc;
}

int multi_upload_particles(vp_ctx* P, const void* particles, int32_t count, const vp_particle_layout* lay, const float* psys_l2w)
{
    vp_multi* M = P->multi;
    // every rank bins ALL particles against its own slab: the array goes to every device (the copies of different devices overlap)
    const int rc = run_all(M, [&](Kid& k) -> int {
        int r = kid_wait(M, k, "vp_upload_particles (previous frame's exchanges)");
        if (!r) r = api_upload_particles(k.c, particles, count, lay, psys_l2w, false);
        return r ? r : api_stream_sync(k.c);
    }, false);
    if (!rc) { P->have_particles = true; P->binned = P->filled = false; }
    return rc;
}

int multi_bin_resident(vp_ctx* P)
{
    vp_multi* M = P->multi;
    if (!P->have_frame) return vp_fail(P, VP_ERR_STATE, "vp_bin before vp_set_frame");
    if (!P->have_particles) return vp_fail(P, VP_ERR_STATE, "vp_bin_resident before vp_upload_particles");
    const bool plan = M->need_plan;
    const int rc = run_all(M, [&](Kid& k) -> int {
        vp_ctx* c = k.c;
        (void)hipSetDevice(k.device);
        VP_VOTE(kid_wait(M, k, "vp_bin (previous frame's exchanges)"));       // the bin's host polls its totals: never behind a stuck exchange
        if (plan) {
            VP_VOTE(VP_OK);                                    // every local rank is here: the collective inside plan_slabs can complete
            int r = plan_slabs(M, k); if (r) return r;
        }
        // the bin itself exchanges nothing: its result is agreed on, so that a slab whose pair count overflows (VP_ERR_UNSUPPORTED) fails the
        // call on every rank alike instead of aborting the context
        VP_VOTE(vp_bin_resident(c));
        return VP_OK;
    });
    if (!rc) { M->need_plan = false; P->binned = true; P->filled = false; }
    return rc;
}

int multi_fill(vp_ctx* P, const vp_fill_params* p)
{
    vp_multi* M = P->multi;
    if (!P->binned) return vp_fail(P, VP_ERR_STATE, "vp_fill before vp_bin");
    const int rc = run_all(M, [&](Kid& k) -> int {
        vp_ctx* c = k.c;
        (void)hipSetDevice(k.device);
        // (uploads of the fill's inputs synchronise the stream: bounded wait first; nothing to upload -> nothing blocks, the fill queues behind the bin)
        int rc0 = (p->cubemap || p->light_depth_map) ? kid_wait(M, k, "vp_fill (previous frame's exchanges)") : VP_OK;
        if (!rc0 && (!c->have_frame || !c->binned)) rc0 = vp_fail(c, VP_ERR_STATE, "vp_fill before vp_set_frame / vp_bin");
        if (!rc0) rc0 = api_stage_fill_inputs(c, p);
        if (!rc0) rc0 = api_ensure_bricks(c, k.rank != 0);          // (rank 0 needs no (density, ao) scratch: fused fill, below)
        VP_VOTE(rc0);
        // The slab nearest the light knows its incoming light (1): it runs the FUSED fill -- bricks stored at once, no (density, ao) scratch
        // round trip, no finish pass -- and its light map IS its transmittance map.  It is also the slab that usually holds most of the
        // ray-march (light and camera on the same side, the benchmark's case), which it can now start while the others finish.
        const bool fused = k.rank == 0;
        int r;
        c->filled = false;
        // (the previous frame's all-gather read this rank's slot of d_tau_all on the exchange stream: over before the fill below rewrites it)
        { const int rf = comm_fence(k); if (rf) return rf; }
        if (fused) {
            r = launch_fill(c, 0, nullptr, c->d_lightmap);
            if (!r && hipMemcpyAsync(k.d_tau_all, c->d_lightmap, M->lm * sizeof(float), hipMemcpyDeviceToDevice, k.stream) != hipSuccess)
                r = vp_fail(c, VP_ERR_HIP, "hipMemcpyAsync (transmittance map) failed");
            c->local_done = false; c->ev_valid[3] = false;
        } else {
            // slab-local pass (T_in = 1): density / ao to scratch, the slab's transmittance map straight into its slot of the gather buffer
            r = launch_fill(c, 1, nullptr, k.d_tau_all + (size_t)k.rank * M->lm);
            c->local_done = r == VP_OK;
        }
        VP_VOTE(r);                                             // a rank whose launch failed must not leave the others inside the all-gather
        // the all-gather runs on the exchange stream, after this rank's map is written ...
        VP_HIP(hipEventRecord(k.ev_local, k.stream));
        VP_HIP(hipStreamWaitEvent(k.xstream, k.ev_local, 0));
        VP_HIP(hipEventRecord(k.ev[0][0], k.xstream));
        r = all_gather_inplace(M, k, k.d_tau_all, M->lm, k.xstream); if (r) return r;
        VP_HIP(hipEventRecord(k.ev[0][1], k.xstream));
        VP_HIP(hipEventRecord(k.ev_tau, k.xstream));
        k.ev_valid[0] = true;
        k.tau_pending = true;
        if (!fused) {
            // ... and only the finish pass waits for it (rank 0's compute stream carries on: its next wait is in front of its next exchange)
            { const int rf = comm_fence(k); if (rf) return rf; }
            // finish pass: T_in = tau[0] * ... * tau[rank - 1] formed inside the kernel, straight from the receive buffer
            c->finish_tau_all = k.d_tau_all;
            c->finish_n_before = k.rank;
            r = launch_fill(c, 2, nullptr, c->d_lightmap);
            c->finish_tau_all = nullptr; c->finish_n_before = 0;
            if (r) return r;
        }
        c->filled = true;
        return VP_OK;
    });
    if (!rc) P->filled = true;
    return rc;
}

int multi_raymarch(vp_ctx* P, const vp_camera* cam, const vp_raymarch_params* rp, float* host_out, void* d_out)
{
    vp_multi* M = P->multi;
    if (!cam || !rp) return vp_fail(P, VP_ERR_BAD_ARG, "null camera / params");
    if (!P->filled) return vp_fail(P, VP_ERR_STATE, "vp_raymarch before vp_fill");
    const int world = M->world;
    const bool gather_all = (M->flags & VP_MULTI_EXCHANGE_ALL_GATHER) != 0;
    const int rc = run_all(M, [&](Kid& k) -> int {
        vp_ctx* c = k.c;
        (void)hipSetDevice(k.device);
        RmConsts kc;
        // (a scene-depth upload from pageable memory blocks in the runtime: bounded wait first; without one the march queues behind the fill)
        int rc0 = rp && rp->scene_depth ? kid_wait(M, k, "vp_raymarch (previous exchanges)") : VP_OK;
        if (!rc0) rc0 = api_stage_raymarch(c, cam, rp, &kc);
        VP_VOTE(rc0);
        kc.partial = 1;
        const int zb = kc.zB;
        int chain[VP_MAX_RANKS], plan_rank[VP_MAX_RANKS + 1], plan_which[VP_MAX_RANKS + 1], plan_kind[VP_MAX_RANKS + 1], pos_group[VP_MAX_RANKS], strad = -1;
        const int n_plan = hl_blend_plan(world, M->cuts.data(), zb, chain, plan_rank, plan_which, plan_kind, &strad);
        // debug views / UNORM8 emulation march without the early-out: nothing to hand on
        const int G = rp->flags ? 1 : M->groups;
        hl_chain_groups(world, G, pos_group);
        int group_of[VP_MAX_RANKS], my_pos = 0;
        for (int p = 0; p < world; ++p) { group_of[chain[p]] = pos_group[p]; if (chain[p] == k.rank) my_pos = p; }
        if (k.index == 0) { memcpy(M->chain, chain, sizeof chain); memcpy(M->group_of, group_of, sizeof group_of); }
        const int my_group = group_of[k.rank];
        // (1) saturation hand-off in: the transmittance maps of every slab of an earlier group (all of them are composited in front of this one)
        VP_HIP(hipEventRecord(k.ev[1][0], k.stream));
        std::vector<P2P> ops;
        int n_in = 0;
        for (int p = 0; p < world; ++p) {
            const int s = chain[p];
            if (group_of[s] < my_group) { ops.push_back(P2P{false, s, k.d_tmaps + (size_t)n_in * M->npix, M->npix}); ++n_in; }
        }
        int r = p2p_batch(M, k, ops); if (r) return r;
        VP_HIP(hipEventRecord(k.ev[1][1], k.stream));
        k.ev_valid[1] = true;
        // (2) the slab's partial images
        float* keep = c->d_scene_depth;
        if (!rp->scene_depth && c->n_occluders == 0) c->d_scene_depth = nullptr;
        RmHandoff ho{};
        ho.t_in = n_in ? k.d_tmaps : nullptr; ho.n_in = n_in; ho.plane = M->npix;
        ho.t_out0 = k.d_tout[0]; ho.t_out1 = k.d_tout[1]; ho.zsamples = (M->want_profile && !c->no_zprofile) ? c->d_zsamples : nullptr;   // the profile costs ~3 %: only when a re-cut was asked for
        // cleared only on a frame that records into it: a frame between the profiled one and the re-cut (updateInterval > 1) must not wipe it
        hipError_t he = ho.zsamples ? hipMemsetAsync(c->d_zsamples, 0, (size_t)VPFX_ZPROF_COPIES * c->g.Nz * sizeof(unsigned), k.stream) : hipSuccess;
        r = he == hipSuccess ? launch_raymarch(c, kc, k.d_img[0], k.d_img[1], &ho) : vp_fail(c, VP_ERR_HIP, "hipMemsetAsync failed");
        c->d_scene_depth = keep;
        if (r) return r;       // (no vote here: with hand-off groups the later groups are still waiting for this rank's maps, step (3); an unvoted
                               //  failure ABORTS the context on the way out -- worker_main / run_all -- which wakes them)
        // (3) hand-off out: a phase-A-only slab behind this one is hidden by this slab's phase-A image only (t_out0), every other by both
        ops.clear();
        for (int p = my_pos + 1; p < world; ++p) {
            const int t = chain[p];
            if (group_of[t] > my_group) ops.push_back(P2P{true, t, is_a_only(M, t, zb) ? k.d_tout[0] : k.d_tout[1], M->npix});
        }
        r = p2p_batch(M, k, ops); if (r) return r;
        // (4) image exchange + ordered blend (VPR.cs:652-711 at slab granularity)
        VP_HIP(hipEventRecord(k.ev[2][0], k.stream));
        float* primary = k.d_img[M->cuts[k.rank] <= zb ? 0 : 1];          // first image in blend order: phase A if the slab has one
        float* second = k.d_img[1];                                       // the straddler's phase-B image
        // The schedule -- who sends which unit to whom, what is copied locally -- is the library's ONE definition of the exchange
        // (hl_exchange_plan = vp_exchange_plan, also executed over gloo on host buffers by tests/test_fanout_gloo.py); here its sends and
        // receives become grouped ncclSend / ncclRecv (or peer copies), its all-gather ncclAllGather, its copies stream copies.
        const size_t unit = (gather_all ? M->pixpad : M->piece) * 4;           // floats per unit: a whole padded image or one screen piece
        float* const xbuf[5] = {primary, second, k.d_pieces, k.d_piece_out, k.d_final};
        const void* images[VP_MAX_RANKS + 1];
        auto run_phase = [&](int phase) -> int {
            vp_xop xo[4 * VP_MAX_RANKS + 8];
            const int nx = hl_exchange_plan(world, k.rank, strad, gather_all ? 1 : 0, phase, xo, (int)(sizeof xo / sizeof xo[0]));
            if (nx > (int)(sizeof xo / sizeof xo[0])) return vp_fail(c, VP_ERR_STATE, "exchange plan of %d operations", nx);
            ops.clear();
            int rr = VP_OK;
            for (int i = 0; i < nx && !rr; ++i) {
                const vp_xop& o = xo[i];
                if (o.kind == VP_XOP_SEND || o.kind == VP_XOP_RECV) {
                    ops.push_back(P2P{o.kind == VP_XOP_SEND, o.peer, xbuf[o.buf] + (size_t)o.index * unit, unit * sizeof(float)});
                    continue;
                }
                rr = p2p_batch(M, k, ops); ops.clear();                       // a batch ends where the sends / receives end
                if (rr) break;
                if (o.kind == VP_XOP_COPY) rr = copy_on_stream(k, xbuf[o.dst_buf] + (size_t)o.dst_index * unit, xbuf[o.buf] + (size_t)o.index * unit, unit * sizeof(float));
                else if (o.kind == VP_XOP_ALL_GATHER) rr = all_gather_inplace(M, k, xbuf[o.buf], unit);
                else rr = vp_fail(c, VP_ERR_STATE, "exchange plan: unknown operation %d", o.kind);
            }
            return rr ? rr : p2p_batch(M, k, ops);
        };
        r = run_phase(0); if (r) return r;
        for (int i = 0; i < n_plan; ++i) images[i] = k.d_pieces + (size_t)(plan_which[i] ? world : plan_rank[i]) * unit;
        if (!gather_all) { r = launch_blend(c, images, plan_kind, n_plan, k.d_piece_out, M->piece); if (r) return r; }
        else if (k.rank == 0) { r = launch_blend(c, images, plan_kind, n_plan, k.d_final, M->npix); if (r) return r; }
        r = run_phase(1); if (r) return r;
        VP_HIP(hipEventRecord(k.ev[2][1], k.stream));
        k.ev_valid[2] = true;
        // (5) deliver on the display rank
        if (k.rank == 0) {
            if (d_out) VP_HIP(hipMemcpyAsync(d_out, k.d_final, M->npix * 4 * sizeof(float), hipMemcpyDeviceToDevice, k.stream));
            if (host_out) {
                // the bounded wait comes BEFORE the copy: a device-to-host copy into pageable memory blocks inside the runtime until the stream
                // has reached it -- for ever, if an exchange in front of it lost its peer (found with the checking RCCL stand-in, round 5: the
                // copy used to be queued first and swallowed the stall the time-out was there to catch)
                { const int rw = kid_wait(M, k, "image exchange"); if (rw) return rw; }
                VP_HIP(hipMemcpyAsync(host_out, k.d_final, M->npix * 4 * sizeof(float), hipMemcpyDeviceToHost, k.stream));
                return api_stream_sync(c);
            }
        }
        return VP_OK;
    });
    if (!rc && M->want_profile) { M->have_profile = true; M->need_plan = true; M->want_profile = false; }    // re-cut at the next bin
    return rc;
}

int multi_sync(vp_ctx* P)
{
    vp_multi* M = P->multi;
    return run_all(M, [&](Kid& k) -> int { const int rw = kid_wait(M, k, "vp_sync"); return rw ? rw : vp_sync(k.c); }, false);
}

int multi_set_occluders(vp_ctx* P, const vp_occluder* boxes, int32_t n)
{
    vp_multi* M = P->multi;
    return run_all(M, [&](Kid& k) -> int { const int rw = kid_wait(M, k, "vp_set_occluders (previous frame's exchanges)"); return rw ? rw : vp_set_occluders2(k.c, boxes, n); }, false);
}

vp_ctx* multi_owner_of_slice(vp_ctx* P, int zz)
{
    vp_multi* M = P->multi;
    if (zz == -2) return M->kids[0].c;
    if (zz == -1) { Kid* k = local_kid(M, 0); return k ? k->c : nullptr; }
    for (Kid& k : M->kids)
        if (zz >= M->cuts[k.rank] && zz < M->cuts[k.rank + 1]) return k.c;
    return nullptr;
}

int multi_get_stats(vp_ctx* P, vp_stats* st)
{
    vp_multi* M = P->multi;
    if (M->aborted.load()) return aborted_fail(M, P);          // (these touch the slab contexts' streams directly: never on an aborted fan-out)
    memset(st, 0, sizeof *st);
    for (Kid& k : M->kids) {
        vp_stats s;
        const int rc = vp_get_stats(k.c, &s);
        if (rc) { P->err = k.c->err; return rc; }
        st->particles = s.particles;
        st->occupied_mv += s.occupied_mv; st->pairs += s.pairs; st->voxels_filled += s.voxels_filled; st->samples += s.samples;
        st->brick_bytes += s.brick_bytes; st->bricks_sampled += s.bricks_sampled;
        st->max_pairs_per_mv = std::max(st->max_pairs_per_mv, s.max_pairs_per_mv);
        st->brick_bytes_per_voxel = s.brick_bytes_per_voxel; st->brick_format = s.brick_format;
    }
    return VP_OK;
}

int multi_last_kernel_ms(vp_ctx* P, int stage, float* ms)
{
    vp_multi* M = P->multi;
    if (M->aborted.load()) return aborted_fail(M, P);          // (these touch the slab contexts' streams directly: never on an aborted fan-out)
    float worst = 0.f;
    for (Kid& k : M->kids) {
        float v = 0.f;
        if (stage == 3 && k.rank == 0) continue;                // rank 0 runs the fused fill: no finish pass (0 ms)
        const int rc = vp_last_kernel_ms(k.c, stage, &v);
        if (rc) { P->err = k.c->err; return rc; }
        worst = std::max(worst, v);
    }
    *ms = worst;                                                // the frame waits for the slowest local rank
    return VP_OK;
}

int multi_read_bincounts(vp_ctx* P, int32_t* counts)
{
    vp_multi* M = P->multi;
    if (M->aborted.load()) return aborted_fail(M, P);          // (these touch the slab contexts' streams directly: never on an aborted fan-out)
    const vp_config& cfg = P->cfg;
    const size_t nxy = (size_t)cfg.num_mv[0] * cfg.num_mv[1], n3 = nxy * cfg.num_mv[2];
    memset(counts, 0, n3 * sizeof(int32_t));
    std::vector<int32_t> tmp(n3);
    for (Kid& k : M->kids) {                                    // every rank counts only its own slab: merge the local slabs
        const int rc = vp_read_bincounts(k.c, tmp.data());
        if (rc) { P->err = k.c->err; return rc; }
        const size_t a = (size_t)M->cuts[k.rank] * nxy, b = (size_t)M->cuts[k.rank + 1] * nxy;
        memcpy(counts + a, tmp.data() + a, (b - a) * sizeof(int32_t));
    }
    return VP_OK;
}

int multi_read_lightmap(vp_ctx* P, float* out)
{
    vp_multi* M = P->multi;
    if (M->aborted.load()) return aborted_fail(M, P);          // (these touch the slab contexts' streams directly: never on an aborted fan-out)
    Kid* last = local_kid(M, M->world - 1);                     // lightPropogationTex after the whole grid = the last slab's light map
    if (!last) return vp_fail(P, VP_ERR_STATE, "vp_read_lightmap: the last slab (rank %d) is not on this process", M->world - 1);
    const int rc = vp_read_lightmap(last->c, out);
    if (rc) P->err = last->c->err;
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------------------
VP_EXPORT int vp_rebalance(vp_ctx* c)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!c->multi) return VP_OK;                               // a single-device context has nothing to cut
    c->multi->want_profile = true;     // the next vp_raymarch also records where its samples fall; the vp_bin after it re-cuts the slabs
    return VP_OK;
}

VP_EXPORT int vp_rccl_unique_id(uint8_t out[128])
{
    vp_ctx* c = nullptr;
    if (!out) return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_rccl_unique_id: null output");
    if (!rccl().load()) return vp_fail(nullptr, VP_ERR_RCCL, "vp_rccl_unique_id: %s", rccl().err.c_str());
    ncclUniqueId id;
    VP_NCCL(rccl().GetUniqueId(&id));
    memcpy(out, &id, 128);
    return VP_OK;
}

VP_EXPORT int vp_get_multi_info(vp_ctx* P, vp_multi_info* out)
{
    if (!P) return VP_ERR_BAD_ARG;
    if (!out) return vp_fail(P, VP_ERR_BAD_ARG, "vp_get_multi_info: null output");
    memset(out, 0, sizeof *out);
    vp_multi* M = P->multi;
    if (!M) { out->world_size = 1; out->num_local = 1; out->rm_groups = 1; out->slab_cuts[1] = P->g.Nz; return VP_OK; }
    out->world_size = M->world; out->num_local = M->nlocal; out->first_rank = M->first_rank;
    out->rccl_ranks = M->use_rccl ? M->rccl_ranks : 0;
    out->exchange = (M->flags & VP_MULTI_EXCHANGE_ALL_GATHER) ? 1 : 0;
    out->rm_groups = M->groups;
    for (int i = 0; i <= M->world; ++i) out->slab_cuts[i] = M->cuts[i];
    for (int i = 0; i < M->world; ++i) { out->chain[i] = M->chain[i]; out->group_of[i] = M->group_of[i]; }
    for (Kid& k : M->kids) {
        vp_ctx* c = k.c;
        (void)hipSetDevice(k.device);
        { const int rw = kid_wait(M, k, "vp_get_multi_info"); if (rw) { P->err = c->err; return rw; } }
        unsigned long long s = 0;
        VP_HIP(hipMemcpy(&s, c->d_samples, sizeof s, hipMemcpyDeviceToHost));
        out->samples[k.rank] = (int64_t)s;
        for (int st = 0; st < 4; ++st) {
            float ms = 0.f;
            if (c->ev_valid[st] && hipEventElapsedTime(&ms, c->ev[st][0], c->ev[st][1]) == hipSuccess) out->stage_ms[k.rank][st] = ms;
        }
        if (k.index == 0)
            for (int e = 0; e < 3; ++e) {
                float ms = 0.f;
                if (k.ev_valid[e] && hipEventElapsedTime(&ms, k.ev[e][0], k.ev[e][1]) == hipSuccess) out->exchange_ms[e] = ms;
            }
    }
    return VP_OK;
}
