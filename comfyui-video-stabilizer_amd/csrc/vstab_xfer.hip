// vstab_xfer.hip -- bulk host <-> device transfer for the node boundary (host code only, no kernels).
//
// ComfyUI hands the nodes CPU tensors and expects CPU tensors back (nodes/stabilizer_utils.py:200-221): a 256 x 1080p
// clip is 6.37 GB in and 8.49 GB out, against ~10 ms of GPU work.  The caller's tensors are ordinary pageable memory;
// a plain hipMemcpy from / to pageable memory is staged by the runtime through one internal buffer on one thread
// (measured ~20 GB/s here, round 1: 760-890 ms per clip).  These two entry points run the same staging as a pipeline:
// a ring of page-locked buffers, several host threads doing the pageable <-> pinned copies (which also spreads the
// first-touch page faults of a freshly allocated output tensor), and the DMA engine moving the previous chunk
// meanwhile, so the transfer runs at min(host copy rate, PCIe rate).
#include "vstab_internal.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <pthread.h>
#include <sys/mman.h>
#include <system_error>
#include <thread>

namespace {

constexpr size_t CHUNK = size_t(32) << 20;   // bytes per ring slot
constexpr int SLOTS = 4;

int xfer_threads()
{
    int t = 16;
    if (const char* e = getenv("VSTAB_XFER_THREADS")) t = atoi(e);
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && t > hw) t = hw;
    return t < 1 ? 1 : t;
}

// A team of host threads that walks the chunks of one transfer together: member 0 (the caller's thread) makes the HIP
// calls, all members copy their 64-byte-aligned share of every chunk; two barrier waits per chunk.
//
// The helpers are started BEFORE the barrier is sized: a box that refuses a thread (process / thread limits) leaves a
// smaller team -- down to the caller alone -- instead of an exception escaping an extern "C" function with started
// helpers parked in a barrier nobody else will reach.
struct Team {
    int members = 1;
    pthread_barrier_t bar;
    bool bar_ready = false;
    std::atomic<int> failed{0};
    std::vector<std::thread> helpers;
    std::mutex gate_m;
    std::condition_variable gate_cv;
    bool go = false;

    Team() = default;
    Team(const Team&) = delete;
    ~Team()
    {
        release();
        for (auto& th : helpers) th.join();
        if (bar_ready) pthread_barrier_destroy(&bar);
    }
    void release()
    {
        {
            std::lock_guard<std::mutex> lk(gate_m);
            go = true;
        }
        gate_cv.notify_all();
    }
    // body(me) is run by every member; returns when all of them are done
    template <class Body>
    void run(int want, Body body)
    {
#ifdef VSTAB_TEST_HOOKS   // fault injector of the test build: pretend the box refuses every helper after the first k (k >= 0)
        const char* e = getenv("VSTAB_DEBUG_XFER_SPAWN_FAIL");
        const int refuse_after = e ? atoi(e) : -1;
#else
        const int refuse_after = -1;
#endif
        try {
            for (int t = 1; t < want; t++) {
                if (refuse_after >= 0 && t > refuse_after) throw std::system_error(EAGAIN, std::generic_category());
                helpers.emplace_back([this, body, t] {
                    std::unique_lock<std::mutex> lk(gate_m);
                    gate_cv.wait(lk, [this] { return go; });
                    lk.unlock();
                    if (t < members) body(t);
                });
                members = t + 1;
            }
        } catch (...) {
            // std::system_error (the box refused a thread) or std::bad_alloc: fewer helpers than asked for -- members counts
            // the ones that exist; nothing may propagate out of an extern "C" entry point
        }
        if (pthread_barrier_init(&bar, nullptr, (unsigned)members) != 0) {
            members = 1;   // no barrier: the caller's thread does all the copying, parked helpers fall through
        } else {
            bar_ready = true;
        }
        release();
        body(0);
        for (auto& th : helpers) th.join();
        helpers.clear();
    }
    void sync()
    {
        if (members > 1) pthread_barrier_wait(&bar);
    }
    void copy_share(int me, char* dst, const char* src, size_t bytes) const
    {
        const size_t part = (((bytes + members - 1) / members) + 63) & ~size_t(63);   // ceil: a tail shorter than the team is still copied
        const size_t lo = std::min(bytes, part * me), hi = std::min(bytes, part * (me + 1));
        if (hi > lo) memcpy(dst + lo, src + lo, hi - lo);
    }
};

struct Ring {
    char* slot[SLOTS] = {};
    hipEvent_t done[SLOTS] = {};
};

int ring_get(vstab_ctx* ctx, Ring& r)
{
    if (ctx->h_xfer.reserve(CHUNK * SLOTS)) return 1;
    for (int i = 0; i < SLOTS; i++) {
        r.slot[i] = static_cast<char*>(ctx->h_xfer.ptr) + CHUNK * i;
        if (!ctx->ev_xfer[i]) VSTAB_HIP(hipEventCreateWithFlags(&ctx->ev_xfer[i], hipEventDisableTiming));
        r.done[i] = ctx->ev_xfer[i];
    }
    if (!ctx->xfer_stream) VSTAB_HIP(hipStreamCreateWithFlags(&ctx->xfer_stream, hipStreamNonBlocking));
    return 0;
}

}  // namespace

extern "C" int vstab_upload(vstab_ctx* ctx, const void* host_src, void* dev_dst, size_t bytes)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_upload: ctx is NULL");
    VSTAB_REQUIRE(bytes == 0 || (host_src && dev_dst), "vstab_upload: NULL pointer argument");
    if (bytes == 0) return 0;
    VSTAB_HIP(hipSetDevice(ctx->device));
    Ring r;
    if (ring_get(ctx, r)) return 1;
    // dev_dst may be a block the caller's allocator handed back while its previous user is still queued on the
    // context's stream (e.g. the frames of an earlier clip that a running blur warp still reads): order the copy
    // stream behind everything launched so far, as vstab_download does
    if (!ctx->ev_xfer_sync) VSTAB_HIP(hipEventCreateWithFlags(&ctx->ev_xfer_sync, hipEventDisableTiming));
    VSTAB_HIP(hipEventRecord(ctx->ev_xfer_sync, ctx->stream));
    VSTAB_HIP(hipStreamWaitEvent(ctx->xfer_stream, ctx->ev_xfer_sync, 0));
    const char* src = static_cast<const char*>(host_src);
    char* dst = static_cast<char*>(dev_dst);
    const size_t chunks = (bytes + CHUNK - 1) / CHUNK;
    Team team;
    auto body = [&](int me) {
        for (size_t c = 0; c < chunks; c++) {
            const int s = (int)(c % SLOTS);
            const size_t off = c * CHUNK, len = std::min(CHUNK, bytes - off);
            // the DMA that last read this slot (also of an earlier call) has finished
            if (me == 0 && hipEventSynchronize(r.done[s]) != hipSuccess) team.failed = 1;
            team.sync();
            team.copy_share(me, r.slot[s], src + off, len);   // overlaps the DMA of the previous chunks
            team.sync();
            if (me == 0 && !team.failed) {
                if (hipMemcpyAsync(dst + off, r.slot[s], len, hipMemcpyHostToDevice, ctx->xfer_stream) != hipSuccess ||
                    hipEventRecord(r.done[s], ctx->xfer_stream) != hipSuccess)
                    team.failed = 1;
            }
        }
    };
    team.run(bytes < (size_t(4) << 20) ? 1 : xfer_threads(), body);
    VSTAB_REQUIRE(!team.failed, "vstab_upload: a HIP call failed: %s", hipGetErrorString(hipGetLastError()));
    // later work on the context's stream must see the data; the host may reuse / free host_src as soon as we return
    // (every byte has left it), and the ring is only touched again by a later call, which waits on these events
    const int last = (int)((chunks - 1) % SLOTS);
    VSTAB_HIP(hipStreamWaitEvent(ctx->stream, r.done[last], 0));
    return 0;
}

extern "C" int vstab_download(vstab_ctx* ctx, const void* dev_src, void* host_dst, size_t bytes)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_download: ctx is NULL");
    VSTAB_REQUIRE(bytes == 0 || (dev_src && host_dst), "vstab_download: NULL pointer argument");
    if (bytes == 0) return 0;
    VSTAB_HIP(hipSetDevice(ctx->device));
    Ring r;
    if (ring_get(ctx, r)) return 1;
    // the producer of dev_src ran on the context's stream: order the copy stream behind it
    if (!ctx->ev_xfer_sync) VSTAB_HIP(hipEventCreateWithFlags(&ctx->ev_xfer_sync, hipEventDisableTiming));
    VSTAB_HIP(hipEventRecord(ctx->ev_xfer_sync, ctx->stream));
    VSTAB_HIP(hipStreamWaitEvent(ctx->xfer_stream, ctx->ev_xfer_sync, 0));
    const char* src = static_cast<const char*>(dev_src);
    char* dst = static_cast<char*>(host_dst);
    const size_t chunks = (bytes + CHUNK - 1) / CHUNK;
    Team team;
    // A freshly allocated output tensor is untouched memory: every 4 KiB page faults on first write, which is what
    // bounds this copy (not PCIe).  Ask for transparent huge pages on the 2 MiB-aligned interior -- 512x fewer faults
    // where the kernel grants it, no effect otherwise.
    if (bytes >= (size_t(8) << 20) && !(getenv("VSTAB_XFER_THP") && atoi(getenv("VSTAB_XFER_THP")) == 0)) {
        const uintptr_t huge = uintptr_t(2) << 20;
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(dst) + huge - 1) & ~(huge - 1);
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(dst) + bytes) & ~(huge - 1);
        if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);
    }
    auto issue = [&](size_t c) {
        const int s = (int)(c % SLOTS);
        const size_t off = c * CHUNK, len = std::min(CHUNK, bytes - off);
        if (hipMemcpyAsync(r.slot[s], src + off, len, hipMemcpyDeviceToHost, ctx->xfer_stream) != hipSuccess ||
            hipEventRecord(r.done[s], ctx->xfer_stream) != hipSuccess)
            team.failed = 1;
    };
    const size_t ahead = std::min<size_t>(SLOTS - 1, chunks);   // DMAs in flight while the host drains a slot
    for (size_t c = 0; c < ahead; c++) issue(c);
    auto body = [&](int me) {
        for (size_t c = 0; c < chunks; c++) {
            const int s = (int)(c % SLOTS);
            const size_t off = c * CHUNK, len = std::min(CHUNK, bytes - off);
            if (me == 0) {
                if (hipEventSynchronize(r.done[s]) != hipSuccess) team.failed = 1;
                if (c + ahead < chunks) issue(c + ahead);   // its slot was drained in the previous round
            }
            team.sync();
            if (!team.failed) team.copy_share(me, dst + off, r.slot[s], len);
            team.sync();
        }
    };
    team.run(bytes < (size_t(4) << 20) ? 1 : xfer_threads(), body);
    VSTAB_REQUIRE(!team.failed, "vstab_download: a HIP call failed: %s", hipGetErrorString(hipGetLastError()));
    return vstab_check_device_status(ctx, "vstab_download");
}
