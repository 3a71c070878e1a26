// vstab_xfer.hip -- bulk host <-> device transfer for the node boundary (host code + the two byte <-> float kernels of the
// coded forms at the end of the file).
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
    // this member's share [lo, hi) of `units` items, in 64-item steps
    void share(int me, size_t units, size_t& lo, size_t& hi) const
    {
        const size_t part = (((units + members - 1) / members) + 63) & ~size_t(63);   // ceil: a tail shorter than the team is still covered
        lo = std::min(units, part * me);
        hi = std::min(units, part * (me + 1));
    }
    void copy_share(int me, char* dst, const char* src, size_t bytes) const
    {
        size_t lo, hi;
        share(me, bytes, lo, hi);
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

// ---- coded forms ---------------------------------------------------------------------------------------------------
// Both directions of the node boundary are PCIe-bound (55 GB/s against ~7 ms of GPU work per 256 x 1080p clip), so bytes
// that need not cross do not:
//  * a ComfyUI IMAGE decoded from 8-bit video holds float32(k) / float32(255), k = 0 .. 255, and nothing else
//    (`np.array(img).astype(np.float32) / 255.0`).  The staging threads look at every value they copy anyway: a chunk whose
//    values ALL have exactly the bits of such a quotient crosses as bytes (a quarter of the traffic) and is expanded by
//    a kernel with the same correctly rounded float32 division; the first chunk with any other value, and everything
//    behind it, crosses as float32 (vstab_upload).  The device tensor has the source's bits either way.  A uint8 clip
//    (nodes/stabilizer_utils.py:122-126) takes the same road without the test: its bytes ARE the k.
//  * the Flow node's padding mask is 0.0f or 1.0f per pixel (nodes/video_stabilizer_flow.py:583-586), Motion Apply's under
//    motion blur 1 - c / S for c = 0 .. S covered samples (nodes/motion_apply.py:195-199): a kernel packs the mask to bytes
//    and reports whether every value was one of those; if so the bytes cross and the host threads expand them, else the
//    plain download runs.
// vstab_codec.cpp (host loops, baseline + AVX2 builds of the same source):
// values -> bytes; false (output unspecified) unless every value has exactly the bits of float32(k) / 255.0f, k = 0 .. 255
extern "C" bool vstab_host_encode_q8(const float* src, unsigned char* dst, size_t n);
// bytes -> 0.0f / 1.0f
extern "C" void vstab_host_expand_mask(const unsigned char* src, float* dst, size_t n);
// bytes c -> lut[c]
extern "C" void vstab_host_expand_levels(const unsigned char* src, float* dst, size_t n, const float* lut);

namespace {

constexpr size_t QCHUNK = CHUNK;   // values per coded chunk: its byte form fills one ring slot (128 MB of float32 per chunk)

__global__ __launch_bounds__(256) void expand_q8_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, size_t n)
{
    const size_t vec = n / 16;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < vec; i += stride) {
        const uint4 q = reinterpret_cast<const uint4*>(src)[i];
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
        float4* o = reinterpret_cast<float4*>(dst) + 4 * i;
#pragma unroll
        for (int j = 0; j < 4; j++)
            o[j] = make_float4((float)(w[j] & 255u) / 255.0f, (float)((w[j] >> 8) & 255u) / 255.0f,
                               (float)((w[j] >> 16) & 255u) / 255.0f, (float)(w[j] >> 24) / 255.0f);
    }
    for (size_t i = vec * 16 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (float)src[i] / 255.0f;
}

// mask values -> bytes (0 / 1); *other is raised if a value is neither 0.0f nor 1.0f (by its bits: -0.0f counts as other)
__global__ __launch_bounds__(256) void pack_mask_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, size_t n, int* other)
{
    const size_t vec = n / 16;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned odd = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < vec; i += stride) {
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint4 f = reinterpret_cast<const uint4*>(src)[4 * i + j];
            const unsigned b[4] = {f.x, f.y, f.z, f.w};
            unsigned pack = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                odd |= (b[k] != 0u) & (b[k] != 0x3f800000u);
                pack |= (b[k] != 0u ? 1u : 0u) << (8 * k);
            }
            w[j] = pack;
        }
        reinterpret_cast<uint4*>(dst)[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    for (size_t i = vec * 16 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned b = reinterpret_cast<const unsigned*>(src)[i];
        odd |= (b != 0u) & (b != 0x3f800000u);
        dst[i] = b != 0u;
    }
    if (__ballot(odd != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr(other, 1);
}

// The motion-blur warp's mask is 1 - c / S for c = 0 .. S covered samples, values below 1e-3 set to 0 (vstab_warp.hip,
// nodes/motion_apply.py:195-199): S + 1 different floats.  value -> c; *other is raised if a value is not one of them by its bits.
__device__ __forceinline__ float level_value(int c, float fs)
{
    const float m = 1.0f - (float)c / fs;
    return (m < 1e-3f) ? 0.f : m;
}

__global__ __launch_bounds__(256) void pack_levels_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, size_t n, int levels, int* other)
{
    const float fs = (float)levels;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned odd = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n + 3) / 4; i += stride) {
        unsigned pack = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t j = 4 * i + k;
            const float m = j < n ? src[j] : 1.0f;
            const float t = (1.0f - m) * fs + 0.5f;
            int c = (t >= 0.f && t <= fs + 1.0f) ? (int)t : 0;     // NaN: 0, caught by the comparison below
            c = c > levels ? levels : c;
            odd |= __float_as_uint(level_value(c, fs)) != __float_as_uint(m);
            pack |= (unsigned)c << (8 * k);
        }
        reinterpret_cast<unsigned*>(dst)[i] = pack;       // (dst holds a multiple of 16 bytes)
    }
    if (__ballot(odd != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr(other, 1);
}

unsigned grid_for(size_t items16)
{
    const size_t blocks = (items16 + 255) / 256;
    return (unsigned)std::min<size_t>(std::max<size_t>(blocks, 1), 256 * 16);
}

}  // namespace

// `count` values to dev_dst as float32(k) / 255: from float32 values that are such quotients (SRC_FLOAT: checked value by
// value, stops at the first chunk that holds anything else) or from the bytes k themselves.  *chunks_done = leading chunks
// that crossed as bytes.
template <bool SRC_FLOAT>
static int upload_as_bytes(vstab_ctx* ctx, const void* host_src, float* dev_dst, size_t count, size_t* chunks_done, const char* who)
{
    Ring r;
    if (ring_get(ctx, r)) return 1;
    if (ctx->d_xfer.reserve(QCHUNK * SLOTS)) return 1;        // the byte chunks' landing slots on the device
    unsigned char* dslot = static_cast<unsigned char*>(ctx->d_xfer.ptr);
    if (!ctx->ev_xfer_sync) VSTAB_HIP(hipEventCreateWithFlags(&ctx->ev_xfer_sync, hipEventDisableTiming));
    VSTAB_HIP(hipEventRecord(ctx->ev_xfer_sync, ctx->stream));                 // as vstab_upload: dev_dst may still be in use on the stream
    VSTAB_HIP(hipStreamWaitEvent(ctx->xfer_stream, ctx->ev_xfer_sync, 0));
    const size_t chunks = (count + QCHUNK - 1) / QCHUNK;
    size_t coded = 0;                                         // written by member 0 only, read after the team has finished
    {
        Team team;
        std::atomic<int> chunk_bad[2];
        chunk_bad[0] = chunk_bad[1] = 0;
        auto body = [&](int me) {
            for (size_t c = 0; c < chunks; c++) {
                const int s = (int)(c % SLOTS);
                const size_t off = c * QCHUNK, len = std::min(QCHUNK, count - off);
                if (me == 0 && hipEventSynchronize(r.done[s]) != hipSuccess) team.failed = 1;   // the DMA that last read this slot
                team.sync();
                size_t lo, hi;
                team.share(me, len, lo, hi);
                unsigned char* slot = reinterpret_cast<unsigned char*>(r.slot[s]);
                if (hi > lo) {
                    if (SRC_FLOAT) {
                        if (!vstab_host_encode_q8(static_cast<const float*>(host_src) + off + lo, slot + lo, hi - lo)) chunk_bad[c & 1] = 1;
                    } else {
                        memcpy(slot + lo, static_cast<const unsigned char*>(host_src) + off + lo, hi - lo);
                    }
                }
                team.sync();
                if (chunk_bad[c & 1] != 0 || team.failed) return;   // every member takes the same decision from the same flags
                if (me == 0) {
                    chunk_bad[(c + 1) & 1] = 0;               // nobody touches it before the next chunk's first barrier
                    hipError_t e = hipMemcpyAsync(dslot + (size_t)s * QCHUNK, slot, len, hipMemcpyHostToDevice, ctx->xfer_stream);
                    if (e == hipSuccess) e = hipEventRecord(r.done[s], ctx->xfer_stream);
                    if (e == hipSuccess) {
                        // (the next copy into this device slot is queued behind the kernel on the same stream)
                        hipLaunchKernelGGL(expand_q8_kernel, dim3(grid_for(len / 16)), dim3(256), 0, ctx->xfer_stream,
                                           dslot + (size_t)s * QCHUNK, dev_dst + off, len);
                        e = hipGetLastError();
                    }
                    if (e != hipSuccess) team.failed = 1;     // (seen by the others at the next chunk's second barrier at the latest)
                    else coded = c + 1;
                }
            }
        };
        team.run(count < (size_t(1) << 20) ? 1 : xfer_threads(), body);
        VSTAB_REQUIRE(!team.failed, "%s: a HIP call failed: %s", who, hipGetErrorString(hipGetLastError()));
    }
    // the last chunk's expansion is the last thing on the copy stream: later work on the context's stream waits for it
    VSTAB_HIP(hipEventRecord(ctx->ev_xfer_sync, ctx->xfer_stream));
    VSTAB_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_xfer_sync, 0));
    *chunks_done = coded;
    return 0;
}

extern "C" int vstab_upload_f32_coded(vstab_ctx* ctx, const float* host_src, float* dev_dst, size_t count, size_t* coded_chunks)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_upload_f32_coded: ctx is NULL");
    VSTAB_REQUIRE(count == 0 || (host_src && dev_dst), "vstab_upload_f32_coded: NULL pointer argument");
    if (coded_chunks) *coded_chunks = 0;
    if (count == 0) return 0;
    VSTAB_REQUIRE((reinterpret_cast<uintptr_t>(dev_dst) & 15) == 0, "vstab_upload_f32_coded: dev_dst must be 16-byte aligned");
    VSTAB_HIP(hipSetDevice(ctx->device));
    size_t coded = 0;
    if (int rc = upload_as_bytes<true>(ctx, host_src, dev_dst, count, &coded, "vstab_upload_f32_coded")) return rc;
    if (coded_chunks) *coded_chunks = coded;
    // a chunk with a value of another kind: it and everything behind it cross as float32 (the ring's slots are guarded by their events)
    const size_t done = coded * QCHUNK;
    if (done < count) return vstab_upload(ctx, host_src + done, dev_dst + done, (count - done) * sizeof(float));
    return 0;
}

extern "C" int vstab_upload_u8_as_f32(vstab_ctx* ctx, const unsigned char* host_src, float* dev_dst, size_t count)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_upload_u8_as_f32: ctx is NULL");
    VSTAB_REQUIRE(count == 0 || (host_src && dev_dst), "vstab_upload_u8_as_f32: NULL pointer argument");
    if (count == 0) return 0;
    VSTAB_REQUIRE((reinterpret_cast<uintptr_t>(dev_dst) & 15) == 0, "vstab_upload_u8_as_f32: dev_dst must be 16-byte aligned");
    VSTAB_HIP(hipSetDevice(ctx->device));
    size_t done = 0;
    return upload_as_bytes<false>(ctx, host_src, dev_dst, count, &done, "vstab_upload_u8_as_f32");
}

static int download_mask_levels(vstab_ctx* ctx, const float* dev_src, float* host_dst, size_t count, int levels, int* coded);

extern "C" int vstab_download_mask_coded(vstab_ctx* ctx, const float* dev_src, float* host_dst, size_t count, int* coded)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_download_mask_coded: ctx is NULL");
    VSTAB_REQUIRE(count == 0 || (dev_src && host_dst), "vstab_download_mask_coded: NULL pointer argument");
    return download_mask_levels(ctx, dev_src, host_dst, count, 1, coded);
}

extern "C" int vstab_download_mask_levels(vstab_ctx* ctx, const float* dev_src, float* host_dst, size_t count, int levels, int* coded)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_download_mask_levels: ctx is NULL");
    VSTAB_REQUIRE(count == 0 || (dev_src && host_dst), "vstab_download_mask_levels: NULL pointer argument");
    VSTAB_REQUIRE(levels >= 1 && levels <= 255, "vstab_download_mask_levels: levels=%d (1 .. 255)", levels);
    return download_mask_levels(ctx, dev_src, host_dst, count, levels, coded);
}

static int download_mask_levels(vstab_ctx* ctx, const float* dev_src, float* host_dst, size_t count, int levels, int* coded)
{
    if (coded) *coded = 0;
    if (count == 0) return 0;
    VSTAB_HIP(hipSetDevice(ctx->device));
    const bool aligned = (reinterpret_cast<uintptr_t>(dev_src) & 15) == 0;
    if (!aligned || count < (size_t(1) << 20)) return vstab_download(ctx, dev_src, host_dst, count * sizeof(float));
    Ring r;
    if (ring_get(ctx, r)) return 1;
    const size_t packed_bytes = (count + 15) & ~size_t(15);
    if (ctx->d_xfer.reserve(packed_bytes + 16)) return 1;
    unsigned char* packed = static_cast<unsigned char*>(ctx->d_xfer.ptr);
    volatile int* other = ctx->h_status + VSTAB_XFER_OTHER_WORD;   // coherent host word, device-visible
    int* d_other = ctx->d_status + VSTAB_XFER_OTHER_WORD;
    // d_xfer may still be the source of an upload's expansion on the copy stream: the pack kernel runs behind it
    if (!ctx->ev_xfer_sync) VSTAB_HIP(hipEventCreateWithFlags(&ctx->ev_xfer_sync, hipEventDisableTiming));
    VSTAB_HIP(hipEventRecord(ctx->ev_xfer_sync, ctx->xfer_stream));
    VSTAB_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_xfer_sync, 0));
    *other = 0;
    float lut[256];
    for (int c = 0; c < 256; c++) {       // the host's float32 arithmetic of level_value (IEEE division and subtraction on both sides)
        const float m = 1.0f - (float)(c <= levels ? c : levels) / (float)levels;
        lut[c] = (m < 1e-3f) ? 0.f : m;
    }
    if (levels == 1) hipLaunchKernelGGL(pack_mask_kernel, dim3(grid_for(count / 16)), dim3(256), 0, ctx->stream, dev_src, packed, count, d_other);
    else hipLaunchKernelGGL(pack_levels_kernel, dim3(grid_for(count / 16)), dim3(256), 0, ctx->stream, dev_src, packed, count, levels, d_other);
    VSTAB_HIP(hipGetLastError());
    VSTAB_HIP(hipStreamSynchronize(ctx->stream));
    if (vstab_check_device_status(ctx, "vstab_download_mask_coded")) return 3;
    if (*other != 0) return vstab_download(ctx, dev_src, host_dst, count * sizeof(float));
    // the bytes through the ring, expanded by the team: one ring slot of bytes becomes four slots' worth of floats
    const size_t chunks = (count + CHUNK - 1) / CHUNK;
    Team team;
    if (!(getenv("VSTAB_XFER_THP") && atoi(getenv("VSTAB_XFER_THP")) == 0)) {
        const uintptr_t huge = uintptr_t(2) << 20;
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(host_dst) + huge - 1) & ~(huge - 1);
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(host_dst) + count * sizeof(float)) & ~(huge - 1);
        if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);
    }
    auto issue = [&](size_t c) {
        const int s = (int)(c % SLOTS);
        const size_t off = c * CHUNK, len = std::min(CHUNK, count - off);
        if (hipMemcpyAsync(r.slot[s], packed + off, len, hipMemcpyDeviceToHost, ctx->xfer_stream) != hipSuccess ||
            hipEventRecord(r.done[s], ctx->xfer_stream) != hipSuccess)
            team.failed = 1;
    };
    const size_t ahead = std::min<size_t>(SLOTS - 1, chunks);
    for (size_t c = 0; c < ahead; c++) issue(c);
    auto body = [&](int me) {
        for (size_t c = 0; c < chunks; c++) {
            const int s = (int)(c % SLOTS);
            const size_t off = c * CHUNK, len = std::min(CHUNK, count - off);
            if (me == 0) {
                if (hipEventSynchronize(r.done[s]) != hipSuccess) team.failed = 1;
                if (c + ahead < chunks) issue(c + ahead);
            }
            team.sync();
            if (!team.failed) {
                size_t lo, hi;
                team.share(me, len, lo, hi);
                const unsigned char* q = reinterpret_cast<const unsigned char*>(r.slot[s]) + lo;
                if (hi > lo && levels == 1) vstab_host_expand_mask(q, host_dst + off + lo, hi - lo);
                else if (hi > lo) vstab_host_expand_levels(q, host_dst + off + lo, hi - lo, lut);
            }
            team.sync();
        }
    };
    team.run(xfer_threads(), body);
    VSTAB_REQUIRE(!team.failed, "vstab_download_mask_coded: a HIP call failed: %s", hipGetErrorString(hipGetLastError()));
    if (coded) *coded = 1;
    return 0;
}
