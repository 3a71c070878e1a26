// vstab_internal.h -- shared between the translation units of libvstab.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include "../../include/vstab.h"

void vstab_set_error(const char* fmt, ...);

#define VSTAB_HIP(call)                                                                  \
    do {                                                                                 \
        hipError_t _e = (call);                                                          \
        if (_e != hipSuccess) {                                                          \
            vstab_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e),       \
                            __FILE__, __LINE__);                                         \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

#define VSTAB_REQUIRE(cond, ...)                                                         \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            vstab_set_error(__VA_ARGS__);                                                \
            return 2;                                                                    \
        }                                                                                \
    } while (0)

// A grow-only device/pinned-host scratch buffer.
struct ScratchBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
    bool pinned_host = false;
    int reserve(size_t need);
    void release();
};

struct EventPair {
    hipEvent_t start = nullptr, stop = nullptr;
    bool pending = false;        // a start/stop pair has been recorded and not yet folded into the totals
    double total_ms = 0.0;       // folded launches since vstab_set_timing(ctx, 1)
    int launches = 0;
};
// Folds a finished, still pending measurement into the totals (blocks until the stop event has passed).
int vstab_timer_fold(EventPair* ev);

struct vstab_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool timing = false;
    bool timing_detail = false;                 // vstab_set_timing(ctx, 2): also events around the stages INSIDE a DIS call
    bool timing_warp_only = false;              // vstab_set_timing(ctx, 3): events around the warp launches only (the roofline kernel)
    bool dis_first_pair_is_clip_start = true;   // see vstab_dis_set_clip_start
    std::map<std::string, EventPair> timers;
    // staging for small per-call parameter tables (pinned host + device mirror)
    ScratchBuf h_params, d_params;
    hipEvent_t ev_params_free = nullptr;  // recorded after the H2D copy of h_params
    // DIS / fit workspaces (grow-only)
    ScratchBuf d_dis, d_fit, h_fit, d_gray_tmp, d_range;
    // Device-side failure reports: one host-resident word (coherent, device-mapped).  A kernel ORs a VSTAB_STATUS_*
    // bit into it through d_status; the host reads h_status after any stream synchronisation at no cost.
    volatile int* h_status = nullptr;
    int* d_status = nullptr;
    // F0's per-frame maxima as the gray pass reports them, mirrored into coherent host memory by the kernel that forms them
    // (frame_max_kernel) -- no copy and no event on the stream (those cost the stream ~25 us between the gray pass and the
    // pyramid, profiles/r05_dis_small_steps.md).  peaks_target counts the frames of all range passes since the context was created; the pass
    // that completes it writes that number into h_status[VSTAB_PEAKS_DONE_WORD] behind the values (vstab_last_frame_peaks polls it).
    float* h_peaks = nullptr;       // host pointer
    float* d_peaks_mirror = nullptr;   // the same memory as the device sees it
    int h_peaks_cap = 0, peaks_frames = 0;
    unsigned peaks_target = 0;
    unsigned* d_peaks_count = nullptr;   // device memory: frames finished since the context was created
    // The plain warp's per-frame padded-pixel counts reach the host the same way: a one-workgroup kernel behind the warp
    // copies them into coherent host memory and sets h_status[VSTAB_COUNTS_DONE_WORD] to the call's generation number, which
    // vstab_last_pad_counts polls -- the step's last host wait costs the flag's round trip instead of a copy + a stream
    // synchronisation (~30 us of GPU idle between two steps).
    unsigned* h_counts = nullptr;
    unsigned* d_counts_mirror = nullptr;
    int h_counts_cap = 0, counts_n = 0;
    unsigned counts_gen = 0;
    // bulk host <-> device transfers (vstab_xfer.hip): pinned ring, its events, a copy stream
    ScratchBuf h_xfer;
    ScratchBuf d_xfer;   // device side of the coded forms: landing slots of byte-coded upload chunks / the packed mask
    hipEvent_t ev_xfer[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_xfer_sync = nullptr;
    hipStream_t xfer_stream = nullptr;
    // speculative device-side plan (vstab_plan.hip): the plan's outputs on the device (final float32 matrices, warp
    // table, path / target) and their pinned host copy, an event behind each asynchronous download
    ScratchBuf d_plan, h_plan;
    hipEvent_t ev_fit_done = nullptr, ev_plan_done = nullptr;
    int fit_pairs_pending = 0, plan_frames = 0, plan_params = 0;
    // The downloads behind the speculative plan (fit records, plan results) run on a stream of their own, behind ONE event
    // recorded after plan_kernel: a copy queued on the call's stream sat between fit -> plan -> warp as two engine hand-overs
    // (19 + 27 us of idle GPU per C2 step, profiles/r05_dis_small_steps.md).  fit_copy_bytes > 0: the fit records'
    // download has not been issued yet (vstab_flow_plan_device issues it on the side stream, vstab_sample_fit_batch_end on
    // the call's stream if no plan call came in between).
    unsigned* plan_zero_ptr = nullptr;     // registered for the next plan kernel to zero (vstab_flow_plan_zero_counts)
    int plan_zero_n = 0;
    unsigned* plan_zeroed_ptr = nullptr;   // what the last plan kernel zeroed: the planned warp skips its own fill for it
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_side = nullptr;
    size_t fit_copy_bytes = 0;
    // DIS: the per-level image preparation (pad, gradients, structure tensor) runs on its own stream beside the
    // coarse-to-fine chain, one event per pyramid level (vstab_dis.hip)
    hipStream_t prep_stream = nullptr;
    hipEvent_t ev_prep[16] = {};   // MAX_LEVELS of vstab_dis.hip
    hipEvent_t ev_pyramid = nullptr;
};

enum { VSTAB_PEAKS_DONE_WORD = 4, VSTAB_COUNTS_DONE_WORD = 5, VSTAB_XFER_OTHER_WORD = 6 };      // index into h_status / d_status (the status word itself is [0])
enum { VSTAB_STATUS_PIS_TIMEOUT = 1 };   // DIS patch search: a bounded intra-workgroup dependency wait expired

// Call after a host synchronisation of ctx->stream: turns a device-side failure report into a non-zero return
// (rc 3) with vstab_last_error() set, and clears the word.
int vstab_check_device_status(vstab_ctx* ctx, const char* who);

// Upload `bytes` of host data through the pinned staging buffer; returns device pointer.
int vstab_stage_params(vstab_ctx* ctx, const void* host, size_t bytes, void** dev_out);

// Brackets the kernels of one API call with HIP events recorded on the call's stream (no host
// sync here: vstab_last_kernel_ms waits for the stop event when the number is asked for).
EventPair* vstab_timer_slot(vstab_ctx* ctx, const char* kind);

struct KernelTimer {
    vstab_ctx* ctx;
    EventPair* ev = nullptr;
    KernelTimer(vstab_ctx* c, const char* k) : ctx(c) {
        // (level 3: an event pair costs the stream ~10 us -- 0.05-0.1 ms over the four stages of a C2 step, measured -- so a
        // timed loop keeps them around the one kernel whose per-launch time it reports, profiles/r05_dis_small_steps.md)
        if (ctx->timing && (!ctx->timing_warp_only || strncmp(k, "warp", 4) == 0)) {
            ev = vstab_timer_slot(ctx, k);
            // the previous launch of this kind finished long ago (every pipeline pass synchronises with the host
            // between two launches of the same kind), so folding it costs no wait
            if (ev && ev->pending) (void)vstab_timer_fold(ev);
            if (ev) (void)hipEventRecord(ev->start, ctx->stream);
        }
    }
    ~KernelTimer() {
        if (ev) {
            (void)hipEventRecord(ev->stop, ctx->stream);
            ev->pending = true;
        }
    }
};

// The same around one stage inside a call, only under vstab_set_timing(ctx, 2): the events sit between dependent kernels,
// where they lengthen the chain a little, and a kind used twice in one call folds (waits for) its previous measurement --
// for a measurement pass of its own, never inside a timed loop.
struct DetailTimer {
    vstab_ctx* ctx;
    EventPair* ev = nullptr;
    DetailTimer(vstab_ctx* c, const char* k) : ctx(c) {
        if (ctx->timing && ctx->timing_detail) {
            ev = vstab_timer_slot(ctx, k);
            if (ev && ev->pending) (void)vstab_timer_fold(ev);
            if (ev) (void)hipEventRecord(ev->start, ctx->stream);
        }
    }
    ~DetailTimer() {
        if (ev) {
            (void)hipEventRecord(ev->stop, ctx->stream);
            ev->pending = true;
        }
    }
};

// cv::invert for a 3x3 CV_64F matrix (closed form); returns false (and zeros) if singular.  One definition for the host
// (vstab_warp_batch inverts the caller's matrices) and the device (plan_kernel writes the warp's table itself): IEEE fp64
// operations in one fixed order, no contraction (-ffp-contract=off), so both produce the same bits.
__host__ __device__ inline bool vstab_invert3x3_hd(const double* S, double* D)
{
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) +
               S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d == 0.0) {
        for (int i = 0; i < 9; i++) D[i] = 0.0;
        return false;
    }
    d = 1.0 / d;
    double t[9];
    t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    for (int i = 0; i < 9; i++) D[i] = t[i];
    return true;
}
bool vstab_invert3x3(const double* S, double* D);

// The warp kernels' per-(frame, sample) transform record and how it is filled from a float32 matrix
// (cv::warpPerspective: M.convertTo(CV_64F), invert).
struct WarpXform {
    double m[9];     // inverse (output -> source) matrix, as cv::warpPerspective builds it
    double wq;       // affine only: m8 ? 32/m8 : 0
    double wn;       // affine only: m8 ? 1/m8 : 0
    int affine;      // m6 == 0 && m7 == 0
    int pad_;
};
__host__ __device__ inline void vstab_fill_xform(const float* m32, WarpXform* xf)
{
    double M[9];
    for (int i = 0; i < 9; i++) M[i] = (double)m32[i];
    vstab_invert3x3_hd(M, xf->m);
    xf->affine = (xf->m[6] == 0.0 && xf->m[7] == 0.0) ? 1 : 0;
    const double W = xf->m[8];
    xf->wq = (W != 0.0) ? 32.0 / W : 0.0;
    xf->wn = (W != 0.0) ? 1.0 / W : 0.0;
    xf->pad_ = 0;
}
