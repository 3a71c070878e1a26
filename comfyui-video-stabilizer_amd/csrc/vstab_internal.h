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
    // bulk host <-> device transfers (vstab_xfer.hip): pinned ring, its events, a copy stream
    ScratchBuf h_xfer;
    hipEvent_t ev_xfer[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_xfer_sync = nullptr;
    hipStream_t xfer_stream = nullptr;
    // DIS: the per-level image preparation (pad, gradients, structure tensor) runs on its own stream beside the
    // coarse-to-fine chain, one event per pyramid level (vstab_dis.hip)
    hipStream_t prep_stream = nullptr;
    hipEvent_t ev_prep[16] = {};   // MAX_LEVELS of vstab_dis.hip
    hipEvent_t ev_pyramid = nullptr;
};

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
        if (ctx->timing) {
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

// cv::invert for a 3x3 CV_64F matrix (closed form); returns false (and zeros) if singular.
bool vstab_invert3x3(const double* S, double* D);
