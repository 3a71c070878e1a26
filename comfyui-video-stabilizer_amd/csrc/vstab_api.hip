// vstab_api.hip -- context, error reporting and staging helpers of libvstab.so.
#include "vstab_internal.h"
#include <cmath>
#include <cstdarg>

static thread_local char g_err[1024] = "";

void vstab_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ScratchBuf::reserve(size_t need)
{
    if (need <= bytes) return 0;
    size_t want = need + need / 4 + 256;
    if (ptr) {
        if (pinned_host) (void)hipHostFree(ptr);
        else (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    hipError_t e = pinned_host ? hipHostMalloc(&ptr, want, hipHostMallocDefault) : hipMalloc(&ptr, want);
    if (e != hipSuccess) {
        vstab_set_error("scratch allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        ptr = nullptr;
        return 1;
    }
    bytes = want;
    return 0;
}

void ScratchBuf::release()
{
    if (ptr) {
        if (pinned_host) (void)hipHostFree(ptr);
        else (void)hipFree(ptr);
    }
    ptr = nullptr;
    bytes = 0;
}

bool vstab_invert3x3(const double* S, double* D) { return vstab_invert3x3_hd(S, D); }

int vstab_stage_params(vstab_ctx* ctx, const void* host, size_t bytes, void** dev_out)
{
    // the pinned buffer may still be the source of an in-flight copy from the previous call
    VSTAB_HIP(hipEventSynchronize(ctx->ev_params_free));
    if (ctx->h_params.reserve(bytes)) return 1;
    if (ctx->d_params.reserve(bytes)) return 1;
    memcpy(ctx->h_params.ptr, host, bytes);
    VSTAB_HIP(hipMemcpyAsync(ctx->d_params.ptr, ctx->h_params.ptr, bytes, hipMemcpyHostToDevice, ctx->stream));
    VSTAB_HIP(hipEventRecord(ctx->ev_params_free, ctx->stream));
    *dev_out = ctx->d_params.ptr;
    return 0;
}

EventPair* vstab_timer_slot(vstab_ctx* ctx, const char* kind)
{
    auto it = ctx->timers.find(kind);
    if (it == ctx->timers.end()) {
        EventPair ep;
        if (hipEventCreate(&ep.start) != hipSuccess || hipEventCreate(&ep.stop) != hipSuccess) return nullptr;
        it = ctx->timers.emplace(kind, ep).first;
    }
    return &it->second;
}

int vstab_timer_fold(EventPair* ev)
{
    if (!ev->pending) return 0;
    float ms = 0.f;
    VSTAB_HIP(hipEventSynchronize(ev->stop));
    VSTAB_HIP(hipEventElapsedTime(&ms, ev->start, ev->stop));
    ev->total_ms += ms;
    ev->launches += 1;
    ev->pending = false;
    return 0;
}

int vstab_check_device_status(vstab_ctx* ctx, const char* who)
{
    if (!ctx || !ctx->h_status) return 0;
    const int word = *ctx->h_status;
    if (word == 0) return 0;
    *ctx->h_status = 0;
    if (word & VSTAB_STATUS_PIS_TIMEOUT)
        vstab_set_error("%s: the DIS patch inverse search reported an expired dependency wait (status 0x%x): a wave did not see "
                        "its neighbour row's progress counter advance within the spin bound, the flow of this call is invalid",
                        who, word);
    else
        vstab_set_error("%s: device-side failure report, status 0x%x", who, word);
    return 3;
}

extern "C" {

int vstab_abi_version(void) { return VSTAB_ABI_VERSION; }

int vstab_test_hooks(void)
{
#ifdef VSTAB_TEST_HOOKS
    return 1;
#else
    return 0;
#endif
}

const char* vstab_last_error(void) { return g_err; }

int vstab_create(vstab_ctx** out, int device)
{
    VSTAB_REQUIRE(out != nullptr, "vstab_create: out is NULL");
    int count = 0;
    VSTAB_HIP(hipGetDeviceCount(&count));
    VSTAB_REQUIRE(count > 0, "vstab_create: no HIP device visible");
    if (device < 0) VSTAB_HIP(hipGetDevice(&device));
    VSTAB_REQUIRE(device < count, "vstab_create: device %d out of range (%d visible)", device, count);
    VSTAB_HIP(hipSetDevice(device));
    vstab_ctx* ctx = new vstab_ctx();
    ctx->device = device;
    ctx->h_params.pinned_host = true;
    ctx->h_fit.pinned_host = true;
    ctx->h_xfer.pinned_host = true;
    VSTAB_HIP(hipEventCreateWithFlags(&ctx->ev_params_free, hipEventDisableTiming));
    VSTAB_HIP(hipEventRecord(ctx->ev_params_free, nullptr));
    {
        void* h = nullptr;
        VSTAB_HIP(hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent));
        memset(h, 0, 64);
        void* d = nullptr;
        VSTAB_HIP(hipHostGetDevicePointer(&d, h, 0));
        ctx->h_status = static_cast<volatile int*>(h);
        ctx->d_status = static_cast<int*>(d);
    }
    *out = ctx;
    return 0;
}

int vstab_destroy(vstab_ctx* ctx)
{
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->h_params.release();
    ctx->d_params.release();
    ctx->d_dis.release();
    ctx->d_fit.release();
    ctx->h_fit.release();
    ctx->d_gray_tmp.release();
    ctx->d_range.release();
    ctx->d_plan.release();
    ctx->h_plan.release();
    if (ctx->h_peaks) (void)hipHostFree(ctx->h_peaks);
    if (ctx->d_peaks_count) (void)hipFree(ctx->d_peaks_count);
    if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
    if (ctx->ev_fit_done) (void)hipEventDestroy(ctx->ev_fit_done);
    if (ctx->ev_plan_done) (void)hipEventDestroy(ctx->ev_plan_done);
    if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
    if (ctx->ev_side) (void)hipEventDestroy(ctx->ev_side);
    for (auto& kv : ctx->timers) { (void)hipEventDestroy(kv.second.start); (void)hipEventDestroy(kv.second.stop); }
    (void)hipEventDestroy(ctx->ev_params_free);
    if (ctx->h_status) (void)hipHostFree(const_cast<int*>(ctx->h_status));
    if (ctx->xfer_stream) { (void)hipStreamSynchronize(ctx->xfer_stream); (void)hipStreamDestroy(ctx->xfer_stream); }
    for (auto& ev : ctx->ev_xfer) if (ev) (void)hipEventDestroy(ev);
    if (ctx->ev_xfer_sync) (void)hipEventDestroy(ctx->ev_xfer_sync);
    ctx->h_xfer.release();
    ctx->d_xfer.release();
    if (ctx->prep_stream) { (void)hipStreamSynchronize(ctx->prep_stream); (void)hipStreamDestroy(ctx->prep_stream); }
    for (auto& ev : ctx->ev_prep) if (ev) (void)hipEventDestroy(ev);
    if (ctx->ev_pyramid) (void)hipEventDestroy(ctx->ev_pyramid);
    delete ctx;
    return 0;
}

int vstab_set_stream(vstab_ctx* ctx, void* hip_stream)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_set_stream: ctx is NULL");
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return 0;
}

int vstab_synchronize(vstab_ctx* ctx)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_synchronize: ctx is NULL");
    VSTAB_HIP(hipStreamSynchronize(ctx->stream));
    return vstab_check_device_status(ctx, "vstab_synchronize");
}


int vstab_set_timing(vstab_ctx* ctx, int enabled)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_set_timing: ctx is NULL");
    ctx->timing = enabled != 0;
    ctx->timing_detail = enabled == 2;
    ctx->timing_warp_only = enabled == 3;
    for (auto& kv : ctx->timers) { kv.second.pending = false; kv.second.total_ms = 0.0; kv.second.launches = 0; }
    return 0;
}

int vstab_last_kernel_ms(vstab_ctx* ctx, const char* kind, float* ms_out)
{
    VSTAB_REQUIRE(ctx != nullptr && kind != nullptr && ms_out != nullptr, "vstab_last_kernel_ms: NULL argument");
    auto it = ctx->timers.find(kind);
    VSTAB_REQUIRE(it != ctx->timers.end() && it->second.start != nullptr && (it->second.pending || it->second.launches > 0),
                  "vstab_last_kernel_ms: no timing recorded for '%s'", kind);
    VSTAB_HIP(hipEventSynchronize(it->second.stop));
    VSTAB_HIP(hipEventElapsedTime(ms_out, it->second.start, it->second.stop));
    return 0;
}

int vstab_kernel_ms_stats(vstab_ctx* ctx, const char* kind, double* total_ms, int* launches)
{
    VSTAB_REQUIRE(ctx != nullptr && kind != nullptr && total_ms != nullptr && launches != nullptr, "vstab_kernel_ms_stats: NULL argument");
    auto it = ctx->timers.find(kind);
    VSTAB_REQUIRE(it != ctx->timers.end(), "vstab_kernel_ms_stats: no timing recorded for '%s'", kind);
    if (int rc = vstab_timer_fold(&it->second)) return rc;
    *total_ms = it->second.total_ms;
    *launches = it->second.launches;
    return 0;
}

// Element-wise libm calls for the host-side parameter maps (stabilizer_utils.py:300-358 uses Python's math.*,
// i.e. this process's libm): the same functions without 6 Python-level calls per frame.
int vstab_host_math(int op, const double* a, const double* b, int n, double* out)
{
    VSTAB_REQUIRE(a != nullptr && out != nullptr && n >= 0, "vstab_host_math: bad argument");
    VSTAB_REQUIRE(op >= VSTAB_HOST_SQRT && op <= VSTAB_HOST_SIN, "vstab_host_math: unknown op %d", op);
    VSTAB_REQUIRE(op != VSTAB_HOST_ATAN2 || b != nullptr, "vstab_host_math: atan2 needs two inputs");
    for (int i = 0; i < n; i++) {
        switch (op) {
        case VSTAB_HOST_SQRT: out[i] = std::sqrt(a[i]); break;
        case VSTAB_HOST_ATAN2: out[i] = std::atan2(a[i], b[i]); break;
        case VSTAB_HOST_LOG: out[i] = std::log(a[i]); break;
        case VSTAB_HOST_EXP: out[i] = std::exp(a[i]); break;
        case VSTAB_HOST_COS: out[i] = std::cos(a[i]); break;
        default: out[i] = std::sin(a[i]); break;
        }
    }
    return 0;
}

// flow.py:340-346 for a whole clip in one call: _rescale_transform_to_full (stabilizer_utils.py:279-297: S^-1 M S in fp64,
// stored as f32; S diagonal, so every entry is the product chain fl(fl(up_i * M_ij) * down_j)) followed by
// _matrix_to_params (stabilizer_utils.py:300-324) on the f32 result -- its float32-scalar arithmetic (a*a + c*c, m00 - 1)
// in f32, its math.* calls through this process's libm, as the per-item form does.
int vstab_transitions_to_params(const float* work_mats, int count, int mode, const double* up, const double* down,
                                float* full_mats, double* params)
{
    VSTAB_REQUIRE(count >= 0 && (count == 0 || (work_mats && full_mats && params)), "vstab_transitions_to_params: bad argument");
    VSTAB_REQUIRE(mode >= VSTAB_MODE_TRANSLATION && mode <= VSTAB_MODE_PERSPECTIVE, "vstab_transitions_to_params: unknown mode %d", mode);
    VSTAB_REQUIRE((up == nullptr) == (down == nullptr), "vstab_transitions_to_params: up and down come together");
    for (int n = 0; n < count; n++) {
        const float* M = work_mats + (size_t)n * 9;
        float* F = full_mats + (size_t)n * 9;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                if (up) {
                    const double t = up[i] * (double)M[i * 3 + j];
                    F[i * 3 + j] = (float)(t * down[j]);
                } else {
                    F[i * 3 + j] = M[i * 3 + j];
                }
            }
        if (mode == VSTAB_MODE_TRANSLATION) {
            double* P = params + (size_t)n * 2;
            P[0] = (double)F[2]; P[1] = (double)F[5];
        } else if (mode == VSTAB_MODE_SIMILARITY) {
            double* P = params + (size_t)n * 4;
            const float a = F[0], c = F[3];
            const float aa = a * a, cc = c * c;
            const float sq = aa + cc;
            const double sq64 = (1e-10f > sq) ? 1e-10 : (double)sq;
            P[0] = (double)F[2]; P[1] = (double)F[5];
            P[2] = std::atan2((double)c, (double)a);
            P[3] = std::log(std::sqrt(sq64));
        } else {
            double* P = params + (size_t)n * 8;
            const float d0 = F[0] - 1.0f, d4 = F[4] - 1.0f;
            P[0] = (double)d0; P[1] = (double)F[1]; P[2] = (double)F[2]; P[3] = (double)F[3];
            P[4] = (double)d4; P[5] = (double)F[5]; P[6] = (double)F[6]; P[7] = (double)F[7];
        }
    }
    return 0;
}

// _params_to_matrix (stabilizer_utils.py:327-358) for a whole clip: fp64 arithmetic, f32 result.
int vstab_params_to_matrices(const double* params, int count, int mode, float* mats)
{
    VSTAB_REQUIRE(count >= 0 && (count == 0 || (params && mats)), "vstab_params_to_matrices: bad argument");
    VSTAB_REQUIRE(mode >= VSTAB_MODE_TRANSLATION && mode <= VSTAB_MODE_PERSPECTIVE, "vstab_params_to_matrices: unknown mode %d", mode);
    for (int n = 0; n < count; n++) {
        float* F = mats + (size_t)n * 9;
        if (mode == VSTAB_MODE_TRANSLATION) {
            const double* P = params + (size_t)n * 2;
            F[0] = 1.f; F[1] = 0.f; F[2] = (float)P[0];
            F[3] = 0.f; F[4] = 1.f; F[5] = (float)P[1];
            F[6] = 0.f; F[7] = 0.f; F[8] = 1.f;
        } else if (mode == VSTAB_MODE_SIMILARITY) {
            const double* P = params + (size_t)n * 4;
            const double s = std::exp(P[3]), ct = std::cos(P[2]), st = std::sin(P[2]);
            const double sc = s * ct, ss = s * st;
            F[0] = (float)sc; F[1] = (float)(-ss); F[2] = (float)P[0];
            F[3] = (float)ss; F[4] = (float)sc; F[5] = (float)P[1];
            F[6] = 0.f; F[7] = 0.f; F[8] = 1.f;
        } else {
            const double* P = params + (size_t)n * 8;
            F[0] = (float)(P[0] + 1.0); F[1] = (float)P[1]; F[2] = (float)P[2];
            F[3] = (float)P[3]; F[4] = (float)(P[4] + 1.0); F[5] = (float)P[5];
            F[6] = (float)P[6]; F[7] = (float)P[7]; F[8] = 1.f;
        }
    }
    return 0;
}

// _compute_bounding_boxes (stabilizer_utils.py:1010-1034) for a whole clip: the four corners (0,0) (w,0) (0,h) (w,h) through
// each matrix in fp64 (`m @ corners` with m cast to fp64: products in corner order, summed left to right, unfused),
// divided by the third row, min / max per axis with NumPy's NaN-propagating minimum / maximum.  Equal to THIS container's
// NumPy bit for bit (tests/test_abi_cpu.py); the reference's `matrix @ corners` goes through whatever BLAS its NumPy was
// built with, and a dgemm micro-kernel that fuses multiply-adds can differ from this in the last bit of the (w, h) corner:
// parity with the reference is BLAS-dependent to 1 ulp there (it enters the meta as min_content_ratio / safe_region_*).
// The end-to-end fixtures produced by the reference's own run (tests/test_e2e_golden_*.py) are the arbiter.
static inline double np_minimum(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }
static inline double np_maximum(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }

int vstab_bounding_boxes(const float* mats, int count, double width, double height, double* mins, double* maxs)
{
    VSTAB_REQUIRE(count >= 0 && (count == 0 || (mats && mins && maxs)), "vstab_bounding_boxes: bad argument");
    const double cx[4] = {0.0, width, 0.0, width}, cy[4] = {0.0, 0.0, height, height};
    for (int n = 0; n < count; n++) {
        const float* M = mats + (size_t)n * 9;
        double xs[4], ys[4];
        for (int j = 0; j < 4; j++) {
            double p[3];
            for (int i = 0; i < 3; i++) {
                const double a = (double)M[i * 3 + 0] * cx[j];
                const double b = (double)M[i * 3 + 1] * cy[j];
                const double c = (double)M[i * 3 + 2] * 1.0;
                p[i] = (a + b) + c;
            }
            xs[j] = p[0] / p[2];
            ys[j] = p[1] / p[2];
        }
        mins[(size_t)n * 2 + 0] = np_minimum(np_minimum(xs[0], xs[1]), np_minimum(xs[2], xs[3]));
        mins[(size_t)n * 2 + 1] = np_minimum(np_minimum(ys[0], ys[1]), np_minimum(ys[2], ys[3]));
        maxs[(size_t)n * 2 + 0] = np_maximum(np_maximum(xs[0], xs[1]), np_maximum(xs[2], xs[3]));
        maxs[(size_t)n * 2 + 1] = np_maximum(np_maximum(ys[0], ys[1]), np_maximum(ys[2], ys[3]));
    }
    return 0;
}

}  // extern "C"
