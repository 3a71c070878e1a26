// vstab_traj.hip -- F7+F8: camera path (prefix sum), box-filter smoothing and strength blend, fp64.
//
// Replaces nodes/video_stabilizer_flow.py:356-371 and nodes/stabilizer_utils.py:361-383
// (_smooth_path: symmetric moving average, window derived from fps, edge padding).  The work is
// O(N*P*window) on a few thousand doubles: one block, one lane per (column, output row); the
// prefix sum is the sequential recurrence of the reference (path[i] = path[i-1] + delta[i]).
#include "vstab_internal.h"
#include <cmath>

namespace {

__global__ __launch_bounds__(256) void trajectory_kernel(const double* __restrict__ deltas, double* __restrict__ path,
                                                         double* __restrict__ target, int n, int p, int window, int do_smooth,
                                                         double strength, int camera_lock)
{
    // phase 1: sequential prefix sum per column
    if (threadIdx.x < p) {
        const int c = threadIdx.x;
        double acc = 0.0;
        path[c] = 0.0;
        for (int i = 1; i < n; i++) {
            acc = acc + deltas[(size_t)(i - 1) * p + c];
            path[(size_t)i * p + c] = acc;
        }
    }
    __syncthreads();
    // phase 2: moving average over the edge-padded series, then blend
    const int pad = window / 2;
    const double kv = 1.0 / (double)window;
    for (int t = threadIdx.x; t < n * p; t += blockDim.x) {
        const int i = t / p, c = t - i * p;
        const double cur = path[t];
        double sm = cur;
        if (do_smooth) {
            double acc = 0.0;
            for (int k = 0; k < window; k++) {
                int s = i + k - pad;
                s = s < 0 ? 0 : (s > n - 1 ? n - 1 : s);
                acc += path[(size_t)s * p + c] * kv;
            }
            sm = acc;
        }
        target[t] = camera_lock ? 0.0 : cur + strength * (sm - cur);
    }
}

// Same arithmetic with the path held in LDS (n*p doubles): the deltas arrive with one coalesced pass, the sequential
// prefix sums (one lane per parameter, fixed order) run on LDS latency instead of a global load/store per frame, and
// the window sums read LDS.  Used whenever the path fits (multi-GPU runs smooth the whole clip on every rank).
constexpr int TRAJ_T = 1024;
constexpr size_t TRAJ_LDS_MAX = 144 * 1024;

__global__ __launch_bounds__(TRAJ_T) void trajectory_lds_kernel(const double* __restrict__ deltas, double* __restrict__ path,
                                                                double* __restrict__ target, int n, int p, int window, int do_smooth,
                                                                double strength, int camera_lock)
{
    extern __shared__ double s_path[];
    const int total = n * p;
    for (int t = threadIdx.x; t < total; t += TRAJ_T) s_path[t] = t < p ? 0.0 : deltas[t - p];
    __syncthreads();
    if ((int)threadIdx.x < p) {
        const int c = threadIdx.x;
        double acc = 0.0;
#pragma unroll 8
        for (int i = 1; i < n; i++) {
            acc = acc + s_path[i * p + c];
            s_path[i * p + c] = acc;
        }
    }
    __syncthreads();
    const int pad = window / 2;
    const double kv = 1.0 / (double)window;
    for (int t = threadIdx.x; t < total; t += TRAJ_T) {
        const int i = t / p, c = t - i * p;
        const double cur = s_path[t];
        double sm = cur;
        if (do_smooth) {
            double acc = 0.0;
            for (int k = 0; k < window; k++) {
                int s = i + k - pad;
                s = s < 0 ? 0 : (s > n - 1 ? n - 1 : s);
                acc += s_path[s * p + c] * kv;
            }
            sm = acc;
        }
        path[t] = cur;
        target[t] = camera_lock ? 0.0 : cur + strength * (sm - cur);
    }
}

}  // namespace

extern "C" int vstab_trajectory(vstab_ctx* ctx, const double* deltas, int n, int p, double smooth, double fps,
                                double strength, int camera_lock, double* path, double* target)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_trajectory: ctx is NULL");
    VSTAB_REQUIRE(deltas && path && target, "vstab_trajectory: NULL pointer argument");
    VSTAB_REQUIRE(n >= 2 && p >= 1 && p <= 8, "vstab_trajectory: unsupported shape n=%d p=%d", n, p);
    VSTAB_HIP(hipSetDevice(ctx->device));
    // stabilizer_utils.py:363-375
    smooth = smooth < 0.0 ? 0.0 : (smooth > 1.0 ? 1.0 : smooth);
    strength = strength < 0.0 ? 0.0 : (strength > 1.0 ? 1.0 : strength);
    const int do_smooth = !(smooth <= 0.0 || n <= 2);
    fps = fps > 1.0 ? fps : 1.0;
    const double min_seconds = 3.0 / 16.0, max_seconds = 13.0 / 16.0;
    const double window_seconds = min_seconds + smooth * (max_seconds - min_seconds);
    int window = (int)std::nearbyint(window_seconds * fps);  // Python round(): half to even
    window = window > 3 ? window : 3;
    if (window % 2 == 0) window += 1;

    const size_t in_bytes = sizeof(double) * (size_t)(n - 1) * p, out_bytes = sizeof(double) * (size_t)n * p;
    void* d_in = nullptr;
    if (vstab_stage_params(ctx, deltas, in_bytes, &d_in)) return 1;
    if (ctx->d_fit.reserve(2 * out_bytes + 512)) return 1;
    if (ctx->h_fit.reserve(2 * out_bytes)) return 1;
    double* d_path = static_cast<double*>(ctx->d_fit.ptr);
    double* d_target = d_path + (size_t)n * p;
    const size_t lds = out_bytes;
    if (lds <= TRAJ_LDS_MAX) {
        if (lds > 64 * 1024)
            VSTAB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trajectory_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(trajectory_lds_kernel, dim3(1), dim3(TRAJ_T), lds, ctx->stream, static_cast<const double*>(d_in), d_path, d_target, n,
                           p, window, do_smooth, strength, camera_lock);
    } else {
        hipLaunchKernelGGL(trajectory_kernel, dim3(1), dim3(256), 0, ctx->stream, static_cast<const double*>(d_in), d_path, d_target, n, p,
                           window, do_smooth, strength, camera_lock);
    }
    VSTAB_HIP(hipGetLastError());
    VSTAB_HIP(hipMemcpyAsync(ctx->h_fit.ptr, d_path, 2 * out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    VSTAB_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(path, ctx->h_fit.ptr, out_bytes);
    memcpy(target, static_cast<char*>(ctx->h_fit.ptr) + out_bytes, out_bytes);
    return 0;
}
