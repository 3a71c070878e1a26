// vstab_fit.hip -- F4+F5: model fit on the grid-sampled flow, one 256-thread block per frame pair.
//
// Replaces nodes/video_stabilizer_flow.py:141-210 of the reference: the stride-8 sampling of the
// dense flow, cv2.estimateAffinePartial2D (RANSAC 2.0 px / 2000 / 0.992 + 10 LM iterations),
// cv2.findHomography (RANSAC 2.5 px / 2000 / 0.992 + DLT on inliers + 10 LM iterations) and the
// per-axis median translation.
//
// RANSAC keeps OpenCV's sequential semantics (RNG seeded with 2^64-1, adaptive iteration count)
// but runs data-parallel: one lane replays the RNG and draws a batch of K minimal samples, K lanes
// build the K models, all 256 lanes score every model against every sample in one pass over the
// points (integer inlier counts -> order independent), then one lane replays OpenCV's
// "best so far / update niters" loop over the batch.  The final least-squares refit is the closed
// form of what the 10 LM iterations converge to (the similarity problem is linear): fp64 sums
// reduced in a fixed tree, 4x4 solve on one lane.
#include "vstab_internal.h"
#include <cmath>

namespace {

// One block per pair.  Every pass over the ~8 k samples is a loop of dependent loads (index map -> sample); with 256
// threads a SIMD held one wavefront and nothing hid that latency (0.21 ms per clip); 512 or 1024 threads: 0.16 ms
// (512 is the default: at 1024 the 128-register cap spills 560 B per thread into scratch -- 238 MB of HBM writes per
// launch in the PMC view -- for no gain)
// (tools/fit_phases.py: validity scan 32 us, RANSAC +32, least squares + residual +43, the two medians +44; what is
// left is the ~60 workgroup barriers between short passes and the single-lane sections).
#ifndef VSTAB_FIT_THREADS
#define VSTAB_FIT_THREADS 512
#endif
constexpr int FIT_THREADS = VSTAB_FIT_THREADS;
constexpr int FIT_WAVES = FIT_THREADS / 64;
static_assert(FIT_THREADS >= 256 && FIT_THREADS % 64 == 0 && FIT_THREADS <= 1024, "radix_select uses 256 histogram bins");
constexpr int RANSAC_BATCH = 16;   // typical clips converge in < 10 iterations; 16 keeps the scoring loop in registers
constexpr int MAX_SORT = 16384;

struct Rng { unsigned long long state; };
__device__ __forceinline__ unsigned rng_next(Rng& r)
{
    r.state = (unsigned long long)(unsigned)r.state * 4164903690ULL + (unsigned)(r.state >> 32);
    return (unsigned)r.state;
}
__device__ __forceinline__ int rng_uniform(Rng& r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (unsigned)(b - a) + a); }

__device__ int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > 2.2250738585072014e-308 ? 1. - p : 2.2250738585072014e-308;
    double denom = 1. - pow(1. - ep, (double)modelPoints);
    if (denom < 2.2250738585072014e-308) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)__builtin_rint(num / denom);
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch /* >= FIT_WAVES */)
{
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    T total = scratch[0];
#pragma unroll
    for (int i = 1; i < FIT_WAVES; i++) total += scratch[i];
    __syncthreads();
    return total;
}

struct FitArgs {
    const float* grid_flow;   // [pairs][gh][gw][2]
    int* vmap;                // [pairs][gh*gw] compacted index -> grid index
    vstab_fit_record* out;    // [pairs*3]
    int pairs, gh, gw, step, requested_mode;
    const int* counts;        // point-pair mode (gw == 0): detected features per pair, rows of `cap` entries
    int cap;
};

// gw > 0: stride-`step` grid over a flow field (F = [gh][gw][2] displacements, flow.py:141-147);
// gw == 0: explicit point pairs (F = [cap][4]: prev.x, prev.y, next.x, next.y; untracked points carry NaN)
__device__ __forceinline__ void load_point(const float* __restrict__ F, int gw, int step, int g, float& px, float& py, float& cx, float& cy)
{
    if (gw == 0) {
        const float4 v = *reinterpret_cast<const float4*>(F + (size_t)g * 4);
        px = v.x; py = v.y; cx = v.z; cy = v.w;
        return;
    }
    const int gy = g / gw, gx = g - gy * gw;
    px = (float)(gx * step);
    py = (float)(gy * step);
    cx = px + F[(size_t)g * 2];
    cy = py + F[(size_t)g * 2 + 1];
}

__device__ void identity_record(vstab_fit_record& r, int nv, int total)
{
    for (int i = 0; i < 9; i++) r.matrix[i] = (i % 4 == 0) ? 1.f : 0.f;
    r.confidence = 0.0; r.residual = 0.0; r.accepted = 0; r.computed = 0; r.valid_points = nv; r.total_points = total;
}

// Order-preserving map float -> unsigned (and back): np.median's ordering on finite values.
__device__ __forceinline__ unsigned sort_key(float v)
{
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_value(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Exact selection of the element of ascending rank r among keys[0..n) (radix select, 4 passes of 8 bits):
// one LDS-atomic histogram of the still-matching keys per pass, wavefront 0 locates the bin that holds the rank.
// Replaces a full bitonic sort (0.62 of the kernel's 0.71 ms went there).  All threads must call; result block-uniform.
__device__ unsigned radix_select(const unsigned* keys, int n, int r, int* hist /*256*/, int* ctl /*2*/)
{
    unsigned prefix = 0, mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;   // 256 bins
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += FIT_THREADS) {
            const unsigned key = keys[k];
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            const int h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
            const int mine = h0 + h1 + h2 + h3;
            int incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d);
                if (l >= d) incl += o;
            }
            const int excl = incl - mine;
            if (r >= excl && r < incl) {   // exactly one lane
                int rr = r - excl, b = 4 * l;
                if (rr >= h0) { rr -= h0; b++; if (rr >= h1) { rr -= h1; b++; if (rr >= h2) { rr -= h2; b++; } } }
                ctl[0] = b; ctl[1] = rr;
            }
        }
        __syncthreads();
        prefix |= (unsigned)ctl[0] << shift;
        mask |= 0xffu << shift;
        r = ctl[1];
        __syncthreads();
    }
    return prefix;
}

__global__ __launch_bounds__(FIT_THREADS) void fit_kernel(FitArgs a)
{
    __shared__ unsigned s_sort[MAX_SORT];
    __shared__ int s_hist[256];
    __shared__ int s_sel[2];
    __shared__ double s_model[RANSAC_BATCH][6];
    __shared__ float s_modelf[RANSAC_BATCH][6];
    __shared__ int s_idx[RANSAC_BATCH][2];
    __shared__ int s_cnt[RANSAC_BATCH];
    __shared__ double s_red[FIT_WAVES];
    __shared__ int s_redi[FIT_WAVES];
    __shared__ int s_scan[FIT_THREADS];
    __shared__ double s_best[6];
    __shared__ int s_ctl[4];   // 0: done flag, 1: niters, 2: iter, 3: maxGood
    __shared__ Rng s_rng;

    const int pair = blockIdx.x, tid = threadIdx.x;
    const bool points = a.gw == 0;
    const int total = points ? a.counts[pair] : a.gh * a.gw;
    const float* __restrict__ F = a.grid_flow + (size_t)pair * a.cap * (points ? 4 : 2);
    int* __restrict__ vmap = a.vmap + (size_t)pair * a.cap;
    vstab_fit_record* out = a.out + (size_t)pair * 3;

    // ---- validity scan (ordered compaction, flow.py:150-152) ----
    const int per = (total + FIT_THREADS - 1) / FIT_THREADS;
    const int g0 = tid * per, g1 = min(g0 + per, total);
    int local = 0;
    for (int g = g0; g < g1; g++) {
        float px, py, cx, cy;
        load_point(F, a.gw, a.step, g, px, py, cx, cy);
        local += (isfinite(cx) && isfinite(cy)) ? 1 : 0;
    }
    {   // exclusive prefix sum of `local` over the block: wavefront scan, then the wavefront totals
        int incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if ((tid & 63) >= d) incl += o;
        }
        if ((tid & 63) == 63) s_redi[tid >> 6] = incl;
        __syncthreads();
        int before = 0;
        for (int i = 0; i < (tid >> 6); i++) before += s_redi[i];
        s_scan[tid] = before + incl - local;
        if (tid == FIT_THREADS - 1) s_sel[0] = before + incl;
    }
    __syncthreads();
    const int nv = s_sel[0];
    {
        int o = s_scan[tid];
        for (int g = g0; g < g1; g++) {
            float px, py, cx, cy;
            load_point(F, a.gw, a.step, g, px, py, cx, cy);
            if (isfinite(cx) && isfinite(cy)) vmap[o++] = g;
        }
    }
    __syncthreads();
    if (tid < 3) identity_record(out[tid], nv, total);
    if (points ? (total < 12 || nv < 8) : nv < 12) return;   // flow.py:153-154 / classic.py:84-86,102-103 (block-uniform)
    __threadfence_block();
    __syncthreads();

    // ---- similarity: RANSAC over 2-point samples (AffinePartial2DEstimatorCallback) ----
    if (a.requested_mode >= VSTAB_MODE_SIMILARITY && nv >= 3) {
        if (tid == 0) { s_rng.state = ~0ULL; s_ctl[0] = 0; s_ctl[1] = 2000; s_ctl[2] = 0; s_ctl[3] = 0; }
        __syncthreads();
        const float thr = (float)(2.0 * 2.0);
        while (true) {
            if (tid == 0) {
                Rng r = s_rng;
                for (int c = 0; c < RANSAC_BATCH; c++) {
                    const int i0 = rng_uniform(r, 0, nv);
                    int i1;
                    do { i1 = rng_uniform(r, 0, nv); } while (i1 == i0);
                    s_idx[c][0] = i0; s_idx[c][1] = i1;
                }
                s_rng = r;
            }
            __syncthreads();
            if (tid < RANSAC_BATCH) {
                float x1f, y1f, X1f, Y1f, x2f, y2f, X2f, Y2f;
                load_point(F, a.gw, a.step, vmap[s_idx[tid][0]], x1f, y1f, X1f, Y1f);
                load_point(F, a.gw, a.step, vmap[s_idx[tid][1]], x2f, y2f, X2f, Y2f);
                const double x1 = x1f, y1 = y1f, x2 = x2f, y2 = y2f, X1 = X1f, Y1 = Y1f, X2 = X2f, Y2 = Y2f;
                const double d = 1. / ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
                const double S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2));
                const double S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2));
                const double S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) - (X1 * x2 - X2 * x1) * (x1 - x2));
                const double S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) - (Y1 * y2 - Y2 * y1) * (y1 - y2));
                const double M[6] = {S0, -S1, S2, S1, S0, S3};
                for (int k = 0; k < 6; k++) { s_model[tid][k] = M[k]; s_modelf[tid][k] = (float)M[k]; }
                s_cnt[tid] = 0;
            }
            __syncthreads();
            // score every model of the batch in one pass over the points
            int cnt[RANSAC_BATCH];
#pragma unroll
            for (int c = 0; c < RANSAC_BATCH; c++) cnt[c] = 0;
            for (int k = tid; k < nv; k += FIT_THREADS) {
                float px, py, cx, cy;
                load_point(F, a.gw, a.step, vmap[k], px, py, cx, cy);
#pragma unroll
                for (int c = 0; c < RANSAC_BATCH; c++) {
                    const float ea = s_modelf[c][0] * px + s_modelf[c][1] * py + s_modelf[c][2] - cx;
                    const float eb = s_modelf[c][3] * px + s_modelf[c][4] * py + s_modelf[c][5] - cy;
                    cnt[c] += (ea * ea + eb * eb <= thr) ? 1 : 0;
                }
            }
#pragma unroll
            for (int c = 0; c < RANSAC_BATCH; c++) {
                int v = cnt[c];
#pragma unroll
                for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s);
                if ((tid & 63) == 0 && v) atomicAdd(&s_cnt[c], v);
            }
            __syncthreads();
            if (tid == 0) {
                int niters = s_ctl[1], iter = s_ctl[2], maxGood = s_ctl[3];
                int c = 0;
                for (; c < RANSAC_BATCH && iter < niters; c++, iter++) {
                    const int good = s_cnt[c];
                    if (good > (maxGood > 1 ? maxGood : 1)) {
                        for (int k = 0; k < 6; k++) s_best[k] = s_model[c][k];
                        maxGood = good;
                        niters = ransac_update_num_iters(0.992, (double)(nv - good) / nv, 2, niters);
                    }
                }
                s_ctl[1] = niters; s_ctl[2] = iter; s_ctl[3] = maxGood;
                s_ctl[0] = (iter >= niters) ? 1 : 0;
            }
            __syncthreads();
            if (s_ctl[0]) break;
        }
        const int maxGood = s_ctl[3];
        if (maxGood > 0) {
            // inliers of the best minimal-sample model -> closed-form least squares (what the LM refinement converges to)
            const float F0 = (float)s_best[0], F1 = (float)s_best[1], F2 = (float)s_best[2];
            const float F3 = (float)s_best[3], F4 = (float)s_best[4], F5 = (float)s_best[5];
            double sxx = 0, sx = 0, sy = 0, sX = 0, sY = 0, sxX = 0, syX = 0;
            int ninl = 0;
            for (int k = tid; k < nv; k += FIT_THREADS) {
                float px, py, cx, cy;
                load_point(F, a.gw, a.step, vmap[k], px, py, cx, cy);
                const float ea = F0 * px + F1 * py + F2 - cx;
                const float eb = F3 * px + F4 * py + F5 - cy;
                if (ea * ea + eb * eb <= thr) {
                    const double x = px, y = py, X = cx, Y = cy;
                    sxx += x * x + y * y; sx += x; sy += y; sX += X; sY += Y;
                    sxX += x * X + y * Y; syX += x * Y - y * X;
                    ninl++;
                }
            }
            sxx = block_sum(sxx, s_red); sx = block_sum(sx, s_red); sy = block_sum(sy, s_red);
            sX = block_sum(sX, s_red); sY = block_sum(sY, s_red); sxX = block_sum(sxX, s_red); syX = block_sum(syX, s_red);
            ninl = block_sum(ninl, s_redi);
            if (tid == 0) {
                // normal equations of [x -y 1 0; y x 0 1] (a, b, tx, ty) = (X, Y)
                double A[4][5] = {{sxx, 0, sx, sy, sxX}, {0, sxx, -sy, sx, syX}, {sx, -sy, (double)ninl, 0, sX}, {sy, sx, 0, (double)ninl, sY}};
                bool ok = true;
                for (int col = 0; col < 4 && ok; col++) {
                    int piv = col;
                    for (int r = col + 1; r < 4; r++) if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
                    if (fabs(A[piv][col]) < 1e-300) { ok = false; break; }
                    if (piv != col) for (int k = 0; k < 5; k++) { const double t = A[piv][k]; A[piv][k] = A[col][k]; A[col][k] = t; }
                    for (int r = col + 1; r < 4; r++) {
                        const double f = A[r][col] / A[col][col];
                        for (int k = col; k < 5; k++) A[r][k] -= f * A[col][k];
                    }
                }
                double sol[4] = {s_best[0], s_best[3], s_best[2], s_best[5]};
                if (ok && ninl > 0) {
                    for (int r = 3; r >= 0; r--) {
                        double v = A[r][4];
                        for (int k = r + 1; k < 4; k++) v -= A[r][k] * sol[k];
                        sol[r] = v / A[r][r];
                    }
                }
                s_best[0] = sol[0]; s_best[1] = -sol[1]; s_best[2] = sol[2];
                s_best[3] = sol[1]; s_best[4] = sol[0]; s_best[5] = sol[3];
            }
            __syncthreads();
            const double M0 = s_best[0], M1 = s_best[1], M2 = s_best[2], M3 = s_best[3], M4 = s_best[4], M5 = s_best[5];
            double res = 0;
            for (int k = tid; k < nv; k += FIT_THREADS) {
                float px, py, cx, cy;
                load_point(F, a.gw, a.step, vmap[k], px, py, cx, cy);
                const double x = px, y = py;
                res += fabs(x * M0 + y * M1 + M2 - (double)cx) + fabs(x * M3 + y * M4 + M5 - (double)cy);
            }
            res = block_sum(res, s_red);
            if (tid == 0) {
                vstab_fit_record& r = out[VSTAB_MODE_SIMILARITY];
                r.computed = 1;
                r.confidence = (double)maxGood / (double)nv;
                if (r.confidence >= 0.1) {
                    r.matrix[0] = (float)M0; r.matrix[1] = (float)M1; r.matrix[2] = (float)M2;
                    r.matrix[3] = (float)M3; r.matrix[4] = (float)M4; r.matrix[5] = (float)M5;
                    r.matrix[6] = 0.f; r.matrix[7] = 0.f; r.matrix[8] = 1.f;
                    r.residual = res / (2.0 * nv);
                    r.accepted = 1;
                }
            }
        } else if (tid == 0) {
            out[VSTAB_MODE_SIMILARITY].computed = 1;
        }
        __syncthreads();
    }

    // ---- translation: per-axis median of the shifts (flow.py:191-208) ----
    {
        float med[2];
        for (int axis = 0; axis < 2; axis++) {
            for (int k = tid; k < nv; k += FIT_THREADS) {
                float px, py, cx, cy;
                load_point(F, a.gw, a.step, vmap[k], px, py, cx, cy);
                s_sort[k] = sort_key(axis == 0 ? cx - px : cy - py);
            }
            __syncthreads();
            const int mid = nv / 2;
            const unsigned hi = radix_select(s_sort, nv, mid, s_hist, s_sel);
            if (nv & 1) {
                med[axis] = key_value(hi);
            } else {
                // lower middle element: the largest key below `hi` if exactly `mid` keys are smaller, else a duplicate of `hi`
                int less = 0;
                unsigned below = 0;
                for (int k = tid; k < nv; k += FIT_THREADS) {
                    const unsigned key = s_sort[k];
                    if (key < hi) { less++; below = key > below ? key : below; }
                }
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) {
                    less += __shfl_down(less, sft);
                    const unsigned o = __shfl_down(below, sft);
                    below = o > below ? o : below;
                }
                if ((tid & 63) == 0) { s_redi[tid >> 6] = less; s_hist[tid >> 6] = (int)below; }
                __syncthreads();
                int total_less = 0;
                unsigned mx = 0;
                for (int i = 0; i < FIT_WAVES; i++) { total_less += s_redi[i]; mx = (unsigned)s_hist[i] > mx ? (unsigned)s_hist[i] : mx; }
                const unsigned lo = total_less == mid ? mx : hi;
                med[axis] = (key_value(lo) + key_value(hi)) / 2.0f;
            }
            __syncthreads();
        }
        const float tx = med[0], ty = med[1];
        double res = 0;
        for (int k = tid; k < nv; k += FIT_THREADS) {
            float px, py, cx, cy;
            load_point(F, a.gw, a.step, vmap[k], px, py, cx, cy);
            res += (double)__builtin_fabsf((px + tx) - cx) + (double)__builtin_fabsf((py + ty) - cy);
        }
        res = block_sum(res, s_red);
        if (tid == 0) {
            vstab_fit_record& r = out[VSTAB_MODE_TRANSLATION];
            r.computed = 1; r.accepted = 1;
            r.matrix[2] = tx; r.matrix[5] = ty;
            r.confidence = (double)nv / (double)total;
            r.residual = res / (2.0 * nv);
        }
    }
}

}  // namespace

int vstab_fit_homography(vstab_ctx* ctx, const float* grid_flow, const int* vmap, int pairs, int gh, int gw, int step,
                         int cap, vstab_fit_record* d_out);

// results == nullptr: launch only -- the records stay on the device (vstab_fit_records_device), their download is queued
// behind the kernels with an event, and vstab_sample_fit_batch_end collects it (the speculative plan and the warp are
// queued in between, flow_pipeline.py).
static int run_fit(vstab_ctx* ctx, const float* data, const int* counts, int pairs, int gh, int gw, int step, int cap,
                   int requested_mode, vstab_fit_record* results)
{
    VSTAB_HIP(hipSetDevice(ctx->device));
    // a pending fit whose records were downloaded on the side stream but never collected (a call abandoned between _begin
    // and _end): that copy reads d_fit and has to be over before this call's kernels write it again
    if (ctx->fit_pairs_pending > 0 && ctx->fit_copy_bytes == 0 && ctx->ev_fit_done) VSTAB_HIP(hipEventSynchronize(ctx->ev_fit_done));
    ctx->fit_pairs_pending = 0; ctx->fit_copy_bytes = 0;
    const size_t rec_bytes = sizeof(vstab_fit_record) * (size_t)pairs * 3;
    const size_t map_bytes = sizeof(int) * (size_t)pairs * cap;
    if (ctx->d_fit.reserve(rec_bytes + map_bytes + 512)) return 1;
    if (ctx->h_fit.reserve(rec_bytes)) return 1;
    char* base = static_cast<char*>(ctx->d_fit.ptr);
    vstab_fit_record* d_out = reinterpret_cast<vstab_fit_record*>(base);
    int* d_map = reinterpret_cast<int*>(base + ((rec_bytes + 255) & ~size_t(255)));
    {
        KernelTimer timer(ctx, "fit");
        FitArgs a{data, d_map, d_out, pairs, gh, gw, step, requested_mode, counts, cap};
        hipLaunchKernelGGL(fit_kernel, dim3((unsigned)pairs), dim3(FIT_THREADS), 0, ctx->stream, a);
        VSTAB_HIP(hipGetLastError());
        if (requested_mode >= VSTAB_MODE_PERSPECTIVE) {
            if (int rc = vstab_fit_homography(ctx, data, d_map, pairs, gh, gw, step, cap, d_out)) return rc;
        }
    }
    if (results == nullptr) {
        // the records' download is NOT queued here: a copy on this stream would sit between the fit kernel and whatever the
        // caller queues next (the plan kernel, the warp).  vstab_flow_plan_device issues it on the side stream behind the plan
        // kernel; vstab_sample_fit_batch_end issues it itself if nobody has by then.
        if (!ctx->ev_fit_done) VSTAB_HIP(hipEventCreateWithFlags(&ctx->ev_fit_done, hipEventDisableTiming));
        ctx->fit_pairs_pending = pairs;
        ctx->fit_copy_bytes = rec_bytes;
        return 0;
    }
    VSTAB_HIP(hipMemcpyAsync(ctx->h_fit.ptr, d_out, rec_bytes, hipMemcpyDeviceToHost, ctx->stream));
    VSTAB_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(results, ctx->h_fit.ptr, rec_bytes);
    // the flow these fits were computed from was produced asynchronously by vstab_dis_flow_batch on the same stream:
    // this is the first host synchronisation after it, so a device-side failure report of DIS surfaces here
    return vstab_check_device_status(ctx, "vstab_sample_fit_batch");
}

extern "C" int vstab_sample_fit_batch(vstab_ctx* ctx, const float* grid_flow, int pairs, int gh, int gw, int step,
                                      int requested_mode, vstab_fit_record* results)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_sample_fit_batch: ctx is NULL");
    VSTAB_REQUIRE(grid_flow && results, "vstab_sample_fit_batch: NULL pointer argument");
    VSTAB_REQUIRE(pairs > 0 && gh > 0 && gw > 0 && step > 0, "vstab_sample_fit_batch: non-positive size");
    VSTAB_REQUIRE(requested_mode >= VSTAB_MODE_TRANSLATION && requested_mode <= VSTAB_MODE_PERSPECTIVE, "vstab_sample_fit_batch: unknown mode %d", requested_mode);
    VSTAB_REQUIRE((long long)gh * gw <= MAX_SORT, "vstab_sample_fit_batch: %d sample points exceed the supported %d", gh * gw, MAX_SORT);
    return run_fit(ctx, grid_flow, nullptr, pairs, gh, gw, step, gh * gw, requested_mode, results);
}

extern "C" int vstab_sample_fit_batch_begin(vstab_ctx* ctx, const float* grid_flow, int pairs, int gh, int gw, int step,
                                            int requested_mode)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_sample_fit_batch_begin: ctx is NULL");
    VSTAB_REQUIRE(grid_flow != nullptr, "vstab_sample_fit_batch_begin: NULL pointer argument");
    VSTAB_REQUIRE(pairs > 0 && gh > 0 && gw > 0 && step > 0, "vstab_sample_fit_batch_begin: non-positive size");
    VSTAB_REQUIRE(requested_mode >= VSTAB_MODE_TRANSLATION && requested_mode <= VSTAB_MODE_PERSPECTIVE, "vstab_sample_fit_batch_begin: unknown mode %d", requested_mode);
    VSTAB_REQUIRE((long long)gh * gw <= MAX_SORT, "vstab_sample_fit_batch_begin: %d sample points exceed the supported %d", gh * gw, MAX_SORT);
    return run_fit(ctx, grid_flow, nullptr, pairs, gh, gw, step, gh * gw, requested_mode, nullptr);
}

extern "C" const vstab_fit_record* vstab_fit_records_device(vstab_ctx* ctx)
{
    return (ctx && ctx->fit_pairs_pending > 0) ? static_cast<const vstab_fit_record*>(ctx->d_fit.ptr) : nullptr;
}

extern "C" int vstab_fit_records_copy(vstab_ctx* ctx, void* dst_dev, int pairs)
{
    VSTAB_REQUIRE(ctx != nullptr && dst_dev != nullptr, "vstab_fit_records_copy: NULL argument");
    VSTAB_REQUIRE(ctx->fit_pairs_pending > 0 && pairs == ctx->fit_pairs_pending, "vstab_fit_records_copy: no fit of %d pairs is pending", pairs);
    VSTAB_HIP(hipMemcpyAsync(dst_dev, ctx->d_fit.ptr, sizeof(vstab_fit_record) * (size_t)pairs * 3, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

extern "C" int vstab_sample_fit_batch_end(vstab_ctx* ctx, int pairs, vstab_fit_record* results)
{
    VSTAB_REQUIRE(ctx != nullptr && results != nullptr, "vstab_sample_fit_batch_end: NULL argument");
    VSTAB_REQUIRE(ctx->fit_pairs_pending > 0 && pairs == ctx->fit_pairs_pending, "vstab_sample_fit_batch_end: no fit of %d pairs is pending", pairs);
    if (ctx->fit_copy_bytes) {   // no plan call took the download with it: queue it behind the fit kernels now
        VSTAB_HIP(hipMemcpyAsync(ctx->h_fit.ptr, ctx->d_fit.ptr, ctx->fit_copy_bytes, hipMemcpyDeviceToHost, ctx->stream));
        VSTAB_HIP(hipEventRecord(ctx->ev_fit_done, ctx->stream));
        ctx->fit_copy_bytes = 0;
    }
    VSTAB_HIP(hipEventSynchronize(ctx->ev_fit_done));
    memcpy(results, ctx->h_fit.ptr, sizeof(vstab_fit_record) * (size_t)pairs * 3);
    ctx->fit_pairs_pending = 0;
    // the event lies behind the DIS kernels of the same stream: a device-side failure report of theirs has landed
    return vstab_check_device_status(ctx, "vstab_sample_fit_batch_end");
}

extern "C" int vstab_points_fit_batch(vstab_ctx* ctx, const float* point_pairs, const int* counts, int pairs, int max_points,
                                      int requested_mode, vstab_fit_record* results)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_points_fit_batch: ctx is NULL");
    VSTAB_REQUIRE(point_pairs && counts && results, "vstab_points_fit_batch: NULL pointer argument");
    VSTAB_REQUIRE(pairs > 0 && max_points > 0, "vstab_points_fit_batch: non-positive size");
    VSTAB_REQUIRE(requested_mode >= VSTAB_MODE_TRANSLATION && requested_mode <= VSTAB_MODE_PERSPECTIVE, "vstab_points_fit_batch: unknown mode %d", requested_mode);
    VSTAB_REQUIRE(max_points <= MAX_SORT, "vstab_points_fit_batch: %d points per pair exceed the supported %d", max_points, MAX_SORT);
    return run_fit(ctx, point_pairs, counts, pairs, 0, 0, 1, max_points, requested_mode, results);
}
