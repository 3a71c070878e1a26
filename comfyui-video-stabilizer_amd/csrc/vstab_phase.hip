// vstab_phase.hip -- phase-correlation fallback estimator, gfx950.
//
// Replaces cv2.phaseCorrelate(prev.astype(float32), curr.astype(float32)) of
// nodes/video_stabilizer_flow.py:110-130 for every frame pair of a clip at once.  The reference reaches that
// call only when cv2.DISOpticalFlow cannot be created (flow.py:90-107); this build's DIS always exists, so the
// host selects this estimator only on request (VSTAB_FLOW_BACKEND=phase_correlate) -- it is here so that the
// `flow_backend == "phase_correlate"` branch of the node has the same GPU-resident implementation as the rest.
//
// Algorithm restated from OpenCV's published phasecorr.cpp (no window):
//   pad both images with zeros to getOptimalDFTSize (smallest 2^a 3^b 5^c >= size); forward real DFT;
//   P = F1 conj(F2) (float, products in double); |P|; C = P |P| / (|P|^2 + FLT_EPSILON), except the purely real
//   bins where OpenCV's packed-format helpers yield C = P / (P^2 + FLT_EPSILON); unscaled inverse DFT; circular shift
//   by (N/2, M/2); first maximum in raster order; 5 x 5 weighted centroid (clamped at the borders, double sums);
//   response = sum / (M N); shift = (N/2 - cx, M/2 - cy).
// The DFT itself is a mixed-radix (5, 3, 2) Stockham transform written here; oracle/vo_phase.c performs the same
// butterflies in the same order with the same twiddle table, so both sides are compared bit for bit.  Against a
// real OpenCV the DFT rounding differs (its factorisation is not restated): "parity unpinned", see DESIGN.md.
//
// Mapping to the machine: one workgroup per image row / spectrum column, the whole 1-D transform in LDS
// (two ping-pong arrays of N complex floats, N <= 2048), every butterfly computed by exactly one thread in a
// fixed operation order.  Each frame is transformed once and serves both pairs it belongs to.
#include "vstab_internal.h"
#include <cmath>
#include <cfloat>
#include <cstdlib>

namespace {

constexpr int PC_T = 256;
constexpr int PC_MAX_LEN = 2048;    // the estimation image is at most 960 px on its long side (stabilizer_utils.py:248-268)
constexpr int PC_MAX_STAGES = 11;   // 2^11 = 2048

struct FftPlan {
    int len;
    int stages;
    int radix[PC_MAX_STAGES];
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }

// One Stockham stage of radix R over `len` points: a -> b.  Every butterfly is computed by one thread, operations in
// the order of oracle/vo_phase.c (twiddle the inputs, then the R-point DFT as a running sum over q).
template <int R>
__device__ __forceinline__ void fft_stage(const float2* a, float2* b, int len, int ns, const float2* __restrict__ tw, bool inverse)
{
    const int m = len / R;
    const int tstep = len / (ns * R);
    for (int j = threadIdx.x; j < m; j += PC_T) {
        const int k = j % ns;
        float2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            v[q] = a[j + q * m];
            if (q) {
                float2 t = tw[q * k * tstep];
                if (inverse) t.y = -t.y;
                v[q] = cmul(v[q], t);
            }
        }
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int p = 0; p < R; ++p) {
            float2 acc = v[0];
#pragma unroll
            for (int q = 1; q < R; ++q) {
                float2 t = tw[((p * q) % R) * m];
                if (inverse) t.y = -t.y;
                acc = cadd(acc, cmul(v[q], t));
            }
            b[j0 + p * ns] = acc;
        }
    }
}

// In-LDS Stockham transform of `plan.len` points held in `a` (scratch `b`); returns the array holding the result.
// tw[t] = exp(-2 pi i t / len); `inverse` conjugates the twiddles (no scaling).
__device__ float2* fft_lds(float2* a, float2* b, const FftPlan& plan, const float2* __restrict__ tw, bool inverse)
{
    const int len = plan.len;
    int ns = 1;
    for (int s = 0; s < plan.stages; ++s) {
        const int r = plan.radix[s];
        __syncthreads();
        if (r == 2) fft_stage<2>(a, b, len, ns, tw, inverse);
        else if (r == 3) fft_stage<3>(a, b, len, ns, tw, inverse);
        else fft_stage<5>(a, b, len, ns, tw, inverse);
        ns *= r;
        float2* t = a; a = b; b = t;
    }
    __syncthreads();
    return a;
}

// rows of the zero-padded float image -> F[f][y][0..nh)
__global__ __launch_bounds__(PC_T) void pc_row_fwd_kernel(const uint8_t* __restrict__ gray, int h, int w, int M, int N, int nh,
                                                          FftPlan plan, const float2* __restrict__ tw, float2* __restrict__ F)
{
    extern __shared__ float2 lds[];
    const int y = blockIdx.x, f = blockIdx.y;
    float2* out = F + ((size_t)f * M + y) * nh;
    if (y >= h) {   // a zero row transforms to zeros
        for (int x = threadIdx.x; x < nh; x += PC_T) out[x] = make_float2(0.0f, 0.0f);
        return;
    }
    float2* a = lds;
    float2* b = lds + N;
    const uint8_t* row = gray + ((size_t)f * h + y) * w;
    for (int x = threadIdx.x; x < N; x += PC_T) a[x] = make_float2(x < w ? (float)row[x] : 0.0f, 0.0f);
    float2* res = fft_lds(a, b, plan, tw, false);
    for (int x = threadIdx.x; x < nh; x += PC_T) out[x] = res[x];
}

// in-place transform of column kx of F[f]
__global__ __launch_bounds__(PC_T) void pc_col_fwd_kernel(float2* __restrict__ F, int M, int nh, FftPlan plan, const float2* __restrict__ tw)
{
    extern __shared__ float2 lds[];
    const int kx = blockIdx.x, f = blockIdx.y;
    float2* col = F + (size_t)f * M * nh + kx;
    float2* a = lds;
    float2* b = lds + M;
    for (int y = threadIdx.x; y < M; y += PC_T) a[y] = col[(size_t)y * nh];
    float2* res = fft_lds(a, b, plan, tw, false);
    for (int y = threadIdx.x; y < M; y += PC_T) col[(size_t)y * nh] = res[y];
}

// normalised cross-power spectrum of pair p, column kx, then the inverse transform along y -> G[p][y][kx]
__global__ __launch_bounds__(PC_T) void pc_col_inv_kernel(const float2* __restrict__ F, int M, int N, int nh, FftPlan plan,
                                                          const float2* __restrict__ tw, float2* __restrict__ G)
{
    extern __shared__ float2 lds[];
    const int kx = blockIdx.x, p = blockIdx.y;
    const float2* c1 = F + (size_t)p * M * nh + kx;
    const float2* c2 = c1 + (size_t)M * nh;
    float2* a = lds;
    float2* b = lds + M;
    const bool real_col = kx == 0 || ((N & 1) == 0 && kx == N / 2);
    const float eps = FLT_EPSILON;
    for (int ky = threadIdx.x; ky < M; ky += PC_T) {
        const float2 u = c1[(size_t)ky * nh], v = c2[(size_t)ky * nh];
        float2 c;
        if (real_col && (ky == 0 || ((M & 1) == 0 && ky == M / 2))) {
            const float pr = u.x * v.x;
            const float pm = pr * pr;
            c = make_float2(pr / (pm + eps), 0.0f);
        } else {
            const float re = (float)((double)u.x * v.x + (double)u.y * v.y);
            const float im = (float)((double)u.y * v.x - (double)u.x * v.y);
            const float mag = (float)sqrt((double)re * re + (double)im * im);
            const double denom = (double)mag * mag + (double)eps;
            c = make_float2((float)(((double)re * mag) / denom), (float)(((double)im * mag) / denom));
        }
        a[ky] = c;
    }
    float2* res = fft_lds(a, b, plan, tw, true);
    float2* out = G + (size_t)p * M * nh + kx;
    for (int y = threadIdx.x; y < M; y += PC_T) out[(size_t)y * nh] = res[y];
}

struct RowBest {
    float value;
    int xs;   // column in the shifted array
};

// inverse transform along x of row y of G[p] (Hermitian-extended) -> real R[p][y][0..N), plus the row's first maximum
// in shifted column order
__global__ __launch_bounds__(PC_T) void pc_row_inv_kernel(const float2* __restrict__ G, int M, int N, int nh, FftPlan plan,
                                                          const float2* __restrict__ tw, float* __restrict__ R, RowBest* __restrict__ best)
{
    extern __shared__ float2 lds[];
    __shared__ float s_val[PC_T];
    __shared__ int s_xs[PC_T];
    const int y = blockIdx.x, p = blockIdx.y;
    const float2* in = G + ((size_t)p * M + y) * nh;
    float2* a = lds;
    float2* b = lds + N;
    for (int x = threadIdx.x; x < N; x += PC_T) {
        float2 g;
        if (x < nh) g = in[x];
        else { g = in[N - x]; g.y = -g.y; }
        a[x] = g;
    }
    float2* res = fft_lds(a, b, plan, tw, true);
    float* out = R + ((size_t)p * M + y) * N;
    const int half = N / 2;
    float bv = -INFINITY;
    int bx = 0x7fffffff;
    for (int x = threadIdx.x; x < N; x += PC_T) {
        const float v = res[x].x;
        out[x] = v;
        const int xs = (x + half) % N;
        if (v > bv || (v == bv && xs < bx)) { bv = v; bx = xs; }
    }
    s_val[threadIdx.x] = bv;
    s_xs[threadIdx.x] = bx;
    __syncthreads();
    for (int off = PC_T / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float ov = s_val[threadIdx.x + off];
            const int ox = s_xs[threadIdx.x + off];
            if (ov > s_val[threadIdx.x] || (ov == s_val[threadIdx.x] && ox < s_xs[threadIdx.x])) { s_val[threadIdx.x] = ov; s_xs[threadIdx.x] = ox; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) best[(size_t)p * M + y] = RowBest{s_val[0], s_xs[0]};
}

// first maximum over the shifted array, 5 x 5 weighted centroid -> (tx, ty, response)
__global__ __launch_bounds__(PC_T) void pc_peak_kernel(const float* __restrict__ R, const RowBest* __restrict__ best, int M, int N,
                                                       double* __restrict__ out)
{
    __shared__ float s_val[PC_T];
    __shared__ int s_pos[PC_T];   // ys * N + xs
    const int p = blockIdx.x;
    const int halfy = M / 2, halfx = N / 2;
    float bv = -INFINITY;
    int bp = 0x7fffffff;
    for (int y = threadIdx.x; y < M; y += PC_T) {
        const RowBest rb = best[(size_t)p * M + y];
        const int pos = ((y + halfy) % M) * N + rb.xs;
        if (rb.value > bv || (rb.value == bv && pos < bp)) { bv = rb.value; bp = pos; }
    }
    s_val[threadIdx.x] = bv;
    s_pos[threadIdx.x] = bp;
    __syncthreads();
    for (int off = PC_T / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float ov = s_val[threadIdx.x + off];
            const int op = s_pos[threadIdx.x + off];
            if (ov > s_val[threadIdx.x] || (ov == s_val[threadIdx.x] && op < s_pos[threadIdx.x])) { s_val[threadIdx.x] = ov; s_pos[threadIdx.x] = op; }
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    int py = 0, px = 0;
    if (s_pos[0] != 0x7fffffff) { py = s_pos[0] / N; px = s_pos[0] % N; }   // an all-NaN plane keeps (0,0); its centroid is non-finite and the caller zeroes it
    int minr = py - 2, maxr = py + 2, minc = px - 2, maxc = px + 2;
    if (minr < 0) minr = 0;
    if (minc < 0) minc = 0;
    if (maxr > M - 1) maxr = M - 1;
    if (maxc > N - 1) maxc = N - 1;
    const float* plane = R + (size_t)p * M * N;
    double cx = 0.0, cy = 0.0, sum = 0.0;
    for (int ys = minr; ys <= maxr; ++ys) {
        const int y = (ys + M - halfy) % M;   // undo the circular shift
        for (int xs = minc; xs <= maxc; ++xs) {
            const int x = (xs + N - halfx) % N;
            const double v = (double)plane[(size_t)y * N + x];
            cx += (double)xs * v;
            cy += (double)ys * v;
            sum += v;
        }
    }
    double response = sum;
    sum += DBL_EPSILON;
    cx /= sum;
    cy /= sum;
    response /= (double)(M * N);
    out[p * 3 + 0] = (double)N / 2.0 - cx;
    out[p * 3 + 1] = (double)M / 2.0 - cy;
    out[p * 3 + 2] = response;
}

int optimal_dft_size(int n)
{
    for (int v = n;; ++v) {
        int t = v;
        while (t % 2 == 0) t /= 2;
        while (t % 3 == 0) t /= 3;
        while (t % 5 == 0) t /= 5;
        if (t == 1) return v;
    }
}

FftPlan make_plan(int len)
{
    FftPlan p{};
    p.len = len;
    int t = len;
    const int radices[3] = {5, 3, 2};
    for (int r : radices)
        while (t % r == 0) { p.radix[p.stages++] = r; t /= r; }
    return p;
}

void fill_twiddles(float* dst, int len)
{
    for (int t = 0; t < len; ++t) {
        const double ang = -2.0 * M_PI * (double)t / (double)len;
        dst[2 * t + 0] = (float)cos(ang);
        dst[2 * t + 1] = (float)sin(ang);
    }
}

size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

}  // namespace

extern "C" int vstab_phase_correlate_batch(vstab_ctx* ctx, const uint8_t* gray, int n, int h, int w, vstab_fit_record* results, double* shifts)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_phase_correlate_batch: ctx is NULL");
    VSTAB_REQUIRE(gray && (results || shifts), "vstab_phase_correlate_batch: NULL pointer argument");
    VSTAB_REQUIRE(n >= 2 && h > 0 && w > 0, "vstab_phase_correlate_batch: need at least two frames of positive size");
    const int M = optimal_dft_size(h), N = optimal_dft_size(w);
    VSTAB_REQUIRE(M <= PC_MAX_LEN && N <= PC_MAX_LEN, "vstab_phase_correlate_batch: padded size %dx%d exceeds %d", N, M, PC_MAX_LEN);
    VSTAB_HIP(hipSetDevice(ctx->device));
    const int nh = N / 2 + 1, pairs = n - 1;
    const FftPlan plan_x = make_plan(N), plan_y = make_plan(M);

    std::vector<float> tw((size_t)2 * (N + M));
    fill_twiddles(tw.data(), N);
    fill_twiddles(tw.data() + 2 * (size_t)N, M);
    void* d_tw = nullptr;
    if (vstab_stage_params(ctx, tw.data(), tw.size() * sizeof(float), &d_tw)) return 1;
    const float2* tw_x = static_cast<const float2*>(d_tw);
    const float2* tw_y = tw_x + N;

    // pairs per pass: spectra of chunk+1 frames, cross-power planes, real planes and row maxima under ~1 GiB
    const size_t spec_b = (size_t)M * nh * sizeof(float2), real_b = (size_t)M * N * sizeof(float);
    int chunk = (int)((size_t(1) << 30) / (2 * spec_b + real_b + (size_t)M * sizeof(RowBest)));
    if (const char* e = getenv("VSTAB_PHASE_CHUNK")) { const int v = atoi(e); if (v > 0 && v < chunk) chunk = v; }   // tests: force several passes
    chunk = chunk < 1 ? 1 : (chunk > pairs ? pairs : chunk);
    const size_t f_b = align256(spec_b * (chunk + 1)), g_b = align256(spec_b * chunk), r_b = align256(real_b * chunk);
    const size_t best_b = align256(sizeof(RowBest) * (size_t)M * chunk), out_b = align256(sizeof(double) * 3 * (size_t)pairs);
    if (ctx->d_dis.reserve(f_b + g_b + r_b + best_b + out_b)) return 1;
    char* base = static_cast<char*>(ctx->d_dis.ptr);
    float2* F = reinterpret_cast<float2*>(base);
    float2* G = reinterpret_cast<float2*>(base + f_b);
    float* R = reinterpret_cast<float*>(base + f_b + g_b);
    RowBest* best = reinterpret_cast<RowBest*>(base + f_b + g_b + r_b);
    double* d_out = reinterpret_cast<double*>(base + f_b + g_b + r_b + best_b);
    const size_t lds_x = (size_t)2 * N * sizeof(float2), lds_y = (size_t)2 * M * sizeof(float2);
    {
        KernelTimer timer(ctx, "phase");
        for (int p0 = 0; p0 < pairs; p0 += chunk) {
            const int pc = (pairs - p0) < chunk ? (pairs - p0) : chunk;
            const int fc = pc + 1;
            hipLaunchKernelGGL(pc_row_fwd_kernel, dim3((unsigned)M, (unsigned)fc), dim3(PC_T), lds_x, ctx->stream, gray + (size_t)p0 * h * w, h, w, M,
                               N, nh, plan_x, tw_x, F);
            hipLaunchKernelGGL(pc_col_fwd_kernel, dim3((unsigned)nh, (unsigned)fc), dim3(PC_T), lds_y, ctx->stream, F, M, nh, plan_y, tw_y);
            hipLaunchKernelGGL(pc_col_inv_kernel, dim3((unsigned)nh, (unsigned)pc), dim3(PC_T), lds_y, ctx->stream, F, M, N, nh, plan_y, tw_y, G);
            hipLaunchKernelGGL(pc_row_inv_kernel, dim3((unsigned)M, (unsigned)pc), dim3(PC_T), lds_x, ctx->stream, G, M, N, nh, plan_x, tw_x, R, best);
            hipLaunchKernelGGL(pc_peak_kernel, dim3((unsigned)pc), dim3(PC_T), 0, ctx->stream, R, best, M, N, d_out + (size_t)p0 * 3);
        }
        VSTAB_HIP(hipGetLastError());
    }
    const size_t out_bytes = sizeof(double) * 3 * (size_t)pairs;
    if (ctx->h_fit.reserve(out_bytes)) return 1;
    VSTAB_HIP(hipMemcpyAsync(ctx->h_fit.ptr, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    VSTAB_HIP(hipStreamSynchronize(ctx->stream));
    const double* host = static_cast<const double*>(ctx->h_fit.ptr);
    if (shifts) memcpy(shifts, host, out_bytes);
    if (results) {
        // flow.py:110-130: non-finite results become a zero shift with confidence 0; the estimate is reported as a
        // "translation" fit whatever mode was requested, residual 0
        memset(results, 0, sizeof(vstab_fit_record) * 3 * (size_t)pairs);
        for (int p = 0; p < pairs; ++p) {
            double tx = host[p * 3 + 0], ty = host[p * 3 + 1], conf = host[p * 3 + 2];
            if (!(std::isfinite(tx) && std::isfinite(ty) && std::isfinite(conf))) tx = ty = conf = 0.0;
            for (int m = 0; m < 3; ++m) {
                vstab_fit_record& r = results[(size_t)p * 3 + m];
                r.matrix[0] = r.matrix[4] = r.matrix[8] = 1.0f;
            }
            vstab_fit_record& r = results[(size_t)p * 3 + 0];
            r.matrix[2] = (float)tx;
            r.matrix[5] = (float)ty;
            r.confidence = conf;
            r.residual = 0.0;
            r.accepted = 1;
            r.computed = 1;
        }
    }
    return 0;
}

#ifdef VSTAB_PHASE_DEBUG
// developer build only (make EXTRA=-DVSTAB_PHASE_DEBUG): spectrum of frame 0 and correlation surface of pair 0 of the
// last vstab_phase_correlate_batch call (single chunk), for stage-by-stage comparison with oracle/vo_phase.c
extern "C" int vstab_phase_debug_dump(vstab_ctx* ctx, int n, int h, int w, float* spectrum, float* surface, float* colinv)
{
    const int M = optimal_dft_size(h), N = optimal_dft_size(w), nh = N / 2 + 1, pairs = n - 1;
    const size_t spec_b = (size_t)M * nh * sizeof(float2), real_b = (size_t)M * N * sizeof(float);
    const size_t f_b = align256(spec_b * (pairs + 1)), g_b = align256(spec_b * pairs);
    char* base = static_cast<char*>(ctx->d_dis.ptr);
    VSTAB_HIP(hipMemcpy(spectrum, base, spec_b, hipMemcpyDeviceToHost));
    VSTAB_HIP(hipMemcpy(surface, base + f_b + g_b, real_b, hipMemcpyDeviceToHost));
    VSTAB_HIP(hipMemcpy(colinv, base + f_b, spec_b, hipMemcpyDeviceToHost));
    return 0;
}
#endif
