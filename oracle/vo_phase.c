/*
 * vo_phase.c -- CPU oracle (TEST INFRASTRUCTURE ONLY) for the phase-correlation fallback estimator:
 * cv2.phaseCorrelate(prev.astype(float32), curr.astype(float32)) as called at
 * nodes/video_stabilizer_flow.py:110-130, restated from the published OpenCV 4.x
 * imgproc/phasecorr.cpp (phaseCorrelate, magSpectrums, divSpectrums, fftShift, weightedCentroid; no window):
 *
 *   M, N = getOptimalDFTSize(rows), getOptimalDFTSize(cols); zero padding at the bottom / right
 *   FFT1, FFT2 = dft(image)                      (real input, unscaled)
 *   P  = FFT1 * conj(FFT2)                       float results of double products
 *   Pm = |P| as (float)sqrt(double)              purely real bins: P^2 (the packed-format helper squares them)
 *   C  = P / Pm with eps = FLT_EPSILON           complex bins: (re*Pm, im*Pm) / (Pm^2 + eps) in double;
 *                                                purely real bins: P / (P^2 + eps) in float
 *   C  = idft(C) (unscaled), circular shift by (N/2, M/2)
 *   peak = first maximum in raster order; 5x5 weighted centroid around it, clamped to the array, double sums
 *   response = sum / (M*N); shift = (N/2 - cx, M/2 - cy)
 *
 * PARITY UNPINNED against a real OpenCV (none is importable here or on the GPU box): OpenCV's DFT factorises
 * and orders its butterflies differently, so its float spectra differ from these in the last bits.  The DFT
 * here is a mixed-radix (5, 3, 2) Stockham transform; csrc/vstab_phase.hip performs the same butterflies in
 * the same order with the same twiddle table and is compared bit for bit.  tests/ additionally check this
 * file against a float64 numpy.fft restatement and against known shifts.
 */
#include "vo_common.h"
#include "vstab_oracle.h"
#include <float.h>

typedef struct { float x, y; } cf;

static inline cf cmul(cf a, cf b) { cf r = {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; return r; }
static inline cf cadd(cf a, cf b) { cf r = {a.x + b.x, a.y + b.y}; return r; }

int vo_optimal_dft_size(int n)
{
    for (int v = n;; ++v) {
        int t = v;
        while (t % 2 == 0) t /= 2;
        while (t % 3 == 0) t /= 3;
        while (t % 5 == 0) t /= 5;
        if (t == 1) return v;
    }
}

static void make_twiddles(cf* tw, int len)
{
    for (int t = 0; t < len; ++t) {
        const double ang = -2.0 * M_PI * (double)t / (double)len;
        tw[t].x = (float)cos(ang);
        tw[t].y = (float)sin(ang);
    }
}

/* Stockham transform of `len` points with stride-1 arrays a (input) / b (scratch); returns the result array. */
static cf* fft_run(cf* a, cf* b, int len, const cf* tw, int inverse)
{
    int radix[16], stages = 0, t = len;
    const int radices[3] = {5, 3, 2};
    for (int i = 0; i < 3; ++i)
        while (t % radices[i] == 0) { radix[stages++] = radices[i]; t /= radices[i]; }
    int ns = 1;
    for (int s = 0; s < stages; ++s) {
        const int r = radix[s], m = len / r, tstep = len / (ns * r), rstep = len / r;
        for (int j = 0; j < m; ++j) {
            const int k = j % ns;
            cf v[5];
            for (int q = 0; q < r; ++q) {
                v[q] = a[j + q * m];
                if (q) {
                    cf w = tw[q * k * tstep];
                    if (inverse) w.y = -w.y;
                    v[q] = cmul(v[q], w);
                }
            }
            const int j0 = (j - k) * r + k;
            for (int p = 0; p < r; ++p) {
                cf acc = v[0];
                for (int q = 1; q < r; ++q) {
                    cf w = tw[((p * q) % r) * rstep];
                    if (inverse) w.y = -w.y;
                    acc = cadd(acc, cmul(v[q], w));
                }
                b[j0 + p * ns] = acc;
            }
        }
        ns *= r;
        cf* sw = a; a = b; b = sw;
    }
    return a;
}

/* forward transform of one zero-padded u8 image -> F [M][nh] */
static void forward_image(const uint8_t* img, int h, int w, int M, int N, const cf* twx, const cf* twy, cf* F)
{
    const int nh = N / 2 + 1, L = M > N ? M : N;
    cf* a = (cf*)malloc(sizeof(cf) * 2 * (size_t)L);
    cf* b = a + L;
    for (int y = 0; y < M; ++y) {
        cf* out = F + (size_t)y * nh;
        if (y >= h) { memset(out, 0, sizeof(cf) * nh); continue; }
        for (int x = 0; x < N; ++x) { a[x].x = x < w ? (float)img[(size_t)y * w + x] : 0.0f; a[x].y = 0.0f; }
        const cf* res = fft_run(a, b, N, twx, 0);
        memcpy(out, res, sizeof(cf) * nh);
    }
    for (int kx = 0; kx < nh; ++kx) {
        for (int y = 0; y < M; ++y) a[y] = F[(size_t)y * nh + kx];
        const cf* res = fft_run(a, b, M, twy, 0);
        for (int y = 0; y < M; ++y) F[(size_t)y * nh + kx] = res[y];
    }
    free(a);
}

static float* g_debug_G = NULL;   /* vo_phase_debug_pair only (single-threaded use) */

static void correlate_pair(const cf* F1, const cf* F2, int M, int N, const cf* twx, const cf* twy, double* out, float* surface)
{
    const int nh = N / 2 + 1, L = M > N ? M : N;
    const float eps = FLT_EPSILON;
    cf* a = (cf*)malloc(sizeof(cf) * 2 * (size_t)L);
    cf* b = a + L;
    cf* G = (cf*)malloc(sizeof(cf) * (size_t)M * nh);
    float* R = surface ? surface : (float*)malloc(sizeof(float) * (size_t)M * N);
    for (int kx = 0; kx < nh; ++kx) {
        const int real_col = kx == 0 || ((N & 1) == 0 && kx == N / 2);
        for (int ky = 0; ky < M; ++ky) {
            const cf u = F1[(size_t)ky * nh + kx], v = F2[(size_t)ky * nh + kx];
            cf c;
            if (real_col && (ky == 0 || ((M & 1) == 0 && ky == M / 2))) {
                const float pr = u.x * v.x;
                const float pm = pr * pr;
                c.x = pr / (pm + eps);
                c.y = 0.0f;
            } else {
                const float re = (float)((double)u.x * v.x + (double)u.y * v.y);
                const float im = (float)((double)u.y * v.x - (double)u.x * v.y);
                const float mag = (float)sqrt((double)re * re + (double)im * im);
                const double denom = (double)mag * mag + (double)eps;
                c.x = (float)(((double)re * mag) / denom);
                c.y = (float)(((double)im * mag) / denom);
            }
            a[ky] = c;
        }
        const cf* res = fft_run(a, b, M, twy, 1);
        for (int y = 0; y < M; ++y) G[(size_t)y * nh + kx] = res[y];
    }
    for (int y = 0; y < M; ++y) {
        const cf* in = G + (size_t)y * nh;
        for (int x = 0; x < N; ++x) {
            if (x < nh) a[x] = in[x];
            else { a[x] = in[N - x]; a[x].y = -a[x].y; }
        }
        const cf* res = fft_run(a, b, N, twx, 1);
        for (int x = 0; x < N; ++x) R[(size_t)y * N + x] = res[x].x;
    }
    if (g_debug_G && surface) memcpy(g_debug_G, G, sizeof(cf) * (size_t)M * nh);
    /* minMaxLoc on the shifted surface: first maximum in raster order */
    const int halfy = M / 2, halfx = N / 2;
    float bv = -INFINITY;
    int py = 0, px = 0;
    for (int ys = 0; ys < M; ++ys) {
        const int y = (ys + M - halfy) % M;
        for (int xs = 0; xs < N; ++xs) {
            const int x = (xs + N - halfx) % N;
            const float v = R[(size_t)y * N + x];
            if (v > bv) { bv = v; py = ys; px = xs; }
        }
    }
    int minr = py - 2, maxr = py + 2, minc = px - 2, maxc = px + 2;
    if (minr < 0) minr = 0;
    if (minc < 0) minc = 0;
    if (maxr > M - 1) maxr = M - 1;
    if (maxc > N - 1) maxc = N - 1;
    double cx = 0.0, cy = 0.0, sum = 0.0;
    for (int ys = minr; ys <= maxr; ++ys) {
        const int y = (ys + M - halfy) % M;
        for (int xs = minc; xs <= maxc; ++xs) {
            const int x = (xs + N - halfx) % N;
            const double v = (double)R[(size_t)y * N + x];
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
    out[0] = (double)N / 2.0 - cx;
    out[1] = (double)M / 2.0 - cy;
    out[2] = response;
    if (!surface) free(R);
    free(G);
    free(a);
}

/* gray [n][h][w] u8 -> shifts [n-1][3] (tx, ty, response) exactly as cv2.phaseCorrelate returns them.
 * surface: optional [M][N] float output of the (unshifted) correlation surface of the FIRST pair (tests). */
void vo_phase_correlate_clip(const uint8_t* gray, int n, int h, int w, double* shifts, float* surface)
{
    const int M = vo_optimal_dft_size(h), N = vo_optimal_dft_size(w), nh = N / 2 + 1;
    cf* twx = (cf*)malloc(sizeof(cf) * N);
    cf* twy = (cf*)malloc(sizeof(cf) * M);
    make_twiddles(twx, N);
    make_twiddles(twy, M);
    cf* F = (cf*)malloc(sizeof(cf) * (size_t)n * M * nh);
#pragma omp parallel for schedule(dynamic)
    for (int f = 0; f < n; ++f) forward_image(gray + (size_t)f * h * w, h, w, M, N, twx, twy, F + (size_t)f * M * nh);
#pragma omp parallel for schedule(dynamic)
    for (int p = 0; p < n - 1; ++p)
        correlate_pair(F + (size_t)p * M * nh, F + (size_t)(p + 1) * M * nh, M, N, twx, twy, shifts + (size_t)p * 3, p == 0 ? surface : NULL);
    free(F);
    free(twy);
    free(twx);
}

/* spectrum of one zero-padded image, [M][N/2+1] complex float (stage-level comparisons in tests / tools) */
void vo_phase_spectrum(const uint8_t* img, int h, int w, float* spectrum)
{
    const int M = vo_optimal_dft_size(h), N = vo_optimal_dft_size(w);
    cf* twx = (cf*)malloc(sizeof(cf) * N);
    cf* twy = (cf*)malloc(sizeof(cf) * M);
    make_twiddles(twx, N);
    make_twiddles(twy, M);
    forward_image(img, h, w, M, N, twx, twy, (cf*)spectrum);
    free(twy);
    free(twx);
}

/* stage-level view of one pair for tools/phase_debug.py: G = column-inverse of the cross-power spectrum [M][N/2+1][2] */
void vo_phase_debug_pair(const uint8_t* pair /*[2][h][w]*/, int h, int w, float* G, float* surface, double* shift)
{
    g_debug_G = G;
    vo_phase_correlate_clip(pair, 2, h, w, shift, surface);
    g_debug_G = NULL;
}
