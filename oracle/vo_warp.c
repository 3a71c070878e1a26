/*
 * vo_warp.c -- CPU oracle for the warp stage (TEST INFRASTRUCTURE, see vo_common.h).
 *
 * Restates what the reference obtains from
 *   cv2.warpPerspective(frame, M, size, INTER_LINEAR|INTER_CUBIC, BORDER_CONSTANT, rgb/255)
 *   cv2.warpPerspective(ones,  M, size, INTER_NEAREST,            BORDER_CONSTANT, 0)
 * at /root/reference/nodes/video_stabilizer_flow.py:560-588 (F13),
 *    /root/reference/nodes/motion_apply.py:75-122 (A3) and :137-202 (A5).
 *
 * Algorithm source: OpenCV 4.x modules/imgproc/src/imgwarp.cpp (legacy kernels)
 *   cv::warpPerspective      -> M (f32) widened to f64, inverted in f64 (closed form 3x3)
 *   WarpPerspectiveInvoker   -> per 64x16 block: X0=M0*x+M1*y+M2 ... ; per pixel
 *                               X=cvRound((X0+M0*x1)*(32/W)), integer part X>>5,
 *                               fraction X&31  (1/32-pixel quantisation)
 *   remapBilinear<float>     -> f32 weights from a 32x32 table, sum of 4 products, left to right
 *   remapBicubic<float>      -> A=-0.75 separable table, row-wise partial sums
 *   remapNearest<float>      -> in-bounds test on the rounded coordinate
 * Written from the published algorithm (no OpenCV source is available in this
 * container); parity against a real cv2 is UNPINNED.
 *
 * mode VO_SUBPIX_EXACT restates the newer (OpenCV >= 4.11) INTER_LINEAR kernels,
 * which interpolate at full f32 coordinate precision; it is even less pinned.
 */
#include "vo_common.h"
#include "vstab_oracle.h"

#define TAB_BITS 5
#define TAB_SIZE 32

int vo_invert3x3(const double* S, double* D)
{
    /* cv::invert, n == 3, CV_64F branch */
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) +
               S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d == 0.0) {
        for (int i = 0; i < 9; i++) D[i] = 0.0;
        return 0;
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
    return 1;
}

/* initInterTab1D: linear -> {1-x, x}; cubic -> A=-0.75 kernel; x = i*(1/32) in f32 */
void vo_interp_tables(float* lin /*32*2*/, float* cub /*32*4*/)
{
    const float scale = 1.f / TAB_SIZE;
    for (int i = 0; i < TAB_SIZE; i++) {
        float x = i * scale;
        lin[i * 2 + 0] = 1.f - x;
        lin[i * 2 + 1] = x;
        const float A = -0.75f;
        float* c = cub + i * 4;
        c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
        c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
    }
}

static inline int clamp_to_int(double v)
{
    /* std::max((double)INT_MIN, std::min((double)INT_MAX, v)) then cvRound */
    double lo = (double)INT_MIN, hi = (double)INT_MAX;
    double m = (v < hi) ? v : hi;   /* std::min(hi, v): (v < hi) ? v : hi ; NaN -> hi */
    double r = (lo < m) ? m : lo;   /* std::max(lo, m): (lo < m) ? m : lo */
    return vo_round_d(r);
}

/* block width used by WarpPerspectiveInvoker for a destination of (dw x dh) */
static int warp_block_width(int dw, int dh)
{
    const int BLOCK_SZ = 32;
    int bh0 = BLOCK_SZ / 2 < dh ? BLOCK_SZ / 2 : dh;
    int bw0 = BLOCK_SZ * BLOCK_SZ / bh0 < dw ? BLOCK_SZ * BLOCK_SZ / bh0 : dw;
    return bw0;
}

static void fetch_bilinear(const float* src, int sh, int sw, int sx, int sy, int a,
                           const float* lin, const float* border, float* out)
{
    const int fx = a & (TAB_SIZE - 1), fy = a >> TAB_BITS;
    const float wy0 = lin[fy * 2], wy1 = lin[fy * 2 + 1];
    const float wx0 = lin[fx * 2], wx1 = lin[fx * 2 + 1];
    const float w0 = wy0 * wx0, w1 = wy0 * wx1, w2 = wy1 * wx0, w3 = wy1 * wx1;
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
        out[0] = border[0]; out[1] = border[1]; out[2] = border[2];
        return;
    }
    const int x0ok = sx >= 0 && sx < sw, x1ok = sx + 1 >= 0 && sx + 1 < sw;
    const int y0ok = sy >= 0 && sy < sh, y1ok = sy + 1 >= 0 && sy + 1 < sh;
    const float* v0 = (x0ok && y0ok) ? src + ((size_t)sy * sw + sx) * 3 : border;
    const float* v1 = (x1ok && y0ok) ? src + ((size_t)sy * sw + sx + 1) * 3 : border;
    const float* v2 = (x0ok && y1ok) ? src + ((size_t)(sy + 1) * sw + sx) * 3 : border;
    const float* v3 = (x1ok && y1ok) ? src + ((size_t)(sy + 1) * sw + sx + 1) * 3 : border;
    for (int k = 0; k < 3; k++)
        out[k] = v0[k] * w0 + v1[k] * w1 + v2[k] * w2 + v3[k] * w3;
}

static void fetch_bicubic(const float* src, int sh, int sw, int sx0, int sy0, int a,
                          const float* cub, const float* border, float* out)
{
    const int fx = a & (TAB_SIZE - 1), fy = a >> TAB_BITS;
    float w[16];
    for (int k1 = 0; k1 < 4; k1++)
        for (int k2 = 0; k2 < 4; k2++)
            w[k1 * 4 + k2] = cub[fy * 4 + k1] * cub[fx * 4 + k2];
    const int sx = sx0 - 1, sy = sy0 - 1;
    const unsigned width1 = (unsigned)(sw - 3 > 0 ? sw - 3 : 0);
    const unsigned height1 = (unsigned)(sh - 3 > 0 ? sh - 3 : 0);
    if ((unsigned)sx < width1 && (unsigned)sy < height1) {
        for (int k = 0; k < 3; k++) {
            const float* S = src + ((size_t)sy * sw + sx) * 3 + k;
            const size_t st = (size_t)sw * 3;
            float sum = S[0] * w[0] + S[3] * w[1] + S[6] * w[2] + S[9] * w[3];
            S += st;
            sum += S[0] * w[4] + S[3] * w[5] + S[6] * w[6] + S[9] * w[7];
            S += st;
            sum += S[0] * w[8] + S[3] * w[9] + S[6] * w[10] + S[9] * w[11];
            S += st;
            sum += S[0] * w[12] + S[3] * w[13] + S[6] * w[14] + S[9] * w[15];
            out[k] = sum;
        }
        return;
    }
    if (sx >= sw || sx + 4 <= 0 || sy >= sh || sy + 4 <= 0) {
        out[0] = border[0]; out[1] = border[1]; out[2] = border[2];
        return;
    }
    int x[4], y[4];
    for (int i = 0; i < 4; i++) {
        x[i] = (sx + i >= 0 && sx + i < sw) ? (sx + i) : -1;
        y[i] = (sy + i >= 0 && sy + i < sh) ? (sy + i) : -1;
    }
    for (int k = 0; k < 3; k++) {
        float cv = border[k], sum = cv;
        for (int i = 0; i < 4; i++) {
            if (y[i] < 0) continue;
            const float* S = src + (size_t)y[i] * sw * 3 + k;
            const float* wr = w + i * 4;
            if (x[0] >= 0) sum += (S[x[0] * 3] - cv) * wr[0];
            if (x[1] >= 0) sum += (S[x[1] * 3] - cv) * wr[1];
            if (x[2] >= 0) sum += (S[x[2] * 3] - cv) * wr[2];
            if (x[3] >= 0) sum += (S[x[3] * 3] - cv) * wr[3];
        }
        out[k] = sum;
    }
}

/* OpenCV >= 4.11 style full-precision bilinear (unpinned, see header) */
static void fetch_bilinear_exact(const float* src, int sh, int sw, float fsx, float fsy,
                                 const float* border, float* out)
{
    int ix = vo_floor_f(fsx), iy = vo_floor_f(fsy);
    float ax = fsx - ix, ay = fsy - iy;
    if (!(fsx == fsx) || !(fsy == fsy) || ix >= sw || ix + 1 < 0 || iy >= sh || iy + 1 < 0) {
        out[0] = border[0]; out[1] = border[1]; out[2] = border[2];
        return;
    }
    const int x0ok = ix >= 0 && ix < sw, x1ok = ix + 1 >= 0 && ix + 1 < sw;
    const int y0ok = iy >= 0 && iy < sh, y1ok = iy + 1 >= 0 && iy + 1 < sh;
    const float* p00 = (x0ok && y0ok) ? src + ((size_t)iy * sw + ix) * 3 : border;
    const float* p01 = (x1ok && y0ok) ? src + ((size_t)iy * sw + ix + 1) * 3 : border;
    const float* p10 = (x0ok && y1ok) ? src + ((size_t)(iy + 1) * sw + ix) * 3 : border;
    const float* p11 = (x1ok && y1ok) ? src + ((size_t)(iy + 1) * sw + ix + 1) * 3 : border;
    for (int k = 0; k < 3; k++) {
        float v0 = p00[k] + ax * (p01[k] - p00[k]);
        float v1 = p10[k] + ax * (p11[k] - p10[k]);
        out[k] = v0 + ay * (v1 - v0);
    }
}

void vo_warp_frame_inv(const float* src, int sh, int sw, const double* Minv, int dh, int dw,
                       int interp, const float* border, int subpix, float* dst, float* coverage)
{
    float lin[TAB_SIZE * 2], cub[TAB_SIZE * 4];
    vo_interp_tables(lin, cub);
    const double* M = Minv;
    const int bw0 = warp_block_width(dw, dh);
    float Mf[9];
    for (int i = 0; i < 9; i++) Mf[i] = (float)M[i];

    for (int y = 0; y < dh; y++) {
        for (int xb = 0; xb < dw; xb += bw0) {
            const int bw = bw0 < dw - xb ? bw0 : dw - xb;
            const double X0 = M[0] * xb + M[1] * y + M[2];
            const double Y0 = M[3] * xb + M[4] * y + M[5];
            const double W0 = M[6] * xb + M[7] * y + M[8];
            for (int x1 = 0; x1 < bw; x1++) {
                const int x = xb + x1;
                float* D = dst ? dst + ((size_t)y * dw + x) * 3 : 0;
                if (D) {
                    if (subpix == VO_SUBPIX_EXACT && interp == VO_INTERP_BILINEAR) {
                        float w = x * Mf[6] + y * Mf[7] + Mf[8];
                        float fsx = (x * Mf[0] + y * Mf[1] + Mf[2]) / w;
                        float fsy = (x * Mf[3] + y * Mf[4] + Mf[5]) / w;
                        fetch_bilinear_exact(src, sh, sw, fsx, fsy, border, D);
                    } else {
                        double W = W0 + M[6] * x1;
                        W = W ? TAB_SIZE / W : 0;
                        int X = clamp_to_int((X0 + M[0] * x1) * W);
                        int Y = clamp_to_int((Y0 + M[3] * x1) * W);
                        int sx = vo_sat_short(X >> TAB_BITS), sy = vo_sat_short(Y >> TAB_BITS);
                        int a = (Y & (TAB_SIZE - 1)) * TAB_SIZE + (X & (TAB_SIZE - 1));
                        if (interp == VO_INTERP_BICUBIC)
                            fetch_bicubic(src, sh, sw, sx, sy, a, cub, border, D);
                        else
                            fetch_bilinear(src, sh, sw, sx, sy, a, lin, border, D);
                    }
                }
                if (coverage) {
                    double W = W0 + M[6] * x1;
                    W = W ? 1. / W : 0;
                    int X = clamp_to_int((X0 + M[0] * x1) * W);
                    int Y = clamp_to_int((Y0 + M[3] * x1) * W);
                    int sx = vo_sat_short(X), sy = vo_sat_short(Y);
                    coverage[(size_t)y * dw + x] =
                        ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) ? 1.f : 0.f;
                }
            }
        }
    }
}

/* cv2.warpPerspective without WARP_INVERSE_MAP: M given as float32 3x3 */
void vo_warp_frame(const float* src, int sh, int sw, const float* M32, int dh, int dw, int interp,
                   const float* border, int subpix, float* dst, float* coverage)
{
    double M[9], Mi[9];
    for (int i = 0; i < 9; i++) M[i] = (double)M32[i];
    vo_invert3x3(M, Mi);
    vo_warp_frame_inv(src, sh, sw, Mi, dh, dw, interp, border, subpix, dst, coverage);
}

/*
 * Plain warp of a clip (flow.py:560-588 / motion_apply.py:92-120):
 * frame = warp(frame, M), mask = 1 - (warp(ones, NEAREST) > 0.5), mask < 1e-3 -> 0.
 * pad_count[i] = number of mask==1 pixels (mask.mean() = count / (h*w) in f32, exact).
 */
void vo_warp_clip(const float* src, int n, int sh, int sw, const float* M32, int dh, int dw,
                  int interp, const float* border, int subpix, float* dst, float* mask,
                  unsigned* pad_count)
{
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < n; i++) {
        const float* S = src + (size_t)i * sh * sw * 3;
        float* D = dst + (size_t)i * dh * dw * 3;
        float* Mk = mask ? mask + (size_t)i * dh * dw : 0;
        vo_warp_frame(S, sh, sw, M32 + i * 9, dh, dw, interp, border, subpix, D, Mk);
        if (Mk) {
            unsigned cnt = 0;
            for (size_t p = 0; p < (size_t)dh * dw; p++) {
                float m = 1.0f - (Mk[p] > 0.5f ? 1.0f : 0.0f);
                if (m < 1e-3f) m = 0.0f;
                Mk[p] = m;
                cnt += m > 0.5f;
            }
            if (pad_count) pad_count[i] = cnt;
        }
    }
}

/*
 * Motion-blur warp of a clip (motion_apply.py:125-202).
 * matrices: [n,9] f64 (motion_meta matrices).  For frame i: delta = M[i+1]-M[i]
 * (last frame: M[i]-M[i-1]) element-wise in f64; ts = linspace(0, blur, S);
 * sample k = f32(M[i] + delta*ts[k]); accumulate S warps in f32 in sample order,
 * divide by float(S); mask = 1 - coverage_sum/S, < 1e-3 -> 0.
 */
void vo_linspace(double a, double b, int n, double* out)
{
    /* numpy.linspace(a, b, n): step = (b-a)/(n-1); y = arange(n)*step + a; y[-1] = b */
    if (n == 1) { out[0] = a; return; }
    double step = (b - a) / (double)(n - 1);
    for (int i = 0; i < n; i++) out[i] = (double)i * step + a;
    out[n - 1] = b;
}

void vo_blur_sample_matrices(const double* matrices, int n, int idx, double blur, int samples,
                             float* out32 /* samples*9 */)
{
    const double* base = matrices + (size_t)idx * 9;
    double delta[9];
    if (n <= 1) {
        for (int j = 0; j < 9; j++) out32[j] = (float)base[j];
        return;
    }
    if (idx < n - 1)
        for (int j = 0; j < 9; j++) delta[j] = matrices[(size_t)(idx + 1) * 9 + j] - base[j];
    else
        for (int j = 0; j < 9; j++) delta[j] = base[j] - matrices[(size_t)(idx - 1) * 9 + j];
    double ts[64];
    vo_linspace(0.0, blur, samples, ts);
    for (int k = 0; k < samples; k++)
        for (int j = 0; j < 9; j++) out32[k * 9 + j] = (float)(base[j] + delta[j] * ts[k]);
}

void vo_warp_blur_clip(const float* src, int n, int sh, int sw, const double* matrices, int dh,
                       int dw, int interp, const float* border, int subpix, double blur,
                       int samples, float* dst, float* mask)
{
    const int S = (n <= 1) ? 1 : samples;
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < n; i++) {
        const size_t np = (size_t)dh * dw;
        float* tmp = (float*)malloc(np * 3 * sizeof(float));
        float* cov = (float*)malloc(np * sizeof(float));
        float* acc = dst + (size_t)i * np * 3;
        float* cacc = mask ? mask + (size_t)i * np : 0;
        float m32[64 * 9];
        vo_blur_sample_matrices(matrices, n, i, blur, samples, m32);
        for (size_t p = 0; p < np * 3; p++) acc[p] = 0.f;
        if (cacc) for (size_t p = 0; p < np; p++) cacc[p] = 0.f;
        for (int k = 0; k < S; k++) {
            vo_warp_frame(src + (size_t)i * sh * sw * 3, sh, sw, m32 + k * 9, dh, dw, interp,
                          border, subpix, tmp, cacc ? cov : 0);
            for (size_t p = 0; p < np * 3; p++) acc[p] += tmp[p];
            if (cacc) for (size_t p = 0; p < np; p++) cacc[p] += (cov[p] > 0.5f ? 1.f : 0.f);
        }
        /* reference divides by float(sample_count) even when a 1-frame clip yields one sample */
        const float fs = (float)samples;
        for (size_t p = 0; p < np * 3; p++) acc[p] = acc[p] / fs;
        if (cacc)
            for (size_t p = 0; p < np; p++) {
                float m = 1.0f - cacc[p] / fs;
                if (m < 1e-3f) m = 0.f;
                cacc[p] = m;
            }
        free(tmp);
        free(cov);
    }
}
