/*
 * vo_classic.c -- CPU oracle (TEST INFRASTRUCTURE ONLY) for the sparse estimator of the
 * `Video Stabilizer Classic` node: the two OpenCV calls of
 * nodes/video_stabilizer_classic.py:76-96 restated from the published OpenCV 4.x algorithms
 *
 *   cv2.goodFeaturesToTrack(gray, maxCorners=400, qualityLevel=0.01, minDistance=7, blockSize=21)
 *       imgproc/featureselect.cpp (goodFeaturesToTrack) + corner.cpp (cornerMinEigenVal):
 *       Sobel 3x3 scaled by 1/(4*blockSize*255) -> dx^2, dxdy, dy^2 -> unnormalised 21x21 box sums
 *       -> min eigenvalue -> threshold at quality*max -> 3x3 local maxima -> sort by strength
 *       -> greedy minimum-distance selection
 *   cv2.calcOpticalFlowPyrLK(prev, next, pts, None, winSize=(31,31), maxLevel=3,
 *                            criteria=(EPS|COUNT, 50, 0.01))
 *       video/lkpyramid.cpp: pyrDown pyramids, Scharr derivatives, 14-bit fixed-point bilinear
 *       windows, iterative 2x2 solve per point and level.
 *
 * PARITY UNPINNED against a real OpenCV (none is importable here or on the GPU box).  Two
 * deliberate, documented choices where OpenCV's own result depends on its SIMD dispatch:
 *   * the float accumulations of the LK normal matrix / mismatch vector and of the box filter
 *     are replaced by exact integer (resp. double) sums converted once -- the value every
 *     OpenCV summation order approximates;
 *   * the filter engine's multiply-adds are written as explicit fmaf (the AVX2 dispatch).
 * The HIP kernels (csrc/vstab_classic.hip) follow the same definitions and are compared bit for bit.
 */
#include "vo_common.h"
#include "vstab_oracle.h"

static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

/* ---- cornerMinEigenVal (corner.cpp), u8 source, aperture 3 -------------------------------- */
void vo_min_eigen_val(const uint8_t* img, int h, int w, int block, float* eig)
{
    const float s = (float)(1.0 / (4.0 * block * 255.0));
    const float k2 = s * 2.f;
    float* cov = (float*)malloc(sizeof(float) * 3 * (size_t)h * w);
    /* Sobel dx: rows [-1 0 1] (exact), columns [s 2s s];  dy: rows [s 2s s], columns [-1 0 1] */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        const uint8_t* r0 = img + (size_t)reflect101(y - 1, h) * w;
        const uint8_t* r1 = img + (size_t)y * w;
        const uint8_t* r2 = img + (size_t)reflect101(y + 1, h) * w;
        for (int x = 0; x < w; x++) {
            const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
            const float d0 = (float)(r0[xr] - r0[xl]), d1 = (float)(r1[xr] - r1[xl]), d2 = (float)(r2[xr] - r2[xl]);
            const float dx = fmaf(d0 + d2, s, d1 * k2);
            const float s0 = fmaf(s, (float)r0[xr], fmaf(k2, (float)r0[x], s * (float)r0[xl]));
            const float s2 = fmaf(s, (float)r2[xr], fmaf(k2, (float)r2[x], s * (float)r2[xl]));
            const float dy = s2 - s0;
            float* c = cov + ((size_t)y * w + x) * 3;
            c[0] = dx * dx; c[1] = dx * dy; c[2] = dy * dy;
        }
    }
    /* unnormalised block x block box sums, double accumulators, reflect-101 border */
    const int r = block / 2;
    double* rows = (double*)malloc(sizeof(double) * 3 * (size_t)h * w);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double a = 0, b = 0, c = 0;
            for (int k = -r; k < block - r; k++) {
                const float* p = cov + ((size_t)y * w + reflect101(x + k, w)) * 3;
                a += p[0]; b += p[1]; c += p[2];
            }
            double* o = rows + ((size_t)y * w + x) * 3;
            o[0] = a; o[1] = b; o[2] = c;
        }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double a = 0, b = 0, c = 0;
            for (int k = -r; k < block - r; k++) {
                const double* p = rows + ((size_t)reflect101(y + k, h) * w + x) * 3;
                a += p[0]; b += p[1]; c += p[2];
            }
            const float fa = (float)a * 0.5f, fb = (float)b, fc = (float)c * 0.5f;
            const float t = fa - fc;
            eig[(size_t)y * w + x] = (fa + fc) - sqrtf(fmaf(fb, fb, t * t));
        }
    free(cov); free(rows);
}

typedef struct { float v; int idx; } vo_cand;
static int cand_cmp(const void* pa, const void* pb)
{
    const vo_cand* a = (const vo_cand*)pa; const vo_cand* b = (const vo_cand*)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return a->idx > b->idx ? -1 : (a->idx < b->idx ? 1 : 0);   /* greaterThanPtr: higher address first */
}

/* goodFeaturesToTrack without mask / Harris.  corners [max_corners][2]; returns the count. */
int vo_good_features(const uint8_t* img, int h, int w, int max_corners, double quality, double min_distance,
                     int block, float* corners)
{
    float* eig = (float*)malloc(sizeof(float) * (size_t)h * w);
    vo_min_eigen_val(img, h, w, block, eig);
    float mx = eig[0];
    for (size_t i = 1; i < (size_t)h * w; i++) mx = eig[i] > mx ? eig[i] : mx;
    const float thr = (float)((double)mx * quality);
    for (size_t i = 0; i < (size_t)h * w; i++) eig[i] = eig[i] > thr ? eig[i] : 0.f;
    vo_cand* cand = (vo_cand*)malloc(sizeof(vo_cand) * (size_t)h * w);
    int total = 0;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            const float v = eig[(size_t)y * w + x];
            if (v == 0.f) continue;
            float m = v;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    const float q = eig[(size_t)(y + dy) * w + x + dx];
                    m = q > m ? q : m;
                }
            if (v == m) { cand[total].v = v; cand[total].idx = y * w + x; total++; }
        }
    qsort(cand, (size_t)total, sizeof(vo_cand), cand_cmp);
    int n = 0;
    const float md2 = (float)(min_distance * min_distance);
    for (int i = 0; i < total; i++) {
        const int y = cand[i].idx / w, x = cand[i].idx - y * w;
        int good = 1;
        if (min_distance >= 1)
            for (int j = 0; j < n; j++) {
                const float dx = (float)x - corners[j * 2], dy = (float)y - corners[j * 2 + 1];
                if (dx * dx + dy * dy < md2) { good = 0; break; }
            }
        if (good) {
            corners[n * 2] = (float)x; corners[n * 2 + 1] = (float)y;
            n++;
            if (max_corners > 0 && n == max_corners) break;
        }
    }
    free(cand); free(eig);
    return n;
}

/* ---- pyrDown, u8, 5x5 Gaussian in integers (imgproc/pyramids.cpp) -------------------------- */
void vo_pyr_down_u8(const uint8_t* src, int sh, int sw, uint8_t* dst)
{
    const int dh = (sh + 1) / 2, dw = (sw + 1) / 2;
    static const int kw[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int acc = 0;
            for (int j = 0; j < 5; j++) {
                const uint8_t* row = src + (size_t)reflect101(2 * y - 2 + j, sh) * sw;
                int racc = 0;
                for (int i = 0; i < 5; i++) racc += kw[i] * row[reflect101(2 * x - 2 + i, sw)];
                acc += kw[j] * racc;
            }
            dst[(size_t)y * dw + x] = (uint8_t)((acc + 128) >> 8);
        }
}

/* calcSharrDeriv (lkpyramid.cpp): interleaved (dx,dy) shorts, reflect-101 at the image border */
void vo_scharr_deriv(const uint8_t* src, int h, int w, short* deriv)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* r0 = src + (size_t)reflect101(y - 1, h) * w;
        const uint8_t* r1 = src + (size_t)y * w;
        const uint8_t* r2 = src + (size_t)reflect101(y + 1, h) * w;
        for (int x = 0; x < w; x++) {
            const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
            const int t0l = (r0[xl] + r2[xl]) * 3 + r1[xl] * 10, t0r = (r0[xr] + r2[xr]) * 3 + r1[xr] * 10;
            const int t1l = r2[xl] - r0[xl], t1c = r2[x] - r0[x], t1r = r2[xr] - r0[xr];
            deriv[((size_t)y * w + x) * 2] = (short)(t0r - t0l);
            deriv[((size_t)y * w + x) * 2 + 1] = (short)((t1r + t1l) * 3 + t1c * 10);
        }
    }
}

#define LK_MAX_LEVELS 8
typedef struct { int h, w; uint8_t* img; short* deriv; } lk_level;

static int build_pyramid(const uint8_t* img, int h, int w, int win, int max_level, lk_level* lv, int with_deriv)
{
    /* buildOpticalFlowPyramid: level 0 is the image; stop once the NEXT level would not exceed the window */
    int levels = 0;
    lv[0].h = h; lv[0].w = w;
    lv[0].img = (uint8_t*)malloc((size_t)h * w);
    memcpy(lv[0].img, img, (size_t)h * w);
    for (int level = 0; level <= max_level; level++) {
        if (level != 0) {
            lv[level].h = (lv[level - 1].h + 1) / 2; lv[level].w = (lv[level - 1].w + 1) / 2;
            lv[level].img = (uint8_t*)malloc((size_t)lv[level].h * lv[level].w);
            vo_pyr_down_u8(lv[level - 1].img, lv[level - 1].h, lv[level - 1].w, lv[level].img);
        }
        levels = level;
        const int nh = (lv[level].h + 1) / 2, nw = (lv[level].w + 1) / 2;
        if (nw <= win || nh <= win) break;
    }
    for (int l = 0; l <= levels; l++) {
        lv[l].deriv = NULL;
        if (with_deriv) {
            lv[l].deriv = (short*)malloc(sizeof(short) * 2 * (size_t)lv[l].h * lv[l].w);
            vo_scharr_deriv(lv[l].img, lv[l].h, lv[l].w, lv[l].deriv);
        }
    }
    return levels;
}

int vo_lk_levels(int h, int w, int win, int max_level)
{
    int level = 0;
    for (; level <= max_level; level++) {
        const int nh = (h + 1) / 2, nw = (w + 1) / 2;
        if (nw <= win || nh <= win) break;
        h = nh; w = nw;
    }
    return level > max_level ? max_level : level;
}

static inline int img_at(const lk_level* L, int x, int y)   /* pyramid border: BORDER_REFLECT_101 */
{
    return L->img[(size_t)reflect101(y, L->h) * L->w + reflect101(x, L->w)];
}
static inline int der_at(const lk_level* L, int x, int y, int c)   /* derivative border: constant 0 */
{
    if ((unsigned)x >= (unsigned)L->w || (unsigned)y >= (unsigned)L->h) return 0;
    return L->deriv[((size_t)y * L->w + x) * 2 + c];
}

#define DESCALE(v, n) (((v) + (1 << ((n) - 1))) >> (n))

/* calcOpticalFlowPyrLK without initial flow; status as OpenCV (1 = tracked).  The err output the Python
 * binding always requests only matters through its side effect on status (final window out of range). */
void vo_lk_track(const uint8_t* prev, const uint8_t* next, int h, int w, const float* pts, int count, int win,
                 int max_level, int max_count, double epsilon, float* out_pts, uint8_t* status)
{
    lk_level P[LK_MAX_LEVELS], N[LK_MAX_LEVELS];
    if (max_level > LK_MAX_LEVELS - 1) max_level = LK_MAX_LEVELS - 1;
    const int lp = build_pyramid(prev, h, w, win, max_level, P, 1);
    const int ln = build_pyramid(next, h, w, win, max_level, N, 0);
    const int levels = lp < ln ? lp : ln;
    max_count = max_count < 0 ? 0 : (max_count > 100 ? 100 : max_count);
    epsilon = epsilon < 0. ? 0. : (epsilon > 10. ? 10. : epsilon);
    epsilon *= epsilon;
    const float half = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int W_BITS = 14;
    for (int i = 0; i < count; i++) status[i] = 1;
    for (int level = levels; level >= 0; level--) {
        const lk_level* I = &P[level]; const lk_level* J = &N[level];
#pragma omp parallel
        {
        short* Iw = (short*)malloc(sizeof(short) * 3 * (size_t)win * win);
#pragma omp for schedule(dynamic, 8)
        for (int p = 0; p < count; p++) {
            const float sc = (float)(1. / (1 << level));
            float px = pts[p * 2] * sc, py = pts[p * 2 + 1] * sc;
            float nx, ny;
            if (level == levels) { nx = px; ny = py; }
            else { nx = out_pts[p * 2] * 2.f; ny = out_pts[p * 2 + 1] * 2.f; }
            out_pts[p * 2] = nx; out_pts[p * 2 + 1] = ny;
            px -= half; py -= half;
            const int ipx = vo_floor_f(px), ipy = vo_floor_f(py);
            if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
                if (level == 0) status[p] = 0;
                continue;
            }
            float a = px - ipx, b = py - ipy;
            int iw00 = vo_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = vo_round_f(a * (1.f - b) * (1 << W_BITS));
            int iw10 = vo_round_f((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            long long sA11 = 0, sA12 = 0, sA22 = 0;
            for (int y = 0; y < win; y++)
                for (int x = 0; x < win; x++) {
                    const int X = ipx + x, Y = ipy + y;
                    const int ival = DESCALE(img_at(I, X, Y) * iw00 + img_at(I, X + 1, Y) * iw01 + img_at(I, X, Y + 1) * iw10 + img_at(I, X + 1, Y + 1) * iw11, W_BITS - 5);
                    const int ixv = DESCALE(der_at(I, X, Y, 0) * iw00 + der_at(I, X + 1, Y, 0) * iw01 + der_at(I, X, Y + 1, 0) * iw10 + der_at(I, X + 1, Y + 1, 0) * iw11, W_BITS);
                    const int iyv = DESCALE(der_at(I, X, Y, 1) * iw00 + der_at(I, X + 1, Y, 1) * iw01 + der_at(I, X, Y + 1, 1) * iw10 + der_at(I, X + 1, Y + 1, 1) * iw11, W_BITS);
                    short* o = Iw + ((size_t)y * win + x) * 3;
                    o[0] = (short)ival; o[1] = (short)ixv; o[2] = (short)iyv;
                    sA11 += (long long)ixv * ixv; sA12 += (long long)ixv * iyv; sA22 += (long long)iyv * iyv;
                }
            const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
            if (min_eig < 1e-4f || D < FLT_EPSILON) {
                if (level == 0) status[p] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= half; ny -= half;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < max_count; j++) {
                const int inx = vo_floor_f(nx), iny = vo_floor_f(ny);
                if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
                    if (level == 0) status[p] = 0;
                    break;
                }
                a = nx - inx; b = ny - iny;
                iw00 = vo_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = vo_round_f(a * (1.f - b) * (1 << W_BITS));
                iw10 = vo_round_f((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                long long sb1 = 0, sb2 = 0;
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++) {
                        const int X = inx + x, Y = iny + y;
                        const short* o = Iw + ((size_t)y * win + x) * 3;
                        const int diff = DESCALE(img_at(J, X, Y) * iw00 + img_at(J, X + 1, Y) * iw01 + img_at(J, X, Y + 1) * iw10 + img_at(J, X + 1, Y + 1) * iw11, W_BITS - 5) - o[0];
                        sb1 += (long long)diff * o[1]; sb2 += (long long)diff * o[2];
                    }
                const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
                const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
                nx += dx; ny += dy;
                out_pts[p * 2] = nx + half; out_pts[p * 2 + 1] = ny + half;
                if ((double)dx * dx + (double)dy * dy <= epsilon) break;
                if (j > 0 && fabsf(dx + pdx) < 0.01 && fabsf(dy + pdy) < 0.01) {
                    out_pts[p * 2] -= dx * 0.5f; out_pts[p * 2 + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
            if (status[p] && level == 0) {   /* err block of the tracker: final window position must be in range */
                const float fx = out_pts[p * 2] - half, fy = out_pts[p * 2 + 1] - half;
                const int ix = vo_floor_f(fx), iy = vo_floor_f(fy);
                if (ix < -win || ix >= J->w || iy < -win || iy >= J->h) status[p] = 0;
            }
        }
        free(Iw);
        }
    }
    for (int l = 0; l <= lp; l++) { free(P[l].img); free(P[l].deriv); }
    for (int l = 0; l <= ln; l++) { free(N[l].img); }
}
