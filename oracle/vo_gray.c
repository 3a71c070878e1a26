/*
 * vo_gray.c -- CPU oracle for F2 (TEST INFRASTRUCTURE, see vo_common.h): grayscale,
 * INTER_AREA downscale, plus the bilinear float resize DIS uses for its flow fields.
 *
 * Reference call sites:
 *   nodes/stabilizer_utils.py:236-242  gray = cv2.cvtColor(frame, COLOR_RGB2GRAY)   (f32)
 *                                      np.clip(gray*255.0, 0, 255).astype(np.uint8) (truncation)
 *   nodes/stabilizer_utils.py:271-276  cv2.resize(gray, working_size, INTER_AREA)
 * Algorithm source (published OpenCV 4.x, restated from memory -- parity UNPINNED):
 *   imgproc/src/color_rgb.simd.hpp RGB2Gray<float>: coefficients 0.299f,0.587f,0.114f;
 *       vector body d = fma(c2, k2, fma(c1, k1, c0*k0)) (AVX2 dispatch, 8 lanes), scalar tail
 *       c0*k0 + c1*k1 + c2*k2 for the last (width % 8) pixels of every row.
 *   imgproc/src/resize.cpp: INTER_AREA integer-ratio fast path (2x2: (a+b+c+d+2)>>2; k x k:
 *       saturate_cast<uchar>(sum * (1.f/(k*k)))) and the general DecimateAlpha path
 *       (computeResizeAreaTab + ResizeArea_Invoker, f32 accumulation);
 *       INTER_LINEAR for f32 (HResizeLinear / VResizeLinear, 2 taps, f32).
 */
#include "vo_common.h"
#include "vstab_oracle.h"

/* cv2.cvtColor(frame f32, COLOR_RGB2GRAY) on its own (f32 out): what the cv2 stand-in of
 * tests/golden/cv2_standin.py hands to the reference's `np.clip(gray * 255.0, 0, 255).astype(np.uint8)` */
void vo_rgb2gray_f32(const float* rgb, int h, int w, int fused_body, float* gray)
{
    const float k0 = 0.299f, k1 = 0.587f, k2 = 0.114f;
    const int body = fused_body ? (w & ~7) : 0;
    for (int y = 0; y < h; y++) {
        const float* s = rgb + (size_t)y * w * 3;
        float* d = gray + (size_t)y * w;
        for (int x = 0; x < w; x++, s += 3)
            d[x] = (x < body) ? fmaf(s[2], k2, fmaf(s[1], k1, s[0] * k0)) : s[0] * k0 + s[1] * k1 + s[2] * k2;
    }
}

/* F0's range sniff, `float(arr.max())` per frame (stabilizer_utils.py:127-131): NaN-propagating like numpy's max */
void vo_frame_max(const float* frames, int n, long long per_frame, float* out)
{
#pragma omp parallel for schedule(dynamic)
    for (int f = 0; f < n; f++) {
        const float* p = frames + (size_t)f * per_frame;
        float m = -INFINITY;
        int nan = 0;
        for (long long k = 0; k < per_frame; k++) {
            const float v = p[k];
            nan |= (v != v);
            m = v > m ? v : m;
        }
        out[f] = nan ? NAN : m;
    }
}

void vo_rgb2gray_u8(const float* rgb, int h, int w, int fused_body, uint8_t* gray)
{
    const float k0 = 0.299f, k1 = 0.587f, k2 = 0.114f;
    const int body = fused_body ? (w & ~7) : 0;
    for (int y = 0; y < h; y++) {
        const float* s = rgb + (size_t)y * w * 3;
        uint8_t* d = gray + (size_t)y * w;
        for (int x = 0; x < w; x++, s += 3) {
            float g;
            if (x < body)
                g = fmaf(s[2], k2, fmaf(s[1], k1, s[0] * k0));
            else
                g = s[0] * k0 + s[1] * k1 + s[2] * k2;
            float v = g * 255.0f;
            v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
            d[x] = (uint8_t)v; /* astype(uint8): truncation */
        }
    }
}

typedef struct { int si, di; float alpha; } DecimateAlpha;

static int area_tab(int ssize, int dsize, double scale, DecimateAlpha* tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale;
        double fsx2 = fsx1 + scale;
        double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = vo_ceil_d(fsx1), sx2 = vo_floor_d(fsx2);
        sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
        sx1 = sx1 < sx2 ? sx1 : sx2;
        if (sx1 - fsx1 > 1e-3) {
            tab[k].di = dx;
            tab[k].si = sx1 - 1;
            tab[k++].alpha = (float)((sx1 - fsx1) / cellWidth);
        }
        for (int sx = sx1; sx < sx2; sx++) {
            tab[k].di = dx;
            tab[k].si = sx;
            tab[k++].alpha = (float)(1.0 / cellWidth);
        }
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2;
            a = a < 1. ? a : 1.;
            a = a < cellWidth ? a : cellWidth;
            tab[k].di = dx;
            tab[k].si = sx2;
            tab[k++].alpha = (float)(a / cellWidth);
        }
    }
    return k;
}

void vo_resize_area_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw)
{
    if (sh == dh && sw == dw) {
        memcpy(dst, src, (size_t)sh * sw);
        return;
    }
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    const int iscale_x = vo_round_d(scale_x), iscale_y = vo_round_d(scale_y);
    const int fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (fast) {
        if (iscale_x == 2 && iscale_y == 2) {
            for (int y = 0; y < dh; y++) {
                const uint8_t* s0 = src + (size_t)(2 * y) * sw;
                const uint8_t* s1 = s0 + sw;
                for (int x = 0; x < dw; x++)
                    dst[(size_t)y * dw + x] =
                        (uint8_t)((s0[2 * x] + s0[2 * x + 1] + s1[2 * x] + s1[2 * x + 1] + 2) >> 2);
            }
            return;
        }
        const float scale = 1.f / (iscale_x * iscale_y);
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++) {
                int sum = 0;
                for (int j = 0; j < iscale_y; j++)
                    for (int i = 0; i < iscale_x; i++)
                        sum += src[(size_t)(y * iscale_y + j) * sw + x * iscale_x + i];
                dst[(size_t)y * dw + x] = vo_sat_u8_f(sum * scale);
            }
        return;
    }
    /* general area path (requires scale >= 1 on both axes, as INTER_AREA decimation does) */
    DecimateAlpha* xtab = (DecimateAlpha*)malloc(sizeof(DecimateAlpha) * (size_t)(sw + sh) * 2);
    DecimateAlpha* ytab = xtab + (size_t)sw * 2;
    const int xn = area_tab(sw, dw, scale_x, xtab);
    const int yn = area_tab(sh, dh, scale_y, ytab);
    float* buf = (float*)malloc(sizeof(float) * (size_t)dw * 2);
    float* sum = buf + dw;
    for (int x = 0; x < dw; x++) sum[x] = 0.f;
    int prev_dy = ytab[0].di;
    for (int j = 0; j < yn; j++) {
        const float beta = ytab[j].alpha;
        const int dy = ytab[j].di, sy = ytab[j].si;
        const uint8_t* S = src + (size_t)sy * sw;
        for (int x = 0; x < dw; x++) buf[x] = 0.f;
        for (int k = 0; k < xn; k++) buf[xtab[k].di] += S[xtab[k].si] * xtab[k].alpha;
        if (dy != prev_dy) {
            for (int x = 0; x < dw; x++) {
                dst[(size_t)prev_dy * dw + x] = vo_sat_u8_f(sum[x]);
                sum[x] = beta * buf[x];
            }
            prev_dy = dy;
        } else {
            for (int x = 0; x < dw; x++) sum[x] += beta * buf[x];
        }
    }
    for (int x = 0; x < dw; x++) dst[(size_t)prev_dy * dw + x] = vo_sat_u8_f(sum[x]);
    free(buf);
    free(xtab);
}

/* cv::resize(..., INTER_LINEAR) for CV_32FC{cn}: 2-tap horizontal then 2-tap vertical, f32 */
void vo_resize_linear_f32(const float* src, int sh, int sw, int cn, float* dst, int dh, int dw)
{
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
    float* xa = (float*)malloc(sizeof(float) * (size_t)dw * 2);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = vo_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        xa[dx * 2] = 1.f - fx;
        xa[dx * 2 + 1] = fx;
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = vo_floor_f(fy);
        fy -= sy;
        /* vertical: rows are clipped (sy, sy+1 clamped into the image), weights kept */
        int sy0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
        const float b0 = 1.f - fy, b1 = fy;
        const float* S0 = src + (size_t)sy0 * sw * cn;
        const float* S1 = src + (size_t)sy1 * sw * cn;
        float* D = dst + (size_t)dy * dw * cn;
        for (int dx = 0; dx < dw; dx++) {
            const int sx = xofs[dx];
            const float a0 = xa[dx * 2], a1 = xa[dx * 2 + 1];
            const int sx1 = sx + 1 < sw ? sx + 1 : sx; /* a1 == 0 whenever sx is the last column */
            for (int c = 0; c < cn; c++) {
                float r0 = S0[sx * cn + c] * a0 + S0[sx1 * cn + c] * a1;
                float r1 = S1[sx * cn + c] * a0 + S1[sx1 * cn + c] * a1;
                D[dx * cn + c] = r0 * b0 + r1 * b1;
            }
        }
    }
    free(xofs);
    free(xa);
}

void vo_gray_for_estimation(const float* rgb, int n, int h, int w, int wh, int ww, int fused_body,
                            uint8_t* out)
{
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < n; i++) {
        uint8_t* g = (uint8_t*)malloc((size_t)h * w);
        vo_rgb2gray_u8(rgb + (size_t)i * h * w * 3, h, w, fused_body, g);
        vo_resize_area_u8(g, h, w, out + (size_t)i * wh * ww, wh, ww);
        free(g);
    }
}
