/*
 * vo_crop.c -- CPU oracle for the coverage analysis behind framing_mode="crop" (TEST INFRASTRUCTURE).
 * Restates nodes/stabilizer_utils.py:611-643 (warpPerspective(ones, INTER_NEAREST) > 0.5, cv2.dilate 3x3,
 * cv2.erode 3x3, bounding box) and :763-787 (AND over frames, cv2.erode 3x3).  cv2.dilate / cv2.erode
 * with the default border: out-of-image pixels never win the max (dilate) nor the min (erode).
 */
#include "vo_common.h"
#include "vstab_oracle.h"

static void morph3(const uint8_t* src, uint8_t* dst, int h, int w, int dilate)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v = dilate ? 0 : 1;
            for (int ey = -1; ey <= 1; ey++)
                for (int ex = -1; ex <= 1; ex++) {
                    int qy = y + ey, qx = x + ex;
                    if (qy < 0 || qy >= h || qx < 0 || qx >= w) continue;
                    if (dilate) v |= src[(size_t)qy * w + qx];
                    else v &= src[(size_t)qy * w + qx];
                }
            dst[(size_t)y * w + x] = (uint8_t)v;
        }
}

void vo_crop_analysis(const float* matrices, int n, int sh, int sw, int oh, int ow, int32_t* bbox, uint8_t* common)
{
    const size_t npx = (size_t)oh * ow;
    float* cov = (float*)malloc(sizeof(float) * npx);
    uint8_t* bin = (uint8_t*)malloc(npx);
    uint8_t* t1 = (uint8_t*)malloc(npx);
    uint8_t* t2 = (uint8_t*)malloc(npx);
    uint8_t* all = (uint8_t*)malloc(npx);
    memset(all, 1, npx);
    const float border[3] = {0, 0, 0};
    for (int f = 0; f < n; f++) {
        vo_warp_frame(0, sh, sw, matrices + (size_t)f * 9, oh, ow, VO_INTERP_BILINEAR, border, VO_SUBPIX_Q5, 0, cov);
        for (size_t p = 0; p < npx; p++) { bin[p] = cov[p] > 0.5f; all[p] &= bin[p]; }
        morph3(bin, t1, oh, ow, 1);
        morph3(t1, t2, oh, ow, 0);
        int x0 = INT_MAX, y0 = INT_MAX, x1 = -1, y1 = -1;
        for (int y = 0; y < oh; y++)
            for (int x = 0; x < ow; x++)
                if (t2[(size_t)y * ow + x]) {
                    if (x < x0) x0 = x;
                    if (y < y0) y0 = y;
                    if (x > x1) x1 = x;
                    if (y > y1) y1 = y;
                }
        if (x1 < 0) { bbox[f * 4] = bbox[f * 4 + 1] = bbox[f * 4 + 2] = bbox[f * 4 + 3] = -1; }
        else { bbox[f * 4] = x0; bbox[f * 4 + 1] = y0; bbox[f * 4 + 2] = x1; bbox[f * 4 + 3] = y1; }
    }
    morph3(all, common, oh, ow, 0);
    free(cov); free(bin); free(t1); free(t2); free(all);
}
