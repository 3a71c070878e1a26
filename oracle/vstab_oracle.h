/*
 * vstab_oracle.h -- entry points of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 * See vo_common.h for what the oracle is and what pins it.  Product code must
 * never include, link or load anything from oracle/.
 */
#ifndef VSTAB_ORACLE_H
#define VSTAB_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { VO_INTERP_BILINEAR = 0, VO_INTERP_BICUBIC = 1 };
enum { VO_SUBPIX_Q5 = 0, VO_SUBPIX_EXACT = 1 };
enum { VO_MODE_TRANSLATION = 0, VO_MODE_SIMILARITY = 1, VO_MODE_PERSPECTIVE = 2 };

/* ---- warp (vo_warp.c) ---- */
int vo_invert3x3(const double* S, double* D);
void vo_interp_tables(float* lin, float* cub);
void vo_warp_frame(const float* src, int sh, int sw, const float* M32, int dh, int dw, int interp,
                   const float* border, int subpix, float* dst, float* coverage);
void vo_warp_frame_inv(const float* src, int sh, int sw, const double* Minv, int dh, int dw,
                       int interp, const float* border, int subpix, float* dst, float* coverage);
void vo_warp_clip(const float* src, int n, int sh, int sw, const float* M32, int dh, int dw,
                  int interp, const float* border, int subpix, float* dst, float* mask,
                  unsigned* pad_count);
void vo_linspace(double a, double b, int n, double* out);
void vo_blur_sample_matrices(const double* matrices, int n, int idx, double blur, int samples,
                             float* out32);
void vo_warp_blur_clip(const float* src, int n, int sh, int sw, const double* matrices, int dh,
                       int dw, int interp, const float* border, int subpix, double blur,
                       int samples, float* dst, float* mask);

/* ---- crop coverage analysis (vo_crop.c) ---- */
void vo_crop_analysis(const float* matrices, int n, int sh, int sw, int oh, int ow, int32_t* bbox, uint8_t* common);

/* ---- gray + resize (vo_gray.c) ---- */
void vo_rgb2gray_u8(const float* rgb, int h, int w, int fused_body, uint8_t* gray);
void vo_frame_max(const float* frames, int n, long long per_frame, float* out);
void vo_rgb2gray_f32(const float* rgb, int h, int w, int fused_body, float* gray);
void vo_resize_area_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw);
void vo_resize_linear_f32(const float* src, int sh, int sw, int cn, float* dst, int dh, int dw);
void vo_gray_for_estimation(const float* rgb, int n, int h, int w, int wh, int ww, int fused_body,
                            uint8_t* out);

/* ---- DIS optical flow (vo_dis.c) ---- */
typedef struct vo_dis_params {
    int finest_scale;      /* 2  (flow.py:83) */
    int patch_size;        /* 8  (flow.py:84) */
    int patch_stride;      /* 4  (flow.py:85) */
    int grad_descent_iter; /* 25 (PRESET_MEDIUM) */
    int var_iter;          /* 5  (PRESET_MEDIUM) */
    float alpha, delta, gamma; /* 20, 5, 10 */
    int use_mean_norm;     /* 1 */
    int use_spatial_prop;  /* 1 (flow.py:86) */
} vo_dis_params;
void vo_dis_default_params(vo_dis_params* p);
/* I0, I1: u8 [h,w]; flow out: f32 [h,w,2] */
int vo_dis_calc(const uint8_t* I0, const uint8_t* I1, int h, int w, const vo_dis_params* p,
                float* flow);
int vo_dis_calc_stateful(const uint8_t* I0, const uint8_t* I1, int h, int w, vo_dis_params* p, float* flow);
/* batch over consecutive pairs of a gray clip [n,h,w]; flow [n-1,h,w,2] */
int vo_dis_calc_clip(const uint8_t* gray, int n, int h, int w, const vo_dis_params* p, float* flow);
int vo_dis_coarsest_scale(int h, int w, int patch_size);
/* association of the four 8x8 patch sums: 1 = OpenCV's SIMD128 row accumulators (default; what the HIP kernel does),
 * 0 = the XOR butterfly of round 1 (deviation measurement only).  Process-global; set before calling vo_dis_calc*. */
void vo_dis_set_sum_order(int order);
int vo_dis_get_sum_order(void);
/* test-only: Sobel gradients (s16) and the five structure-tensor planes [5][hs][ws] (xx, yy, xy, x, y) of one level image */
void vo_dis_gradients(const uint8_t* I, int h, int w, int psz, int pstr, short* Ix, short* Iy, float* tensor);
/* test-only: the patch inverse search of one level (sparse flow [hs][ws] from a dense initial flow [h][w]) */
int vo_dis_patch_search_debug(const uint8_t* I0, const uint8_t* I1, int h, int w, const float* Ux, const float* Uy, float* Sx, float* Sy,
                              const vo_dis_params* p);
/* test-only: variational refinement of a given flow on one level with `sor_iters` SOR iterations per fixed-point iteration */
int vo_variational_refine_debug(const uint8_t* I0, const uint8_t* I1, int h, int w, float* U, float* V,
                                const vo_dis_params* p, int sor_iters);

/* ---- sampling + model fit (vo_fit.c) ---- */
typedef struct vo_fit_result {
    float matrix[9];
    int mode;          /* VO_MODE_* actually used; identity fallback reports translation */
    double confidence;
    double residual;
    int valid;         /* 1 if this mode's acceptance test passed */
} vo_fit_result;
/* flow [h,w,2]; restates flow.py:141-210 starting from requested mode */
void vo_fit_from_flow(const float* flow, int h, int w, int step, int requested_mode,
                      vo_fit_result* out);
void vo_fit_all_modes(const float* flow, int h, int w, int step, int requested_mode,
                      vo_fit_result* rec /*3, indexed by mode; rec[m].mode == -1: not computed*/,
                      int* valid_points, int* total_points);
void vo_fit_all_modes_points(const float* from, const float* to, const uint8_t* status, int count, int requested_mode,
                             vo_fit_result* rec /*3*/, int* valid_points);
int vo_estimate_affine_partial2d(const float* from, const float* to, int count, double thresh,
                                 int max_iters, double confidence, int refine_iters,
                                 double* M /*2x3*/, uint8_t* inliers);
int vo_find_homography_ransac(const float* from, const float* to, int count, double thresh,
                              int max_iters, double confidence, double* H /*3x3*/,
                              uint8_t* inliers);

/* ---- Classic estimator: GFTT + pyramidal LK (vo_classic.c) ---- */
void vo_min_eigen_val(const uint8_t* img, int h, int w, int block, float* eig);
int vo_good_features(const uint8_t* img, int h, int w, int max_corners, double quality, double min_distance,
                     int block, float* corners /*[max_corners][2]*/);
void vo_pyr_down_u8(const uint8_t* src, int sh, int sw, uint8_t* dst /*[(sh+1)/2][(sw+1)/2]*/);
void vo_scharr_deriv(const uint8_t* src, int h, int w, short* deriv /*[h][w][2]*/);
int vo_lk_levels(int h, int w, int win, int max_level);
void vo_lk_track(const uint8_t* prev, const uint8_t* next, int h, int w, const float* pts, int count, int win,
                 int max_level, int max_count, double epsilon, float* out_pts, uint8_t* status);

/* ---- fallback estimator: phase correlation (vo_phase.c) ---- */
int vo_optimal_dft_size(int n);
void vo_phase_spectrum(const uint8_t* img, int h, int w, float* spectrum /*[M][N/2+1][2]*/);
void vo_phase_correlate_clip(const uint8_t* gray, int n, int h, int w, double* shifts /*[n-1][3]*/,
                             float* surface /*[M][N] of pair 0 or NULL*/);

#ifdef __cplusplus
}
#endif
#endif
