/*
 * vo_dis.c -- CPU oracle for F3, DIS dense optical flow (TEST INFRASTRUCTURE, see vo_common.h).
 *
 * Reference call site: nodes/video_stabilizer_flow.py:82-86 (DISOpticalFlow PRESET_MEDIUM,
 * finestScale 2, patchSize 8, patchStride 4, spatial propagation) and :140 (calc(prev,curr,None)).
 *
 * Algorithm source: OpenCV 4.x modules/video/src/dis_flow.cpp and variational_refinement.cpp,
 * restated from the published algorithm (Kroeger et al., "Fast Optical Flow using Dense Inverse
 * Search", ECCV 2016, as implemented by OpenCV).  No OpenCV exists in this container: parity
 * with a real cv2 is UNPINNED.  Stages, in OpenCV's order:
 *   prepareBuffers            INTER_AREA pyramid (finest = size/4, then /2 per level), I1 replicate-
 *                             padded by 16, spatialGradient (3x3 Sobel, s16, reflect-101)
 *   precomputeStructureTensor separable 8-tap running box sums of Ix^2, Iy^2, IxIy, Ix, Iy (f32)
 *   PatchInverseSearch        8 fixed stripes, 2 passes (forward / backward raster), spatial
 *                             propagation from the already-visited neighbours, <= 12 inverse-
 *                             compositional steps per pass, mean-normalised SSD
 *   Densification             per-pixel weighted mean over overlapping patches
 *   VariationalRefinement     5 fixed-point iterations x 5 red-black SOR sweeps (omega 1.6),
 *                             alpha 20, delta 5, gamma 10, zeta 0.1, epsilon 0.001
 *   upsample                  bilinear resize x2 between levels, final resize to full size, x4
 *
 * The four per-patch sums (sum d, sum d^2, sum d*Ix, sum d*Iy over the 8x8 patch) are reduced in OpenCV's own f32
 * association -- the 4-lane row accumulators + horizontal add of the CV_SIMD128 branch (opencv_rows4_sum below); the
 * HIP kernel reproduces it lane for lane.  (Round 1 shipped a 6-level XOR butterfly over lane = row*8 + col instead;
 * it was measured outside the flow bound in round 2 and is kept behind vo_dis_set_sum_order(0) only so that
 * tests/test_dis_sum_order_cpu.py can keep showing what it cost.)  No deliberate deviation from OpenCV's operation
 * order remains.
 */
#include "vo_common.h"
#include "vstab_oracle.h"
#include <stdio.h>

#define DIS_EPS 0.001f
#define DIS_INF 1e10f
#define DIS_BORDER 16
#define MAX_LEVELS 16

void vo_dis_default_params(vo_dis_params* p)
{
    p->finest_scale = 2;
    p->patch_size = 8;
    p->patch_stride = 4;
    p->grad_descent_iter = 25;
    p->var_iter = 5;
    p->alpha = 20.0f;
    p->delta = 5.0f;
    p->gamma = 10.0f;
    p->use_mean_norm = 1;
    p->use_spatial_prop = 1;
}

int vo_dis_coarsest_scale(int h, int w, int patch_size)
{
    int mx = w > h ? w : h, mn = w < h ? w : h;
    int a = (int)(log(mx / (4.0 * patch_size)) / log(2.0) + 0.5);
    int b = (int)(log((double)(mn / patch_size)) / log(2.0)); /* integer division, as OpenCV */
    return a < b ? a : b;
}

typedef struct {
    int w, h, ws, hs;
    uint8_t* I;     /* h*w */
    uint8_t* Iext;  /* (h+32)*(w+32) replicate padded */
    short *Ix, *Iy; /* Sobel */
    float *xx, *yy, *xy, *sx, *sy; /* structure tensor, hs*ws */
} Level;

static void level_free(Level* L)
{
    free(L->I); free(L->Iext); free(L->Ix); free(L->Iy);
    free(L->xx); free(L->yy); free(L->xy); free(L->sx); free(L->sy);
    memset(L, 0, sizeof(*L));
}

static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

static void sobel_s16(const uint8_t* I, int h, int w, short* Ix, short* Iy)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* r0 = I + (size_t)reflect101(y - 1, h) * w;
        const uint8_t* r1 = I + (size_t)y * w;
        const uint8_t* r2 = I + (size_t)reflect101(y + 1, h) * w;
        for (int x = 0; x < w; x++) {
            int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
            int gx = (r0[xr] + 2 * r1[xr] + r2[xr]) - (r0[xl] + 2 * r1[xl] + r2[xl]);
            int gy = (r2[xl] + 2 * r2[x] + r2[xr]) - (r0[xl] + 2 * r0[x] + r0[xr]);
            Ix[(size_t)y * w + x] = (short)gx;
            Iy[(size_t)y * w + x] = (short)gy;
        }
    }
}

static void pad_replicate(const uint8_t* I, int h, int w, int b, uint8_t* out)
{
    const int we = w + 2 * b;
    for (int y = 0; y < h + 2 * b; y++) {
        int sy = y - b;
        sy = sy < 0 ? 0 : (sy >= h ? h - 1 : sy);
        for (int x = 0; x < we; x++) {
            int sx = x - b;
            sx = sx < 0 ? 0 : (sx >= w ? w - 1 : sx);
            out[(size_t)y * we + x] = I[(size_t)sy * w + sx];
        }
    }
}

/* precomputeStructureTensor: running sums exactly in OpenCV's order */
static void structure_tensor(Level* L, int psz, int pstr)
{
    const int w = L->w, h = L->h, ws = L->ws, hs = L->hs;
    float* aux[5];
    for (int k = 0; k < 5; k++) aux[k] = (float*)calloc((size_t)h * ws, sizeof(float));
    for (int i = 0; i < h; i++) {
        float s_xx = 0.f, s_yy = 0.f, s_xy = 0.f, s_x = 0.f, s_y = 0.f;
        const short* xr = L->Ix + (size_t)i * w;
        const short* yr = L->Iy + (size_t)i * w;
        for (int j = 0; j < psz; j++) {
            s_xx += xr[j] * xr[j];
            s_yy += yr[j] * yr[j];
            s_xy += xr[j] * yr[j];
            s_x += xr[j];
            s_y += yr[j];
        }
        aux[0][(size_t)i * ws] = s_xx; aux[1][(size_t)i * ws] = s_yy; aux[2][(size_t)i * ws] = s_xy;
        aux[3][(size_t)i * ws] = s_x; aux[4][(size_t)i * ws] = s_y;
        int js = 1;
        for (int j = psz; j < w; j++) {
            s_xx += (xr[j] * xr[j] - xr[j - psz] * xr[j - psz]);
            s_yy += (yr[j] * yr[j] - yr[j - psz] * yr[j - psz]);
            s_xy += (xr[j] * yr[j] - xr[j - psz] * yr[j - psz]);
            s_x += (xr[j] - xr[j - psz]);
            s_y += (yr[j] - yr[j - psz]);
            if ((j - psz + 1) % pstr == 0) {
                aux[0][(size_t)i * ws + js] = s_xx; aux[1][(size_t)i * ws + js] = s_yy;
                aux[2][(size_t)i * ws + js] = s_xy; aux[3][(size_t)i * ws + js] = s_x;
                aux[4][(size_t)i * ws + js] = s_y;
                js++;
            }
        }
    }
    float* dst[5] = {L->xx, L->yy, L->xy, L->sx, L->sy};
    for (int k = 0; k < 5; k++) {
        float* sum = (float*)calloc((size_t)ws, sizeof(float));
        for (int i = 0; i < psz; i++)
            for (int j = 0; j < ws; j++) sum[j] += aux[k][(size_t)i * ws + j];
        for (int j = 0; j < ws; j++) dst[k][j] = sum[j];
        int is = 1;
        for (int i = psz; i < h; i++) {
            for (int j = 0; j < ws; j++)
                sum[j] += (aux[k][(size_t)i * ws + j] - aux[k][(size_t)(i - psz) * ws + j]);
            if ((i - psz + 1) % pstr == 0) {
                for (int j = 0; j < ws; j++) dst[k][(size_t)is * ws + j] = sum[j];
                is++;
            }
        }
        free(sum);
        free(aux[k]);
    }
}

/* Test-only view of prepareBuffers' per-level products for ONE image (tests/test_referee_cpu.py compares them with an
 * independent torch.nn.functional.conv2d evaluation): Sobel gradients and, if `tensor` is given, the five structure-tensor
 * planes [5][hs][ws] in the order xx, yy, xy, x, y. */
void vo_dis_gradients(const uint8_t* I, int h, int w, int psz, int pstr, short* Ix, short* Iy, float* tensor)
{
    sobel_s16(I, h, w, Ix, Iy);
    if (!tensor) return;
    Level L;
    memset(&L, 0, sizeof(L));
    L.w = w; L.h = h;
    L.ws = 1 + (w - psz) / pstr;
    L.hs = 1 + (h - psz) / pstr;
    L.Ix = Ix; L.Iy = Iy;
    const size_t ns = (size_t)L.ws * L.hs;
    L.xx = tensor; L.yy = tensor + ns; L.xy = tensor + 2 * ns; L.sx = tensor + 3 * ns; L.sy = tensor + 4 * ns;
    structure_tensor(&L, psz, pstr);
}

/* pyramid of one frame: levels [finest, coarsest] */
static void build_levels(const uint8_t* gray, int h, int w, const vo_dis_params* p, int coarsest,
                         Level* L /* indexed by scale */, int want_grad, int want_ext)
{
    int fraction = 1;
    int cur_h = 0, cur_w = 0;
    for (int i = 0; i <= coarsest; i++) {
        if (i == p->finest_scale) {
            cur_h = h / fraction;
            cur_w = w / fraction;
            L[i].I = (uint8_t*)malloc((size_t)cur_h * cur_w);
            vo_resize_area_u8(gray, h, w, L[i].I, cur_h, cur_w);
        } else if (i > p->finest_scale) {
            int ph = cur_h, pw = cur_w;
            cur_h = ph / 2;
            cur_w = pw / 2;
            L[i].I = (uint8_t*)malloc((size_t)cur_h * cur_w);
            vo_resize_area_u8(L[i - 1].I, ph, pw, L[i].I, cur_h, cur_w);
        }
        if (i >= p->finest_scale) {
            L[i].h = cur_h;
            L[i].w = cur_w;
            L[i].ws = 1 + (cur_w - p->patch_size) / p->patch_stride;
            L[i].hs = 1 + (cur_h - p->patch_size) / p->patch_stride;
            if (want_ext) {
                L[i].Iext = (uint8_t*)malloc((size_t)(cur_h + 2 * DIS_BORDER) * (cur_w + 2 * DIS_BORDER));
                pad_replicate(L[i].I, cur_h, cur_w, DIS_BORDER, L[i].Iext);
            }
            if (want_grad) {
                L[i].Ix = (short*)malloc(sizeof(short) * (size_t)cur_h * cur_w);
                L[i].Iy = (short*)malloc(sizeof(short) * (size_t)cur_h * cur_w);
                sobel_s16(L[i].I, cur_h, cur_w, L[i].Ix, L[i].Iy);
                size_t ns = (size_t)L[i].ws * L[i].hs;
                L[i].xx = (float*)malloc(sizeof(float) * ns);
                L[i].yy = (float*)malloc(sizeof(float) * ns);
                L[i].xy = (float*)malloc(sizeof(float) * ns);
                L[i].sx = (float*)malloc(sizeof(float) * ns);
                L[i].sy = (float*)malloc(sizeof(float) * ns);
                structure_tensor(&L[i], p->patch_size, p->patch_stride);
            }
        }
        fraction *= 2;
    }
}

/* 64-element XOR-butterfly sum (wavefront shuffle reduction order; round-1 association, measurement only) */
static inline float butterfly64(float* v)
{
    float t[64];
    for (int s = 1; s < 64; s <<= 1) {
        for (int l = 0; l < 64; l++) t[l] = v[l] + v[l ^ s];
        memcpy(v, t, sizeof(t));
    }
    return v[0];
}

/* OpenCV's own association for patch size 8 (video/src/dis_flow.cpp, processPatchMeanNorm / computeSSDMeanNorm,
 * CV_SIMD128 branch): one v_float32x4 accumulator per sum; every row adds (left half + right half) of its 8 terms to
 * it -- lane l collects columns l and l+4, rows in order -- and v_reduce_sum folds the four lanes as
 * (a0 + a2) + (a1 + a3) (the SSE implementation: add the upper half onto the lower, then lane 1 onto lane 0).
 * The per-element terms (d, d*d, d*Ix, d*Iy) are the same f32 values in both orders. */
static inline float opencv_rows4_sum(const float* t)
{
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < 8; r++)
        for (int l = 0; l < 4; l++) acc[l] = acc[l] + (t[r * 8 + l] + t[r * 8 + l + 4]);
    return (acc[0] + acc[2]) + (acc[1] + acc[3]);
}

/* 1 (default): OpenCV's 4-lane row accumulators -- the definition; the HIP kernel reproduces it lane for lane.
 * 0: the XOR butterfly round 1 shipped (a plain wavefront shuffle reduction).  Kept only so that
 *    tests/test_dis_sum_order_cpu.py can keep measuring what that shortcut cost: the two orders differ in the last
 *    bit of a sum, a branch of the descent flips, and a few sampled flow vectors move by 1e-3 .. 2e-2 px. */
static int g_sum_order = 1;
void vo_dis_set_sum_order(int order) { g_sum_order = order ? 1 : 0; }
int vo_dis_get_sum_order(void) { return g_sum_order; }
static inline float patch_sum(float* v) { return g_sum_order ? opencv_rows4_sum(v) : butterfly64(v); }

typedef struct { float sum_diff, sum_sq, sum_x, sum_y; } PatchSums;

/* one evaluation of an 8x8 patch: bilinear I1 window minus I0, reduced */
static PatchSums patch_eval(const uint8_t* I0p, int s0, const uint8_t* I1p, int s1, const short* Ixp,
                            const short* Iyp, float w00, float w01, float w10, float w11, int grad)
{
    float d[64], d2[64], dx[64], dy[64];
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) {
            const uint8_t* q = I1p + (size_t)r * s1 + c;
            float v = w00 * (float)q[0] + w01 * (float)q[1] + w10 * (float)q[s1] + w11 * (float)q[s1 + 1] -
                      (float)I0p[(size_t)r * s0 + c];
            const int l = r * 8 + c;
            d[l] = v;
            d2[l] = v * v;
            if (grad) {
                dx[l] = v * (float)Ixp[(size_t)r * s0 + c];
                dy[l] = v * (float)Iyp[(size_t)r * s0 + c];
            }
        }
    PatchSums s;
    s.sum_diff = patch_sum(d);
    s.sum_sq = patch_sum(d2);
    s.sum_x = grad ? patch_sum(dx) : 0.f;
    s.sum_y = grad ? patch_sum(dy) : 0.f;
    return s;
}

typedef struct { float i_I1, j_I1, w00, w01, w10, w11; } Bilin;

static inline Bilin bilin_weights(int i, int j, float Ux, float Uy, int bsz, float i_lo, float i_hi,
                                  float j_lo, float j_hi)
{
    Bilin b;
    float ii = (float)i + Uy + (float)bsz;
    float jj = (float)j + Ux + (float)bsz;
    ii = ii > i_lo ? ii : i_lo; /* max(a, lo) */
    ii = ii < i_hi ? ii : i_hi; /* min(.., hi) */
    jj = jj > j_lo ? jj : j_lo;
    jj = jj < j_hi ? jj : j_hi;
    const float fi = floorf(ii), fj = floorf(jj);
    b.i_I1 = ii;
    b.j_I1 = jj;
    b.w11 = (ii - fi) * (jj - fj);
    b.w10 = (ii - fi) * (fj + 1 - jj);
    b.w01 = (fi + 1 - ii) * (jj - fj);
    b.w00 = (fi + 1 - ii) * (fj + 1 - jj);
    return b;
}

static void patch_inverse_search(const Level* L0, const Level* L1, const float* Ux, const float* Uy,
                                 float* Sx, float* Sy, const vo_dis_params* p)
{
    const int w = L0->w, h = L0->h, ws = L0->ws, hs = L0->hs;
    const int psz = p->patch_size, psz2 = psz / 2, pstr = p->patch_stride, bsz = DIS_BORDER;
    const int w_ext = w + 2 * bsz;
    const int nstripes = 8, num_iter = 2;
    const int stripe_sz = (int)ceil(hs / (double)nstripes);
    const float i_lo = bsz - psz + 1.0f, i_hi = bsz + h - 1.0f;
    const float j_lo = bsz - psz + 1.0f, j_hi = bsz + w - 1.0f;
    const int num_inner_iter = (int)floor(p->grad_descent_iter / (float)num_iter);
    const float n = (float)psz * psz;

    for (int stripe = 0; stripe < nstripes; stripe++) {
        for (int iter = 0; iter < num_iter; iter++) {
            int dir, start_is, end_is, start_js, end_js, start_i, start_j;
            if (iter % 2 == 0) {
                dir = 1;
                start_is = stripe * stripe_sz < hs ? stripe * stripe_sz : hs;
                end_is = (stripe + 1) * stripe_sz < hs ? (stripe + 1) * stripe_sz : hs;
                start_js = 0;
                end_js = ws;
                start_i = start_is * pstr;
                start_j = 0;
            } else {
                dir = -1;
                start_is = ((stripe + 1) * stripe_sz < hs ? (stripe + 1) * stripe_sz : hs) - 1;
                end_is = (stripe * stripe_sz < hs ? stripe * stripe_sz : hs) - 1;
                start_js = ws - 1;
                end_js = -1;
                start_i = start_is * pstr;
                start_j = (ws - 1) * pstr;
            }
            int i = start_i;
            for (int is = start_is; dir * is < dir * end_is; is += dir) {
                int j = start_j;
                for (int js = start_js; dir * js < dir * end_js; js += dir) {
                    const size_t sidx = (size_t)is * ws + js;
                    const uint8_t* I0p = L0->I + (size_t)i * w + j;
                    const short* Ixp = L0->Ix + (size_t)i * w + j;
                    const short* Iyp = L0->Iy + (size_t)i * w + j;
                    if (iter == 0) {
                        Sx[sidx] = Ux[(size_t)(i + psz2) * w + j + psz2];
                        Sy[sidx] = Uy[(size_t)(i + psz2) * w + j + psz2];
                    }
#define SSD_AT(dst, ux, uy)                                                                          \
    do {                                                                                             \
        Bilin b_ = bilin_weights(i, j, (ux), (uy), bsz, i_lo, i_hi, j_lo, j_hi);                     \
        PatchSums s_ = patch_eval(I0p, w, L1->Iext + (size_t)(int)b_.i_I1 * w_ext + (int)b_.j_I1,    \
                                  w_ext, Ixp, Iyp, b_.w00, b_.w01, b_.w10, b_.w11, 0);               \
        dst = s_.sum_sq - s_.sum_diff * s_.sum_diff / n;                                             \
    } while (0)
                    float min_SSD = DIS_INF, cur_SSD;
                    SSD_AT(min_SSD, Sx[sidx], Sy[sidx]);
                    if (dir * js > dir * start_js) {
                        SSD_AT(cur_SSD, Sx[sidx - dir], Sy[sidx - dir]);
                        if (cur_SSD < min_SSD) {
                            min_SSD = cur_SSD;
                            Sx[sidx] = Sx[sidx - dir];
                            Sy[sidx] = Sy[sidx - dir];
                        }
                    }
                    if (dir * is > dir * start_is) {
                        const size_t nidx = (size_t)(is - dir) * ws + js;
                        SSD_AT(cur_SSD, Sx[nidx], Sy[nidx]);
                        if (cur_SSD < min_SSD) {
                            min_SSD = cur_SSD;
                            Sx[sidx] = Sx[nidx];
                            Sy[sidx] = Sy[nidx];
                        }
                    }
#undef SSD_AT
                    float cur_Ux = Sx[sidx], cur_Uy = Sy[sidx];
                    float detH = L0->xx[sidx] * L0->yy[sidx] - L0->xy[sidx] * L0->xy[sidx];
                    if (fabsf(detH) < DIS_EPS) detH = DIS_EPS;
                    const float invH11 = L0->yy[sidx] / detH;
                    const float invH12 = -L0->xy[sidx] / detH;
                    const float invH22 = L0->xx[sidx] / detH;
                    float prev_SSD = DIS_INF, SSD;
                    const float x_grad_sum = L0->sx[sidx], y_grad_sum = L0->sy[sidx];
                    for (int t = 0; t < num_inner_iter; t++) {
                        Bilin b = bilin_weights(i, j, cur_Ux, cur_Uy, bsz, i_lo, i_hi, j_lo, j_hi);
                        PatchSums s = patch_eval(I0p, w, L1->Iext + (size_t)(int)b.i_I1 * w_ext + (int)b.j_I1,
                                                 w_ext, Ixp, Iyp, b.w00, b.w01, b.w10, b.w11, 1);
                        const float dUx = s.sum_x - s.sum_diff * x_grad_sum / n;
                        const float dUy = s.sum_y - s.sum_diff * y_grad_sum / n;
                        SSD = s.sum_sq - s.sum_diff * s.sum_diff / n;
                        const float dx = invH11 * dUx + invH12 * dUy;
                        const float dy = invH12 * dUx + invH22 * dUy;
                        cur_Ux -= dx;
                        cur_Uy -= dy;
                        if (SSD >= prev_SSD) break;
                        prev_SSD = SSD;
                    }
                    {
                        const double ddx = (double)(cur_Ux - Sx[sidx]), ddy = (double)(cur_Uy - Sy[sidx]);
                        if (sqrt(ddx * ddx + ddy * ddy) <= (double)psz) {
                            Sx[sidx] = cur_Ux;
                            Sy[sidx] = cur_Uy;
                        }
                    }
                    j += dir * pstr;
                }
                i += dir * pstr;
            }
        }
    }
}

static void densify(const Level* L0, const Level* L1, const float* Sx, const float* Sy, float* Ux,
                    float* Uy, const vo_dis_params* p)
{
    const int w = L0->w, h = L0->h, ws = L0->ws, hs = L0->hs;
    const int psz = p->patch_size, pstr = p->patch_stride;
    const uint8_t* I0 = L0->I;
    const uint8_t* I1 = L1->I;
    for (int i = 0; i < h; i++) {
        int end_is = i / pstr < hs - 1 ? i / pstr : hs - 1;
        int start_is = i - psz >= 0 ? (i - psz) / pstr + 1 : 0;
        if (start_is > end_is) start_is = end_is;
        for (int j = 0; j < w; j++) {
            int end_js = j / pstr < ws - 1 ? j / pstr : ws - 1;
            int start_js = j - psz >= 0 ? (j - psz) / pstr + 1 : 0;
            if (start_js > end_js) start_js = end_js;
            float sum_coef = 0.f, sum_Ux = 0.f, sum_Uy = 0.f;
            for (int is = start_is; is <= end_is; is++)
                for (int js = start_js; js <= end_js; js++) {
                    const float sxv = Sx[(size_t)is * ws + js], syv = Sy[(size_t)is * ws + js];
                    float j_m = (float)j + sxv, i_m = (float)i + syv;
                    j_m = j_m > 0.0f ? j_m : 0.0f;
                    j_m = j_m < (float)w - 1.0f - DIS_EPS ? j_m : (float)w - 1.0f - DIS_EPS;
                    i_m = i_m > 0.0f ? i_m : 0.0f;
                    i_m = i_m < (float)h - 1.0f - DIS_EPS ? i_m : (float)h - 1.0f - DIS_EPS;
                    const int j_l = (int)j_m, j_u = j_l + 1, i_l = (int)i_m, i_u = i_l + 1;
                    const float diff = (j_m - j_l) * (i_m - i_l) * I1[(size_t)i_u * w + j_u] +
                                       (j_u - j_m) * (i_m - i_l) * I1[(size_t)i_u * w + j_l] +
                                       (j_m - j_l) * (i_u - i_m) * I1[(size_t)i_l * w + j_u] +
                                       (j_u - j_m) * (i_u - i_m) * I1[(size_t)i_l * w + j_l] -
                                       I0[(size_t)i * w + j];
                    const float ad = fabsf(diff);
                    const float coef = 1 / (ad > 1.0f ? ad : 1.0f);
                    sum_Ux += coef * sxv;
                    sum_Uy += coef * syv;
                    sum_coef += coef;
                }
            Ux[(size_t)i * w + j] = sum_Ux / sum_coef;
            Uy[(size_t)i * w + j] = sum_Uy / sum_coef;
        }
    }
}

/* ---------------- variational refinement (calcUV) ---------------- */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void variational_refine_n(const uint8_t* I0, const uint8_t* I1, int h, int w, float* U, float* V,
                                 const vo_dis_params* p, int sor_iters)
{
    const size_t np = (size_t)h * w;
    const int fixed_iters = p->var_iter;
    const float omega = 1.6f, zeta = 0.1f, epsilon = 0.001f;
    const float zeta_squared = zeta * zeta, epsilon_squared = epsilon * epsilon;
    const float gamma2 = p->gamma / 2, delta2 = p->delta / 2, alpha2 = p->alpha / 4;

    float* buf = (float*)malloc(sizeof(float) * np * 24);
    float *avg = buf, *Iz = buf + np, *Ix = buf + 2 * np, *Iy = buf + 3 * np, *Ixx = buf + 4 * np,
          *Ixy = buf + 5 * np, *Iyy = buf + 6 * np, *Ixz = buf + 7 * np, *Iyz = buf + 8 * np,
          *A11 = buf + 9 * np, *A12 = buf + 10 * np, *A22 = buf + 11 * np, *b1 = buf + 12 * np,
          *b2 = buf + 13 * np, *wgt = buf + 14 * np, *tU = buf + 15 * np, *tV = buf + 16 * np,
          *dU = buf + 17 * np, *dV = buf + 18 * np;

    /* warpImage + average + temporal derivative */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t q = (size_t)y * w + x;
            const float mx = x + U[q], my = y + V[q];
            const int sx = vo_round_f(mx * 32.f), sy = vo_round_f(my * 32.f);
            const int ix = vo_sat_short(sx >> 5), iy = vo_sat_short(sy >> 5);
            const int fx = sx & 31, fy = sy & 31;
            const float wx1 = fx * (1.f / 32), wx0 = 1.f - wx1, wy1 = fy * (1.f / 32), wy0 = 1.f - wy1;
            const int x0 = clampi(ix, 0, w - 1), x1 = clampi(ix + 1, 0, w - 1);
            const int y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
            const float v00 = (float)I1[(size_t)y0 * w + x0], v01 = (float)I1[(size_t)y0 * w + x1];
            const float v10 = (float)I1[(size_t)y1 * w + x0], v11 = (float)I1[(size_t)y1 * w + x1];
            const float warped = v00 * (wy0 * wx0) + v01 * (wy0 * wx1) + v10 * (wy1 * wx0) + v11 * (wy1 * wx1);
            const float i0 = (float)I0[q];
            avg[q] = i0 * 0.5f + warped * 0.5f + 0.f;
            Iz[q] = warped - i0;
        }
#define DX(src, y, x) ((src)[(size_t)(y) * w + clampi((x) + 1, 0, w - 1)] - (src)[(size_t)(y) * w + clampi((x) - 1, 0, w - 1)])
#define DY(src, y, x) ((src)[(size_t)clampi((y) + 1, 0, h - 1) * w + (x)] - (src)[(size_t)clampi((y) - 1, 0, h - 1) * w + (x)])
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t q = (size_t)y * w + x;
            Ix[q] = DX(avg, y, x);
            Iy[q] = DY(avg, y, x);
            Ixz[q] = DX(Iz, y, x);
            Iyz[q] = DY(Iz, y, x);
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t q = (size_t)y * w + x;
            Ixx[q] = DX(Ix, y, x);
            Ixy[q] = DY(Ix, y, x);
            Iyy[q] = DY(Iy, y, x);
        }
#undef DX
#undef DY
    memcpy(tU, U, sizeof(float) * np);
    memcpy(tV, V, sizeof(float) * np);
    memset(dU, 0, sizeof(float) * np);
    memset(dV, 0, sizeof(float) * np);

    for (int it = 0; it < fixed_iters; it++) {
        /* smoothness weights (all pixels; replicate border makes the missing difference zero) */
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const size_t q = (size_t)y * w + x;
                const size_t qr = (size_t)y * w + (x + 1 < w ? x + 1 : x);
                const size_t qd = (size_t)(y + 1 < h ? y + 1 : y) * w + x;
                const float ux = tU[qr] - tU[q], vx = tV[qr] - tV[q];
                const float uy = tU[qd] - tU[q], vy = tV[qd] - tV[q];
                wgt[q] = alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + epsilon_squared);
            }
        /* linear system */
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const size_t q = (size_t)y * w + x;
                float a11, a12, a22, B1, B2;
                {
                    float derivNorm = Ix[q] * Ix[q] + Iy[q] * Iy[q] + zeta_squared;
                    float Ik1z = Iz[q] + Ix[q] * dU[q] + Iy[q] * dV[q];
                    float weight = delta2 / sqrtf(Ik1z * Ik1z / derivNorm + epsilon_squared);
                    a11 = weight * (Ix[q] * Ix[q] / derivNorm) + zeta_squared;
                    a12 = weight * (Ix[q] * Iy[q] / derivNorm);
                    a22 = weight * (Iy[q] * Iy[q] / derivNorm) + zeta_squared;
                    B1 = -weight * (Iz[q] * Ix[q] / derivNorm);
                    B2 = -weight * (Iz[q] * Iy[q] / derivNorm);
                    derivNorm = Ixx[q] * Ixx[q] + Ixy[q] * Ixy[q] + zeta_squared;
                    float derivNorm2 = Iyy[q] * Iyy[q] + Ixy[q] * Ixy[q] + zeta_squared;
                    float Ik1zx = Ixz[q] + Ixx[q] * dU[q] + Ixy[q] * dV[q];
                    float Ik1zy = Iyz[q] + Ixy[q] * dU[q] + Iyy[q] * dV[q];
                    weight = gamma2 / sqrtf(Ik1zx * Ik1zx / derivNorm + Ik1zy * Ik1zy / derivNorm2 + epsilon_squared);
                    a11 += weight * (Ixx[q] * Ixx[q] / derivNorm + Ixy[q] * Ixy[q] / derivNorm2);
                    a12 += weight * (Ixx[q] * Ixy[q] / derivNorm + Ixy[q] * Iyy[q] / derivNorm2);
                    a22 += weight * (Ixy[q] * Ixy[q] / derivNorm + Iyy[q] * Iyy[q] / derivNorm2);
                    B1 += -weight * (Ixx[q] * Ixz[q] / derivNorm + Ixy[q] * Iyz[q] / derivNorm2);
                    B2 += -weight * (Ixy[q] * Ixz[q] / derivNorm + Iyy[q] * Iyz[q] / derivNorm2);
                }
                /* smoothness contributions in OpenCV's red/black scatter order (red = (x+y) even) */
                const int red = ((x + y) & 1) == 0;
                const int has_r = x + 1 < w, has_l = x > 0, has_d = y + 1 < h, has_u = y > 0;
                float own_hu = 0, own_hv = 0, left_hu = 0, left_hv = 0, wl = 0;
                if (has_r) { own_hu = wgt[q] * (U[q + 1] - U[q]); own_hv = wgt[q] * (V[q + 1] - V[q]); }
                if (has_l) { wl = wgt[q - 1]; left_hu = wl * (U[q] - U[q - 1]); left_hv = wl * (V[q] - V[q - 1]); }
                float own_vu = 0, own_vv = 0, up_vu = 0, up_vv = 0, wu = 0;
                if (has_d) { own_vu = wgt[q] * (U[q + w] - U[q]); own_vv = wgt[q] * (V[q + w] - V[q]); }
                if (has_u) { wu = wgt[q - w]; up_vu = wu * (U[q] - U[q - w]); up_vv = wu * (V[q] - V[q - w]); }
                if (red) {
                    if (has_r) { B1 += own_hu; a11 += wgt[q]; B2 += own_hv; a22 += wgt[q]; }
                    if (has_l) { B1 -= left_hu; a11 += wl; B2 -= left_hv; a22 += wl; }
                    if (has_d) { B1 += own_vu; a11 += wgt[q]; B2 += own_vv; a22 += wgt[q]; }
                    if (has_u) { B1 -= up_vu; a11 += wu; B2 -= up_vv; a22 += wu; }
                } else {
                    if (has_l) { B1 -= left_hu; a11 += wl; B2 -= left_hv; a22 += wl; }
                    if (has_r) { B1 += own_hu; a11 += wgt[q]; B2 += own_hv; a22 += wgt[q]; }
                    if (has_u) { B1 -= up_vu; a11 += wu; B2 -= up_vv; a22 += wu; }
                    if (has_d) { B1 += own_vu; a11 += wgt[q]; B2 += own_vv; a22 += wgt[q]; }
                }
                A11[q] = a11; A12[q] = a12; A22[q] = a22; b1[q] = B1; b2[q] = B2;
            }
        /* red-black SOR */
        for (int s = 0; s < sor_iters; s++)
            for (int color = 0; color < 2; color++)
                for (int y = 0; y < h; y++)
                    for (int x = (y + color) & 1; x < w; x += 2) {
                        const size_t q = (size_t)y * w + x;
                        const float wl = x > 0 ? wgt[q - 1] : 0.f, wu = y > 0 ? wgt[q - w] : 0.f;
                        const float dul = x > 0 ? dU[q - 1] : 0.f, dvl = x > 0 ? dV[q - 1] : 0.f;
                        const float dur = x + 1 < w ? dU[q + 1] : 0.f, dvr = x + 1 < w ? dV[q + 1] : 0.f;
                        const float duu = y > 0 ? dU[q - w] : 0.f, dvu = y > 0 ? dV[q - w] : 0.f;
                        const float dud = y + 1 < h ? dU[q + w] : 0.f, dvd = y + 1 < h ? dV[q + w] : 0.f;
                        const float sigmaU = wl * dul + wgt[q] * dur + wu * duu + wgt[q] * dud;
                        const float sigmaV = wl * dvl + wgt[q] * dvr + wu * dvu + wgt[q] * dvd;
                        dU[q] += omega * ((sigmaU + b1[q] - dV[q] * A12[q]) / A11[q] - dU[q]);
                        dV[q] += omega * ((sigmaV + b2[q] - dU[q] * A12[q]) / A22[q] - dV[q]);
                    }
        for (size_t q = 0; q < np; q++) {
            tU[q] = U[q] + dU[q];
            tV[q] = V[q] + dV[q];
        }
    }
    memcpy(U, tU, sizeof(float) * np);
    memcpy(V, tV, sizeof(float) * np);
    free(buf);
}

/* Test-only view (tests/test_referee_cpu.py): the patch inverse search of ONE level -- I0 / I1 are the level images, Ux / Uy the
 * dense initial flow [h][w] (what the coarser level handed down), Sx / Sy receive the sparse flow [hs][ws] of the patch grid. */
int vo_dis_patch_search_debug(const uint8_t* I0, const uint8_t* I1, int h, int w, const float* Ux, const float* Uy, float* Sx, float* Sy,
                              const vo_dis_params* p)
{
    if (!I0 || !I1 || !Ux || !Uy || !Sx || !Sy || !p || p->patch_size != 8 || h < 8 || w < 8) return 1;
    Level L0, L1;
    memset(&L0, 0, sizeof(L0));
    memset(&L1, 0, sizeof(L1));
    const int psz = p->patch_size, pstr = p->patch_stride;
    L0.w = L1.w = w; L0.h = L1.h = h;
    L0.ws = L1.ws = 1 + (w - psz) / pstr;
    L0.hs = L1.hs = 1 + (h - psz) / pstr;
    const size_t np = (size_t)h * w, ns = (size_t)L0.ws * L0.hs;
    L0.I = (uint8_t*)malloc(np); memcpy(L0.I, I0, np);
    L0.Ix = (short*)malloc(sizeof(short) * np); L0.Iy = (short*)malloc(sizeof(short) * np);
    sobel_s16(I0, h, w, L0.Ix, L0.Iy);
    L0.xx = (float*)malloc(sizeof(float) * ns); L0.yy = (float*)malloc(sizeof(float) * ns); L0.xy = (float*)malloc(sizeof(float) * ns);
    L0.sx = (float*)malloc(sizeof(float) * ns); L0.sy = (float*)malloc(sizeof(float) * ns);
    structure_tensor(&L0, psz, pstr);
    L1.I = (uint8_t*)malloc(np); memcpy(L1.I, I1, np);
    L1.Iext = (uint8_t*)malloc((size_t)(h + 2 * DIS_BORDER) * (w + 2 * DIS_BORDER));
    pad_replicate(I1, h, w, DIS_BORDER, L1.Iext);
    patch_inverse_search(&L0, &L1, Ux, Uy, Sx, Sy, p);
    level_free(&L0);
    level_free(&L1);
    return 0;
}

/* DIS runs OpenCV's VariationalRefinement with five SOR iterations per fixed-point iteration (dis_flow.cpp) */
static void variational_refine(const uint8_t* I0, const uint8_t* I1, int h, int w, float* U, float* V,
                               const vo_dis_params* p)
{
    variational_refine_n(I0, I1, h, w, U, V, p, 5);
}

/* Test-only view (tests/test_referee_cpu.py): the refinement of a given flow on ONE level with a chosen number of SOR
 * iterations -- with one fixed-point iteration and many sweeps the increment converges to the exact solution of that
 * iteration's linear system, which an independent sparse direct solve can referee. */
int vo_variational_refine_debug(const uint8_t* I0, const uint8_t* I1, int h, int w, float* U, float* V,
                                const vo_dis_params* p, int sor_iters)
{
    if (!I0 || !I1 || !U || !V || !p || h < 2 || w < 2 || sor_iters < 1) return 1;
    variational_refine_n(I0, I1, h, w, U, V, p, sor_iters);
    return 0;
}

/* flow between two prepared frames; L0 needs gradients, L1 needs Iext */
static int dis_pair(const Level* L0, const Level* L1, int h, int w, const vo_dis_params* p, int coarsest,
                    float* flow)
{
    float* Ux[MAX_LEVELS] = {0};
    float* Uy[MAX_LEVELS] = {0};
    const Level* F = &L0[p->finest_scale];
    float* Sx = (float*)malloc(sizeof(float) * (size_t)F->ws * F->hs);
    float* Sy = (float*)malloc(sizeof(float) * (size_t)F->ws * F->hs);
    for (int i = p->finest_scale; i <= coarsest; i++) {
        Ux[i] = (float*)calloc((size_t)L0[i].w * L0[i].h, sizeof(float));
        Uy[i] = (float*)calloc((size_t)L0[i].w * L0[i].h, sizeof(float));
    }
    for (int i = coarsest; i >= p->finest_scale; i--) {
        patch_inverse_search(&L0[i], &L1[i], Ux[i], Uy[i], Sx, Sy, p);
        densify(&L0[i], &L1[i], Sx, Sy, Ux[i], Uy[i], p);
        if (p->var_iter > 0) variational_refine(L0[i].I, L1[i].I, L0[i].h, L0[i].w, Ux[i], Uy[i], p);
        if (i > p->finest_scale) {
            vo_resize_linear_f32(Ux[i], L0[i].h, L0[i].w, 1, Ux[i - 1], L0[i - 1].h, L0[i - 1].w);
            vo_resize_linear_f32(Uy[i], L0[i].h, L0[i].w, 1, Uy[i - 1], L0[i - 1].h, L0[i - 1].w);
            const size_t np = (size_t)L0[i - 1].h * L0[i - 1].w;
            for (size_t q = 0; q < np; q++) { Ux[i - 1][q] *= 2; Uy[i - 1][q] *= 2; }
        }
    }
    {
        const int fh = F->h, fw = F->w;
        float* Uc = (float*)malloc(sizeof(float) * (size_t)fh * fw * 2);
        for (size_t q = 0; q < (size_t)fh * fw; q++) {
            Uc[q * 2] = Ux[p->finest_scale][q];
            Uc[q * 2 + 1] = Uy[p->finest_scale][q];
        }
        vo_resize_linear_f32(Uc, fh, fw, 2, flow, h, w);
        const float mul = (float)(1 << p->finest_scale);
        for (size_t q = 0; q < (size_t)h * w * 2; q++) flow[q] *= mul;
        free(Uc);
    }
    for (int i = p->finest_scale; i <= coarsest; i++) { free(Ux[i]); free(Uy[i]); }
    free(Sx);
    free(Sy);
    return 0;
}

/* DISOpticalFlowImpl::calc: coarsest scale from the image size; if it is below the finest scale,
 * autoSelectPatchSizeAndScales() (finest_scale 2 -> "default" branch): patch 8,
 * coarsest = max(0, floor(log2(2*w / (5*8)))), finest = max(coarsest - 2, 0).  The DIS object keeps the
 * modified finest_scale for later calls (the reference reuses one object for the whole clip). */
static int dis_scales(int h, int w, vo_dis_params* p, int* coarsest)
{
    if (p->patch_size != 8 || !p->use_mean_norm || !p->use_spatial_prop) return -1;
    *coarsest = vo_dis_coarsest_scale(h, w, p->patch_size);
    if (*coarsest < 0) return -3; /* OpenCV: "The input image must have either width or height >= 12" */
    if (*coarsest < p->finest_scale) {
        int c = (int)floor(log2((2.0f * (float)w) / (5.0f * (float)p->patch_size)));
        *coarsest = c > 0 ? c : 0;
        p->finest_scale = *coarsest - 2 > 0 ? *coarsest - 2 : 0;
    }
    if (*coarsest >= MAX_LEVELS) return -2;
    return 0;
}

int vo_dis_calc(const uint8_t* I0, const uint8_t* I1, int h, int w, const vo_dis_params* p_in, float* flow)
{
    int coarsest;
    vo_dis_params pl = *p_in, *p = &pl;
    int rc = dis_scales(h, w, p, &coarsest);
    if (rc) return rc;
    Level L0[MAX_LEVELS], L1[MAX_LEVELS];
    memset(L0, 0, sizeof(L0));
    memset(L1, 0, sizeof(L1));
    build_levels(I0, h, w, p, coarsest, L0, 1, 0);
    build_levels(I1, h, w, p, coarsest, L1, 0, 1);
    rc = dis_pair(L0, L1, h, w, p, coarsest, flow);
    for (int i = 0; i < MAX_LEVELS; i++) { level_free(&L0[i]); level_free(&L1[i]); }
    return rc;
}

/* calc() on a persistent DIS object: like vo_dis_calc, and the (possibly auto-selected) finest scale is written back
 * into *p as OpenCV keeps it in the object -- used by the cv2 stand-in that serves the reference's own pair loop */
int vo_dis_calc_stateful(const uint8_t* I0, const uint8_t* I1, int h, int w, vo_dis_params* p, float* flow)
{
    int coarsest;
    vo_dis_params probe = *p;
    int rc = dis_scales(h, w, &probe, &coarsest);
    if (rc) return rc;
    rc = vo_dis_calc(I0, I1, h, w, p, flow);
    if (rc == 0) p->finest_scale = probe.finest_scale;
    return rc;
}

static int dis_clip_range(const uint8_t* gray, int n, int h, int w, const vo_dis_params* p, int coarsest, float* flow)
{
    Level* all = (Level*)calloc((size_t)n * MAX_LEVELS, sizeof(Level));
#pragma omp parallel for schedule(dynamic)
    for (int f = 0; f < n; f++)
        build_levels(gray + (size_t)f * h * w, h, w, p, coarsest, all + (size_t)f * MAX_LEVELS, f < n - 1, f > 0);
#pragma omp parallel for schedule(dynamic)
    for (int f = 0; f < n - 1; f++)
        dis_pair(all + (size_t)f * MAX_LEVELS, all + (size_t)(f + 1) * MAX_LEVELS, h, w, p, coarsest,
                 flow + (size_t)f * h * w * 2);
    for (size_t i = 0; i < (size_t)n * MAX_LEVELS; i++) level_free(&all[i]);
    free(all);
    return 0;
}

/* one DIS object for the whole clip (flow.py:316): the first calc() may auto-select scales and keeps the
 * new finest scale; later calls recompute the coarsest scale from the image size */
int vo_dis_calc_clip(const uint8_t* gray, int n, int h, int w, const vo_dis_params* p_in, float* flow)
{
    vo_dis_params p0 = *p_in, p1;
    int c0, c1;
    int rc = dis_scales(h, w, &p0, &c0);
    if (rc) return rc;
    p1 = p0;
    rc = dis_scales(h, w, &p1, &c1);
    if (rc) return rc;
    if (n < 2) return 0;
    if (c1 == c0 && p1.finest_scale == p0.finest_scale) return dis_clip_range(gray, n, h, w, &p0, c0, flow);
    rc = dis_clip_range(gray, 2, h, w, &p0, c0, flow);
    if (rc || n == 2) return rc;
    return dis_clip_range(gray + (size_t)h * w, n - 1, h, w, &p1, c1, flow + (size_t)h * w * 2);
}
