// vstab_dis.hip -- F3: DIS dense optical flow for all consecutive pairs of a clip, batched.
//
// Replaces cv2.DISOpticalFlow (PRESET_MEDIUM + finestScale 2 / patchSize 8 / patchStride 4 /
// spatial propagation; nodes/video_stabilizer_flow.py:82-86,140 of the reference).
// The algorithm is OpenCV's (modules/video/src/dis_flow.cpp, variational_refinement.cpp); the
// arithmetic mirrors oracle/vo_dis.c operation by operation (FMA contraction is off) so the
// two agree bit for bit.
//
// MI355X mapping
//   * every stage is batched over the P = N-1 frame pairs of the clip; per-frame data (pyramid,
//     Sobel gradients, structure tensor) is computed once per frame and shared by the two pairs
//     that use the frame
//   * patch inverse search: the raster recurrence (left/top propagation, 8 fixed stripes) is
//     kept; parallelism = P pairs x 8 stripes wavefronts.  An 8x8 patch is exactly one wave64:
//     lane = row*8+col, the four patch sums are 6-level XOR-butterfly shuffle reductions, the
//     stripe's sparse flow lives in LDS
//   * densification / variational refinement / resizes are per-pixel gather kernels over
//     [P, h, w]; the working set at the finest level (240x135) is L2/Infinity-Cache resident
//   * nothing here is a contraction: no MFMA
#include "vstab_internal.h"
#include <cfloat>
#include <cmath>

namespace {

constexpr float DIS_EPS = 0.001f;
constexpr float DIS_INF = 1e10f;
constexpr int DIS_BORDER = 16;
constexpr int PSZ = 8;
constexpr int PSTR = 4;
constexpr int FINEST = 2;
constexpr int GD_ITERS = 25;
constexpr int VAR_ITERS = 5;
constexpr int SOR_ITERS = 5;
constexpr int MAX_LEVELS = 16;

struct LevelGeom {
    int w, h, ws, hs;
};

unsigned grid_for(long long items, int block = 256)
{
    long long b = (items + block - 1) / block;
    const long long cap = 256LL * 32;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

#define GRID_STRIDE(t, total) \
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (total); t += (long long)gridDim.x * blockDim.x)

__device__ __forceinline__ int d_ceil(double v) { int i = (int)v; return i + (i < v); }
__device__ __forceinline__ int d_floor(double v) { int i = (int)v; return i - (i > v); }
__device__ __forceinline__ int f_floor(float v) { int i = (int)v; return i - (i > v); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int sat_u8_round(float v)
{
    int i = (int)__builtin_rintf(v);
    return i < 0 ? 0 : (i > 255 ? 255 : i);
}
__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

// ---- INTER_AREA u8 -> u8, any ratio >= 1 (tables computed on the fly, OpenCV's order) ----
struct AreaAxis { int n; int si[8]; float a[8]; };

__device__ __forceinline__ void area_axis(int d, int ssize, double scale, AreaAxis& ax)
{
    const double fsx1 = d * scale, fsx2 = fsx1 + scale;
    const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
    int sx1 = d_ceil(fsx1), sx2 = d_floor(fsx2);
    sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
    sx1 = sx1 < sx2 ? sx1 : sx2;
    int k = 0;
    if (sx1 - fsx1 > 1e-3) { ax.si[k] = sx1 - 1; ax.a[k++] = (float)((sx1 - fsx1) / cell); }
    for (int sx = sx1; sx < sx2 && k < 7; sx++) { ax.si[k] = sx; ax.a[k++] = (float)(1.0 / cell); }
    if (fsx2 - sx2 > 1e-3 && k < 8) {
        double a = fsx2 - sx2;
        a = a < 1. ? a : 1.;
        a = a < cell ? a : cell;
        ax.si[k] = sx2; ax.a[k++] = (float)(a / cell);
    }
    ax.n = k;
}

// mode 0: exact 2x2, 1: exact kx x ky integer boxes, 2: general
__global__ __launch_bounds__(256) void area_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int sh,
                                                      int sw, int dh, int dw, int mode, int kx, int ky, double scale_x,
                                                      double scale_y)
{
    const long long total = (long long)n * dh * dw;
    GRID_STRIDE(t, total) {
        const int x = (int)(t % dw);
        const long long r = t / dw;
        const int y = (int)(r % dh);
        const int f = (int)(r / dh);
        const uint8_t* S = src + (size_t)f * sh * sw;
        int o;
        if (mode == 0) {
            const uint8_t* s0 = S + (size_t)(2 * y) * sw + 2 * x;
            o = (s0[0] + s0[1] + s0[sw] + s0[sw + 1] + 2) >> 2;
        } else if (mode == 1) {
            int sum = 0;
            for (int j = 0; j < ky; j++)
                for (int i = 0; i < kx; i++) sum += S[(size_t)(y * ky + j) * sw + x * kx + i];
            o = sat_u8_round(sum * (1.f / (kx * ky)));
        } else {
            AreaAxis ax, ay;
            area_axis(x, sw, scale_x, ax);
            area_axis(y, sh, scale_y, ay);
            float sum = 0.f;
            for (int j = 0; j < ay.n; j++) {
                const uint8_t* row = S + (size_t)ay.si[j] * sw;
                float buf = 0.f;
                for (int k = 0; k < ax.n; k++) buf += row[ax.si[k]] * ax.a[k];
                const float term = ay.a[j] * buf;
                sum = (j == 0) ? term : sum + term;
            }
            o = sat_u8_round(sum);
        }
        dst[t] = (uint8_t)o;
    }
}

__global__ __launch_bounds__(256) void pad_replicate_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int h, int w)
{
    const int we = w + 2 * DIS_BORDER, he = h + 2 * DIS_BORDER;
    const long long total = (long long)n * he * we;
    GRID_STRIDE(t, total) {
        const int x = (int)(t % we);
        const long long r = t / we;
        const int y = (int)(r % he);
        const int f = (int)(r / he);
        const int sx = clampi(x - DIS_BORDER, 0, w - 1), sy = clampi(y - DIS_BORDER, 0, h - 1);
        dst[t] = src[((size_t)f * h + sy) * w + sx];
    }
}

__global__ __launch_bounds__(256) void sobel_kernel(const uint8_t* __restrict__ I, short* __restrict__ Ix, short* __restrict__ Iy, int n, int h, int w)
{
    const long long total = (long long)n * h * w;
    GRID_STRIDE(t, total) {
        const int x = (int)(t % w);
        const long long r = t / w;
        const int y = (int)(r % h);
        const int f = (int)(r / h);
        const uint8_t* S = I + (size_t)f * h * w;
        const uint8_t* r0 = S + (size_t)reflect101(y - 1, h) * w;
        const uint8_t* r1 = S + (size_t)y * w;
        const uint8_t* r2 = S + (size_t)reflect101(y + 1, h) * w;
        const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
        const int gx = (r0[xr] + 2 * r1[xr] + r2[xr]) - (r0[xl] + 2 * r1[xl] + r2[xl]);
        const int gy = (r2[xl] + 2 * r2[x] + r2[xr]) - (r0[xl] + 2 * r0[x] + r0[xr]);
        Ix[t] = (short)gx;
        Iy[t] = (short)gy;
    }
}

// precomputeStructureTensor, horizontal running sums: one thread per (frame,row)
__global__ __launch_bounds__(64) void tensor_h_kernel(const short* __restrict__ Ix, const short* __restrict__ Iy, float* __restrict__ aux,
                                                      int n, int h, int w, int ws)
{
    const long long rows = (long long)n * h;
    const size_t plane = (size_t)n * h * ws;
    GRID_STRIDE(t, rows) {
        const short* xr = Ix + (size_t)t * w;
        const short* yr = Iy + (size_t)t * w;
        float* o = aux + (size_t)t * ws;
        float s_xx = 0.f, s_yy = 0.f, s_xy = 0.f, s_x = 0.f, s_y = 0.f;
        for (int j = 0; j < PSZ; j++) {
            s_xx += xr[j] * xr[j];
            s_yy += yr[j] * yr[j];
            s_xy += xr[j] * yr[j];
            s_x += xr[j];
            s_y += yr[j];
        }
        o[0] = s_xx; o[plane] = s_yy; o[2 * plane] = s_xy; o[3 * plane] = s_x; o[4 * plane] = s_y;
        int js = 1;
        for (int j = PSZ; j < w; j++) {
            s_xx += (xr[j] * xr[j] - xr[j - PSZ] * xr[j - PSZ]);
            s_yy += (yr[j] * yr[j] - yr[j - PSZ] * yr[j - PSZ]);
            s_xy += (xr[j] * yr[j] - xr[j - PSZ] * yr[j - PSZ]);
            s_x += (xr[j] - xr[j - PSZ]);
            s_y += (yr[j] - yr[j - PSZ]);
            if ((j - PSZ + 1) % PSTR == 0) {
                o[js] = s_xx; o[plane + js] = s_yy; o[2 * plane + js] = s_xy; o[3 * plane + js] = s_x; o[4 * plane + js] = s_y;
                js++;
            }
        }
    }
}

// vertical running sums: one thread per (quantity, frame, js)
__global__ __launch_bounds__(64) void tensor_v_kernel(const float* __restrict__ aux, float* __restrict__ out, int n, int h, int ws, int hs)
{
    const long long cols = 5LL * n * ws;
    const size_t aplane = (size_t)n * h * ws, oplane = (size_t)n * hs * ws;
    GRID_STRIDE(t, cols) {
        const int j = (int)(t % ws);
        const long long r = t / ws;
        const int f = (int)(r % n);
        const int k = (int)(r / n);
        const float* a = aux + k * aplane + (size_t)f * h * ws + j;
        float* o = out + k * oplane + (size_t)f * hs * ws + j;
        float sum = 0.f;
        for (int i = 0; i < PSZ; i++) sum += a[(size_t)i * ws];
        o[0] = sum;
        int is = 1;
        for (int i = PSZ; i < h; i++) {
            sum += (a[(size_t)i * ws] - a[(size_t)(i - PSZ) * ws]);
            if ((i - PSZ + 1) % PSTR == 0) { o[(size_t)is * ws] = sum; is++; }
        }
    }
}

// ---- patch inverse search ------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s);
    return v;
}

struct Bilin { int off; float w00, w01, w10, w11; };

__device__ __forceinline__ Bilin bilin_weights(int i, int j, float Ux, float Uy, float i_lo, float i_hi, float j_lo, float j_hi, int w_ext)
{
    float ii = (float)i + Uy + (float)DIS_BORDER;
    float jj = (float)j + Ux + (float)DIS_BORDER;
    ii = ii > i_lo ? ii : i_lo;
    ii = ii < i_hi ? ii : i_hi;
    jj = jj > j_lo ? jj : j_lo;
    jj = jj < j_hi ? jj : j_hi;
    const float fi = __builtin_floorf(ii), fj = __builtin_floorf(jj);
    Bilin b;
    b.w11 = (ii - fi) * (jj - fj);
    b.w10 = (ii - fi) * (fj + 1 - jj);
    b.w01 = (fi + 1 - ii) * (jj - fj);
    b.w00 = (fi + 1 - ii) * (fj + 1 - jj);
    b.off = (int)ii * w_ext + (int)jj;
    return b;
}

struct PisArgs {
    const uint8_t* I;      // [n][h][w]
    const uint8_t* Iext;   // [n][h+32][w+32]
    const short* Ix;       // [n][h][w]
    const short* Iy;
    const float* tensor;   // 5 planes of [n][hs][ws]
    const float* U;        // [P][h][w] dense init
    const float* V;
    float* Sx;             // [P][hs][ws]
    float* Sy;
    int n, w, h, ws, hs, stripe_sz;
};

__global__ __launch_bounds__(64) void pis_kernel(PisArgs a)
{
    extern __shared__ float lds[];  // [2][stripe_sz][ws]
    const int pair = blockIdx.x >> 3, stripe = blockIdx.x & 7;
    const int lane = threadIdx.x, r = lane >> 3, c = lane & 7;
    const int w = a.w, h = a.h, ws = a.ws, hs = a.hs;
    const int w_ext = w + 2 * DIS_BORDER;
    const int row_lo = min(stripe * a.stripe_sz, hs), row_hi = min((stripe + 1) * a.stripe_sz, hs);
    if (row_lo >= row_hi) return;
    float* lSx = lds;
    float* lSy = lds + a.stripe_sz * ws;
    const uint8_t* I0 = a.I + (size_t)pair * h * w;
    const short* Ix = a.Ix + (size_t)pair * h * w;
    const short* Iy = a.Iy + (size_t)pair * h * w;
    const uint8_t* I1e = a.Iext + (size_t)(pair + 1) * (h + 2 * DIS_BORDER) * w_ext;
    const size_t tplane = (size_t)a.n * hs * ws;
    const float* T = a.tensor + (size_t)pair * hs * ws;
    const float* U = a.U + (size_t)pair * h * w;
    const float* V = a.V + (size_t)pair * h * w;
    const float i_lo = DIS_BORDER - PSZ + 1.0f, i_hi = DIS_BORDER + h - 1.0f;
    const float j_lo = DIS_BORDER - PSZ + 1.0f, j_hi = DIS_BORDER + w - 1.0f;
    const int num_inner_iter = GD_ITERS / 2;
    const float nn = (float)(PSZ * PSZ);
    const int lane_off1 = r * w_ext + c;

    for (int iter = 0; iter < 2; iter++) {
        const int dir = (iter == 0) ? 1 : -1;
        const int start_is = (iter == 0) ? row_lo : row_hi - 1;
        const int end_is = (iter == 0) ? row_hi : row_lo - 1;
        const int start_js = (iter == 0) ? 0 : ws - 1;
        const int end_js = (iter == 0) ? ws : -1;
        for (int is = start_is; dir * is < dir * end_is; is += dir) {
            const int i = is * PSTR;
            for (int js = start_js; dir * js < dir * end_js; js += dir) {
                const int j = js * PSTR;
                const int lidx = (is - row_lo) * ws + js;
                const size_t sidx = (size_t)is * ws + js;
                const size_t poff = (size_t)(i + r) * w + j + c;
                const float i0 = (float)I0[poff];
                const float gx = (float)Ix[poff], gy = (float)Iy[poff];
                float Sxv, Syv;
                if (iter == 0) {
                    Sxv = U[(size_t)(i + PSZ / 2) * w + j + PSZ / 2];
                    Syv = V[(size_t)(i + PSZ / 2) * w + j + PSZ / 2];
                } else {
                    Sxv = lSx[lidx];
                    Syv = lSy[lidx];
                }
#define PATCH_DIFF(bw)                                                                                     \
    ({                                                                                                     \
        const uint8_t* q_ = I1e + (bw).off + lane_off1;                                                    \
        (bw).w00 * (float)q_[0] + (bw).w01 * (float)q_[1] + (bw).w10 * (float)q_[w_ext] +                  \
            (bw).w11 * (float)q_[w_ext + 1] - i0;                                                          \
    })
#define SSD_AT(dst, ux, uy)                                                                                \
    do {                                                                                                   \
        Bilin b_ = bilin_weights(i, j, (ux), (uy), i_lo, i_hi, j_lo, j_hi, w_ext);                         \
        const float d_ = PATCH_DIFF(b_);                                                                   \
        const float sd_ = wave_sum(d_), sq_ = wave_sum(d_ * d_);                                           \
        dst = sq_ - sd_ * sd_ / nn;                                                                        \
    } while (0)
                float min_SSD, cur_SSD;
                SSD_AT(min_SSD, Sxv, Syv);
                if (dir * js > dir * start_js) {
                    const float nx = lSx[lidx - dir], ny = lSy[lidx - dir];
                    SSD_AT(cur_SSD, nx, ny);
                    if (cur_SSD < min_SSD) { min_SSD = cur_SSD; Sxv = nx; Syv = ny; }
                }
                if (dir * is > dir * start_is) {
                    const float nx = lSx[lidx - dir * ws], ny = lSy[lidx - dir * ws];
                    SSD_AT(cur_SSD, nx, ny);
                    if (cur_SSD < min_SSD) { min_SSD = cur_SSD; Sxv = nx; Syv = ny; }
                }
                float cur_Ux = Sxv, cur_Uy = Syv;
                const float txx = T[sidx], tyy = T[tplane + sidx], txy = T[2 * tplane + sidx];
                const float x_grad_sum = T[3 * tplane + sidx], y_grad_sum = T[4 * tplane + sidx];
                float detH = txx * tyy - txy * txy;
                if (__builtin_fabsf(detH) < DIS_EPS) detH = DIS_EPS;
                const float invH11 = tyy / detH, invH12 = -txy / detH, invH22 = txx / detH;
                float prev_SSD = DIS_INF;
                for (int t = 0; t < num_inner_iter; t++) {
                    Bilin b = bilin_weights(i, j, cur_Ux, cur_Uy, i_lo, i_hi, j_lo, j_hi, w_ext);
                    const float d = PATCH_DIFF(b);
                    const float sum_diff = wave_sum(d), sum_sq = wave_sum(d * d);
                    const float sum_x = wave_sum(d * gx), sum_y = wave_sum(d * gy);
                    const float dUx = sum_x - sum_diff * x_grad_sum / nn;
                    const float dUy = sum_y - sum_diff * y_grad_sum / nn;
                    const float SSD = sum_sq - sum_diff * sum_diff / nn;
                    const float dx = invH11 * dUx + invH12 * dUy;
                    const float dy = invH12 * dUx + invH22 * dUy;
                    cur_Ux -= dx;
                    cur_Uy -= dy;
                    if (SSD >= prev_SSD) break;
                    prev_SSD = SSD;
                }
#undef SSD_AT
#undef PATCH_DIFF
                {
                    const double ddx = (double)(cur_Ux - Sxv), ddy = (double)(cur_Uy - Syv);
                    if (__builtin_sqrt(ddx * ddx + ddy * ddy) <= (double)PSZ) { Sxv = cur_Ux; Syv = cur_Uy; }
                }
                __syncthreads();
                if (lane == 0) { lSx[lidx] = Sxv; lSy[lidx] = Syv; }
                __syncthreads();
            }
        }
    }
    // write the stripe back
    float* Sx = a.Sx + (size_t)pair * hs * ws + (size_t)row_lo * ws;
    float* Sy = a.Sy + (size_t)pair * hs * ws + (size_t)row_lo * ws;
    const int cnt = (row_hi - row_lo) * ws;
    for (int k = lane; k < cnt; k += 64) { Sx[k] = lSx[k]; Sy[k] = lSy[k]; }
}

// ---- per-pixel phase bodies (shared by every launch shape) -----------------------------------
// All indices below are GLOBAL element indices into the [P][h][w] planes; (x, y) is the pixel.
struct VrBufs {
    float *avg, *Iz, *Ix, *Iy, *Ixx, *Ixy, *Iyy, *Ixz, *Iyz, *A11, *A12, *A22, *b1, *b2, *wgt, *tU, *tV, *dU, *dV;
};

__device__ __forceinline__ void densify_px(const uint8_t* __restrict__ I0, const uint8_t* __restrict__ I1, const float* __restrict__ sx,
                                           const float* __restrict__ sy, float* __restrict__ U, float* __restrict__ V, long long t, int i,
                                           int j, int h, int w, int ws, int hs)
{
    int end_is = i / PSTR < hs - 1 ? i / PSTR : hs - 1;
    int start_is = i - PSZ >= 0 ? (i - PSZ) / PSTR + 1 : 0;
    if (start_is > end_is) start_is = end_is;
    int end_js = j / PSTR < ws - 1 ? j / PSTR : ws - 1;
    int start_js = j - PSZ >= 0 ? (j - PSZ) / PSTR + 1 : 0;
    if (start_js > end_js) start_js = end_js;
    const float i0 = (float)I0[(size_t)i * w + j];
    float sum_coef = 0.f, sum_Ux = 0.f, sum_Uy = 0.f;
    for (int is = start_is; is <= end_is; is++)
        for (int js = start_js; js <= end_js; js++) {
            const float sxv = sx[(size_t)is * ws + js], syv = sy[(size_t)is * ws + js];
            float j_m = (float)j + sxv, i_m = (float)i + syv;
            j_m = j_m > 0.0f ? j_m : 0.0f;
            j_m = j_m < (float)w - 1.0f - DIS_EPS ? j_m : (float)w - 1.0f - DIS_EPS;
            i_m = i_m > 0.0f ? i_m : 0.0f;
            i_m = i_m < (float)h - 1.0f - DIS_EPS ? i_m : (float)h - 1.0f - DIS_EPS;
            const int j_l = (int)j_m, j_u = j_l + 1, i_l = (int)i_m, i_u = i_l + 1;
            const float diff = (j_m - j_l) * (i_m - i_l) * I1[(size_t)i_u * w + j_u] +
                               (j_u - j_m) * (i_m - i_l) * I1[(size_t)i_u * w + j_l] +
                               (j_m - j_l) * (i_u - i_m) * I1[(size_t)i_l * w + j_u] +
                               (j_u - j_m) * (i_u - i_m) * I1[(size_t)i_l * w + j_l] - i0;
            const float ad = __builtin_fabsf(diff);
            const float coef = 1 / (ad > 1.0f ? ad : 1.0f);
            sum_Ux += coef * sxv;
            sum_Uy += coef * syv;
            sum_coef += coef;
        }
    U[t] = sum_Ux / sum_coef;
    V[t] = sum_Uy / sum_coef;
}

__device__ __forceinline__ void vr_warp_px(const uint8_t* __restrict__ I0, const uint8_t* __restrict__ I1, const float* __restrict__ U,
                                           const float* __restrict__ V, const VrBufs& b, long long t, int x, int y, int h, int w)
{
    const float u = U[t], v = V[t];
    const float mx = x + u, my = y + v;
    const int sx = (int)__builtin_rintf(mx * 32.f), sy = (int)__builtin_rintf(my * 32.f);
    const int ix = sat_short(sx >> 5), iy = sat_short(sy >> 5);
    const int fx = sx & 31, fy = sy & 31;
    const float wx1 = fx * (1.f / 32), wx0 = 1.f - wx1, wy1 = fy * (1.f / 32), wy0 = 1.f - wy1;
    const int x0 = clampi(ix, 0, w - 1), x1 = clampi(ix + 1, 0, w - 1);
    const int y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
    const float v00 = (float)I1[(size_t)y0 * w + x0], v01 = (float)I1[(size_t)y0 * w + x1];
    const float v10 = (float)I1[(size_t)y1 * w + x0], v11 = (float)I1[(size_t)y1 * w + x1];
    const float warped = v00 * (wy0 * wx0) + v01 * (wy0 * wx1) + v10 * (wy1 * wx0) + v11 * (wy1 * wx1);
    const float i0 = (float)I0[(size_t)y * w + x];
    b.avg[t] = i0 * 0.5f + warped * 0.5f + 0.f;
    b.Iz[t] = warped - i0;
    b.tU[t] = u;
    b.tV[t] = v;
    b.dU[t] = 0.f;
    b.dV[t] = 0.f;
}

__device__ __forceinline__ void vr_deriv1_px(const VrBufs& b, long long t, int x, int y, int h, int w)
{
    const long long base = t - (long long)y * w - x;
    const long long xl = base + (long long)y * w + clampi(x - 1, 0, w - 1), xr = base + (long long)y * w + clampi(x + 1, 0, w - 1);
    const long long yu = base + (long long)clampi(y - 1, 0, h - 1) * w + x, yd = base + (long long)clampi(y + 1, 0, h - 1) * w + x;
    b.Ix[t] = b.avg[xr] - b.avg[xl];
    b.Iy[t] = b.avg[yd] - b.avg[yu];
    b.Ixz[t] = b.Iz[xr] - b.Iz[xl];
    b.Iyz[t] = b.Iz[yd] - b.Iz[yu];
}

__device__ __forceinline__ void vr_deriv2_px(const VrBufs& b, long long t, int x, int y, int h, int w)
{
    const long long base = t - (long long)y * w - x;
    const long long xl = base + (long long)y * w + clampi(x - 1, 0, w - 1), xr = base + (long long)y * w + clampi(x + 1, 0, w - 1);
    const long long yu = base + (long long)clampi(y - 1, 0, h - 1) * w + x, yd = base + (long long)clampi(y + 1, 0, h - 1) * w + x;
    b.Ixx[t] = b.Ix[xr] - b.Ix[xl];
    b.Ixy[t] = b.Ix[yd] - b.Ix[yu];
    b.Iyy[t] = b.Iy[yd] - b.Iy[yu];
}

__device__ __forceinline__ void vr_weights_px(const VrBufs& b, long long t, int x, int y, int h, int w, float alpha2, float eps2)
{
    const long long qr = (x + 1 < w) ? t + 1 : t;
    const long long qd = (y + 1 < h) ? t + w : t;
    const float ux = b.tU[qr] - b.tU[t], vx = b.tV[qr] - b.tV[t];
    const float uy = b.tU[qd] - b.tU[t], vy = b.tV[qd] - b.tV[t];
    b.wgt[t] = alpha2 / __builtin_sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + eps2);
}

__device__ __forceinline__ void vr_system_px(const VrBufs& b, const float* __restrict__ U, const float* __restrict__ V, long long q, int x,
                                             int y, int h, int w, float delta2, float gamma2, float zeta2, float eps2)
{
    const float Ix = b.Ix[q], Iy = b.Iy[q], Iz = b.Iz[q], Ixx = b.Ixx[q], Ixy = b.Ixy[q], Iyy = b.Iyy[q], Ixz = b.Ixz[q], Iyz = b.Iyz[q];
    const float du = b.dU[q], dv = b.dV[q];
    float a11, a12, a22, B1, B2;
    {
        float derivNorm = Ix * Ix + Iy * Iy + zeta2;
        float Ik1z = Iz + Ix * du + Iy * dv;
        float weight = delta2 / __builtin_sqrtf(Ik1z * Ik1z / derivNorm + eps2);
        a11 = weight * (Ix * Ix / derivNorm) + zeta2;
        a12 = weight * (Ix * Iy / derivNorm);
        a22 = weight * (Iy * Iy / derivNorm) + zeta2;
        B1 = -weight * (Iz * Ix / derivNorm);
        B2 = -weight * (Iz * Iy / derivNorm);
        derivNorm = Ixx * Ixx + Ixy * Ixy + zeta2;
        float derivNorm2 = Iyy * Iyy + Ixy * Ixy + zeta2;
        float Ik1zx = Ixz + Ixx * du + Ixy * dv;
        float Ik1zy = Iyz + Ixy * du + Iyy * dv;
        weight = gamma2 / __builtin_sqrtf(Ik1zx * Ik1zx / derivNorm + Ik1zy * Ik1zy / derivNorm2 + eps2);
        a11 += weight * (Ixx * Ixx / derivNorm + Ixy * Ixy / derivNorm2);
        a12 += weight * (Ixx * Ixy / derivNorm + Ixy * Iyy / derivNorm2);
        a22 += weight * (Ixy * Ixy / derivNorm + Iyy * Iyy / derivNorm2);
        B1 += -weight * (Ixx * Ixz / derivNorm + Ixy * Iyz / derivNorm2);
        B2 += -weight * (Ixy * Ixz / derivNorm + Iyy * Iyz / derivNorm2);
    }
    const bool red = ((x + y) & 1) == 0;
    const bool has_r = x + 1 < w, has_l = x > 0, has_d = y + 1 < h, has_u = y > 0;
    const float wq = b.wgt[q], uq = U[q], vq = V[q];
    float own_hu = 0, own_hv = 0, left_hu = 0, left_hv = 0, wl = 0;
    if (has_r) { own_hu = wq * (U[q + 1] - uq); own_hv = wq * (V[q + 1] - vq); }
    if (has_l) { wl = b.wgt[q - 1]; left_hu = wl * (uq - U[q - 1]); left_hv = wl * (vq - V[q - 1]); }
    float own_vu = 0, own_vv = 0, up_vu = 0, up_vv = 0, wu = 0;
    if (has_d) { own_vu = wq * (U[q + w] - uq); own_vv = wq * (V[q + w] - vq); }
    if (has_u) { wu = b.wgt[q - w]; up_vu = wu * (uq - U[q - w]); up_vv = wu * (vq - V[q - w]); }
    if (red) {
        if (has_r) { B1 += own_hu; a11 += wq; B2 += own_hv; a22 += wq; }
        if (has_l) { B1 -= left_hu; a11 += wl; B2 -= left_hv; a22 += wl; }
        if (has_d) { B1 += own_vu; a11 += wq; B2 += own_vv; a22 += wq; }
        if (has_u) { B1 -= up_vu; a11 += wu; B2 -= up_vv; a22 += wu; }
    } else {
        if (has_l) { B1 -= left_hu; a11 += wl; B2 -= left_hv; a22 += wl; }
        if (has_r) { B1 += own_hu; a11 += wq; B2 += own_hv; a22 += wq; }
        if (has_u) { B1 -= up_vu; a11 += wu; B2 -= up_vv; a22 += wu; }
        if (has_d) { B1 += own_vu; a11 += wq; B2 += own_vv; a22 += wq; }
    }
    b.A11[q] = a11; b.A12[q] = a12; b.A22[q] = a22; b.b1[q] = B1; b.b2[q] = B2;
}

__device__ __forceinline__ void vr_sor_px(const VrBufs& b, long long q, int x, int y, int h, int w, float omega)
{
    const float wq = b.wgt[q];
    const float wl = x > 0 ? b.wgt[q - 1] : 0.f, wu = y > 0 ? b.wgt[q - w] : 0.f;
    const float dul = x > 0 ? b.dU[q - 1] : 0.f, dvl = x > 0 ? b.dV[q - 1] : 0.f;
    const float dur = x + 1 < w ? b.dU[q + 1] : 0.f, dvr = x + 1 < w ? b.dV[q + 1] : 0.f;
    const float duu = y > 0 ? b.dU[q - w] : 0.f, dvu = y > 0 ? b.dV[q - w] : 0.f;
    const float dud = y + 1 < h ? b.dU[q + w] : 0.f, dvd = y + 1 < h ? b.dV[q + w] : 0.f;
    const float sigmaU = wl * dul + wq * dur + wu * duu + wq * dud;
    const float sigmaV = wl * dvl + wq * dvr + wu * dvu + wq * dvd;
    const float a12 = b.A12[q];
    float du = b.dU[q], dv = b.dV[q];
    du += omega * ((sigmaU + b.b1[q] - dv * a12) / b.A11[q] - du);
    dv += omega * ((sigmaV + b.b2[q] - du * a12) / b.A22[q] - dv);
    b.dU[q] = du;
    b.dV[q] = dv;
}

// ---- bilinear f32 resize (flow upsampling between levels), result scaled by `mul` ------------
__device__ __forceinline__ void lin_coord(int d, double scale, int ssize, int& s0, float& f)
{
    float fx = (float)((d + 0.5) * scale - 0.5);
    int sx = f_floor(fx);
    fx -= sx;
    s0 = sx;
    f = fx;
}

__device__ __forceinline__ void upsample_px(const float* __restrict__ sU, const float* __restrict__ sV, float* __restrict__ dU,
                                            float* __restrict__ dV, long long src_base, long long t, int dx, int dy, int sh, int sw,
                                            double scale_x, double scale_y, float mul)
{
    int sx, sy;
    float fx, fy;
    lin_coord(dx, scale_x, sw, sx, fx);
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    lin_coord(dy, scale_y, sh, sy, fy);
    const int sy0 = clampi(sy, 0, sh - 1), sy1 = clampi(sy + 1, 0, sh - 1);
    const int sx1 = sx + 1 < sw ? sx + 1 : sx;
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const float* S0 = sU + src_base + (long long)sy0 * sw;
    const float* S1 = sU + src_base + (long long)sy1 * sw;
    float r0 = S0[sx] * a0 + S0[sx1] * a1, r1 = S1[sx] * a0 + S1[sx1] * a1;
    dU[t] = (r0 * b0 + r1 * b1) * mul;
    S0 = sV + src_base + (long long)sy0 * sw;
    S1 = sV + src_base + (long long)sy1 * sw;
    r0 = S0[sx] * a0 + S0[sx1] * a1; r1 = S1[sx] * a0 + S1[sx1] * a1;
    dV[t] = (r0 * b0 + r1 * b1) * mul;
}

struct LevelArgs {
    const uint8_t* I;   // [n][h][w] pyramid level
    const float* Sx;    // [P][hs][ws]
    const float* Sy;
    float* U;           // [P][h][w]
    float* V;
    float* nextU;       // [P][nh][nw] (finer level) or nullptr
    float* nextV;
    VrBufs vb;
    int P, h, w, ws, hs, nh, nw;
    double up_sx, up_sy;
    float alpha2, delta2, gamma2, zeta2, eps2, omega;
};

// One workgroup per frame pair runs densification, the whole variational refinement (5 fixed-point
// iterations x 5 red-black SOR sweeps) and the x2 upsample to the next finer level.  The ~70 phases are
// separated by workgroup barriers instead of kernel launches; the pair's planes stay in L2/Infinity Cache.
__global__ __launch_bounds__(1024) void level_fused_kernel(LevelArgs a)
{
    const int pair = blockIdx.x;
    const int h = a.h, w = a.w;
    const int npx = h * w;
    const long long base = (long long)pair * npx;
    const uint8_t* I0 = a.I + (size_t)pair * npx;
    const uint8_t* I1 = a.I + (size_t)(pair + 1) * npx;
    const float* sx = a.Sx + (size_t)pair * a.hs * a.ws;
    const float* sy = a.Sy + (size_t)pair * a.hs * a.ws;
    const VrBufs& b = a.vb;
#define FOR_PX(...)                                                         \
    for (int q_ = threadIdx.x; q_ < npx; q_ += blockDim.x) {                \
        const int y = q_ / w, x = q_ - y * w;                               \
        const long long t = base + q_;                                      \
        __VA_ARGS__;                                                        \
    }                                                                       \
    __syncthreads();
    FOR_PX(densify_px(I0, I1, sx, sy, a.U, a.V, t, y, x, h, w, a.ws, a.hs))
    FOR_PX(vr_warp_px(I0, I1, a.U, a.V, b, t, x, y, h, w))
    FOR_PX(vr_deriv1_px(b, t, x, y, h, w))
    FOR_PX(vr_deriv2_px(b, t, x, y, h, w))
    const int half_w = (w + 1) >> 1;
    const int nhalf = h * half_w;
    for (int it = 0; it < VAR_ITERS; it++) {
        FOR_PX(vr_weights_px(b, t, x, y, h, w, a.alpha2, a.eps2))
        FOR_PX(vr_system_px(b, a.U, a.V, t, x, y, h, w, a.delta2, a.gamma2, a.zeta2, a.eps2))
        for (int s = 0; s < SOR_ITERS * 2; s++) {
            const int color = s & 1;
            for (int k_ = threadIdx.x; k_ < nhalf; k_ += blockDim.x) {
                const int y = k_ / half_w;
                const int x = 2 * (k_ - y * half_w) + ((y + color) & 1);
                if (x < w) vr_sor_px(b, base + (long long)y * w + x, x, y, h, w, a.omega);
            }
            __syncthreads();
        }
        const bool last = it == VAR_ITERS - 1;
        FOR_PX({
            const float u = a.U[t] + b.dU[t], v = a.V[t] + b.dV[t];
            b.tU[t] = u; b.tV[t] = v;
            if (last) { a.U[t] = u; a.V[t] = v; }
        })
    }
#undef FOR_PX
    if (a.nextU != nullptr) {
        const int nn = a.nh * a.nw;
        const long long nbase = (long long)pair * nn;
        for (int q_ = threadIdx.x; q_ < nn; q_ += blockDim.x) {
            const int dy = q_ / a.nw, dx = q_ - dy * a.nw;
            upsample_px(a.U, a.V, a.nextU, a.nextV, base, nbase + q_, dx, dy, h, w, a.up_sx, a.up_sy, 2.0f);
        }
    }
}

// final resize of the finest flow to the working size (x 2^finest), evaluated on a strided grid
// (step 1 = the full field).  out is [P][gh][gw][2].
__global__ __launch_bounds__(256) void final_sample_kernel(const float* __restrict__ sU, const float* __restrict__ sV, float* __restrict__ out,
                                                           int P, int sh, int sw, int gh, int gw, int step, double scale_x, double scale_y,
                                                           float mul)
{
    const long long total = (long long)P * gh * gw;
    GRID_STRIDE(t, total) {
        const int gx = (int)(t % gw);
        const long long rr = t / gw;
        const int gy = (int)(rr % gh);
        const long long p = rr / gh;
        const int dx = gx * step, dy = gy * step;
        int sx, sy;
        float fx, fy;
        lin_coord(dx, scale_x, sw, sx, fx);
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        lin_coord(dy, scale_y, sh, sy, fy);
        const int sy0 = clampi(sy, 0, sh - 1), sy1 = clampi(sy + 1, 0, sh - 1);
        const int sx1 = sx + 1 < sw ? sx + 1 : sx;
        const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
        const float* S0 = sU + (p * sh + sy0) * sw;
        const float* S1 = sU + (p * sh + sy1) * sw;
        float r0 = S0[sx] * a0 + S0[sx1] * a1, r1 = S1[sx] * a0 + S1[sx1] * a1;
        out[t * 2] = (r0 * b0 + r1 * b1) * mul;
        S0 = sV + (p * sh + sy0) * sw;
        S1 = sV + (p * sh + sy1) * sw;
        r0 = S0[sx] * a0 + S0[sx1] * a1; r1 = S1[sx] * a0 + S1[sx1] * a1;
        out[t * 2 + 1] = (r0 * b0 + r1 * b1) * mul;
    }
}

int coarsest_scale(int h, int w)
{
    const int mx = w > h ? w : h, mn = w < h ? w : h;
    const int a = (int)(std::log(mx / (4.0 * PSZ)) / std::log(2.0) + 0.5);
    const int b = (int)(std::log((double)(mn / PSZ)) / std::log(2.0));
    return a < b ? a : b;
}

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T> T* take(size_t count)
    {
        off = (off + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

int launch_area(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int sh, int sw, int dh, int dw)
{
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    const int isx = (int)std::lrint(scale_x), isy = (int)std::lrint(scale_y);
    const bool fast = std::fabs(scale_x - isx) < DBL_EPSILON && std::fabs(scale_y - isy) < DBL_EPSILON;
    VSTAB_REQUIRE(scale_x >= 1.0 && scale_y >= 1.0 && scale_x < 6.0 && scale_y < 6.0, "dis: area ratio %.3fx%.3f unsupported", scale_x, scale_y);
    const int mode = fast ? ((isx == 2 && isy == 2) ? 0 : 1) : 2;
    const long long items = (long long)n * dh * dw;
    hipLaunchKernelGGL(area_u8_kernel, dim3(grid_for(items)), dim3(256), 0, st, src, dst, n, sh, sw, dh, dw, mode, isx, isy, scale_x, scale_y);
    VSTAB_HIP(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" int vstab_dis_flow_batch(vstab_ctx* ctx, const uint8_t* gray, int n, int h, int w, float* flow, float* grid_flow,
                                    int sample_step)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_dis_flow_batch: ctx is NULL");
    VSTAB_REQUIRE(gray != nullptr, "vstab_dis_flow_batch: gray is NULL");
    VSTAB_REQUIRE(n >= 2, "vstab_dis_flow_batch: need at least 2 frames, got %d", n);
    VSTAB_REQUIRE(flow != nullptr || grid_flow != nullptr, "vstab_dis_flow_batch: no output requested");
    VSTAB_REQUIRE(sample_step >= 1, "vstab_dis_flow_batch: sample_step must be >= 1");
    VSTAB_REQUIRE(h > 0 && w > 0, "vstab_dis_flow_batch: non-positive size");
    const int coarsest = coarsest_scale(h, w);
    // OpenCV would re-select patch size / scales for tiny images (autoSelectPatchSizeAndScales); that
    // path is not implemented: fail loudly instead of silently differing.
    VSTAB_REQUIRE(coarsest >= FINEST && coarsest < MAX_LEVELS,
                  "vstab_dis_flow_batch: %dx%d is too small for finest scale %d (coarsest %d)", w, h, FINEST, coarsest);
    VSTAB_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int P = n - 1;

    LevelGeom G[MAX_LEVELS];
    {
        int fraction = 1, ch = 0, cw = 0;
        for (int i = 0; i <= coarsest; i++) {
            if (i == FINEST) { ch = h / fraction; cw = w / fraction; }
            else if (i > FINEST) { ch = ch / 2; cw = cw / 2; }
            if (i >= FINEST) {
                G[i].h = ch; G[i].w = cw;
                G[i].ws = 1 + (cw - PSZ) / PSTR;
                G[i].hs = 1 + (ch - PSZ) / PSTR;
                VSTAB_REQUIRE(ch >= PSZ && cw >= PSZ, "vstab_dis_flow_batch: pyramid level %d is %dx%d, smaller than a patch", i, cw, ch);
            }
            fraction *= 2;
        }
    }
    // ---- workspace layout (two passes: size, then carve) ----
    uint8_t* I[MAX_LEVELS]; uint8_t* Iext[MAX_LEVELS]; short* Ixs[MAX_LEVELS]; short* Iys[MAX_LEVELS];
    float* tensor[MAX_LEVELS]; float* Ul[MAX_LEVELS]; float* Vl[MAX_LEVELS];
    float *aux = nullptr, *Sx = nullptr, *Sy = nullptr;
    VrBufs vb{};
    const LevelGeom& F = G[FINEST];
    const size_t npF = (size_t)P * F.h * F.w;
    auto layout = [&](Carver& c) {
        for (int i = FINEST; i <= coarsest; i++) {
            const size_t px = (size_t)n * G[i].h * G[i].w;
            I[i] = c.take<uint8_t>(px);
            Iext[i] = c.take<uint8_t>((size_t)n * (G[i].h + 2 * DIS_BORDER) * (G[i].w + 2 * DIS_BORDER));
            Ixs[i] = c.take<short>(px);
            Iys[i] = c.take<short>(px);
            tensor[i] = c.take<float>(5 * (size_t)n * G[i].hs * G[i].ws);
            Ul[i] = c.take<float>((size_t)P * G[i].h * G[i].w);
            Vl[i] = c.take<float>((size_t)P * G[i].h * G[i].w);
        }
        aux = c.take<float>(5 * (size_t)n * F.h * F.ws);
        Sx = c.take<float>((size_t)P * F.hs * F.ws);
        Sy = c.take<float>((size_t)P * F.hs * F.ws);
        float** planes[] = {&vb.avg, &vb.Iz, &vb.Ix, &vb.Iy, &vb.Ixx, &vb.Ixy, &vb.Iyy, &vb.Ixz, &vb.Iyz, &vb.A11,
                            &vb.A12, &vb.A22, &vb.b1, &vb.b2, &vb.wgt, &vb.tU, &vb.tV, &vb.dU, &vb.dV};
        for (float** pl : planes) *pl = c.take<float>(npF);
    };
    {
        Carver sizer(nullptr);
        layout(sizer);
        if (ctx->d_dis.reserve(sizer.off + 256)) return 1;
    }
    Carver carver(ctx->d_dis.ptr);
    layout(carver);

    KernelTimer timer(ctx, "dis");

    // ---- per-frame preparation: pyramid, padded copies, gradients, structure tensor ----
    for (int i = FINEST; i <= coarsest; i++) {
        const LevelGeom& g = G[i];
        if (i == FINEST) { if (launch_area(st, gray, I[i], n, h, w, g.h, g.w)) return 1; }
        else { if (launch_area(st, I[i - 1], I[i], n, G[i - 1].h, G[i - 1].w, g.h, g.w)) return 1; }
        const long long px = (long long)n * g.h * g.w;
        hipLaunchKernelGGL(pad_replicate_kernel, dim3(grid_for((long long)n * (g.h + 32) * (g.w + 32))), dim3(256), 0, st, I[i], Iext[i], n, g.h, g.w);
        hipLaunchKernelGGL(sobel_kernel, dim3(grid_for(px)), dim3(256), 0, st, I[i], Ixs[i], Iys[i], n, g.h, g.w);
        hipLaunchKernelGGL(tensor_h_kernel, dim3(grid_for((long long)n * g.h, 64)), dim3(64), 0, st, Ixs[i], Iys[i], aux, n, g.h, g.w, g.ws);
        hipLaunchKernelGGL(tensor_v_kernel, dim3(grid_for(5LL * n * g.ws, 64)), dim3(64), 0, st, aux, tensor[i], n, g.h, g.ws, g.hs);
        VSTAB_HIP(hipGetLastError());
    }
    VSTAB_HIP(hipMemsetAsync(Ul[coarsest], 0, sizeof(float) * (size_t)P * G[coarsest].h * G[coarsest].w, st));
    VSTAB_HIP(hipMemsetAsync(Vl[coarsest], 0, sizeof(float) * (size_t)P * G[coarsest].h * G[coarsest].w, st));

    const float zeta = 0.1f, epsilon = 0.001f, alpha = 20.0f, delta = 5.0f, gamma = 10.0f, omega = 1.6f;
    const float zeta2 = zeta * zeta, eps2 = epsilon * epsilon, gamma2 = gamma / 2, delta2 = delta / 2, alpha2 = alpha / 4;

    for (int i = coarsest; i >= FINEST; i--) {
        const LevelGeom& g = G[i];
        PisArgs pa{};
        pa.I = I[i]; pa.Iext = Iext[i]; pa.Ix = Ixs[i]; pa.Iy = Iys[i]; pa.tensor = tensor[i];
        pa.U = Ul[i]; pa.V = Vl[i]; pa.Sx = Sx; pa.Sy = Sy;
        pa.n = n; pa.w = g.w; pa.h = g.h; pa.ws = g.ws; pa.hs = g.hs;
        pa.stripe_sz = (int)std::ceil(g.hs / 8.0);
        const size_t lds_bytes = sizeof(float) * 2 * (size_t)pa.stripe_sz * g.ws;
        VSTAB_REQUIRE(lds_bytes <= 64 * 1024, "vstab_dis_flow_batch: stripe of %d x %d patches does not fit LDS", pa.stripe_sz, g.ws);
        hipLaunchKernelGGL(pis_kernel, dim3((unsigned)P * 8), dim3(64), lds_bytes, st, pa);
        LevelArgs la{};
        la.I = I[i]; la.Sx = Sx; la.Sy = Sy; la.U = Ul[i]; la.V = Vl[i]; la.vb = vb;
        la.P = P; la.h = g.h; la.w = g.w; la.ws = g.ws; la.hs = g.hs;
        la.alpha2 = alpha2; la.delta2 = delta2; la.gamma2 = gamma2; la.zeta2 = zeta2; la.eps2 = eps2; la.omega = omega;
        if (i > FINEST) {
            const LevelGeom& d = G[i - 1];
            la.nextU = Ul[i - 1]; la.nextV = Vl[i - 1]; la.nh = d.h; la.nw = d.w;
            la.up_sx = 1. / ((double)d.w / g.w); la.up_sy = 1. / ((double)d.h / g.h);
        }
        hipLaunchKernelGGL(level_fused_kernel, dim3((unsigned)P), dim3(1024), 0, st, la);
        VSTAB_HIP(hipGetLastError());
    }
    const double fsx = 1. / ((double)w / F.w), fsy = 1. / ((double)h / F.h);
    const float mul = (float)(1 << FINEST);
    if (grid_flow) {
        const int gh = (h + sample_step - 1) / sample_step, gw = (w + sample_step - 1) / sample_step;
        hipLaunchKernelGGL(final_sample_kernel, dim3(grid_for((long long)P * gh * gw)), dim3(256), 0, st, Ul[FINEST], Vl[FINEST], grid_flow, P,
                           F.h, F.w, gh, gw, sample_step, fsx, fsy, mul);
    }
    if (flow) {
        hipLaunchKernelGGL(final_sample_kernel, dim3(grid_for((long long)P * h * w)), dim3(256), 0, st, Ul[FINEST], Vl[FINEST], flow, P, F.h, F.w,
                           h, w, 1, fsx, fsy, mul);
    }
    VSTAB_HIP(hipGetLastError());
    return 0;
}
