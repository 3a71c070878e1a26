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
#include <cstdlib>
#include <cfloat>
#include <algorithm>
#include <cmath>
#include <memory>

namespace {

constexpr float DIS_EPS = 0.001f;
constexpr float DIS_INF = 1e10f;
constexpr int DIS_BORDER = 16;
constexpr int PSZ = 8;
constexpr int PSTR = 4;
constexpr int DEFAULT_FINEST = 2;   // flow.py:83 setFinestScale(2)
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

// blocks along x of a (blocks, frames) grid: enough to cover a frame, but no more than keeps the whole grid near
// 8192 workgroups (a thread then strides over a few elements instead of the launch paying for tens of thousands of
// one-element workgroups)
dim3 frame_grid(long long per_frame, int frames, int block = 256)
{
    long long b = (per_frame + block - 1) / block;
    const long long cap = std::max(1LL, (256LL * 32) / std::max(frames, 1));
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return dim3((unsigned)b, (unsigned)frames);
}

// The per-frame preparation kernels run on a (blocks, frames) grid: blockIdx.y is the frame (or pair) and the element
// index inside it is 32-bit, so that splitting it into (row, column) is one 32-bit division instead of two 64-bit
// divisions of a clip-wide index (which cost more than the work of these kernels).
#define FRAME_STRIDE(t, per_frame) \
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < (unsigned)(per_frame); t += gridDim.x * blockDim.x)

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

// mode 0: exact 2x2, 1: exact kx x ky integer boxes, 2: general, 3: exact 4x4 boxes on dword-aligned rows (the 960x540 ->
// 240x135 level of every 1080p / 4K clip: four dword loads and four v_sad_u8 per output instead of sixteen byte loads)
__global__ __launch_bounds__(256) void area_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int sh,
                                                      int sw, int dh, int dw, int mode, int kx, int ky, double scale_x,
                                                      double scale_y)
{
    const int f = blockIdx.y;
    const uint8_t* S = src + (size_t)f * sh * sw;
    uint8_t* D = dst + (size_t)f * dh * dw;
    FRAME_STRIDE(t, dh * dw) {
        const int y = (int)(t / (unsigned)dw), x = (int)(t - (unsigned)y * (unsigned)dw);
        int o;
        if (mode == 0) {
            const uint8_t* s0 = S + (size_t)(2 * y) * sw + 2 * x;
            o = (s0[0] + s0[1] + s0[sw] + s0[sw + 1] + 2) >> 2;
        } else if (mode == 3) {
            const unsigned* p = reinterpret_cast<const unsigned*>(S + (size_t)(4 * y) * sw + 4 * x);
            const int rs = sw >> 2;
            unsigned sum = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) sum = __builtin_amdgcn_sad_u8(p[(size_t)j * rs], 0u, sum);
            o = sat_u8_round((int)sum * (1.f / 16));
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
        D[t] = (uint8_t)o;
    }
}

// General ratio with the tap tables in LDS: one workgroup per (frame, AREA_ROWS output rows) forms the column table once
// and one row table per output row (area_axis is fp64 with three divisions: per output pixel, as in mode 2 above, it
// made the three small pyramid levels of a 1080p clip cost more than the 4x4 level that reads sixteen times the bytes).
// Same taps, same f32 sums in the same order as mode 2.
constexpr int AREA_ROWS = 32, AREA_MAX_COLS = 1024;
struct AreaTaps { int n, first; float a[8]; };   // the taps of an output sample are consecutive source samples

// area_axis without the indexed tap arrays (which live in scratch memory): a leading partial sample, full samples, a
// trailing partial one.  The host admits ratios below 6, so the k < 7 / k < 8 guards of area_axis never cut a run short.
__device__ __forceinline__ void area_taps(int d, int ssize, double scale, AreaTaps& t)
{
    const double fsx1 = d * scale, fsx2 = fsx1 + scale;
    const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
    int sx1 = d_ceil(fsx1), sx2 = d_floor(fsx2);
    sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
    sx1 = sx1 < sx2 ? sx1 : sx2;
    const int lead = (sx1 - fsx1 > 1e-3) ? 1 : 0;
    const int k_end = lead + (sx2 - sx1);
    const bool tail = fsx2 - sx2 > 1e-3;
    double at = fsx2 - sx2;
    at = at < 1. ? at : 1.;
    at = at < cell ? at : cell;
    const float a_lead = (float)((sx1 - fsx1) / cell), a_full = (float)(1.0 / cell), a_tail = (float)(at / cell);
    t.n = k_end + (tail ? 1 : 0);
    t.first = sx1 - lead;
#pragma unroll
    for (int k = 0; k < 8; k++) t.a[k] = (k < lead) ? a_lead : (k < k_end) ? a_full : (k == k_end && tail) ? a_tail : 0.f;
}

__global__ __launch_bounds__(256) void area_general_rows_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int sh,
                                                                int sw, int dh, int dw, double scale_x, double scale_y)
{
    __shared__ AreaTaps s_x[AREA_MAX_COLS];
    __shared__ AreaTaps s_y[AREA_ROWS];
    const int groups = (dh + AREA_ROWS - 1) / AREA_ROWS;
    const int f = (int)blockIdx.x / groups, y0 = ((int)blockIdx.x % groups) * AREA_ROWS;
    const int rows = min(AREA_ROWS, dh - y0);
    for (int x = threadIdx.x; x < dw; x += blockDim.x) area_taps(x, sw, scale_x, s_x[x]);
    if ((int)threadIdx.x < rows) area_taps(y0 + (int)threadIdx.x, sh, scale_y, s_y[threadIdx.x]);
    __syncthreads();
    const uint8_t* S = src + (size_t)f * sh * sw;
    uint8_t* D = dst + ((size_t)f * dh + y0) * dw;
    for (int t = threadIdx.x; t < rows * dw; t += blockDim.x) {
        const int ry = t / dw, x = t - ry * dw;
        const AreaTaps& ax = s_x[x];
        const AreaTaps& ay = s_y[ry];
        float sum = 0.f;
        for (int j = 0; j < ay.n; j++) {
            const uint8_t* row = S + (size_t)(ay.first + j) * sw + ax.first;
            float buf = 0.f;
            for (int k = 0; k < ax.n; k++) buf += row[k] * ax.a[k];
            const float term = ay.a[j] * buf;
            sum = (j == 0) ? term : sum + term;
        }
        D[t] = (uint8_t)sat_u8_round(sum);
    }
}

// ---- the pyramid's tail in ONE launch ------------------------------------------------------------------------------------
// The levels above the finest one are tiny (120 x 67, 60 x 33, 30 x 16 for a 1080p clip) and each is the INTER_AREA of the
// previous: as separate launches they are three latency-bound kernels in a row on the call's stream (28 + 16 + 6 us per
// 256-frame clip, profiles/r05_dis_small_steps.md).  One workgroup per frame keeps the chain on chip: the finest level is
// read into LDS once, every further level is formed from the previous one's LDS copy (tap tables of a level computed once
// per workgroup, as in area_general_rows_kernel) and written to both LDS and its plane.  Per output the same taps, the same
// f32 sums in the same order as area_u8_kernel's modes 0 / 1 / 2 (whichever launch_area would pick for that level).
constexpr int TAIL_MAX_LEVELS = 8;
#ifdef VSTAB_FUSED_TRACE
#define TAIL_MARK(i) do { if (prof_) { const long long now_ = wall_clock64(); a.dbg[i] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define TAIL_MARK(i)
#endif
struct TailLevel { uint8_t* dst; int h, w, mode, kx, ky; double scale_x, scale_y; };
struct TailArgs {
#ifdef VSTAB_FUSED_TRACE
    long long* dbg;
#endif
    const uint8_t* src;   // [n][h0][w0]: the finest level
    int n, h0, w0, levels;
    TailLevel lv[TAIL_MAX_LEVELS];
    // the COARSEST level's preparation in the same launch (it is needed at once, by the chain's first patch search): padded
    // copy, Sobel gradients, structure tensor -- what pad_replicate / sobel / tensor_h / tensor_v_kernel do for the other
    // levels on the second stream.  prep_ext == nullptr: not requested.
    uint8_t* prep_ext;    // [n][h + 32][w + 32]
    short* prep_ix;       // [n][h][w]
    short* prep_iy;
    float* prep_tensor;   // 5 planes of [n][hs][ws]
    int prep_ws, prep_hs, prep_lds_off;   // patch grid of the coarsest level; byte offset of the preparation's LDS scratch
};

// 1024 threads: the phases are chains of LDS round trips; with four wavefronts on a CU (one per SIMD) every instruction's latency was
// exposed (level 3 alone 25 us), with sixteen they overlap.
__global__ __launch_bounds__(1024) void pyramid_tail_kernel(TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_lds[];
    const int f = blockIdx.x;
#ifdef VSTAB_FUSED_TRACE
    long long tprev_ = wall_clock64();
    const bool prof_ = a.dbg && blockIdx.x == 7 && threadIdx.x == 0;
#endif
    const int cap0 = (a.h0 * a.w0 + 15) & ~15;                // level images ping-pong between two buffers
    const int cap1 = (a.lv[0].h * a.lv[0].w + 15) & ~15;
    unsigned char* bufA = tail_lds;
    unsigned char* bufB = tail_lds + cap0;
    AreaTaps* s_x = reinterpret_cast<AreaTaps*>(tail_lds + cap0 + cap1);
    AreaTaps* s_y = s_x + a.lv[0].w;
    {
        const uint8_t* S = a.src + (size_t)f * a.h0 * a.w0;
        const int npx = a.h0 * a.w0;
        if ((reinterpret_cast<uintptr_t>(S) & 3) == 0 && (npx & 3) == 0) {
            const unsigned* s4 = reinterpret_cast<const unsigned*>(S);
            unsigned* d4 = reinterpret_cast<unsigned*>(bufA);
            for (int k = threadIdx.x; k < npx / 4; k += blockDim.x) d4[k] = s4[k];
        } else {
            for (int k = threadIdx.x; k < npx; k += blockDim.x) bufA[k] = S[k];
        }
    }
    __syncthreads();
    TAIL_MARK(0);
    int sh = a.h0, sw = a.w0;
    unsigned char* cur = bufA;
    unsigned char* nxt = bufB;
    for (int l = 0; l < a.levels; l++) {
        const TailLevel& L = a.lv[l];
        const int dh = L.h, dw = L.w;
        if (L.mode == 2) {
            for (int x = threadIdx.x; x < dw; x += blockDim.x) area_taps(x, sw, L.scale_x, s_x[x]);
            for (int y = threadIdx.x; y < dh; y += blockDim.x) area_taps(y, sh, L.scale_y, s_y[y]);
            __syncthreads();
        }
        uint8_t* D = L.dst + (size_t)f * dh * dw;
        if (L.mode == 2 && L.kx <= 4 && L.ky <= 4 && dw <= (int)blockDim.x) {
            // General ratio with at most four taps per axis (any halving step: ratio 2 .. 2.1): a thread keeps its COLUMN's taps
            // in registers and walks down the rows; the sixteen candidate source bytes of an output are read together
            // (clamped addresses) and the absent taps are skipped by selects -- the same products added in the same order as
            // the tap loops below, without their chain of dependent LDS reads (49 -> ~20 us per 256-frame clip).
            const int rows_par = (int)blockDim.x / dw;
            const int x = (int)threadIdx.x % dw, ry = (int)threadIdx.x / dw;
            if (ry < rows_par) {
                const AreaTaps ax = s_x[x];
                const int x0 = ax.first;
                for (int y = ry; y < dh; y += rows_par) {
                    const AreaTaps& ay = s_y[y];
                    float sum = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const unsigned char* row = cur + min(ay.first + j, sh - 1) * sw;
                        float buf = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const float v = (float)row[min(x0 + k, sw - 1)];
                            if (k < ax.n) buf += v * ax.a[k];
                        }
                        const float term = ay.a[j] * buf;
                        if (j < ay.n) sum = (j == 0) ? term : sum + term;
                    }
                    const int o = sat_u8_round(sum);
                    nxt[y * dw + x] = (unsigned char)o;
                    D[y * dw + x] = (uint8_t)o;
                }
            }
        } else
        for (int t = threadIdx.x; t < dh * dw; t += blockDim.x) {
            const int y = t / dw, x = t - y * dw;
            int o;
            if (L.mode == 0) {
                const unsigned char* s0 = cur + (2 * y) * sw + 2 * x;
                o = (s0[0] + s0[1] + s0[sw] + s0[sw + 1] + 2) >> 2;
            } else if (L.mode == 1) {
                int sum = 0;
                for (int j = 0; j < L.ky; j++)
                    for (int i = 0; i < L.kx; i++) sum += cur[(y * L.ky + j) * sw + x * L.kx + i];
                o = sat_u8_round(sum * (1.f / (L.kx * L.ky)));
            } else {
                const AreaTaps& ax = s_x[x];
                const AreaTaps& ay = s_y[y];
                float sum = 0.f;
                for (int j = 0; j < ay.n; j++) {
                    const unsigned char* row = cur + (ay.first + j) * sw + ax.first;
                    float buf = 0.f;
                    for (int k = 0; k < ax.n; k++) buf += row[k] * ax.a[k];
                    const float term = ay.a[j] * buf;
                    sum = (j == 0) ? term : sum + term;
                }
                o = sat_u8_round(sum);
            }
            nxt[t] = (unsigned char)o;
            D[t] = (uint8_t)o;
        }
        __syncthreads();
        TAIL_MARK(1 + l);
        unsigned char* tmp = cur; cur = nxt; nxt = tmp;
        sh = dh; sw = dw;
    }
    if (a.prep_ext != nullptr) {
        // `cur` = the coarsest level [sh][sw].  Same values as the per-level kernels produce (integers throughout, except the
        // vertical running sums, which keep OpenCV's sequential f32 order exactly as tensor_v_kernel does).
        const int h = sh, w = sw, ws = a.prep_ws, hs = a.prep_hs;
        const int we = w + 2 * DIS_BORDER, he = h + 2 * DIS_BORDER;
        short* lx = reinterpret_cast<short*>(tail_lds + a.prep_lds_off);
        short* ly = lx + h * w;
        float* laux = reinterpret_cast<float*>(tail_lds + a.prep_lds_off + ((4 * h * w + 15) & ~15));   // [5][h][ws]
        uint8_t* E = a.prep_ext + (size_t)f * he * we;
        for (int t = threadIdx.x; t < he * we; t += blockDim.x) {
            const int y = t / we, x = t - y * we;
            E[t] = cur[clampi(y - DIS_BORDER, 0, h - 1) * w + clampi(x - DIS_BORDER, 0, w - 1)];
        }
        short* Dx = a.prep_ix + (size_t)f * h * w;
        short* Dy = a.prep_iy + (size_t)f * h * w;
        for (int t = threadIdx.x; t < h * w; t += blockDim.x) {
            const int y = t / w, x = t - y * w;
            const unsigned char* r0 = cur + reflect101(y - 1, h) * w;
            const unsigned char* r1 = cur + y * w;
            const unsigned char* r2 = cur + reflect101(y + 1, h) * w;
            const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
            const int gx = (r0[xr] + 2 * r1[xr] + r2[xr]) - (r0[xl] + 2 * r1[xl] + r2[xl]);
            const int gy = (r2[xl] + 2 * r2[x] + r2[xr]) - (r0[xl] + 2 * r0[x] + r0[xr]);
            Dx[t] = (short)gx; Dy[t] = (short)gy;
            lx[t] = (short)gx; ly[t] = (short)gy;
        }
        __syncthreads();
        const int aplane = h * ws;
        for (int t = threadIdx.x; t < h * ws; t += blockDim.x) {
            const int row = t / ws, js = t - row * ws;
            const short* xr = lx + row * w + js * PSTR;
            const short* yr = ly + row * w + js * PSTR;
            int s_xx = 0, s_yy = 0, s_xy = 0, s_x = 0, s_y = 0;
#pragma unroll
            for (int j = 0; j < PSZ; j++) {
                const int gx = xr[j], gy = yr[j];
                s_xx += gx * gx; s_yy += gy * gy; s_xy += gx * gy; s_x += gx; s_y += gy;
            }
            laux[t] = (float)s_xx; laux[aplane + t] = (float)s_yy; laux[2 * aplane + t] = (float)s_xy;
            laux[3 * aplane + t] = (float)s_x; laux[4 * aplane + t] = (float)s_y;
        }
        __syncthreads();
        const size_t oplane = (size_t)a.n * hs * ws;
        for (int t = threadIdx.x; t < 5 * ws; t += blockDim.x) {
            const int k = t / ws, j = t - k * ws;
            const float* av = laux + k * aplane + j;
            float* o = a.prep_tensor + k * oplane + (size_t)f * hs * ws + j;
            float ring[PSZ];
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < PSZ; i++) { ring[i] = av[i * ws]; sum += ring[i]; }
            o[0] = sum;
            int is = 1;
            for (int i0 = PSZ; i0 < h; i0 += PSZ) {
                float v[PSZ];
#pragma unroll
                for (int q = 0; q < PSZ; q++) v[q] = (i0 + q < h) ? av[(i0 + q) * ws] : 0.f;
#pragma unroll
                for (int q = 0; q < PSZ; q++) {
                    if (i0 + q < h) {
                        sum += (v[q] - ring[q]);
                        ring[q] = v[q];
                        if ((q + 1) % PSTR == 0) { o[(size_t)is * ws] = sum; is++; }
                    }
                }
            }
        }
#ifdef VSTAB_FUSED_TRACE
        __syncthreads();
#endif
        TAIL_MARK(6);
    }
}

__global__ __launch_bounds__(256) void pad_replicate_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int h, int w)
{
    const int we = w + 2 * DIS_BORDER, he = h + 2 * DIS_BORDER;
    const int f = blockIdx.y;
    const uint8_t* S = src + (size_t)f * h * w;
    uint8_t* D = dst + (size_t)f * he * we;
    FRAME_STRIDE(t, he * we) {
        const int y = (int)(t / (unsigned)we), x = (int)(t - (unsigned)y * (unsigned)we);
        const int sx = clampi(x - DIS_BORDER, 0, w - 1), sy = clampi(y - DIS_BORDER, 0, h - 1);
        D[t] = S[(unsigned)(sy * w + sx)];
    }
}

__global__ __launch_bounds__(256) void sobel_kernel(const uint8_t* __restrict__ I, short* __restrict__ Ix, short* __restrict__ Iy, int n, int h, int w)
{
    const int f = blockIdx.y;
    const uint8_t* S = I + (size_t)f * h * w;
    short* Dx = Ix + (size_t)f * h * w;
    short* Dy = Iy + (size_t)f * h * w;
    FRAME_STRIDE(t, h * w) {
        const int y = (int)(t / (unsigned)w), x = (int)(t - (unsigned)y * (unsigned)w);
        const uint8_t* r0 = S + (unsigned)(reflect101(y - 1, h) * w);
        const uint8_t* r1 = S + (unsigned)(y * w);
        const uint8_t* r2 = S + (unsigned)(reflect101(y + 1, h) * w);
        const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
        const int gx = (r0[xr] + 2 * r1[xr] + r2[xr]) - (r0[xl] + 2 * r1[xl] + r2[xl]);
        const int gy = (r2[xl] + 2 * r2[x] + r2[xr]) - (r0[xl] + 2 * r0[x] + r0[xr]);
        Dx[t] = (short)gx;
        Dy[t] = (short)gy;
    }
}

// precomputeStructureTensor, horizontal sums of every patch column position (frame, row, js): OpenCV keeps five f32
// running sums per row (`sum += x[j]*x[j] - x[j-8]*x[j-8]`, integer increments added to a float).  Every value such a
// sum ever takes is an integer below 2^24 (8 terms of at most 1020^2: Sobel of u8 is within +-1020), so each of those
// float additions is exact and the running sum equals the plain integer window sum -- computed here independently per
// output, no sequential chain (the vertical pass, whose sums exceed 2^24, does keep OpenCV's order).
__global__ __launch_bounds__(256) void tensor_h_kernel(const short* __restrict__ Ix, const short* __restrict__ Iy, float* __restrict__ aux,
                                                       int n, int h, int w, int ws)
{
    const size_t plane = (size_t)n * h * ws;
    const int f = blockIdx.y;
    const short* Fx = Ix + (size_t)f * h * w;
    const short* Fy = Iy + (size_t)f * h * w;
    float* A = aux + (size_t)f * h * ws;
    FRAME_STRIDE(t, h * ws) {
        const int row = (int)(t / (unsigned)ws), js = (int)(t - (unsigned)row * (unsigned)ws);
        const short* xr = Fx + (unsigned)(row * w + js * PSTR);
        const short* yr = Fy + (unsigned)(row * w + js * PSTR);
        int s_xx = 0, s_yy = 0, s_xy = 0, s_x = 0, s_y = 0;
#pragma unroll
        for (int j = 0; j < PSZ; j++) {
            const int gx = xr[j], gy = yr[j];
            s_xx += gx * gx; s_yy += gy * gy; s_xy += gx * gy; s_x += gx; s_y += gy;
        }
        float* o = A + t;
        o[0] = (float)s_xx; o[plane] = (float)s_yy; o[2 * plane] = (float)s_xy; o[3 * plane] = (float)s_x; o[4 * plane] = (float)s_y;
    }
}

// vertical running sums: one thread per (quantity, frame, js)
__global__ __launch_bounds__(64) void tensor_v_kernel(const float* __restrict__ aux, float* __restrict__ out, int n, int h, int ws, int hs)
{
    // grid (column blocks, frames, 5 quantities): no index arithmetic beyond the column
    const size_t aplane = (size_t)n * h * ws, oplane = (size_t)n * hs * ws;
    const int f = blockIdx.y, k = blockIdx.z;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < ws; j += gridDim.x * blockDim.x) {
        const float* a = aux + k * aplane + (size_t)f * h * ws + j;
        float* o = out + k * oplane + (size_t)f * hs * ws + j;
        // the last PSZ rows stay in registers (the row that leaves the window is not read twice), and the PSZ loads of a
        // chunk are independent of the running sum: same additions in the same order
        float ring[PSZ];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < PSZ; i++) { ring[i] = a[(size_t)i * ws]; sum += ring[i]; }
        o[0] = sum;
        int is = 1;
        for (int i0 = PSZ; i0 < h; i0 += PSZ) {
            float v[PSZ];
#pragma unroll
            for (int k = 0; k < PSZ; k++) v[k] = (i0 + k < h) ? a[(size_t)(i0 + k) * ws] : 0.f;
#pragma unroll
            for (int k = 0; k < PSZ; k++) {
                if (i0 + k < h) {
                    sum += (v[k] - ring[k]);
                    ring[k] = v[k];
                    if ((k + 1) % PSTR == 0) { o[(size_t)is * ws] = sum; is++; }   // (i - PSZ + 1) % PSTR == 0 for i = i0 + k, i0 a multiple of PSZ
                }
            }
        }
    }
}

// ---- patch inverse search ------------------------------------------------------------------
// The four per-patch sums (sum d, sum d^2, sum d*Ix, sum d*Iy over the 8x8 patch) are formed in OpenCV's own f32
// association (video/src/dis_flow.cpp, CV_SIMD128 branch of processPatchMeanNorm / computeSSDMeanNorm, patch size 8): one
// 4-lane accumulator; row after row adds (left half + right half) of its 8 terms -- accumulator lane l collects
// columns l and l+4 -- and v_reduce_sum folds the lanes as (a0 + a2) + (a1 + a3).  These sums feed the branches
// `SSD >= prev_SSD` and `cur_SSD < min_SSD`, where a last-bit difference moves a patch by a whole descent step
// (round 1 used an XOR butterfly here; measured against this order it changed 0.4 % of the sampled flow vectors by
// more than 1e-3 px, profiles/r02_dis_sum_order.md), so the association is part of the arithmetic contract.
// Its realisation on a wavefront is quarter_sums<N> below (pis4_kernel).
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
    int spin_limit;        // a wavefront's total spin allowance over its LDS progress-counter waits (the test build's VSTAB_DEBUG_PIS_SPIN_LIMIT overrides it)
    int* status;           // host-mapped status word of the context (vstab_internal.h): a timed-out wait is reported there
};

// Two 1024-thread workgroups per frame pair; each owns 4 of OpenCV's 8 fixed stripes and runs 4 waves per
// stripe.  Inside a stripe the raster recurrence (left + top in the forward pass, right + bottom in the
// backward pass) is kept exactly, but rows are software-pipelined along anti-diagonals: wave k of a stripe
// handles rows k, k+4, ... and waits on an LDS progress counter of the row it depends on.  The padded I1
// level image (<= 74 KB) and the block's sparse flow live in LDS, so the dependent chain
// candidate -> bilinear window -> sums -> update never leaves the CU.
constexpr int PIS_STRIPES_PER_BLOCK = 4;

__device__ __forceinline__ void wait_progress(volatile int* counter, int need, int& budget, int* status)
{
    // bounded spin (every wave of the workgroup is resident, so the producer always makes progress; the bound turns a
    // logic error into a reported failure instead of a hung GPU: the wave records VSTAB_STATUS_PIS_TIMEOUT in the
    // context's host-visible status word and carries on, and the next host synchronisation point of the library
    // (vstab_sample_fit_batch, vstab_synchronize) returns non-zero with vstab_last_error() set).
    // `budget` is the wavefront's TOTAL spin allowance for the kernel, not a per-wait one: a wave of the two-wavefront
    // form waits up to 2 x ws ~ 120 times per level; with a per-wait bound of 2^22 spins (~0.3 s each at ~170 clocks per
    // spin) waits expiring one after another could have stacked to ~40 s per level -- minutes per clip -- before the
    // failure surfaced.  With one allowance a broken dependency costs a wave at most ~0.3 s per kernel.
    while (budget > 0) {
        if (__hip_atomic_load(const_cast<int*>(counter), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) return;
        __builtin_amdgcn_s_sleep(1);
        budget--;
    }
    if (__hip_atomic_load(const_cast<int*>(counter), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) return;
    if (status != nullptr && (threadIdx.x & 63) == 0)
        __hip_atomic_fetch_or(status, VSTAB_STATUS_PIS_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- patch inverse search, FOUR patch rows per wavefront ----------------------------------------
// Two workgroups per frame pair, each owning 4 of OpenCV's 8 fixed stripes (the stripes are what makes the result
// independent of the thread count); the padded I1 level image (<= 74 KB) and the block's sparse flow live in LDS, so the
// dependent chain candidate -> bilinear window -> sums -> update never leaves the CU.  Inside a stripe the raster
// recurrence (left + top in the forward pass, right + bottom in the backward pass) is kept exactly.
// A wavefront holds four 8x8 patches, one per DPP row of 16 lanes, belonging to four consecutive patch rows of a stripe, each row trailing the previous one by one patch (the
// raster dependency: left neighbour = own previous patch, vertical neighbour = the previous DPP row's previous patch).
// Lane (rr, l) of a row owns the patch pixels (2rr, l), (2rr, l+4), (2rr+1, l), (2rr+1, l+4).  OpenCV's association of
// the patch sums -- accumulator lane l adds (left + right) of row 0, 1, ... 7 in order -- becomes: x = left + right of
// row 2rr, y = the same of row 2rr+1 (in-lane); bank 0: x += y; banks 1..3 in turn: x = x(bank-1) + x, then x += y;
// fold; all inside one DPP row, so no cross-row hop, and the total reaches the row's 16 lanes with two `row_ror` moves.
// Per sum 11 DPP instructions serve four patches, and the per-patch uniform arithmetic (bilinear weights, update) is
// issued once for four patches: the kernel is VALU-issue-bound (profiles/r02_pmc_kernels.md).  Its predecessor held two
// patches per wavefront (32 lanes each, 16 chain instructions per sum incl. a v_permlane16_swap hop between its two DPP
// rows): DIS 4.29 -> 3.93 ms per 256-frame clip (profiles/r02_dis_launch_forms.md).
#define Q_B0(X, Y) "v_add_f32_dpp " X ", " Y ", " X " quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x1\n\t"
#define Q_SH(X, K) "v_add_f32_dpp " X ", " X ", " X " row_shr:4 row_mask:0xf bank_mask:" K "\n\t"
#define Q_AD(X, Y, K) "v_add_f32_dpp " X ", " Y ", " X " quad_perm:[0,1,2,3] row_mask:0xf bank_mask:" K "\n\t"
#define Q_F1(X) "v_add_f32_dpp " X ", " X ", " X " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0x8\n\t"
#define Q_F2(X) "v_add_f32_dpp " X ", " X ", " X " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0x8\n\t"
#define Q_C1(X) "v_mov_b32_dpp " X ", " X " row_ror:4 row_mask:0xf bank_mask:0x1\n\t"
#define Q_C2(X) "v_mov_b32_dpp " X ", " X " row_ror:8 row_mask:0xf bank_mask:0x6\n\t"
// x[k] / y[k]: this lane's (left + right) terms of its two patch rows for sum k; on return x[k] = the patch total in all
// 16 lanes of the DPP row.  One asm statement: with four sums interleaved every register written by one instruction is
// read by a DPP operand three instructions later at the earliest (2 wait states needed); with two sums an `s_nop 0` per
// step supplies the missing one.
template <int N>
__device__ __forceinline__ void quarter_sums(float (&x)[N], const float (&y)[N])
{
    static_assert(N == 2 || N == 4, "two sums for an SSD evaluation, four for a descent step");
    if constexpr (N == 4) {
        asm("s_nop 1\n\t"
            Q_B0("%0", "%4") Q_B0("%1", "%5") Q_B0("%2", "%6") Q_B0("%3", "%7")
            Q_SH("%0", "0x2") Q_SH("%1", "0x2") Q_SH("%2", "0x2") Q_SH("%3", "0x2")
            Q_AD("%0", "%4", "0x2") Q_AD("%1", "%5", "0x2") Q_AD("%2", "%6", "0x2") Q_AD("%3", "%7", "0x2")
            Q_SH("%0", "0x4") Q_SH("%1", "0x4") Q_SH("%2", "0x4") Q_SH("%3", "0x4")
            Q_AD("%0", "%4", "0x4") Q_AD("%1", "%5", "0x4") Q_AD("%2", "%6", "0x4") Q_AD("%3", "%7", "0x4")
            Q_SH("%0", "0x8") Q_SH("%1", "0x8") Q_SH("%2", "0x8") Q_SH("%3", "0x8")
            Q_AD("%0", "%4", "0x8") Q_AD("%1", "%5", "0x8") Q_AD("%2", "%6", "0x8") Q_AD("%3", "%7", "0x8")
            Q_F1("%0") Q_F1("%1") Q_F1("%2") Q_F1("%3") Q_F2("%0") Q_F2("%1") Q_F2("%2") Q_F2("%3")
            Q_C1("%0") Q_C1("%1") Q_C1("%2") Q_C1("%3") Q_C2("%0") Q_C2("%1") Q_C2("%2") Q_C2("%3")
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));
    } else {
#define Q_N "s_nop 0\n\t"
        asm("s_nop 1\n\t"
            Q_B0("%0", "%2") Q_B0("%1", "%3")
            Q_N Q_SH("%0", "0x2") Q_SH("%1", "0x2") Q_N Q_AD("%0", "%2", "0x2") Q_AD("%1", "%3", "0x2")
            Q_N Q_SH("%0", "0x4") Q_SH("%1", "0x4") Q_N Q_AD("%0", "%2", "0x4") Q_AD("%1", "%3", "0x4")
            Q_N Q_SH("%0", "0x8") Q_SH("%1", "0x8") Q_N Q_AD("%0", "%2", "0x8") Q_AD("%1", "%3", "0x8")
            Q_N Q_F1("%0") Q_F1("%1") Q_N Q_F2("%0") Q_F2("%1")
            Q_N Q_C1("%0") Q_C1("%1") Q_N Q_C2("%0") Q_C2("%1")
            : "+v"(x[0]), "+v"(x[1]) : "v"(y[0]), "v"(y[1]));
#undef Q_N
    }
}
#undef Q_B0
#undef Q_SH
#undef Q_AD
#undef Q_F1
#undef Q_F2
#undef Q_C1
#undef Q_C2

// PIS4_WAVES wavefronts per stripe share its groups of four rows round-robin (1 up to four rows per stripe, else 2).
template <int PIS4_WAVES>
__global__ __launch_bounds__(64 * PIS4_WAVES * PIS_STRIPES_PER_BLOCK) void pis4_kernel(PisArgs a)
{
    extern __shared__ unsigned char pis_lds[];
    const int pair = blockIdx.x >> 1, half = blockIdx.x & 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int Q = lane >> 4, t = lane & 15, rr = t >> 2, c2 = t & 3;
    const int stripe = half * PIS_STRIPES_PER_BLOCK + wave / PIS4_WAVES, pw = wave % PIS4_WAVES;
    const int w = a.w, h = a.h, ws = a.ws, hs = a.hs;
    const int w_ext = w + 2 * DIS_BORDER, h_ext = h + 2 * DIS_BORDER;
    const int img_bytes = (w_ext * h_ext + 15) & ~15;
    unsigned char* lI1 = pis_lds;
    float* lSx = reinterpret_cast<float*>(pis_lds + img_bytes);
    float* lSy = lSx + hs * ws;
    int* done0 = reinterpret_cast<int*>(lSy + hs * ws);
    int* done1 = done0 + hs;
    {
        const unsigned char* sb = a.Iext + (size_t)(pair + 1) * h_ext * w_ext;
        const bool aligned = ((reinterpret_cast<uintptr_t>(sb) & 15) == 0);
        const int nvec = aligned ? (w_ext * h_ext) / 16 : 0;
        const uint4* src = reinterpret_cast<const uint4*>(sb);
        uint4* dst = reinterpret_cast<uint4*>(lI1);
        for (int k = threadIdx.x; k < nvec; k += blockDim.x) dst[k] = src[k];
        for (int k = nvec * 16 + threadIdx.x; k < w_ext * h_ext; k += blockDim.x) lI1[k] = sb[k];
        for (int k = threadIdx.x; k < 2 * hs; k += blockDim.x) done0[k] = 0;
    }
    __syncthreads();
    const int row_lo = min(stripe * a.stripe_sz, hs), row_hi = min((stripe + 1) * a.stripe_sz, hs);
    const int nrows = row_hi - row_lo, ngroups = (nrows + 3) >> 2;
    const uint8_t* I0 = a.I + (size_t)pair * h * w;
    const short* Ix = a.Ix + (size_t)pair * h * w;
    const short* Iy = a.Iy + (size_t)pair * h * w;
    const size_t tplane = (size_t)a.n * hs * ws;
    const float* T = a.tensor + (size_t)pair * hs * ws;
    const float* U = a.U + (size_t)pair * h * w;
    const float* V = a.V + (size_t)pair * h * w;
    const float i_lo = DIS_BORDER - PSZ + 1.0f, i_hi = DIS_BORDER + h - 1.0f;
    const float j_lo = DIS_BORDER - PSZ + 1.0f, j_hi = DIS_BORDER + w - 1.0f;
    const int num_inner_iter = GD_ITERS / 2;
    const float nn = (float)(PSZ * PSZ);
    const int lane_off4 = 2 * rr * w_ext + c2;

    int spin_budget = a.spin_limit;   // all dependency waits of this wavefront together (see wait_progress)
    for (int iter = 0; iter < 2; iter++) {
        const int dir = (iter == 0) ? 1 : -1;
        const int start_is = (iter == 0) ? row_lo : row_hi - 1;
        const int start_js = (iter == 0) ? 0 : ws - 1;
        volatile int* done = (iter == 0) ? done0 : done1;
        for (int k = pw; k < ngroups; k += PIS4_WAVES) {
            const int row_a = start_is + dir * 4 * k;                 // leading row of the group in this pass
            const int nq = min(4, nrows - 4 * k);                     // rows in this group
            const int is = row_a + dir * Q;
            const bool row_ok = Q < nq;
            const int i = is * PSTR;
            if (iter == 1) {
                // the backward pass starts from each row's forward-pass result; the grouping (hence the wavefront that
                // produced it) may differ between the passes, so wait for the forward pass of every row of the group
                for (int q = 0; q < nq; q++) wait_progress(done0 + (row_a + dir * q), ws, spin_budget, a.status);
            }
            // A step's inputs from memory (the lane's four I0 pixels and gradients, the patch's structure tensor and, in the
            // forward pass, its initial flow) depend on the patch position only, never on what the search computes: they
            // are requested ONE STEP AHEAD, before the current step's descent starts, so their L2 / HBM round trip runs
            // under the ~10^4 cycles of dependent arithmetic of a step instead of at the head of the next one (the kernel
            // has two wavefronts per SIMD and nothing else to hide it behind).
            struct StepIn { uint8_t i[4]; short gx[4], gy[4]; float t[5], u0, v0; };
            auto fetch = [&](int visited_) {
                StepIn r;
                const int js_ = start_js + dir * visited_;
                const int j_ = js_ * PSTR;
                const int sidx_ = is * ws + js_;
                const size_t p0 = (size_t)(i + 2 * rr) * w + j_ + c2, p1 = p0 + w;   // patch rows 2rr, 2rr+1; columns c2, c2+4
                r.i[0] = I0[p0]; r.i[1] = I0[p0 + 4]; r.i[2] = I0[p1]; r.i[3] = I0[p1 + 4];
                r.gx[0] = Ix[p0]; r.gx[1] = Ix[p0 + 4]; r.gx[2] = Ix[p1]; r.gx[3] = Ix[p1 + 4];
                r.gy[0] = Iy[p0]; r.gy[1] = Iy[p0 + 4]; r.gy[2] = Iy[p1]; r.gy[3] = Iy[p1 + 4];
#pragma unroll
                for (int c = 0; c < 5; c++) r.t[c] = T[c * tplane + sidx_];
                r.u0 = r.v0 = 0.f;
                if (iter == 0) {
                    r.u0 = U[(size_t)(i + PSZ / 2) * w + j_ + PSZ / 2];
                    r.v0 = V[(size_t)(i + PSZ / 2) * w + j_ + PSZ / 2];
                }
                return r;
            };
            StepIn nxt{};
            if (row_ok && Q == 0) nxt = fetch(0);                      // step 0: only the leading row is active
            for (int s = 0; s < ws + nq - 1; s++) {
                const int visited = s - Q;                             // patches this row finished before this step
                const bool act = row_ok && visited >= 0 && visited < ws;
                const StepIn cur = nxt;
                if (row_ok && visited + 1 >= 0 && visited + 1 < ws) nxt = fetch(visited + 1);
                if (k > 0 && s < ws) wait_progress(done + (row_a - dir), s + 1, spin_budget, a.status);   // leading row's vertical neighbour (previous group)
                if (act) {
                    const int js = start_js + dir * visited;
                    const int j = js * PSTR;
                    const int sidx = is * ws + js;
                    const float i00 = (float)cur.i[0], i01 = (float)cur.i[1], i10 = (float)cur.i[2], i11 = (float)cur.i[3];
                    const float gx00 = (float)cur.gx[0], gx01 = (float)cur.gx[1], gx10 = (float)cur.gx[2], gx11 = (float)cur.gx[3];
                    const float gy00 = (float)cur.gy[0], gy01 = (float)cur.gy[1], gy10 = (float)cur.gy[2], gy11 = (float)cur.gy[3];
                    const float txx = cur.t[0], tyy = cur.t[1], txy = cur.t[2];
                    const float x_grad_sum = cur.t[3], y_grad_sum = cur.t[4];
                    float Sxv, Syv;
                    if (iter == 0) {
                        Sxv = cur.u0;
                        Syv = cur.v0;
                    } else {
                        Sxv = lSx[sidx];
                        Syv = lSy[sidx];
                    }
#define PATCH_DIFF4(bw, d00_, d01_, d10_, d11_)                                                            \
    do {                                                                                                   \
        const unsigned char* q_ = lI1 + (bw).off + lane_off4;                                              \
        const float a0_ = (float)q_[0], a1_ = (float)q_[1], a4_ = (float)q_[4], a5_ = (float)q_[5];        \
        const float b0_ = (float)q_[w_ext], b1_ = (float)q_[w_ext + 1];                                    \
        const float b4_ = (float)q_[w_ext + 4], b5_ = (float)q_[w_ext + 5];                                \
        const float c0_ = (float)q_[2 * w_ext], c1_ = (float)q_[2 * w_ext + 1];                            \
        const float c4_ = (float)q_[2 * w_ext + 4], c5_ = (float)q_[2 * w_ext + 5];                        \
        d00_ = (bw).w00 * a0_ + (bw).w01 * a1_ + (bw).w10 * b0_ + (bw).w11 * b1_ - i00;                    \
        d01_ = (bw).w00 * a4_ + (bw).w01 * a5_ + (bw).w10 * b4_ + (bw).w11 * b5_ - i01;                    \
        d10_ = (bw).w00 * b0_ + (bw).w01 * b1_ + (bw).w10 * c0_ + (bw).w11 * c1_ - i10;                    \
        d11_ = (bw).w00 * b4_ + (bw).w01 * b5_ + (bw).w10 * c4_ + (bw).w11 * c5_ - i11;                    \
    } while (0)
#define SSD_AT4(dst, ux, uy)                                                                               \
    do {                                                                                                   \
        Bilin b_ = bilin_weights(i, j, (ux), (uy), i_lo, i_hi, j_lo, j_hi, w_ext);                         \
        float e00_, e01_, e10_, e11_;                                                                      \
        PATCH_DIFF4(b_, e00_, e01_, e10_, e11_);                                                           \
        float px_[2] = {e00_ + e01_, e00_ * e00_ + e01_ * e01_};                                           \
        const float py_[2] = {e10_ + e11_, e10_ * e10_ + e11_ * e11_};                                     \
        quarter_sums<2>(px_, py_);                                                                         \
        dst = px_[1] - px_[0] * px_[0] / nn;                                                               \
    } while (0)
                    float min_SSD, cur_SSD;
                    SSD_AT4(min_SSD, Sxv, Syv);
                    if (visited > 0) {
                        const float nx = lSx[sidx - dir], ny = lSy[sidx - dir];
                        SSD_AT4(cur_SSD, nx, ny);
                        if (cur_SSD < min_SSD) { min_SSD = cur_SSD; Sxv = nx; Syv = ny; }
                    }
                    if (Q >= 1 || k > 0) {                              // a previously visited row exists in this pass
                        const float nx = lSx[sidx - dir * ws], ny = lSy[sidx - dir * ws];
                        SSD_AT4(cur_SSD, nx, ny);
                        if (cur_SSD < min_SSD) { min_SSD = cur_SSD; Sxv = nx; Syv = ny; }
                    }
                    float cur_Ux = Sxv, cur_Uy = Syv;
                    float detH = txx * tyy - txy * txy;
                    if (__builtin_fabsf(detH) < DIS_EPS) detH = DIS_EPS;
                    const float invH11 = tyy / detH, invH12 = -txy / detH, invH22 = txx / detH;
                    float prev_SSD = DIS_INF;
                    for (int tt = 0; tt < num_inner_iter; tt++) {
                        Bilin b = bilin_weights(i, j, cur_Ux, cur_Uy, i_lo, i_hi, j_lo, j_hi, w_ext);
                        float d00, d01, d10, d11;
                        PATCH_DIFF4(b, d00, d01, d10, d11);
                        float px[4] = {d00 + d01, d00 * d00 + d01 * d01, d00 * gx00 + d01 * gx01, d00 * gy00 + d01 * gy01};
                        const float py[4] = {d10 + d11, d10 * d10 + d11 * d11, d10 * gx10 + d11 * gx11, d10 * gy10 + d11 * gy11};
                        quarter_sums<4>(px, py);
                        const float sum_diff = px[0], sum_sq = px[1], sum_x = px[2], sum_y = px[3];
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
#undef SSD_AT4
#undef PATCH_DIFF4
                    {
                        const double ddx = (double)(cur_Ux - Sxv), ddy = (double)(cur_Uy - Syv);
                        if (__builtin_sqrt(ddx * ddx + ddy * ddy) <= (double)PSZ) { Sxv = cur_Ux; Syv = cur_Uy; }
                    }
                    if (t == 0) {
                        lSx[sidx] = Sxv;
                        lSy[sidx] = Syv;
                        __hip_atomic_store(const_cast<int*>(done + is), visited + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
    const int blk_lo = min(half * PIS_STRIPES_PER_BLOCK * a.stripe_sz, hs);
    const int blk_hi = min((half + 1) * PIS_STRIPES_PER_BLOCK * a.stripe_sz, hs);
    float* Sx = a.Sx + (size_t)pair * hs * ws;
    float* Sy = a.Sy + (size_t)pair * hs * ws;
    for (int k = blk_lo * ws + threadIdx.x; k < blk_hi * ws; k += blockDim.x) { Sx[k] = lSx[k]; Sy[k] = lSy[k]; }
}

// ---- per-pixel phase bodies (shared by every launch shape) -----------------------------------
// All indices below are GLOBAL element indices into the [P][h][w] planes; (x, y) is the pixel.
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f2_t __attribute__((ext_vector_type(2)));
// Planes of the variational refinement, [P][h][w] each.  What a pixel's stage reads together is stored together, so that
// one wide load replaces several dword loads: the vector-memory pipeline of a CU (address processing + L1, shared by its
// four SIMDs) was what the tile stages waited for -- a linear-system pixel issued 18 global_load_dword, and thirteen more
// of them cost +250 us per finest level (profiles/r04_level_kernel.md).
//   avg, Iz   f32     0.5 (I0 + warped I1) and warped I1 - I0: the derivative phase's inputs
//   D0, D1    float4  {Ix, Iy, Iz, Ixx} and {Ixy, Iyy, Ixz, Iyz}: the eight derivative values of the linear system
//   UV        float2  the level's flow (u, v) as the refinement reads it (a copy of U, V made by the warp phase)
//   dA, dB    float2  the increment (du, dv) of the fixed-point iterations, ping-pong
// (the linear system's coefficients live in registers)
struct VrBufs {
    float *avg, *Iz;
    f4_t *D0, *D1;
    f2_t *UV, *dA, *dB;
};

// Plane accesses of the per-pixel phases: every pointer below is the PAIR's plane (a uniform base, SGPR pair) and every
// index a 32-bit in-plane index, turned into a 32-bit byte offset -- the `global_load_dword v, v_off, s[base]` form, with
// no 64-bit address arithmetic per access (a level of a <= 960-px working image has far fewer than 2^30 pixels).
__device__ __forceinline__ float ldf(const float* p, int i) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + (unsigned)i * 4u); }
__device__ __forceinline__ void stf(float* p, int i, float v) { *reinterpret_cast<float*>(reinterpret_cast<char*>(p) + (unsigned)i * 4u) = v; }
__device__ __forceinline__ float ldb(const uint8_t* p, int i) { return (float)p[(unsigned)i]; }
__device__ __forceinline__ f2_t ld2(const f2_t* p, int i) { return *reinterpret_cast<const f2_t*>(reinterpret_cast<const char*>(p) + (unsigned)i * 8u); }
__device__ __forceinline__ void st2(f2_t* p, int i, f2_t v) { *reinterpret_cast<f2_t*>(reinterpret_cast<char*>(p) + (unsigned)i * 8u) = v; }
__device__ __forceinline__ f4_t ld4(const f4_t* p, int i) { return *reinterpret_cast<const f4_t*>(reinterpret_cast<const char*>(p) + (unsigned)i * 16u); }
__device__ __forceinline__ void st4(f4_t* p, int i, f4_t v) { *reinterpret_cast<f4_t*>(reinterpret_cast<char*>(p) + (unsigned)i * 16u) = v; }

__device__ __forceinline__ void densify_px(const uint8_t* __restrict__ I0, const uint8_t* __restrict__ I1, const float* __restrict__ sx,
                                           const float* __restrict__ sy, float* __restrict__ U, float* __restrict__ V, int q, int i,
                                           int j, int h, int w, int ws, int hs, float& u_out, float& v_out)
{
    int end_is = i / PSTR < hs - 1 ? i / PSTR : hs - 1;
    int start_is = i - PSZ >= 0 ? (i - PSZ) / PSTR + 1 : 0;
    if (start_is > end_is) start_is = end_is;
    int end_js = j / PSTR < ws - 1 ? j / PSTR : ws - 1;
    int start_js = j - PSZ >= 0 ? (j - PSZ) / PSTR + 1 : 0;
    if (start_js > end_js) start_js = end_js;
    const float i0 = ldb(I0, i * w + j);
    float sum_coef = 0.f, sum_Ux = 0.f, sum_Uy = 0.f;
    for (int is = start_is; is <= end_is; is++)
        for (int js = start_js; js <= end_js; js++) {
            const float sxv = ldf(sx, is * ws + js), syv = ldf(sy, is * ws + js);
            float j_m = (float)j + sxv, i_m = (float)i + syv;
            j_m = j_m > 0.0f ? j_m : 0.0f;
            j_m = j_m < (float)w - 1.0f - DIS_EPS ? j_m : (float)w - 1.0f - DIS_EPS;
            i_m = i_m > 0.0f ? i_m : 0.0f;
            i_m = i_m < (float)h - 1.0f - DIS_EPS ? i_m : (float)h - 1.0f - DIS_EPS;
            const int j_l = (int)j_m, j_u = j_l + 1, i_l = (int)i_m, i_u = i_l + 1;
            const float diff = (j_m - j_l) * (i_m - i_l) * ldb(I1, i_u * w + j_u) +
                               (j_u - j_m) * (i_m - i_l) * ldb(I1, i_u * w + j_l) +
                               (j_m - j_l) * (i_u - i_m) * ldb(I1, i_l * w + j_u) +
                               (j_u - j_m) * (i_u - i_m) * ldb(I1, i_l * w + j_l) - i0;
            const float ad = __builtin_fabsf(diff);
            const float coef = 1 / (ad > 1.0f ? ad : 1.0f);
            sum_Ux += coef * sxv;
            sum_Uy += coef * syv;
            sum_coef += coef;
        }
    u_out = sum_Ux / sum_coef;
    v_out = sum_Uy / sum_coef;
    stf(U, q, u_out);
    stf(V, q, v_out);
}

// b: the pair's planes (VrBufs shifted to the pair, see level_kernel)
__device__ __forceinline__ void vr_warp_px(const uint8_t* __restrict__ I0, const uint8_t* __restrict__ I1, float u, float v,
                                           const VrBufs& b, int q, int x, int y, int h, int w)
{
    const float mx = x + u, my = y + v;
    const int sx = (int)__builtin_rintf(mx * 32.f), sy = (int)__builtin_rintf(my * 32.f);
    const int ix = sat_short(sx >> 5), iy = sat_short(sy >> 5);
    const int fx = sx & 31, fy = sy & 31;
    const float wx1 = fx * (1.f / 32), wx0 = 1.f - wx1, wy1 = fy * (1.f / 32), wy0 = 1.f - wy1;
    const int x0 = clampi(ix, 0, w - 1), x1 = clampi(ix + 1, 0, w - 1);
    const int y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
    const float v00 = ldb(I1, y0 * w + x0), v01 = ldb(I1, y0 * w + x1);
    const float v10 = ldb(I1, y1 * w + x0), v11 = ldb(I1, y1 * w + x1);
    const float warped = v00 * (wy0 * wx0) + v01 * (wy0 * wx1) + v10 * (wy1 * wx0) + v11 * (wy1 * wx1);
    const float i0 = ldb(I0, q);
    stf(b.avg, q, i0 * 0.5f + warped * 0.5f + 0.f);
    stf(b.Iz, q, warped - i0);
    st2(b.UV, q, f2_t{u, v});
    st2(b.dA, q, f2_t{0.f, 0.f});
}

// calcDerivatives of OpenCV's variational refinement for one pixel: central differences with replicated borders, first
// order of avg and Iz, second order as the first-order operator applied to Ix / Iy -- evaluated straight from the two
// planes everything comes from (the intermediate Ix / Iy values are recomputed at the neighbours instead of stored and
// re-read: the same f32 subtractions of the same stored values, so the same bits) and written once, together.
__device__ __forceinline__ void vr_deriv_px(const VrBufs& b, int q, int x, int y, int h, int w)
{
    const int xl = max(x - 1, 0), xr = min(x + 1, w - 1), yu = max(y - 1, 0), yd = min(y + 1, h - 1);
    const int row = y * w, rowu = yu * w, rowd = yd * w;
    const float* __restrict__ A = b.avg;
    const float* __restrict__ Z = b.Iz;
    const float Ix = ldf(A, row + xr) - ldf(A, row + xl);
    const float Iy = ldf(A, rowd + x) - ldf(A, rowu + x);
    const float Iz = ldf(Z, q);
    const float Ixz = ldf(Z, row + xr) - ldf(Z, row + xl);
    const float Iyz = ldf(Z, rowd + x) - ldf(Z, rowu + x);
    const float Ixx = (ldf(A, row + min(xr + 1, w - 1)) - ldf(A, row + max(xr - 1, 0))) - (ldf(A, row + min(xl + 1, w - 1)) - ldf(A, row + max(xl - 1, 0)));
    const float Ixy = (ldf(A, rowd + xr) - ldf(A, rowd + xl)) - (ldf(A, rowu + xr) - ldf(A, rowu + xl));
    const float Iyy = (ldf(A, min(yd + 1, h - 1) * w + x) - ldf(A, max(yd - 1, 0) * w + x)) -
                      (ldf(A, min(yu + 1, h - 1) * w + x) - ldf(A, max(yu - 1, 0) * w + x));
    st4(b.D0, q, f4_t{Ix, Iy, Iz, Ixx});
    st4(b.D1, q, f4_t{Ixy, Iyy, Ixz, Iyz});
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
                                            float* __restrict__ dV, int q, int dx, int dy, int sh, int sw,
                                            double scale_x, double scale_y, float mul)
{
    // sU / sV: the pair's source level, dU / dV: the pair's destination level (uniform bases), q: destination pixel
    int sx, sy;
    float fx, fy;
    lin_coord(dx, scale_x, sw, sx, fx);
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    lin_coord(dy, scale_y, sh, sy, fy);
    const int sy0 = clampi(sy, 0, sh - 1), sy1 = clampi(sy + 1, 0, sh - 1);
    const int sx1 = sx + 1 < sw ? sx + 1 : sx;
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const int r0i = sy0 * sw, r1i = sy1 * sw;
    float r0 = ldf(sU, r0i + sx) * a0 + ldf(sU, r0i + sx1) * a1, r1 = ldf(sU, r1i + sx) * a0 + ldf(sU, r1i + sx1) * a1;
    stf(dU, q, (r0 * b0 + r1 * b1) * mul);
    r0 = ldf(sV, r0i + sx) * a0 + ldf(sV, r0i + sx1) * a1; r1 = ldf(sV, r1i + sx) * a0 + ldf(sV, r1i + sx1) * a1;
    stf(dV, q, (r0 * b0 + r1 * b1) * mul);
}

// one sample of the final resize (cv::resize INTER_LINEAR of the finest flow, x mul) at working-size pixel (dx, dy);
// sU / sV: the pair's finest-level planes
__device__ __forceinline__ void final_sample_px(const float* __restrict__ sU, const float* __restrict__ sV, float* __restrict__ o2, int dx, int dy,
                                                int sh, int sw, double scale_x, double scale_y, float mul)
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
    const int r0i = sy0 * sw, r1i = sy1 * sw;
    float r0 = ldf(sU, r0i + sx) * a0 + ldf(sU, r0i + sx1) * a1, r1 = ldf(sU, r1i + sx) * a0 + ldf(sU, r1i + sx1) * a1;
    o2[0] = (r0 * b0 + r1 * b1) * mul;
    r0 = ldf(sV, r0i + sx) * a0 + ldf(sV, r0i + sx1) * a1; r1 = ldf(sV, r1i + sx) * a0 + ldf(sV, r1i + sx1) * a1;
    o2[1] = (r0 * b0 + r1 * b1) * mul;
}

#ifdef VSTAB_FUSED_TRACE   // developer build: per-phase time of one workgroup (tools/fused_phases.py)
static long long* g_dis_dbg = nullptr;
extern "C" void vstab_dis_dbg(long long* p) { g_dis_dbg = p; }
#define FUSED_MARK(i) do { if (prof_) { const long long now_ = wall_clock64(); a.dbg[i] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define FUSED_MARK(i)
#endif
// f32 division as hipcc emits it under -fhip-fp32-correctly-rounded-divide-sqrt, split so that quotients with a common
// denominator share its refined reciprocal: the compiler's sequence is
//   y0 = v_rcp(b); y = fma(fma(-b, y0, 1), y0, y0);  q0 = a*y; q1 = fma(fma(-b, q0, a), y, q0); q = fma(fma(-b, q1, a), y, q1)
// wrapped in v_div_scale (rescales operands only when the denominator is denormal / huge or the numerator's exponent is
// below 2^-104) and v_div_fixup (NaN / inf / zero cases).  For the linear system of the variational refinement the
// denominators are >= zeta^2 = 0.01 and <= ~1e6 and the numerators are 0 or >= ~1e-20, where both wrappers are the
// identity, so `div_shared(a, b, rcp_refined(b))` produces the bits of `a / b` (the oracle's IEEE division) with 5
// instead of 11 instructions per additional numerator -- 18 of the 20 divisions per pixel share three denominators.
__device__ __forceinline__ float rcp_refined(float b)
{
    const float y0 = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, y0, 1.0f), y0, y0);
}
__device__ __forceinline__ float div_shared(float a, float b, float y);
// a / b for a single numerator in the same domain (8 instead of 11 instructions: no v_div_scale / v_div_fixup)
__device__ __forceinline__ float div_plain(float a, float b);
__device__ __forceinline__ float div_shared(float a, float b, float y)
{
    const float q0 = a * y;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-b, q0, a), y, q0);
    return __builtin_fmaf(__builtin_fmaf(-b, q1, a), y, q1);
}

__device__ __forceinline__ float div_plain(float a, float b) { return div_shared(a, b, rcp_refined(b)); }

// sqrtf as hipcc emits it under -fhip-fp32-correctly-rounded-divide-sqrt, minus the two wrappers that are the identity
// for the arguments of the variational refinement (>= eps^2 = 1e-6, finite): the 2^32 pre-scaling of arguments below
// 2^-96 and the pass-through of +-0 / +inf.  What is left is the compiler's own correction of v_sqrt_f32's result by
// one unit in the last place in either direction (s-, s+ tested with exact fma residuals): the correctly rounded root,
// 8 instead of 15 instructions and two instead of five VCC hazards.  NaN and +inf come out as with the full sequence.
__device__ __forceinline__ float sqrt_plain(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float em = __builtin_fmaf(-sm, s, x), ep = __builtin_fmaf(-sp, s, x);
    s = (0.0f >= em) ? sm : s;
    s = (0.0f < ep) ? sp : s;
    return s;
}

struct LevelArgs {
#ifdef VSTAB_FUSED_TRACE
    long long* dbg;
#endif
    const uint8_t* I;   // [n][h][w] pyramid level
    const float* Sx;    // [P][hs][ws]
    const float* Sy;
    float* U;           // [P][h][w]
    float* V;
    float* nextU;       // [P][nh][nw] (finer level) or nullptr
    float* nextV;
    VrBufs vb;
    int P, h, w, ws, hs, nh, nw, nhs, nws;   // n*: the next finer level (image size, patch grid)
    // finest level, fused form: the stride-`g_step` grid the fit reads (final_sample_kernel's output) is formed by the pair's
    // own workgroup behind the merge -- one launch and two kernel boundaries less on the chain
    float* grid_out;        // [P][g_h][g_w][2] or nullptr
    int g_h, g_w, g_step;
    double g_sx, g_sy;
    float g_mul;
    int tiles_x, tiles_y;   // SOR tiling (LDS temporal blocking)
    int lds_plane;          // floats per LDS plane (max padded tile)
    int parts;              // split launches: workgroups per pair of this launch (pixel slices, or tiles)
    int it;                 // split launches: fixed-point iteration of a LEVEL_TILE launch
    double up_sx, up_sy;
    float alpha2, delta2, gamma2, zeta2, eps2, omega;
};

constexpr int SOR_HALO = 2 * SOR_ITERS;   // one pixel of dependency per half-sweep
// Tuned on MI355X (profiles/r01_fused_tile_sweep.md): with 16 waves per CU a wave has 128 VGPRs.  With the tile in
// three separate LDS planes 4 or 5 owned pixels per colour spilled (nine loop-invariant LDS addresses per pixel
// next to the coefficients) and 3 was fastest; with the interleaved float4 tile (three addresses per pixel) 5 fit
// without spilling, and the larger tiles (less halo work) win: 5 is also the LDS limit (160 KB / 16 B per pixel).
#ifndef VSTAB_FUSED_T
#define VSTAB_FUSED_T 1024
#define VSTAB_SOR_NPT 5
#endif
#ifdef VSTAB_LEVEL_WAVES4   // tile sweeps with fewer threads per workgroup: keep 128 VGPRs so that two workgroups share a CU
#define VSTAB_LEVEL_ATTR __attribute__((amdgpu_waves_per_eu(4)))
#else
#define VSTAB_LEVEL_ATTR
#endif
constexpr int FUSED_T = VSTAB_FUSED_T;    // threads of the fused level kernel
constexpr int SOR_NPT = VSTAB_SOR_NPT;    // owned pixels per thread and colour (tile <= 2 * SOR_NPT * FUSED_T px)

// One workgroup per frame pair runs densification, the whole variational refinement and the x2 upsample
// to the next finer level in ONE launch.  The red-black SOR sweeps of a fixed-point iteration are
// temporally blocked: per tile, the smoothness weights and the increment (dU,dV) sit in LDS with a
// 10-pixel halo (one pixel of dependency per half-sweep), the five linear-system coefficients of each
// owned pixel sit in registers and never reach memory, all 10 half-sweeps run on-chip, and only the tile
// interior is written back (double-buffered, because neighbouring tiles still need the old halo).
// tempW of OpenCV (W + dW) is recomputed where needed instead of stored.
//
// MODE: LEVEL_FUSED is that single launch -- one workgroup per pair keeps 255 pairs on 256 CUs busy for a whole
// 256-frame clip.  A rank of a multi-GPU run (or a short clip) has far fewer pairs than the chip has CUs, and the
// fused form would leave most of it idle for the same wall time; the SPLIT modes run the same phases as separate
// launches whose grids also cover the work INSIDE a pair -- pixel slices for the per-pixel phases, the tiles of one
// fixed-point iteration for the temporally blocked part (tiles of an iteration are independent: they read the
// previous iteration's increment and write the other buffer) -- with the kernel boundary as the only
// synchronisation.  Same code, same arithmetic, same bits (tests/test_dis_gpu.py runs both).
enum { LEVEL_FUSED = 0, LEVEL_PRE = 1, LEVEL_DERIV = 2, LEVEL_TILE = 4, LEVEL_MERGE = 5, LEVEL_UPSAMPLE = 6 };

template <int MODE>
__global__ __launch_bounds__(FUSED_T) VSTAB_LEVEL_ATTR void level_kernel(LevelArgs a)
{
#ifdef VSTAB_FUSED_TRACE
    long long tprev_ = wall_clock64();
    const bool prof_ = MODE == LEVEL_FUSED && a.dbg && blockIdx.x == 7 && threadIdx.x == 0;
#endif

    extern __shared__ float vr_lds[];
    constexpr bool FUSED = MODE == LEVEL_FUSED;
    const int pair = FUSED ? (int)blockIdx.x : (int)blockIdx.x / a.parts;
    const int part = FUSED ? 0 : (int)blockIdx.x % a.parts;      // pixel slice or tile of this workgroup
    const int nparts = FUSED ? 1 : a.parts;
    const int h = a.h, w = a.w;
    const int npx = h * w;
    const long long base = (long long)pair * npx;
    const uint8_t* I0 = a.I + (size_t)pair * npx;
    const uint8_t* I1 = a.I + (size_t)(pair + 1) * npx;
    const float* sx = a.Sx + (size_t)pair * a.hs * a.ws;
    const float* sy = a.Sy + (size_t)pair * a.hs * a.ws;
    // every plane shifted to this pair once (uniform: scalar registers); all accesses below are 32-bit in-plane offsets
    VrBufs b = a.vb;
    b.avg += base; b.Iz += base; b.D0 += base; b.D1 += base; b.UV += base; b.dA += base; b.dB += base;
    float* __restrict__ Uw = a.U + base;   // the pair's flow at this level (densify writes it, the merge updates it)
    float* __restrict__ Vw = a.V + base;
    const float* __restrict__ U = Uw;
    const float* __restrict__ V = Vw;
    // the pixels of this workgroup: all of the pair's (fused), or every nparts-th block of FUSED_T (split)
    // (x, y) of a thread's pixels advance incrementally: one integer division per phase instead of one per pixel (a
    // runtime divisor costs ~25 instructions; the per-pixel phases have 50-100)
    const int px_stride_ = nparts * (int)blockDim.x;
    const int px_dy_ = px_stride_ / w, px_dx_ = px_stride_ - px_dy_ * w;
#define FOR_PX(...)                                                                                   \
    {                                                                                                 \
        int q_ = part * (int)blockDim.x + (int)threadIdx.x;                                           \
        int y = q_ / w, x = q_ - y * w;                                                               \
        for (; q_ < npx; q_ += px_stride_) {                                                          \
            const int t = q_;                                                                         \
            __VA_ARGS__;                                                                              \
            x += px_dx_; y += px_dy_;                                                                 \
            if (x >= w) { x -= w; y++; }                                                              \
        }                                                                                             \
    }                                                                                                 \
    if (FUSED) __syncthreads();   /* (last line of FOR_PX: every per-pixel phase of the fused form ends in a barrier) */
    if (FUSED || MODE == LEVEL_PRE) {
        // densification and the warp of the variational refinement in one pass: the warp of a pixel needs the flow of that
        // pixel only, which the thread has just formed (no second loop over the pixels, no re-read of U / V)
        FOR_PX(float u_, v_; densify_px(I0, I1, sx, sy, Uw, Vw, t, y, x, h, w, a.ws, a.hs, u_, v_);
               vr_warp_px(I0, I1, u_, v_, b, t, x, y, h, w))   // also zeroes the increment (buffer A)
        FUSED_MARK(0);
        FUSED_MARK(1);
    }
    if (FUSED || MODE == LEVEL_DERIV) { FOR_PX(vr_deriv_px(b, t, x, y, h, w)) }
    FUSED_MARK(2);

    const f4_t* __restrict__ pD0 = b.D0;
    const f4_t* __restrict__ pD1 = b.D1;
    const f2_t* __restrict__ pUV = b.UV;
    // LDS tile: one float4 (dU, dV, smoothness weight, pad) per padded pixel -- an update reads its own and its
    // left / up neighbours' triples with one 16-B load each and the right / down increments with one 8-B load each
    // (6 LDS instructions instead of 15), and only four loop-invariant addresses per owned pixel are live in the
    // sweep loop instead of nine (three planes x own / up / down), which is what lets a thread own more pixels
    // without spilling.
    //
    // Layout: COLUMN-PARITY SEPARATED.  A padded row py is stored as two arrays of hw = ceil(pw / 2) entries, the even
    // padded columns first, then the odd ones: pixel (py, px) sits at LIDX = (2 py + (px & 1)) hw + (px >> 1).  The
    // pixels a half-sweep updates in one row all have the same column parity (a checkerboard colour), and consecutive
    // lanes own consecutive ones, so every access of the update -- own, left / right (the other array of the row), up /
    // down (the same array two row-slots away) -- runs at a 16-B lane stride: ds_read_b128 conflict-free, ds_read_b64 /
    // ds_write_b64 2-way.  With the pixels in raster order (round 2) the stride was 32 B: 2-way on every b128, 4-way on
    // the b64 reads and on the store -- 2.9 k of a half-sweep's 3.5 k cycles LDS-busy (profiles/r03_sor_lds_layout.md).
    // Same values, same operations, same order: same bits.
    f4_t* lP = reinterpret_cast<f4_t*>(vr_lds);
    // Plane reads of the tile stages: a uniform plane base (SGPR pair) + a 32-bit BYTE offset per lane selects the
    // `global_load_dword v, v_off, s[base]` form; indexing the float pointer instead makes the compiler widen the element
    // index to 64 bits first (one v_lshl_add_u64 per load: 36 of them in the linear-system stage).  A pair's plane has
    // fewer than 2^30 pixels (pyramid level of a <= 960-px working image; checked by the host).
#define LDF(P, idx) (*reinterpret_cast<const float*>(reinterpret_cast<const char*>(P) + (unsigned)(idx) * 4u))
    // increment ping-pong between the two float2 planes of the workspace
    f2_t* dIn = b.dA;
    f2_t* dOut = b.dB;

    // split: one (iteration, tile) per workgroup; the increment ping-pong is a function of the iteration's parity
    const int it_lo = FUSED ? 0 : a.it, it_hi = FUSED ? VAR_ITERS : a.it + 1;
    const int tile_lo = FUSED ? 0 : part, tile_hi = FUSED ? a.tiles_x * a.tiles_y : part + 1;
    if (!FUSED && (a.it & 1)) { f2_t* tmp = dIn; dIn = dOut; dOut = tmp; }
    // A level that is ONE tile (the coarse levels: the tile is the image, no halo) keeps its increment in LDS from one
    // fixed-point iteration to the next in the fused form: the sweeps leave exactly the values stage 4 would write and the
    // next stage 1 would read back, so neither runs through memory (the weights are rebuilt from the LDS values, the final
    // merge reads them there).  Same values, same operations.
    const bool single = FUSED && a.tiles_x * a.tiles_y == 1;
    if (FUSED || MODE == LEVEL_TILE)
    for (int it = it_lo; it < it_hi; it++) {
        for (int tile = tile_lo; tile < tile_hi; tile++) {
            const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
            const int ix0 = (int)((long long)w * tx / a.tiles_x), ix1 = (int)((long long)w * (tx + 1) / a.tiles_x);
            const int iy0 = (int)((long long)h * ty / a.tiles_y), iy1 = (int)((long long)h * (ty + 1) / a.tiles_y);
            const int ox = max(ix0 - SOR_HALO, 0), oy = max(iy0 - SOR_HALO, 0);
            const int lw = min(ix1 + SOR_HALO, w) - ox, lh = min(iy1 + SOR_HALO, h) - oy;
            const int pw = lw + 2;                    // padded row length (1-px zero border)
            const int pn = pw * (lh + 2);
            const int hw = (pw + 1) >> 1;             // entries per column-parity array of a padded row
            const int rs2 = 2 * hw;                   // LDS distance of vertical neighbours
#define LIDX(py_, px_) ((2 * (py_) + ((px_) & 1)) * hw + ((px_) >> 1))
            // ---- stage 1: zero border, load the increment, compute the smoothness weights into LDS
            // (Staging U + dU of the padded tile in LDS first -- two loads per pixel instead of six, the neighbours' sums
            // from LDS -- was measured in round 4 and is slower: 316 vs 265-288 us per finest level.  This stage streams
            // 16 B per pixel from HBM at about the CU's share of the bandwidth; its load count is not what it waits for.)
            const int s1_dy = (int)blockDim.x / pw, s1_dx = (int)blockDim.x - s1_dy * pw;
            int py = (int)threadIdx.x / pw, px = (int)threadIdx.x - py * pw;
            if (single && it > 0) {
                // the increment is already in the tile (ox = oy = 0, lw = w, lh = h): only the weights are formed anew
                for (int k = threadIdx.x; k < pn; k += blockDim.x) {
                    const int gx = px - 1, gy = py - 1;
                    if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
                        const int q = gy * w + gx;
                        const int li = LIDX(py, px);
                        const int lr = (gx + 1 < w) ? LIDX(py, px + 1) : li;
                        const int ld = (gy + 1 < h) ? LIDX(py + 1, px) : li;
                        const f2_t d0 = *reinterpret_cast<const f2_t*>(lP + li), dr = *reinterpret_cast<const f2_t*>(lP + lr),
                                   dd = *reinterpret_cast<const f2_t*>(lP + ld);
                        const int qr = (gx + 1 < w) ? q + 1 : q;
                        const int qd = (gy + 1 < h) ? q + w : q;
                        const f2_t f0 = ld2(pUV, q), fr = ld2(pUV, qr), fd = ld2(pUV, qd);
                        const float tu = f0.x + d0.x, tv = f0.y + d0.y;
                        const float tur = fr.x + dr.x, tvr = fr.y + dr.y;
                        const float tud = fd.x + dd.x, tvd = fd.y + dd.y;
                        const float ux = tur - tu, vx = tvr - tv, uy = tud - tu, vy = tvd - tv;
                        reinterpret_cast<float*>(lP + li)[2] = div_plain(a.alpha2, sqrt_plain(ux * ux + vx * vx + uy * uy + vy * vy + a.eps2));
                    }
                    px += s1_dx; py += s1_dy;
                    if (px >= pw) { px -= pw; py++; }
                }
            } else
            for (int k = threadIdx.x; k < pn; k += blockDim.x) {
                const int lx = px - 1, ly = py - 1;
                float wv = 0.f, du = 0.f, dv = 0.f;
                if (lx >= 0 && lx < lw && ly >= 0 && ly < lh) {
                    const int gx = ox + lx, gy = oy + ly;
                    const int q = gy * w + gx;
                    const int qr = (gx + 1 < w) ? q + 1 : q;
                    const int qd = (gy + 1 < h) ? q + w : q;
                    const f2_t d0 = ld2(dIn, q), dr = ld2(dIn, qr), dd = ld2(dIn, qd);
                    const f2_t f0 = ld2(pUV, q), fr = ld2(pUV, qr), fd = ld2(pUV, qd);
                    du = d0.x; dv = d0.y;
                    const float tu = f0.x + du, tv = f0.y + dv;
                    const float tur = fr.x + dr.x, tvr = fr.y + dr.y;
                    const float tud = fd.x + dd.x, tvd = fd.y + dd.y;
                    const float ux = tur - tu, vx = tvr - tv, uy = tud - tu, vy = tvd - tv;
                    wv = div_plain(a.alpha2, sqrt_plain(ux * ux + vx * vx + uy * uy + vy * vy + a.eps2));
                }
                lP[LIDX(py, px)] = f4_t{du, dv, wv, 0.f};
                px += s1_dx; py += s1_dy;
                if (px >= pw) { px -= pw; py++; }
            }
            __syncthreads();
            FUSED_MARK(3);
            // ---- stage 2: linear system of the owned pixels -> registers
            const int half_lw = (lw + 1) >> 1;
            const int ncol = lh * half_lw;
            float c11[2][SOR_NPT], c12[2][SOR_NPT], c22[2][SOR_NPT], cb1[2][SOR_NPT], cb2[2][SOR_NPT];
            float cy22[2][SOR_NPT];   // refined reciprocal of c22 (that of c11 rides in the LDS tile's spare float)
            int cidx[2][SOR_NPT], cidl[2][SOR_NPT];   // LDS index of the owned pixel and of its left neighbour (right = left + 1)
            // Owned pixel j = 2*u + colour of this thread: k = threadIdx.x + u * FUSED_T = ly * half_lw + xh, advanced
            // incrementally (one division per tile).  The loop is NOT unrolled: one pixel's ~60 temporaries are live at a
            // time next to the persistent coefficient registers (an unrolled version interleaves the bodies and spills);
            // the results are routed to their registers by compile-time-indexed selects.
            const int step_rows = FUSED_T / half_lw, step_rem = FUSED_T - step_rows * half_lw;
            int ly = (int)threadIdx.x / half_lw, xh = (int)threadIdx.x - ly * half_lw;
#pragma unroll
            for (int jj = 0; jj < 2 * SOR_NPT; jj++) cidx[jj & 1][jj >> 1] = -1;
#pragma unroll 1
            for (int j = 0; j < 2 * SOR_NPT; j++) {
                const int color = j & 1;
                const int k = threadIdx.x + (j >> 1) * FUSED_T;
                if (k < ncol) {
                        const int gy = oy + ly;
                        const int lx = 2 * xh + ((gy + ox + color) & 1);
                        if (lx < lw) {
                            const int gx = ox + lx;
                            const int q = gy * w + gx;
                            const int li = LIDX(ly + 1, lx + 1), ll = LIDX(ly + 1, lx);
                            const f4_t e0 = ld4(pD0, q), e1 = ld4(pD1, q);
                            const float Ix = e0.x, Iy = e0.y, Iz = e0.z, Ixx = e0.w, Ixy = e1.x, Iyy = e1.y, Ixz = e1.z, Iyz = e1.w;
                            const f4_t own = lP[li];
                            const float du = own.x, dv = own.y;
                            // The flow of the four neighbours (smoothness term, below) is requested HERE, together with the
                            // eight derivative planes, from clamped (always valid) positions: one wait for everything instead
                            // of one dependent load-and-wait per neighbour after the data term (round 2 had each pair of loads
                            // inside its `if (has_*)`: five serial memory round trips per pixel).
                            const bool has_r = gx + 1 < w, has_l = gx > 0, has_d = gy + 1 < h, has_u = gy > 0;
                            const int q_r = has_r ? q + 1 : q, q_l = has_l ? q - 1 : q, q_d = has_d ? q + w : q, q_u = has_u ? q - w : q;
                            const f2_t fq = ld2(pUV, q), f_r = ld2(pUV, q_r), f_l = ld2(pUV, q_l), f_d = ld2(pUV, q_d), f_u = ld2(pUV, q_u);
                            const float uq = fq.x, vq = fq.y;
                            const float u_r = f_r.x, v_r = f_r.y, u_l = f_l.x, v_l = f_l.y;
                            const float u_d = f_d.x, v_d = f_d.y, u_u = f_u.x, v_u = f_u.y;
                            const float wl = lP[ll].z, wu = lP[li - rs2].z;
                            float a11, a12, a22, B1, B2;
                            {
                                float derivNorm = Ix * Ix + Iy * Iy + a.zeta2;
                                float yn = rcp_refined(derivNorm);
#define DIVN(x) div_shared((x), derivNorm, yn)
                                float Ik1z = Iz + Ix * du + Iy * dv;
                                float weight = div_plain(a.delta2, sqrt_plain(DIVN(Ik1z * Ik1z) + a.eps2));
                                a11 = weight * DIVN(Ix * Ix) + a.zeta2;
                                a12 = weight * DIVN(Ix * Iy);
                                a22 = weight * DIVN(Iy * Iy) + a.zeta2;
                                B1 = -weight * DIVN(Iz * Ix);
                                B2 = -weight * DIVN(Iz * Iy);
                                derivNorm = Ixx * Ixx + Ixy * Ixy + a.zeta2;
                                yn = rcp_refined(derivNorm);
                                const float derivNorm2 = Iyy * Iyy + Ixy * Ixy + a.zeta2;
                                const float yn2 = rcp_refined(derivNorm2);
#define DIVN2(x) div_shared((x), derivNorm2, yn2)
                                float Ik1zx = Ixz + Ixx * du + Ixy * dv;
                                float Ik1zy = Iyz + Ixy * du + Iyy * dv;
                                weight = div_plain(a.gamma2, sqrt_plain(DIVN(Ik1zx * Ik1zx) + DIVN2(Ik1zy * Ik1zy) + a.eps2));
                                a11 += weight * (DIVN(Ixx * Ixx) + DIVN2(Ixy * Ixy));
                                a12 += weight * (DIVN(Ixx * Ixy) + DIVN2(Ixy * Iyy));
                                a22 += weight * (DIVN(Ixy * Ixy) + DIVN2(Iyy * Iyy));
                                B1 += -weight * (DIVN(Ixx * Ixz) + DIVN2(Ixy * Iyz));
                                B2 += -weight * (DIVN(Ixy * Ixz) + DIVN2(Iyy * Iyz));
#undef DIVN
#undef DIVN2
                            }
                            // smoothness term, accumulated in OpenCV's red/black scatter order.  `color` IS the global
                            // checkerboard parity (gx + gy) & 1 of the pixel, so the order is known at compile time.
                            const float wq = own.z;
#define SM_RIGHT() if (has_r) { B1 += wq * (u_r - uq); a11 += wq; B2 += wq * (v_r - vq); a22 += wq; }
#define SM_LEFT()  if (has_l) { B1 -= wl * (uq - u_l); a11 += wl; B2 -= wl * (vq - v_l); a22 += wl; }
#define SM_DOWN()  if (has_d) { B1 += wq * (u_d - uq); a11 += wq; B2 += wq * (v_d - vq); a22 += wq; }
#define SM_UP()    if (has_u) { B1 -= wu * (uq - u_u); a11 += wu; B2 -= wu * (vq - v_u); a22 += wu; }
                            if (color == 0) { SM_RIGHT() SM_LEFT() SM_DOWN() SM_UP() }
                            else            { SM_LEFT() SM_RIGHT() SM_UP() SM_DOWN() }
#undef SM_RIGHT
#undef SM_LEFT
#undef SM_DOWN
#undef SM_UP
                            // The two SOR denominators of a pixel are fixed for the ten half-sweeps: their refined reciprocals
                            // are formed once here (3 instructions each) instead of in each of the pixel's five updates.
                            const float y11 = rcp_refined(a11), y22 = rcp_refined(a22);
#pragma unroll
                            for (int jj = 0; jj < 2 * SOR_NPT; jj++)
                                if (jj == j) {
                                    cidx[jj & 1][jj >> 1] = li; cidl[jj & 1][jj >> 1] = ll;
                                    c11[jj & 1][jj >> 1] = a11; c12[jj & 1][jj >> 1] = a12; c22[jj & 1][jj >> 1] = a22;
                                    cb1[jj & 1][jj >> 1] = B1; cb2[jj & 1][jj >> 1] = B2;
                                    cy22[jj & 1][jj >> 1] = y22;
                                }
                            lP[li].w = y11;   // own pixel's spare slot: nobody else reads or writes this dword
                        }
                }
                if (color == 1) {
                    xh += step_rem; ly += step_rows;
                    if (xh >= half_lw) { xh -= half_lw; ly++; }
                }
            }
            // NOTE: a left/up neighbour outside the tile (but inside the image) reads the zero border here; that
            // only happens for halo pixels at the tile rim, whose values are never written back.
#ifdef VSTAB_FUSED_TRACE
            __syncthreads();
#endif
            FUSED_MARK(4);
            // ---- stage 3: 2*SOR_ITERS half-sweeps entirely in LDS
            for (int s = 0; s < SOR_ITERS * 2; s++) {
#pragma unroll
                for (int color = 0; color < 2; color++) {
                    if (color != (s & 1)) continue;
#pragma unroll
                    for (int u = 0; u < SOR_NPT; u++) {
                        const int li = cidx[color][u];
                        if (li >= 0) {
                            const int ll = cidl[color][u];
                            const f4_t pc = lP[li], pl = lP[ll], pu = lP[li - rs2];
                            const f2_t pr = *reinterpret_cast<const f2_t*>(lP + ll + 1), pd = *reinterpret_cast<const f2_t*>(lP + li + rs2);
                            const float wq = pc.z, wl = pl.z, wu = pu.z;
                            const float sigmaU = wl * pl.x + wq * pr.x + wu * pu.x + wq * pd.x;
                            const float sigmaV = wl * pl.y + wq * pr.y + wu * pu.y + wq * pd.y;
                            float du = pc.x, dv = pc.y;
                            // the two divisions in the split form of `/` (rcp_refined + div_shared, above): denominators >= zeta^2, so
                            // v_div_scale / v_div_fixup are the identity.  The refined reciprocals come from stage 2: the one of
                            // c11 in the tile's spare float (it arrives with `pc`), the one of c22 in a register (ten registers
                            // fit since the plane accesses stopped using 64-bit addresses; twenty did not: round-2 profile notes)
                            du += a.omega * (div_shared(sigmaU + cb1[color][u] - dv * c12[color][u], c11[color][u], pc.w) - du);
                            dv += a.omega * (div_shared(sigmaV + cb2[color][u] - du * c12[color][u], c22[color][u], cy22[color][u]) - dv);
                            *reinterpret_cast<f2_t*>(lP + li) = f2_t{du, dv};
                        }
                        __builtin_amdgcn_sched_barrier(0);   // keep the owned pixels' updates apart (register pressure)
                    }
                }
                __syncthreads();
            }
            FUSED_MARK(5);
            // ---- stage 4: write the tile interior of the new increment
            const int iw = ix1 - ix0, ih = iy1 - iy0;
            const int s4_dy = (int)blockDim.x / iw, s4_dx = (int)blockDim.x - s4_dy * iw;
            int yy = (int)threadIdx.x / iw, xx = (int)threadIdx.x - yy * iw;
            if (!single)
            for (int k = threadIdx.x; k < iw * ih; k += blockDim.x) {
                const int gx = ix0 + xx, gy = iy0 + yy;
                const int li = LIDX(gy - oy + 1, gx - ox + 1);
                st2(dOut, gy * w + gx, *reinterpret_cast<const f2_t*>(lP + li));
                xx += s4_dx; yy += s4_dy;
                if (xx >= iw) { xx -= iw; yy++; }
            }
            __syncthreads();
            FUSED_MARK(6);
#undef LIDX
        }
        f2_t* tmp = dIn; dIn = dOut; dOut = tmp;
    }
    // mergeCheckerboard(W, tempW): W + dW of the last fixed-point iteration
    if (MODE == LEVEL_MERGE && (VAR_ITERS & 1)) dIn = b.dB;   // where iteration VAR_ITERS-1 wrote
    if (single) {
        const int hw = (w + 3) >> 1;              // the one tile's column-parity layout: padded row = w + 2 entries
        for (int q_ = (int)threadIdx.x; q_ < npx; q_ += (int)blockDim.x) {
            const int y = q_ / w, x = q_ - y * w;
            const f2_t f = ld2(pUV, q_);
            const f2_t d = *reinterpret_cast<const f2_t*>(lP + ((2 * (y + 1) + ((x + 1) & 1)) * hw + ((x + 1) >> 1)));
            stf(Uw, q_, f.x + d.x);
            stf(Vw, q_, f.y + d.y);
        }
        __syncthreads();
    } else if (FUSED || MODE == LEVEL_MERGE) {
        for (int q_ = part * (int)blockDim.x + (int)threadIdx.x; q_ < npx; q_ += nparts * (int)blockDim.x) {
            const f2_t f = ld2(pUV, q_), d = ld2(dIn, q_);
            stf(Uw, q_, f.x + d.x);
            stf(Vw, q_, f.y + d.y);
        }
        if (FUSED) __syncthreads();
    }
#undef FOR_PX
#undef LDF
    FUSED_MARK(7);
    if ((FUSED || MODE == LEVEL_UPSAMPLE) && a.nextU != nullptr) {
        // The x2 upsampled flow is read at the finer level's PATCH CENTRES only -- the initial flow of its patch search,
        // pis4_kernel's `U[(i + PSZ / 2) * w + j + PSZ / 2]` -- before that level's densification overwrites the whole
        // plane: only those nhs x nws samples are evaluated (1 pixel in 16), each exactly as the dense resize forms it.
        const int nn = a.nh * a.nw;
        float* __restrict__ nU = a.nextU + (long long)pair * nn;
        float* __restrict__ nV = a.nextV + (long long)pair * nn;
        const int npts = a.nhs * a.nws;
        for (int q_ = part * (int)blockDim.x + (int)threadIdx.x; q_ < npts; q_ += px_stride_) {
            const int is = q_ / a.nws, js = q_ - is * a.nws;
            const int dy = is * PSTR + PSZ / 2, dx = js * PSTR + PSZ / 2;
            upsample_px(U, V, nU, nV, dy * a.nw + dx, dx, dy, h, w, a.up_sx, a.up_sy, 2.0f);
        }
    }
    if (FUSED && a.grid_out != nullptr) {
        // (the merge above ends in a barrier: the pair's whole U / V plane is written)
        float* __restrict__ go = a.grid_out + (size_t)pair * a.g_h * a.g_w * 2;
        for (int t = (int)threadIdx.x; t < a.g_h * a.g_w; t += (int)blockDim.x) {
            const int gy = t / a.g_w, gx = t - gy * a.g_w;
            final_sample_px(U, V, go + 2 * t, gx * a.g_step, gy * a.g_step, h, w, a.g_sx, a.g_sy, a.g_mul);
        }
    }
#ifdef VSTAB_FUSED_TRACE
    __syncthreads();
    FUSED_MARK(8);
#endif
}

// final resize of the finest flow to the working size (x 2^finest), evaluated on a strided grid
// (step 1 = the full field).  out is [P][gh][gw][2].
__global__ __launch_bounds__(256) void final_sample_kernel(const float* __restrict__ sU, const float* __restrict__ sV, float* __restrict__ out,
                                                           int P, int sh, int sw, int gh, int gw, int step, double scale_x, double scale_y,
                                                           float mul)
{
    const long long p = blockIdx.y;
    out += (size_t)p * gh * gw * 2;
    FRAME_STRIDE(t, gh * gw) {
        const int gy = (int)(t / (unsigned)gw), gx = (int)(t - (unsigned)gy * (unsigned)gw);
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

// cv::resize(INTER_AREA) picks its path from the ratios: exact 2 x 2, exact integer boxes, or the general tap tables
struct AreaPlan { double scale_x, scale_y; int isx, isy, mode; };
int plan_area(int sh, int sw, int dh, int dw, AreaPlan& pl)
{
    pl.scale_x = 1. / ((double)dw / sw); pl.scale_y = 1. / ((double)dh / sh);
    pl.isx = (int)std::lrint(pl.scale_x); pl.isy = (int)std::lrint(pl.scale_y);
    const bool fast = std::fabs(pl.scale_x - pl.isx) < DBL_EPSILON && std::fabs(pl.scale_y - pl.isy) < DBL_EPSILON;
    VSTAB_REQUIRE(pl.scale_x >= 1.0 && pl.scale_y >= 1.0 && pl.scale_x < 6.0 && pl.scale_y < 6.0, "dis: area ratio %.3fx%.3f unsupported", pl.scale_x, pl.scale_y);
    pl.mode = fast ? ((pl.isx == 2 && pl.isy == 2) ? 0 : 1) : 2;
    return 0;
}

int launch_area(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int sh, int sw, int dh, int dw)
{
    AreaPlan pl;
    if (int rc = plan_area(sh, sw, dh, dw, pl)) return rc;
    const double scale_x = pl.scale_x, scale_y = pl.scale_y;
    const int isx = pl.isx, isy = pl.isy;
    int mode = pl.mode;
    if (mode == 1 && isx == 4 && isy == 4 && sw % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0) mode = 3;
    const long long row_groups = (long long)n * ((dh + AREA_ROWS - 1) / AREA_ROWS);
    if (mode == 2 && dw <= AREA_MAX_COLS && row_groups < 0x7fffffffLL)
        hipLaunchKernelGGL(area_general_rows_kernel, dim3((unsigned)row_groups), dim3(256), 0, st, src, dst, n, sh, sw, dh, dw, scale_x, scale_y);
    else
        hipLaunchKernelGGL(area_u8_kernel, frame_grid((long long)dh * dw, n), dim3(256), 0, st, src, dst, n, sh, sw, dh, dw, mode, isx, isy, scale_x, scale_y);
    VSTAB_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// DISOpticalFlowImpl::calc scale selection incl. autoSelectPatchSizeAndScales (finest 2 -> default branch)
static bool dis_scales(int h, int w, int& finest, int& coarsest)
{
    coarsest = coarsest_scale(h, w);
    if (coarsest < 0) return false;
    if (coarsest < finest) {
        const int c = (int)std::floor(std::log2((2.0f * (float)w) / (5.0f * (float)PSZ)));
        coarsest = c > 0 ? c : 0;
        finest = coarsest - 2 > 0 ? coarsest - 2 : 0;
    }
    return coarsest < MAX_LEVELS;
}

static int dis_run(vstab_ctx* ctx, const uint8_t* gray, int n, int h, int w, int FINEST, int coarsest, float* flow, float* grid_flow,
                   int sample_step)
{
    hipStream_t st = ctx->stream;
    const int P = n - 1;
    VSTAB_REQUIRE(n <= 65535, "vstab_dis_flow_batch: %d frames in one call (the frame index is a grid dimension: at most 65535)", n);
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
    float *aux = nullptr, *aux_c = nullptr, *Sx = nullptr, *Sy = nullptr;
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
        aux_c = c.take<float>(5 * (size_t)n * G[coarsest].h * G[coarsest].ws);   // the coarsest level's own (it is prepared on the main stream)
        Sx = c.take<float>((size_t)P * F.hs * F.ws);
        Sy = c.take<float>((size_t)P * F.hs * F.ws);
        vb.avg = c.take<float>(npF); vb.Iz = c.take<float>(npF);
        vb.D0 = c.take<f4_t>(npF); vb.D1 = c.take<f4_t>(npF);
        vb.UV = c.take<f2_t>(npF); vb.dA = c.take<f2_t>(npF); vb.dB = c.take<f2_t>(npF);
    };
    {
        Carver sizer(nullptr);
        layout(sizer);
        if (ctx->d_dis.reserve(sizer.off + 256)) return 1;
    }
    Carver carver(ctx->d_dis.ptr);
    layout(carver);

    // ---- per-frame preparation: pyramid, padded copies, gradients, structure tensor ----
    // The pyramid is a chain (each level is the INTER_AREA of the previous one) and stays on the call's stream.  The
    // rest of a level's preparation is only needed when the coarse-to-fine chain reaches that level, and the chain's
    // coarse levels are latency-bound launches that leave most of the chip idle: it runs on a second stream, coarsest
    // level first, one event per level (VSTAB_DIS_PREP_STREAM=0 keeps everything on one stream: A/B measurement).
    static const bool two_streams = [] { const char* e = getenv("VSTAB_DIS_PREP_STREAM"); return !(e && atoi(e) == 0); }();
    static_assert(sizeof(ctx->ev_prep) / sizeof(ctx->ev_prep[0]) >= MAX_LEVELS, "one event per pyramid level");
    hipStream_t ps = st;
    if (two_streams) {
        if (!ctx->prep_stream) {
            VSTAB_HIP(hipStreamCreateWithFlags(&ctx->prep_stream, hipStreamNonBlocking));
            VSTAB_HIP(hipEventCreateWithFlags(&ctx->ev_pyramid, hipEventDisableTiming));
            for (auto& ev : ctx->ev_prep) VSTAB_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
        ps = ctx->prep_stream;
    }
    // (set_timing level 2 only: events around the stages of this call, vstab_internal.h)
    auto prep_timer = std::make_unique<DetailTimer>(ctx, "dis_prep");
    bool coarsest_prepared = false;   // pyramid_tail_kernel also formed the coarsest level's padded copy, gradients and tensor
    {
        if (launch_area(st, gray, I[FINEST], n, h, w, G[FINEST].h, G[FINEST].w)) return 1;
        // the levels above the finest: one launch (pyramid_tail_kernel) where the finest level and its successor fit one
        // workgroup's LDS -- every clip whose working size is DIS's usual <= 960 px -- else one launch per level
        // (VSTAB_DIS_PYRAMID_TAIL=0 forces the latter: A/B measurement, tests)
        static const bool tail_on = [] { const char* e = getenv("VSTAB_DIS_PYRAMID_TAIL"); return !(e && atoi(e) == 0); }();
        const int tail_levels = coarsest - FINEST;
        size_t tail_lds = 0;
        if (tail_levels >= 1) {
            const LevelGeom& a0 = G[FINEST];
            const LevelGeom& a1 = G[FINEST + 1];
            tail_lds = (((size_t)a0.h * a0.w + 15) & ~size_t(15)) + (((size_t)a1.h * a1.w + 15) & ~size_t(15)) +
                       sizeof(AreaTaps) * ((size_t)a1.w + a1.h);
        }
        if (tail_on && tail_levels >= 1 && tail_levels <= TAIL_MAX_LEVELS && tail_lds <= 150 * 1024) {
            TailArgs ta{};
#ifdef VSTAB_FUSED_TRACE
            ta.dbg = g_dis_dbg ? g_dis_dbg + 16 * 7 : nullptr;
#endif
            {   // the coarsest level's preparation rides along where its scratch fits too
                const LevelGeom& c = G[coarsest];
                const size_t off = (tail_lds + 15) & ~size_t(15);
                const size_t need = (((size_t)4 * c.h * c.w + 15) & ~size_t(15)) + sizeof(float) * 5 * (size_t)c.h * c.ws;
                if (off + need <= 150 * 1024) {
                    ta.prep_ext = Iext[coarsest]; ta.prep_ix = Ixs[coarsest]; ta.prep_iy = Iys[coarsest]; ta.prep_tensor = tensor[coarsest];
                    ta.prep_ws = c.ws; ta.prep_hs = c.hs; ta.prep_lds_off = (int)off;
                    tail_lds = off + need;
                    coarsest_prepared = true;
                }
            }
            ta.src = I[FINEST]; ta.n = n; ta.h0 = G[FINEST].h; ta.w0 = G[FINEST].w; ta.levels = tail_levels;
            for (int i = FINEST + 1; i <= coarsest; i++) {
                AreaPlan pl;
                if (int rc = plan_area(G[i - 1].h, G[i - 1].w, G[i].h, G[i].w, pl)) return rc;
                TailLevel& tl = ta.lv[i - FINEST - 1];
                tl.dst = I[i]; tl.h = G[i].h; tl.w = G[i].w; tl.mode = pl.mode; tl.kx = pl.isx; tl.ky = pl.isy;
                if (pl.mode == 2) {   // general ratio: kx / ky = the most taps an output can have along the axis (floor(scale) + 2)
                    tl.kx = (int)std::floor(pl.scale_x) + 2; tl.ky = (int)std::floor(pl.scale_y) + 2;
                }
                tl.scale_x = pl.scale_x; tl.scale_y = pl.scale_y;
            }
            if (tail_lds > 64 * 1024)
                VSTAB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pyramid_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tail_lds));
            hipLaunchKernelGGL(pyramid_tail_kernel, dim3((unsigned)n), dim3(1024), tail_lds, st, ta);
            VSTAB_HIP(hipGetLastError());
        } else {
            for (int i = FINEST + 1; i <= coarsest; i++)
                if (launch_area(st, I[i - 1], I[i], n, G[i - 1].h, G[i - 1].w, G[i].h, G[i].w)) return 1;
        }
    }
    if (two_streams) {
        // (the event also orders the second stream behind every earlier kernel of this stream that still reads the workspace)
        VSTAB_HIP(hipEventRecord(ctx->ev_pyramid, st));
        VSTAB_HIP(hipStreamWaitEvent(ps, ctx->ev_pyramid, 0));
    }
    for (int i = coarsest; i >= FINEST; i--) {
        const LevelGeom& g = G[i];
        // the coarsest level is needed at once: it is prepared on the main stream, behind the pyramid, so that the chain's
        // first patch search does not wait for a cross-stream event (its own scratch: the other levels' `aux` is in use on
        // the second stream at the same time)
        const bool own = i == coarsest;
        if (own && coarsest_prepared) continue;
        hipStream_t s_i = own ? st : ps;
        float* aux_i = own ? aux_c : aux;
        hipLaunchKernelGGL(pad_replicate_kernel, frame_grid((long long)(g.h + 32) * (g.w + 32), n), dim3(256), 0, s_i, I[i], Iext[i], n, g.h, g.w);
        hipLaunchKernelGGL(sobel_kernel, frame_grid((long long)g.h * g.w, n), dim3(256), 0, s_i, I[i], Ixs[i], Iys[i], n, g.h, g.w);
        hipLaunchKernelGGL(tensor_h_kernel, frame_grid((long long)g.h * g.ws, n), dim3(256), 0, s_i, Ixs[i], Iys[i], aux_i, n, g.h, g.w, g.ws);
        hipLaunchKernelGGL(tensor_v_kernel, dim3((unsigned)((g.ws + 63) / 64), (unsigned)n, 5u), dim3(64), 0, s_i, aux_i, tensor[i], n, g.h, g.ws, g.hs);
        VSTAB_HIP(hipGetLastError());
        if (two_streams && !own) VSTAB_HIP(hipEventRecord(ctx->ev_prep[i], ps));
    }
    VSTAB_HIP(hipMemsetAsync(Ul[coarsest], 0, sizeof(float) * (size_t)P * G[coarsest].h * G[coarsest].w, st));
    VSTAB_HIP(hipMemsetAsync(Vl[coarsest], 0, sizeof(float) * (size_t)P * G[coarsest].h * G[coarsest].w, st));

    prep_timer.reset();

    bool grid_fused = false;   // the finest level's fused launch formed the sampled grid itself
    const float zeta = 0.1f, epsilon = 0.001f, alpha = 20.0f, delta = 5.0f, gamma = 10.0f, omega = 1.6f;
    const float zeta2 = zeta * zeta, eps2 = epsilon * epsilon, gamma2 = gamma / 2, delta2 = delta / 2, alpha2 = alpha / 4;

    for (int i = coarsest; i >= FINEST; i--) {
        const LevelGeom& g = G[i];
        PisArgs pa{};
        pa.I = I[i]; pa.Iext = Iext[i]; pa.Ix = Ixs[i]; pa.Iy = Iys[i]; pa.tensor = tensor[i];
        pa.U = Ul[i]; pa.V = Vl[i]; pa.Sx = Sx; pa.Sy = Sy;
        pa.n = n; pa.w = g.w; pa.h = g.h; pa.ws = g.ws; pa.hs = g.hs;
        pa.stripe_sz = (int)std::ceil(g.hs / 8.0);
        pa.spin_limit = 1 << 22;
#ifdef VSTAB_TEST_HOOKS   // fault injector of the test build (lib/libvstab_hooks.so): 0 forces the timeout report
        if (const char* e = getenv("VSTAB_DEBUG_PIS_SPIN_LIMIT")) pa.spin_limit = atoi(e);
#endif
        pa.status = ctx->d_status;
        if (two_streams && i != coarsest) VSTAB_HIP(hipStreamWaitEvent(st, ctx->ev_prep[i], 0));   // this level's padded image, gradients, tensor
        const size_t lds_bytes = (((size_t)(g.w + 32) * (g.h + 32) + 15) & ~size_t(15)) + sizeof(float) * 2 * (size_t)g.hs * g.ws + sizeof(int) * 2 * (size_t)g.hs;
        VSTAB_REQUIRE(lds_bytes <= 160 * 1024, "vstab_dis_flow_batch: level %dx%d needs %zu B of LDS (> 160 KB)", g.w, g.h, lds_bytes);
        char kind_pis[24], kind_level[24];
        snprintf(kind_pis, sizeof(kind_pis), "dis_pis4_L%d", i);
        snprintf(kind_level, sizeof(kind_level), "dis_level_L%d", i);
        auto stage_timer = std::make_unique<DetailTimer>(ctx, kind_pis);
        // one wavefront per stripe walks its rows in groups of four; two share the groups where a stripe has more rows
        if (pa.stripe_sz > 4) {
            if (lds_bytes > 64 * 1024)
                VSTAB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pis4_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            hipLaunchKernelGGL(pis4_kernel<2>, dim3((unsigned)P * 2), dim3(64 * 2 * PIS_STRIPES_PER_BLOCK), lds_bytes, st, pa);
        } else {
            if (lds_bytes > 64 * 1024)
                VSTAB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pis4_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            hipLaunchKernelGGL(pis4_kernel<1>, dim3((unsigned)P * 2), dim3(64 * PIS_STRIPES_PER_BLOCK), lds_bytes, st, pa);
        }
        stage_timer.reset();
        stage_timer = std::make_unique<DetailTimer>(ctx, kind_level);
        LevelArgs la{};
#ifdef VSTAB_FUSED_TRACE
        la.dbg = g_dis_dbg ? g_dis_dbg + 16 * i : nullptr;
#endif
        la.I = I[i]; la.Sx = Sx; la.Sy = Sy; la.U = Ul[i]; la.V = Vl[i]; la.vb = vb;
        la.P = P; la.h = g.h; la.w = g.w; la.ws = g.ws; la.hs = g.hs;
        la.alpha2 = alpha2; la.delta2 = delta2; la.gamma2 = gamma2; la.zeta2 = zeta2; la.eps2 = eps2; la.omega = omega;
        if (i > FINEST) {
            const LevelGeom& d = G[i - 1];
            la.nextU = Ul[i - 1]; la.nextV = Vl[i - 1]; la.nh = d.h; la.nw = d.w; la.nhs = d.hs; la.nws = d.ws;
            la.up_sx = 1. / ((double)d.w / g.w); la.up_sy = 1. / ((double)d.h / g.h);
        }
        {   // smallest tiling whose padded tile fits SOR_TILE_CAP pixels (registers hold SOR_NPT pixels per colour)
            int tx = 1, ty = 1;
            auto tile_px = [&](int tx_, int ty_, int& padded) {
                const int lw = std::min((g.w + tx_ - 1) / tx_ + 2 * SOR_HALO, g.w), lh = std::min((g.h + ty_ - 1) / ty_ + 2 * SOR_HALO, g.h);
                padded = 2 * ((lw + 3) / 2) * (lh + 2);   // two column-parity arrays of ceil((lw + 2) / 2) entries per padded row
                return ((lw + 1) / 2) * lh;   // pixels of one colour
            };
            int padded = 0;
            while (tile_px(tx, ty, padded) > SOR_NPT * FUSED_T || (size_t)padded * 16 > 160 * 1024) {   // registers and LDS
                if ((g.w + tx - 1) / tx >= (g.h + ty - 1) / ty) tx++; else ty++;
                VSTAB_REQUIRE(tx <= 64 && ty <= 64, "vstab_dis_flow_batch: cannot tile a %dx%d level", g.w, g.h);
            }
            la.tiles_x = tx; la.tiles_y = ty; la.lds_plane = (padded + 3) & ~3;
        }
        const size_t vr_lds_bytes = sizeof(float) * 4 * (size_t)la.lds_plane;
        VSTAB_REQUIRE(vr_lds_bytes <= 160 * 1024, "vstab_dis_flow_batch: SOR tile needs %zu B of LDS", vr_lds_bytes);
        // One workgroup per pair fills the chip only when there are about as many pairs as CUs (a 256-frame clip);
        // below that the phases run as separate launches that also spread over the pixels / tiles inside a pair.
        // VSTAB_DIS_SPLIT = 0 | 1 forces a form (A/B measurement, tests); default: split below 7/8 of the CUs (measured crossover, profiles/r02_dis_launch_forms.md).
        static const int n_cu = [&] { int v = 256; (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device); return v; }();
        bool split = 8 * P < 7 * n_cu;
        int forced = -1;
        if (const char* e = getenv("VSTAB_DIS_SPLIT")) { forced = atoi(e); split = forced != 0; }
        // a level that is a single tile has nothing to spread inside a pair: its ten split launches would only add
        // launch boundaries to work that is latency-bound anyway (VSTAB_DIS_SPLIT=2 splits every level regardless)
        if (split && forced != 2 && la.tiles_x * la.tiles_y == 1) split = false;
        const bool fuse_grid = !split && i == FINEST && grid_flow != nullptr;
        if (fuse_grid) {
            la.grid_out = grid_flow;
            la.g_h = (h + sample_step - 1) / sample_step; la.g_w = (w + sample_step - 1) / sample_step; la.g_step = sample_step;
            la.g_sx = 1. / ((double)w / F.w); la.g_sy = 1. / ((double)h / F.h); la.g_mul = (float)(1 << FINEST);
            grid_fused = true;
        }
        if (!split) {
            if (vr_lds_bytes > 64 * 1024)
                VSTAB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(level_kernel<LEVEL_FUSED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)vr_lds_bytes));
            la.parts = 1;
            hipLaunchKernelGGL(level_kernel<LEVEL_FUSED>, dim3((unsigned)P), dim3(FUSED_T), vr_lds_bytes, st, la);
        } else {
            const int px_parts = std::max(1, std::min(64, (g.h * g.w + FUSED_T - 1) / FUSED_T));
            la.parts = px_parts;
            hipLaunchKernelGGL(level_kernel<LEVEL_PRE>, dim3((unsigned)(P * px_parts)), dim3(FUSED_T), 0, st, la);
            hipLaunchKernelGGL(level_kernel<LEVEL_DERIV>, dim3((unsigned)(P * px_parts)), dim3(FUSED_T), 0, st, la);
            if (vr_lds_bytes > 64 * 1024)
                VSTAB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(level_kernel<LEVEL_TILE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)vr_lds_bytes));
            la.parts = la.tiles_x * la.tiles_y;
            for (int it = 0; it < VAR_ITERS; it++) {
                la.it = it;
                hipLaunchKernelGGL(level_kernel<LEVEL_TILE>, dim3((unsigned)(P * la.parts)), dim3(FUSED_T), vr_lds_bytes, st, la);
            }
            la.parts = px_parts;
            hipLaunchKernelGGL(level_kernel<LEVEL_MERGE>, dim3((unsigned)(P * px_parts)), dim3(FUSED_T), 0, st, la);
            if (la.nextU != nullptr) {
                la.parts = std::max(1, std::min(64, (la.nhs * la.nws + FUSED_T - 1) / FUSED_T));
                hipLaunchKernelGGL(level_kernel<LEVEL_UPSAMPLE>, dim3((unsigned)(P * la.parts)), dim3(FUSED_T), 0, st, la);
            }
        }
        VSTAB_HIP(hipGetLastError());
        stage_timer.reset();
    }
    DetailTimer final_timer(ctx, "dis_final");
    const double fsx = 1. / ((double)w / F.w), fsy = 1. / ((double)h / F.h);
    const float mul = (float)(1 << FINEST);
    if (grid_flow && !grid_fused) {
        const int gh = (h + sample_step - 1) / sample_step, gw = (w + sample_step - 1) / sample_step;
        hipLaunchKernelGGL(final_sample_kernel, frame_grid((long long)gh * gw, P), dim3(256), 0, st, Ul[FINEST], Vl[FINEST], grid_flow, P,
                           F.h, F.w, gh, gw, sample_step, fsx, fsy, mul);
    }
    if (flow) {
        hipLaunchKernelGGL(final_sample_kernel, frame_grid((long long)h * w, P), dim3(256), 0, st, Ul[FINEST], Vl[FINEST], flow, P, F.h, F.w,
                           h, w, 1, fsx, fsy, mul);
    }
    VSTAB_HIP(hipGetLastError());
    return 0;
}

extern "C" int vstab_dis_flow_batch(vstab_ctx* ctx, const uint8_t* gray, int n, int h, int w, float* flow, float* grid_flow,
                                    int sample_step)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_dis_flow_batch: ctx is NULL");
    VSTAB_REQUIRE(gray != nullptr, "vstab_dis_flow_batch: gray is NULL");
    VSTAB_REQUIRE(n >= 2, "vstab_dis_flow_batch: need at least 2 frames, got %d", n);
    VSTAB_REQUIRE(flow != nullptr || grid_flow != nullptr, "vstab_dis_flow_batch: no output requested");
    VSTAB_REQUIRE(sample_step >= 1, "vstab_dis_flow_batch: sample_step must be >= 1");
    VSTAB_REQUIRE(h > 0 && w > 0, "vstab_dis_flow_batch: non-positive size");
    // One DIS object serves the whole clip in the reference (flow.py:316): its first calc() may auto-select the
    // scales for a tiny image and keeps the new finest scale; later calls recompute the coarsest scale only.
    int f0 = DEFAULT_FINEST, c0 = 0;
    VSTAB_REQUIRE(dis_scales(h, w, f0, c0), "vstab_dis_flow_batch: %dx%d is too small for DIS (OpenCV needs width or height >= 12)", w, h);
    int f1 = f0, c1 = 0;
    VSTAB_REQUIRE(dis_scales(h, w, f1, c1), "vstab_dis_flow_batch: scale selection failed for %dx%d", w, h);
    VSTAB_HIP(hipSetDevice(ctx->device));
    KernelTimer timer(ctx, "dis");
    const bool stateful = ctx->dis_first_pair_is_clip_start && (c1 != c0 || f1 != f0);
    if (!stateful) return dis_run(ctx, gray, n, h, w, ctx->dis_first_pair_is_clip_start ? f0 : f1, ctx->dis_first_pair_is_clip_start ? c0 : c1, flow, grid_flow, sample_step);
    if (int rc = dis_run(ctx, gray, 2, h, w, f0, c0, flow, grid_flow, sample_step)) return rc;
    if (n == 2) return 0;
    const int gh = (h + sample_step - 1) / sample_step, gw = (w + sample_step - 1) / sample_step;
    return dis_run(ctx, gray + (size_t)h * w, n - 1, h, w, f1, c1, flow ? flow + (size_t)h * w * 2 : nullptr,
                   grid_flow ? grid_flow + (size_t)gh * gw * 2 : nullptr, sample_step);
}

extern "C" int vstab_dis_set_clip_start(vstab_ctx* ctx, int first_pair_is_clip_start)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_dis_set_clip_start: ctx is NULL");
    ctx->dis_first_pair_is_clip_start = first_pair_is_clip_start != 0;
    return 0;
}
