// vstab_classic.hip -- sparse estimator of the `Video Stabilizer Classic` node on gfx950.
//
// Replaces the two OpenCV calls of nodes/video_stabilizer_classic.py:76-96 for every frame / frame pair
// of a clip at once:
//   cv2.goodFeaturesToTrack(prev_gray, 400, 0.01, 7, blockSize=21)      -> vstab_gftt_batch
//   cv2.calcOpticalFlowPyrLK(prev, curr, features, winSize=(31,31), maxLevel=3,
//                            criteria=(EPS|COUNT, 50, 0.01))            -> vstab_lk_track_batch
// The arithmetic definitions are those of oracle/vo_classic.c (same choices where OpenCV's own result
// depends on its SIMD dispatch), so both sides are compared bit for bit.
//
// Mapping to the machine:
//   * min-eigenvalue map: one 1024-thread workgroup per 32 x 32 tile keeps the Sobel products of the tile + box
//     halo and their fp64 row sums in LDS (separable 21x21 box sums, fp64 accumulators, fixed summation order)
//   * corner candidates (3x3 maxima above quality*max) become 64-bit keys (strength bits << 32 | pixel index);
//     their slots come from a count / scan / write pass pair (no per-candidate atomics); one 1024-thread block per
//     frame sorts them (bitonic, in place) and one wavefront runs OpenCV's greedy minimum-distance selection with
//     a 64-wide parallel distance test per candidate
//   * pyramidal LK: one wavefront per tracked point, the 31x31 window lives in registers (16 px per lane),
//     the 2x2 normal matrix / mismatch vector are exact integer wave reductions (64-bit)
#include "vstab_internal.h"
#include <cmath>

namespace {

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// ------------------------------------------------------------------------------------------------
// cornerMinEigenVal
// ------------------------------------------------------------------------------------------------
constexpr int ROW_T = 256;
constexpr int MAX_BLOCK = 31;   // box size limit (LDS halo)

__device__ __forceinline__ void sobel_cov(const uint8_t* __restrict__ img, int h, int w, int x, int y, float s, float k2,
                                          float& cxx, float& cxy, float& cyy)
{
    const uint8_t* r0 = img + (size_t)reflect101(y - 1, h) * w;
    const uint8_t* r1 = img + (size_t)y * w;
    const uint8_t* r2 = img + (size_t)reflect101(y + 1, h) * w;
    const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
    const float d0 = (float)((int)r0[xr] - (int)r0[xl]), d1 = (float)((int)r1[xr] - (int)r1[xl]), d2 = (float)((int)r2[xr] - (int)r2[xl]);
    const float dx = __builtin_fmaf(d0 + d2, s, d1 * k2);
    const float s0 = __builtin_fmaf(s, (float)r0[xr], __builtin_fmaf(k2, (float)r0[x], s * (float)r0[xl]));
    const float s2 = __builtin_fmaf(s, (float)r2[xr], __builtin_fmaf(k2, (float)r2[x], s * (float)r2[xl]));
    const float dy = s2 - s0;
    cxx = dx * dx; cxy = dx * dy; cyy = dy * dy;
}

__device__ __forceinline__ unsigned float_order_key(float v)
{
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float float_from_order_key(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Fused min-eigenvalue map: one 256-thread workgroup per 32 x 32 output tile.  The Sobel products of the tile plus
// its box-filter halo are formed once in LDS (f32), summed along x into fp64 row sums in LDS, then along y -- the
// summation orders of the two-pass version (k ascending), so the bits are the same, but the 504 B per pixel of fp64
// row sums never travel through HBM.  grid (tiles_x, tiles_y, frames); dynamic LDS = cov + row sums.
constexpr int EIG_TILE = 32;
constexpr int EIG_T = 1024;   // threads: the three phases are latency-bound loops over LDS, 32 waves per CU hide it
__global__ __launch_bounds__(EIG_T) void eig_tile_kernel(const uint8_t* __restrict__ gray, int h, int w, int block, float s,
                                                       float* __restrict__ eig, unsigned* __restrict__ max_key)
{
    extern __shared__ unsigned char eig_lds[];
    __shared__ unsigned s_max[EIG_T / 64];
    const int r = block / 2;
    const int ext = EIG_TILE + block - 1;                 // tile + halo, both directions
    float* cov = reinterpret_cast<float*>(eig_lds);       // [3][ext][ext]
    double* rows = reinterpret_cast<double*>(eig_lds + (((size_t)3 * ext * ext * sizeof(float) + 15) & ~size_t(15)));   // [3][ext][EIG_TILE]
    const int x0 = blockIdx.x * EIG_TILE, y0 = blockIdx.y * EIG_TILE, f = blockIdx.z;
    const uint8_t* img = gray + (size_t)f * h * w;
    const float k2 = s * 2.f;
    const int cplane = ext * ext;
    for (int k = threadIdx.x; k < cplane; k += EIG_T) {
        const int ly = k / ext, lx = k - ly * ext;
        float a, b, c;
        sobel_cov(img, h, w, reflect101(x0 - r + lx, w), reflect101(y0 - r + ly, h), s, k2, a, b, c);
        cov[k] = a; cov[cplane + k] = b; cov[2 * cplane + k] = c;
    }
    __syncthreads();
    const int rplane = ext * EIG_TILE;
    for (int k = threadIdx.x; k < rplane; k += EIG_T) {
        const int ly = k / EIG_TILE, lx = k - ly * EIG_TILE;
        const float* p = cov + ly * ext + lx;
        double a = 0, b = 0, c = 0;
        for (int q = 0; q < block; q++) { a += (double)p[q]; b += (double)p[cplane + q]; c += (double)p[2 * cplane + q]; }
        rows[k] = a; rows[rplane + k] = b; rows[2 * rplane + k] = c;
    }
    __syncthreads();
    unsigned key = 0;
    for (int k = threadIdx.x; k < EIG_TILE * EIG_TILE; k += EIG_T) {
        const int ly = k / EIG_TILE, lx = k - ly * EIG_TILE;
        const int x = x0 + lx, y = y0 + ly;
        if (x < w && y < h) {
            const double* p = rows + ly * EIG_TILE + lx;
            double a = 0, b = 0, c = 0;
            for (int q = 0; q < block; q++) { a += p[q * EIG_TILE]; b += p[rplane + q * EIG_TILE]; c += p[2 * rplane + q * EIG_TILE]; }
            const float fa = (float)a * 0.5f, fb = (float)b, fc = (float)c * 0.5f;
            const float t = fa - fc;
            const float e = (fa + fc) - __builtin_sqrtf(__builtin_fmaf(fb, fb, t * t));
            eig[((size_t)f * h + y) * w + x] = e;
            const unsigned ke = float_order_key(e);
            key = ke > key ? ke : key;
        }
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) { const unsigned o = __shfl_down(key, sft); key = o > key ? o : key; }
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = s_max[0];
        for (int i = 1; i < EIG_T / 64; i++) m = s_max[i] > m ? s_max[i] : m;
        atomicMax(&max_key[f], m);
    }
}

// Thresholded 3x3 maxima of the interior -> candidate keys.  Slots are assigned without atomics (a returning atomic
// on one counter per frame serialises: ~0.4 us per allocation, 0.7-0.8 ms per clip): a first pass counts the
// candidates of every row, a scan turns the counts into row offsets, a second pass recomputes the same test and
// writes each candidate to its slot.  grid (h - 2, frames), one workgroup per interior row.
template <bool WRITE>
__global__ __launch_bounds__(ROW_T) void corner_collect_kernel(const float* __restrict__ eig, const unsigned* __restrict__ max_key, int h, int w,
                                                               double quality, unsigned long long* __restrict__ cand, size_t cap,
                                                               int* __restrict__ row_count /*[frames][h]: counts, then offsets*/)
{
    __shared__ int s_wave[ROW_T / 64];
    __shared__ int s_base;
    const int y = blockIdx.x + 1, f = blockIdx.y;
    const float thr = (float)((double)float_from_order_key(max_key[f]) * quality);
    const float* r1 = eig + ((size_t)f * h + y) * w;
    const float* r0 = r1 - w;
    const float* r2 = r1 + w;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int running = WRITE ? row_count[(size_t)f * h + y] : 0;   // WRITE: first slot of this row; else: count so far
    for (int xb = 1; xb < w - 1; xb += ROW_T) {
        const int x = xb + (int)threadIdx.x;
        float v = 0.f;
        bool is_max = false;
        if (x < w - 1) {
            v = r1[x];
            v = v > thr ? v : 0.f;
            if (v != 0.f) {
                float m = v;
#define CC_TAP(p) { float q = (p); q = q > thr ? q : 0.f; m = q > m ? q : m; }
                CC_TAP(r0[x - 1]) CC_TAP(r0[x]) CC_TAP(r0[x + 1]) CC_TAP(r1[x - 1]) CC_TAP(r1[x + 1]) CC_TAP(r2[x - 1]) CC_TAP(r2[x]) CC_TAP(r2[x + 1])
#undef CC_TAP
                is_max = (v == m);
            }
        }
        const unsigned long long vote = __ballot(is_max);
        if (lane == 0) s_wave[wave] = __popcll(vote);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < ROW_T / 64; k++) { const int c = s_wave[k]; before += (k < wave) ? c : 0; total += c; }
        if (WRITE && is_max) {
            const int slot = running + before + __popcll(vote & ((1ull << lane) - 1ull));
            cand[(size_t)f * cap + slot] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(y * w + x);   // v > 0: bits are monotone
        }
        running += total;
        __syncthreads();
    }
    if (!WRITE && threadIdx.x == 0) row_count[(size_t)f * h + y] = running;
    (void)s_base;
}

// one workgroup per frame: exclusive scan of the row counts (rows 1 .. h-2) in place, total -> cand_count[f]
__global__ __launch_bounds__(256) void corner_scan_kernel(int* __restrict__ row_count, int h, int* __restrict__ cand_count)
{
    __shared__ int s_part[256];
    const int f = blockIdx.x;
    int* rc = row_count + (size_t)f * h;
    const int per = (h + 255) / 256;
    const int lo = threadIdx.x * per, hi = min(lo + per, h);
    int sum = 0;
    for (int r = lo; r < hi; r++) sum += (r >= 1 && r < h - 1) ? rc[r] : 0;
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < 256; i++) { const int v = s_part[i]; s_part[i] = acc; acc += v; }
        cand_count[f] = acc;
    }
    __syncthreads();
    int acc = s_part[threadIdx.x];
    for (int r = lo; r < hi; r++) {
        const int c = (r >= 1 && r < h - 1) ? rc[r] : 0;
        rc[r] = acc;
        acc += c;
    }
}

constexpr int SEL_T = 1024;
constexpr int MAX_CORNERS = 4096;

// one block per frame: sort the candidate keys descending (value, then pixel index = OpenCV's greaterThanPtr),
// then the greedy minimum-distance selection of featureselect.cpp on wavefront 0
__global__ __launch_bounds__(SEL_T) void corner_select_kernel(unsigned long long* __restrict__ cand, size_t cap, const int* __restrict__ cand_count,
                                                              int w, int max_corners, float min_dist2, int use_dist,
                                                              float* __restrict__ corners, int* __restrict__ counts)
{
    __shared__ float s_x[MAX_CORNERS], s_y[MAX_CORNERS];
    const int f = blockIdx.x;
    unsigned long long* K = cand + (size_t)f * cap;
    const int total = cand_count[f];
    int n2 = 1;
    while (n2 < total) n2 <<= 1;
    for (int i = total + threadIdx.x; i < n2; i += SEL_T) K[i] = 0ULL;
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < n2; i += SEL_T) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = K[i], b = K[ixj];
                    const bool desc = (i & k) == 0;
                    if ((a < b) == desc) { K[i] = b; K[ixj] = a; }
                }
            }
        }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    int nacc = 0;
    for (int base = 0; base < total && nacc < max_corners; base += 64) {
        const unsigned long long mine = (base + lane < total) ? K[base + lane] : 0ULL;
        const int lim = min(64, total - base);
        for (int t = 0; t < lim; t++) {
            const unsigned idx = (unsigned)__shfl(mine, t);
            const int y = (int)(idx / (unsigned)w), x = (int)(idx - (unsigned)y * (unsigned)w);
            bool good = true;
            if (use_dist)
                for (int j0 = 0; j0 < nacc; j0 += 64) {
                    const int j = j0 + lane;
                    bool bad = false;
                    if (j < nacc) {
                        const float dx = (float)x - s_x[j], dy = (float)y - s_y[j];
                        bad = dx * dx + dy * dy < min_dist2;
                    }
                    if (__ballot(bad) != 0ULL) { good = false; break; }
                }
            if (good) {
                if (lane == 0) {
                    s_x[nacc] = (float)x; s_y[nacc] = (float)y;
                    corners[((size_t)f * max_corners + nacc) * 2] = (float)x;
                    corners[((size_t)f * max_corners + nacc) * 2 + 1] = (float)y;
                }
                nacc++;
                __builtin_amdgcn_wave_barrier();
                if (nacc == max_corners) break;
            }
        }
    }
    if (lane == 0) counts[f] = nacc;
}

// ------------------------------------------------------------------------------------------------
// pyramids for LK
// ------------------------------------------------------------------------------------------------
__global__ void pyr_down_kernel(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ dst, int dh, int dw)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= dw) return;
    const uint8_t* S = src + (size_t)f * sh * sw;
    int cols[5];
#pragma unroll
    for (int i = 0; i < 5; i++) cols[i] = reflect101(2 * x - 2 + i, sw);
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const uint8_t* row = S + (size_t)reflect101(2 * y - 2 + j, sh) * sw;
        const int racc = row[cols[0]] + 4 * row[cols[1]] + 6 * row[cols[2]] + 4 * row[cols[3]] + row[cols[4]];
        acc += (j == 0 || j == 4) ? racc : ((j == 2) ? 6 * racc : 4 * racc);
    }
    dst[((size_t)f * dh + y) * dw + x] = (uint8_t)((acc + 128) >> 8);
}

__global__ void scharr_kernel(const uint8_t* __restrict__ src, int h, int w, short2* __restrict__ deriv)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= w) return;
    const uint8_t* S = src + (size_t)f * h * w;
    const uint8_t* r0 = S + (size_t)reflect101(y - 1, h) * w;
    const uint8_t* r1 = S + (size_t)y * w;
    const uint8_t* r2 = S + (size_t)reflect101(y + 1, h) * w;
    const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
    const int t0l = ((int)r0[xl] + (int)r2[xl]) * 3 + (int)r1[xl] * 10, t0r = ((int)r0[xr] + (int)r2[xr]) * 3 + (int)r1[xr] * 10;
    const int t1l = (int)r2[xl] - (int)r0[xl], t1c = (int)r2[x] - (int)r0[x], t1r = (int)r2[xr] - (int)r0[xr];
    deriv[((size_t)f * h + y) * w + x] = make_short2((short)(t0r - t0l), (short)((t1r + t1l) * 3 + t1c * 10));
}

// ------------------------------------------------------------------------------------------------
// pyramidal LK
// ------------------------------------------------------------------------------------------------
constexpr int LK_MAX_LEVELS = 8;
constexpr int LK_NPL = 16;        // window pixels per lane: 31*31 = 961 <= 16*64
constexpr int LK_WAVES = 4;

struct LkLevels {
    int levels;                   // index of the coarsest level
    int h[LK_MAX_LEVELS], w[LK_MAX_LEVELS];
    size_t img_off[LK_MAX_LEVELS];     // byte offset of level l (all frames) in the image scratch
    size_t der_off[LK_MAX_LEVELS];     // element (short2) offset of level l in the derivative scratch
};

struct LkArgs {
    const uint8_t* level0;        // [frames][h][w] = the caller's gray clip
    const uint8_t* pyr;           // levels >= 1
    const short2* deriv;
    const float* pts;             // [pairs][max_pts][2]
    const int* counts;            // [pairs]
    float* out;                   // [pairs][max_pts][4]: prev.x, prev.y, next.x, next.y (next = NaN when status == 0)
    float* next_pts;              // optional [pairs][max_pts][2]
    uint8_t* status;              // optional [pairs][max_pts]
    int pairs, max_pts, win, max_count;
    double epsilon2;
    LkLevels lv;
};

__device__ __forceinline__ long long wave_sum_ll(long long v)
{
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

#define LK_DESCALE(v, n) (((v) + (1 << ((n) - 1))) >> (n))

__device__ __forceinline__ void lk_weights(float a, float b, int& w00, int& w01, int& w10, int& w11)
{
    w00 = (int)__builtin_rintf((1.f - a) * (1.f - b) * 16384.f);
    w01 = (int)__builtin_rintf(a * (1.f - b) * 16384.f);
    w10 = (int)__builtin_rintf((1.f - a) * b * 16384.f);
    w11 = 16384 - w00 - w01 - w10;
}

__global__ __launch_bounds__(64 * LK_WAVES) void lk_kernel(LkArgs a)
{
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.y;
    const int p = blockIdx.x * LK_WAVES + (threadIdx.x >> 6);
    if (p >= a.counts[pair]) return;   // wave-uniform
    const int win = a.win, npx = win * win;
    const float half = (float)(win - 1) * 0.5f;
    const size_t pidx = (size_t)pair * a.max_pts + p;
    const float ptx = a.pts[pidx * 2], pty = a.pts[pidx * 2 + 1];

    // window coordinates of this lane's pixels
    int wx[LK_NPL], wy[LK_NPL];
#pragma unroll
    for (int t = 0; t < LK_NPL; t++) {
        const int q = lane + 64 * t;
        wy[t] = q / win; wx[t] = q - wy[t] * win;
    }
    float ox = 0.f, oy = 0.f;       // nextPts[ptidx]
    int status = 1;
    for (int level = a.lv.levels; level >= 0; level--) {
        const int lh = a.lv.h[level], lw = a.lv.w[level];
        const size_t fsz = (size_t)lh * lw;
        const uint8_t* base = level == 0 ? a.level0 : a.pyr + a.lv.img_off[level];
        const uint8_t* I = base + (size_t)pair * fsz;
        const uint8_t* J = base + (size_t)(pair + 1) * fsz;
        const short2* dI = a.deriv + a.lv.der_off[level] + (size_t)pair * fsz;
        const float sc = (float)(1. / (double)(1 << level));
        float px = ptx * sc, py = pty * sc;
        float nx, ny;
        if (level == a.lv.levels) { nx = px; ny = py; }
        else { nx = ox * 2.f; ny = oy * 2.f; }
        ox = nx; oy = ny;
        px -= half; py -= half;
        const int ipx = (int)__builtin_floorf(px), ipy = (int)__builtin_floorf(py);
        if (ipx < -win || ipx >= lw || ipy < -win || ipy >= lh) {
            if (level == 0) status = 0;
            continue;
        }
        int w00, w01, w10, w11;
        lk_weights(px - (float)ipx, py - (float)ipy, w00, w01, w10, w11);
        int Iv[LK_NPL], Dv[LK_NPL];   // Dv packs (ix, iy) as two shorts
        int pA11 = 0, pA12 = 0, pA22 = 0;
        const bool inside = ipx >= 0 && ipy >= 0 && ipx + win < lw && ipy + win < lh;   // window + bilinear neighbour in range
#pragma unroll
        for (int t = 0; t < LK_NPL; t++) {
            Iv[t] = 0; Dv[t] = 0;
            if (lane + 64 * t < npx) {
                const int X = ipx + wx[t], Y = ipy + wy[t];
                int i00, i01, i10, i11;
                short2 d00, d01, d10, d11;
                if (inside) {
                    const size_t o = (size_t)Y * lw + X;
                    i00 = I[o]; i01 = I[o + 1]; i10 = I[o + lw]; i11 = I[o + lw + 1];
                    d00 = dI[o]; d01 = dI[o + 1]; d10 = dI[o + lw]; d11 = dI[o + lw + 1];
                } else {
                    const int X0 = reflect101(X, lw), X1 = reflect101(X + 1, lw), Y0 = reflect101(Y, lh), Y1 = reflect101(Y + 1, lh);
                    i00 = I[(size_t)Y0 * lw + X0]; i01 = I[(size_t)Y0 * lw + X1]; i10 = I[(size_t)Y1 * lw + X0]; i11 = I[(size_t)Y1 * lw + X1];
                    const bool xa = (unsigned)X < (unsigned)lw, xb = (unsigned)(X + 1) < (unsigned)lw;
                    const bool ya = (unsigned)Y < (unsigned)lh, yb = (unsigned)(Y + 1) < (unsigned)lh;
                    const short2 z = make_short2(0, 0);
                    d00 = (xa && ya) ? dI[(size_t)Y * lw + X] : z;
                    d01 = (xb && ya) ? dI[(size_t)Y * lw + X + 1] : z;
                    d10 = (xa && yb) ? dI[(size_t)(Y + 1) * lw + X] : z;
                    d11 = (xb && yb) ? dI[(size_t)(Y + 1) * lw + X + 1] : z;
                }
                const int ival = LK_DESCALE(i00 * w00 + i01 * w01 + i10 * w10 + i11 * w11, 9);
                const int ixv = LK_DESCALE((int)d00.x * w00 + (int)d01.x * w01 + (int)d10.x * w10 + (int)d11.x * w11, 14);
                const int iyv = LK_DESCALE((int)d00.y * w00 + (int)d01.y * w01 + (int)d10.y * w10 + (int)d11.y * w11, 14);
                Iv[t] = ival;
                Dv[t] = (ixv & 0xffff) | (int)((unsigned)iyv << 16);
                pA11 += ixv * ixv; pA12 += ixv * iyv; pA22 += iyv * iyv;
            }
        }
        const long long sA11 = wave_sum_ll(pA11), sA12 = wave_sum_ll(pA12), sA22 = wave_sum_ll(pA22);
        const float FLT_SCALE = 1.f / (float)(1 << 20);
        const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float min_eig = (A22 + A11 - __builtin_sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
        if (min_eig < 1e-4f || D < 1.1920928955078125e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_count; j++) {
            const int inx = (int)__builtin_floorf(nx), iny = (int)__builtin_floorf(ny);
            if (inx < -win || inx >= lw || iny < -win || iny >= lh) {
                if (level == 0) status = 0;
                break;
            }
            lk_weights(nx - (float)inx, ny - (float)iny, w00, w01, w10, w11);
            const bool jin = inx >= 0 && iny >= 0 && inx + win < lw && iny + win < lh;
            int pb1 = 0, pb2 = 0;
#pragma unroll
            for (int t = 0; t < LK_NPL; t++) {
                if (lane + 64 * t < npx) {
                    const int X = inx + wx[t], Y = iny + wy[t];
                    int j00, j01, j10, j11;
                    if (jin) {
                        const size_t o = (size_t)Y * lw + X;
                        j00 = J[o]; j01 = J[o + 1]; j10 = J[o + lw]; j11 = J[o + lw + 1];
                    } else {
                        const int X0 = reflect101(X, lw), X1 = reflect101(X + 1, lw), Y0 = reflect101(Y, lh), Y1 = reflect101(Y + 1, lh);
                        j00 = J[(size_t)Y0 * lw + X0]; j01 = J[(size_t)Y0 * lw + X1]; j10 = J[(size_t)Y1 * lw + X0]; j11 = J[(size_t)Y1 * lw + X1];
                    }
                    const int diff = LK_DESCALE(j00 * w00 + j01 * w01 + j10 * w10 + j11 * w11, 9) - Iv[t];
                    pb1 += diff * (int)(short)(Dv[t] & 0xffff);
                    pb2 += diff * (Dv[t] >> 16);
                }
            }
            const long long sb1 = wave_sum_ll(pb1), sb2 = wave_sum_ll(pb2);
            const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
            nx += dx; ny += dy;
            ox = nx + half; oy = ny + half;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= a.epsilon2) break;
            if (j > 0 && (double)__builtin_fabsf(dx + pdx) < 0.01 && (double)__builtin_fabsf(dy + pdy) < 0.01) {
                ox -= dx * 0.5f; oy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status && level == 0) {   // the tracker's error block: the final window must be addressable
            const float fx = ox - half, fy = oy - half;
            const int ix = (int)__builtin_floorf(fx), iy = (int)__builtin_floorf(fy);
            if (ix < -win || ix >= lw || iy < -win || iy >= lh) status = 0;
        }
    }
    if (lane == 0) {
        const float nan = __uint_as_float(0x7fc00000u);
        reinterpret_cast<float4*>(a.out)[pidx] = make_float4(ptx, pty, status ? ox : nan, status ? oy : nan);
        if (a.next_pts) { a.next_pts[pidx * 2] = ox; a.next_pts[pidx * 2 + 1] = oy; }
        if (a.status) a.status[pidx] = (uint8_t)status;
    }
}

__global__ void fill_untracked_kernel(float* out, uint8_t* status, const int* counts, int max_pts, int pairs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, pair = blockIdx.y;
    if (i >= max_pts || i < counts[pair]) return;
    const float nan = __uint_as_float(0x7fc00000u);
    reinterpret_cast<float4*>(out)[(size_t)pair * max_pts + i] = make_float4(0.f, 0.f, nan, nan);
    if (status) status[(size_t)pair * max_pts + i] = 0;
}

size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

}  // namespace

extern "C" int vstab_gftt_batch(vstab_ctx* ctx, const uint8_t* gray, int n, int h, int w, int max_corners, double quality,
                                double min_distance, int block_size, float* corners, int* counts)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_gftt_batch: ctx is NULL");
    VSTAB_REQUIRE(gray && corners && counts, "vstab_gftt_batch: NULL pointer argument");
    VSTAB_REQUIRE(n > 0 && h > 0 && w > 0, "vstab_gftt_batch: non-positive size");
    VSTAB_REQUIRE(max_corners > 0 && max_corners <= MAX_CORNERS, "vstab_gftt_batch: max_corners %d outside 1..%d", max_corners, MAX_CORNERS);
    VSTAB_REQUIRE(block_size >= 1 && block_size <= MAX_BLOCK, "vstab_gftt_batch: block_size %d outside 1..%d", block_size, MAX_BLOCK);
    VSTAB_REQUIRE(quality > 0.0 && min_distance >= 0.0, "vstab_gftt_batch: quality must be positive and min_distance non-negative");
    VSTAB_REQUIRE((long long)h * w < (1LL << 31), "vstab_gftt_batch: image too large");
    VSTAB_HIP(hipSetDevice(ctx->device));
    const size_t px = (size_t)h * w;
    size_t cap = 1;
    while (cap < px) cap <<= 1;                 // bitonic sort pads to a power of two in place
    // frames per pass: keep the eigenvalue maps + candidate keys under ~1 GiB
    const size_t per_frame = px * sizeof(float) + cap * sizeof(unsigned long long) + sizeof(int) * (size_t)h + 64;
    int chunk = (int)((size_t(1) << 30) / per_frame);
    chunk = chunk < 1 ? 1 : (chunk > n ? n : chunk);
    const size_t rows_b = 0, eig_b = align256(px * chunk * sizeof(float));
    const size_t key_b = align256(cap * chunk * sizeof(unsigned long long)), small_b = align256(sizeof(unsigned) * chunk) + align256(sizeof(int) * chunk) +
                 align256(sizeof(int) * (size_t)chunk * h);
    if (ctx->d_dis.reserve(rows_b + eig_b + key_b + small_b)) return 1;
    char* base = static_cast<char*>(ctx->d_dis.ptr);
    float* eig = reinterpret_cast<float*>(base + rows_b);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(base + rows_b + eig_b);
    unsigned* max_key = reinterpret_cast<unsigned*>(base + rows_b + eig_b + key_b);
    int* cand_count = reinterpret_cast<int*>(base + rows_b + eig_b + key_b + align256(sizeof(unsigned) * chunk));
    int* row_count = reinterpret_cast<int*>(base + rows_b + eig_b + key_b + align256(sizeof(unsigned) * chunk) + align256(sizeof(int) * chunk));
    const float s = (float)(1.0 / (4.0 * block_size * 255.0));
    KernelTimer timer(ctx, "gftt");
    for (int f0 = 0; f0 < n; f0 += chunk) {
        const int fc = (n - f0) < chunk ? (n - f0) : chunk;
        VSTAB_HIP(hipMemsetAsync(max_key, 0, sizeof(unsigned) * fc, ctx->stream));
        VSTAB_HIP(hipMemsetAsync(cand_count, 0, sizeof(int) * fc, ctx->stream));
        {
            const int ext = EIG_TILE + block_size - 1;
            const size_t lds = (((size_t)3 * ext * ext * sizeof(float) + 15) & ~size_t(15)) + (size_t)3 * ext * EIG_TILE * sizeof(double);
            if (lds > 64 * 1024)
                VSTAB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(eig_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const dim3 tgrid((unsigned)((w + EIG_TILE - 1) / EIG_TILE), (unsigned)((h + EIG_TILE - 1) / EIG_TILE), (unsigned)fc);
            hipLaunchKernelGGL(eig_tile_kernel, tgrid, dim3(EIG_T), lds, ctx->stream, gray + (size_t)f0 * px, h, w, block_size, s, eig, max_key);
        }
        if (h > 2 && w > 2) {
            const dim3 cgrid((unsigned)(h - 2), (unsigned)fc);
            hipLaunchKernelGGL((corner_collect_kernel<false>), cgrid, dim3(ROW_T), 0, ctx->stream, eig, max_key, h, w, quality, keys, cap, row_count);
            hipLaunchKernelGGL(corner_scan_kernel, dim3((unsigned)fc), dim3(256), 0, ctx->stream, row_count, h, cand_count);
            hipLaunchKernelGGL((corner_collect_kernel<true>), cgrid, dim3(ROW_T), 0, ctx->stream, eig, max_key, h, w, quality, keys, cap, row_count);
        }
        hipLaunchKernelGGL(corner_select_kernel, dim3((unsigned)fc), dim3(SEL_T), 0, ctx->stream, keys, cap, cand_count, w, max_corners,
                           (float)(min_distance * min_distance), min_distance >= 1.0 ? 1 : 0, corners + (size_t)f0 * max_corners * 2, counts + f0);
        VSTAB_HIP(hipGetLastError());
    }
    return 0;
}

extern "C" int vstab_lk_levels(int h, int w, int win, int max_level)
{
    int level = 0;
    for (; level <= max_level; level++) {
        const int nh = (h + 1) / 2, nw = (w + 1) / 2;
        if (nw <= win || nh <= win) break;
        h = nh; w = nw;
    }
    return level > max_level ? max_level : level;
}

extern "C" int vstab_lk_track_batch(vstab_ctx* ctx, const uint8_t* gray, int n, int h, int w, const float* points, const int* counts,
                                    int max_points, int win, int max_level, int max_count, double epsilon, float* point_pairs,
                                    float* next_points, uint8_t* status)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_lk_track_batch: ctx is NULL");
    VSTAB_REQUIRE(gray && points && counts && point_pairs, "vstab_lk_track_batch: NULL pointer argument");
    VSTAB_REQUIRE(n >= 2 && h > 0 && w > 0 && max_points > 0, "vstab_lk_track_batch: need at least two frames and positive sizes");
    VSTAB_REQUIRE(win >= 3 && win * win <= 64 * LK_NPL, "vstab_lk_track_batch: window %d outside 3..31", win);
    VSTAB_REQUIRE(max_level >= 0 && max_level < LK_MAX_LEVELS, "vstab_lk_track_batch: max_level %d outside 0..%d", max_level, LK_MAX_LEVELS - 1);
    VSTAB_HIP(hipSetDevice(ctx->device));
    const int pairs = n - 1;
    LkArgs a{};
    a.lv.levels = vstab_lk_levels(h, w, win, max_level);
    size_t img_bytes = 0, der_elems = 0;
    {
        int lh = h, lw = w;
        for (int l = 0; l <= a.lv.levels; l++) {
            a.lv.h[l] = lh; a.lv.w[l] = lw;
            a.lv.img_off[l] = img_bytes;
            a.lv.der_off[l] = der_elems;
            if (l > 0) img_bytes += align256((size_t)lh * lw * n);
            der_elems += (align256((size_t)lh * lw * n * sizeof(short2))) / sizeof(short2);
            lh = (lh + 1) / 2; lw = (lw + 1) / 2;
        }
    }
    // level 0 is the caller's clip; img_off of levels >= 1 is relative to the pyramid scratch
    const size_t pyr_b = align256(img_bytes + 256);
    if (ctx->d_dis.reserve(pyr_b + der_elems * sizeof(short2) + 256)) return 1;
    uint8_t* pyr = static_cast<uint8_t*>(ctx->d_dis.ptr);
    short2* deriv = reinterpret_cast<short2*>(static_cast<char*>(ctx->d_dis.ptr) + pyr_b);
    KernelTimer timer(ctx, "lk");
    for (int l = 0; l <= a.lv.levels; l++) {
        const uint8_t* cur = l == 0 ? gray : pyr + a.lv.img_off[l];
        if (l > 0) {
            const uint8_t* prev = l == 1 ? gray : pyr + a.lv.img_off[l - 1];
            hipLaunchKernelGGL(pyr_down_kernel, dim3((unsigned)((a.lv.w[l] + 255) / 256), (unsigned)a.lv.h[l], (unsigned)n), dim3(256), 0, ctx->stream,
                               prev, a.lv.h[l - 1], a.lv.w[l - 1], pyr + a.lv.img_off[l], a.lv.h[l], a.lv.w[l]);
        }
        hipLaunchKernelGGL(scharr_kernel, dim3((unsigned)((a.lv.w[l] + 255) / 256), (unsigned)a.lv.h[l], (unsigned)n), dim3(256), 0, ctx->stream,
                           cur, a.lv.h[l], a.lv.w[l], deriv + a.lv.der_off[l]);
    }
    VSTAB_HIP(hipGetLastError());
    a.level0 = gray; a.pyr = pyr; a.deriv = deriv; a.pts = points; a.counts = counts;
    a.out = point_pairs; a.next_pts = next_points; a.status = status;
    a.pairs = pairs; a.max_pts = max_points; a.win = win;
    a.max_count = max_count < 0 ? 0 : (max_count > 100 ? 100 : max_count);
    const double eps = epsilon < 0. ? 0. : (epsilon > 10. ? 10. : epsilon);
    a.epsilon2 = eps * eps;
    hipLaunchKernelGGL(fill_untracked_kernel, dim3((unsigned)((max_points + 255) / 256), (unsigned)pairs), dim3(256), 0, ctx->stream,
                       point_pairs, status, counts, max_points, pairs);
    hipLaunchKernelGGL(lk_kernel, dim3((unsigned)((max_points + LK_WAVES - 1) / LK_WAVES), (unsigned)pairs), dim3(64 * LK_WAVES), 0, ctx->stream, a);
    VSTAB_HIP(hipGetLastError());
    return 0;
}
