// vstab_warp.hip -- perspective/similarity warp with padding mask and multi-sample motion blur.
//
// Replaces the OpenCV calls of nodes/video_stabilizer_flow.py:560-588 (F13) and
// nodes/motion_apply.py:75-202 (A3, A5) of the reference.  Arithmetic follows OpenCV's
// legacy warpPerspective/remap kernels (f64 coordinate math evaluated per 64-wide block,
// 1/32-px quantised fractions, f32 weights, left-to-right f32 sums) so that the result is
// bit-identical to oracle/vo_warp.c.  Built with -ffp-contract=off: no FMA may be formed.
//
// Kernel shape (HBM-bound, 28 B of algorithmic traffic per output pixel):
//   * one 256-thread block = 64 x 8 output tile (32 threads along x, 2 pixels per thread strided by 32, so the
//     lanes of a wavefront cover consecutive pixels: tap loads and stores touch whole cache lines); the bilinear
//     source footprint of the tile stays in the CU's L1, vertically adjacent tiles share a source row through L2
//   * blockIdx is remapped so that the 8 XCDs (private L2 each) own contiguous runs of tiles
//   * padding-pixel count: wave shuffle reduction -> LDS -> one atomic per block, only if non-zero
#include "vstab_internal.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr int TILE_PX = 2;        // pixels per thread along x, strided by the tile width (profiles/r01_warp_tile_sweep.md)
constexpr int MAX_BLUR_SAMPLES = 33;

// WarpXform (per frame / sample transform record): vstab_internal.h

struct WarpArgs {
    const float* src;
    float* dst;
    float* mask;
    unsigned* pad_count;
    const WarpXform* xf;
    int n, sh, sw, dh, dw;
    int bw0;          // column block width of OpenCV's WarpPerspectiveInvoker
    int bw0_pow2;     // 1 if bw0 is a power of two
    int tiles_x, tiles_y;
    int samples;      // 1 for the plain warp
    int nxf_per_frame;  // sample matrices stored per frame (1 for a single-frame blur clip)
    int blur_fast;      // blur: the interior fast path may be used (host-checked preconditions, see warp_blur_kernel)
    float b0, b1, b2;   // border colour
};

__device__ __forceinline__ int clamp_round_i32(double v)
{
    // std::max((double)INT_MIN, std::min((double)INT_MAX, v)) followed by cvRound
    const double hi = 2147483647.0, lo = -2147483648.0;
    double m = (v < hi) ? v : hi;
    double r = (lo < m) ? m : lo;
    return (int)__builtin_rint(r);
}

// Round-half-even of an fp64 value known to satisfy |v| < 2^31: adding 1.5*2^52 leaves the integer in the
// low 32 bits of the sum's bit pattern (one fp64 add instead of clamp + rint + convert).  Exactly what
// cvRound gives for such values.
__device__ __forceinline__ int round_small(double v)
{
    return (int)(unsigned)__double_as_longlong(v + 6755399441055744.0);
}

__device__ __forceinline__ int sat_short(int v)
{
    return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}

// initInterTab1D(INTER_CUBIC): A = -0.75, x = i/32, same operation order as OpenCV's interpolateCubic
__device__ __forceinline__ void cubic_coeffs(int i, float* c)
{
    const float A = -0.75f;
    const float x = i * (1.f / 32);
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

struct Px { float r, g, b; };

__device__ __forceinline__ Px load_px(const float* p)
{
    Px v;
    __builtin_memcpy(&v, p, 12);
    return v;
}

template <int INTERP>
__device__ __forceinline__ Px sample_q5(const float* __restrict__ S, int sh, int sw, int X, int Y,
                                        float b0, float b1, float b2, const float* cub_tab /* LDS [32][4] or nullptr */)
{
    const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
    const int fx = X & 31, fy = Y & 31;
    Px o;
    if (INTERP == VSTAB_INTERP_BILINEAR) {
        const float wx1 = fx * (1.f / 32), wx0 = 1.f - wx1;
        const float wy1 = fy * (1.f / 32), wy0 = 1.f - wy1;
        const float w0 = wy0 * wx0, w1 = wy0 * wx1, w2 = wy1 * wx0, w3 = wy1 * wx1;
        if ((unsigned)sx < (unsigned)(sw - 1) && (unsigned)sy < (unsigned)(sh - 1)) {
            const float* p = S + ((unsigned)sy * (unsigned)sw + (unsigned)sx) * 3u;
            float r0[6], r1[6];
            __builtin_memcpy(r0, p, 24);
            __builtin_memcpy(r1, p + (unsigned)sw * 3u, 24);
            o.r = r0[0] * w0 + r0[3] * w1 + r1[0] * w2 + r1[3] * w3;
            o.g = r0[1] * w0 + r0[4] * w1 + r1[1] * w2 + r1[4] * w3;
            o.b = r0[2] * w0 + r0[5] * w1 + r1[2] * w2 + r1[5] * w3;
            return o;
        }
        if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
            o.r = b0; o.g = b1; o.b = b2;
            return o;
        }
        const bool x0ok = sx >= 0 && sx < sw, x1ok = sx + 1 >= 0 && sx + 1 < sw;
        const bool y0ok = sy >= 0 && sy < sh, y1ok = sy + 1 >= 0 && sy + 1 < sh;
        const Px bd = {b0, b1, b2};
        const Px v0 = (x0ok && y0ok) ? load_px(S + ((unsigned)sy * (unsigned)sw + (unsigned)sx) * 3u) : bd;
        const Px v1 = (x1ok && y0ok) ? load_px(S + ((unsigned)sy * (unsigned)sw + (unsigned)(sx + 1)) * 3u) : bd;
        const Px v2 = (x0ok && y1ok) ? load_px(S + ((unsigned)(sy + 1) * (unsigned)sw + (unsigned)sx) * 3u) : bd;
        const Px v3 = (x1ok && y1ok) ? load_px(S + ((unsigned)(sy + 1) * (unsigned)sw + (unsigned)(sx + 1)) * 3u) : bd;
        o.r = v0.r * w0 + v1.r * w1 + v2.r * w2 + v3.r * w3;
        o.g = v0.g * w0 + v1.g * w1 + v2.g * w2 + v3.g * w3;
        o.b = v0.b * w0 + v1.b * w1 + v2.b * w2 + v3.b * w3;
        return o;
    } else {
        float cx[4], cy[4];
        if (cub_tab != nullptr) {   // OpenCV reads these from its own 32-entry table too (initInterTab1D): same values
            typedef float f4_t __attribute__((ext_vector_type(4)));
            const f4_t tx = reinterpret_cast<const f4_t*>(cub_tab)[fx], ty = reinterpret_cast<const f4_t*>(cub_tab)[fy];
            cx[0] = tx.x; cx[1] = tx.y; cx[2] = tx.z; cx[3] = tx.w;
            cy[0] = ty.x; cy[1] = ty.y; cy[2] = ty.z; cy[3] = ty.w;
        } else {
            cubic_coeffs(fx, cx);
            cubic_coeffs(fy, cy);
        }
        const int x0 = sx - 1, y0 = sy - 1;
        const unsigned width1 = (unsigned)(sw - 3 > 0 ? sw - 3 : 0);
        const unsigned height1 = (unsigned)(sh - 3 > 0 ? sh - 3 : 0);
        if ((unsigned)x0 < width1 && (unsigned)y0 < height1) {
            const float* p = S + ((unsigned)y0 * (unsigned)sw + (unsigned)x0) * 3u;
            float sr = 0.f, sg = 0.f, sb = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float row[12];
                __builtin_memcpy(row, p + (unsigned)i * (unsigned)sw * 3u, 48);
                const float w0 = cy[i] * cx[0], w1 = cy[i] * cx[1], w2 = cy[i] * cx[2], w3 = cy[i] * cx[3];
                const float tr = row[0] * w0 + row[3] * w1 + row[6] * w2 + row[9] * w3;
                const float tg = row[1] * w0 + row[4] * w1 + row[7] * w2 + row[10] * w3;
                const float tb = row[2] * w0 + row[5] * w1 + row[8] * w2 + row[11] * w3;
                if (i == 0) { sr = tr; sg = tg; sb = tb; }
                else { sr += tr; sg += tg; sb += tb; }
            }
            o.r = sr; o.g = sg; o.b = sb;
            return o;
        }
        if (x0 >= sw || x0 + 4 <= 0 || y0 >= sh || y0 + 4 <= 0) {
            o.r = b0; o.g = b1; o.b = b2;
            return o;
        }
        float sr = b0, sg = b1, sb = b2;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int yy = y0 + i;
            if (yy < 0 || yy >= sh) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int xx = x0 + j;
                if (xx < 0 || xx >= sw) continue;
                const Px v = load_px(S + ((unsigned)yy * (unsigned)sw + (unsigned)xx) * 3u);
                const float w = cy[i] * cx[j];
                sr += (v.r - b0) * w;
                sg += (v.g - b1) * w;
                sb += (v.b - b2) * w;
            }
        }
        o.r = sr; o.g = sg; o.b = sb;
        return o;
    }
}

__device__ __forceinline__ Px sample_exact(const float* __restrict__ S, int sh, int sw, float fsx, float fsy,
                                           float b0, float b1, float b2)
{
    Px o;
    const float flx = __builtin_floorf(fsx), fly = __builtin_floorf(fsy);
    const bool bad = !(fsx == fsx) || !(fsy == fsy) || flx >= 2.0e9f || flx <= -2.0e9f || fly >= 2.0e9f || fly <= -2.0e9f;
    const int ix = bad ? 0 : (int)flx, iy = bad ? 0 : (int)fly;
    const float ax = fsx - ix, ay = fsy - iy;
    if (bad || ix >= sw || ix + 1 < 0 || iy >= sh || iy + 1 < 0) {
        o.r = b0; o.g = b1; o.b = b2;
        return o;
    }
    const bool x0ok = ix >= 0 && ix < sw, x1ok = ix + 1 >= 0 && ix + 1 < sw;
    const bool y0ok = iy >= 0 && iy < sh, y1ok = iy + 1 >= 0 && iy + 1 < sh;
    const Px bd = {b0, b1, b2};
    const Px p00 = (x0ok && y0ok) ? load_px(S + ((unsigned)iy * (unsigned)sw + (unsigned)ix) * 3u) : bd;
    const Px p01 = (x1ok && y0ok) ? load_px(S + ((unsigned)iy * (unsigned)sw + (unsigned)(ix + 1)) * 3u) : bd;
    const Px p10 = (x0ok && y1ok) ? load_px(S + ((unsigned)(iy + 1) * (unsigned)sw + (unsigned)ix) * 3u) : bd;
    const Px p11 = (x1ok && y1ok) ? load_px(S + ((unsigned)(iy + 1) * (unsigned)sw + (unsigned)(ix + 1)) * 3u) : bd;
    float v0, v1;
    v0 = p00.r + ax * (p01.r - p00.r); v1 = p10.r + ax * (p11.r - p10.r); o.r = v0 + ay * (v1 - v0);
    v0 = p00.g + ax * (p01.g - p00.g); v1 = p10.g + ax * (p11.g - p10.g); o.g = v0 + ay * (v1 - v0);
    v0 = p00.b + ax * (p01.b - p00.b); v1 = p10.b + ax * (p11.b - p10.b); o.b = v0 + ay * (v1 - v0);
    return o;
}

// XCD-aware bijective remap of the linear block id (8 XCDs, round-robin dispatch):
// blocks that share an XCD get a contiguous run of logical tile ids.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nblk)
{
    const unsigned q = nblk >> 3, r = nblk & 7, x = b & 7, i = b >> 3;
    const unsigned base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + i;
}

template <int INTERP, int SUBPIX, bool WITH_MASK, int TILE_TX>
__global__ __launch_bounds__(256) void warp_kernel(WarpArgs a)
{
    constexpr int TILE_W = TILE_TX * TILE_PX, TILE_H = 256 / TILE_TX;
    __shared__ unsigned s_cnt[4];
    __shared__ __attribute__((aligned(16))) float s_cub[32 * 4];
    const float* cub_tab = nullptr;
    if (INTERP == VSTAB_INTERP_BICUBIC) {
        if (threadIdx.x < 32) cubic_coeffs((int)threadIdx.x, s_cub + threadIdx.x * 4);
        __syncthreads();
        cub_tab = s_cub;
    }
    const unsigned nblk = gridDim.x;
    const unsigned t = xcd_remap(blockIdx.x, nblk);
    const unsigned tiles_per_frame = (unsigned)a.tiles_x * a.tiles_y;
    const int frame = t / tiles_per_frame;
    const unsigned tr = t - frame * tiles_per_frame;
    const int tile_y = tr / a.tiles_x, tile_x = tr - tile_y * a.tiles_x;
    const int tx = threadIdx.x % TILE_TX, ty = threadIdx.x / TILE_TX;
    // Pixel p of a thread is x0 + p * TILE_TX: the lanes of a wavefront cover CONSECUTIVE output pixels, so one load
    // instruction of the bilinear taps touches ~13 cache lines (24 B per lane at a 12-B lane stride) instead of 48
    // (at the 48-B stride of 4 consecutive pixels per thread) -- the texture addresser was 83 % busy that way (PMC
    // TA_BUSY) and limited the read side to 2.8 TB/s -- and the stores are whole lines (12 B per lane, contiguous).
    const int x0 = tile_x * TILE_W + tx;
    const int y = tile_y * TILE_H + ty;
    const bool active = (y < a.dh) && (x0 < a.dw);
    const int npx = active ? (a.dw - x0 + TILE_TX - 1) / TILE_TX : 0;   // pixels x0 + p*TILE_TX < dw (capped at TILE_PX below)

    const float* __restrict__ S = a.src + (size_t)frame * a.sh * a.sw * 3;
    constexpr int nxf = 1;   // one matrix per frame (the S-sample blur is warp_blur_kernel)

    float acc[TILE_PX][3];
    float cov[TILE_PX];
#pragma unroll
    for (int p = 0; p < TILE_PX; p++) { acc[p][0] = acc[p][1] = acc[p][2] = 0.f; cov[p] = 0.f; }

    if (active) {
        const double dy = (double)y;

        for (int k = 0; k < nxf; k++) {
            const WarpXform* __restrict__ xf = a.xf + (size_t)frame * nxf + k;
            const double m0 = xf->m[0], m1 = xf->m[1], m2 = xf->m[2];
            const double m3 = xf->m[3], m4 = xf->m[4], m5 = xf->m[5];
            const double m6 = xf->m[6], m7 = xf->m[7], m8 = xf->m[8];
            const bool affine = xf->affine != 0;
            // OpenCV evaluates the row-start terms X0, Y0, W0 once per 64-wide column block (xb) and adds m * (x - xb)
            // per pixel; a thread's pixels lie in up to TILE_PX different blocks, so the terms are formed per pixel.
            float mf[9];
            if (SUBPIX == VSTAB_SUBPIX_EXACT && INTERP == VSTAB_INTERP_BILINEAR) {
#pragma unroll
                for (int i = 0; i < 9; i++) mf[i] = (float)xf->m[i];
            }
            const bool fast_ok = affine && !(SUBPIX == VSTAB_SUBPIX_EXACT && INTERP == VSTAB_INTERP_BILINEAR);
#pragma unroll
            for (int p = 0; p < TILE_PX; p++) {
                if (p >= npx) continue;
                const int x = x0 + p * TILE_TX;
                int xb;
                if (a.bw0 >= a.dw) xb = 0;
                else if (a.bw0_pow2) xb = x & ~(a.bw0 - 1);
                else xb = (x / a.bw0) * a.bw0;
                const double dxb = (double)xb;
                const double X0 = m0 * dxb + m1 * dy + m2;
                const double Y0 = m3 * dxb + m4 * dy + m5;
                const double W0 = m6 * dxb + m7 * dy + m8;
                const double dx1 = (double)(x - xb);
                const double Xn = X0 + m0 * dx1, Yn = Y0 + m3 * dx1;
                Px v;
                float c = 0.f;
                // Fast path (the common case): affine map whose 1/32-px coordinates stay far inside the range where
                // OpenCV's INT clamp and short saturation are no-ops.
                const bool small = fast_ok && __builtin_fabs(Xn * xf->wq) < 1.0e6 && __builtin_fabs(Yn * xf->wq) < 1.0e6;
                if (small) {
                    const int X = round_small(Xn * xf->wq), Y = round_small(Yn * xf->wq);
                    v = sample_q5<INTERP>(S, a.sh, a.sw, X, Y, a.b0, a.b1, a.b2, cub_tab);
                    if (WITH_MASK) {
                        const int nx = round_small(Xn * xf->wn), ny = round_small(Yn * xf->wn);
                        c = ((unsigned)nx < (unsigned)a.sw && (unsigned)ny < (unsigned)a.sh) ? 1.f : 0.f;
                    }
                } else {
                    double Wq, Wn;
                    if (affine) { Wq = xf->wq; Wn = xf->wn; }
                    else {
                        // one fp64 division serves both: 32/W == 32 * (1/W) bit for bit (scaling a correctly rounded
                        // quotient by a power of two is exact)
                        const double W = W0 + m6 * dx1;
                        Wn = (W != 0.0) ? 1.0 / W : 0.0;
                        Wq = 32.0 * Wn;
                    }
                    if (SUBPIX == VSTAB_SUBPIX_EXACT && INTERP == VSTAB_INTERP_BILINEAR) {
                        const float w = x * mf[6] + y * mf[7] + mf[8];
                        const float fsx = (x * mf[0] + y * mf[1] + mf[2]) / w;
                        const float fsy = (x * mf[3] + y * mf[4] + mf[5]) / w;
                        v = sample_exact(S, a.sh, a.sw, fsx, fsy, a.b0, a.b1, a.b2);
                    } else {
                        const int X = clamp_round_i32(Xn * Wq);
                        const int Y = clamp_round_i32(Yn * Wq);
                        v = sample_q5<INTERP>(S, a.sh, a.sw, X, Y, a.b0, a.b1, a.b2, cub_tab);
                    }
                    if (WITH_MASK) {
                        const int nx = sat_short(clamp_round_i32(Xn * Wn));
                        const int ny = sat_short(clamp_round_i32(Yn * Wn));
                        c = ((unsigned)nx < (unsigned)a.sw && (unsigned)ny < (unsigned)a.sh) ? 1.f : 0.f;
                    }
                }
                acc[p][0] = v.r; acc[p][1] = v.g; acc[p][2] = v.b;
                if (WITH_MASK) cov[p] = c;
            }
        }
    }

    // ---- epilogue ----
    float mk[TILE_PX];
    unsigned padded = 0;
#pragma unroll
    for (int p = 0; p < TILE_PX; p++) {
        float m = 1.0f - cov[p];
        mk[p] = (m < 1e-3f) ? 0.f : m;
        if (WITH_MASK && p < npx) padded += (mk[p] > 0.5f) ? 1u : 0u;
    }

    if (active) {
        // per-frame bases are uniform (scalar registers); inside a frame 32-bit element offsets suffice (the host checks
        // that a frame has fewer than 2^30 pixels), which keeps the address arithmetic out of 64-bit VALU multiplies
        float* __restrict__ D = a.dst + (size_t)frame * a.dh * a.dw * 3;
        float* __restrict__ Mk = WITH_MASK ? a.mask + (size_t)frame * a.dh * a.dw : nullptr;
        const unsigned row = (unsigned)y * (unsigned)a.dw;
        typedef float f3 __attribute__((ext_vector_type(3)));
#pragma unroll
        for (int p = 0; p < TILE_PX; p++) {
            if (p >= npx) continue;
            const unsigned pix = row + (unsigned)(x0 + p * TILE_TX);
            // 12 B per lane, consecutive lanes -> consecutive pixels: every store instruction writes whole lines
            f3 rgb = {acc[p][0], acc[p][1], acc[p][2]};
            __builtin_memcpy(D + pix * 3u, &rgb, 12);
            if (WITH_MASK) Mk[pix] = mk[p];
        }
    }

    if (WITH_MASK && a.pad_count != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) padded += __shfl_down(padded, off);
        if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = padded;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            if (total) atomicAdd(a.pad_count + frame, total);
        }
    }
}

// ---- motion blur: S samples per output pixel (motion_apply.py:137-202) ---------------------------------------------
//
// What bounds the S-sample loop (profiles/r03_blur_kernel.md): a bicubic sample reads 16 taps x 12 B = 192 B per output
// pixel; served by the CU's vector L1 at 64 B per clock that is 3 clocks per pixel-sample -- 12.6 ms for 64 x 1080p x 17
// samples at 2.2 GHz, which is what round 2's kernel took (12.6 ms) and what every rearrangement of its arithmetic took
// too (fewer bounds tests, shared block-origin terms, product table, scalar-base addressing: 12.85-12.88 ms each).  The
// VALU work (176 instructions per pixel-sample, 84 % issue) was only just hidden behind it.  The loop is therefore fed
// from LDS (256 B per clock) instead:
//
//   * A block owns a 64 x TILE_H output tile for ALL samples.  Once per block, the source coordinates of the tile's four
//     corners are evaluated for every sample with the kernel's own arithmetic (4 x S threads).  For affine samples the
//     per-pixel evaluation is monotone in x and in y (a chain of correctly rounded operations, same column-block origin
//     for the whole tile), so the corner values bound every pixel's: their bounding box + the tap footprint is the
//     source window the tile can touch.  If all samples are affine, the window lies inside the source and fits the LDS
//     budget, the block stages it (each texel read from memory once, padded to a float4) and runs the STAGED loop;
//     otherwise (border tiles, perspective samples, very fast motion) it runs the general loop (sample_q5), which is
//     the arithmetic definition.  Both give the oracle's bits (tests/test_warp_gpu.py, tests/test_configs_gpu.py).
//   * The staged loop has no bounds tests, no short saturation (no-ops in the interior) and no coverage arithmetic
//     (all S samples covered: mask = 1 - S/S = 0, the float the general loop produces); the f64 block-origin terms
//     X0, Y0 are formed once per thread and sample, not per pixel; taps are 16-byte LDS reads at a 16-byte lane stride.
//   * bilinear: the four tap weights are formed in registers by the expressions of OpenCV's 32 x 32 product table
//     (initInterTab2D; bilinear_weights above -- round 3 read an LDS copy of the table: 35 % bank conflicts); bicubic keeps
//     the 1-D table in LDS + 16 products per sample: its 64 KB product table would halve the blocks per CU for 16 of ~140
//     instructions.
// The four bilinear tap weights of a 1/32-px fraction pair, formed in registers by the expressions that fill OpenCV's 32 x 32
// table of products (initInterTab2D) -- the same float32 products.  Round 3 read them from a copy of that table in LDS: its
// random 16-B reads were 35 % bank conflicts of a loop that keeps the LDS array 73 % busy; ten VALU instructions instead:
// 16 x 4K x 33 samples 5.87 -> 5.55 ms (profiles/r04_blur_kernel.md).
typedef float blur_f4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ blur_f4_t bilinear_weights(int fx, int fy)
{
    const float wx1 = fx * (1.f / 32), wx0 = 1.f - wx1;
    const float wy1 = fy * (1.f / 32), wy0 = 1.f - wy1;
    return blur_f4_t{wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};
}

template <int INTERP, int SUBPIX>
struct BlurGeom {
    // the staged loops exist for every sampler; `exact` (the OpenCV >= 4.11 bilinear form: float32 coordinates, lerp
    // arithmetic) runs the border-capable one for all of its staged tiles
    static constexpr bool FAST = true;
    static constexpr bool EXACT = SUBPIX == VSTAB_SUBPIX_EXACT;
    static constexpr bool BICUBIC = INTERP == VSTAB_INTERP_BICUBIC;
    static constexpr int NT = 512;                                    // 64 x 16 output pixels
    static constexpr int FOOT_TEXELS = !FAST ? 0 : (BICUBIC ? 4864 : 2304);   // staged source window (float4 per texel)
    static constexpr int TAPS = BICUBIC ? 4 : 2, LEAD = BICUBIC ? 1 : 0;      // taps per axis, taps left of / above (sx, sy)
    // LDS per block: bicubic 76 KB + 0.5 KB (two blocks per CU), bilinear 36 + 0.5 KB (four)
    static constexpr size_t LDS_BYTES = sizeof(float) * (32 * 4 + (size_t)FOOT_TEXELS * 4);
};

// amdgpu_waves_per_eu(4): at most 128 VGPRs, so that two 512-thread bicubic blocks share a CU.
// `xfs` is a kernel argument of its own (not the WarpArgs member): a `const T* __restrict__` PARAMETER is what lets the
// compiler read the per-sample matrices with scalar loads (s_load into SGPRs) -- through the by-value struct it used
// per-lane vector loads, a full memory round trip at the head of every sample's dependent chain.
template <int INTERP, int SUBPIX, bool WITH_MASK>
__global__ __launch_bounds__((BlurGeom<INTERP, SUBPIX>::NT)) __attribute__((amdgpu_waves_per_eu(4))) void warp_blur_kernel(WarpArgs a, const WarpXform* __restrict__ xfs)
{
    using G = BlurGeom<INTERP, SUBPIX>;
    constexpr int NT = G::NT, TILE_TX = 32, TILE_W = TILE_TX * TILE_PX, TILE_H = NT / TILE_TX;
    typedef float f4_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float s_mem[];   // [1-D cubic table][staged window]
    float* s_cub = s_mem;
    f4_t* s_foot = reinterpret_cast<f4_t*>(s_mem + 32 * 4);
    __shared__ int s_box[4];   // min sx, min sy, max sx, max sy over corners x samples
    const float* cub_tab = nullptr;
    if (INTERP == VSTAB_INTERP_BICUBIC) {
        if (threadIdx.x < 32) cubic_coeffs((int)threadIdx.x, s_cub + threadIdx.x * 4);
        cub_tab = s_cub;
    }
    if (threadIdx.x == 0) { s_box[0] = s_box[1] = 0x7fffffff; s_box[2] = s_box[3] = (int)0x80000000; }
    __syncthreads();

    const unsigned t = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned tiles_per_frame = (unsigned)a.tiles_x * a.tiles_y;
    const int frame = t / tiles_per_frame;
    const unsigned tr = t - frame * tiles_per_frame;
    const int tile_y = tr / a.tiles_x, tile_x = tr - tile_y * a.tiles_x;
    const int tx = threadIdx.x % TILE_TX, ty = threadIdx.x / TILE_TX;
    const int x0 = tile_x * TILE_W + tx;
    const int y = tile_y * TILE_H + ty;
    const bool active = (y < a.dh) && (x0 < a.dw);
    const int npx = active ? (a.dw - x0 + TILE_TX - 1) / TILE_TX : 0;
    const float* __restrict__ S = a.src + (size_t)frame * a.sh * a.sw * 3;
    const int nxf = a.nxf_per_frame;
    const WarpXform* __restrict__ xf0 = xfs + (size_t)frame * nxf;

    auto block_origin = [&](int x) {
        if (a.bw0 >= a.dw) return 0;
        if (a.bw0_pow2) return x & ~(a.bw0 - 1);
        return (x / a.bw0) * a.bw0;
    };

    // ---- the source window this tile can touch, over all samples
    bool fast = G::FAST && a.blur_fast != 0;
    bool inside = false;   // the staged window lies inside the source: no tap of the tile meets the border
    bool inside_persp = false;   // ... and some sample of the frame has a perspective row (its own copy of the interior loop)
    int ox = 0, oy = 0, fw = 0;
    if (fast) {
        bool ok = true, any_persp = false;
        int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = (int)0x80000000, by1 = (int)0x80000000;
        if ((int)threadIdx.x < 4 * nxf) {
            const int k = threadIdx.x >> 2, c = threadIdx.x & 3;
            const WarpXform* __restrict__ xf = xf0 + k;
            const int cx_ = (c & 1) ? min(tile_x * TILE_W + TILE_W, a.dw) - 1 : tile_x * TILE_W;
            const int cy_ = (c & 2) ? min(tile_y * TILE_H + TILE_H, a.dh) - 1 : tile_y * TILE_H;
            const int xb = block_origin(cx_);
            const double dxb = (double)xb, dyc = (double)cy_, dx1 = (double)(cx_ - xb);
            const double Xn = (xf->m[0] * dxb + xf->m[1] * dyc + xf->m[2]) + xf->m[0] * dx1;
            const double Yn = (xf->m[3] * dxb + xf->m[4] * dyc + xf->m[5]) + xf->m[3] * dx1;
            // A sample with a perspective row (Flow in perspective mode -> Motion Apply: BASELINE C3) maps the tile onto a
            // convex quadrilateral as long as its denominator W = m6 x + m7 y + m8 keeps one sign over the tile -- W is
            // affine in (x, y), so that is a test of the four corners -- and a convex quadrilateral lies inside the
            // bounding box of its corners.  The kernel's rounded coordinates can leave that box by one 1/32-px unit at
            // most (fp64 evaluation error ~1e-10 of a unit), which the window's one-texel margin absorbs.
            double wq_c = xf->wq;
            bool w_ok = true;
            if (!xf->affine) {
                const double W = (xf->m[6] * dxb + xf->m[7] * dyc + xf->m[8]) + xf->m[6] * dx1;
                const bool pos = W > 0.0;
                w_ok = !G::EXACT && W != 0.0 && W == W && (__shfl_xor((int)pos, 1) == (int)pos) && (__shfl_xor((int)pos, 2) == (int)pos);
                wq_c = 32.0 * ((W != 0.0) ? 1.0 / W : 0.0);
                any_persp = true;
            }
            const double Xq = Xn * wq_c, Yq = Yn * wq_c;
            ok = w_ok && __builtin_fabs(Xq) < 9.0e5 && __builtin_fabs(Yq) < 9.0e5;   // NaN compares false
            if (ok) { bx0 = bx1 = round_small(Xq) >> 5; by0 = by1 = round_small(Yq) >> 5; }
            if (G::EXACT && ok) {
                // the exact sampler's own source position of this corner (float32 chain, monotone in x and in y for an
                // affine sample: every operation is correctly rounded and the divisor is the constant m8)
                float mf[9];
#pragma unroll
                for (int i = 0; i < 9; i++) mf[i] = (float)xf->m[i];
                const float w = cx_ * mf[6] + cy_ * mf[7] + mf[8];
                const float fsx = (cx_ * mf[0] + cy_ * mf[1] + mf[2]) / w, fsy = (cx_ * mf[3] + cy_ * mf[4] + mf[5]) / w;
                ok = __builtin_fabsf(fsx) < 28000.f && __builtin_fabsf(fsy) < 28000.f;
                if (ok) { bx0 = bx1 = (int)__builtin_floorf(fsx); by0 = by1 = (int)__builtin_floorf(fsy); }
                else { bx0 = by0 = 0x7fffffff; bx1 = by1 = (int)0x80000000; }
            }
        }
        // bounding box: butterfly over the wavefront (idle lanes hold the neutral elements), one LDS atomic set per wavefront
        if ((int)(threadIdx.x & ~63u) < 4 * nxf) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                bx0 = min(bx0, __shfl_xor(bx0, off)); by0 = min(by0, __shfl_xor(by0, off));
                bx1 = max(bx1, __shfl_xor(bx1, off)); by1 = max(by1, __shfl_xor(by1, off));
            }
            if ((threadIdx.x & 63) == 0) {
                atomicMin(&s_box[0], bx0); atomicMin(&s_box[1], by0);
                atomicMax(&s_box[2], bx1); atomicMax(&s_box[3], by1);
            }
        }
        fast = __syncthreads_and(ok) != 0;
        const bool persp = __syncthreads_or(any_persp) != 0;   // uniform: some sample of this frame has a perspective row
        if (fast) {
            // taps of (sx, sy): columns sx - LEAD .. sx - LEAD + TAPS - 1; one texel of margin on every side
            ox = s_box[0] - G::LEAD - 1;
            oy = s_box[1] - G::LEAD - 1;
            fw = s_box[2] - s_box[0] + G::TAPS + 2;
            const int fh = s_box[3] - s_box[1] + G::TAPS + 2;
            fast = fw * fh <= G::FOOT_TEXELS;   // uniform
            // (the exact sampler takes the border-capable loop, which carries its coordinate form; a window that lies inside
            // the source serves perspective samples through the interior loop too since round 5: the window bounds every
            // tap of every pixel-sample, so no tap meets the border and the nearest-neighbour coverage pixel -- one of the
            // two bilinear columns / rows around the sample -- lies inside as well)
            inside = !G::EXACT && ox >= 0 && oy >= 0 && ox + fw <= a.sw && oy + fh <= a.sh;
            inside_persp = inside && persp;
            if (fast) {
                // stage the window: one texel (12 contiguous bytes) per lane and step -> whole cache lines per row.  A
                // window that leaves the source (the ring of tiles along the content's edge) is filled with the border
                // colour there: BORDER_CONSTANT replaces a tap outside the source by the border value, tap by tap.
                typedef float f3_t __attribute__((ext_vector_type(3), aligned(4)));
                for (int i = threadIdx.x; i < fw * fh; i += NT) {
                    const int r = i / fw, c = i - r * fw;
                    const int gy = oy + r, gx = ox + c;
                    f4_t e = f4_t{a.b0, a.b1, a.b2, 0.f};
                    if ((unsigned)gx < (unsigned)a.sw && (unsigned)gy < (unsigned)a.sh) {
                        const f3_t v = *reinterpret_cast<const f3_t*>(S + ((unsigned)gy * (unsigned)a.sw + (unsigned)gx) * 3u);
                        e = f4_t{v.x, v.y, v.z, 0.f};
                    }
                    s_foot[i] = e;
                }
            }
            __syncthreads();
        }
    }

    float acc[TILE_PX][3];
    float cov[TILE_PX];
#pragma unroll
    for (int p = 0; p < TILE_PX; p++) { acc[p][0] = acc[p][1] = acc[p][2] = 0.f; cov[p] = 0.f; }

    if (active && fast && !inside) {
        // ---- the STAGED loop for a tile whose window leaves the source: the taps still come from LDS (border-filled),
        // and what the interior loop may leave out is put back per pixel-sample -- the three cases of cv::remap with
        // BORDER_CONSTANT (all taps inside: the plain sums; no tap inside: the border value itself, not a weighted sum of
        // sixteen copies of it; otherwise bilinear: the same sums over the border-filled taps, bicubic: OpenCV's separate
        // formula border + sum over the VALID taps of (tap - border) * weight), and the nearest-neighbour coverage test.
        // Round 3 ran these tiles (the ring along the content's edge: ~10 % of a crop_and_pad clip's tiles) through the
        // general loop, at the L1-bound rate of round 2.
        const int xb = block_origin(x0);
        const double dy = (double)y, dxb = (double)xb;
        double dx1[TILE_PX];
#pragma unroll
        for (int p = 0; p < TILE_PX; p++) dx1[p] = (double)((p < npx ? x0 + p * TILE_TX : x0) - xb);
        for (int k = 0; k < nxf; k++) {
            const WarpXform* __restrict__ xf = xf0 + k;
            const double m0 = xf->m[0], m3 = xf->m[3], m6 = xf->m[6];
            const bool aff = xf->affine != 0;
            const double X0 = m0 * dxb + xf->m[1] * dy + xf->m[2];
            const double Y0 = m3 * dxb + xf->m[4] * dy + xf->m[5];
            const double W0 = m6 * dxb + xf->m[7] * dy + xf->m[8];
            float mf[9];
            if (G::EXACT) {
#pragma unroll
                for (int i = 0; i < 9; i++) mf[i] = (float)xf->m[i];
            }
#pragma unroll
            for (int p = 0; p < TILE_PX; p++) {
                const double Xn = X0 + m0 * dx1[p], Yn = Y0 + m3 * dx1[p];
                double wq = xf->wq, wn = xf->wn;
                if (!aff) {   // the general loop's per-pixel denominator (uniform branch: a property of the sample)
                    const double W = W0 + m6 * dx1[p];
                    wn = (W != 0.0) ? 1.0 / W : 0.0;
                    wq = 32.0 * wn;
                }
                const int X = round_small(Xn * wq), Y = round_small(Yn * wq);
                int sx = X >> 5, sy = Y >> 5;           // |sx|, |sy| < 2^15 (corner test): sat_short is the identity
                const int fx = X & 31, fy = Y & 31;
                float ax = 0.f, ay = 0.f;
                if (G::EXACT) {   // sample_exact's coordinates: float32, no 1/32-px quantisation
                    const int x = p < npx ? x0 + p * TILE_TX : x0;
                    const float w = x * mf[6] + y * mf[7] + mf[8];
                    const float fsx = (x * mf[0] + y * mf[1] + mf[2]) / w, fsy = (x * mf[3] + y * mf[4] + mf[5]) / w;
                    sx = (int)__builtin_floorf(fsx); sy = (int)__builtin_floorf(fsy);
                    ax = fsx - sx; ay = fsy - sy;
                }
                const f4_t* __restrict__ T = s_foot + ((int)__umul24((unsigned)(sy - oy - G::LEAD), (unsigned)fw) + (sx - ox - G::LEAD));
                float vr, vg, vb;
                if (G::EXACT) {
                    const f4_t p00 = T[0], p01 = T[1], p10 = T[fw], p11 = T[fw + 1];
                    const bool none = sx >= a.sw || sx + 1 < 0 || sy >= a.sh || sy + 1 < 0;
                    float v0, v1;
                    v0 = p00.x + ax * (p01.x - p00.x); v1 = p10.x + ax * (p11.x - p10.x); vr = v0 + ay * (v1 - v0);
                    v0 = p00.y + ax * (p01.y - p00.y); v1 = p10.y + ax * (p11.y - p10.y); vg = v0 + ay * (v1 - v0);
                    v0 = p00.z + ax * (p01.z - p00.z); v1 = p10.z + ax * (p11.z - p10.z); vb = v0 + ay * (v1 - v0);
                    if (none) { vr = a.b0; vg = a.b1; vb = a.b2; }
                } else if (INTERP == VSTAB_INTERP_BICUBIC) {
                    const f4_t cxv = reinterpret_cast<const f4_t*>(s_cub)[fx], cyv = reinterpret_cast<const f4_t*>(s_cub)[fy];
                    const float cx[4] = {cxv.x, cxv.y, cxv.z, cxv.w}, cy[4] = {cyv.x, cyv.y, cyv.z, cyv.w};
                    const int bx0 = sx - 1, by0 = sy - 1;
                    const unsigned width1 = (unsigned)(a.sw - 3 > 0 ? a.sw - 3 : 0), height1 = (unsigned)(a.sh - 3 > 0 ? a.sh - 3 : 0);
                    if ((unsigned)bx0 < width1 && (unsigned)by0 < height1) {
                        float sr = 0.f, sg = 0.f, sb = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const f4_t t0 = T[i * fw], t1 = T[i * fw + 1], t2 = T[i * fw + 2], t3 = T[i * fw + 3];
                            const float w0 = cy[i] * cx[0], w1 = cy[i] * cx[1], w2 = cy[i] * cx[2], w3 = cy[i] * cx[3];
                            const float tr_ = t0.x * w0 + t1.x * w1 + t2.x * w2 + t3.x * w3;
                            const float tg_ = t0.y * w0 + t1.y * w1 + t2.y * w2 + t3.y * w3;
                            const float tb_ = t0.z * w0 + t1.z * w1 + t2.z * w2 + t3.z * w3;
                            if (i == 0) { sr = tr_; sg = tg_; sb = tb_; }
                            else { sr += tr_; sg += tg_; sb += tb_; }
                        }
                        vr = sr; vg = sg; vb = sb;
                    } else if (bx0 >= a.sw || bx0 + 4 <= 0 || by0 >= a.sh || by0 + 4 <= 0) {
                        vr = a.b0; vg = a.b1; vb = a.b2;
                    } else {
                        float sr = a.b0, sg = a.b1, sb = a.b2;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int yy = by0 + i;
                            if (yy < 0 || yy >= a.sh) continue;
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int xx = bx0 + j;
                                if (xx < 0 || xx >= a.sw) continue;
                                const f4_t v = T[i * fw + j];
                                const float w = cy[i] * cx[j];
                                sr += (v.x - a.b0) * w;
                                sg += (v.y - a.b1) * w;
                                sb += (v.z - a.b2) * w;
                            }
                        }
                        vr = sr; vg = sg; vb = sb;
                    }
                } else {
                    const f4_t t00 = T[0], t01 = T[1], t10 = T[fw], t11 = T[fw + 1];
                    const f4_t w = bilinear_weights(fx, fy);
                    const bool none = sx >= a.sw || sx + 1 < 0 || sy >= a.sh || sy + 1 < 0;
                    vr = t00.x * w.x + t01.x * w.y + t10.x * w.z + t11.x * w.w;
                    vg = t00.y * w.x + t01.y * w.y + t10.y * w.z + t11.y * w.w;
                    vb = t00.z * w.x + t01.z * w.y + t10.z * w.z + t11.z * w.w;
                    if (none) { vr = a.b0; vg = a.b1; vb = a.b2; }
                }
                acc[p][0] += vr; acc[p][1] += vg; acc[p][2] += vb;
                if (WITH_MASK) {
                    const int nx = round_small(Xn * wn), ny = round_small(Yn * wn);
                    cov[p] += ((unsigned)nx < (unsigned)a.sw && (unsigned)ny < (unsigned)a.sh) ? 1.f : 0.f;
                }
                if (INTERP == VSTAB_INTERP_BICUBIC) __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (active && fast) {
        // ---- the INTERIOR loop.  Two copies, chosen per block: frames whose samples are all affine keep the loop of rounds
        // 3-4 untouched (it sits at the 128-VGPR limit); a frame with a perspective sample (Flow in perspective mode ->
        // Motion Apply: BASELINE C3) takes the copy that forms the general loop's per-pixel denominator -- 1 / W in fp64 per
        // pixel-sample is the contract, W's tile-uniform part W0 is formed once per thread and sample like X0 / Y0 -- and
        // nothing else of the border-capable loop: no bounds tests, no per-tap validity, no coverage arithmetic
        // (C3's blur warp: see profiles/r05_c3_chain.md).
        const int xb = block_origin(x0);                 // == block_origin(x0 + TILE_TX): blur_fast
        const double dy = (double)y, dxb = (double)xb;
        // a pixel of this thread beyond the right edge (ragged last tile) re-evaluates pixel 0 instead of being skipped:
        // no divergent branch inside the sample loop, its sum is never stored
        double dx1[TILE_PX];
#pragma unroll
        for (int p = 0; p < TILE_PX; p++) dx1[p] = (double)((p < npx ? x0 + p * TILE_TX : x0) - xb);
        const int tap0 = (oy + G::LEAD) * fw + ox + G::LEAD;   // window index of tap (0,0) = (sy * fw + sx) - tap0
        auto interior = [&](auto persp_tag) {
            constexpr bool PERSP = decltype(persp_tag)::value;
            for (int k = 0; k < nxf; k++) {
                const WarpXform* __restrict__ xf = xf0 + k;
                const double m0 = xf->m[0], m3 = xf->m[3];
                const double X0 = m0 * dxb + xf->m[1] * dy + xf->m[2];
                const double Y0 = m3 * dxb + xf->m[4] * dy + xf->m[5];
                const bool aff = !PERSP || xf->affine != 0;            // uniform: a property of the sample
                const double m6 = PERSP ? xf->m[6] : 0.0;
                const double W0 = PERSP ? m6 * dxb + xf->m[7] * dy + xf->m[8] : 0.0;
#pragma unroll
                for (int p = 0; p < TILE_PX; p++) {
                    const double Xn = X0 + m0 * dx1[p], Yn = Y0 + m3 * dx1[p];
                    double wq = xf->wq;
                    if (PERSP && !aff) {
                        const double W = W0 + m6 * dx1[p];
                        wq = 32.0 * ((W != 0.0) ? 1.0 / W : 0.0);
                    }
                    const int X = round_small(Xn * wq), Y = round_small(Yn * wq);
                    const int sx = X >> 5, sy = Y >> 5;
                    const int fx = X & 31, fy = Y & 31;
                    const f4_t* __restrict__ T = s_foot + ((int)__umul24((unsigned)sy, (unsigned)fw) + sx - tap0);
                    if (INTERP == VSTAB_INTERP_BICUBIC) {
                        const f4_t cxv = reinterpret_cast<const f4_t*>(s_cub)[fx], cyv = reinterpret_cast<const f4_t*>(s_cub)[fy];
                        const float cx[4] = {cxv.x, cxv.y, cxv.z, cxv.w}, cy[4] = {cyv.x, cyv.y, cyv.z, cyv.w};
                        float sr = 0.f, sg = 0.f, sb = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const f4_t t0 = T[i * fw], t1 = T[i * fw + 1], t2 = T[i * fw + 2], t3 = T[i * fw + 3];
                            const float w0 = cy[i] * cx[0], w1 = cy[i] * cx[1], w2 = cy[i] * cx[2], w3 = cy[i] * cx[3];
                            const float tr_ = t0.x * w0 + t1.x * w1 + t2.x * w2 + t3.x * w3;
                            const float tg_ = t0.y * w0 + t1.y * w1 + t2.y * w2 + t3.y * w3;
                            const float tb_ = t0.z * w0 + t1.z * w1 + t2.z * w2 + t3.z * w3;
                            if (i == 0) { sr = tr_; sg = tg_; sb = tb_; }
                            else { sr += tr_; sg += tg_; sb += tb_; }
                        }
                        acc[p][0] += sr; acc[p][1] += sg; acc[p][2] += sb;
                        __builtin_amdgcn_sched_barrier(0);   // one pixel's 16 taps at a time: both in flight need > 128 VGPRs (spills)
                    } else {
                        const f4_t t00 = T[0], t01 = T[1], t10 = T[fw], t11 = T[fw + 1];
                        const f4_t w = bilinear_weights(fx, fy);
                        acc[p][0] += t00.x * w.x + t01.x * w.y + t10.x * w.z + t11.x * w.w;
                        acc[p][1] += t00.y * w.x + t01.y * w.y + t10.y * w.z + t11.y * w.w;
                        acc[p][2] += t00.z * w.x + t01.z * w.y + t10.z * w.z + t11.z * w.w;
                    }
                }
            }
        };
        if (inside_persp) interior(std::true_type{});
        else interior(std::false_type{});
        if (WITH_MASK) {
#pragma unroll
            for (int p = 0; p < TILE_PX; p++) cov[p] = (float)nxf;   // every sample covered: 1.f added nxf times
        }
    } else if (active) {
        const double dy = (double)y;
        for (int k = 0; k < nxf; k++) {
            const WarpXform* __restrict__ xf = xf0 + k;
            const double m0 = xf->m[0], m1 = xf->m[1], m2 = xf->m[2];
            const double m3 = xf->m[3], m4 = xf->m[4], m5 = xf->m[5];
            const double m6 = xf->m[6], m7 = xf->m[7], m8 = xf->m[8];
            const bool affine = xf->affine != 0;
            float mf[9];
            if (SUBPIX == VSTAB_SUBPIX_EXACT && INTERP == VSTAB_INTERP_BILINEAR) {
#pragma unroll
                for (int i = 0; i < 9; i++) mf[i] = (float)xf->m[i];
            }
#pragma unroll
            for (int p = 0; p < TILE_PX; p++) {
                if (p >= npx) continue;
                const int x = x0 + p * TILE_TX;
                const int xb = block_origin(x);
                const double dxb = (double)xb;
                const double X0 = m0 * dxb + m1 * dy + m2;
                const double Y0 = m3 * dxb + m4 * dy + m5;
                const double W0 = m6 * dxb + m7 * dy + m8;
                const double dx1 = (double)(x - xb);
                const double Xn = X0 + m0 * dx1, Yn = Y0 + m3 * dx1;
                double Wq, Wn;
                if (affine) { Wq = xf->wq; Wn = xf->wn; }
                else {
                    const double W = W0 + m6 * dx1;
                    Wn = (W != 0.0) ? 1.0 / W : 0.0;
                    Wq = 32.0 * Wn;
                }
                Px v;
                if (SUBPIX == VSTAB_SUBPIX_EXACT && INTERP == VSTAB_INTERP_BILINEAR) {
                    const float w = x * mf[6] + y * mf[7] + mf[8];
                    const float fsx = (x * mf[0] + y * mf[1] + mf[2]) / w;
                    const float fsy = (x * mf[3] + y * mf[4] + mf[5]) / w;
                    v = sample_exact(S, a.sh, a.sw, fsx, fsy, a.b0, a.b1, a.b2);
                } else {
                    const int X = clamp_round_i32(Xn * Wq);
                    const int Y = clamp_round_i32(Yn * Wq);
                    v = sample_q5<INTERP>(S, a.sh, a.sw, X, Y, a.b0, a.b1, a.b2, cub_tab);
                }
                acc[p][0] += v.r; acc[p][1] += v.g; acc[p][2] += v.b;
                if (WITH_MASK) {
                    const int nx = sat_short(clamp_round_i32(Xn * Wn));
                    const int ny = sat_short(clamp_round_i32(Yn * Wn));
                    cov[p] += ((unsigned)nx < (unsigned)a.sw && (unsigned)ny < (unsigned)a.sh) ? 1.f : 0.f;
                }
            }
        }
    }

    if (active) {
        const float fs = (float)a.samples;   // the reference divides by the sample count even when a 1-frame clip yields one sample
        float* __restrict__ D = a.dst + (size_t)frame * a.dh * a.dw * 3;
        float* __restrict__ Mk = WITH_MASK ? a.mask + (size_t)frame * a.dh * a.dw : nullptr;
        const unsigned row = (unsigned)y * (unsigned)a.dw;
        typedef float f3 __attribute__((ext_vector_type(3)));
#pragma unroll
        for (int p = 0; p < TILE_PX; p++) {
            if (p >= npx) continue;
            const unsigned pix = row + (unsigned)(x0 + p * TILE_TX);
            f3 rgb = {acc[p][0] / fs, acc[p][1] / fs, acc[p][2] / fs};
            __builtin_memcpy(D + pix * 3u, &rgb, 12);
            if (WITH_MASK) {
                const float m = 1.0f - cov[p] / fs;
                Mk[pix] = (m < 1e-3f) ? 0.f : m;
            }
        }
    }
}

template <int INTERP, int SUBPIX>
int launch_blur(WarpArgs a, bool with_mask, hipStream_t st)
{
    using G = BlurGeom<INTERP, SUBPIX>;
    constexpr int TILE_W = 32 * TILE_PX, TILE_H = G::NT / 32;
    a.tiles_x = (a.dw + TILE_W - 1) / TILE_W;
    a.tiles_y = (a.dh + TILE_H - 1) / TILE_H;
    const unsigned long long blocks = (unsigned long long)a.tiles_x * a.tiles_y * a.n;
    VSTAB_REQUIRE(blocks > 0 && blocks < 0x7fffffffULL, "warp: grid of %llu blocks is out of range", blocks);
    // staged-path precondition that does not depend on the block: a thread's two pixels share OpenCV's column block (the
    // u24 multiply of the window index is exact: check_common has sh, sw <= 32767 and the window is at most 4864 texels)
    a.blur_fast = ((a.bw0 >= a.dw) || (a.bw0 % TILE_W == 0)) ? 1 : 0;
    // the prologue (corner classification, staging, two barriers) is amortised over the samples: measured break-even
    // S = 4 for bilinear (S = 3: 2.00 vs 1.86 ms per 64 x 1080p; S = 5: 2.38 vs 2.50), below 3 for bicubic
    if (INTERP == VSTAB_INTERP_BILINEAR && a.nxf_per_frame < 4) a.blur_fast = 0;
    // 0: general loop everywhere (tests toggle it inside one process, so it is read per launch: ~0.1 us against a kernel
    // of milliseconds)
    if (const char* e = getenv("VSTAB_BLUR_FAST")) a.blur_fast = a.blur_fast && atoi(e) != 0;
    // The staged window needs more than the default 64 KB of dynamic LDS for bicubic (76.5 KB): asked for ONCE per kernel
    // instance AND DEVICE (hipFuncSetAttribute applies to the current device's function object, and one process may hold a
    // context per GPU); a device (or runtime) that refuses it gets the general loop everywhere with the small allocation
    // instead of an error -- the staged path is an optimisation, the general loop is the definition.
    constexpr int MAX_DEV = 64;
    static signed char opt_in[2][MAX_DEV] = {};   // [with_mask][device]: 0 not asked yet, 1 granted, -1 refused
    int dev = 0;
    VSTAB_HIP(hipGetDevice(&dev));
    bool big_lds = G::LDS_BYTES <= 64 * 1024;
    if (!big_lds) {
        signed char* slot = (dev >= 0 && dev < MAX_DEV) ? &opt_in[with_mask ? 1 : 0][dev] : nullptr;
        signed char state = slot ? __atomic_load_n(slot, __ATOMIC_RELAXED) : 0;
        if (state == 0) {
            const void* fn = with_mask ? reinterpret_cast<const void*>(warp_blur_kernel<INTERP, SUBPIX, true>)
                                       : reinterpret_cast<const void*>(warp_blur_kernel<INTERP, SUBPIX, false>);
            state = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) == hipSuccess ? 1 : -1;
            if (slot) __atomic_store_n(slot, state, __ATOMIC_RELAXED);
        }
        big_lds = state > 0;
    }
    size_t lds = G::LDS_BYTES;
    if (!big_lds) {
        (void)hipGetLastError();
        a.blur_fast = 0;
        lds = sizeof(float) * 32 * 4;
    }
    if (with_mask) hipLaunchKernelGGL((warp_blur_kernel<INTERP, SUBPIX, true>), dim3((unsigned)blocks), dim3(G::NT), lds, st, a, a.xf);
    else hipLaunchKernelGGL((warp_blur_kernel<INTERP, SUBPIX, false>), dim3((unsigned)blocks), dim3(G::NT), lds, st, a, a.xf);
    VSTAB_HIP(hipGetLastError());
    return 0;
}

template <int INTERP, int SUBPIX>
void launch_mask(const WarpArgs& a, bool with_mask, unsigned grid, hipStream_t st, int tx)
{
#define LAUNCH_TX(TX)                                                                                              \
    do {                                                                                                           \
        if (with_mask) hipLaunchKernelGGL((warp_kernel<INTERP, SUBPIX, true, TX>), dim3(grid), dim3(256), 0, st, a);  \
        else hipLaunchKernelGGL((warp_kernel<INTERP, SUBPIX, false, TX>), dim3(grid), dim3(256), 0, st, a);  \
    } while (0)
    if (tx == 16) LAUNCH_TX(16);
    else if (tx == 64) LAUNCH_TX(64);
    else if (tx == 8) LAUNCH_TX(8);
    else LAUNCH_TX(32);
#undef LAUNCH_TX
}

int launch_warp(WarpArgs a, int interp, int subpix, bool with_mask, hipStream_t st)
{
    // threads along x of the 256-thread tile: 32 (64 x 8 px) by default; VSTAB_WARP_TX = 8|16|64 for sweeps
    // (profiles/r01_warp_tile_sweep.md)
    int tx = 32;
    if (const char* e = getenv("VSTAB_WARP_TX")) { const int v = atoi(e); if (v == 8 || v == 16 || v == 64) tx = v; }
    a.tiles_x = (a.dw + tx * TILE_PX - 1) / (tx * TILE_PX);
    a.tiles_y = (a.dh + 256 / tx - 1) / (256 / tx);
    const unsigned long long blocks = (unsigned long long)a.tiles_x * a.tiles_y * a.n;
    VSTAB_REQUIRE(blocks > 0 && blocks < 0x7fffffffULL, "warp: grid of %llu blocks is out of range", blocks);
    const unsigned grid = (unsigned)blocks;
    if (interp == VSTAB_INTERP_BICUBIC) launch_mask<VSTAB_INTERP_BICUBIC, VSTAB_SUBPIX_Q5>(a, with_mask, grid, st, tx);
    else if (subpix == VSTAB_SUBPIX_EXACT) launch_mask<VSTAB_INTERP_BILINEAR, VSTAB_SUBPIX_EXACT>(a, with_mask, grid, st, tx);
    else launch_mask<VSTAB_INTERP_BILINEAR, VSTAB_SUBPIX_Q5>(a, with_mask, grid, st, tx);
    VSTAB_HIP(hipGetLastError());
    return 0;
}

int launch_warp_blur(const WarpArgs& a, int interp, int subpix, bool with_mask, hipStream_t st)
{
    if (interp == VSTAB_INTERP_BICUBIC) return launch_blur<VSTAB_INTERP_BICUBIC, VSTAB_SUBPIX_Q5>(a, with_mask, st);
    if (subpix == VSTAB_SUBPIX_EXACT) return launch_blur<VSTAB_INTERP_BILINEAR, VSTAB_SUBPIX_EXACT>(a, with_mask, st);
    return launch_blur<VSTAB_INTERP_BILINEAR, VSTAB_SUBPIX_Q5>(a, with_mask, st);
}

void fill_xform(const float* m32, WarpXform* xf) { vstab_fill_xform(m32, xf); }

int check_common(const char* who, vstab_ctx* ctx, const void* src, int n, int sh, int sw, const void* mats,
                 int dh, int dw, int interp, const float* border, int subpix, const void* dst)
{
    VSTAB_REQUIRE(ctx != nullptr, "%s: ctx is NULL", who);
    VSTAB_REQUIRE(src && mats && border && dst, "%s: NULL pointer argument", who);
    VSTAB_REQUIRE(n > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0, "%s: non-positive size (n=%d src=%dx%d out=%dx%d)", who, n, sw, sh, dw, dh);
    VSTAB_REQUIRE(sh <= 32767 && sw <= 32767, "%s: source larger than 32767 px is not representable in OpenCV's short maps", who);
    VSTAB_REQUIRE((long long)sh * sw < (1LL << 30) && (long long)dh * dw < (1LL << 30), "%s: frames of 2^30 pixels or more are not supported (32-bit in-frame offsets)", who);
    VSTAB_REQUIRE(interp == VSTAB_INTERP_BILINEAR || interp == VSTAB_INTERP_BICUBIC, "%s: unknown interpolation %d", who, interp);
    VSTAB_REQUIRE(subpix == VSTAB_SUBPIX_Q5 || subpix == VSTAB_SUBPIX_EXACT, "%s: unknown subpix mode %d", who, subpix);
    VSTAB_REQUIRE(!(subpix == VSTAB_SUBPIX_EXACT && interp == VSTAB_INTERP_BICUBIC), "%s: exact sub-pixel mode exists for bilinear only", who);
    return 0;
}

void fill_geometry(WarpArgs& a, int n, int sh, int sw, int dh, int dw, const float* border, const float* dst, const float* mask)
{
    a.n = n; a.sh = sh; a.sw = sw; a.dh = dh; a.dw = dw;
    const int BLOCK_SZ = 32;
    int bh0 = BLOCK_SZ / 2 < dh ? BLOCK_SZ / 2 : dh;
    int bw0 = BLOCK_SZ * BLOCK_SZ / bh0 < dw ? BLOCK_SZ * BLOCK_SZ / bh0 : dw;
    a.bw0 = bw0;
    a.bw0_pow2 = (bw0 & (bw0 - 1)) == 0;
    a.b0 = border[0]; a.b1 = border[1]; a.b2 = border[2];
}

}  // namespace

// ---- the counts' way to the host (see vstab_internal.h) ----
namespace {
__global__ __launch_bounds__(256) void mirror_counts_kernel(const uint32_t* __restrict__ counts, unsigned* host, int n, unsigned* flag, unsigned gen)
{
    for (int i = threadIdx.x; i < n; i += 256) host[i] = counts[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int mirror_counts(vstab_ctx* ctx, const uint32_t* pad_count, int n)
{
    if (ctx->h_counts_cap < n) {
        if (ctx->h_counts) { VSTAB_HIP(hipStreamSynchronize(ctx->stream)); VSTAB_HIP(hipHostFree(ctx->h_counts)); ctx->h_counts = nullptr; ctx->h_counts_cap = 0; }
        const int cap = n > 4096 ? n : 4096;
        void* hp = nullptr;
        VSTAB_HIP(hipHostMalloc(&hp, sizeof(unsigned) * (size_t)cap, hipHostMallocMapped | hipHostMallocCoherent));
        void* dp = nullptr;
        VSTAB_HIP(hipHostGetDevicePointer(&dp, hp, 0));
        ctx->h_counts = static_cast<unsigned*>(hp); ctx->d_counts_mirror = static_cast<unsigned*>(dp); ctx->h_counts_cap = cap;
    }
    ctx->counts_gen += 1;
    ctx->counts_n = n;
    hipLaunchKernelGGL(mirror_counts_kernel, dim3(1), dim3(256), 0, ctx->stream, pad_count, ctx->d_counts_mirror, n,
                       reinterpret_cast<unsigned*>(ctx->d_status) + VSTAB_COUNTS_DONE_WORD, ctx->counts_gen);
    VSTAB_HIP(hipGetLastError());
    return 0;
}
}  // namespace

// The padded-pixel counts of the latest vstab_warp_batch / vstab_warp_batch_planned call that asked for them, on the host:
// waits for that call's kernels only (a word in coherent host memory), nothing is copied by the caller.
extern "C" int vstab_last_pad_counts(vstab_ctx* ctx, int n, uint32_t* out)
{
    VSTAB_REQUIRE(ctx != nullptr && out != nullptr, "vstab_last_pad_counts: NULL argument");
    VSTAB_REQUIRE(ctx->h_counts != nullptr && n == ctx->counts_n, "vstab_last_pad_counts: no warp with counts over %d frames is pending", n);
    volatile unsigned* done = reinterpret_cast<volatile unsigned*>(ctx->h_status) + VSTAB_COUNTS_DONE_WORD;
    const unsigned gen = ctx->counts_gen;
    for (long spins = 0; *done != gen; spins++) {
        if (spins > 200000000L) {   // ~10 s: a warp of a long 4K clip takes tens of milliseconds; then the runtime's own report
            VSTAB_HIP(hipStreamSynchronize(ctx->stream));
            VSTAB_REQUIRE(*done == gen, "vstab_last_pad_counts: the warp finished without reporting its counts");
            break;
        }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    memcpy(out, ctx->h_counts, sizeof(uint32_t) * (size_t)n);
    return 0;
}

extern "C" int vstab_warp_batch(vstab_ctx* ctx, const float* src, int n, int src_h, int src_w, const float* matrices,
                                int out_h, int out_w, int interp, const float* border_rgb, int subpix, float* dst,
                                float* mask, uint32_t* pad_count)
{
    if (int rc = check_common("vstab_warp_batch", ctx, src, n, src_h, src_w, matrices, out_h, out_w, interp, border_rgb, subpix, dst)) return rc;
    VSTAB_HIP(hipSetDevice(ctx->device));
    std::vector<WarpXform> xf((size_t)n);
    for (int i = 0; i < n; i++) fill_xform(matrices + (size_t)i * 9, &xf[i]);
    void* d_xf = nullptr;
    if (vstab_stage_params(ctx, xf.data(), xf.size() * sizeof(WarpXform), &d_xf)) return 1;

    WarpArgs a{};
    a.src = src; a.dst = dst; a.mask = mask; a.pad_count = pad_count;
    a.xf = static_cast<const WarpXform*>(d_xf);
    a.samples = 1; a.nxf_per_frame = 1;
    fill_geometry(a, n, src_h, src_w, out_h, out_w, border_rgb, dst, mask);
    if (pad_count) VSTAB_HIP(hipMemsetAsync(pad_count, 0, sizeof(uint32_t) * (size_t)n, ctx->stream));
    {
        KernelTimer timer(ctx, "warp");
        if (int rc = launch_warp(a, interp, subpix, mask != nullptr, ctx->stream)) return rc;
    }
    return pad_count ? mirror_counts(ctx, pad_count, n) : 0;
}

const WarpXform* vstab_plan_xforms(vstab_ctx* ctx, int first, int n);   // vstab_traj.hip

// vstab_warp_batch for frames [first, first + n) of the clip whose plan vstab_flow_plan_device left on the device: the
// transform table is read where the plan kernel wrote it, nothing crosses the host.
extern "C" int vstab_warp_batch_planned(vstab_ctx* ctx, const float* src, int first, int n, int src_h, int src_w, int out_h,
                                        int out_w, int interp, const float* border_rgb, int subpix, float* dst, float* mask,
                                        uint32_t* pad_count)
{
    if (int rc = check_common("vstab_warp_batch_planned", ctx, src, n, src_h, src_w, border_rgb, out_h, out_w, interp, border_rgb, subpix, dst)) return rc;
    const WarpXform* xf = vstab_plan_xforms(ctx, first, n);
    VSTAB_REQUIRE(xf != nullptr, "vstab_warp_batch_planned: frames [%d, %d) are outside the pending device plan", first, first + n);
    VSTAB_HIP(hipSetDevice(ctx->device));
    WarpArgs a{};
    a.src = src; a.dst = dst; a.mask = mask; a.pad_count = pad_count;
    a.xf = xf;
    a.samples = 1; a.nxf_per_frame = 1;
    fill_geometry(a, n, src_h, src_w, out_h, out_w, border_rgb, dst, mask);
    if (pad_count && pad_count != ctx->plan_zeroed_ptr) VSTAB_HIP(hipMemsetAsync(pad_count, 0, sizeof(uint32_t) * (size_t)n, ctx->stream));
    ctx->plan_zeroed_ptr = nullptr;   // (the plan kernel zeroed a registered array once: a second warp into it fills it itself)
    {
        KernelTimer timer(ctx, "warp");
        if (int rc = launch_warp(a, interp, subpix, mask != nullptr, ctx->stream)) return rc;
    }
    return pad_count ? mirror_counts(ctx, pad_count, n) : 0;
}

// motion_apply.py:125-134 (_blurred_matrix_samples) + the f32 cast of motion_apply.py:172, for frames
// [first, first+count) of a clip of `total` matrices.  Host arithmetic only.
extern "C" int vstab_blur_sample_matrices(const double* matrices, int total, int first, int count, const double* ts,
                                          int samples, float* out)
{
    VSTAB_REQUIRE(matrices && ts && out, "vstab_blur_sample_matrices: NULL pointer argument");
    VSTAB_REQUIRE(total >= 1 && first >= 0 && count >= 1 && first + count <= total,
                  "vstab_blur_sample_matrices: frames [%d, %d) outside a clip of %d matrices", first, first + count, total);
    VSTAB_REQUIRE(samples >= 1 && samples <= MAX_BLUR_SAMPLES, "vstab_blur_sample_matrices: samples=%d outside [1,%d]", samples, MAX_BLUR_SAMPLES);
    // a single-frame clip yields one sample matrix per frame (the caller still divides by `samples`)
    const int per_frame = (total <= 1) ? 1 : samples;
    for (int c = 0; c < count; c++) {
        const int i = first + c;
        const double* base = matrices + (size_t)i * 9;
        double delta[9];
        if (total > 1) {
            if (i < total - 1) for (int j = 0; j < 9; j++) delta[j] = matrices[(size_t)(i + 1) * 9 + j] - base[j];
            else for (int j = 0; j < 9; j++) delta[j] = base[j] - matrices[(size_t)(i - 1) * 9 + j];
        }
        for (int k = 0; k < per_frame; k++) {
            float* m32 = out + ((size_t)c * per_frame + k) * 9;
            for (int j = 0; j < 9; j++) m32[j] = (total > 1) ? (float)(base[j] + delta[j] * ts[k]) : (float)base[j];
        }
    }
    return 0;
}

extern "C" int vstab_warp_blur_clip_batch(vstab_ctx* ctx, const float* src, int n, int src_h, int src_w,
                                          const double* clip_matrices, int clip_total, int clip_first, const double* ts,
                                          int samples, int out_h, int out_w, int interp, const float* border_rgb, int subpix,
                                          float* dst, float* mask)
{
    if (int rc = check_common("vstab_warp_blur_clip_batch", ctx, src, n, src_h, src_w, clip_matrices, out_h, out_w, interp, border_rgb, subpix, dst)) return rc;
    VSTAB_REQUIRE(ts != nullptr, "vstab_warp_blur_clip_batch: ts is NULL");
    VSTAB_REQUIRE(samples >= 1 && samples <= MAX_BLUR_SAMPLES, "vstab_warp_blur_clip_batch: samples=%d outside [1,%d]", samples, MAX_BLUR_SAMPLES);
    VSTAB_REQUIRE(clip_first >= 0 && clip_first + n <= clip_total,
                  "vstab_warp_blur_clip_batch: frames [%d, %d) outside a clip of %d matrices", clip_first, clip_first + n, clip_total);
    VSTAB_HIP(hipSetDevice(ctx->device));
    const int per_frame = (clip_total <= 1) ? 1 : samples;
    std::vector<float> m32((size_t)n * per_frame * 9);
    if (int rc = vstab_blur_sample_matrices(clip_matrices, clip_total, clip_first, n, ts, samples, m32.data())) return rc;
    std::vector<WarpXform> xf((size_t)n * per_frame);
    for (size_t i = 0; i < xf.size(); i++) fill_xform(m32.data() + i * 9, &xf[i]);
    void* d_xf = nullptr;
    if (vstab_stage_params(ctx, xf.data(), xf.size() * sizeof(WarpXform), &d_xf)) return 1;

    WarpArgs a{};
    a.src = src; a.dst = dst; a.mask = mask; a.pad_count = nullptr;
    a.xf = static_cast<const WarpXform*>(d_xf);
    a.samples = samples; a.nxf_per_frame = per_frame;
    fill_geometry(a, n, src_h, src_w, out_h, out_w, border_rgb, dst, mask);
    KernelTimer timer(ctx, "warp_blur");
    return launch_warp_blur(a, interp, subpix, mask != nullptr, ctx->stream);
}

extern "C" int vstab_warp_blur_batch(vstab_ctx* ctx, const float* src, int n, int src_h, int src_w,
                                     const double* matrices, const double* ts, int samples, int out_h, int out_w,
                                     int interp, const float* border_rgb, int subpix, float* dst, float* mask)
{
    return vstab_warp_blur_clip_batch(ctx, src, n, src_h, src_w, matrices, n, 0, ts, samples, out_h, out_w, interp, border_rgb,
                                      subpix, dst, mask);
}
