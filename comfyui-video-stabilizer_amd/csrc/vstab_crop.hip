// vstab_crop.hip -- coverage analysis behind framing_mode="crop" (SURVEY 8f N1).
//
// Replaces, for all frames at once, the cv2 calls inside the reference's keep_fov crop solver:
//   nodes/stabilizer_utils.py:611-643  warpPerspective(ones, INTER_NEAREST) > 0.5 -> dilate 3x3 -> erode 3x3
//                                      -> bounding box of the remaining content per frame
//   nodes/stabilizer_utils.py:763-787  AND of all frames' coverage -> erode 3x3
// The bisection over the stabilisation scale and the integral-image rectangle search stay on the host
// (they are scalar / O(h*w) NumPy in the reference too).
#include "vstab_internal.h"

namespace {

struct CovXform { double m[9]; double wn; int affine; int pad_; };

__device__ __forceinline__ int clamp_round_i32(double v)
{
    const double hi = 2147483647.0, lo = -2147483648.0;
    double m = (v < hi) ? v : hi;
    double r = (lo < m) ? m : lo;
    return (int)__builtin_rint(r);
}
__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// nearest-neighbour coverage exactly as the warp kernel computes its mask (same per-64-column block terms)
__global__ __launch_bounds__(256) void coverage_kernel(const CovXform* __restrict__ xf, uint8_t* __restrict__ cov, int n, int sh, int sw,
                                                       int dh, int dw, int bw0)
{
    const long long total = (long long)n * dh * dw;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(t % dw);
        const long long r = t / dw;
        const int y = (int)(r % dh);
        const int f = (int)(r / dh);
        const CovXform& X = xf[f];
        const int xb = (bw0 >= dw) ? 0 : (x / bw0) * bw0;
        const double dxb = (double)xb, dy = (double)y, dx1 = (double)(x - xb);
        const double X0 = X.m[0] * dxb + X.m[1] * dy + X.m[2];
        const double Y0 = X.m[3] * dxb + X.m[4] * dy + X.m[5];
        const double W0 = X.m[6] * dxb + X.m[7] * dy + X.m[8];
        double Wn;
        if (X.affine) Wn = X.wn;
        else { const double W = W0 + X.m[6] * dx1; Wn = (W != 0.0) ? 1.0 / W : 0.0; }
        const int nx = sat_short(clamp_round_i32((X0 + X.m[0] * dx1) * Wn));
        const int ny = sat_short(clamp_round_i32((Y0 + X.m[3] * dx1) * Wn));
        cov[t] = ((unsigned)nx < (unsigned)sw && (unsigned)ny < (unsigned)sh) ? 1 : 0;
    }
}

// closing = dilate 3x3 then erode 3x3 (out-of-image pixels ignored by both), in two passes over u8 planes, and the
// per-frame bounding box of the result.  One workgroup per (8-row band, frame): the box is reduced in registers,
// wavefront shuffles and LDS, and reaches memory as 4 atomics per workgroup -- per-pixel atomics on the four words of
// a frame serialised to a full second per 256 x 1080p clip.
constexpr int BAND = 8;

__global__ __launch_bounds__(256) void dilate3_kernel(const uint8_t* __restrict__ cov, uint8_t* __restrict__ dil, int dh, int dw)
{
    const int f = blockIdx.y, y0 = blockIdx.x * BAND;
    const uint8_t* C = cov + (size_t)f * dh * dw;
    uint8_t* D = dil + (size_t)f * dh * dw;
    for (int k = threadIdx.x; k < BAND * dw; k += 256) {
        const int y = y0 + k / dw, x = k % dw;
        if (y >= dh) break;
        int v = 0;
        for (int gy = -1; gy <= 1; gy++)
            for (int gx = -1; gx <= 1; gx++) {
                const int py = y + gy, px = x + gx;
                if (py < 0 || py >= dh || px < 0 || px >= dw) continue;   // dilate border = -inf
                v |= C[(size_t)py * dw + px];
            }
        D[(size_t)y * dw + x] = (uint8_t)v;
    }
}

__global__ __launch_bounds__(256) void erode_bbox_kernel(const uint8_t* __restrict__ dil, int* __restrict__ bbox, int dh, int dw)
{
    __shared__ int s_box[4][4];
    const int f = blockIdx.y, y0 = blockIdx.x * BAND;
    const uint8_t* D = dil + (size_t)f * dh * dw;
    int x_lo = 0x7fffffff, y_lo = 0x7fffffff, x_hi = -1, y_hi = -1;
    for (int k = threadIdx.x; k < BAND * dw; k += 256) {
        const int y = y0 + k / dw, x = k % dw;
        if (y >= dh) break;
        int v = 1;
        for (int ey = -1; ey <= 1; ey++)
            for (int ex = -1; ex <= 1; ex++) {
                const int qy = y + ey, qx = x + ex;
                if (qy < 0 || qy >= dh || qx < 0 || qx >= dw) continue;   // erode border = +inf
                v &= D[(size_t)qy * dw + qx];
            }
        if (v) { x_lo = min(x_lo, x); y_lo = min(y_lo, y); x_hi = max(x_hi, x); y_hi = max(y_hi, y); }
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
        x_lo = min(x_lo, __shfl_down(x_lo, sft)); y_lo = min(y_lo, __shfl_down(y_lo, sft));
        x_hi = max(x_hi, __shfl_down(x_hi, sft)); y_hi = max(y_hi, __shfl_down(y_hi, sft));
    }
    if ((threadIdx.x & 63) == 0) { int* b = s_box[threadIdx.x >> 6]; b[0] = x_lo; b[1] = y_lo; b[2] = x_hi; b[3] = y_hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) {
            x_lo = min(x_lo, s_box[i][0]); y_lo = min(y_lo, s_box[i][1]); x_hi = max(x_hi, s_box[i][2]); y_hi = max(y_hi, s_box[i][3]);
        }
        if (x_hi >= 0) {
            atomicMin(bbox + f * 4 + 0, x_lo); atomicMin(bbox + f * 4 + 1, y_lo);
            atomicMax(bbox + f * 4 + 2, x_hi); atomicMax(bbox + f * 4 + 3, y_hi);
        }
    }
}

__global__ __launch_bounds__(256) void common_kernel(const uint8_t* __restrict__ cov, uint8_t* __restrict__ common, int n, int npx)
{
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += gridDim.x * blockDim.x) {
        int all = 1;
        for (int f = 0; f < n; f++) all &= cov[(size_t)f * npx + p];
        common[p] = (uint8_t)all;
    }
}

__global__ __launch_bounds__(256) void erode3_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int dh, int dw)
{
    const int npx = dh * dw;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += gridDim.x * blockDim.x) {
        const int y = p / dw, x = p - y * dw;
        int v = 1;
        for (int ey = -1; ey <= 1; ey++)
            for (int ex = -1; ex <= 1; ex++) {
                const int qy = y + ey, qx = x + ex;
                if (qy < 0 || qy >= dh || qx < 0 || qx >= dw) continue;
                v &= src[(size_t)qy * dw + qx];
            }
        dst[p] = (uint8_t)v;
    }
}

// AND over all frames of the nearest coverage, one thread per output pixel (no per-frame planes in memory)
__global__ __launch_bounds__(256) void common_coverage_kernel(const CovXform* __restrict__ xf, uint8_t* __restrict__ common, int n, int sh,
                                                              int sw, int dh, int dw, int bw0)
{
    const int npx = dh * dw;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += gridDim.x * blockDim.x) {
        const int y = p / dw, x = p - y * dw;
        const int xb = (bw0 >= dw) ? 0 : (x / bw0) * bw0;
        const double dxb = (double)xb, dy = (double)y, dx1 = (double)(x - xb);
        int all = 1;
        for (int f = 0; f < n && all; f++) {
            const CovXform& X = xf[f];
            const double X0 = X.m[0] * dxb + X.m[1] * dy + X.m[2];
            const double Y0 = X.m[3] * dxb + X.m[4] * dy + X.m[5];
            const double W0 = X.m[6] * dxb + X.m[7] * dy + X.m[8];
            double Wn;
            if (X.affine) Wn = X.wn;
            else { const double W = W0 + X.m[6] * dx1; Wn = (W != 0.0) ? 1.0 / W : 0.0; }
            const int nx = sat_short(clamp_round_i32((X0 + X.m[0] * dx1) * Wn));
            const int ny = sat_short(clamp_round_i32((Y0 + X.m[3] * dx1) * Wn));
            all &= ((unsigned)nx < (unsigned)sw && (unsigned)ny < (unsigned)sh) ? 1 : 0;
        }
        common[p] = (uint8_t)all;
    }
}

unsigned grid_for(long long items)
{
    long long b = (items + 255) / 256;
    if (b > 256LL * 32) b = 256LL * 32;
    if (b < 1) b = 1;
    return (unsigned)b;
}

int stage_cov_xforms(vstab_ctx* ctx, const float* matrices, int n, void** d_xf)
{
    std::vector<CovXform> xf((size_t)n);
    for (int i = 0; i < n; i++) {
        double M[9];
        for (int j = 0; j < 9; j++) M[j] = (double)matrices[(size_t)i * 9 + j];
        vstab_invert3x3(M, xf[i].m);
        xf[i].affine = (xf[i].m[6] == 0.0 && xf[i].m[7] == 0.0) ? 1 : 0;
        xf[i].wn = (xf[i].m[8] != 0.0) ? 1.0 / xf[i].m[8] : 0.0;
        xf[i].pad_ = 0;
    }
    return vstab_stage_params(ctx, xf.data(), xf.size() * sizeof(CovXform), d_xf);
}

}  // namespace

extern "C" int vstab_common_coverage(vstab_ctx* ctx, const float* matrices, int n, int src_h, int src_w, int out_h, int out_w,
                                     uint8_t* common)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_common_coverage: ctx is NULL");
    VSTAB_REQUIRE(matrices && common, "vstab_common_coverage: NULL pointer argument");
    VSTAB_REQUIRE(n > 0 && src_h > 0 && src_w > 0 && out_h > 0 && out_w > 0, "vstab_common_coverage: non-positive size");
    VSTAB_REQUIRE(src_h <= 32767 && src_w <= 32767, "vstab_common_coverage: source larger than 32767 px");
    VSTAB_REQUIRE((long long)out_h * out_w < 0x7fffffffLL, "vstab_common_coverage: output too large");
    VSTAB_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    void* d_xf = nullptr;
    if (stage_cov_xforms(ctx, matrices, n, &d_xf)) return 1;
    const size_t npx = (size_t)out_h * out_w;
    if (ctx->d_gray_tmp.reserve(npx + 256)) return 1;
    if (ctx->h_fit.reserve(npx)) return 1;
    uint8_t* d_common = static_cast<uint8_t*>(ctx->d_gray_tmp.ptr);
    const int BLOCK_SZ = 32;
    int bh0 = BLOCK_SZ / 2 < out_h ? BLOCK_SZ / 2 : out_h;
    int bw0 = BLOCK_SZ * BLOCK_SZ / bh0 < out_w ? BLOCK_SZ * BLOCK_SZ / bh0 : out_w;
    hipLaunchKernelGGL(common_coverage_kernel, dim3(grid_for((long long)npx)), dim3(256), 0, st, static_cast<const CovXform*>(d_xf), d_common,
                       n, src_h, src_w, out_h, out_w, bw0);
    VSTAB_HIP(hipGetLastError());
    VSTAB_HIP(hipMemcpyAsync(ctx->h_fit.ptr, d_common, npx, hipMemcpyDeviceToHost, st));
    VSTAB_HIP(hipStreamSynchronize(st));
    memcpy(common, ctx->h_fit.ptr, npx);
    return 0;
}

extern "C" int vstab_crop_analysis(vstab_ctx* ctx, const float* matrices, int n, int src_h, int src_w, int out_h, int out_w,
                                   int32_t* bbox, uint8_t* common)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_crop_analysis: ctx is NULL");
    VSTAB_REQUIRE(matrices && bbox && common, "vstab_crop_analysis: NULL pointer argument");
    VSTAB_REQUIRE(n > 0 && src_h > 0 && src_w > 0 && out_h > 0 && out_w > 0, "vstab_crop_analysis: non-positive size");
    VSTAB_REQUIRE(src_h <= 32767 && src_w <= 32767, "vstab_crop_analysis: source larger than 32767 px");
    VSTAB_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    void* d_xf = nullptr;
    if (stage_cov_xforms(ctx, matrices, n, &d_xf)) return 1;
    const size_t npx = (size_t)out_h * out_w;
    const size_t need = 2 * (size_t)n * npx + 2 * npx + sizeof(int) * 4 * (size_t)n + 1024;
    if (ctx->d_gray_tmp.reserve(need)) return 1;
    const size_t npx_al = (npx + 15) & ~size_t(15);
    if (ctx->h_fit.reserve(npx_al + sizeof(int) * 4 * (size_t)n)) return 1;
    uint8_t* d_cov = static_cast<uint8_t*>(ctx->d_gray_tmp.ptr);
    uint8_t* d_dil = d_cov + (size_t)n * npx;
    uint8_t* d_common = d_dil + (size_t)n * npx;
    uint8_t* d_eroded = d_common + npx;
    int* d_bbox = reinterpret_cast<int*>((reinterpret_cast<uintptr_t>(d_eroded + npx) + 255) & ~uintptr_t(255));
    std::vector<int> init((size_t)n * 4);
    for (int i = 0; i < n; i++) { init[i * 4] = 0x7fffffff; init[i * 4 + 1] = 0x7fffffff; init[i * 4 + 2] = -1; init[i * 4 + 3] = -1; }
    VSTAB_HIP(hipMemcpyAsync(d_bbox, init.data(), init.size() * sizeof(int), hipMemcpyHostToDevice, st));
    VSTAB_HIP(hipStreamSynchronize(st));   // `init` is pageable and goes out of scope

    const int BLOCK_SZ = 32;
    int bh0 = BLOCK_SZ / 2 < out_h ? BLOCK_SZ / 2 : out_h;
    int bw0 = BLOCK_SZ * BLOCK_SZ / bh0 < out_w ? BLOCK_SZ * BLOCK_SZ / bh0 : out_w;
    const long long items = (long long)n * out_h * out_w;
    hipLaunchKernelGGL(coverage_kernel, dim3(grid_for(items)), dim3(256), 0, st, static_cast<const CovXform*>(d_xf), d_cov, n, src_h, src_w,
                       out_h, out_w, bw0);
    {
        const dim3 bgrid((unsigned)((out_h + BAND - 1) / BAND), (unsigned)n);
        hipLaunchKernelGGL(dilate3_kernel, bgrid, dim3(256), 0, st, d_cov, d_dil, out_h, out_w);
        hipLaunchKernelGGL(erode_bbox_kernel, bgrid, dim3(256), 0, st, d_dil, d_bbox, out_h, out_w);
    }
    hipLaunchKernelGGL(common_kernel, dim3(grid_for((long long)npx)), dim3(256), 0, st, d_cov, d_common, n, (int)npx);
    hipLaunchKernelGGL(erode3_kernel, dim3(grid_for((long long)npx)), dim3(256), 0, st, d_common, d_eroded, out_h, out_w);
    VSTAB_HIP(hipGetLastError());
    char* hbuf = static_cast<char*>(ctx->h_fit.ptr);
    VSTAB_HIP(hipMemcpyAsync(hbuf, d_eroded, npx, hipMemcpyDeviceToHost, st));
    VSTAB_HIP(hipMemcpyAsync(hbuf + npx_al, d_bbox, sizeof(int) * 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    VSTAB_HIP(hipStreamSynchronize(st));
    memcpy(common, hbuf, npx);
    const int* hb = reinterpret_cast<const int*>(hbuf + npx_al);
    for (int i = 0; i < n; i++) {
        const bool empty = hb[i * 4 + 2] < 0;
        for (int k = 0; k < 4; k++) bbox[i * 4 + k] = empty ? -1 : hb[i * 4 + k];
    }
    return 0;
}
