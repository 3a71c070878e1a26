// vstab_gray.hip -- F2: RGB f32 -> gray u8 (truncating) -> INTER_AREA downscale to the working size.
//
// Replaces nodes/stabilizer_utils.py:236-242 (_make_gray) and :271-276 (cv2.resize INTER_AREA).
// This is the only estimation-stage pass that touches full-resolution data: 24.9 MB read per
// 1080p frame, 0.5 MB written -> HBM-bound, one fused pass for the integer-ratio cases
// (1080p -> 960x540 is 2x2, 4K -> 960x540 is 4x4).  Each thread produces 4 consecutive output
// pixels: it streams 4*K RGB pixels (K*48 contiguous bytes) from each of K source rows.
//
// Gray arithmetic follows OpenCV's RGB2Gray<float>: an 8-lane FMA body
// fma(b, 0.114, fma(g, 0.587, r*0.299)) and an unfused scalar tail for the last (w % 8) pixels
// of a row; then v = gray*255 (f32), clip to [0,255], truncate.
#include "vstab_internal.h"
#include <cfloat>
#include <cmath>

namespace {

__device__ __forceinline__ int gray_u8(float r, float g, float b, bool fused)
{
    const float k0 = 0.299f, k1 = 0.587f, k2 = 0.114f;
    float y;
    if (fused) y = __builtin_fmaf(b, k2, __builtin_fmaf(g, k1, r * k0));
    else y = r * k0 + g * k1 + b * k2;
    float v = y * 255.0f;
    v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
    return (int)v;
}

__device__ __forceinline__ int sat_u8_round(float v)
{
    int i = (int)__builtin_rintf(v);
    return i < 0 ? 0 : (i > 255 ? 255 : i);
}

// Integer-ratio INTER_AREA (K x K boxes) fused with the gray conversion. K == 1: gray only.
// One workgroup per output row; thread t produces the pixels t, t + 256, ...: the lanes of a wavefront read
// CONSECUTIVE K-pixel groups (K * 12 B apart), so one load instruction touches K * 12 * 64 contiguous bytes
// instead of 64 scattered 16-B pieces (the texture addresser, not HBM, was the limit with 4 pixels per thread).
template <int K>
__global__ __launch_bounds__(256) void gray_area_int_kernel(const float* __restrict__ frames, uint8_t* __restrict__ out,
                                                             int n, int sh, int sw, int dh, int dw, int body)
{
    const int y = blockIdx.x % dh, f = blockIdx.x / dh;
    const float* rowbase = frames + ((size_t)f * sh + (size_t)y * K) * sw * 3;
    uint8_t* D = out + ((size_t)f * dh + y) * dw;
#pragma unroll 4
    for (int x = threadIdx.x; x < dw; x += 256) {
        int sum = 0;
        const float* base = rowbase + (size_t)x * K * 3;
#pragma unroll
        for (int j = 0; j < K; j++) {
            float row[K * 3];
            __builtin_memcpy(row, base + (size_t)j * sw * 3, sizeof(row));
#pragma unroll
            for (int i = 0; i < K; i++) sum += gray_u8(row[i * 3], row[i * 3 + 1], row[i * 3 + 2], (x * K + i) < body);
        }
        int o;
        if (K == 1) o = sum;
        else if (K == 2) o = (sum + 2) >> 2;
        else o = sat_u8_round(sum * (1.f / (K * K)));
        D[x] = (uint8_t)o;
    }
}

// Generic integer ratio (kx, ky) on an already-gray image.
__global__ __launch_bounds__(256) void area_int_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n,
                                                          int sh, int sw, int dh, int dw, int kx, int ky)
{
    const long long total = (long long)n * dh * dw;
    const float scale = 1.f / (kx * ky);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(t % dw);
        const long long r = t / dw;
        const int y = (int)(r % dh);
        const int f = (int)(r / dh);
        const uint8_t* S = src + ((size_t)f * sh + (size_t)y * ky) * sw + (size_t)x * kx;
        int sum = 0;
        for (int j = 0; j < ky; j++)
            for (int i = 0; i < kx; i++) sum += S[(size_t)j * sw + i];
        int o;
        if (kx == 2 && ky == 2) o = (sum + 2) >> 2;
        else o = sat_u8_round(sum * scale);
        dst[t] = (uint8_t)o;
    }
}

struct AreaTabEntry { int si; float alpha; };

// General INTER_AREA (non-integer ratio): f32 accumulation in OpenCV's order.
__global__ __launch_bounds__(256) void area_general_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n,
                                                              int sh, int sw, int dh, int dw,
                                                              const AreaTabEntry* __restrict__ xtab, const int* __restrict__ xstart,
                                                              const AreaTabEntry* __restrict__ ytab, const int* __restrict__ ystart)
{
    const long long total = (long long)n * dh * dw;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(t % dw);
        const long long r = t / dw;
        const int y = (int)(r % dh);
        const int f = (int)(r / dh);
        const uint8_t* S = src + (size_t)f * sh * sw;
        float sum = 0.f;
        const int j0 = ystart[y], j1 = ystart[y + 1];
        const int k0 = xstart[x], k1 = xstart[x + 1];
        for (int j = j0; j < j1; j++) {
            const uint8_t* row = S + (size_t)ytab[j].si * sw;
            float buf = 0.f;
            for (int k = k0; k < k1; k++) buf += row[xtab[k].si] * xtab[k].alpha;
            const float term = ytab[j].alpha * buf;
            sum = (j == j0) ? term : sum + term;
        }
        dst[t] = (uint8_t)sat_u8_round(sum);
    }
}

int ceil_d(double v) { int i = (int)v; return i + (i < v); }
int floor_d(double v) { int i = (int)v; return i - (i > v); }

// computeResizeAreaTab of OpenCV's resize.cpp
void build_area_tab(int ssize, int dsize, double scale, std::vector<AreaTabEntry>& tab, std::vector<int>& start)
{
    tab.clear();
    start.assign((size_t)dsize + 1, 0);
    for (int dx = 0; dx < dsize; dx++) {
        start[dx] = (int)tab.size();
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = ceil_d(fsx1), sx2 = floor_d(fsx2);
        sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
        sx1 = sx1 < sx2 ? sx1 : sx2;
        if (sx1 - fsx1 > 1e-3) tab.push_back({sx1 - 1, (float)((sx1 - fsx1) / cellWidth)});
        for (int sx = sx1; sx < sx2; sx++) tab.push_back({sx, (float)(1.0 / cellWidth)});
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2;
            a = a < 1. ? a : 1.;
            a = a < cellWidth ? a : cellWidth;
            tab.push_back({sx2, (float)(a / cellWidth)});
        }
    }
    start[dsize] = (int)tab.size();
}

unsigned grid_for(long long items)
{
    long long b = (items + 255) / 256;
    const long long cap = 256LL * 16;  // 256 CUs x 16 blocks, grid-stride beyond that
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int vstab_gray_downscale(vstab_ctx* ctx, const float* frames, int n, int src_h, int src_w, int work_h,
                                    int work_w, uint8_t* gray)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_gray_downscale: ctx is NULL");
    VSTAB_REQUIRE(frames && gray, "vstab_gray_downscale: NULL pointer argument");
    VSTAB_REQUIRE(n > 0 && src_h > 0 && src_w > 0 && work_h > 0 && work_w > 0, "vstab_gray_downscale: non-positive size");
    VSTAB_REQUIRE(work_h <= src_h && work_w <= src_w, "vstab_gray_downscale: working size %dx%d larger than source %dx%d", work_w, work_h, src_w, src_h);
    VSTAB_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int body = src_w & ~7;
    KernelTimer timer(ctx, "gray");

    if (work_h == src_h && work_w == src_w) {
        hipLaunchKernelGGL((gray_area_int_kernel<1>), dim3((unsigned)(n * work_h)), dim3(256), 0, st, frames, gray, n, src_h, src_w, work_h, work_w, body);
        VSTAB_HIP(hipGetLastError());
        return 0;
    }
    // cv::resize: scale = 1/(dsize/ssize); integer-ratio fast path when both are integers within DBL_EPSILON
    const double scale_x = 1. / ((double)work_w / src_w), scale_y = 1. / ((double)work_h / src_h);
    const int isx = (int)std::lrint(scale_x), isy = (int)std::lrint(scale_y);
    const bool fast = std::fabs(scale_x - isx) < DBL_EPSILON && std::fabs(scale_y - isy) < DBL_EPSILON;
    if (fast && isx == isy && (isx == 2 || isx == 4)) {
        if (isx == 2)
            hipLaunchKernelGGL((gray_area_int_kernel<2>), dim3((unsigned)(n * work_h)), dim3(256), 0, st, frames, gray, n, src_h, src_w, work_h, work_w, body);
        else
            hipLaunchKernelGGL((gray_area_int_kernel<4>), dim3((unsigned)(n * work_h)), dim3(256), 0, st, frames, gray, n, src_h, src_w, work_h, work_w, body);
        VSTAB_HIP(hipGetLastError());
        return 0;
    }
    // two passes: full-resolution gray into scratch, then the area resize
    const size_t full = (size_t)n * src_h * src_w;
    if (ctx->d_gray_tmp.reserve(full)) return 1;
    uint8_t* tmp = static_cast<uint8_t*>(ctx->d_gray_tmp.ptr);
    {
        hipLaunchKernelGGL((gray_area_int_kernel<1>), dim3((unsigned)(n * src_h)), dim3(256), 0, st, frames, tmp, n, src_h, src_w, src_h, src_w, body);
        VSTAB_HIP(hipGetLastError());
    }
    const long long out_items = (long long)n * work_h * work_w;
    if (fast) {
        hipLaunchKernelGGL(area_int_u8_kernel, dim3(grid_for(out_items)), dim3(256), 0, st, tmp, gray, n, src_h, src_w, work_h, work_w, isx, isy);
        VSTAB_HIP(hipGetLastError());
        return 0;
    }
    std::vector<AreaTabEntry> xtab, ytab;
    std::vector<int> xstart, ystart;
    build_area_tab(src_w, work_w, scale_x, xtab, xstart);
    build_area_tab(src_h, work_h, scale_y, ytab, ystart);
    // pack the four tables into one staged upload
    const size_t b_xt = xtab.size() * sizeof(AreaTabEntry), b_yt = ytab.size() * sizeof(AreaTabEntry);
    const size_t b_xs = xstart.size() * sizeof(int), b_ys = ystart.size() * sizeof(int);
    std::vector<unsigned char> blob(b_xt + b_yt + b_xs + b_ys);
    memcpy(blob.data(), xtab.data(), b_xt);
    memcpy(blob.data() + b_xt, ytab.data(), b_yt);
    memcpy(blob.data() + b_xt + b_yt, xstart.data(), b_xs);
    memcpy(blob.data() + b_xt + b_yt + b_xs, ystart.data(), b_ys);
    void* d_blob = nullptr;
    if (vstab_stage_params(ctx, blob.data(), blob.size(), &d_blob)) return 1;
    unsigned char* db = static_cast<unsigned char*>(d_blob);
    hipLaunchKernelGGL(area_general_u8_kernel, dim3(grid_for(out_items)), dim3(256), 0, st, tmp, gray, n, src_h, src_w, work_h, work_w,
                       reinterpret_cast<const AreaTabEntry*>(db), reinterpret_cast<const int*>(db + b_xt + b_yt),
                       reinterpret_cast<const AreaTabEntry*>(db + b_xt), reinterpret_cast<const int*>(db + b_xt + b_yt + b_xs));
    VSTAB_HIP(hipGetLastError());
    return 0;
}
