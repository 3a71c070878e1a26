// vstab_gray.hip -- F2: RGB f32 -> gray u8 (truncating) -> INTER_AREA downscale to the working size.
//
// Replaces nodes/stabilizer_utils.py:236-242 (_make_gray) and :271-276 (cv2.resize INTER_AREA).
// This is the only estimation-stage pass that touches full-resolution data: 24.9 MB read per
// 1080p frame, 0.5 MB written -> HBM-bound, one fused pass for the integer-ratio cases
// (1080p -> 960x540 is 2x2, 4K -> 960x540 is 4x4): a lane per source pixel column, see gray_area_int_kernel.
//
// Gray arithmetic follows OpenCV's RGB2Gray<float>: an 8-lane FMA body
// fma(b, 0.114, fma(g, 0.587, r*0.299)) and an unfused scalar tail for the last (w % 8) pixels
// of a row; then v = gray*255 (f32), clip to [0,255], truncate.
#include "vstab_internal.h"
#include <algorithm>
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
// One workgroup per output row, one SOURCE pixel column per lane: a lane reads its pixel (12 B) from each of the K rows
// of the box and the K lanes of a box add up over DPP, so every load instruction of a wavefront covers 768 contiguous
// bytes with no holes.  The frames are read once and never again by this pass: non-temporal loads (measured on
// MI355X, profiles/r02_gray_forms.md: 256 x 1080p 1.09 -> 0.96 ms, 6.6 TB/s; 4K 0.82 -> 0.71 ms).  The launch picks a
// workgroup size that divides the row into whole passes (640 threads for 1920 and 3840 columns).
//
// RANGE: the same pass also yields the largest sample of every source row it reads (NaN if the row holds one, as
// numpy's max does) -- the value-range sniff of nodes/stabilizer_utils.py:127-131 (`float(arr.max()) > 1.5` per frame)
// for free, instead of a second 24.9 MB read per frame.  Written per block to row_max[f * dh + y]; only valid when the
// K x K boxes tile the whole source (the host checks dh * K == sh and dw * K == sw).
template <int K, bool RANGE>
__global__ __launch_bounds__(1024) void gray_area_int_kernel(const float* __restrict__ frames, uint8_t* __restrict__ out,
                                                             int n, int sh, int sw, int dh, int dw, int body,
                                                             float* __restrict__ row_max)
{
    __shared__ float s_max[16];
    __shared__ int s_nan[16];
    const int y = (int)blockIdx.x % dh, f = (int)blockIdx.x / dh;
    const float* rowbase = frames + ((size_t)f * sh + (size_t)y * K) * sw * 3;
    uint8_t* D = out + ((size_t)f * dh + y) * dw;
    float vmax = -INFINITY;
    int has_nan = 0;
    const int used = dw * K;   // == sw on the fused path
    constexpr int U = 4 / K;   // columns per thread and pass: four 12-B loads in flight
    typedef float f3_t __attribute__((ext_vector_type(3), aligned(4)));
    for (int sx0 = threadIdx.x; sx0 < used; sx0 += U * blockDim.x) {
        f3_t px[U][K];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int sx = sx0 + u * blockDim.x;
            if (sx < used) {
#pragma unroll
                for (int j = 0; j < K; j++)
                    px[u][j] = __builtin_nontemporal_load(reinterpret_cast<const f3_t*>(rowbase + (size_t)j * sw * 3 + (size_t)sx * 3));
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int sx = sx0 + u * blockDim.x;
            if (sx < used) {
                const bool fused = sx < body;
                int sum = 0;
#pragma unroll
                for (int j = 0; j < K; j++) {
                    const float cr = px[u][j].x, cg = px[u][j].y, cb = px[u][j].z;
                    sum += gray_u8(cr, cg, cb, fused);
                    if (RANGE) {
                        vmax = __builtin_fmaxf(vmax, __builtin_fmaxf(__builtin_fmaxf(cr, cg), cb));
                        // per component: a sum probe would also fire on +inf next to -inf, where numpy's max is +inf.
                        // (Round 5 tried the NaN-propagating v_maximum3_f32 and v_max3_f32 + a wavefront NaN mask through
                        // inline asm -- 8 instead of ~70 VALU instructions per four pixels: both SLOWER, 1.27 vs 1.13 ms per
                        // 256 x 1080p on one box; the asm statements keep the compiler from overlapping the next loads with
                        // this arithmetic, which is what the pass lives on.  profiles/r05_dis_small_steps.md)
                        has_nan |= ((cr != cr) | (cg != cg) | (cb != cb)) ? 1 : 0;
                    }
                }
                // the K lanes of a box sit in one quad (blockDim.x and `used` are multiples of K): integer adds, any order
                if (K >= 2) sum += __builtin_amdgcn_update_dpp(0, sum, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
                if (K == 4) sum += __builtin_amdgcn_update_dpp(0, sum, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
                if ((sx & (K - 1)) == 0) {
                    int o;
                    if (K == 1) o = sum;
                    else if (K == 2) o = (sum + 2) >> 2;
                    else o = sat_u8_round(sum * (1.f / (K * K)));
                    D[sx / K] = (uint8_t)o;
                }
            }
        }
    }
    if (RANGE) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            vmax = __builtin_fmaxf(vmax, __shfl_down(vmax, off));
            has_nan |= __shfl_down(has_nan, off);
        }
        const int nw = (blockDim.x + 63) >> 6;
        if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = vmax; s_nan[threadIdx.x >> 6] = has_nan; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = s_max[0];
            int nn = s_nan[0];
            for (int k = 1; k < nw; k++) { m = __builtin_fmaxf(m, s_max[k]); nn |= s_nan[k]; }
            row_max[(size_t)f * dh + y] = nn ? NAN : m;
        }
    }
}

// per-frame maximum of the per-row maxima (NaN wins, as in numpy): one workgroup per frame
// host_max (optional): the maxima also go to coherent host memory -- by the LAST workgroup to finish (a device-scope count
// finds it), as one coalesced store of all `gridDim.x` values followed by a release store of `target` into host_flag.  (One
// store + one system-scope atomic per workgroup, the first form of this, serialised 256 PCIe round trips: 276 us.)
__global__ __launch_bounds__(256) void frame_max_kernel(const float* __restrict__ row_max, float* __restrict__ frame_max, int rows,
                                                        float* host_max, unsigned* host_flag, unsigned* dev_count, unsigned target)
{
    __shared__ float s_max[4];
    __shared__ int s_nan[4];
    __shared__ int s_last;
    const float* R = row_max + (size_t)blockIdx.x * rows;
    float vmax = -INFINITY;
    int has_nan = 0;
    for (int k = threadIdx.x; k < rows; k += 256) {
        const float v = R[k];
        has_nan |= (v != v) ? 1 : 0;
        vmax = __builtin_fmaxf(vmax, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        vmax = __builtin_fmaxf(vmax, __shfl_down(vmax, off));
        has_nan |= __shfl_down(has_nan, off);
    }
    if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = vmax; s_nan[threadIdx.x >> 6] = has_nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = __builtin_fmaxf(__builtin_fmaxf(s_max[0], s_max[1]), __builtin_fmaxf(s_max[2], s_max[3]));
        frame_max[blockIdx.x] = (s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3]) ? NAN : m;
        s_last = 0;
        if (host_max != nullptr) {
            // release the value, count this workgroup; the one that completes the pass's count mirrors all of them
            const unsigned prev = __hip_atomic_fetch_add(dev_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (prev + 1u == target) ? 1 : 0;
        }
    }
    if (host_max == nullptr) return;
    __syncthreads();
    if (s_last) {
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 256)
            host_max[i] = __hip_atomic_load(frame_max + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(host_flag, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// stabilizer_utils.py:127-131 applied to a clip: frames whose maximum exceeds 1.5 are divided by 255 (IEEE f32
// division, as numpy's `arr /= 255.0`), the others are copied; NaN maxima compare False.  Out of place: the caller's
// tensor is never modified.
__global__ __launch_bounds__(256) void value_range_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          const float* __restrict__ frame_max, long long per_frame, long long total)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const float v = src[t];
        dst[t] = (frame_max[t / per_frame] > 1.5f) ? v / 255.0f : v;
    }
}

// per-row maximum of a clip without producing gray (Motion Apply has no estimation pass): one workgroup per source row
// VEC: pieces are a multiple of 4 floats and 16-B aligned (checked by the host); otherwise plain 4-B loads
template <bool VEC>
__global__ __launch_bounds__(256) void row_max_kernel(const float* __restrict__ frames, float* __restrict__ row_max, int row_floats)
{
    __shared__ float s_max[4];
    __shared__ int s_nan[4];
    const float* S = frames + (size_t)blockIdx.x * row_floats;
    float vmax = -INFINITY;
    int has_nan = 0;
    if (VEC) {
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const f4_t* R = reinterpret_cast<const f4_t*>(S);
        const int nvec = row_floats / 4;
#pragma unroll 4
        for (int k = threadIdx.x; k < nvec; k += 256) {
            const f4_t v = __builtin_nontemporal_load(R + k);   // read once by this pass (see gray_area_int_kernel)
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fmaxf(v.x, v.y)), __builtin_fmaxf(v.z, v.w));
            has_nan |= ((v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w)) ? 1 : 0;
        }
    } else {
        for (int k = threadIdx.x; k < row_floats; k += 256) {
            const float v = S[k];
            vmax = __builtin_fmaxf(vmax, v);
            has_nan |= (v != v) ? 1 : 0;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        vmax = __builtin_fmaxf(vmax, __shfl_down(vmax, off));
        has_nan |= __shfl_down(has_nan, off);
    }
    if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = vmax; s_nan[threadIdx.x >> 6] = has_nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = __builtin_fmaxf(__builtin_fmaxf(s_max[0], s_max[1]), __builtin_fmaxf(s_max[2], s_max[3]));
        row_max[blockIdx.x] = (s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3]) ? NAN : m;
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

static int gray_run(vstab_ctx* ctx, const float* frames, int n, int src_h, int src_w, int work_h, int work_w, uint8_t* gray,
                    float* frame_max)
{
    VSTAB_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int body = src_w & ~7;
    KernelTimer timer(ctx, "gray");
    float* row_max = nullptr;
    int range_rows = 0;   // rows of row_max per frame once the gray pass has filled it

    // workgroup size: the multiple of 64 (<= 1024) that covers a row of `cols` columns in whole passes with the fewest
    // idle lane slots; ties go to the size nearest 640 (measured best for 1920 / 3840 columns)
    auto threads_for = [](int cols) {
        int best = 256;
        long long best_cost = -1;
        for (int t = 256; t <= 1024; t += 64) {
            const long long cost = (long long)((cols + t - 1) / t) * t;
            if (best_cost < 0 || cost < best_cost || (cost == best_cost && std::abs(t - 640) < std::abs(best - 640))) { best = t; best_cost = cost; }
        }
        if (const char* e = getenv("VSTAB_GRAY_THREADS")) best = atoi(e);   // A/B measurement (tools/gray_forms.py)
        return best;
    };
#define LAUNCH_GRAY(K, ROWS, OUT)                                                                                         \
    do {                                                                                                                  \
        const int threads = threads_for(src_w / K * K);                                                                   \
        if (row_max)                                                                                                      \
            hipLaunchKernelGGL((gray_area_int_kernel<K, true>), dim3((unsigned)(n * (ROWS))), dim3(threads), 0, st, frames, OUT, n, src_h, \
                               src_w, (ROWS), src_w / K, body, row_max);                                                  \
        else                                                                                                              \
            hipLaunchKernelGGL((gray_area_int_kernel<K, false>), dim3((unsigned)(n * (ROWS))), dim3(threads), 0, st, frames, OUT, n, src_h, \
                               src_w, (ROWS), src_w / K, body, row_max);                                                  \
        VSTAB_HIP(hipGetLastError());                                                                                     \
    } while (0)

    // cv::resize: scale = 1/(dsize/ssize); integer-ratio fast path when both are integers within DBL_EPSILON
    const double scale_x = 1. / ((double)work_w / src_w), scale_y = 1. / ((double)work_h / src_h);
    const int isx = (int)std::lrint(scale_x), isy = (int)std::lrint(scale_y);
    const bool same = (work_h == src_h && work_w == src_w);
    const bool fast = !same && std::fabs(scale_x - isx) < DBL_EPSILON && std::fabs(scale_y - isy) < DBL_EPSILON;
    const bool fused = fast && isx == isy && (isx == 2 || isx == 4);
    const int K = same ? 1 : (fused ? isx : 1);
    if (frame_max) {
        // the gray pass covers every source sample exactly when its K x K boxes tile the source
        const int rows = (same || !fused) ? src_h : work_h;
        if (ctx->d_range.reserve(sizeof(float) * (size_t)n * rows)) return 1;
        row_max = static_cast<float*>(ctx->d_range.ptr);
        range_rows = rows;
    }
    if (same) {
        LAUNCH_GRAY(1, work_h, gray);
    } else if (fused) {
        VSTAB_REQUIRE(work_w * K == src_w && work_h * K == src_h, "vstab_gray_downscale: %dx%d is not %d x %dx%d", src_w, src_h, K, work_w, work_h);
        if (K == 2) LAUNCH_GRAY(2, work_h, gray);
        else LAUNCH_GRAY(4, work_h, gray);
    } else {
        // two passes: full-resolution gray into scratch, then the area resize
        const size_t full = (size_t)n * src_h * src_w;
        if (ctx->d_gray_tmp.reserve(full)) return 1;
        uint8_t* tmp = static_cast<uint8_t*>(ctx->d_gray_tmp.ptr);
        LAUNCH_GRAY(1, src_h, tmp);
        const long long out_items = (long long)n * work_h * work_w;
        if (fast) {
            hipLaunchKernelGGL(area_int_u8_kernel, dim3(grid_for(out_items)), dim3(256), 0, st, tmp, gray, n, src_h, src_w, work_h, work_w, isx, isy);
            VSTAB_HIP(hipGetLastError());
        } else {
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
        }
    }
#undef LAUNCH_GRAY
    if (frame_max) {
        // the host's copy of the maxima: written by the kernel itself into coherent host memory (no copy, no event on the stream)
        if (ctx->h_peaks_cap < n) {
            if (ctx->h_peaks) { VSTAB_HIP(hipStreamSynchronize(st)); VSTAB_HIP(hipHostFree(ctx->h_peaks)); ctx->h_peaks = nullptr; ctx->h_peaks_cap = 0; }
            const int cap = std::max(1024, n);
            void* hp = nullptr;
            VSTAB_HIP(hipHostMalloc(&hp, sizeof(float) * (size_t)cap, hipHostMallocMapped | hipHostMallocCoherent));
            void* dp = nullptr;
            VSTAB_HIP(hipHostGetDevicePointer(&dp, hp, 0));
            ctx->h_peaks = static_cast<float*>(hp); ctx->d_peaks_mirror = static_cast<float*>(dp); ctx->h_peaks_cap = cap;
        }
        if (!ctx->d_peaks_count) {   // the device-side count of finished frames (never reset: passes are told their target)
            VSTAB_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_peaks_count), 256));
            VSTAB_HIP(hipMemsetAsync(ctx->d_peaks_count, 0, 256, st));
        }
        ctx->peaks_target += (unsigned)n;
        ctx->peaks_frames = n;
        hipLaunchKernelGGL(frame_max_kernel, dim3((unsigned)n), dim3(256), 0, st, row_max, frame_max, range_rows, ctx->d_peaks_mirror,
                           reinterpret_cast<unsigned*>(ctx->d_status) + VSTAB_PEAKS_DONE_WORD, ctx->d_peaks_count, ctx->peaks_target);
        VSTAB_HIP(hipGetLastError());
    }
    return 0;
}

static int gray_check(const char* who, vstab_ctx* ctx, const float* frames, int n, int src_h, int src_w, int work_h, int work_w,
                      const uint8_t* gray)
{
    VSTAB_REQUIRE(ctx != nullptr, "%s: ctx is NULL", who);
    VSTAB_REQUIRE(frames && gray, "%s: NULL pointer argument", who);
    VSTAB_REQUIRE(n > 0 && src_h > 0 && src_w > 0 && work_h > 0 && work_w > 0, "%s: non-positive size", who);
    VSTAB_REQUIRE(work_h <= src_h && work_w <= src_w, "%s: working size %dx%d larger than source %dx%d", who, work_w, work_h, src_w, src_h);
    return 0;
}

extern "C" int vstab_gray_downscale(vstab_ctx* ctx, const float* frames, int n, int src_h, int src_w, int work_h,
                                    int work_w, uint8_t* gray)
{
    if (int rc = gray_check("vstab_gray_downscale", ctx, frames, n, src_h, src_w, work_h, work_w, gray)) return rc;
    return gray_run(ctx, frames, n, src_h, src_w, work_h, work_w, gray, nullptr);
}

extern "C" int vstab_gray_downscale_range(vstab_ctx* ctx, const float* frames, int n, int src_h, int src_w, int work_h,
                                          int work_w, uint8_t* gray, float* frame_max)
{
    if (int rc = gray_check("vstab_gray_downscale_range", ctx, frames, n, src_h, src_w, work_h, work_w, gray)) return rc;
    VSTAB_REQUIRE(frame_max != nullptr, "vstab_gray_downscale_range: frame_max is NULL");
    return gray_run(ctx, frames, n, src_h, src_w, work_h, work_w, gray, frame_max);
}

// The maxima of the latest vstab_gray_downscale_range call, on the host: waits for that call's frame_max_kernel only (a
// word in coherent host memory that the kernel's last workgroup sets to the pass's target), not for anything queued behind it.
extern "C" int vstab_last_frame_peaks(vstab_ctx* ctx, int n, float* out)
{
    VSTAB_REQUIRE(ctx != nullptr && out != nullptr, "vstab_last_frame_peaks: NULL argument");
    VSTAB_REQUIRE(ctx->h_peaks != nullptr && n == ctx->peaks_frames, "vstab_last_frame_peaks: no range pass over %d frames is pending", n);
    volatile unsigned* done = reinterpret_cast<volatile unsigned*>(ctx->h_status) + VSTAB_PEAKS_DONE_WORD;
    const unsigned target = ctx->peaks_target;
    // the gray pass of a 256 x 1080p clip takes ~1 ms; should the count never arrive (a lost launch), a stream
    // synchronisation after ~2 s turns the wait into the runtime's own error report
    for (long spins = 0; (int)(*done - target) < 0; spins++) {
        if (spins > 20000000L) {
            VSTAB_HIP(hipStreamSynchronize(ctx->stream));
            VSTAB_REQUIRE((int)(*done - target) >= 0, "vstab_last_frame_peaks: the range pass finished without reporting its %d frames", n);
            break;
        }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    memcpy(out, ctx->h_peaks, sizeof(float) * (size_t)n);
    return 0;
}

extern "C" int vstab_frame_range(vstab_ctx* ctx, const float* frames, int n, int h, int w, float* frame_max)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_frame_range: ctx is NULL");
    VSTAB_REQUIRE(frames && frame_max, "vstab_frame_range: NULL pointer argument");
    VSTAB_REQUIRE(n > 0 && h > 0 && w > 0, "vstab_frame_range: non-positive size");
    VSTAB_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // a frame is h*w*3 contiguous floats: cut it into pieces of a multiple of 4 floats that divides it evenly
    const long long per_frame = (long long)h * w * 3;
    long long piece = (long long)w * 3;   // one source row per workgroup
    bool vec = false;
    for (long long cand : {(long long)w * 3, (long long)w * 3 * 2, (long long)w * 3 * 4})
        if (cand % 4 == 0 && per_frame % cand == 0 && (reinterpret_cast<uintptr_t>(frames) % 16) == 0) { piece = cand; vec = true; break; }
    const long long pieces = per_frame / piece;
    VSTAB_REQUIRE((long long)n * pieces < 0x7fffffffLL && piece < 0x7fffffffLL, "vstab_frame_range: clip too large");
    if (ctx->d_range.reserve(sizeof(float) * (size_t)n * pieces)) return 1;
    float* row_max = static_cast<float*>(ctx->d_range.ptr);
    KernelTimer timer(ctx, "range");
    if (vec) hipLaunchKernelGGL(row_max_kernel<true>, dim3((unsigned)(n * pieces)), dim3(256), 0, st, frames, row_max, (int)piece);
    else hipLaunchKernelGGL(row_max_kernel<false>, dim3((unsigned)(n * pieces)), dim3(256), 0, st, frames, row_max, (int)piece);
    hipLaunchKernelGGL(frame_max_kernel, dim3((unsigned)n), dim3(256), 0, st, row_max, frame_max, (int)pieces, (float*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, 0u);
    VSTAB_HIP(hipGetLastError());
    return 0;
}

extern "C" int vstab_apply_value_range(vstab_ctx* ctx, const float* frames, int n, int h, int w, const float* frame_max, float* out)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_apply_value_range: ctx is NULL");
    VSTAB_REQUIRE(frames && frame_max && out, "vstab_apply_value_range: NULL pointer argument");
    VSTAB_REQUIRE(n > 0 && h > 0 && w > 0, "vstab_apply_value_range: non-positive size");
    VSTAB_HIP(hipSetDevice(ctx->device));
    const long long per_frame = (long long)h * w * 3, total = per_frame * n;
    hipLaunchKernelGGL(value_range_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, frames, out, frame_max, per_frame, total);
    VSTAB_HIP(hipGetLastError());
    return 0;
}
