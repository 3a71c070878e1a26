// vstab_traj.hip -- F6-F12: from the per-pair candidate fits to the warp's transform table.
//
//   vstab_trajectory        F7+F8 on the HOST: camera path (prefix sum), box-filter smoothing and strength blend, fp64
//                           (nodes/video_stabilizer_flow.py:356-371, nodes/stabilizer_utils.py:361-383: symmetric moving
//                           average, window derived from fps, edge padding).  A few thousand doubles: the host loop takes
//                           microseconds, a kernel cost two transfers and a synchronisation in the middle of the plan.
//   vstab_flow_plan_device  the WHOLE plan of a crop_and_pad clip on the device, speculatively: sticky-mode walk
//                           (flow.py:324-339), rescale + parameter deltas (flow.py:340-346), path / smoothing / blend,
//                           parameters -> float32 matrices (stabilizer_utils.py:327-358), bounding boxes and the common
//                           region (stabilizer_utils.py:1010-1034, flow.py:500-504), recentring shift and final = T @ M
//                           (flow.py:505-521), inverted into the warp kernels' table -- one small fp64 kernel behind the fit
//                           kernel on the same stream, so the warp starts without a host round trip.
//
// Why "speculatively": the reference forms atan2 / log / exp / cos / sin with Python's math module, i.e. the HOST's libm,
// and the reported motion (path, target_path, applied_matrix in the meta) must equal what it computes.  The device's fp64
// math library is not glibc's: it may differ in the last unit of a double, which survives the float32 cast of a matrix
// entry with probability ~1e-8.  So the host still computes the plan exactly (while the warp runs -- it needs the fit
// records for the meta anyway) and compares its float32 final matrices with the device's, bit for bit; a frame whose
// matrices differ is warped again with the host's matrix (flow_pipeline.py).  The outputs are therefore always those of
// the host plan; the device plan only removes the wait.  A perspective plan has no libm call at all (its parameters are the
// matrix entries), but its final = T @ M has sums of two inexact terms whose float32 rounding depends on how NumPy's matmul
// forms them (fused or not): the kernel takes the unfused form, and the verification catches the rare frame where that matters.
#include "vstab_internal.h"
#include <cmath>
#include <cstdlib>

namespace {

constexpr int PLAN_T = 1024;
constexpr size_t PLAN_LDS_MAX = 144 * 1024;

constexpr int PLAN_MAX_SEG = 64;

struct PlanArgs {
    const vstab_fit_record* rec;   // [pairs][3], indexed [pair][mode]; or a gathered table, see seg_*
    int segments;                  // 0: contiguous.  > 0: the all-gather's receive buffer as it is -- one block of seg_rows
    int seg_rows;                  //    records rows per rank, of which the first seg_start[r + 1] - seg_start[r] are valid
    int seg_start[PLAN_MAX_SEG + 1];
    int pairs, mode;               // requested model: VSTAB_MODE_TRANSLATION | VSTAB_MODE_SIMILARITY | VSTAB_MODE_PERSPECTIVE
    int rescale;                   // 1: F = f32(fl(up_i * M_ij) * down_j), 0: F = M
    double up[3], down[3];
    int window, do_smooth, camera_lock;
    double strength, width, height;
    float* final32;                // [frames][9]
    WarpXform* xf;                 // [frames]
    double* path;                  // [frames][p]
    double* target;                // [frames][p]
    double* region;                // [4]: x0, y0, x1, y1 of the common region (tests)
    int perturb;                   // tests (VSTAB_DEBUG_PLAN_PERTURB=frame): that frame's matrix is made wrong by one ulp
    int framing;                   // 0: crop_and_pad (recentre on the frames' common region), 1: expand (shift the frames' union to the origin)
    unsigned* zero;                // the planned warp's padded-pixel counts: zeroed here (vstab_flow_plan_zero_counts), or nullptr
    int zero_n;
};

// the records of pair i (three, indexed by mode)
__device__ __forceinline__ const vstab_fit_record* pair_records(const PlanArgs& a, int i)
{
    if (a.segments <= 0) return a.rec + (size_t)i * 3;
    int s = 0;
    while (s + 1 < a.segments && i >= a.seg_start[s + 1]) s++;
    return a.rec + ((size_t)s * a.seg_rows + (size_t)(i - a.seg_start[s])) * 3;
}

__device__ __forceinline__ double np_min(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }
__device__ __forceinline__ double np_max(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }

// block-wide reduction of four doubles with NumPy's NaN-propagating minimum / maximum: (max, max, min, min) -- the frames'
// common region -- or, with `uni`, (min, min, max, max) -- their union (expand framing)
__device__ void reduce_region(double v[4], double* s_red /*[4][PLAN_T / 64]*/, bool uni)
{
    for (int off = 32; off > 0; off >>= 1) {
        const double o0 = __shfl_xor(v[0], off), o1 = __shfl_xor(v[1], off), o2 = __shfl_xor(v[2], off), o3 = __shfl_xor(v[3], off);
        v[0] = uni ? np_min(v[0], o0) : np_max(v[0], o0); v[1] = uni ? np_min(v[1], o1) : np_max(v[1], o1);
        v[2] = uni ? np_max(v[2], o2) : np_min(v[2], o2); v[3] = uni ? np_max(v[3], o3) : np_min(v[3], o3);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = PLAN_T / 64;
    if (lane == 0) for (int k = 0; k < 4; k++) s_red[k * nw + wave] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 4; k++) {
            double acc = s_red[k * nw];
            for (int i = 1; i < nw; i++) acc = ((k < 2) != uni) ? np_max(acc, s_red[k * nw + i]) : np_min(acc, s_red[k * nw + i]);
            s_red[k * nw] = acc;
        }
    }
    __syncthreads();
    for (int k = 0; k < 4; k++) v[k] = s_red[k * nw];
}

// One workgroup.  LDS: the path [frames][p] fp64 (as the former trajectory kernel held it), the chosen mode per pair, and
// the reduction scratch.
__global__ __launch_bounds__(PLAN_T) void plan_kernel(PlanArgs a)
{
    extern __shared__ double s_path[];
    const int p = a.mode == VSTAB_MODE_PERSPECTIVE ? 8 : (a.mode == VSTAB_MODE_SIMILARITY ? 4 : 2);
    for (int i = threadIdx.x; i < a.zero_n; i += PLAN_T) a.zero[i] = 0u;   // (a fill kernel of its own cost the stream ~17 us before the warp)
    const int frames = a.pairs + 1, total = frames * p;
    double* s_red = s_path + total;                                  // [4][PLAN_T / 64]
    signed char* s_mode = reinterpret_cast<signed char*>(s_red + 4 * (PLAN_T / 64));   // [pairs]

    // ---- sticky active_mode walk (flow.py:324-339 + 153-210).  Sequential as written, but the active mode only ever
    // steps DOWN (it becomes the mode that was used), and below similarity there is only translation: with `first` = the
    // first pair whose similarity fit is unusable, every pair before it uses similarity and every pair from it on uses
    // translation if that fit is usable, else nothing (identity, reported as "translation", and the active mode stays at
    // translation).  One block-wide minimum instead of a pair-by-pair walk of one lane through global memory (measured:
    // ~0.2 ms for 255 pairs, most of what the device plan was meant to save).
    __shared__ int s_first;
    if (threadIdx.x == 0) s_first = a.pairs;
    __syncthreads();
    if (a.mode == VSTAB_MODE_PERSPECTIVE) {
        // three models: which fits of each pair are usable goes to LDS in parallel (one byte per pair), then one lane
        // replays the walk as written -- the active mode keeps winning until its fit is rejected, then the best usable
        // model below it is used and becomes the active one (none: identity, active = translation) -- ~50 cycles a pair
        for (int i = threadIdx.x; i < a.pairs; i += PLAN_T) {
            const vstab_fit_record* r = pair_records(a, i);
            int bits = 0;
            for (int m = 0; m < 3; m++) bits |= (r[m].computed != 0 && r[m].accepted != 0) ? (1 << m) : 0;
            s_mode[i] = (signed char)bits;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int active = VSTAB_MODE_PERSPECTIVE;
            for (int i = 0; i < a.pairs; i++) {
                const int bits = s_mode[i];
                int pick = -1;
                for (int m = active; m >= 0; m--)
                    if (bits & (1 << m)) { pick = m; break; }
                s_mode[i] = (signed char)pick;
                active = pick >= 0 ? pick : VSTAB_MODE_TRANSLATION;
            }
        }
        __syncthreads();
    } else {
    if (a.mode == VSTAB_MODE_SIMILARITY) {
        int mine = a.pairs;
        for (int i = threadIdx.x; i < a.pairs; i += PLAN_T) {
            const vstab_fit_record& r = pair_records(a, i)[VSTAB_MODE_SIMILARITY];
            if (!(r.computed != 0 && r.accepted != 0)) { mine = i; break; }   // this thread's pairs ascend
        }
        if (mine < a.pairs) atomicMin(&s_first, mine);
    } else if (threadIdx.x == 0) {
        s_first = 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.pairs; i += PLAN_T) {
        int pick = VSTAB_MODE_SIMILARITY;
        if (i >= s_first) {
            const vstab_fit_record& r = pair_records(a, i)[VSTAB_MODE_TRANSLATION];
            pick = (r.computed != 0 && r.accepted != 0) ? VSTAB_MODE_TRANSLATION : -1;
        }
        s_mode[i] = (signed char)pick;
    }
    __syncthreads();
    }
    // ---- rescale to full resolution + parameter deltas of the REQUESTED model (flow.py:340-346), into the path array
    for (int t = threadIdx.x; t < p; t += PLAN_T) s_path[t] = 0.0;
    for (int i = threadIdx.x; i < a.pairs; i += PLAN_T) {
        float M[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
        const int pick = s_mode[i];
        if (pick >= 0) {
            const float* src = pair_records(a, i)[pick].matrix;
            for (int k = 0; k < 9; k++) M[k] = src[k];
        }
        float F[9];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                if (a.rescale) {
                    const double t = a.up[r] * (double)M[r * 3 + c];
                    F[r * 3 + c] = (float)(t * a.down[c]);
                } else {
                    F[r * 3 + c] = M[r * 3 + c];
                }
            }
        double* P = s_path + (size_t)(i + 1) * p;
        if (a.mode == VSTAB_MODE_PERSPECTIVE) {   // the matrix entries themselves, identity removed in float32 (vstab_transitions_to_params)
            const float d0 = F[0] - 1.0f, d4 = F[4] - 1.0f;
            P[0] = (double)d0; P[1] = (double)F[1]; P[2] = (double)F[2]; P[3] = (double)F[3];
            P[4] = (double)d4; P[5] = (double)F[5]; P[6] = (double)F[6]; P[7] = (double)F[7];
            continue;
        }
        P[0] = (double)F[2]; P[1] = (double)F[5];
        if (a.mode == VSTAB_MODE_SIMILARITY) {
            const float fa = F[0], fc = F[3];
            const float aa = fa * fa, cc = fc * fc;
            const float sq = aa + cc;
            const double sq64 = (1e-10f > sq) ? 1e-10 : (double)sq;
            P[2] = atan2((double)fc, (double)fa);
            P[3] = log(sqrt(sq64));
        }
    }
    __syncthreads();
    // ---- path = prefix sum of the deltas (flow.py:356-358), one lane per parameter, in order
    //      The additions stay sequential (np.cumsum's order); a batch of deltas is read from LDS before its chain of adds
    //      starts -- element by element the loop pays an LDS round trip per frame (~80 cycles: 70 us of a 2048-frame plan).
    if ((int)threadIdx.x < p) {
        const int c = threadIdx.x;
        constexpr int B = 16;
        double acc = 0.0;
        int i = 1;
        for (; i + B <= frames; i += B) {
            double v[B];
#pragma unroll
            for (int k = 0; k < B; k++) v[k] = s_path[(i + k) * p + c];
#pragma unroll
            for (int k = 0; k < B; k++) { acc = acc + v[k]; v[k] = acc; }
#pragma unroll
            for (int k = 0; k < B; k++) s_path[(i + k) * p + c] = v[k];
        }
        for (; i < frames; i++) {
            acc = acc + s_path[i * p + c];
            s_path[i * p + c] = acc;
        }
    }
    __syncthreads();
    // ---- per frame: smoothed target, correction, float32 matrix, corners (same arithmetic as vstab_trajectory /
    //      vstab_params_to_matrices / vstab_bounding_boxes on the host)
    const int pad = a.window / 2;
    const double kv = 1.0 / (double)a.window;
    const bool uni = a.framing == 1;
    // crop_and_pad: max of mins.x, mins.y; min of maxs.x, maxs.y (the common region).  expand: min of mins, max of maxs (the union)
    double reg[4] = {uni ? INFINITY : -INFINITY, uni ? INFINITY : -INFINITY, uni ? -INFINITY : INFINITY, uni ? -INFINITY : INFINITY};
    float A[9];
    for (int i = threadIdx.x; i < frames; i += PLAN_T) {
        double diff[8];
        for (int c = 0; c < p; c++) {
            const double cur = s_path[i * p + c];
            double sm = cur;
            if (a.do_smooth) {
                double acc = 0.0;
                for (int k = 0; k < a.window; k++) {
                    int s = i + k - pad;
                    s = s < 0 ? 0 : (s > frames - 1 ? frames - 1 : s);
                    acc += s_path[s * p + c] * kv;
                }
                sm = acc;
            }
            const double tgt = a.camera_lock ? 0.0 : cur + a.strength * (sm - cur);
            a.path[(size_t)i * p + c] = cur;
            a.target[(size_t)i * p + c] = tgt;
            diff[c] = tgt - cur;
        }
        A[6] = 0.f; A[7] = 0.f; A[8] = 1.f;
        if (a.mode == VSTAB_MODE_PERSPECTIVE) {
            A[0] = (float)(diff[0] + 1.0); A[1] = (float)diff[1]; A[2] = (float)diff[2];
            A[3] = (float)diff[3]; A[4] = (float)(diff[4] + 1.0); A[5] = (float)diff[5];
            A[6] = (float)diff[6]; A[7] = (float)diff[7];
        } else if (a.mode == VSTAB_MODE_SIMILARITY) {
            const double s = exp(diff[3]), ct = cos(diff[2]), st = sin(diff[2]);
            const double sc = s * ct, ss = s * st;
            A[0] = (float)sc; A[1] = (float)(-ss); A[2] = (float)diff[0];
            A[3] = (float)ss; A[4] = (float)sc; A[5] = (float)diff[1];
        } else {
            A[0] = 1.f; A[1] = 0.f; A[2] = (float)diff[0];
            A[3] = 0.f; A[4] = 1.f; A[5] = (float)diff[1];
        }
        for (int k = 0; k < 9; k++) a.final32[(size_t)i * 9 + k] = A[k];   // the apply matrix, recentred below
        const double cx[4] = {0.0, a.width, 0.0, a.width}, cy[4] = {0.0, 0.0, a.height, a.height};
        double xs[4], ys[4];
        for (int j = 0; j < 4; j++) {
            double q[3];
            for (int r = 0; r < 3; r++) {
                const double t0 = (double)A[r * 3 + 0] * cx[j];
                const double t1 = (double)A[r * 3 + 1] * cy[j];
                const double t2 = (double)A[r * 3 + 2] * 1.0;
                q[r] = (t0 + t1) + t2;
            }
            xs[j] = q[0] / q[2];
            ys[j] = q[1] / q[2];
        }
        const double mnx = np_min(np_min(xs[0], xs[1]), np_min(xs[2], xs[3])), mny = np_min(np_min(ys[0], ys[1]), np_min(ys[2], ys[3]));
        const double mxx = np_max(np_max(xs[0], xs[1]), np_max(xs[2], xs[3])), mxy = np_max(np_max(ys[0], ys[1]), np_max(ys[2], ys[3]));
        reg[0] = uni ? np_min(reg[0], mnx) : np_max(reg[0], mnx);
        reg[1] = uni ? np_min(reg[1], mny) : np_max(reg[1], mny);
        reg[2] = uni ? np_max(reg[2], mxx) : np_min(reg[2], mxx);
        reg[3] = uni ? np_max(reg[3], mxy) : np_min(reg[3], mxy);
    }
    // ---- common region over all frames -> recentring shift (flow.py:501-511); or their union -> the shift that brings it to the
    //      origin (stabilizer_utils.py:386-406, flow.py:530-533; the canvas size is the host's to form from `region`): float32
    //      like the reference's matrix either way
    reduce_region(reg, s_red, uni);
    if (threadIdx.x == 0) for (int k = 0; k < 4; k++) a.region[k] = reg[k];
    const float off_x = uni ? (float)(-reg[0]) : (float)(a.width * 0.5 - (reg[0] + reg[2]) * 0.5);
    const float off_y = uni ? (float)(-reg[1]) : (float)(a.height * 0.5 - (reg[1] + reg[3]) * 0.5);
    // ---- final = T @ A in float32 (affine A: every sum has one inexact term; perspective A: off * A[6 + c] is rounded before it
    //      is added, the unfused form), inverted into the warp's table
    for (int i = threadIdx.x; i < frames; i += PLAN_T) {
        float* Fm = a.final32 + (size_t)i * 9;
        float R[9];
        for (int c = 0; c < 3; c++) {
            R[c] = (1.f * Fm[c] + 0.f * Fm[3 + c]) + off_x * Fm[6 + c];
            R[3 + c] = (0.f * Fm[c] + 1.f * Fm[3 + c]) + off_y * Fm[6 + c];
            R[6 + c] = (0.f * Fm[c] + 0.f * Fm[3 + c]) + 1.f * Fm[6 + c];
        }
        if (i == a.perturb) R[2] = __uint_as_float(__float_as_uint(R[2]) + 1u);   // a disagreement for the verification to catch
        for (int k = 0; k < 9; k++) Fm[k] = R[k];
        vstab_fill_xform(R, a.xf + i);
    }
}

int smoothing_window(double smooth, double fps)   // stabilizer_utils.py:363-375
{
    fps = fps > 1.0 ? fps : 1.0;
    const double min_seconds = 3.0 / 16.0, max_seconds = 13.0 / 16.0;
    const double window_seconds = min_seconds + smooth * (max_seconds - min_seconds);
    int window = (int)std::nearbyint(window_seconds * fps);  // Python round(): half to even
    window = window > 3 ? window : 3;
    if (window % 2 == 0) window += 1;
    return window;
}

struct PlanLayout { size_t final32, xf, path, target, region, total; };
PlanLayout plan_layout(int frames, int p)
{
    PlanLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~size_t(255); return o; };
    L.final32 = take(sizeof(float) * 9 * (size_t)frames);
    L.path = take(sizeof(double) * (size_t)frames * p);
    L.target = take(sizeof(double) * (size_t)frames * p);
    L.region = take(sizeof(double) * 4);
    L.xf = take(sizeof(WarpXform) * (size_t)frames);
    L.total = off;
    return L;
}

}  // namespace

extern "C" int vstab_trajectory(vstab_ctx* ctx, const double* deltas, int n, int p, double smooth, double fps,
                                double strength, int camera_lock, double* path, double* target)
{
    (void)ctx;   // host arithmetic (kept in the signature: the entry point predates the host form)
    VSTAB_REQUIRE(deltas && path && target, "vstab_trajectory: NULL pointer argument");
    VSTAB_REQUIRE(n >= 2 && p >= 1 && p <= 8, "vstab_trajectory: unsupported shape n=%d p=%d", n, p);
    smooth = smooth < 0.0 ? 0.0 : (smooth > 1.0 ? 1.0 : smooth);
    strength = strength < 0.0 ? 0.0 : (strength > 1.0 ? 1.0 : strength);
    const int do_smooth = !(smooth <= 0.0 || n <= 2);
    const int window = smoothing_window(smooth, fps);
    // the operation order of plan_kernel: sequential prefix sum per parameter; window sum in tap order, each tap scaled
    for (int c = 0; c < p; c++) {
        double acc = 0.0;
        path[c] = 0.0;
        for (int i = 1; i < n; i++) {
            acc = acc + deltas[(size_t)(i - 1) * p + c];
            path[(size_t)i * p + c] = acc;
        }
    }
    const int pad = window / 2;
    const double kv = 1.0 / (double)window;
    for (int i = 0; i < n; i++)
        for (int c = 0; c < p; c++) {
            const double cur = path[(size_t)i * p + c];
            double sm = cur;
            if (do_smooth) {
                double acc = 0.0;
                for (int k = 0; k < window; k++) {
                    int s = i + k - pad;
                    s = s < 0 ? 0 : (s > n - 1 ? n - 1 : s);
                    acc += path[(size_t)s * p + c] * kv;
                }
                sm = acc;
            }
            target[(size_t)i * p + c] = camera_lock ? 0.0 : cur + strength * (sm - cur);
        }
    return 0;
}

extern "C" int vstab_flow_plan_device(vstab_ctx* ctx, const vstab_fit_record* d_records, int pairs, int requested_mode,
                                      const double* up, const double* down, double smooth, double fps, double strength,
                                      int camera_lock, int width, int height, int segments, const int* seg_pairs, int seg_rows, int framing)
{
    // a registered count array belongs to THIS call only, whether it succeeds or not (a refused call must not leave the
    // pointer behind for a later kernel to write through)
    unsigned* zero_ptr = ctx ? ctx->plan_zero_ptr : nullptr;
    const int zero_n = ctx ? ctx->plan_zero_n : 0;
    if (ctx) { ctx->plan_zero_ptr = nullptr; ctx->plan_zero_n = 0; ctx->plan_zeroed_ptr = nullptr; }
    VSTAB_REQUIRE(ctx != nullptr && d_records != nullptr, "vstab_flow_plan_device: NULL argument");
    VSTAB_REQUIRE(pairs >= 1, "vstab_flow_plan_device: needs at least one transition");
    VSTAB_REQUIRE(requested_mode == VSTAB_MODE_TRANSLATION || requested_mode == VSTAB_MODE_SIMILARITY || requested_mode == VSTAB_MODE_PERSPECTIVE,
                  "vstab_flow_plan_device: unknown model %d", requested_mode);
    VSTAB_REQUIRE((up == nullptr) == (down == nullptr), "vstab_flow_plan_device: up and down come together");
    VSTAB_REQUIRE(width > 0 && height > 0, "vstab_flow_plan_device: non-positive frame size");
    VSTAB_REQUIRE(framing == 0 || framing == 1, "vstab_flow_plan_device: framing %d (0 = crop_and_pad, 1 = expand; crop is the host's keep_fov solver)", framing);
    const int p = requested_mode == VSTAB_MODE_PERSPECTIVE ? 8 : (requested_mode == VSTAB_MODE_SIMILARITY ? 4 : 2), frames = pairs + 1;
    const size_t lds = sizeof(double) * ((size_t)frames * p + 4 * (PLAN_T / 64)) + (((size_t)pairs + 15) & ~size_t(15));
    VSTAB_REQUIRE(lds <= PLAN_LDS_MAX, "vstab_flow_plan_device: a clip of %d frames does not fit the plan kernel's LDS", frames);
    VSTAB_HIP(hipSetDevice(ctx->device));
    const PlanLayout L = plan_layout(frames, p);
    if (ctx->d_plan.reserve(L.total)) return 1;
    ctx->h_plan.pinned_host = true;
    if (ctx->h_plan.reserve(L.xf)) return 1;   // everything but the warp table comes back
    if (!ctx->ev_plan_done) VSTAB_HIP(hipEventCreateWithFlags(&ctx->ev_plan_done, hipEventDisableTiming));
    // a previous plan whose results nobody collected (a discarded speculative run): its download from d_plan, on the side
    // stream, has to be over before this call's kernel writes d_plan again (a host wait on an event long past: no stream cost)
    else if (ctx->plan_frames > 0) VSTAB_HIP(hipEventSynchronize(ctx->ev_plan_done));
    char* base = static_cast<char*>(ctx->d_plan.ptr);
    PlanArgs a{};
    a.rec = d_records; a.pairs = pairs; a.mode = requested_mode;
    VSTAB_REQUIRE(segments >= 0 && segments <= PLAN_MAX_SEG, "vstab_flow_plan_device: %d segments (at most %d)", segments, PLAN_MAX_SEG);
    a.segments = segments; a.seg_rows = seg_rows;
    if (segments > 0) {
        VSTAB_REQUIRE(seg_pairs != nullptr && seg_rows >= 1, "vstab_flow_plan_device: a segmented table needs seg_pairs and seg_rows");
        int acc = 0;
        for (int r = 0; r < segments; r++) {
            VSTAB_REQUIRE(seg_pairs[r] >= 0 && seg_pairs[r] <= seg_rows, "vstab_flow_plan_device: segment %d holds %d pairs of %d rows", r, seg_pairs[r], seg_rows);
            a.seg_start[r] = acc;
            acc += seg_pairs[r];
        }
        a.seg_start[segments] = acc;
        VSTAB_REQUIRE(acc == pairs, "vstab_flow_plan_device: the segments hold %d pairs, the clip has %d", acc, pairs);
    }
    a.rescale = up != nullptr;
    for (int i = 0; i < 3; i++) { a.up[i] = up ? up[i] : 1.0; a.down[i] = down ? down[i] : 1.0; }
    smooth = smooth < 0.0 ? 0.0 : (smooth > 1.0 ? 1.0 : smooth);
    a.strength = strength < 0.0 ? 0.0 : (strength > 1.0 ? 1.0 : strength);
    a.do_smooth = !(smooth <= 0.0 || frames <= 2);
    a.window = smoothing_window(smooth, fps);
    a.camera_lock = camera_lock; a.width = (double)width; a.height = (double)height;
    a.final32 = reinterpret_cast<float*>(base + L.final32);
    a.xf = reinterpret_cast<WarpXform*>(base + L.xf);
    a.path = reinterpret_cast<double*>(base + L.path);
    a.target = reinterpret_cast<double*>(base + L.target);
    a.region = reinterpret_cast<double*>(base + L.region);
    a.perturb = -1;
    a.framing = framing;
    a.zero = zero_ptr; a.zero_n = zero_ptr ? zero_n : 0;
    ctx->plan_zeroed_ptr = zero_ptr;
#ifdef VSTAB_TEST_HOOKS   // fault injector of the test build (lib/libvstab_hooks.so); the shipped library reads no such variable
    if (const char* e = getenv("VSTAB_DEBUG_PLAN_PERTURB")) a.perturb = atoi(e);
#endif
    if (lds > 64 * 1024)
        VSTAB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(plan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(PLAN_T), lds, ctx->stream, a);
    VSTAB_HIP(hipGetLastError());
    // The host's copies -- this rank's fit records, if their download is still owed, and everything of the plan but the warp
    // table -- go out on a stream of their own behind ONE event recorded here: the warp that follows on the call's stream
    // starts without waiting for a copy engine, and the host has both while the warp runs.
    if (!ctx->side_stream) {
        VSTAB_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        VSTAB_HIP(hipEventCreateWithFlags(&ctx->ev_side, hipEventDisableTiming));
    }
    VSTAB_HIP(hipEventRecord(ctx->ev_side, ctx->stream));
    VSTAB_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->ev_side, 0));
    if (ctx->fit_copy_bytes) {
        VSTAB_HIP(hipMemcpyAsync(ctx->h_fit.ptr, ctx->d_fit.ptr, ctx->fit_copy_bytes, hipMemcpyDeviceToHost, ctx->side_stream));
        VSTAB_HIP(hipEventRecord(ctx->ev_fit_done, ctx->side_stream));
        ctx->fit_copy_bytes = 0;
    }
    VSTAB_HIP(hipMemcpyAsync(ctx->h_plan.ptr, base, L.xf, hipMemcpyDeviceToHost, ctx->side_stream));
    VSTAB_HIP(hipEventRecord(ctx->ev_plan_done, ctx->side_stream));
    ctx->plan_frames = frames; ctx->plan_params = p;
    return 0;
}

// Registers the planned warp's padded-pixel count array with the NEXT vstab_flow_plan_device call: its kernel zeroes the
// array, and vstab_warp_batch_planned, handed the same pointer, skips its own fill (one launch less between plan and warp).
extern "C" int vstab_flow_plan_zero_counts(vstab_ctx* ctx, uint32_t* pad_count, int n)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_flow_plan_zero_counts: ctx is NULL");
    VSTAB_REQUIRE(pad_count != nullptr && n > 0, "vstab_flow_plan_zero_counts: needs a count array of at least one frame");
    ctx->plan_zero_ptr = pad_count; ctx->plan_zero_n = n;
    return 0;
}

extern "C" int vstab_flow_plan_result(vstab_ctx* ctx, int frames, float* final32, double* path, double* target, double* region)
{
    VSTAB_REQUIRE(ctx != nullptr, "vstab_flow_plan_result: ctx is NULL");
    VSTAB_REQUIRE(ctx->plan_frames > 0 && frames == ctx->plan_frames, "vstab_flow_plan_result: no device plan of %d frames is pending", frames);
    VSTAB_HIP(hipEventSynchronize(ctx->ev_plan_done));
    const PlanLayout L = plan_layout(frames, ctx->plan_params);
    const char* base = static_cast<const char*>(ctx->h_plan.ptr);
    if (final32) memcpy(final32, base + L.final32, sizeof(float) * 9 * (size_t)frames);
    if (path) memcpy(path, base + L.path, sizeof(double) * (size_t)frames * ctx->plan_params);
    if (target) memcpy(target, base + L.target, sizeof(double) * (size_t)frames * ctx->plan_params);
    if (region) memcpy(region, base + L.region, sizeof(double) * 4);
    return 0;
}

// the warp table of frames [first, first + n) of the last device plan (vstab_warp.hip)
const WarpXform* vstab_plan_xforms(vstab_ctx* ctx, int first, int n)
{
    if (!ctx || ctx->plan_frames <= 0 || first < 0 || n < 1 || first + n > ctx->plan_frames) return nullptr;
    const PlanLayout L = plan_layout(ctx->plan_frames, ctx->plan_params);
    return reinterpret_cast<const WarpXform*>(static_cast<const char*>(ctx->d_plan.ptr) + L.xf) + first;
}
