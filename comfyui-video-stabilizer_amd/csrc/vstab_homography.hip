// vstab_homography.hip -- perspective branch of F5: cv2.findHomography(RANSAC, 2.5 px, 2000, 0.992)
// (nodes/video_stabilizer_flow.py:162-175 of the reference), one 256-thread block per frame pair.
//
// OpenCV's pipeline is kept: 4-point minimal samples (same RNG stream, same duplicate / checkSubset
// rejection, so the same samples are drawn), normalised DLT solved through the 9x9 symmetric Jacobi
// eigen-decomposition, f32 reprojection-error test, adaptive iteration count; then a DLT over all inliers
// and 10 Levenberg-Marquardt iterations on the 8 parameters.  Data-parallel mapping: one lane replays the
// RNG and draws a batch of 16 samples, 16 lanes run the 16 Jacobi solves, all lanes score the batch in one
// pass over the points, one lane replays the "best so far / update niters" loop.  The normal equations of
// the refit and of every LM step are fp64 block reductions; the 8x8 solves run on one lane (Gaussian
// elimination instead of OpenCV's eigen-solve: same solution to ~1e-13).
#include "vstab_internal.h"
#include <cmath>

#define VSTAB_HAVE_HOMOGRAPHY 1

namespace {

constexpr int HT = 256;
constexpr int HBATCH = 16;

struct Rng { unsigned long long state; };
__device__ __forceinline__ unsigned rng_next(Rng& r)
{
    r.state = (unsigned long long)(unsigned)r.state * 4164903690ULL + (unsigned)(r.state >> 32);
    return (unsigned)r.state;
}
__device__ __forceinline__ int rng_uniform(Rng& r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (unsigned)(b - a) + a); }

__device__ int update_num_iters(double p, double ep, int modelPoints, int maxIters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > 2.2250738585072014e-308 ? 1. - p : 2.2250738585072014e-308;
    double denom = 1. - pow(1. - ep, (double)modelPoints);
    if (denom < 2.2250738585072014e-308) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)__builtin_rint(num / denom);
}

// gw > 0: stride-`step` grid over a flow field (F = [gh][gw][2] displacements, flow.py:141-147);
// gw == 0: explicit point pairs (F = [cap][4]: prev.x, prev.y, next.x, next.y; untracked points carry NaN)
__device__ __forceinline__ void load_point(const float* __restrict__ F, int gw, int step, int g, float& px, float& py, float& cx, float& cy)
{
    if (gw == 0) {
        const float4 v = *reinterpret_cast<const float4*>(F + (size_t)g * 4);
        px = v.x; py = v.y; cx = v.z; cy = v.w;
        return;
    }
    const int gy = g / gw, gx = g - gy * gw;
    px = (float)(gx * step);
    py = (float)(gy * step);
    cx = px + F[(size_t)g * 2];
    cy = py + F[(size_t)g * 2 + 1];
}

// JacobiImpl_ of OpenCV (core/src/lapack.cpp) for a symmetric n x n fp64 matrix; eigenvalues sorted descending,
// eigenvectors are the rows of V.
// Workspace of one eigen-solve / one 8x8 elimination.  It lives in LDS: the solvers index their matrices with
// run-time pivots, which would otherwise put them in scratch memory, and since round 5 the sixteen lanes of a group
// share one solve through it.
struct JacWs {
    double A[81];
    double V[81];
    double W[9];
    int indR[9], indC[9];
    double norm[8];   // cmx, cmy, cMx, cMy, smx, smy, sMx, sMy of the solve in progress (written by the group's lane 0)
    int go, pad_;     // lane 0's verdict: the normalisation is not degenerate, the group runs the eigen-solve
    double pad;
};

// Sixteen lanes of one wavefront (an aligned group: lane g = threadIdx.x & 15) work on one JacWs together.  They run in
// lockstep and exchange everything through LDS, whose operations a wavefront issues and completes in order: between a phase
// that writes and a phase that reads only the COMPILER has to be kept from moving accesses across -- no s_barrier, so
// groups of one wavefront may sit in different iterations (or have left the loop) without harm.
__device__ __forceinline__ void group_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The eigen-solve by a group of 16 lanes (n = 9; the serial form is oracle/vo_fit.c's jacobi_eigen).  OpenCV's algorithm is
// sequential in its ROTATIONS; inside one
// rotation the sixteen element pairs it turns -- seven of A (the rows / columns k and l without their crossing), nine of V
// (rows k and l) -- are independent, and so are the four index scans that follow: one pair, one scan per lane, the same
// fp64 operations on the same values as the serial loop, so the same bits.  The pivot search and c / s / t are computed by every lane redundantly (LDS broadcast reads).  A phase
// profile had the 16 serial solves of a batch at 535-553 us (one lane each, diverging) and the refit's at 269 us of the
// kernel's 1.17 ms per pair (profiles/r05_homography.md).
__device__ void jacobi_eigen_g16(JacWs& ws, int g)
{
    constexpr int n = 9;
    const double eps = 2.220446049250313e-16;
    double* A = ws.A; double* V = ws.V; double* W = ws.W;
    int* indR = ws.indR; int* indC = ws.indC;
    for (int e = g; e < n * n; e += 16) V[e] = (e / n == e % n) ? 1.0 : 0.0;
    if (g < n) {
        const int k = g;
        W[k] = A[(n + 1) * k];
        if (k < n - 1) {
            int m = k + 1;
            double mv = fabs(A[n * k + m]);
            for (int i = k + 2; i < n; i++) { const double val = fabs(A[n * k + i]); if (mv < val) mv = val, m = i; }
            indR[k] = m;
        }
        if (k > 0) {
            int m = 0;
            double mv = fabs(A[k]);
            for (int i = 1; i < k; i++) { const double val = fabs(A[n * i + k]); if (mv < val) mv = val, m = i; }
            indC[k] = m;
        }
    }
    group_sync();
    const int maxIters = n * n * 30;
    for (int iters = 0; iters < maxIters; iters++) {
        int k = 0;
        double mv = fabs(A[indR[0]]);
        for (int i = 1; i < n - 1; i++) { const double val = fabs(A[n * i + indR[i]]); if (mv < val) mv = val, k = i; }
        int l = indR[k];
        for (int i = 1; i < n; i++) { const double val = fabs(A[n * indC[i] + i]); if (mv < val) mv = val, k = indC[i], l = i; }
        const double p = A[n * k + l];
        if (fabs(p) <= eps) break;                       // uniform over the group
        const double y = (W[l] - W[k]) * 0.5;
        double t = fabs(y) + hypot(p, y);
        double sn = hypot(p, t);
        const double c = t / sn;
        sn = p / sn;
        t = (p / t) * p;
        if (y < 0) sn = -sn, t = -t;
        group_sync();                                    // every lane has read A[k][l], W[k], W[l] and the index tables
        if (g == 0) { A[n * k + l] = 0; W[k] -= t; W[l] += t; }
        {
            double *e0, *e1;
            if (g < n - 2) {                             // the i-th index of 0 .. n-1 that is neither k nor l (k < l always)
                int i = g;
                if (i >= k) i++;
                if (i >= l) i++;
                if (i < k) { e0 = &A[n * i + k]; e1 = &A[n * i + l]; }
                else if (i < l) { e0 = &A[n * k + i]; e1 = &A[n * i + l]; }
                else { e0 = &A[n * k + i]; e1 = &A[n * l + i]; }
            } else {
                const int i = g - (n - 2);
                e0 = &V[n * k + i]; e1 = &V[n * l + i];
            }
            const double a0 = *e0, b0 = *e1;
            *e0 = a0 * c - b0 * sn;
            *e1 = a0 * sn + b0 * c;
        }
        group_sync();
        if (g < 4) {
            const int idx = g < 2 ? k : l;
            if ((g & 1) == 0) {
                if (idx < n - 1) {
                    int m = idx + 1;
                    double mx = fabs(A[n * idx + m]);
                    for (int i = idx + 2; i < n; i++) { const double val = fabs(A[n * idx + i]); if (mx < val) mx = val, m = i; }
                    indR[idx] = m;
                }
            } else if (idx > 0) {
                int m = 0;
                double mx = fabs(A[idx]);
                for (int i = 1; i < idx; i++) { const double val = fabs(A[n * i + idx]); if (mx < val) mx = val, m = i; }
                indC[idx] = m;
            }
        }
        group_sync();
    }
    // eigenvalues descending, the rows of V with them (selection sort: n - 1 sequential steps, each swap spread over lanes)
    for (int k = 0; k < n - 1; k++) {
        int m = k;
        for (int i = k + 1; i < n; i++)
            if (W[m] < W[i]) m = i;
        group_sync();
        if (k != m) {
            if (g < n) { const double tv = V[n * m + g]; V[n * m + g] = V[n * k + g]; V[n * k + g] = tv; }
            else if (g == n) { const double tw = W[m]; W[m] = W[k]; W[k] = tw; }
        }
        group_sync();
    }
}

// H from the normalisation parameters and LtL (HomographyEstimatorCallback::runKernel after the accumulation)
// ws.A holds the upper triangle of LtL, ws.norm the normalisation: mirror, eigen-solve (the group), compose H (lane 0)
__device__ void homography_from_ltl_g16(JacWs& ws, int g, double* H /* lane 0's result */)
{
    double* LtL = ws.A;
    for (int e = g; e < 81; e += 16) {
        const int j = e / 9, k = e - j * 9;
        if (k < j) LtL[j * 9 + k] = LtL[k * 9 + j];
    }
    group_sync();
    jacobi_eigen_g16(ws, g);
    if (g == 0) {
        const double cmx = ws.norm[0], cmy = ws.norm[1], cMx = ws.norm[2], cMy = ws.norm[3];
        const double smx = ws.norm[4], smy = ws.norm[5], sMx = ws.norm[6], sMy = ws.norm[7];
        const double* H0 = ws.V + 72;
        const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
        const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
        double T[9], R[9];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += invHnorm[r * 3 + k] * H0[k * 3 + c];
                T[r * 3 + c] = s;
            }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += T[r * 3 + k] * Hnorm2[k * 3 + c];
                R[r * 3 + c] = s;
            }
        const double sc = 1. / R[8];
        for (int i = 0; i < 9; i++) H[i] = R[i] * sc;
    }
}

// lane 0 of a group: normalisation + LtL of a 4-point sample into ws (runKernel's accumulation); false: degenerate
__device__ bool homography_4pt_setup(JacWs& ws, const float* M, const float* m)
{
    double* LtL = ws.A;
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    const int count = 4;
    for (int i = 0; i < count; i++) { cmx += m[i * 2]; cmy += m[i * 2 + 1]; cMx += M[i * 2]; cMy += M[i * 2 + 1]; }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; i++) {
        smx += fabs(m[i * 2] - cmx); smy += fabs(m[i * 2 + 1] - cmy);
        sMx += fabs(M[i * 2] - cMx); sMy += fabs(M[i * 2 + 1] - cMy);
    }
    const double eps = 2.220446049250313e-16;
    if (fabs(smx) < eps || fabs(smy) < eps || fabs(sMx) < eps || fabs(sMy) < eps) return false;
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    for (int i = 0; i < 81; i++) LtL[i] = 0;
    for (int i = 0; i < count; i++) {
        double x = (m[i * 2] - cmx) * smx, y = (m[i * 2 + 1] - cmy) * smy;
        double X = (M[i * 2] - cMx) * sMx, Y = (M[i * 2 + 1] - cMy) * sMy;
        double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++)
            for (int k = j; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    ws.norm[0] = cmx; ws.norm[1] = cmy; ws.norm[2] = cMx; ws.norm[3] = cMy;
    ws.norm[4] = smx; ws.norm[5] = smy; ws.norm[6] = sMx; ws.norm[7] = sMy;
    return true;
}

__device__ bool collinear(const float* p, int count)
{
    const int i = count - 1;
    for (int j = 0; j < i; j++) {
        double dx1 = p[j * 2] - p[i * 2], dy1 = p[j * 2 + 1] - p[i * 2 + 1];
        for (int k = 0; k < j; k++) {
            double dx2 = p[k * 2] - p[i * 2], dy2 = p[k * 2 + 1] - p[i * 2 + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
        }
    }
    return false;
}
__device__ double det3pts(const float* p, int t0, int t1, int t2)
{
    const double a0 = p[t0 * 2], a1 = p[t0 * 2 + 1], a3 = p[t1 * 2], a4 = p[t1 * 2 + 1], a6 = p[t2 * 2], a7 = p[t2 * 2 + 1];
    return a0 * (a4 * 1. - 1. * a7) - a1 * (a3 * 1. - 1. * a6) + 1. * (a3 * a7 - a4 * a6);
}
__device__ bool check_subset(const float* ms1, const float* ms2)
{
    if (collinear(ms1, 4) || collinear(ms2, 4)) return false;
    const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    int negative = 0;
    for (int i = 0; i < 4; i++) negative += det3pts(ms1, tt[i][0], tt[i][1], tt[i][2]) * det3pts(ms2, tt[i][0], tt[i][1], tt[i][2]) < 0;
    return negative == 0 || negative == 4;
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch)
{
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    const T total = ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
    __syncthreads();
    return total;
}

// N block sums at once: the same wavefront shuffle tree and the same ((w0 + w1) + w2) + w3 order as N calls of block_sum
// -- the same bits -- behind two barriers instead of 3 N (the normal equations of an LM step are 44 sums: 132 barriers).
// part: LDS [HT / 64][N]; out: LDS [N], valid for every thread on return.
template <int N>
__device__ __forceinline__ void block_sum_many(const double (&acc)[N], double* part, double* out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double v = acc[i];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s);
        if (lane == 0) part[wave * N + i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += HT) out[i] = ((part[i] + part[N + i]) + part[2 * N + i]) + part[3 * N + i];
    __syncthreads();
}

// solve the symmetric n x n system A x = b (n <= 8) by Gaussian elimination with partial pivoting
__device__ bool solve_sym(double (*A)[9] /* 8 rows of LDS workspace */, const double* A_in, const double* b, int n, double* x)
{
    for (int r = 0; r < n; r++) {
        for (int c = 0; c < n; c++) A[r][c] = A_in[r * n + c];
        A[r][n] = b[r];
    }
    for (int col = 0; col < n; col++) {
        int piv = col;
        for (int r = col + 1; r < n; r++) if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
        if (fabs(A[piv][col]) < 1e-300) return false;
        if (piv != col) for (int k = 0; k <= n; k++) { const double t = A[piv][k]; A[piv][k] = A[col][k]; A[col][k] = t; }
        for (int r = col + 1; r < n; r++) {
            const double f = A[r][col] / A[col][col];
            for (int k = col; k <= n; k++) A[r][k] -= f * A[col][k];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double v = A[r][n];
        for (int k = r + 1; k < n; k++) v -= A[r][k] * x[k];
        x[r] = v / A[r][r];
    }
    return true;
}

#ifdef VSTAB_HOMOGRAPHY_TRACE   // developer build: per-phase time of one workgroup (tools/homography_phases.py)
static long long* g_h_dbg = nullptr;
extern "C" void vstab_homography_dbg(long long* p) { g_h_dbg = p; }
#define H_MARK(i) do { if (prof_) { const long long now_ = wall_clock64(); a.dbg[i] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define H_MARK(i)
#endif

struct HArgs {
#ifdef VSTAB_HOMOGRAPHY_TRACE
    long long* dbg;
#endif
    const float* grid_flow;
    const int* vmap;
    vstab_fit_record* out;
    int pairs, gh, gw, step;
    int cap;   // row stride (entries) of grid_flow / vmap per pair
};

// residuals + (optionally) the normal equations of HomographyRefineCallback for parameters h[8] over the inliers
// (inlier test against Hbest in f32, threshold thr).  Returns S = |r|^2; JtJ (36 upper entries) / Jtr (8) if wanted.
__device__ double lm_accumulate(const HArgs& a, const float* F, const int* vmap, int nv, const float* Hbf, float thr, const double* h,
                                bool want_jac, double* JtJ /*64*/, double* Jtr /*8*/, double* rinf, double* s_red, double* s_part /*[4][45]*/,
                                double* s_sum /*[45]*/)
{
    double acc[45];
    for (int i = 0; i < 45; i++) acc[i] = 0;
    double S = 0, rmax = 0;
    for (int k = threadIdx.x; k < nv; k += HT) {
        float px, py, cx, cy;
        load_point(F, a.gw, a.step, vmap[k], px, py, cx, cy);
        const float ww = 1.f / (Hbf[6] * px + Hbf[7] * py + 1.f);
        const float ex = (Hbf[0] * px + Hbf[1] * py + Hbf[2]) * ww - cx;
        const float ey = (Hbf[3] * px + Hbf[4] * py + Hbf[5]) * ww - cy;
        if (!(ex * ex + ey * ey <= thr)) continue;
        const double Mx = px, My = py;
        double w = h[6] * Mx + h[7] * My + 1.;
        w = fabs(w) > 2.220446049250313e-16 ? 1. / w : 0;
        const double xi = (h[0] * Mx + h[1] * My + h[2]) * w;
        const double yi = (h[3] * Mx + h[4] * My + h[5]) * w;
        const double r0 = xi - (double)cx, r1 = yi - (double)cy;
        S += r0 * r0 + r1 * r1;
        rmax = fmax(rmax, fmax(fabs(r0), fabs(r1)));
        if (want_jac) {
            const double J0[8] = {Mx * w, My * w, w, 0, 0, 0, -Mx * w * xi, -My * w * xi};
            const double J1[8] = {0, 0, 0, Mx * w, My * w, w, -Mx * w * yi, -My * w * yi};
            int idx = 0;
            for (int p = 0; p < 8; p++)
                for (int q = p; q < 8; q++) acc[idx++] += J0[p] * J0[q] + J1[p] * J1[q];
            for (int p = 0; p < 8; p++) acc[36 + p] += J0[p] * r0 + J1[p] * r1;
        }
    }
    S = block_sum(S, s_red);
    // max via sum trick is not possible: reduce rmax with shuffles
    for (int s = 32; s > 0; s >>= 1) rmax = fmax(rmax, __shfl_down(rmax, s));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = rmax;
    __syncthreads();
    *rinf = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
    __syncthreads();
    if (want_jac) {
        block_sum_many<45>(acc, s_part, s_sum);   // (entry 44 is unused padding of the accumulator array)
        if (threadIdx.x < 44) {
            const int i = threadIdx.x;
            if (i < 36) {
                int p = 0, base = 0;
                while (i >= base + (8 - p)) { base += 8 - p; p++; }
                const int q = p + (i - base);
                JtJ[p * 8 + q] = s_sum[i]; JtJ[q * 8 + p] = s_sum[i];
            } else {
                Jtr[i - 36] = s_sum[i];
            }
        }
        __syncthreads();
    }
    return S;
}

__global__ __launch_bounds__(HT) void homography_kernel(HArgs a)
{
    __shared__ double s_model[HBATCH][9];
    __shared__ float s_modelf[HBATCH][8];
    __shared__ int s_ok[HBATCH];
    __shared__ int s_idx[HBATCH][4];
    __shared__ int s_cnt[HBATCH];
    __shared__ double s_red[8];
    __shared__ double s_part[4 * 45], s_sum[45];
    __shared__ int s_redi[8];
    __shared__ double s_best[9];
    __shared__ double s_A[64], s_v[8], s_x[8], s_d[8];
    __shared__ double s_lm[8];   // 0 lambda, 1 lc, 2 S, 3 flag
    __shared__ int s_ctl[5];     // 0 done, 1 niters, 2 iter, 3 maxGood, 4 subset_failed
    __shared__ Rng s_rng;
    __shared__ JacWs s_ws[HBATCH];
    double (*elim)[9] = reinterpret_cast<double (*)[9]>(s_ws[1].A);   // 8x9 elimination tableau of the LM steps (lane 0 only)

    const int pair = blockIdx.x, tid = threadIdx.x;
#ifdef VSTAB_HOMOGRAPHY_TRACE
    long long tprev_ = wall_clock64();
    const bool prof_ = a.dbg && blockIdx.x == 7 && threadIdx.x == 0;
#endif
    const bool points = a.gw == 0;
    const float* __restrict__ F = a.grid_flow + (size_t)pair * a.cap * (points ? 4 : 2);
    const int* __restrict__ vmap = a.vmap + (size_t)pair * a.cap;
    vstab_fit_record* out = a.out + (size_t)pair * 3 + VSTAB_MODE_PERSPECTIVE;
    const int nv = a.out[(size_t)pair * 3].valid_points;
    // flow.py:153-154 / classic.py:84-86,102-103 (record stays "not computed")
    if (points ? (a.out[(size_t)pair * 3].total_points < 12 || nv < 8) : nv < 12) return;
    const float thr = (float)(2.5 * 2.5);

    if (tid == 0) { s_rng.state = ~0ULL; s_ctl[0] = 0; s_ctl[1] = 2000; s_ctl[2] = 0; s_ctl[3] = 0; s_ctl[4] = 0; }
    __syncthreads();
    while (true) {
        if (tid == 0) {
            Rng r = s_rng;
            int failed = 0;
            for (int c = 0; c < HBATCH && !failed; c++) {
                bool found = false;
                for (int attempt = 0; attempt < 10000 && !found; attempt++) {
                    int idx[4];
                    float ms1[8], ms2[8];
                    for (int i = 0; i < 4; i++) {
                        int v;
                        bool dup;
                        do {
                            v = rng_uniform(r, 0, nv);
                            dup = false;
                            for (int q = 0; q < i; q++) dup |= (idx[q] == v);
                        } while (dup);
                        idx[i] = v;
                        load_point(F, a.gw, a.step, vmap[v], ms1[i * 2], ms1[i * 2 + 1], ms2[i * 2], ms2[i * 2 + 1]);
                    }
                    if (check_subset(ms1, ms2)) {
                        for (int i = 0; i < 4; i++) s_idx[c][i] = idx[i];
                        found = true;
                    }
                }
                if (!found) { failed = c + 1; for (int i = 0; i < 4; i++) s_idx[c][i] = -1; }
            }
            if (failed) for (int c = failed; c < HBATCH; c++) for (int i = 0; i < 4; i++) s_idx[c][i] = -1;
            s_ctl[4] = failed;
            s_rng = r;
        }
        __syncthreads();
        H_MARK(0);
        {   // the batch's 16 minimal solves: one per group of 16 lanes (HT = 16 x 16)
            const int grp = tid >> 4, g = tid & 15;
            JacWs& ws = s_ws[grp];
            if (g == 0) {
                s_ok[grp] = 0;
                s_cnt[grp] = 0;
                ws.go = 0;
                if (s_idx[grp][0] >= 0) {
                    float ms1[8], ms2[8];
                    for (int i = 0; i < 4; i++) load_point(F, a.gw, a.step, vmap[s_idx[grp][i]], ms1[i * 2], ms1[i * 2 + 1], ms2[i * 2], ms2[i * 2 + 1]);
                    ws.go = homography_4pt_setup(ws, ms1, ms2) ? 1 : 0;
                }
            }
            group_sync();
            if (ws.go) {
                double H[9];
                homography_from_ltl_g16(ws, g, H);
                if (g == 0) {
                    for (int k = 0; k < 9; k++) s_model[grp][k] = H[k];
                    for (int k = 0; k < 8; k++) s_modelf[grp][k] = (float)H[k];
                    s_ok[grp] = 1;
                }
            }
        }
        __syncthreads();
        H_MARK(1);
        int cnt[HBATCH];
#pragma unroll
        for (int c = 0; c < HBATCH; c++) cnt[c] = 0;
        for (int k = tid; k < nv; k += HT) {
            float px, py, cx, cy;
            load_point(F, a.gw, a.step, vmap[k], px, py, cx, cy);
#pragma unroll
            for (int c = 0; c < HBATCH; c++) {
                const float ww = 1.f / (s_modelf[c][6] * px + s_modelf[c][7] * py + 1.f);
                const float dx = (s_modelf[c][0] * px + s_modelf[c][1] * py + s_modelf[c][2]) * ww - cx;
                const float dy = (s_modelf[c][3] * px + s_modelf[c][4] * py + s_modelf[c][5]) * ww - cy;
                cnt[c] += (dx * dx + dy * dy <= thr) ? 1 : 0;
            }
        }
#pragma unroll
        for (int c = 0; c < HBATCH; c++) {
            int v = cnt[c];
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s);
            if ((tid & 63) == 0 && v) atomicAdd(&s_cnt[c], v);
        }
        __syncthreads();
        H_MARK(2);
        if (tid == 0) {
            int niters = s_ctl[1], iter = s_ctl[2], maxGood = s_ctl[3];
            bool stop = false;
            for (int c = 0; c < HBATCH && iter < niters; c++, iter++) {
                if (s_idx[c][0] < 0) { stop = true; break; }   // getSubset failed: iter == 0 -> no model, else break
                if (!s_ok[c]) continue;                         // runKernel returned 0 models
                const int good = s_cnt[c];
                if (good > (maxGood > 3 ? maxGood : 3)) {
                    for (int k = 0; k < 9; k++) s_best[k] = s_model[c][k];
                    maxGood = good;
                    niters = update_num_iters(0.992, (double)(nv - good) / nv, 4, niters);
                }
            }
            s_ctl[1] = niters; s_ctl[2] = iter; s_ctl[3] = maxGood;
            s_ctl[0] = (stop || iter >= niters) ? 1 : 0;
        }
        __syncthreads();
        H_MARK(3);
        if (s_ctl[0]) break;
    }
    const int maxGood = s_ctl[3];
    if (maxGood <= 0) {
        if (tid == 0) out->computed = 1;
        return;
    }
    // ---- refit: normalised DLT over all inliers of the best minimal model (runKernel on the compressed set)
    float Hbf[8];
    for (int k = 0; k < 8; k++) Hbf[k] = (float)s_best[k];
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int ninl = 0;
    for (int k = tid; k < nv; k += HT) {
        float px, py, cx, cy;
        load_point(F, a.gw, a.step, vmap[k], px, py, cx, cy);
        const float ww = 1.f / (Hbf[6] * px + Hbf[7] * py + 1.f);
        const float ex = (Hbf[0] * px + Hbf[1] * py + Hbf[2]) * ww - cx;
        const float ey = (Hbf[3] * px + Hbf[4] * py + Hbf[5]) * ww - cy;
        if (ex * ex + ey * ey <= thr) { c0 += cx; c1 += cy; c2 += px; c3 += py; ninl++; }
    }
    double cmx = block_sum(c0, s_red), cmy = block_sum(c1, s_red), cMx = block_sum(c2, s_red), cMy = block_sum(c3, s_red);
    ninl = block_sum(ninl, s_redi);
    cmx /= ninl; cmy /= ninl; cMx /= ninl; cMy /= ninl;
    c0 = c1 = c2 = c3 = 0;
    for (int k = tid; k < nv; k += HT) {
        float px, py, cx, cy;
        load_point(F, a.gw, a.step, vmap[k], px, py, cx, cy);
        const float ww = 1.f / (Hbf[6] * px + Hbf[7] * py + 1.f);
        const float ex = (Hbf[0] * px + Hbf[1] * py + Hbf[2]) * ww - cx;
        const float ey = (Hbf[3] * px + Hbf[4] * py + Hbf[5]) * ww - cy;
        if (ex * ex + ey * ey <= thr) { c0 += fabs(cx - cmx); c1 += fabs(cy - cmy); c2 += fabs(px - cMx); c3 += fabs(py - cMy); }
    }
    double smx = block_sum(c0, s_red), smy = block_sum(c1, s_red), sMx = block_sum(c2, s_red), sMy = block_sum(c3, s_red);
    const double deps = 2.220446049250313e-16;
    const bool degenerate = fabs(smx) < deps || fabs(smy) < deps || fabs(sMx) < deps || fabs(sMy) < deps;
    if (!degenerate) {
        smx = ninl / smx; smy = ninl / smy; sMx = ninl / sMx; sMy = ninl / sMy;
        double acc[45];
        for (int i = 0; i < 45; i++) acc[i] = 0;
        for (int k = tid; k < nv; k += HT) {
            float px, py, cx, cy;
            load_point(F, a.gw, a.step, vmap[k], px, py, cx, cy);
            const float ww = 1.f / (Hbf[6] * px + Hbf[7] * py + 1.f);
            const float ex = (Hbf[0] * px + Hbf[1] * py + Hbf[2]) * ww - cx;
            const float ey = (Hbf[3] * px + Hbf[4] * py + Hbf[5]) * ww - cy;
            if (!(ex * ex + ey * ey <= thr)) continue;
            const double x = (cx - cmx) * smx, y = (cy - cmy) * smy;
            const double X = (px - cMx) * sMx, Y = (py - cMy) * sMy;
            const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
            const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
            int idx = 0;
            for (int j = 0; j < 9; j++)
                for (int q = j; q < 9; q++) acc[idx++] += Lx[j] * Lx[q] + Ly[j] * Ly[q];
        }
        H_MARK(4);
        double* s_LtL = s_ws[0].A;   // the refit reuses slot 0 (the batch solves are finished)
        block_sum_many<45>(acc, s_part, s_sum);
        if (tid < 45) {
            int j = 0, base = 0;
            while (tid >= base + (9 - j)) { base += 9 - j; j++; }
            s_LtL[j * 9 + j + (tid - base)] = s_sum[tid];
        }
        __syncthreads();
        H_MARK(5);
        if (tid < 16) {   // the first group of 16 lanes (one wavefront's DPP row 0)
            if (tid == 0) {
                double* nm = s_ws[0].norm;
                nm[0] = cmx; nm[1] = cmy; nm[2] = cMx; nm[3] = cMy; nm[4] = smx; nm[5] = smy; nm[6] = sMx; nm[7] = sMy;
            }
            group_sync();
            double H[9];
            homography_from_ltl_g16(s_ws[0], tid, H);
            if (tid == 0) for (int i = 0; i < 9; i++) s_best[i] = H[i];
        }
        __syncthreads();
    }
    H_MARK(6);
    // ---- Levenberg-Marquardt, 10 iterations (LMSolverImpl::run, eps = FLT_EPSILON), parameters h[0..7]
    double x[8];
    for (int i = 0; i < 8; i++) x[i] = s_best[i];
    double rinf;
    double S = lm_accumulate(a, F, vmap, nv, Hbf, thr, x, true, s_A, s_v, &rinf, s_red, s_part, s_sum);
    __shared__ double s_D[8];
    if (tid == 0) { for (int i = 0; i < 8; i++) s_D[i] = s_A[i * 8 + i]; s_lm[0] = 1.0; s_lm[1] = 0.75; }
    __syncthreads();
    const double Rlo = 0.25, Rhi = 0.75;
    H_MARK(7);
    for (int iter = 0; iter < 10; iter++) {
        if (tid == 0) {
            double Ap[64];
            for (int i = 0; i < 64; i++) Ap[i] = s_A[i];
            for (int i = 0; i < 8; i++) Ap[i * 8 + i] += s_lm[0] * s_D[i];
            double d[8];
            if (!solve_sym(elim, Ap, s_v, 8, d)) for (int i = 0; i < 8; i++) d[i] = 0;
            for (int i = 0; i < 8; i++) { s_d[i] = d[i]; s_x[i] = x[i] - d[i]; }
        }
        __syncthreads();
        H_MARK(8);
        double xd[8];
        for (int i = 0; i < 8; i++) xd[i] = s_x[i];
        double rinf_d;
        const double Sd = lm_accumulate(a, F, vmap, nv, Hbf, thr, xd, false, nullptr, nullptr, &rinf_d, s_red, s_part, s_sum);
        H_MARK(9);
        if (tid == 0) {
            double dS = 0, tdv = 0;
            for (int i = 0; i < 8; i++) {
                double s = 0;
                for (int j = 0; j < 8; j++) s += s_A[i * 8 + j] * s_d[j];
                dS += s_d[i] * (-s + 2 * s_v[i]);
                tdv += s_d[i] * s_v[i];
            }
            const double R = (S - Sd) / (fabs(dS) > 2.220446049250313e-16 ? dS : 1);
            double lambda = s_lm[0], lc = s_lm[1];
            if (R > Rhi) {
                lambda *= 0.5;
                if (lambda < lc) lambda = 0;
            } else if (R < Rlo) {
                double nu = (Sd - S) / (fabs(tdv) > 2.220446049250313e-16 ? tdv : 1) + 2;
                nu = nu < 2. ? 2. : (nu > 10. ? 10. : nu);
                if (lambda == 0) {
                    double maxval = 2.220446049250313e-16;
                    for (int c = 0; c < 8; c++) {   // diagonal of A^-1, column by column
                        double e[8], col[8];
                        for (int i = 0; i < 8; i++) e[i] = (i == c);
                        if (solve_sym(elim, s_A, e, 8, col)) maxval = fmax(maxval, fabs(col[c]));
                    }
                    lambda = lc = 1. / maxval;
                    nu *= 0.5;
                }
                lambda *= nu;
            }
            s_lm[0] = lambda; s_lm[1] = lc;
            s_lm[3] = (Sd < S) ? 1.0 : 0.0;
        }
        __syncthreads();
        H_MARK(10);
        const bool accept = s_lm[3] != 0.0;
        double dinf = 0;
        for (int i = 0; i < 8; i++) dinf = fmax(dinf, fabs(s_d[i]));
        if (accept) {
            S = Sd;
            for (int i = 0; i < 8; i++) x[i] = xd[i];
            __syncthreads();
            S = lm_accumulate(a, F, vmap, nv, Hbf, thr, x, true, s_A, s_v, &rinf, s_red, s_part, s_sum);
        }
        __syncthreads();
        H_MARK(11);
#ifdef VSTAB_HOMOGRAPHY_TRACE
        if (prof_) a.dbg[14] += 1;
#endif
        if (!(iter + 1 < 10 && dinf >= 1.1920928955078125e-07 && rinf >= 1.1920928955078125e-07)) break;
    }
    // ---- record (flow.py:171-175: confidence = inliers/valid, residual uses the affine part of H only)
    double res = 0;
    for (int k = tid; k < nv; k += HT) {
        float px, py, cx, cy;
        load_point(F, a.gw, a.step, vmap[k], px, py, cx, cy);
        const double X = px, Y = py;
        res += fabs(X * x[0] + Y * x[1] + x[2] - (double)cx) + fabs(X * x[3] + Y * x[4] + x[5] - (double)cy);
    }
    res = block_sum(res, s_red);
    H_MARK(12);
    if (tid == 0) {
        out->computed = 1;
        out->confidence = (double)maxGood / (double)nv;
        if (out->confidence >= 0.15) {
            for (int i = 0; i < 8; i++) out->matrix[i] = (float)x[i];
            out->matrix[8] = 1.f;
            out->residual = res / (2.0 * nv);
            out->accepted = 1;
        }
    }
}

}  // namespace

int vstab_fit_homography(vstab_ctx* ctx, const float* grid_flow, const int* vmap, int pairs, int gh, int gw, int step,
                         int cap, vstab_fit_record* d_out)
{
    HArgs a{
#ifdef VSTAB_HOMOGRAPHY_TRACE
        g_h_dbg,
#endif
        grid_flow, vmap, d_out, pairs, gh, gw, step, cap};
    hipLaunchKernelGGL(homography_kernel, dim3((unsigned)pairs), dim3(HT), 0, ctx->stream, a);
    VSTAB_HIP(hipGetLastError());
    return 0;
}
