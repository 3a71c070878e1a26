/*
 * vo_fit.c -- CPU oracle for F4+F5 (TEST INFRASTRUCTURE, see vo_common.h): grid sampling of the
 * dense flow and the RANSAC model-fit cascade.
 *
 * Reference call site: nodes/video_stabilizer_flow.py:141-210
 *   perspective : cv2.findHomography(prev, curr, RANSAC, 2.5, maxIters=2000, confidence=0.992), accept >= 0.15
 *   similarity  : cv2.estimateAffinePartial2D(prev, curr, RANSAC, 2.0, 2000, 0.992),            accept >= 0.1
 *   translation : per-axis np.median of the shifts
 * Algorithm source (published OpenCV 4.x, restated from memory -- parity UNPINNED):
 *   calib3d/src/ptsetreg.cpp  RANSACPointSetRegistrator (RNG seeded with (uint64)-1, getSubset,
 *                             findInliers, RANSACUpdateNumIters), AffinePartial2DEstimatorCallback,
 *                             AffinePartial2DRefineCallback, estimateAffinePartial2D (10 LM iterations)
 *   calib3d/src/fundam.cpp    HomographyEstimatorCallback (normalised DLT + 9x9 symmetric eigen,
 *                             checkSubset), HomographyRefineCallback, findHomography
 *   calib3d/src/levmarq.cpp   LMSolverImpl::run (eps = FLT_EPSILON)
 *   core/src/lapack.cpp       JacobiImpl_ (cv::eigen / solve(DECOMP_EIG) without LAPACK)
 *   core/src/rand.cpp         RNG: multiply-with-carry, coefficient 4164903690
 */
#include "vo_common.h"
#include "vstab_oracle.h"

/* ------------------------------------------------------------------ RNG */
typedef struct { uint64_t state; } vo_rng;
static inline unsigned rng_next(vo_rng* r)
{
    r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32);
    return (unsigned)r->state;
}
static inline int rng_uniform(vo_rng* r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (unsigned)(b - a) + a); }

/* ------------------------------------------------------------------ Jacobi eigen (symmetric, f64) */
static void jacobi_eigen(double* A, int n, double* W, double* V)
{
    const double eps = DBL_EPSILON;
    int indR[16], indC[16];
    int i, j, k, m;
    double mv;
    for (i = 0; i < n; i++) {
        for (j = 0; j < n; j++) V[i * n + j] = 0;
        V[i * n + i] = 1;
    }
    for (k = 0; k < n; k++) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) {
            for (m = k + 1, mv = fabs(A[n * k + m]), i = k + 2; i < n; i++) {
                double val = fabs(A[n * k + i]);
                if (mv < val) mv = val, m = i;
            }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabs(A[k]), i = 1; i < k; i++) {
                double val = fabs(A[n * i + k]);
                if (mv < val) mv = val, m = i;
            }
            indC[k] = m;
        }
    }
    const int maxIters = n * n * 30;
    if (n > 1)
        for (int iters = 0; iters < maxIters; iters++) {
            for (k = 0, mv = fabs(A[indR[0]]), i = 1; i < n - 1; i++) {
                double val = fabs(A[n * i + indR[i]]);
                if (mv < val) mv = val, k = i;
            }
            int l = indR[k];
            for (i = 1; i < n; i++) {
                double val = fabs(A[n * indC[i] + i]);
                if (mv < val) mv = val, k = indC[i], l = i;
            }
            double p = A[n * k + l];
            if (fabs(p) <= eps) break;
            double y = (W[l] - W[k]) * 0.5;
            double t = fabs(y) + hypot(p, y);
            double s = hypot(p, t);
            double c = t / s;
            s = p / s;
            t = (p / t) * p;
            if (y < 0) s = -s, t = -t;
            A[n * k + l] = 0;
            W[k] -= t;
            W[l] += t;
            double a0, b0;
#define ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
            for (i = 0; i < k; i++) ROT(A[n * i + k], A[n * i + l]);
            for (i = k + 1; i < l; i++) ROT(A[n * k + i], A[n * i + l]);
            for (i = l + 1; i < n; i++) ROT(A[n * k + i], A[n * l + i]);
            for (i = 0; i < n; i++) ROT(V[n * k + i], V[n * l + i]);
#undef ROT
            for (j = 0; j < 2; j++) {
                int idx = j == 0 ? k : l;
                if (idx < n - 1) {
                    for (m = idx + 1, mv = fabs(A[n * idx + m]), i = idx + 2; i < n; i++) {
                        double val = fabs(A[n * idx + i]);
                        if (mv < val) mv = val, m = i;
                    }
                    indR[idx] = m;
                }
                if (idx > 0) {
                    for (m = 0, mv = fabs(A[idx]), i = 1; i < idx; i++) {
                        double val = fabs(A[n * i + idx]);
                        if (mv < val) mv = val, m = i;
                    }
                    indC[idx] = m;
                }
            }
        }
    for (k = 0; k < n - 1; k++) {
        m = k;
        for (i = k + 1; i < n; i++)
            if (W[m] < W[i]) m = i;
        if (k != m) {
            double tw = W[m]; W[m] = W[k]; W[k] = tw;
            for (i = 0; i < n; i++) { double tv = V[n * m + i]; V[n * m + i] = V[n * k + i]; V[n * k + i] = tv; }
        }
    }
}

/* solve(A, b, x, DECOMP_EIG) for symmetric A (n <= 8): x = sum_k v_k (v_k.b) / w_k, tiny w_k dropped */
static void solve_eig(const double* A_in, const double* b, int n, double* x)
{
    double A[64], W[8], V[64];
    memcpy(A, A_in, sizeof(double) * n * n);
    jacobi_eigen(A, n, W, V);
    double threshold = 0;
    for (int i = 0; i < n; i++) threshold += fabs(W[i]);
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < n; i++) x[i] = 0;
    for (int k = 0; k < n; k++) {
        if (fabs(W[k]) <= threshold) continue;
        double s = 0;
        for (int i = 0; i < n; i++) s += V[k * n + i] * b[i];
        s /= W[k];
        for (int i = 0; i < n; i++) x[i] += s * V[k * n + i];
    }
}

static void invert_eig(const double* A_in, int n, double* Ainv)
{
    double e[8], col[8];
    for (int c = 0; c < n; c++) {
        for (int i = 0; i < n; i++) e[i] = (i == c);
        solve_eig(A_in, e, n, col);
        for (int i = 0; i < n; i++) Ainv[i * n + c] = col[i];
    }
}

/* ------------------------------------------------------------------ Levenberg-Marquardt (LMSolverImpl::run) */
typedef void (*lm_compute_fn)(const double* x, int np, const float* src, const float* dst, int count,
                              double* err /*2*count*/, double* J /*2*count*np or NULL*/);

static int lm_run(lm_compute_fn fn, double* x /*np*/, int np, const float* src, const float* dst, int count,
                  int maxIters)
{
    const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
    const int m = 2 * count;
    double* r = (double*)malloc(sizeof(double) * m);
    double* rd = (double*)malloc(sizeof(double) * m);
    double* J = (double*)malloc(sizeof(double) * (size_t)m * np);
    double A[64], Ap[64], v[8], d[8], xd[8], D[8], temp_d[8];
#define NORM2(p_, n_, out_) do { double s_ = 0; for (int q_ = 0; q_ < (n_); q_++) s_ += (p_)[q_] * (p_)[q_]; out_ = s_; } while (0)
#define NORMAL_EQ()                                                               \
    do {                                                                          \
        for (int a_ = 0; a_ < np; a_++) {                                         \
            for (int b_ = 0; b_ < np; b_++) {                                     \
                double s_ = 0;                                                    \
                for (int q_ = 0; q_ < m; q_++) s_ += J[(size_t)q_ * np + a_] * J[(size_t)q_ * np + b_]; \
                A[a_ * np + b_] = s_;                                             \
            }                                                                     \
            double t_ = 0;                                                        \
            for (int q_ = 0; q_ < m; q_++) t_ += J[(size_t)q_ * np + a_] * r[q_]; \
            v[a_] = t_;                                                           \
        }                                                                         \
    } while (0)
    fn(x, np, src, dst, count, r, J);
    double S;
    NORM2(r, m, S);
    NORMAL_EQ();
    for (int i = 0; i < np; i++) D[i] = A[i * np + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        memcpy(Ap, A, sizeof(double) * np * np);
        for (int i = 0; i < np; i++) Ap[i * np + i] += lambda * D[i];
        solve_eig(Ap, v, np, d);
        for (int i = 0; i < np; i++) xd[i] = x[i] - d[i];
        fn(xd, np, src, dst, count, rd, 0);
        double Sd;
        NORM2(rd, m, Sd);
        for (int i = 0; i < np; i++) {
            double s = 0;
            for (int j = 0; j < np; j++) s += A[i * np + j] * d[j];
            temp_d[i] = -s + 2 * v[i];
        }
        double dS = 0;
        for (int i = 0; i < np; i++) dS += d[i] * temp_d[i];
        double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            double t = 0;
            for (int i = 0; i < np; i++) t += d[i] * v[i];
            double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = nu < 2. ? 2. : (nu > 10. ? 10. : nu);
            if (lambda == 0) {
                invert_eig(A, np, Ap);
                double maxval = DBL_EPSILON;
                for (int i = 0; i < np; i++) maxval = maxval > fabs(Ap[i * np + i]) ? maxval : fabs(Ap[i * np + i]);
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            memcpy(x, xd, sizeof(double) * np);
            fn(x, np, src, dst, count, r, J);
            NORMAL_EQ();
        }
        iter++;
        double dn = 0, rn = 0;
        for (int i = 0; i < np; i++) dn = dn > fabs(d[i]) ? dn : fabs(d[i]);
        for (int q = 0; q < m; q++) rn = rn > fabs(r[q]) ? rn : fabs(r[q]);
        if (!(iter < maxIters && dn >= epsx && rn >= epsf)) break;
    }
#undef NORM2
#undef NORMAL_EQ
    free(r); free(rd); free(J);
    return iter;
}

/* ------------------------------------------------------------------ RANSAC core */
static int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : vo_round_d(num / denom);
}

static int have_collinear(const float* pts, int count)
{
    int i = count - 1;
    for (int j = 0; j < i; j++) {
        double dx1 = pts[j * 2] - pts[i * 2], dy1 = pts[j * 2 + 1] - pts[i * 2 + 1];
        for (int k = 0; k < j; k++) {
            double dx2 = pts[k * 2] - pts[i * 2], dy2 = pts[k * 2 + 1] - pts[i * 2 + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return 1;
        }
    }
    return 0;
}

typedef struct {
    int model_points;
    int (*kernel)(const float* m1, const float* m2, int count, double* model);
    void (*error)(const float* m1, const float* m2, int count, const double* model, float* err);
    int (*check)(const float* ms1, const float* ms2, int count);
    int model_size;
} ransac_cb;

static int get_subset(const ransac_cb* cb, const float* m1, const float* m2, int count, float* ms1, float* ms2,
                      vo_rng* rng, int maxAttempts)
{
    int idx[8];
    const int mp = cb->model_points;
    for (int iters = 0; iters < maxAttempts; ++iters) {
        int i;
        for (i = 0; i < mp; ++i) {
            int idx_i;
            for (;;) {
                idx_i = rng_uniform(rng, 0, count);
                int dup = 0;
                for (int q = 0; q < i; q++) dup |= (idx[q] == idx_i);
                if (!dup) break;
            }
            idx[i] = idx_i;
            ms1[i * 2] = m1[idx_i * 2]; ms1[i * 2 + 1] = m1[idx_i * 2 + 1];
            ms2[i * 2] = m2[idx_i * 2]; ms2[i * 2 + 1] = m2[idx_i * 2 + 1];
        }
        if (cb->check(ms1, ms2, i)) return 1;
    }
    return 0;
}

static int ransac_run(const ransac_cb* cb, const float* m1, const float* m2, int count, double thresh,
                      double confidence, int maxIters, double* bestModel, uint8_t* bestMask)
{
    int niters = maxIters > 1 ? maxIters : 1;
    int maxGoodCount = 0;
    vo_rng rng = {(uint64_t)-1};
    if (count < cb->model_points) return 0;
    float* err = (float*)malloc(sizeof(float) * count);
    uint8_t* mask = (uint8_t*)malloc(count);
    float ms1[16], ms2[16];
    double model[16];
    const float t = (float)(thresh * thresh);
    for (int iter = 0; iter < niters; iter++) {
        if (count > cb->model_points) {
            if (!get_subset(cb, m1, m2, count, ms1, ms2, &rng, 10000)) {
                if (iter == 0) { free(err); free(mask); return 0; }
                break;
            }
        } else {
            memcpy(ms1, m1, sizeof(float) * 2 * count);
            memcpy(ms2, m2, sizeof(float) * 2 * count);
        }
        int nmodels = cb->kernel(ms1, ms2, cb->model_points, model);
        if (nmodels <= 0) continue;
        cb->error(m1, m2, count, model, err);
        int good = 0;
        for (int i = 0; i < count; i++) {
            int f = err[i] <= t;
            mask[i] = (uint8_t)f;
            good += f;
        }
        const int floor_cnt = maxGoodCount > cb->model_points - 1 ? maxGoodCount : cb->model_points - 1;
        if (good > floor_cnt) {
            memcpy(bestMask, mask, count);
            memcpy(bestModel, model, sizeof(double) * cb->model_size);
            maxGoodCount = good;
            niters = ransac_update_num_iters(confidence, (double)(count - good) / count, cb->model_points, niters);
        }
    }
    free(err);
    free(mask);
    return maxGoodCount > 0;
}

static int compress(float* pts, const uint8_t* mask, int count)
{
    int j = 0;
    for (int i = 0; i < count; i++)
        if (mask[i]) {
            if (i > j) { pts[j * 2] = pts[i * 2]; pts[j * 2 + 1] = pts[i * 2 + 1]; }
            j++;
        }
    return j;
}

/* ------------------------------------------------------------------ similarity (AffinePartial2D) */
static int ap2d_kernel(const float* from, const float* to, int count, double* M)
{
    (void)count;
    double x1 = from[0], y1 = from[1], x2 = from[2], y2 = from[3];
    double X1 = to[0], Y1 = to[1], X2 = to[2], Y2 = to[3];
    double d = 1. / ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
    double S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2));
    double S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2));
    double S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) - (X1 * x2 - X2 * x1) * (x1 - x2));
    double S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) - (Y1 * y2 - Y2 * y1) * (y1 - y2));
    M[0] = M[4] = S0;
    M[1] = -S1;
    M[2] = S2;
    M[3] = S1;
    M[5] = S3;
    return 1;
}
static void affine_error(const float* from, const float* to, int count, const double* F, float* err)
{
    const float F0 = (float)F[0], F1 = (float)F[1], F2 = (float)F[2], F3 = (float)F[3], F4 = (float)F[4], F5 = (float)F[5];
    for (int i = 0; i < count; i++) {
        float fx = from[i * 2], fy = from[i * 2 + 1];
        float a = F0 * fx + F1 * fy + F2 - to[i * 2];
        float b = F3 * fx + F4 * fy + F5 - to[i * 2 + 1];
        err[i] = a * a + b * b;
    }
}
static int affine_check(const float* ms1, const float* ms2, int count)
{
    return !have_collinear(ms1, count) && !have_collinear(ms2, count);
}
static void ap2d_refine(const double* h, int np, const float* M, const float* m, int count, double* err, double* J)
{
    (void)np;
    for (int i = 0; i < count; i++) {
        double Mx = M[i * 2], My = M[i * 2 + 1];
        double xi = h[0] * Mx - h[1] * My + h[2];
        double yi = h[1] * Mx + h[0] * My + h[3];
        err[i * 2] = xi - m[i * 2];
        err[i * 2 + 1] = yi - m[i * 2 + 1];
        if (J) {
            double* Jp = J + (size_t)i * 8;
            Jp[0] = Mx; Jp[1] = -My; Jp[2] = 1.; Jp[3] = 0.;
            Jp[4] = My; Jp[5] = Mx; Jp[6] = 0.; Jp[7] = 1.;
        }
    }
}

int vo_estimate_affine_partial2d(const float* from_in, const float* to_in, int count, double thresh, int max_iters,
                                 double confidence, int refine_iters, double* H /*2x3*/, uint8_t* inliers)
{
    static const ransac_cb cb = {2, ap2d_kernel, affine_error, affine_check, 6};
    float* from = (float*)malloc(sizeof(float) * 2 * count);
    float* to = (float*)malloc(sizeof(float) * 2 * count);
    memcpy(from, from_in, sizeof(float) * 2 * count);
    memcpy(to, to_in, sizeof(float) * 2 * count);
    memset(inliers, 0, count);
    int result = ransac_run(&cb, from, to, count, thresh, confidence, max_iters, H, inliers);
    if (result && count > 2 && refine_iters) {
        compress(from, inliers, count);
        int ni = compress(to, inliers, count);
        if (ni > 0) {
            double hv[4] = {H[0], H[3], H[2], H[5]};
            lm_run(ap2d_refine, hv, 4, from, to, ni, refine_iters);
            H[0] = H[4] = hv[0];
            H[1] = -hv[1];
            H[2] = hv[2];
            H[3] = hv[1];
            H[5] = hv[3];
        }
    }
    if (!result) memset(inliers, 0, count);
    free(from);
    free(to);
    return result;
}

/* ------------------------------------------------------------------ homography */
static int homography_kernel(const float* M, const float* m, int count, double* H)
{
    double LtL[81], W[9], V[81];
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) {
        cmx += m[i * 2]; cmy += m[i * 2 + 1];
        cMx += M[i * 2]; cMy += M[i * 2 + 1];
    }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; i++) {
        smx += fabs(m[i * 2] - cmx); smy += fabs(m[i * 2 + 1] - cmy);
        sMx += fabs(M[i * 2] - cMx); sMy += fabs(M[i * 2 + 1] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON) return 0;
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    memset(LtL, 0, sizeof(LtL));
    for (int i = 0; i < count; i++) {
        double x = (m[i * 2] - cmx) * smx, y = (m[i * 2 + 1] - cmy) * smy;
        double X = (M[i * 2] - cMx) * sMx, Y = (M[i * 2 + 1] - cMy) * sMy;
        double Lx[] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        double Ly[] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++)
            for (int k = j; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; j++)
        for (int k = 0; k < j; k++) LtL[j * 9 + k] = LtL[k * 9 + j];
    jacobi_eigen(LtL, 9, W, V);
    const double* H0 = V + 8 * 9;
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
    return 1;
}
static void homography_error(const float* M, const float* m, int count, const double* H, float* err)
{
    const float Hf[] = {(float)H[0], (float)H[1], (float)H[2], (float)H[3], (float)H[4], (float)H[5], (float)H[6], (float)H[7]};
    for (int i = 0; i < count; i++) {
        float Mx = M[i * 2], My = M[i * 2 + 1];
        float ww = 1.f / (Hf[6] * Mx + Hf[7] * My + 1.f);
        float dx = (Hf[0] * Mx + Hf[1] * My + Hf[2]) * ww - m[i * 2];
        float dy = (Hf[3] * Mx + Hf[4] * My + Hf[5]) * ww - m[i * 2 + 1];
        err[i] = dx * dx + dy * dy;
    }
}
static double det3pts(const float* p, const int* t)
{
    double a[9] = {p[t[0] * 2], p[t[0] * 2 + 1], 1., p[t[1] * 2], p[t[1] * 2 + 1], 1., p[t[2] * 2], p[t[2] * 2 + 1], 1.};
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}
static int homography_check(const float* ms1, const float* ms2, int count)
{
    if (have_collinear(ms1, count) || have_collinear(ms2, count)) return 0;
    if (count == 4) {
        static const int tt[][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int negative = 0;
        for (int i = 0; i < 4; i++) negative += det3pts(ms1, tt[i]) * det3pts(ms2, tt[i]) < 0;
        if (negative != 0 && negative != 4) return 0;
    }
    return 1;
}
static void homography_refine(const double* h, int np, const float* M, const float* m, int count, double* err, double* J)
{
    (void)np;
    for (int i = 0; i < count; i++) {
        double Mx = M[i * 2], My = M[i * 2 + 1];
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        err[i * 2] = xi - m[i * 2];
        err[i * 2 + 1] = yi - m[i * 2 + 1];
        if (J) {
            double* Jp = J + (size_t)i * 16;
            Jp[0] = Mx * ww; Jp[1] = My * ww; Jp[2] = ww;
            Jp[3] = Jp[4] = Jp[5] = 0.;
            Jp[6] = -Mx * ww * xi; Jp[7] = -My * ww * xi;
            Jp[8] = Jp[9] = Jp[10] = 0.;
            Jp[11] = Mx * ww; Jp[12] = My * ww; Jp[13] = ww;
            Jp[14] = -Mx * ww * yi; Jp[15] = -My * ww * yi;
        }
    }
}

int vo_find_homography_ransac(const float* from_in, const float* to_in, int count, double thresh, int max_iters,
                              double confidence, double* H, uint8_t* inliers)
{
    static const ransac_cb cb = {4, homography_kernel, homography_error, homography_check, 9};
    if (count < 4) return 0;
    float* src = (float*)malloc(sizeof(float) * 2 * count);
    float* dst = (float*)malloc(sizeof(float) * 2 * count);
    memcpy(src, from_in, sizeof(float) * 2 * count);
    memcpy(dst, to_in, sizeof(float) * 2 * count);
    memset(inliers, 0, count);
    int result;
    if (count == 4) {
        result = homography_kernel(src, dst, 4, H) > 0;
        if (result) memset(inliers, 1, count);
    } else {
        result = ransac_run(&cb, src, dst, count, thresh, confidence, max_iters, H, inliers);
    }
    if (result && count > 4) {
        compress(src, inliers, count);
        int ni = compress(dst, inliers, count);
        if (ni > 0) {
            homography_kernel(src, dst, ni, H);
            lm_run(homography_refine, H, 8, src, dst, ni, 10);
        }
    }
    if (!result) memset(inliers, 0, count);
    free(src);
    free(dst);
    return result;
}

/* ------------------------------------------------------------------ flow.py:141-210 */
static int cmp_f32(const void* a, const void* b)
{
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}
static float median_f32(float* v, int n)
{
    qsort(v, n, sizeof(float), cmp_f32);
    if (n & 1) return v[n / 2];
    return (v[n / 2 - 1] + v[n / 2]) / 2.0f; /* np.mean of two float32 values */
}

/* Evaluates every mode at or below `requested`; rec[mode] as vo_fit_result with valid = accepted. */
/* candidate fits of every mode at or below `requested_mode` on nv point pairs (flow.py:156-210, classic.py:105-160) */
static void fit_modes_on_points(const float* prev, const float* curr, int nv, int total, int requested_mode, vo_fit_result* rec);

static void reset_records(vo_fit_result* rec)
{
    for (int m = 0; m < 3; m++) {
        memset(&rec[m], 0, sizeof(rec[m]));
        rec[m].matrix[0] = rec[m].matrix[4] = rec[m].matrix[8] = 1.f;
        rec[m].mode = -1; /* not computed */
    }
}

void vo_fit_all_modes(const float* flow, int h, int w, int step, int requested_mode, vo_fit_result* rec /*3*/,
                      int* valid_points, int* total_points)
{
    const int gh = (h + step - 1) / step, gw = (w + step - 1) / step;
    const int total = gh * gw;
    float* prev = (float*)malloc(sizeof(float) * 2 * total);
    float* curr = (float*)malloc(sizeof(float) * 2 * total);
    int nv = 0;
    for (int gy = 0; gy < gh; gy++)
        for (int gx = 0; gx < gw; gx++) {
            const int x = gx * step, y = gy * step;
            const float px = (float)x, py = (float)y;
            const float cx = px + flow[((size_t)y * w + x) * 2], cy = py + flow[((size_t)y * w + x) * 2 + 1];
            if (isfinite(cx) && isfinite(cy)) {
                prev[nv * 2] = px; prev[nv * 2 + 1] = py;
                curr[nv * 2] = cx; curr[nv * 2 + 1] = cy;
                nv++;
            }
        }
    *valid_points = nv;
    *total_points = total;
    reset_records(rec);
    if (nv >= 12) fit_modes_on_points(prev, curr, nv, total, requested_mode, rec);   /* flow.py:153-154 */
    free(prev); free(curr);
}

/* Classic estimator (classic.py:86-104): `count` detected features, status from the LK tracker.
 * Fewer than 12 features or fewer than 8 tracked ones -> no candidate at all. */
void vo_fit_all_modes_points(const float* from, const float* to, const uint8_t* status, int count, int requested_mode,
                             vo_fit_result* rec /*3*/, int* valid_points)
{
    float* prev = (float*)malloc(sizeof(float) * 2 * (count > 0 ? count : 1));
    float* curr = (float*)malloc(sizeof(float) * 2 * (count > 0 ? count : 1));
    int nv = 0;
    for (int i = 0; i < count; i++)
        if (status[i] == 1) {
            prev[nv * 2] = from[i * 2]; prev[nv * 2 + 1] = from[i * 2 + 1];
            curr[nv * 2] = to[i * 2]; curr[nv * 2 + 1] = to[i * 2 + 1];
            nv++;
        }
    *valid_points = nv;
    reset_records(rec);
    if (count >= 12 && nv >= 8) fit_modes_on_points(prev, curr, nv, count, requested_mode, rec);
    free(prev); free(curr);
}

static void fit_modes_on_points(const float* prev, const float* curr, int nv, int total, int requested_mode, vo_fit_result* rec)
{
    uint8_t* inl = (uint8_t*)malloc(nv);
    if (requested_mode >= VO_MODE_PERSPECTIVE && nv >= 4) {
        double H[9];
        vo_fit_result* r = &rec[VO_MODE_PERSPECTIVE];
        r->mode = VO_MODE_PERSPECTIVE;
        if (vo_find_homography_ransac(prev, curr, nv, 2.5, 2000, 0.992, H, inl)) {
            int cnt = 0;
            for (int i = 0; i < nv; i++) cnt += inl[i] != 0;
            const double conf = (double)cnt / (double)nv;
            r->confidence = conf;
            if (conf >= 0.15) {
                double s = 0; /* residual uses only the affine part of H (flow.py:174) */
                for (int i = 0; i < nv; i++) {
                    double px = prev[i * 2], py = prev[i * 2 + 1];
                    s += fabs(px * H[0] + py * H[1] + H[2] - curr[i * 2]);
                    s += fabs(px * H[3] + py * H[4] + H[5] - curr[i * 2 + 1]);
                }
                r->residual = s / (2.0 * nv);
                for (int i = 0; i < 9; i++) r->matrix[i] = (float)H[i];
                r->valid = 1;
            }
        }
    }
    if (requested_mode >= VO_MODE_SIMILARITY && nv >= 3) {
        double M[6];
        vo_fit_result* r = &rec[VO_MODE_SIMILARITY];
        r->mode = VO_MODE_SIMILARITY;
        if (vo_estimate_affine_partial2d(prev, curr, nv, 2.0, 2000, 0.992, 10, M, inl)) {
            int cnt = 0;
            for (int i = 0; i < nv; i++) cnt += inl[i] != 0;
            const double conf = (double)cnt / (double)nv;
            r->confidence = conf;
            if (conf >= 0.1) {
                double s = 0;
                for (int i = 0; i < nv; i++) {
                    double px = prev[i * 2], py = prev[i * 2 + 1];
                    s += fabs(px * M[0] + py * M[1] + M[2] - curr[i * 2]);
                    s += fabs(px * M[3] + py * M[4] + M[5] - curr[i * 2 + 1]);
                }
                r->residual = s / (2.0 * nv);
                for (int i = 0; i < 6; i++) r->matrix[i] = (float)M[i];
                r->matrix[6] = 0.f; r->matrix[7] = 0.f; r->matrix[8] = 1.f;
                r->valid = 1;
            }
        }
    }
    {
        vo_fit_result* r = &rec[VO_MODE_TRANSLATION];
        r->mode = VO_MODE_TRANSLATION;
        float* sx = (float*)malloc(sizeof(float) * nv);
        float* sy = (float*)malloc(sizeof(float) * nv);
        for (int i = 0; i < nv; i++) { sx[i] = curr[i * 2] - prev[i * 2]; sy[i] = curr[i * 2 + 1] - prev[i * 2 + 1]; }
        const float tx = median_f32(sx, nv), ty = median_f32(sy, nv);
        double s = 0;
        for (int i = 0; i < nv; i++) {
            s += fabsf((prev[i * 2] + tx) - curr[i * 2]);
            s += fabsf((prev[i * 2 + 1] + ty) - curr[i * 2 + 1]);
        }
        r->matrix[2] = tx; r->matrix[5] = ty;
        r->confidence = (double)nv / (double)total;
        r->residual = s / (2.0 * nv);
        r->valid = 1;
        free(sx); free(sy);
    }
    free(inl);
}

void vo_fit_from_flow(const float* flow, int h, int w, int step, int requested_mode, vo_fit_result* out)
{
    vo_fit_result rec[3];
    int nv, total;
    vo_fit_all_modes(flow, h, w, step, requested_mode, rec, &nv, &total);
    for (int m = requested_mode; m >= 0; m--)
        if (rec[m].mode == m && rec[m].valid) { *out = rec[m]; return; }
    memset(out, 0, sizeof(*out));
    out->matrix[0] = out->matrix[4] = out->matrix[8] = 1.f;
    out->mode = VO_MODE_TRANSLATION;
}
