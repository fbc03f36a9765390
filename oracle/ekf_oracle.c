/*
 * ekf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  "parity unpinned"
 * by the reference's own tests (see ekf_oracle.h for how it is pinned instead).
 *
 * CPU restatement (plain C, IEEE fp64, no FMA contraction: build with
 * -ffp-contract=off) of
 *   Tools/GenericExtendedKalmanFilter.m            (whole file)
 *   Tools/SIAlphaModelEKF.m:27-109                 (3-state callbacks)
 *   Tools/SIAlphaModelEKFOptControlled.m:27-168    (6-state callbacks)
 *   Tools/SIAlphaModelBackwardEKF.m:19-130         (flipped 3-state)
 *   Tools/SIAlphaModelBackwardEKFOptControlled.m:19-189 (flipped 6-state)
 *   Tools/NewCaseEKFEstimatorWithOptimalNPI.m:1-290 (+ MatlabCodeGenerator twin)
 *   Tools/SIalpha_Controlled.m, SI_Controlled.m, SEIRP.m,
 *   Tools/SEIRPSaturatedResource.m, Tools/NPICost.m
 *   Tools/Rt_ExpFitEKF.m                            (whole file)
 *   Tools/TrainPredictPrescribeNPI.m:142-198,201-202,240 (per-region preprocessing),
 *       :251-276 (NNLS regression between the EKF rounds), :496-521 (random-NPI
 *       scenarios), :624-633 (Pareto front and optimum)
 * MATLAB built-ins restated: pinv (symmetric argument: cyclic Jacobi
 * eigen-decomposition + tol = max(size)*eps(norm)), mrdivide for a square
 * right operand (LU with partial pivoting, LAPACK dgetf2/dgetrs operation
 * order), NaN-ignoring min/max (fmin/fmax), eps, squeeze; filter / filtfilt
 * (Signal Processing Toolbox) and lsqnonneg from their published definitions --
 * see the sections that use them.
 *
 * Matrix products are evaluated as MATLAB writes them, left to right.  Every
 * BLAS-class operation (matrix-matrix, matrix-vector and dot products -- what
 * MATLAB hands to its BLAS, which uses fused multiply-add on every CPU since
 * 2013) accumulates with fma():  acc = a(i,0)*b(0,j); acc = fma(a(i,k), b(k,j),
 * acc) for k = 1..m-1.  Scalar / element-wise MATLAB expressions (the model
 * maps, Jacobian entries, gains, clamps, the innovation monitor) are NOT fused:
 * each written * + - / is one IEEE rounding.  fma() is the C99 correctly-rounded
 * fused multiply-add (build with -mfma so it is one instruction); the HIP
 * kernels use the same fma() in the same places, so CPU and GPU results are
 * bit-identical.
 */
#include "ekf_oracle.h"
#include "../include/epiekf_layout.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MM 6            /* max state dimension */
#define DBL_EPS 2.220446049250313e-16 /* MATLAB eps */

/* ---------- small dense helpers (column-major, leading dimension m) ---------- */
#define IX(i, j, m) ((i) + (m) * (j))

static void mat_mul(int m, const double *A, const double *B, double *C) /* C = A*B */
{
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++) {
            double acc = A[IX(i, 0, m)] * B[IX(0, j, m)];
            for (int k = 1; k < m; k++) acc = fma(A[IX(i, k, m)], B[IX(k, j, m)], acc);
            C[IX(i, j, m)] = acc;
        }
}
static void mat_mul_bt(int m, const double *A, const double *B, double *C) /* C = A*B' */
{
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++) {
            double acc = A[IX(i, 0, m)] * B[IX(j, 0, m)];
            for (int k = 1; k < m; k++) acc = fma(A[IX(i, k, m)], B[IX(j, k, m)], acc);
            C[IX(i, j, m)] = acc;
        }
}
static void symmetrize(int m, double *P) /* P = (P + P')/2.0 */
{
    for (int j = 0; j < m; j++)
        for (int i = j + 1; i < m; i++) {
            double v = (P[IX(i, j, m)] + P[IX(j, i, m)]) / 2.0;
            double w = (P[IX(j, i, m)] + P[IX(i, j, m)]) / 2.0; /* commutative: v == w */
            P[IX(i, j, m)] = v;
            P[IX(j, i, m)] = w;
        }
    for (int i = 0; i < m; i++) P[IX(i, i, m)] = (P[IX(i, i, m)] + P[IX(i, i, m)]) / 2.0;
}

/* ---------- MATLAB pinv for a symmetric argument ----------
 * pinv(A): [U,S,V]=svd(A); tol = max(size(A))*eps(norm(s,inf)); keep s>tol;
 * X = V(:,keep)*diag(1./s(keep))*U(:,keep)'.  For symmetric A the singular
 * triplets are (|lambda_i|, sign(lambda_i) v_i, v_i).  The eigen-decomposition is
 * the cyclic Jacobi method (Rutishauser / Numerical-Recipes formulation with the
 * b/z accumulators), run on A scaled by an exact power of two.  The kept terms
 * are accumulated in the eigenvalue index order the iteration leaves them in;
 * only the upper triangle of X is accumulated and then mirrored. */
/* ---- exp and tanh with a fixed operation order ----------------------------------------------------------------
 * Rt_ExpFitEKF.m and SEIRPSaturatedResource.m call exp / tanh.  libm and the GPU's math library round them
 * differently, and a nearly singular 2 x 2 smoother gain amplifies that last-bit difference (seen: 1e-9 on
 * P_SMOOTH).  Oracle and kernels therefore evaluate the SAME sequence: k = rint(x/ln2), two-part Cody-Waite
 * reduction, degree-13 Taylor polynomial of expm1 in Horner form with fma, exact scaling by 2^k.  Error < 1 ulp for
 * exp, a few ulp for tanh (tests/test_oracle.py checks both against libm). */
static double epi_expm1_reduced(double r)
{
    /* sum_{n>=2} r^(n-2)/n!, Horner with fma; the constants are correctly rounded quotients on every IEEE compiler */
    double q = 1.0 / 6227020800.0;
    q = fma(q, r, 1.0 / 479001600.0);
    q = fma(q, r, 1.0 / 39916800.0);
    q = fma(q, r, 1.0 / 3628800.0);
    q = fma(q, r, 1.0 / 362880.0);
    q = fma(q, r, 1.0 / 40320.0);
    q = fma(q, r, 1.0 / 5040.0);
    q = fma(q, r, 1.0 / 720.0);
    q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    return fma(q * r, r, r);
}
static double epi_reduce_ln2(double y, double *k)
{
    *k = rint(y * 1.44269504088896338700e+00);
    double r = fma(-*k, 6.93147180369123816490e-01, y);      /* ln2 high part: 32 significant bits, k*hi exact */
    return fma(-*k, 1.90821492927058770002e-10, r);          /* ln2 low part */
}
static double epi_exp(double x)
{
    if (x != x) return x;
    if (x > 709.78271289338397) return (double)INFINITY;
    if (x < -745.13321910194122) return 0.0;
    double k;
    const double r = epi_reduce_ln2(x, &k);
    return ldexp(1.0 + epi_expm1_reduced(r), (int)k);
}
static double epi_tanh(double x)
{
    if (x != x) return x;
    const double ax = fabs(x);
    double res = 1.0;                                         /* |x| > 22: 1 - 2e-19 rounds to 1 */
    if (ax <= 22.0) {
        double k;
        const double r = epi_reduce_ln2(ax + ax, &k);
        const double q = epi_expm1_reduced(r);                /* e^(2|x|) = 2^k (1 + q) */
        const double s = ldexp(1.0, (int)k);                  /* tanh = (e - 1)/(e + 1), both formed with one rounding */
        res = fma(s, q, s - 1.0) / fma(s, q, s + 1.0);
    }
    return copysign(res, x);
}

double orc_exp(double x) { return epi_exp(x); }
double orc_tanh(double x) { return epi_tanh(x); }

static double eps_of(double x) /* MATLAB eps(x) for finite x >= 0 */
{
    if (x == 0.0) return 4.9406564584124654e-324;
    int e = ilogb(x);
    if (e < -1022) return 4.9406564584124654e-324;
    return ldexp(1.0, e - 52);
}

#define ORC_JACOBI_MAX_SWEEPS 50

static void jacobi_eig(int m, double *a /* in: sym matrix (destroyed) */, double *d, double *v, int skip_dead)
{
    double b[MM], z[MM];
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++) v[IX(i, j, m)] = (i == j) ? 1.0 : 0.0;
    for (int i = 0; i < m; i++) {
        b[i] = d[i] = a[IX(i, i, m)];
        z[i] = 0.0;
    }
    for (int sweep = 1; sweep <= ORC_JACOBI_MAX_SWEEPS; sweep++) {
        /* The caller (orc_sym_pinv) discards every eigenpair below tol = m*eps(max|d|).  Call an index "dead" when
         * its diagonal entry is below 2^-10 of that cut-off.  Two kinds of pairs are left alone for the sweep (neither
         * rotated, zeroed nor counted in the convergence sum), judged on the values at the START of the sweep:
         *   - both indices dead: the rotation only mixes directions that are discarded anyway;
         *   - one dead, one live, and 100|a_pq| vanishes against the live diagonal entry: the rotation angle is
         *     below eps/100, so the live eigenpair would not move; only the dead entry's relative accuracy is at
         *     stake, and that entry is discarded.
         * X changes by <1e-18 relative, the rotations by -30 %. */
        double dmax = 0.0;
        for (int i = 0; i < m; i++) dmax = fmax(dmax, fabs(d[i]));
        const double cut = ((double)m * eps_of(dmax)) * 0x1p-10;
        int dead[MM], la[MM * MM];
        for (int i = 0; i < m; i++) dead[i] = skip_dead && (fabs(d[i]) < cut);
        double sm = 0.0;
        for (int p = 0; p < m - 1; p++)
            for (int q = p + 1; q < m; q++) {
                const double apq = a[IX(p, q, m)];
                const double dl = fabs(d[dead[p] ? q : p]); /* the live entry of a mixed pair */
                la[IX(p, q, m)] = (dead[p] && dead[q]) || (dead[p] != dead[q] && (dl + 100.0 * fabs(apq)) == dl);
                if (!la[IX(p, q, m)]) sm = sm + fabs(apq);
            }
        if (sm == 0.0) break;
        double tresh = (sweep < 4) ? 0.2 * sm / (double)(m * m) : 0.0;
        for (int p = 0; p < m - 1; p++)
            for (int q = p + 1; q < m; q++) {
                if (la[IX(p, q, m)]) continue;
                double apq = a[IX(p, q, m)];
                double g = 100.0 * fabs(apq);
                if (sweep > 4 && (fabs(d[p]) + g) == fabs(d[p]) && (fabs(d[q]) + g) == fabs(d[q])) {
                    a[IX(p, q, m)] = 0.0;
                } else if (fabs(apq) > tresh) {
                    double h = d[q] - d[p];
                    double t;
                    if ((fabs(h) + g) == fabs(h)) {
                        t = apq / h;
                    } else {
                        /* t = sgn(theta)/(|theta| + sqrt(theta^2+1)), theta = h/(2 apq), multiplied
                         * through by |2 apq| (the matrix is pre-scaled to max|a| in [1,2): no overflow) */
                        double two_apq = 2.0 * apq;
                        t = two_apq / (fabs(h) + sqrt(fma(h, h, two_apq * two_apq)));
                        if (h < 0.0) t = -t;
                    }
                    /* c = 1/r, tau = s/(1+c) = t/(1+r) with r = sqrt(1+t^2): one division for both */
                    double r = sqrt(fma(t, t, 1.0));
                    double ir = 1.0 / fma(r, r, r);
                    double c = (1.0 + r) * ir;
                    double s = t * c;
                    double tau = (t * r) * ir;
                    h = t * apq;
                    z[p] = z[p] - h;
                    z[q] = z[q] + h;
                    d[p] = d[p] - h;
                    d[q] = d[q] + h;
                    a[IX(p, q, m)] = 0.0;
#define ROT(x, y)                                  \
    do {                                           \
        double g_ = (x), h_ = (y);                 \
        (x) = fma(-s, fma(g_, tau, h_), g_);       \
        (y) = fma(s, fma(-h_, tau, g_), h_);       \
    } while (0)
                    for (int j = 0; j < p; j++) ROT(a[IX(j, p, m)], a[IX(j, q, m)]);
                    for (int j = p + 1; j < q; j++) ROT(a[IX(p, j, m)], a[IX(j, q, m)]);
                    for (int j = q + 1; j < m; j++) ROT(a[IX(p, j, m)], a[IX(q, j, m)]);
                    for (int j = 0; j < m; j++) ROT(v[IX(j, p, m)], v[IX(j, q, m)]);
#undef ROT
                }
            }
        for (int i = 0; i < m; i++) {
            b[i] = b[i] + z[i];
            d[i] = b[i];
            z[i] = 0.0;
        }
    }
}

/* MATLAB pinv of a symmetric matrix through the eigen-decomposition by two-sided cyclic Jacobi (jacobi_eig above): valid
 * for ANY symmetric matrix.  orc_sym_pinv uses it for the matrices its faster route declines (indefinite ones). */
static int sym_pinv_two_sided(int m, const double *A, double *X)
{
    double a[MM * MM], d[MM], v[MM * MM];
    double amax = 0.0;
    for (int i = 0; i < m * m; i++) amax = fmax(amax, fabs(A[i]));
    for (int i = 0; i < m * m; i++) X[i] = 0.0;
    if (amax == 0.0) return 0;
    int e = ilogb(amax);
    /* only the upper triangle (i<=j) is referenced, as the iteration does */
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++) a[IX(i, j, m)] = ldexp(A[IX(i <= j ? i : j, i <= j ? j : i, m)], -e);
    jacobi_eig(m, a, d, v, m > 3); /* the eigenpairs below tol are discarded right below; 3 x 3: plain iteration */
    double smax = 0.0;
    for (int i = 0; i < m; i++) smax = fmax(smax, fabs(d[i]));
    double tol = (double)m * eps_of(smax);
    int rank = 0;
    for (int i = 0; i < m; i++) {
        double sv = fabs(d[i]);
        if (!(sv > tol)) continue;
        rank++;
        double inv = 1.0 / sv;
        double sg = (d[i] < 0.0) ? -1.0 : 1.0;
        /* the pseudo-inverse of a symmetric matrix is symmetric: the upper triangle is accumulated and
         * mirrored, so that X is symmetric bit for bit (and can be stored packed) */
        for (int c = 0; c < m; c++)
            for (int r = 0; r <= c; r++)
                X[IX(r, c, m)] = fma(v[IX(r, i, m)] * inv, sg * v[IX(c, i, m)], X[IX(r, c, m)]);
    }
    for (int c = 0; c < m; c++)
        for (int r = 0; r <= c; r++) {
            X[IX(r, c, m)] = ldexp(X[IX(r, c, m)], -e);
            X[IX(c, r, m)] = X[IX(r, c, m)];
        }
    return rank;
}

/* pinv(A), A symmetric (GenericExtendedKalmanFilter.m:215 applies it to the covariance P(k+1|k)): MATLAB's rule -- singular
 * values s, tol = max(size(A)) * eps(max(s)), keep s > tol -- evaluated for a positive semi-definite argument without
 * ever forming what the rule throws away:
 *   1. A, scaled by a power of two, is factored  A = G G' + S  by a Cholesky factorisation with diagonal pivoting (the pivot
 *      is SELECTED, rows stay where they are: column k of G belongs to the k-th pivot).  It ends when the trace of what is
 *      left, an upper bound of its eigenvalues, is below 2^-20 of MATLAB's cut-off (so dropping S moves the kept eigenvalues
 *      by < 1e-6 of the SMALLEST value the rule can keep), or when the pivot column violates a_ip^2 <= a_pp a_ii beyond
 *      rounding (tested for pivots at or below the cut-off's scale), i.e. when rounding noise has taken over.  For the filter's covariances 45 % of the singular values fall
 *      under the cut-off: G then has 2 or 3 columns instead of 6.
 *   2. one-sided Jacobi (Hestenes) rotations make the columns of G orthogonal; A's non-zero eigenvalues are then the
 *      squared column norms, its eigenvectors the normalised columns -- no eigenvector matrix is accumulated, and
 *      pre-conditioned by the pivoted factorisation the iteration needs 2 - 3.5 sweeps (Veselic & Hari 1989, Drmac 1997).
 *   3. X = sum over the kept columns of g g' / (g'g)^2.
 * Where the factorisation runs through all m pivots and a cheap certificate shows every eigenvalue above the cut-off,
 * steps 2-3 are skipped: X is the inverse, formed from the triangular factor (see the block after the loop).
 * A matrix that is not positive semi-definite up to rounding (a diagonal entry, or what is left when the factorisation
 * stops, that is not negligible against the cut-off) takes the two-sided Jacobi route above, which handles any
 * symmetric matrix.  On the headline sweep's covariances the two routes agree on the rank everywhere and on X to 2e-12,
 * and both stand at the same distance from a LAPACK SVD evaluation (tests/test_oracle.py). */
#define ORC_PINV_MAX_SWEEPS 30
/* route (may be NULL): 0 = factorisation + one-sided Jacobi, 1 = handed to the two-sided Jacobi, 2 = full rank certified: the
 * inverse from the factor; sweeps (may be NULL): one-sided sweeps run (the last one finds nothing to rotate) */
int orc_sym_pinv_ex(int m, const double *A, double *X, int *route, int *sweeps)
{
    if (route) *route = 0;
    if (sweeps) *sweeps = 0;
    double a[MM * MM], G[MM * MM];
    double amax = 0.0;
    for (int i = 0; i < m * m; i++) amax = fmax(amax, fabs(A[i]));
    for (int i = 0; i < m * m; i++) X[i] = 0.0;
    if (amax == 0.0) return 0;
    const int e = ilogb(amax);
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++) a[IX(i, j, m)] = ldexp(A[IX(i <= j ? i : j, i <= j ? j : i, m)], -e);
    double dmax0 = 0.0;
    for (int i = 0; i < m; i++) dmax0 = fmax(dmax0, a[IX(i, i, m)]);
    const double noise = (double)m * eps_of(dmax0);     /* the scale of MATLAB's cut-off (max(s) >= max diagonal entry) */
    const double stop = noise * 0x1p-20;
    int indefinite = 0;
    for (int i = 0; i < m; i++) indefinite |= (a[IX(i, i, m)] < -0.25 * noise);
    int used[MM] = {0}, order[MM] = {0};
    double ilv[MM];
    int r = 0;
    double tr0 = 0.0;                                    /* trace of the scaled matrix: >= its largest eigenvalue */
    for (int i = 0; i < m; i++) tr0 = tr0 + a[IX(i, i, m)];
    for (int i = 0; i < m * m; i++) G[i] = 0.0;
    for (int k = 0; k < m && !indefinite; k++) {
        int p = -1;
        double d = 0.0;
        for (int i = 0; i < m; i++)
            if (!used[i] && (p < 0 || a[IX(i, i, m)] > d)) { p = i; d = a[IX(i, i, m)]; }
        int quit = !(d * (double)(m - k) > stop);
        if (d <= noise)     /* (a pivot above the noise scale cannot be rounding noise: the test is not needed there) */
            for (int i = 0; i < m; i++)
                if (!used[i] && i != p) {
                    const double x = a[IX(i < p ? i : p, i < p ? p : i, m)];
                    quit |= (x * x > (4.0 * d) * fabs(a[IX(i, i, m)]));
                }
        if (quit) {
            double rest = 0.0;
            for (int j = 0; j < m; j++)
                for (int i = 0; i <= j; i++)
                    if (!used[i] && !used[j]) rest = fmax(rest, fabs(a[IX(i, j, m)]));
            indefinite |= (rest > 0.25 * noise);
            break;
        }
        const double l = sqrt(d), il = 1.0 / l;
        double col[MM];
        for (int i = 0; i < m; i++) col[i] = used[i] ? 0.0 : a[IX(i < p ? i : p, i < p ? p : i, m)] * il;
        col[p] = l;
        for (int i = 0; i < m; i++) G[IX(i, k, m)] = col[i];
        used[p] = 1;
        order[k] = p;
        ilv[k] = il;
        for (int j = 0; j < m; j++)
            for (int i = 0; i <= j; i++)
                if (!used[i] && !used[j]) a[IX(i, j, m)] = fma(-col[i], col[j], a[IX(i, j, m)]);
        for (int i = 0; i < m; i++) indefinite |= (!used[i] && a[IX(i, i, m)] < -0.25 * noise);
        r = k + 1;
    }
    if (indefinite) {
        if (route) *route = 1;
        return sym_pinv_two_sided(m, A, X);
    }
    if (r == m) {
        /* Every index has been a pivot: A is positive definite as far as the factorisation can tell, and if ALL its
         * eigenvalues lie above MATLAB's cut-off the pseudo-inverse is the inverse, which the factor gives directly:
         * with Lp the rows of G in pivot order (a lower triangle), A = P' Lp Lp' P, so X = P' (inv(Lp)' inv(Lp)) P.
         * The certificate needs no eigenvalue: lambda_min(A) >= 1 / trace(inv(A)) and trace(inv(A)) = the squared
         * Frobenius norm of inv(Lp); the cut-off is at most m eps(trace(A)).  Held with a factor 4 in hand; a matrix
         * that fails it (an eigenvalue within a few cut-offs of the cut-off) goes on to the Jacobi iteration below.
         * (The filter's covariances of the first ~100 days of the headline sweep, and every 3 x 3 one: 4 300 -> 1 300
         * instructions per matrix on the GPU.) */
        double Lp[MM * MM], Li[MM * MM], Xp[MM * MM];
        for (int k = 0; k < m; k++)
            for (int c = 0; c <= k; c++) Lp[IX(k, c, m)] = G[IX(order[k], c, m)];
        double fro = 0.0;
        for (int j = 0; j < m; j++) {                      /* column j of inv(Lp), top to bottom */
            Li[IX(j, j, m)] = ilv[j];
            for (int i = j + 1; i < m; i++) {
                double acc = Lp[IX(i, j, m)] * Li[IX(j, j, m)];
                for (int k = j + 1; k < i; k++) acc = fma(Lp[IX(i, k, m)], Li[IX(k, j, m)], acc);
                Li[IX(i, j, m)] = -(acc * ilv[i]);
            }
        }
        for (int j = 0; j < m; j++)
            for (int i = j; i < m; i++) fro = fma(Li[IX(i, j, m)], Li[IX(i, j, m)], fro);
        if (4.0 * ((double)m * eps_of(tr0)) * fro < 1.0) {
            for (int j = 0; j < m; j++)                     /* Xp = inv(Lp)' inv(Lp), upper triangle */
                for (int i = 0; i <= j; i++) {
                    double acc = Li[IX(j, i, m)] * Li[IX(j, j, m)];
                    for (int k = j + 1; k < m; k++) acc = fma(Li[IX(k, i, m)], Li[IX(k, j, m)], acc);
                    Xp[IX(i, j, m)] = acc;
                    Xp[IX(j, i, m)] = acc;
                }
            for (int j = 0; j < m; j++)
                for (int i = 0; i < m; i++) X[IX(order[i], order[j], m)] = ldexp(Xp[IX(i, j, m)], -e);
            if (route) *route = 2;
            return m;
        }
    }
    for (int sweep = 1; sweep <= ORC_PINV_MAX_SWEEPS; sweep++) {
        int rotated = 0;
        if (sweeps) *sweeps = sweep;
        for (int p = 0; p < r - 1; p++)
            for (int q = p + 1; q < r; q++) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int i = 0; i < m; i++) {
                    al = fma(G[IX(i, p, m)], G[IX(i, p, m)], al);
                    be = fma(G[IX(i, q, m)], G[IX(i, q, m)], be);
                    ga = fma(G[IX(i, p, m)], G[IX(i, q, m)], ga);
                }
                if (!(ga * ga > 0x1p-106 * (al * be))) continue;   /* orthogonal to working precision */
                rotated = 1;
                /* tan = sgn(zeta)/(|zeta| + sqrt(zeta^2+1)), zeta = (be - al)/(2 ga); multiplied through by |2 ga|:
                 * tan = 2 ga / (|h| + w), h = be - al, w = sqrt(h^2 + (2 ga)^2), and 1 + tan^2 = 2 w (|h| + w) / (|h| + w)^2,
                 * so cos = (|h| + w) / D and sin = 2 ga / D with D = sqrt(2 w (|h| + w)): two square roots, ONE division */
                const double h = be - al, two = 2.0 * ga;
                const double w = sqrt(fma(h, h, two * two)), sum = fabs(h) + w;
                const double iD = 1.0 / sqrt((2.0 * w) * sum);
                const double c = sum * iD;
                double s = two * iD;
                if (h < 0.0) s = -s;
                for (int i = 0; i < m; i++) {
                    const double gp = G[IX(i, p, m)], gq = G[IX(i, q, m)];
                    G[IX(i, p, m)] = fma(c, gp, -(s * gq));
                    G[IX(i, q, m)] = fma(s, gp, c * gq);
                }
            }
        if (!rotated) break;
    }
    double lam[MM], lmax = 0.0;
    for (int k = 0; k < r; k++) {
        double s2 = 0.0;
        for (int i = 0; i < m; i++) s2 = fma(G[IX(i, k, m)], G[IX(i, k, m)], s2);
        lam[k] = s2;
        lmax = fmax(lmax, s2);
    }
    const double tol = (double)m * eps_of(lmax);
    int rank = 0;
    for (int k = 0; k < r; k++) {
        if (!(lam[k] > tol)) continue;
        rank++;
        const double w = 1.0 / (lam[k] * lam[k]);
        /* symmetric by construction: the upper triangle is accumulated and mirrored (X can be stored packed) */
        for (int c = 0; c < m; c++)
            for (int rr = 0; rr <= c; rr++) X[IX(rr, c, m)] = fma(G[IX(rr, k, m)] * w, G[IX(c, k, m)], X[IX(rr, c, m)]);
    }
    for (int c = 0; c < m; c++)
        for (int rr = 0; rr <= c; rr++) {
            X[IX(rr, c, m)] = ldexp(X[IX(rr, c, m)], -e);
            X[IX(c, rr, m)] = X[IX(rr, c, m)];
        }
    return rank;
}

int orc_sym_pinv(int m, const double *A, double *X) { return orc_sym_pinv_ex(m, A, X, NULL, NULL); }

/* ---------- MATLAB mrdivide, square right operand: X = B/A = (A'\B')' ----------
 * dgetf2 (unblocked right-looking LU, first-max partial pivoting, reciprocal
 * scaling of the sub-column as in the reference LAPACK when |pivot| >= sfmin)
 * followed by dgetrs (dlaswp, unit-lower dtrsm, upper dtrsm; reference-BLAS
 * column-oriented operation order). */
void orc_mrdivide(int m, const double *B, const double *A, double *X)
{
    double M[MM * MM], Y[MM * MM];
    int piv[MM];
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++) {
            M[IX(i, j, m)] = A[IX(j, i, m)]; /* A' */
            Y[IX(i, j, m)] = B[IX(j, i, m)]; /* B' */
        }
    for (int j = 0; j < m; j++) {
        int p = j;
        double best = fabs(M[IX(j, j, m)]);
        for (int i = j + 1; i < m; i++) {
            double v = fabs(M[IX(i, j, m)]);
            if (v > best) { best = v; p = i; }
        }
        piv[j] = p;
        if (M[IX(p, j, m)] != 0.0) {
            if (p != j)
                for (int c = 0; c < m; c++) {
                    double t = M[IX(j, c, m)];
                    M[IX(j, c, m)] = M[IX(p, c, m)];
                    M[IX(p, c, m)] = t;
                }
            if (fabs(M[IX(j, j, m)]) >= 2.2250738585072014e-308) {
                double r = 1.0 / M[IX(j, j, m)];
                for (int i = j + 1; i < m; i++) M[IX(i, j, m)] = M[IX(i, j, m)] * r;
            } else {
                for (int i = j + 1; i < m; i++) M[IX(i, j, m)] = M[IX(i, j, m)] / M[IX(j, j, m)];
            }
        }
        for (int c = j + 1; c < m; c++)
            for (int i = j + 1; i < m; i++)
                M[IX(i, c, m)] = M[IX(i, c, m)] - M[IX(i, j, m)] * M[IX(j, c, m)];
    }
    /* row interchanges on the right-hand sides */
    for (int j = 0; j < m; j++)
        if (piv[j] != j)
            for (int c = 0; c < m; c++) {
                double t = Y[IX(j, c, m)];
                Y[IX(j, c, m)] = Y[IX(piv[j], c, m)];
                Y[IX(piv[j], c, m)] = t;
            }
    for (int c = 0; c < m; c++) {
        /* L y = b, unit diagonal */
        for (int k = 0; k < m; k++)
            if (Y[IX(k, c, m)] != 0.0)
                for (int i = k + 1; i < m; i++)
                    Y[IX(i, c, m)] = Y[IX(i, c, m)] - Y[IX(k, c, m)] * M[IX(i, k, m)];
        /* U x = y */
        for (int k = m - 1; k >= 0; k--)
            if (Y[IX(k, c, m)] != 0.0) {
                Y[IX(k, c, m)] = Y[IX(k, c, m)] / M[IX(k, k, m)];
                for (int i = 0; i < k; i++)
                    Y[IX(i, c, m)] = Y[IX(i, c, m)] - Y[IX(k, c, m)] * M[IX(i, k, m)];
            }
    }
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++) X[IX(i, j, m)] = Y[IX(j, i, m)];
}

/* ---------- model callbacks ---------- */
typedef struct model_ops {
    int m;
    int flipped;       /* sign of every dt term reversed */
    int lo_is_zero;    /* s,i clamps use 0 instead of s_min/i_min */
    int phi_ge;        /* bang-bang test phi >= 0 (NewCase...:175) instead of phi > 0 */
    int obs_clamp;     /* ObsHardMargins = max(0,.) ; 0 in the codegen twin */
    int obs_type_fixed;/* codegen twin: always NEWCASES */
} model_ops;

static const model_ops MODEL_TABLE[6] = {
    /* SIA3          */ {3, 0, 0, 0, 1, 0},
    /* SIA6          */ {6, 0, 1, 0, 1, 0},
    /* SIA3_BWD      */ {3, 1, 1, 0, 1, 0},
    /* SIA6_BWD      */ {6, 1, 1, 0, 1, 0},
    /* NEWCASE6      */ {6, 0, 1, 1, 1, 0},
    /* NEWCASE6 twin */ {6, 0, 1, 1, 0, 1},
};

int orc_model_dim(int model) { return (model >= 0 && model < 6) ? MODEL_TABLE[model].m : -1; }

/* StateHardMargins: SIAlphaModelEKF.m:27-31 ; OptControlled :27-31 ; Backward :48-52 */
static void state_hard_margins(const model_ops *mo, const orc_params *p, double *s)
{
    double slo = mo->lo_is_zero ? 0.0 : p->s_min;
    double ilo = mo->lo_is_zero ? 0.0 : p->i_min;
    s[0] = fmin(1.0, fmax(slo, s[0]));
    s[1] = fmin(1.0, fmax(ilo, s[1]));
    s[2] = fmin(p->alpha_max, fmax(p->alpha_min, s[2]));
}

/* bang-bang substitution: SIAlphaModelEKFOptControlled.m:49-58 */
static void resolve_control(const model_ops *mo, const orc_params *p, const double *s, double *u)
{
    if (mo->m != 6) return;
    for (int kk = 0; kk < p->n_npi; kk++) {
        if (isnan(u[kk])) {
            double phi = p->epsilon * p->w_eff[kk] - p->gamma * s[5] * p->a[kk];
            int lo = mo->phi_ge ? (phi >= 0.0) : (phi > 0.0);
            u[kk] = lo ? p->u_min[kk] : p->u_max[kk];
        }
    }
}

/* NlinStateUpdate: SIAlphaModelEKF.m:39-48 ; OptControlled :39-74 ; flipped: Backward*.m:60-95 */
static void nlin_state_update(const model_ops *mo, const orc_params *p, double *u /* in/out */,
                              const double *s, double *sn)
{
    double sg = mo->flipped ? -1.0 : 1.0;
    double slo = mo->lo_is_zero ? 0.0 : p->s_min;
    double ilo = mo->lo_is_zero ? 0.0 : p->i_min;
    resolve_control(mo, p, s, u);
    /* params.gamma * params.a'*(params.u_max - u): MATLAB evaluates left to
     * right, i.e. the row vector (gamma*a') times the column (u_max - u) */
    double dot = 0.0;
    for (int kk = 0; kk < p->n_npi; kk++) {
        dot = (kk == 0) ? (p->gamma * p->a[kk]) * (p->u_max[kk] - u[kk])
                        : fma(p->gamma * p->a[kk], p->u_max[kk] - u[kk], dot);
    }
    double asi = s[2] * s[0] * s[1]; /* s_k(3) * s_k(1) * s_k(2) */
    double f3 = -p->gamma * s[2] + p->gamma * p->b + dot;
    if (!mo->flipped) {
        sn[0] = fmax(slo, fmin(1.0, s[0] - p->dt * s[2] * s[0] * s[1]));
        sn[1] = fmax(ilo, fmin(1.0, s[1] + p->dt * (asi - p->beta * s[1])));
        sn[2] = fmax(p->alpha_min, fmin(p->alpha_max, s[2] + p->dt * f3));
    } else {
        sn[0] = fmax(slo, fmin(1.0, s[0] + p->dt * s[2] * s[0] * s[1]));
        sn[1] = fmax(ilo, fmin(1.0, s[1] - p->dt * (asi - p->beta * s[1])));
        sn[2] = fmax(p->alpha_min, fmin(p->alpha_max, s[2] - p->dt * f3));
    }
    if (mo->m == 6) {
        double rho = s[3] - s[4] - (1.0 - p->epsilon);
        double g4 = p->dt * rho * s[2] * s[1];
        double g5 = p->dt * (rho * s[2] * s[0] + p->beta * s[4]);
        double g6 = p->dt * (rho * s[0] * s[1] + p->gamma * s[5]);
        sn[3] = s[3] + sg * g4;
        sn[4] = s[4] + sg * g5;
        sn[5] = s[5] + sg * g6;
    }
}

/* NlinObsUpdate + ObsHardMargins: SIAlphaModelEKF.m:34-36,51-59 */
static int predict_obs(const model_ops *mo, const orc_params *p, const double *s, double v_bar, double *xk)
{
    int ot = mo->obs_type_fixed ? ORC_OBS_NEWCASES : p->obs_type;
    double v;
    if (ot == ORC_OBS_NEWCASES) v = s[0] * s[1] * s[2] + v_bar;
    else if (ot == ORC_OBS_TOTALCASES) v = 1.0 - s[0] + v_bar;
    else return ORC_ERR_OBS_TYPE;
    *xk = mo->obs_clamp ? fmax(0.0, v) : v;
    return ORC_OK;
}

/* ObsJacobian: SIAlphaModelEKF.m:79-89 ; OptControlled :138-148 */
static int obs_jacobian(const model_ops *mo, const orc_params *p, const double *s, double *C)
{
    int ot = p->obs_type; /* the codegen twin's ObsJacobian keeps the obs_type test only in Tools/ */
    if (mo->obs_type_fixed) ot = ORC_OBS_NEWCASES;
    for (int i = 0; i < mo->m; i++) C[i] = 0.0;
    if (ot == ORC_OBS_NEWCASES) {
        C[0] = s[1] * s[2];
        C[1] = s[0] * s[2];
        C[2] = s[0] * s[1];
    } else if (ot == ORC_OBS_TOTALCASES) {
        C[0] = -1.0;
    } else return ORC_ERR_OBS_TYPE;
    return ORC_OK;
}

/* StateJacobians: SIAlphaModelEKF.m:62-76 ; OptControlled :89-135 ; flipped twins */
static void state_jacobians(const model_ops *mo, const orc_params *p, const double *u /* original, NaNs kept */,
                            const double *s, double *A)
{
    int m = mo->m;
    double dt = p->dt;
    for (int i = 0; i < m * m; i++) A[i] = 0.0;
    if (!mo->flipped) {
        A[IX(0, 0, m)] = 1.0 - dt * s[2] * s[1];
        A[IX(0, 1, m)] = -dt * s[2] * s[0];
        A[IX(0, 2, m)] = -dt * s[0] * s[1];
        A[IX(1, 0, m)] = dt * s[1] * s[2];
        A[IX(1, 1, m)] = 1.0 + dt * (s[0] * s[2] - p->beta);
        A[IX(1, 2, m)] = dt * s[0] * s[1];
        A[IX(2, 2, m)] = 1.0 - dt * p->gamma;
    } else {
        A[IX(0, 0, m)] = 1.0 + dt * s[2] * s[1];
        A[IX(0, 1, m)] = dt * s[2] * s[0];
        A[IX(0, 2, m)] = dt * s[0] * s[1];
        A[IX(1, 0, m)] = -dt * s[1] * s[2];
        A[IX(1, 1, m)] = 1.0 - dt * (s[0] * s[2] - p->beta);
        A[IX(1, 2, m)] = -dt * s[0] * s[1];
        A[IX(2, 2, m)] = 1.0 + dt * p->gamma;
    }
    if (m != 6) return;
    /* linear-slope term :107-114 */
    for (int kk = 0; kk < p->n_npi; kk++) {
        if (isnan(u[kk])) {
            double phi = p->epsilon * p->w_eff[kk] - p->gamma * s[5] * p->a[kk];
            if (phi > -1.0 / p->sigma && phi < 1.0 / p->sigma) {
                double term = p->gamma * dt * (p->sigma / 2.0) * p->a[kk] * (p->u_max[kk] - p->u_min[kk]);
                if (!mo->flipped) A[IX(2, 5, m)] = A[IX(2, 5, m)] - term;
                else A[IX(2, 5, m)] = A[IX(2, 5, m)] + term;
            }
        }
    }
    double rho = s[3] - s[4] - (1.0 - p->epsilon);
    if (!mo->flipped) {
        A[IX(3, 1, m)] = dt * s[2] * rho;
        A[IX(3, 2, m)] = dt * s[1] * rho;
        A[IX(3, 3, m)] = 1.0 + dt * s[1] * s[2];
        A[IX(3, 4, m)] = -dt * s[1] * s[2];
        A[IX(4, 0, m)] = dt * s[2] * rho;
        A[IX(4, 2, m)] = dt * s[0] * rho;
        A[IX(4, 3, m)] = dt * s[0] * s[2];
        A[IX(4, 4, m)] = 1.0 - dt * (s[0] * s[2] - p->beta);
        A[IX(5, 0, m)] = dt * s[1] * rho;
        A[IX(5, 1, m)] = dt * s[0] * rho;
        A[IX(5, 3, m)] = dt * s[0] * s[1];
        A[IX(5, 4, m)] = -dt * s[0] * s[1];
        A[IX(5, 5, m)] = 1.0 + dt * p->gamma;
    } else {
        A[IX(3, 1, m)] = -dt * s[2] * rho;
        A[IX(3, 2, m)] = -dt * s[1] * rho;
        A[IX(3, 3, m)] = 1.0 - dt * s[1] * s[2];
        A[IX(3, 4, m)] = dt * s[1] * s[2];
        A[IX(4, 0, m)] = -dt * s[2] * rho;
        A[IX(4, 2, m)] = -dt * s[0] * rho;
        A[IX(4, 3, m)] = -dt * s[0] * s[2];
        A[IX(4, 4, m)] = 1.0 + dt * (s[0] * s[2] - p->beta);
        A[IX(5, 0, m)] = -dt * s[1] * rho;
        A[IX(5, 1, m)] = -dt * s[0] * rho;
        A[IX(5, 3, m)] = -dt * s[0] * s[1];
        A[IX(5, 4, m)] = dt * s[0] * s[1];
        A[IX(5, 5, m)] = 1.0 - dt * p->gamma;
    }
}

static int has_nonfinite(const double *P, int n)
{
    for (int i = 0; i < n; i++)
        if (isnan(P[i]) || isinf(P[i])) return 1;
    return 0;
}

/* ---------- the filter ---------- */
static int ekf_core(int model, int T, const double *u_in, const double *x, const orc_params *prm,
                    const double *s_init, const double *Ps_init, const double *s_final,
                    const double *Ps_final, double v_bar, const double *Q_w, int q_len,
                    const double *R_v, int r_len, double beta, double gamma, int L, int order,
                    double *u_opt, double *u_opt_smooth, double *S_MINUS, double *S_PLUS,
                    double *S_SMOOTH, double *P_MINUS, double *P_PLUS, double *P_SMOOTH,
                    double *K_GAIN, double *innovations, double *rho, int *pinv_rank)
{
    const model_ops *mo = &MODEL_TABLE[model];
    const int m = mo->m, mm = m * m, nn = prm->n_npi;
    const int generic = (model <= ORC_MODEL_SIA6_BWD); /* GenericExtendedKalmanFilter vs NewCase... */
    if (T < 1 || L < 1 || nn < 1 || nn > ORC_MAX_NPI) return ORC_ERR_BAD_ARG;
    if (q_len != 1 && q_len != T) return ORC_ERR_Q_MISMATCH;
    if (r_len != 1 && r_len != T) return ORC_ERR_R_MISMATCH;
    if (!generic && (q_len != 1 || r_len != 1)) return ORC_ERR_BAD_ARG; /* Q = Q_w; R = R_v scalars/matrices */
    if (order != 1 && order != 2) return ORC_ERR_UNDEFINED_ORDER;

    double *R = (double *)malloc(sizeof(double) * (size_t)T);
    double *winMean = (double *)calloc((size_t)L, sizeof(double));
    double *winCov = (double *)calloc((size_t)L, sizeof(double));
    double *winCovN = (double *)calloc((size_t)L, sizeof(double));
    double *uo_s = (double *)malloc(sizeof(double) * (size_t)nn * (size_t)T);
    int rc = ORC_OK;
    const int fixed_R = (r_len == 1); /* GenericEKF:79-85 */
    for (int k = 0; k < T; k++) R[k] = fixed_R ? R_v[0] : R_v[k];
    double Rs = R_v[0]; /* NewCase...:31 scalar running R */

    double sk_minus[MM], Pk_minus[MM * MM], sk_plus[MM], Pk_plus[MM * MM];
    double C[MM], K[MM], A[MM * MM], T1[MM * MM], T2[MM * MM], IKC[MM * MM], uk[ORC_MAX_NPI];
    for (int i = 0; i < m; i++) sk_minus[i] = s_init[i];
    for (int i = 0; i < mm; i++) Pk_minus[i] = Ps_init[i];

    /* Forward Kalman filtering stage: GenericEKF:98-186 / NewCase...:37-113 */
    for (int k = 0; k < T; k++) {
        const double *Qk = Q_w + (q_len == 1 ? 0 : (size_t)mm * k);
        double Rk = generic ? R[k] : Rs;
        if (S_MINUS) memcpy(S_MINUS + (size_t)m * k, sk_minus, sizeof(double) * m);
        if (P_MINUS) memcpy(P_MINUS + (size_t)mm * k, Pk_minus, sizeof(double) * mm);

        if ((rc = obs_jacobian(mo, prm, sk_minus, C)) != ORC_OK) goto done;
        double xk_minus;
        if ((rc = predict_obs(mo, prm, sk_minus, v_bar, &xk_minus)) != ORC_OK) goto done;

        double innov;
        if (!isnan(x[k])) {
            innov = x[k] - xk_minus;
            /* Kgain = Pk_minus*Ck' / (Ck*Pk_minus*Ck' + gamma*(D*R*D')) */
            double PCt[MM];
            for (int i = 0; i < m; i++) {
                double acc = Pk_minus[IX(i, 0, m)] * C[0];
                for (int j = 1; j < m; j++) acc = fma(Pk_minus[IX(i, j, m)], C[j], acc);
                PCt[i] = acc;
            }
            double CP[MM]; /* Ck*Pk_minus (row vector) */
            for (int j = 0; j < m; j++) {
                double acc = C[0] * Pk_minus[IX(0, j, m)];
                for (int i = 1; i < m; i++) acc = fma(C[i], Pk_minus[IX(i, j, m)], acc);
                CP[j] = acc;
            }
            double CPCt = CP[0] * C[0];
            for (int j = 1; j < m; j++) CPCt = fma(CP[j], C[j], CPCt);
            double den = CPCt + gamma * Rk;
            for (int i = 0; i < m; i++) K[i] = PCt[i] / den;
            /* eye(m) - Kgain*Ck */
            for (int j = 0; j < m; j++)
                for (int i = 0; i < m; i++) IKC[IX(i, j, m)] = ((i == j) ? 1.0 : 0.0) - K[i] * C[j];
            if (generic) {
                /* ((I-KC)*P*(I-KC)' + K*(D*R*D')*K')/gamma   GenericEKF:127 */
                mat_mul(m, IKC, Pk_minus, T1);
                mat_mul_bt(m, T1, IKC, T2);
                for (int j = 0; j < m; j++)
                    for (int i = 0; i < m; i++)
                        Pk_plus[IX(i, j, m)] = (T2[IX(i, j, m)] + (K[i] * Rk) * K[j]) / gamma;
            } else {
                /* (I-KC)*P/gamma   NewCase...:64 */
                mat_mul(m, IKC, Pk_minus, T1);
                for (int i = 0; i < mm; i++) Pk_plus[i] = T1[i] / gamma;
            }
            for (int i = 0; i < m; i++) sk_plus[i] = sk_minus[i] + K[i] * innov;
        } else {
            innov = 0.0;
            for (int i = 0; i < m; i++) K[i] = 0.0;
            for (int i = 0; i < mm; i++) Pk_plus[i] = Pk_minus[i];
            for (int i = 0; i < m; i++) sk_plus[i] = sk_minus[i];
        }
        if (generic) symmetrize(m, Pk_plus); /* :138 */
        state_hard_margins(mo, prm, sk_plus); /* :141 */

        /* state propagation :155-164 */
        for (int kk = 0; kk < nn; kk++) uk[kk] = u_in[kk + (size_t)nn * k];
        nlin_state_update(mo, prm, uk, sk_plus, sk_minus);
        if (u_opt) memcpy(u_opt + (size_t)nn * k, uk, sizeof(double) * nn);
        state_jacobians(mo, prm, u_in + (size_t)nn * k, sk_plus, A);
        mat_mul(m, A, Pk_plus, T1);
        mat_mul_bt(m, T1, A, T2);
        for (int i = 0; i < mm; i++) Pk_minus[i] = T2[i] + Qk[i]; /* B = I */
        if (generic) symmetrize(m, Pk_minus); /* :161 */
        state_hard_margins(mo, prm, sk_minus); /* :164 */

        if (S_PLUS) memcpy(S_PLUS + (size_t)m * k, sk_plus, sizeof(double) * m);
        if (P_PLUS) memcpy(P_PLUS + (size_t)mm * k, Pk_plus, sizeof(double) * mm);
        if (K_GAIN) memcpy(K_GAIN + (size_t)m * k, K, sizeof(double) * m);
        if (innovations) innovations[k] = innov;

        /* innovation monitor :172-185 (windows are newest-first, summed front to back) */
        int cnt = (k + 1 < L) ? (k + 1) : L;
        memmove(winMean + 1, winMean, sizeof(double) * (size_t)(L - 1));
        winMean[0] = innov;
        double sum = winMean[0];
        for (int j = 1; j < L; j++) sum = sum + winMean[j];
        double mu = sum / (double)cnt;
        double cc = (innov - mu) * (innov - mu);
        memmove(winCov + 1, winCov, sizeof(double) * (size_t)(L - 1));
        winCov[0] = cc;
        memmove(winCovN + 1, winCovN, sizeof(double) * (size_t)(L - 1));
        winCovN[0] = generic ? cc / (Rk + DBL_EPS) : cc / Rk;
        double sumN = winCovN[0];
        for (int j = 1; j < L; j++) sumN = sumN + winCovN[j];
        if (rho) rho[k] = sumN / (double)cnt;
        if (generic) {
            if (beta != 1.0 && !isnan(x[k]) && fixed_R && k < T - 1) {
                double sumC = winCov[0];
                for (int j = 1; j < L; j++) sumC = sumC + winCov[j];
                R[k + 1] = beta * R[k] + (1.0 - beta) * (sumC / (double)cnt);
            }
        } else {
            if (beta != 1.0 && !isnan(x[k])) {
                double sumC = winCov[0];
                for (int j = 1; j < L; j++) sumC = sumC + winCov[j];
                Rs = beta * Rs + (1.0 - beta) * sumC / (double)cnt;
            }
        }
    }

    /* Backward smoothing stage: GenericEKF:189-230 / NewCase...:115-139.
     * It needs the stored forward quantities; if the caller did not ask for
     * them we cannot smooth, so the batch driver always supplies scratch. */
    if (S_SMOOTH && P_SMOOTH && S_MINUS && S_PLUS && P_MINUS && P_PLUS) {
        double Ss[MM], Ps[MM * MM], J[MM * MM], Xp[MM * MM], D[MM * MM], dv[MM];
        for (int i = 0; i < m; i++) Ss[i] = S_PLUS[i + (size_t)m * (T - 1)];
        for (int i = 0; i < mm; i++) Ps[i] = P_PLUS[i + (size_t)mm * (T - 1)];
        for (int i = 0; i < m; i++)
            if (!isnan(s_final[i])) Ss[i] = s_final[i];
        if (generic) {
            for (int i = 0; i < mm; i++)
                if (!isnan(Ps_final[i])) Ps[i] = Ps_final[i];
        } else {
            /* P_SMOOTH(row, col, T) = Ps_final(row, col): cross-product sub-assignment :125-127 */
            int rows[MM] = {0}, cols[MM] = {0}, any = 0;
            for (int j = 0; j < m; j++)
                for (int i = 0; i < m; i++)
                    if (!isnan(Ps_final[IX(i, j, m)])) { rows[i] = 1; cols[j] = 1; any = 1; }
            if (any)
                for (int j = 0; j < m; j++)
                    for (int i = 0; i < m; i++)
                        if (rows[i] && cols[j]) Ps[IX(i, j, m)] = Ps_final[IX(i, j, m)];
        }
        memcpy(S_SMOOTH + (size_t)m * (T - 1), Ss, sizeof(double) * m);
        memcpy(P_SMOOTH + (size_t)mm * (T - 1), Ps, sizeof(double) * mm);
        for (int kk = 0; kk < nn; kk++) uo_s[kk + (size_t)nn * (T - 1)] = 0.0; /* column T never written :95,204 */
        if (pinv_rank) pinv_rank[T - 1] = -1;

        for (int k = T - 2; k >= 0; k--) {
            const double *Sp = S_PLUS + (size_t)m * k;
            const double *Pp = P_PLUS + (size_t)mm * k;
            const double *Sm1 = S_MINUS + (size_t)m * (k + 1);
            const double *Pm1 = P_MINUS + (size_t)mm * (k + 1);
            state_jacobians(mo, prm, u_in + (size_t)nn * k, Sp, A);
            int rank = -1;
            if (generic) {
                if (has_nonfinite(Pm1, mm)) {
                    for (int i = 0; i < mm; i++) J[i] = 0.0; /* :211-213 */
                } else {
                    mat_mul_bt(m, Pp, A, T1); /* P_PLUS*A' */
                    rank = orc_sym_pinv(m, Pm1, Xp);
                    mat_mul(m, T1, Xp, J);
                }
            } else {
                mat_mul_bt(m, Pp, A, T1);
                orc_mrdivide(m, T1, Pm1, J); /* :132 */
            }
            if (pinv_rank) pinv_rank[k] = rank;
            for (int i = 0; i < m; i++) dv[i] = Ss[i] - Sm1[i];
            double Sn[MM];
            for (int i = 0; i < m; i++) {
                double acc = J[IX(i, 0, m)] * dv[0];
                for (int j = 1; j < m; j++) acc = fma(J[IX(i, j, m)], dv[j], acc);
                Sn[i] = Sp[i] + acc;
            }
            state_hard_margins(mo, prm, Sn);
            for (int i = 0; i < mm; i++) D[i] = Pm1[i] - Ps[i];
            mat_mul(m, J, D, T1);
            mat_mul_bt(m, T1, J, T2);
            for (int i = 0; i < mm; i++) Ps[i] = Pp[i] - T2[i];
            if (generic) symmetrize(m, Ps);
            for (int i = 0; i < m; i++) Ss[i] = Sn[i];
            memcpy(S_SMOOTH + (size_t)m * k, Ss, sizeof(double) * m);
            memcpy(P_SMOOTH + (size_t)mm * k, Ps, sizeof(double) * mm);
            if (generic) {
                /* rerun the state equation to find the optimal input :229 */
                double sn_dummy[MM];
                for (int kk = 0; kk < nn; kk++) uk[kk] = u_in[kk + (size_t)nn * k];
                nlin_state_update(mo, prm, uk, Ss, sn_dummy);
                memcpy(uo_s + (size_t)nn * k, uk, sizeof(double) * nn);
            }
        }
        if (generic && u_opt_smooth) memcpy(u_opt_smooth, uo_s, sizeof(double) * (size_t)nn * T);
    }
done:
    free(R); free(winMean); free(winCov); free(winCovN); free(uo_s);
    return rc;
}

static void flip_cols(double *a, int rows, int T)
{
    if (!a) return;
    for (int k = 0; k < T / 2; k++)
        for (int i = 0; i < rows; i++) {
            double t = a[i + (size_t)rows * k];
            a[i + (size_t)rows * k] = a[i + (size_t)rows * (T - 1 - k)];
            a[i + (size_t)rows * (T - 1 - k)] = t;
        }
}

int orc_ekf_run(int model, int T, const double *u, const double *x, const orc_params *prm,
                const double *s_init, const double *Ps_init, const double *s_final,
                const double *Ps_final, double v_bar, const double *Q_w, int q_len,
                const double *R_v, int r_len, double beta, double gamma, int L, int order,
                double *u_opt, double *u_opt_smooth, double *S_MINUS, double *S_PLUS,
                double *S_SMOOTH, double *P_MINUS, double *P_PLUS, double *P_SMOOTH,
                double *K_GAIN, double *innovations, double *rho, int *pinv_rank)
{
    if (model < 0 || model > 5 || T < 1) return ORC_ERR_BAD_ARG;
    const model_ops *mo = &MODEL_TABLE[model];
    const int m = mo->m, mm = m * m, nn = prm->n_npi;
    /* scratch for forward quantities the smoother needs when the caller passes NULL */
    double *sm_ = S_MINUS ? NULL : (double *)malloc(sizeof(double) * (size_t)m * T);
    double *sp_ = S_PLUS ? NULL : (double *)malloc(sizeof(double) * (size_t)m * T);
    double *pm_ = P_MINUS ? NULL : (double *)malloc(sizeof(double) * (size_t)mm * T);
    double *pp_ = P_PLUS ? NULL : (double *)malloc(sizeof(double) * (size_t)mm * T);
    double *ss_ = S_SMOOTH ? NULL : (double *)malloc(sizeof(double) * (size_t)m * T);
    double *ps_ = P_SMOOTH ? NULL : (double *)malloc(sizeof(double) * (size_t)mm * T);
    double *SM = S_MINUS ? S_MINUS : sm_, *SP = S_PLUS ? S_PLUS : sp_;
    double *PM = P_MINUS ? P_MINUS : pm_, *PP = P_PLUS ? P_PLUS : pp_;
    double *SS = S_SMOOTH ? S_SMOOTH : ss_, *PS = P_SMOOTH ? P_SMOOTH : ps_;
    int rc;
    if (!mo->flipped) {
        rc = ekf_core(model, T, u, x, prm, s_init, Ps_init, s_final, Ps_final, v_bar, Q_w, q_len, R_v,
                      r_len, beta, gamma, L, order, u_opt, u_opt_smooth, SM, SP, SS, PM, PP, PS,
                      K_GAIN, innovations, rho, pinv_rank);
    } else {
        /* Backward wrappers: SIAlphaModelBackwardEKF.m:19-40 -- flip u and x in
         * time, swap init/final, run, flip all 11 outputs back.  Q_w and R_v are
         * passed through UN-flipped, as the reference does (:27). */
        double *uf = (double *)malloc(sizeof(double) * (size_t)nn * T);
        double *xf = (double *)malloc(sizeof(double) * (size_t)T);
        for (int k = 0; k < T; k++) {
            xf[k] = x[T - 1 - k];
            for (int kk = 0; kk < nn; kk++) uf[kk + (size_t)nn * k] = u[kk + (size_t)nn * (T - 1 - k)];
        }
        rc = ekf_core(model, T, uf, xf, prm, /*s_init_flipped=*/s_final, /*Ps_init_flipped=*/Ps_final,
                      /*s_final_flipped=*/s_init, /*Ps_final_flipped=*/Ps_init, v_bar, Q_w, q_len, R_v,
                      r_len, beta, gamma, L, order, u_opt, u_opt_smooth, SM, SP, SS, PM, PP, PS,
                      K_GAIN, innovations, rho, pinv_rank);
        if (rc == ORC_OK) {
            flip_cols(u_opt, nn, T); flip_cols(u_opt_smooth, nn, T);
            flip_cols(S_MINUS, m, T); flip_cols(S_PLUS, m, T); flip_cols(S_SMOOTH, m, T);
            flip_cols(P_MINUS, mm, T); flip_cols(P_PLUS, mm, T); flip_cols(P_SMOOTH, mm, T);
            flip_cols(K_GAIN, m, T); flip_cols(innovations, 1, T);
            /* rho is NOT reversed.  GenericExtendedKalmanFilter.m:233 squeezes rho (1 x 1 x T) to a T x 1 column, and
             * SIAlphaModelBackwardEKF.m:40 / ...BackwardEKFOptControlled.m:40 then index it `rho_flipped(:, :, end:-1:1)`:
             * on a T x 1 array the third dimension has size 1, so `end` is 1 there and the expression returns the column
             * as it stands -- rho(k) of the wrapper is the monitor value of FILTER step k, i.e. of caller day T+1-k. */
            if (pinv_rank)
                for (int k = 0; k < T / 2; k++) { int t = pinv_rank[k]; pinv_rank[k] = pinv_rank[T - 1 - k]; pinv_rank[T - 1 - k] = t; }
        }
        free(uf); free(xf);
    }
    free(sm_); free(sp_); free(pm_); free(pp_); free(ss_); free(ps_);
    return rc;
}

/* ---------- forward simulators ---------- */
void orc_sialpha_controlled(const double *u, int n_npi, double s0, double i0, double alpha0,
                            const double *u_max, double alpha_min, double alpha_max, double gamma,
                            const double *a, double b, double beta, double s_noise_std,
                            double i_noise_std, double alpha_noise_std, int K, double dt,
                            const double *z, double *s, double *i, double *alpha)
{
    double sp = s0, ip = i0, ap = alpha0; /* SIalpha_Controlled.m:19-21 */
    for (int t = 0; t < K; t++) {
        double z1 = z ? z[3 * t + 0] : 0.0, z2 = z ? z[3 * t + 1] : 0.0, z3 = z ? z[3 * t + 2] : 0.0;
        double dot = 0.0;
        for (int kk = 0; kk < n_npi; kk++) {
            double du = u_max[kk] - u[kk + (size_t)n_npi * t]; /* (gamma*a')*(u_max-u) */
            dot = (kk == 0) ? (gamma * a[kk]) * du : fma(gamma * a[kk], du, dot);
        }
        double sn = fmax(0.0, fmin(1.0, sp - dt * (ap * sp * ip + z1 * s_noise_std)));                 /* :25 */
        double in = fmax(0.0, fmin(1.0, ip + dt * (ap * sp * ip - beta * ip + z2 * i_noise_std)));     /* :26 */
        double an = fmax(alpha_min, fmin(alpha_max, ap + dt * (-gamma * ap + gamma * b + dot + z3 * alpha_noise_std))); /* :27 */
        s[t] = sn; i[t] = in; alpha[t] = an; /* initial sample dropped :30-32 */
        sp = sn; ip = in; ap = an;
    }
}

void orc_si_controlled(const double *alpha, double beta, double s0, double i0, int K, double dt,
                       double *s, double *i)
{
    s[0] = s0; i[0] = i0; /* SI_Controlled.m:15-16 */
    for (int t = 0; t < K - 1; t++) {
        s[t + 1] = fmax(0.0, fmin(1.0, s[t] - dt * alpha[t] * s[t] * i[t]));
        i[t + 1] = fmax(0.0, fmin(1.0, i[t] + dt * (alpha[t] * s[t] * i[t] - beta * i[t])));
    }
}

/* testScripts/testSIR01.m:15-36 -- the 3-compartment SIR with return flow r -> s (BASELINE config 1), forward Euler,
 * no clamps; each line evaluated as written: (rhs) * dt + state. */
void orc_sir(double alpha, double beta, double gamma, double s0, double i0, double r0, int K, double dt, double *s, double *i,
             double *r)
{
    s[0] = s0; i[0] = i0; r[0] = r0; /* :28-30 */
    for (int t = 0; t < K - 1; t++) {
        s[t + 1] = (-alpha * s[t] * i[t] + gamma * r[t]) * dt + s[t]; /* :33 */
        i[t + 1] = (alpha * s[t] * i[t] - beta * i[t]) * dt + i[t];   /* :34 */
        r[t + 1] = (beta * i[t] - gamma * r[t]) * dt + r[t];          /* :35 */
    }
}

void orc_seirp(const double *alpha_e, const double *alpha_i, const double *kappa, const double *rho,
               const double *beta, const double *mu, const double *gamma, double s0, double e0,
               double i0, double r0, double p0, int K, double dt, double *s, double *e, double *i,
               double *r, double *p)
{
    s[0] = s0; e[0] = e0; i[0] = i0; r[0] = r0; p[0] = p0; /* SEIRP.m:20-24 */
    for (int t = 0; t < K - 1; t++) { /* :26-32 */
        s[t + 1] = (-alpha_e[t] * s[t] * e[t] - alpha_i[t] * s[t] * i[t] + gamma[t] * r[t]) * dt + s[t];
        e[t + 1] = (alpha_e[t] * s[t] * e[t] + alpha_i[t] * s[t] * i[t] - kappa[t] * e[t] - rho[t] * e[t]) * dt + e[t];
        i[t + 1] = (kappa[t] * e[t] - beta[t] * i[t] - mu[t] * i[t]) * dt + i[t];
        r[t + 1] = (beta[t] * i[t] + rho[t] * e[t] - gamma[t] * r[t]) * dt + r[t];
        p[t + 1] = (mu[t] * i[t]) * dt + p[t];
    }
}

void orc_seirp_saturated(const double *alpha_e, const double *alpha_i, const double *kappa,
                         const double *rho, const double *gamma, double s0, double e0, double i0,
                         double r0, double p0, int K, double dt, double beta_0, double beta_s,
                         double mu_0, double mu_s, double sigma, double i_0, double *s, double *e,
                         double *i, double *r, double *p)
{
    s[0] = s0; e[0] = e0; i[0] = i0; r[0] = r0; p[0] = p0;
    for (int t = 0; t < K - 1; t++) { /* SEIRPSaturatedResource.m:26-36 */
        double h = (epi_tanh((i[t] - i_0) / sigma) + 1.0) / 2.0;
        double beta = (beta_s - beta_0) * h + beta_0;
        double mu = (mu_s - mu_0) * h + mu_0;
        s[t + 1] = (-alpha_e[t] * s[t] * e[t] - alpha_i[t] * s[t] * i[t] + gamma[t] * r[t]) * dt + s[t];
        e[t + 1] = (alpha_e[t] * s[t] * e[t] + alpha_i[t] * s[t] * i[t] - kappa[t] * e[t] - rho[t] * e[t]) * dt + e[t];
        i[t + 1] = (kappa[t] * e[t] - beta * i[t] - mu * i[t]) * dt + i[t];
        r[t + 1] = (beta * i[t] + rho[t] * e[t] - gamma[t] * r[t]) * dt + r[t];
        p[t + 1] = (mu * i[t]) * dt + p[t];
    }
}

void orc_npi_cost(const double *newcases, const double *inputs, const double *weights, int n_npi,
                  int T, double *J0, double *J1)
{
    double a0 = 0.0; /* NPICost.m:6  mean(newcases) */
    for (int t = 0; t < T; t++) a0 = (t == 0) ? newcases[0] : a0 + newcases[t];
    *J0 = a0 / (double)T;
    double a1 = 0.0; /* :9-10  mean(weights(:).*inputs(:)) in column-major order */
    for (size_t e = 0; e < (size_t)n_npi * (size_t)T; e++) {
        double term = weights[e] * inputs[e];
        a1 = (e == 0) ? term : a1 + term;
    }
    *J1 = a1 / (double)((size_t)n_npi * (size_t)T);
}

/* ---------- batched SoA driver ---------- */
int orc_ekf_run_batch(const orc_batch *bt, int n_threads)
{
    const int m = orc_model_dim(bt->model);
    if (m < 0) return ORC_ERR_BAD_ARG;
    const int mm = m * m, T = bt->T, B = bt->B, Sx = bt->Sx, Su = bt->Su, nn = bt->n_npi;
    int rc_all = ORC_OK;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        double *u = (double *)malloc(sizeof(double) * (size_t)nn * T);
        double *x = (double *)malloc(sizeof(double) * (size_t)T);
        double *Rv = (double *)malloc(sizeof(double) * (size_t)T);
        double *uo = (double *)malloc(sizeof(double) * (size_t)nn * T);
        double *uos = (double *)malloc(sizeof(double) * (size_t)nn * T);
        double *SM = (double *)malloc(sizeof(double) * (size_t)m * T);
        double *SP = (double *)malloc(sizeof(double) * (size_t)m * T);
        double *SS = (double *)malloc(sizeof(double) * (size_t)m * T);
        double *PM = (double *)malloc(sizeof(double) * (size_t)mm * T);
        double *PP = (double *)malloc(sizeof(double) * (size_t)mm * T);
        double *PS = (double *)malloc(sizeof(double) * (size_t)mm * T);
        double *KG = (double *)malloc(sizeof(double) * (size_t)m * T);
        double *inn = (double *)malloc(sizeof(double) * (size_t)T);
        double *rh = (double *)malloc(sizeof(double) * (size_t)T);
        int *rk = (int *)malloc(sizeof(int) * (size_t)T);
        double *Qt = bt->q_mode ? (double *)malloc(sizeof(double) * (size_t)mm * T) : NULL;
#pragma omp for schedule(dynamic, 4)
        for (int c = 0; c < B; c++) {
            const int sx = bt->x_series_of_chain ? bt->x_series_of_chain[c] : c;
            const int su = bt->u_series_of_chain ? bt->u_series_of_chain[c] : c;
            orc_params p;
            const double *pr = bt->prm;
#define PRM(f) pr[(size_t)(f) * B + c]
            p.dt = PRM(EPI_PRM_DT); p.beta = PRM(EPI_PRM_BETA); p.gamma = PRM(EPI_PRM_GAMMA);
            p.sigma = PRM(EPI_PRM_SIGMA); p.b = PRM(EPI_PRM_B); p.epsilon = PRM(EPI_PRM_EPSILON);
            p.s_min = PRM(EPI_PRM_S_MIN); p.i_min = PRM(EPI_PRM_I_MIN);
            p.alpha_min = PRM(EPI_PRM_ALPHA_MIN); p.alpha_max = PRM(EPI_PRM_ALPHA_MAX);
            for (int kk = 0; kk < ORC_MAX_NPI; kk++) {
                p.a[kk] = PRM(EPI_PRM_A + kk); p.u_min[kk] = PRM(EPI_PRM_U_MIN + kk);
                p.u_max[kk] = PRM(EPI_PRM_U_MAX + kk); p.w_eff[kk] = PRM(EPI_PRM_W_EFF + kk);
            }
            double v_bar = PRM(EPI_PRM_V_BAR), beta_ekf = PRM(EPI_PRM_BETA_EKF), gamma_ekf = PRM(EPI_PRM_GAMMA_EKF);
#undef PRM
            p.n_npi = nn; p.obs_type = bt->obs_type;
            for (int k = 0; k < T; k++) {
                x[k] = bt->x[(size_t)k * Sx + sx];
                for (int kk = 0; kk < nn; kk++) u[kk + (size_t)nn * k] = bt->u[((size_t)k * nn + kk) * Su + su];
                if (bt->r_mode == 1) Rv[k] = bt->R_series[(size_t)k * Sx + sx];
            }
            if (bt->r_mode == 0) Rv[0] = bt->R_scalar[c];
            double si[MM], sf[MM], Pi[MM * MM], Pf[MM * MM], Q[MM * MM];
            for (int i = 0; i < m; i++) { si[i] = bt->s_init[(size_t)i * B + c]; sf[i] = bt->s_final[(size_t)i * B + c]; }
            for (int i = 0; i < mm; i++) {
                Pi[i] = bt->Ps_init[(size_t)i * B + c]; Pf[i] = bt->Ps_final[(size_t)i * B + c];
                Q[i] = bt->Q[(size_t)i * B + c];
            }
            if (bt->q_mode)
                for (int k = 0; k < T; k++)
                    for (int i = 0; i < mm; i++) Qt[i + (size_t)mm * k] = bt->Q[((size_t)k * mm + i) * B + c];
            int rc = orc_ekf_run(bt->model, T, u, x, &p, si, Pi, sf, Pf, v_bar, bt->q_mode ? Qt : Q, bt->q_mode ? T : 1, Rv,
                                 bt->r_mode == 1 ? T : 1, beta_ekf, gamma_ekf, bt->L, bt->order,
                                 uo, uos, SM, SP, SS, PM, PP, PS, KG, inn, rh, rk);
            if (rc != ORC_OK) {
#pragma omp critical
                rc_all = rc;
                continue;
            }
            int has_uos = (bt->model <= ORC_MODEL_SIA6_BWD);
            for (int k = 0; k < T; k++) {
                for (int kk = 0; kk < nn; kk++) {
                    if (bt->u_opt) bt->u_opt[((size_t)k * nn + kk) * B + c] = uo[kk + (size_t)nn * k];
                    if (bt->u_opt_smooth && has_uos) bt->u_opt_smooth[((size_t)k * nn + kk) * B + c] = uos[kk + (size_t)nn * k];
                }
                for (int i = 0; i < m; i++) {
                    if (bt->S_MINUS) bt->S_MINUS[((size_t)k * m + i) * B + c] = SM[i + (size_t)m * k];
                    if (bt->S_PLUS) bt->S_PLUS[((size_t)k * m + i) * B + c] = SP[i + (size_t)m * k];
                    if (bt->S_SMOOTH) bt->S_SMOOTH[((size_t)k * m + i) * B + c] = SS[i + (size_t)m * k];
                    if (bt->K_GAIN) bt->K_GAIN[((size_t)k * m + i) * B + c] = KG[i + (size_t)m * k];
                }
                for (int i = 0; i < mm; i++) {
                    if (bt->P_MINUS) bt->P_MINUS[((size_t)k * mm + i) * B + c] = PM[i + (size_t)mm * k];
                    if (bt->P_PLUS) bt->P_PLUS[((size_t)k * mm + i) * B + c] = PP[i + (size_t)mm * k];
                    if (bt->P_SMOOTH) bt->P_SMOOTH[((size_t)k * mm + i) * B + c] = PS[i + (size_t)mm * k];
                }
                if (bt->innovations) bt->innovations[(size_t)k * B + c] = inn[k];
                if (bt->rho) bt->rho[(size_t)k * B + c] = rh[k];
                if (bt->pinv_rank) bt->pinv_rank[(size_t)k * B + c] = rk[k];
            }
        }
        free(u); free(x); free(Rv); free(uo); free(uos); free(SM); free(SP); free(SS);
        free(PM); free(PP); free(PS); free(KG); free(inn); free(rh); free(rk); free(Qt);
    }
    return rc_all;
}

/* ------------------------------------------------------------------------------------------------
 * Scenario generation / selection around the sweep
 * ---------------------------------------------------------------------------------------------- */
/* Philox4x32-10, Salmon et al. SC'11 (Random123): the generator the HIP library draws NPI levels from. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Tools/TrainPredictPrescribeNPI.m:499-511: the random plan of (region, scenario) on K forecast days,
 * u n_npi x K column-major.  scenario is 0-based here; `scenario < runs/2` is on the 1-based index (:502). */
void orc_random_npi_plan(uint32_t seed_lo, uint32_t seed_hi, int region, int scenario, int n_scen, int n_npi, int K,
                         const double *npi_mins, const double *npi_maxes, double *u)
{
    const uint32_t key[2] = {seed_lo, seed_hi};
    const int constant_plan = 2 * (scenario + 1) < n_scen;
    for (int t = 0; t < K; t++) {
        for (int k = 0; k < n_npi; k++) {
            if (constant_plan && t > 0) { u[k + (size_t)n_npi * t] = u[k]; continue; }
            const uint32_t ctr[4] = {(uint32_t)region, (uint32_t)scenario, (uint32_t)(k / 4),
                                     constant_plan ? 0u : (uint32_t)t + 1u};
            uint32_t x[4];
            orc_philox4x32_10(ctr, key, x);
            const double w = npi_maxes[k] - npi_mins[k];
            const uint32_t span = (w >= 0.0 && w < 4294967295.0) ? (uint32_t)w + 1u : 1u;
            /* randi([lo, hi]): lo + floor(U * span), U = x / 2^32 */
            u[k + (size_t)n_npi * t] = npi_mins[k] + (double)(uint32_t)(((uint64_t)x[k % 4] * span) >> 32);
        }
    }
}

/* Tools/TrainPredictPrescribeNPI.m:624-633 for one region: is_on_pareto_front and I_opt (0-based). */
void orc_pareto_front(int P, const double *J0, const double *J1, int *on_front, int *i_opt)
{
    double m0 = NAN, m1 = NAN; /* max() ignores NaN */
    for (int q = 0; q < P; q++) { m0 = fmax(m0, J0[q]); m1 = fmax(m1, J1[q]); }
    double best = NAN;
    int bi = 0; /* min of an all-NaN vector returns index 1 */
    for (int ii = 0; ii < P; ii++) {
        int cnt = 0;
        for (int o = 0; o < P; o++) cnt += (J0[o] < J0[ii]) && (J1[o] < J1[ii]);
        if (on_front) on_front[ii] = (cnt == 0);
        const double na = J0[ii] / m0, nb = J1[ii] / m1;
        const double sc = na * na + nb * nb;
        if (!isnan(sc) && (isnan(best) || sc < best)) { best = sc; bi = ii; }
    }
    if (i_opt) *i_opt = bi;
}

/* ------------------------------------------------------------------------------------------------
 * Tools/Rt_ExpFitEKF.m:1-227 -- 2-state exponential-fit EKF/EKS with second-order (Hessian) terms
 * (SURVEY.md 8(f3)).  exp / tanh are epi_exp / epi_tanh above (the fixed operation order the HIP kernels use
 * too), so parity for this function is bit for bit like the other filters.
 * ---------------------------------------------------------------------------------------------- */
static double trace2(const double *M) { return M[0] + M[3]; }

/* [fs, Cs] = the trace terms of StateHessianTerms (:158-199) for one matrix pair list {F1, F2} */
static void hessian_terms2(const double *Pk, const double *F1, const double *F2, double *f, double *Cm)
{
    const double *F[2] = {F1, F2};
    double T1[4], T2[4], T3[4];
    for (int ii = 0; ii < 2; ii++) {
        mat_mul(2, Pk, F[ii], T1);
        f[ii] = trace2(T1) / 2;                         /* :183 */
        for (int jj = 0; jj < 2; jj++) {
            mat_mul(2, T1, Pk, T2);                     /* Pk*Fs{ii}*Pk*Fs{jj}, left to right */
            mat_mul(2, T2, F[jj], T3);
            Cm[IX(ii, jj, 2)] = trace2(T3) / 2;         /* :185 */
        }
    }
}

int orc_rt_expfit_ekf(int T, const double *x, const double *s_init, const double *params, const double *w_bar,
                      double v_bar, const double *Ps_init, const double *Q_w, double R_v, double beta, double gamma,
                      int L, int order, double *S_MINUS, double *S_PLUS, double *P_MINUS, double *P_PLUS,
                      double *K_GAIN, double *S_SMOOTH, double *P_SMOOTH, double *innovations, double *rho)
{
    if (order != 1 && order != 2) return ORC_ERR_UNDEFINED_ORDER;   /* :46,77 */
    if (T < 1 || L < 1) return ORC_ERR_BAD_ARG;
    const double ts = params[0], alpha = params[1], sigma = params[2];
    double *winMean = (double *)calloc((size_t)L, sizeof(double));
    double *winCov = (double *)calloc((size_t)L, sizeof(double));
    double *winCovN = (double *)calloc((size_t)L, sizeof(double));
    double sm[2] = {s_init[0], s_init[1]}, Pm[4], R = R_v;      /* :29-32 */
    memcpy(Pm, Ps_init, sizeof Pm);
    const double Cj[2] = {1.0, 0.0}, Dj = 1.0;                  /* ObsJacobian :151-154 */
    for (int k = 0; k < T; k++) {
        memcpy(S_MINUS + 2 * (size_t)k, sm, sizeof sm);         /* :37-38 */
        memcpy(P_MINUS + 4 * (size_t)k, Pm, sizeof Pm);
        /* ObsHessianTerms :202-227: Gs = Gv = {0} => every trace is 0 for either order */
        const double gs = 0.0, Gsp = 0.0, gv = 0.0, Gvp = 0.0;
        const double xk_minus = ((sm[0] + v_bar) + gs) + gv;    /* :52 */
        double innov, K[2], sp[2], Pp[4];
        const int valid = !isnan(x[k]);
        if (valid) {                                            /* :55-59 */
            innov = x[k] - xk_minus;
            double PCt[2], CP[2];
            for (int i = 0; i < 2; i++) PCt[i] = fma(Pm[IX(i, 1, 2)], Cj[1], Pm[IX(i, 0, 2)] * Cj[0]);
            for (int j = 0; j < 2; j++) CP[j] = fma(Cj[1], Pm[IX(1, j, 2)], Cj[0] * Pm[IX(0, j, 2)]);
            const double CPCt = fma(CP[1], Cj[1], CP[0] * Cj[0]);
            const double den = ((CPCt + gamma * ((Dj * R) * Dj)) + Gsp) + Gvp;
            for (int i = 0; i < 2; i++) K[i] = PCt[i] / den;
            double IKC[4], T1[4];
            for (int j = 0; j < 2; j++)
                for (int i = 0; i < 2; i++) IKC[IX(i, j, 2)] = ((i == j) ? 1.0 : 0.0) - K[i] * Cj[j];
            mat_mul(2, IKC, Pm, T1);
            for (int e = 0; e < 4; e++) Pp[e] = T1[e] / gamma;
            for (int i = 0; i < 2; i++) sp[i] = sm[i] + K[i] * innov;
        } else {                                                /* :60-65 */
            innov = 0.0; K[0] = K[1] = 0.0;
            memcpy(Pp, Pm, sizeof Pp); memcpy(sp, sm, sizeof sp);
        }
        /* NlinStateUpdate :133-140, StateJacobians :148-160 */
        const double E = epi_exp(ts * sp[1]);
        const double tnh = epi_tanh((alpha * sp[1] + w_bar[1]) / sigma);
        const double omt = 1 - tnh * tnh;
        double fs[2] = {0, 0}, fw[2] = {0, 0}, Fsp[4] = {0, 0, 0, 0}, Fwp[4] = {0, 0, 0, 0};
        if (order == 2) {                                       /* StateHessianTerms :163-199 */
            double Fs1[4] = {0, 0, 0, 0}, Fs2[4] = {0, 0, 0, 0}, Fw1[4] = {0, 0, 0, 0}, Fw2[4] = {0, 0, 0, 0};
            Fs1[IX(0, 1, 2)] = ts * E; Fs1[IX(1, 0, 2)] = Fs1[IX(0, 1, 2)];
            Fs1[IX(1, 1, 2)] = ((ts * ts) * sp[0]) * E;
            Fs2[IX(1, 1, 2)] = (((-2 * (alpha * alpha)) / sigma) * tnh) * omt;
            Fw2[IX(1, 1, 2)] = ((-2 / sigma) * tnh) * omt;
            hessian_terms2(Pp, Fs1, Fs2, fs, Fsp);
            hessian_terms2(Q_w, Fw1, Fw2, fw, Fwp);
        }
        sm[0] = ((sp[0] * E + w_bar[0]) + fs[0]) + fw[0];       /* :81 */
        sm[1] = ((sigma * tnh) + fs[1]) + fw[1];
        {
            double A[4], Bm[4] = {1, 0, 0, omt}, T1[4], T2[4], T3[4];
            A[IX(0, 0, 2)] = E; A[IX(0, 1, 2)] = (ts * sp[0]) * E; A[IX(1, 0, 2)] = 0; A[IX(1, 1, 2)] = alpha * omt;
            mat_mul(2, A, Pp, T1); mat_mul_bt(2, T1, A, T2);
            mat_mul(2, Bm, Q_w, T1); mat_mul_bt(2, T1, Bm, T3);
            for (int e = 0; e < 4; e++) Pm[e] = ((T2[e] + T3[e]) + Fsp[e]) + Fwp[e];   /* :83 */
        }
        memcpy(S_PLUS + 2 * (size_t)k, sp, sizeof sp);          /* :86-88 */
        memcpy(P_PLUS + 4 * (size_t)k, Pp, sizeof Pp);
        if (K_GAIN) { K_GAIN[2 * (size_t)k] = K[0]; K_GAIN[2 * (size_t)k + 1] = K[1]; }
        if (innovations) innovations[k] = innov;
        /* :91-101 */
        const int cnt = (k + 1 < L) ? (k + 1) : L;
        memmove(winMean + 1, winMean, sizeof(double) * (size_t)(L - 1)); winMean[0] = innov;
        double sum = winMean[0];
        for (int j = 1; j < L; j++) sum = sum + winMean[j];
        const double mu = sum / cnt;
        const double cc = (innov - mu) * (innov - mu);
        memmove(winCov + 1, winCov, sizeof(double) * (size_t)(L - 1)); winCov[0] = cc;
        memmove(winCovN + 1, winCovN, sizeof(double) * (size_t)(L - 1)); winCovN[0] = cc / R;
        double sumN = winCovN[0];
        for (int j = 1; j < L; j++) sumN = sumN + winCovN[j];
        if (rho) rho[k] = sumN / cnt;
        if (beta != 1.0 && valid) {
            double sumC = winCov[0];
            for (int j = 1; j < L; j++) sumC = sumC + winCov[j];
            R = beta * R + (1 - beta) * sumC / cnt;
        }
    }
    free(winMean); free(winCov); free(winCovN);
    if (S_SMOOTH && P_SMOOTH) {                                  /* :105-116 */
        double Ss[2], Ps[4];
        memcpy(Ss, S_PLUS + 2 * (size_t)(T - 1), sizeof Ss); memcpy(Ps, P_PLUS + 4 * (size_t)(T - 1), sizeof Ps);
        memcpy(S_SMOOTH + 2 * (size_t)(T - 1), Ss, sizeof Ss); memcpy(P_SMOOTH + 4 * (size_t)(T - 1), Ps, sizeof Ps);
        for (int k = T - 2; k >= 0; k--) {
            const double *sp = S_PLUS + 2 * (size_t)k, *Pp = P_PLUS + 4 * (size_t)k;
            const double *Sm1 = S_MINUS + 2 * (size_t)(k + 1), *Pm1 = P_MINUS + 4 * (size_t)(k + 1);
            const double E = epi_exp(ts * sp[1]);
            const double tnh = epi_tanh((alpha * sp[1] + w_bar[1]) / sigma);
            double A[4], T1[4], J[4], D[4], T2[4];
            A[IX(0, 0, 2)] = E; A[IX(0, 1, 2)] = (ts * sp[0]) * E; A[IX(1, 0, 2)] = 0; A[IX(1, 1, 2)] = alpha * (1 - tnh * tnh);
            mat_mul_bt(2, Pp, A, T1);
            orc_mrdivide(2, T1, Pm1, J);                         /* :112 */
            double dv[2] = {Ss[0] - Sm1[0], Ss[1] - Sm1[1]};
            for (int i = 0; i < 2; i++) Ss[i] = sp[i] + fma(J[IX(i, 1, 2)], dv[1], J[IX(i, 0, 2)] * dv[0]);
            for (int e = 0; e < 4; e++) D[e] = Pm1[e] - Ps[e];
            mat_mul(2, J, D, T1); mat_mul_bt(2, T1, J, T2);
            for (int e = 0; e < 4; e++) Ps[e] = Pp[e] - T2[e];
            memcpy(S_SMOOTH + 2 * (size_t)k, Ss, sizeof Ss); memcpy(P_SMOOTH + 4 * (size_t)k, Ps, sizeof Ps);
        }
    }
    return ORC_OK;
}

/* batched SoA driver with the HIP library's layout: x [T][Sx], rp [EPI_RT_PRM_COUNT][B], outputs [T][rows][B] */
int orc_rt_expfit_batch(int B, int T, int Sx, const int *x_series, const double *x, const double *rp, int L, int order,
                        double *S_MINUS, double *S_PLUS, double *P_MINUS, double *P_PLUS, double *K_GAIN,
                        double *S_SMOOTH, double *P_SMOOTH, double *innovations, double *rho, int n_threads)
{
    int rc_all = ORC_OK;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        double *xs = (double *)malloc(sizeof(double) * (size_t)T);
        double *buf = (double *)malloc(sizeof(double) * (size_t)T * 22);
        double *SM = buf, *SP = SM + 2 * (size_t)T, *PM = SP + 2 * (size_t)T, *PP = PM + 4 * (size_t)T, *KG = PP + 4 * (size_t)T;
        double *SS = KG + 2 * (size_t)T, *PS = SS + 2 * (size_t)T, *inn = PS + 4 * (size_t)T, *rh = inn + T;
#pragma omp for schedule(dynamic, 16)
        for (int c = 0; c < B; c++) {
            const int sx = x_series ? x_series[c] : c;
            for (int k = 0; k < T; k++) xs[k] = x[(size_t)k * Sx + sx];
#define RP(f) rp[(size_t)(f) * B + c]
            const double params[3] = {RP(0), RP(1), RP(2)}, w_bar[2] = {RP(3), RP(4)}, s_init[2] = {RP(9), RP(10)};
            const double Pi[4] = {RP(11), RP(12), RP(13), RP(14)}, Q[4] = {RP(15), RP(16), RP(17), RP(18)};
            int rc = orc_rt_expfit_ekf(T, xs, s_init, params, w_bar, RP(5), Pi, Q, RP(6), RP(7), RP(8), L, order,
                                       SM, SP, PM, PP, KG, SS, PS, inn, rh);
#undef RP
            if (rc != ORC_OK) {
#pragma omp critical
                rc_all = rc;
                continue;
            }
            for (int k = 0; k < T; k++) {
                for (int i = 0; i < 2; i++) {
                    if (S_MINUS) S_MINUS[((size_t)k * 2 + i) * B + c] = SM[i + 2 * (size_t)k];
                    if (S_PLUS) S_PLUS[((size_t)k * 2 + i) * B + c] = SP[i + 2 * (size_t)k];
                    if (K_GAIN) K_GAIN[((size_t)k * 2 + i) * B + c] = KG[i + 2 * (size_t)k];
                    if (S_SMOOTH) S_SMOOTH[((size_t)k * 2 + i) * B + c] = SS[i + 2 * (size_t)k];
                }
                for (int e = 0; e < 4; e++) {
                    if (P_MINUS) P_MINUS[((size_t)k * 4 + e) * B + c] = PM[e + 4 * (size_t)k];
                    if (P_PLUS) P_PLUS[((size_t)k * 4 + e) * B + c] = PP[e + 4 * (size_t)k];
                    if (P_SMOOTH) P_SMOOTH[((size_t)k * 4 + e) * B + c] = PS[e + 4 * (size_t)k];
                }
                if (innovations) innovations[(size_t)k * B + c] = inn[k];
                if (rho) rho[(size_t)k * B + c] = rh[k];
            }
        }
        free(xs); free(buf);
    }
    return rc_all;
}

/* ------------------------------------------------------------------------------------------------
 * Per-region preprocessing feeding the filters: Tools/TrainPredictPrescribeNPI.m:152-198,240
 * (the same block again at :97-112 for the "ENTIRE" span and in ForecastQualityAssessment.m).
 *
 * `filter` is a MATLAB built-in and `filtfilt` belongs to the Signal Processing Toolbox -- neither is in the
 * reference checkout.  Restated from their published definitions:
 *   filter(b, a, x): coefficients normalised by a(1); direct form II transposed,
 *       y(n) = b1 x(n) + z1(n-1),  z_i(n) = b_{i+1} x(n) + z_{i+1}(n-1),  z_{nb-1}(n) = b_nb x(n).
 *   filtfilt(b, a, x) (Gustafsson 1996, as in the toolbox and SciPy): nfact = 3 (max(nb, na) - 1) samples of odd
 *       reflection at both ends, zi = steady-state DF-II-T state for a unit step (zi(nb-1) = b(nb),
 *       zi(i) = b(i+1) + zi(i+1)), forward pass from zi * e(1), time reversal, second pass from zi * y(1), reversal,
 *       padding removed.  Coefficients normalised by a(1) first, like filter does.
 * ---------------------------------------------------------------------------------------------- */
#define PRE_MAX_TAPS 32

/* y = filter(ones(1, W), W, x) */
static void causal_ma(int T, const double *x, int W, double *y)
{
    const double c = 1.0 / (double)W; /* b / a(1) */
    for (int n = 0; n < T; n++) {
        double acc = 0.0;
        for (int k = W - 1; k >= 1; k--) {   /* oldest first: z_{W-1} inwards to z_1 */
            double p = (n - k >= 0) ? c * x[n - k] : 0.0;
            acc = p + acc;
        }
        y[n] = c * x[n] + acc;
    }
}

/* DF-II-T pass of an nb-tap FIR with equal taps c over e[0..n), initial state zi * e[0]; out may alias nothing */
static void fir_df2t_zi(int n, const double *e, int nb, double c, const double *zi, double *out)
{
    double z[PRE_MAX_TAPS];
    for (int i = 0; i < nb - 1; i++) z[i] = zi[i] * e[0];
    for (int t = 0; t < n; t++) {
        const double x = e[t];
        const double y = (nb > 1) ? c * x + z[0] : c * x;
        for (int i = 0; i < nb - 2; i++) z[i] = c * x + z[i + 1];
        if (nb > 1) z[nb - 2] = c * x;
        out[t] = y;
    }
}

/* y = filtfilt(ones(1, W), W, x); returns 0, or -1 when T <= 3 (W - 1) ('Data length must be larger than ...') */
static int zero_phase_ma(int T, const double *x, int W, double *y, double *scratch /* 2 * (T + 6 (W - 1)) */)
{
    const int nfact = (3 * (W - 1) > 1) ? 3 * (W - 1) : 1;
    if (T <= nfact || W > PRE_MAX_TAPS) return -1;
    const double c = 1.0 / (double)W;
    double zi[PRE_MAX_TAPS];
    if (W > 1) {
        zi[W - 2] = c;
        for (int i = W - 3; i >= 0; i--) zi[i] = c + zi[i + 1];
    }
    const int n = T + 2 * nfact;
    double *e = scratch, *f = scratch + n;
    for (int i = 0; i < nfact; i++) e[i] = 2 * x[0] - x[nfact - i];
    for (int i = 0; i < T; i++) e[nfact + i] = x[i];
    for (int i = 0; i < nfact; i++) e[nfact + T + i] = 2 * x[T - 1] - x[T - 2 - i];
    fir_df2t_zi(n, e, W, c, zi, f);
    for (int i = 0; i < n; i++) e[i] = f[n - 1 - i];
    fir_df2t_zi(n, e, W, c, zi, f);
    for (int i = 0; i < T; i++) y[i] = f[n - 1 - (nfact + i)];
    return 0;
}

/* diff([c(1); c]), clamp negatives, fill a missing last day with the last valid one, other NaNs -> 0  (:166-178) */
static void refine_counts(int T, const double *cum, double *out)
{
    int last_valid = -1;
    for (int t = 0; t < T; t++) {
        double d = cum[t] - cum[t > 0 ? t - 1 : 0];
        if (d < 0) d = 0;                       /* NaN < 0 is false */
        out[t] = d;
        if (!isnan(d)) last_valid = t;
    }
    if (isnan(out[T - 1]) && last_valid >= 0) out[T - 1] = out[last_valid];
    for (int t = 0; t < T; t++)
        if (isnan(out[t])) out[t] = 0.0;
}

/* One region.  cases/deaths: cumulative confirmed counts [T] (deaths may be NULL).  Outputs (each may be NULL):
 * new_refined, new_smoothed, zero_lag, x_new (= smoothed / N), x_total (= cumsum(smoothed) / N), R_v, fatality [T];
 * *I0.  Returns 0, ORC_ERR_BAD_ARG for W out of range, T < 2 (:168) or T too short for filtfilt. */
int orc_preprocess_region(int T, const double *cases, const double *deaths, double N_population, int W,
                          double min_cases, int first_num_days, double *new_refined, double *new_smoothed,
                          double *zero_lag, double *x_new, double *x_total, double *R_v, double *fatality, double *I0)
{
    if (W < 1 || W > PRE_MAX_TAPS || T < 2) return ORC_ERR_BAD_ARG;
    const int W2 = (int)floor((double)W / 2 + 0.5);          /* MATLAB round(): halves away from zero */
    double *ref = (double *)malloc(sizeof(double) * (size_t)T * 4), *sm = ref + T, *zl = sm + T, *cs = zl + T;
    double *scratch = (double *)malloc(sizeof(double) * 2 * ((size_t)T + 6 * PRE_MAX_TAPS));
    int rc = ORC_OK;
    refine_counts(T, cases, ref);
    causal_ma(T, ref, W, sm);                                /* :173 */
    if (zero_phase_ma(T, ref, W2 < 1 ? 1 : W2, zl, scratch) != 0) rc = ORC_ERR_BAD_ARG;   /* :174 */
    if (rc == ORC_OK) {
        double run = 0.0;
        for (int t = 0; t < T; t++) {
            run = (t == 0) ? sm[0] : run + sm[t];            /* cumsum :178 */
            cs[t] = run;
            if (new_refined) new_refined[t] = ref[t];
            if (new_smoothed) new_smoothed[t] = sm[t];
            if (zero_lag) zero_lag[t] = zl[t];
            if (x_new) x_new[t] = sm[t] / N_population;      /* :175 */
            if (x_total) x_total[t] = run / N_population;    /* :180 */
            if (R_v) { const double d = (zl[t] - ref[t]) / N_population; R_v[t] = 0.1 * (d * d); }   /* :240 */
        }
        if (I0) {                                            /* :201-202 */
            double s = 0.0; int cnt = 0;
            for (int t = 0; t < T && cnt < first_num_days; t++)
                if (sm[t] > 0) { s = (cnt == 0) ? sm[t] : s + sm[t]; cnt++; }
            const double mean = cnt ? s / cnt : NAN;
            *I0 = fmax(min_cases, mean);
        }
        if (fatality && deaths) {                            /* :183-197 */
            double *dr = ref, *ds = zl;                      /* reuse */
            refine_counts(T, deaths, dr);
            causal_ma(T, dr, W, ds);
            double drun = 0.0;
            for (int t = 0; t < T; t++) {
                drun = (t == 0) ? ds[0] : drun + ds[t];
                double fr = drun / cs[t];
                fatality[t] = isnan(fr) ? 0.0 : fr;
            }
        }
    }
    free(ref); free(scratch);
    return rc;
}

/* Tools/TrainPredictPrescribeNPI.m:142-150: carry the previous day's level over N/A days, leading N/A -> 0.
 * ip: T x n_npi column-major (days down the rows, as read from the table) is NOT assumed here: ip[t * n_npi + j]. */
void orc_npi_fill(int T, int n_npi, const double *ip, double *out)
{
    for (int j = 0; j < n_npi; j++) {
        for (int t = 0; t < T; t++) {
            double v = ip[(size_t)t * n_npi + j];
            if (t > 0 && isnan(v) && !isnan(out[(size_t)(t - 1) * n_npi + j])) v = out[(size_t)(t - 1) * n_npi + j];
            out[(size_t)t * n_npi + j] = v;
        }
        for (int t = 0; t < T; t++)
            if (isnan(out[(size_t)t * n_npi + j])) out[(size_t)t * n_npi + j] = 0.0;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Regression between the EKF rounds: Tools/TrainPredictPrescribeNPI.m:251-276,
 * REGRESSION_TYPE = 'NONNEGATIVELS':  alpha(t) ~ a' (NPI_MAXES - u(t)) + b,  a >= 0.
 *
 * lsqnonneg is a MATLAB built-in that is not in the reference checkout; it implements Lawson & Hanson's
 * active-set NNLS (1974) with tol = 10*eps*norm(C,1)*length(C) and itmax = 3n.  Restated here on the normal
 * equations (G = C'C, h = C'd): the active-set logic is Lawson-Hanson's verbatim, the passive-set least-squares
 * solve C(:,P)\d is a Cholesky factorisation of G(P,P) with diagonal pivoting -- in exact arithmetic the same pivot
 * order and the same basic solution (dependent columns 0) as the pivoted QR behind MATLAB's backslash.  A pivot
 * below 100*n*eps*max(diag) ends the factorisation (rank decision on the squared scale of the Gram matrix).
 * ---------------------------------------------------------------------------------------------- */
#define NN_MAX 12

/* z(P) = argmin ||C(:,P) z - d||, z(~P) = 0, from G and h */
static void nnls_solve_passive(int n, const double *G, const double *h, const int *inP, double *z)
{
    int idx[NN_MAX], k = 0;
    double A[NN_MAX * NN_MAX], b[NN_MAX], y[NN_MAX];
    for (int j = 0; j < n; j++) { z[j] = 0.0; if (inP[j]) idx[k++] = j; }
    if (k == 0) return;
    for (int i = 0; i < k; i++) {
        b[i] = h[idx[i]];
        for (int j = 0; j < k; j++) A[i + NN_MAX * j] = G[idx[i] + n * idx[j]];
    }
    double dmax = 0.0;
    for (int i = 0; i < k; i++) dmax = fmax(dmax, A[i + NN_MAX * i]);
    const double ptol = 100.0 * n * DBL_EPS * dmax;
    int rank = 0;
    for (int i = 0; i < k; i++) {                        /* lower-triangular Cholesky with diagonal pivoting */
        int p = i;
        for (int j = i + 1; j < k; j++)
            if (A[j + NN_MAX * j] > A[p + NN_MAX * p]) p = j;
        if (!(A[p + NN_MAX * p] > ptol)) break;
        if (p != i) {
            for (int c = 0; c < k; c++) { double t = A[i + NN_MAX * c]; A[i + NN_MAX * c] = A[p + NN_MAX * c]; A[p + NN_MAX * c] = t; }
            for (int r = 0; r < k; r++) { double t = A[r + NN_MAX * i]; A[r + NN_MAX * i] = A[r + NN_MAX * p]; A[r + NN_MAX * p] = t; }
            { double t = b[i]; b[i] = b[p]; b[p] = t; }
            { int t = idx[i]; idx[i] = idx[p]; idx[p] = t; }
        }
        const double d = sqrt(A[i + NN_MAX * i]);
        A[i + NN_MAX * i] = d;
        for (int j = i + 1; j < k; j++) A[j + NN_MAX * i] = A[j + NN_MAX * i] / d;
        for (int c = i + 1; c < k; c++)                  /* whole trailing block: stays exactly symmetric for the swaps */
            for (int r = i + 1; r < k; r++) A[r + NN_MAX * c] = fma(-A[r + NN_MAX * i], A[c + NN_MAX * i], A[r + NN_MAX * c]);
        rank = i + 1;
    }
    for (int i = 0; i < rank; i++) {                     /* L y = b */
        double acc = b[i];
        for (int j = 0; j < i; j++) acc = fma(-A[i + NN_MAX * j], y[j], acc);
        y[i] = acc / A[i + NN_MAX * i];
    }
    for (int i = rank - 1; i >= 0; i--) {                /* L' z = y */
        double acc = y[i];
        for (int j = i + 1; j < rank; j++) acc = fma(-A[j + NN_MAX * i], y[j], acc);
        y[i] = acc / A[i + NN_MAX * i];
    }
    for (int i = 0; i < rank; i++) z[idx[i]] = y[i];
}

/* x = lsqnonneg(C, d) given G = C'C (n x n column-major), h = C'd and tol.  Returns the exit flag (1, or 0 when the
 * inner loop hit 3n iterations). */
int orc_nnls_gram(int n, const double *G, const double *h, double tol, double *x)
{
    int inP[NN_MAX] = {0};
    double w[NN_MAX], z[NN_MAX];
    for (int j = 0; j < n; j++) { x[j] = 0.0; w[j] = h[j]; }   /* resid = d - C*0 */
    int iter = 0;
    const int itmax = 3 * n;
    for (;;) {
        int t = -1;
        for (int j = 0; j < n; j++)
            if (!inP[j] && w[j] > tol && (t < 0 || w[j] > w[t])) t = j;   /* any(Z) && any(w(Z) > tol); first max */
        if (t < 0) return 1;
        inP[t] = 1;
        nnls_solve_passive(n, G, h, inP, z);
        for (;;) {
            int any = 0;
            for (int j = 0; j < n; j++) any |= (inP[j] && z[j] <= 0.0);
            if (!any) break;
            if (++iter > itmax) { for (int j = 0; j < n; j++) x[j] = z[j]; return 0; }
            double alpha = INFINITY;
            for (int j = 0; j < n; j++)
                if (inP[j] && z[j] <= 0.0) alpha = fmin(alpha, x[j] / (x[j] - z[j]));
            for (int j = 0; j < n; j++) x[j] = x[j] + alpha * (z[j] - x[j]);
            for (int j = 0; j < n; j++)
                if (inP[j] && fabs(x[j]) < tol) inP[j] = 0;
            nnls_solve_passive(n, G, h, inP, z);
        }
        for (int j = 0; j < n; j++) x[j] = z[j];
        for (int i = 0; i < n; i++) {                     /* w = C'(d - C x) = h - G x */
            double acc = G[i] * x[0];
            for (int j = 1; j < n; j++) acc = fma(G[i + n * j], x[j], acc);
            w[i] = h[i] - acc;
        }
    }
}

/* G, h, tol of lsqnonneg(X, y - shift) for X [D x n] row t at X[t * n + j] */
static void nnls_normal_equations(int D, int n, const double *X, const double *y, double shift, double *G, double *h, double *tol)
{
    for (int i = 0; i < n; i++) {
        double cs = 0.0;
        for (int t = 0; t < D; t++) cs = cs + fabs(X[(size_t)t * n + i]);
        if (i == 0 || cs > *tol) *tol = cs;                /* norm(C, 1) */
        double hv = 0.0;
        for (int t = 0; t < D; t++) hv = fma(X[(size_t)t * n + i], y[t] - shift, hv);
        h[i] = hv;
        for (int j = 0; j <= i; j++) {
            double g = 0.0;
            for (int t = 0; t < D; t++) g = fma(X[(size_t)t * n + i], X[(size_t)t * n + j], g);
            G[i + n * j] = g; G[j + n * i] = g;
        }
    }
    *tol = 10.0 * DBL_EPS * (*tol) * (double)(D > n ? D : n);   /* 10*eps*norm(C,1)*length(C) */
}

/* TrainPredictPrescribeNPI.m:262-276 for one region: X [D][n] (row-major: day, NPI), y [D].  Outputs a [n], *b, *min_err,
 * *iters (passes of the :266 loop that were accepted). */
int orc_nnls_affine_fit(int D, int n, const double *X, const double *y, int max_iters, double *a, double *b,
                        double *min_err, int *iters)
{
    if (n < 1 || n > NN_MAX || D < 1) return ORC_ERR_BAD_ARG;
    double G[NN_MAX * NN_MAX], h[NN_MAX], tol = 0.0, coef[NN_MAX];
    nnls_normal_equations(D, n, X, y, 0.0, G, h, &tol);
    orc_nnls_gram(n, G, h, tol, a);                          /* :263 */
    double bb = 0.0, err = 0.0;
    for (int t = 0; t < D; t++) {
        double xa = X[(size_t)t * n] * a[0];
        for (int j = 1; j < n; j++) xa = fma(X[(size_t)t * n + j], a[j], xa);
        const double r = y[t] - xa;
        err = (t == 0) ? r * r : err + r * r;                /* :265 */
    }
    int acc = 0;
    for (int jj = 0; jj < max_iters; jj++) {                 /* :266-276 */
        nnls_normal_equations(D, n, X, y, bb, G, h, &tol);
        orc_nnls_gram(n, G, h, tol, coef);                   /* lsqnonneg(X, y - reg_coef_b) */
        double s = 0.0;
        for (int t = 0; t < D; t++) {                        /* mean(y - X*reg_coef_a): the CURRENT reg_coef_a */
            double xa = X[(size_t)t * n] * a[0];
            for (int j = 1; j < n; j++) xa = fma(X[(size_t)t * n + j], a[j], xa);
            const double r = y[t] - xa;
            s = (t == 0) ? r : s + r;
        }
        const double c0 = s / D;
        double e = 0.0;
        for (int t = 0; t < D; t++) {
            double xa = X[(size_t)t * n] * a[0];
            for (int j = 1; j < n; j++) xa = fma(X[(size_t)t * n + j], a[j], xa);
            const double r = (y[t] - xa) - c0;
            e = (t == 0) ? r * r : e + r * r;
        }
        if (e < err) { for (int j = 0; j < n; j++) a[j] = coef[j]; bb = c0; err = e; acc++; }
        else break;
    }
    *b = bb; *min_err = err; if (iters) *iters = acc;
    return ORC_OK;
}
