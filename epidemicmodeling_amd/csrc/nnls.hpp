// Non-negative least squares between the EKF rounds:
// Tools/TrainPredictPrescribeNPI.m:251-276, REGRESSION_TYPE = 'NONNEGATIVELS' -- for every region
//     alpha(t) ~ a' (NPI_MAXES - u(t)) + b,   a >= 0
// over the last num_regression_days days: a = lsqnonneg(X, y), then the :266-276 loop that introduces the intercept.
// Included by epiekf.hip.
//
// One lane per region.  lsqnonneg (Lawson & Hanson's active-set NNLS, tol = 10*eps*norm(C,1)*length(C), 3n inner
// iterations at most) runs on the normal equations: G = X'X, h = X'(y - b) are accumulated in one pass over the
// region's D x n block (read coalesced across regions), after which every passive-set solve is a pivoted Cholesky
// of at most 12 x 12 held in LDS (dynamic indexing -- one conflict-free column per lane).  Operation for operation
// the oracle's algorithm, so the results are bit-identical to it.
#pragma once

constexpr int kNnMax = 12;
constexpr int kNnLanes = 32;                                   // lanes per workgroup (LDS: 3 KiB per lane)
constexpr int kNnDoubles = 2 * kNnMax * kNnMax + 8 * kNnMax;   // G, A, h, b, yv, z, x, w, a, coef
constexpr int kNnInts = 2 * kNnMax;                            // inP, idx

struct NnArgs {
    int S, D, n, max_iters;
    const double *X, *y;       // [D][n][S], [D][S]
    double *a, *b, *min_err;   // [n][S], [S], [S]
    int32_t *iters, *flag;     // [S] accepted passes of the :266 loop; lsqnonneg exit flag of the first solve
};

struct NnLds {                 // element e of this lane's arrays: base[e * kNnLanes]
    double *G, *A, *h, *b, *yv, *z, *x, *w, *a, *coef;
    int *inP, *idx;
};
#define NN_AT(p, e) (p)[(e) * kNnLanes]

// z(P) = argmin ||C(:,P) z - d||, z(~P) = 0: Cholesky of G(P,P) with diagonal pivoting, dependent columns dropped
EPI_DEV void nnls_solve_passive(const NnLds &m, int n)
{
    int k = 0;
    for (int j = 0; j < n; j++) { NN_AT(m.z, j) = 0.0; if (NN_AT(m.inP, j)) { NN_AT(m.idx, k) = j; k++; } }
    if (k == 0) return;
    for (int i = 0; i < k; i++) {
        NN_AT(m.b, i) = NN_AT(m.h, NN_AT(m.idx, i));
        for (int j = 0; j < k; j++) NN_AT(m.A, i + kNnMax * j) = NN_AT(m.G, NN_AT(m.idx, i) + n * NN_AT(m.idx, j));
    }
    double dmax = 0.0;
    for (int i = 0; i < k; i++) dmax = fmax(dmax, NN_AT(m.A, i + kNnMax * i));
    const double ptol = 100.0 * n * kEps * dmax;
    int rank = 0;
    for (int i = 0; i < k; i++) {
        int p = i;
        for (int j = i + 1; j < k; j++)
            if (NN_AT(m.A, j + kNnMax * j) > NN_AT(m.A, p + kNnMax * p)) p = j;
        if (!(NN_AT(m.A, p + kNnMax * p) > ptol)) break;
        if (p != i) {
            for (int c = 0; c < k; c++) { const double t = NN_AT(m.A, i + kNnMax * c); NN_AT(m.A, i + kNnMax * c) = NN_AT(m.A, p + kNnMax * c); NN_AT(m.A, p + kNnMax * c) = t; }
            for (int r = 0; r < k; r++) { const double t = NN_AT(m.A, r + kNnMax * i); NN_AT(m.A, r + kNnMax * i) = NN_AT(m.A, r + kNnMax * p); NN_AT(m.A, r + kNnMax * p) = t; }
            { const double t = NN_AT(m.b, i); NN_AT(m.b, i) = NN_AT(m.b, p); NN_AT(m.b, p) = t; }
            { const int t = NN_AT(m.idx, i); NN_AT(m.idx, i) = NN_AT(m.idx, p); NN_AT(m.idx, p) = t; }
        }
        const double d = sqrt(NN_AT(m.A, i + kNnMax * i));
        NN_AT(m.A, i + kNnMax * i) = d;
        for (int j = i + 1; j < k; j++) NN_AT(m.A, j + kNnMax * i) = NN_AT(m.A, j + kNnMax * i) / d;
        for (int c = i + 1; c < k; c++)                  // whole trailing block: stays exactly symmetric for the swaps
            for (int r = i + 1; r < k; r++)
                NN_AT(m.A, r + kNnMax * c) = fma(-NN_AT(m.A, r + kNnMax * i), NN_AT(m.A, c + kNnMax * i), NN_AT(m.A, r + kNnMax * c));
        rank = i + 1;
    }
    for (int i = 0; i < rank; i++) {                     // L y = b
        double acc = NN_AT(m.b, i);
        for (int j = 0; j < i; j++) acc = fma(-NN_AT(m.A, i + kNnMax * j), NN_AT(m.yv, j), acc);
        NN_AT(m.yv, i) = acc / NN_AT(m.A, i + kNnMax * i);
    }
    for (int i = rank - 1; i >= 0; i--) {                // L' z = y
        double acc = NN_AT(m.yv, i);
        for (int j = i + 1; j < rank; j++) acc = fma(-NN_AT(m.A, j + kNnMax * i), NN_AT(m.yv, j), acc);
        NN_AT(m.yv, i) = acc / NN_AT(m.A, i + kNnMax * i);
    }
    for (int i = 0; i < rank; i++) NN_AT(m.z, NN_AT(m.idx, i)) = NN_AT(m.yv, i);
}

// x = lsqnonneg(C, d) from G, h (Lawson & Hanson 1974 as in lsqnonneg.m); result in m.x; returns the exit flag
EPI_DEV int nnls_gram(const NnLds &m, int n, double tol)
{
    for (int j = 0; j < n; j++) { NN_AT(m.x, j) = 0.0; NN_AT(m.w, j) = NN_AT(m.h, j); NN_AT(m.inP, j) = 0; }
    int iter = 0;
    const int itmax = 3 * n;
    for (;;) {
        int t = -1;
        for (int j = 0; j < n; j++)
            if (!NN_AT(m.inP, j) && NN_AT(m.w, j) > tol && (t < 0 || NN_AT(m.w, j) > NN_AT(m.w, t))) t = j;
        if (t < 0) return 1;
        NN_AT(m.inP, t) = 1;
        nnls_solve_passive(m, n);
        for (;;) {
            int any = 0;
            for (int j = 0; j < n; j++) any |= (NN_AT(m.inP, j) && NN_AT(m.z, j) <= 0.0);
            if (!any) break;
            if (++iter > itmax) { for (int j = 0; j < n; j++) NN_AT(m.x, j) = NN_AT(m.z, j); return 0; }
            double alpha = __builtin_inf();
            for (int j = 0; j < n; j++)
                if (NN_AT(m.inP, j) && NN_AT(m.z, j) <= 0.0) alpha = fmin(alpha, NN_AT(m.x, j) / (NN_AT(m.x, j) - NN_AT(m.z, j)));
            for (int j = 0; j < n; j++) NN_AT(m.x, j) = NN_AT(m.x, j) + alpha * (NN_AT(m.z, j) - NN_AT(m.x, j));
            for (int j = 0; j < n; j++)
                if (NN_AT(m.inP, j) && fabs(NN_AT(m.x, j)) < tol) NN_AT(m.inP, j) = 0;
            nnls_solve_passive(m, n);
        }
        for (int j = 0; j < n; j++) NN_AT(m.x, j) = NN_AT(m.z, j);
        for (int i = 0; i < n; i++) {                    // w = C'(d - C x) = h - G x
            double acc = NN_AT(m.G, i) * NN_AT(m.x, 0);
            for (int j = 1; j < n; j++) acc = fma(NN_AT(m.G, i + n * j), NN_AT(m.x, j), acc);
            NN_AT(m.w, i) = NN_AT(m.h, i) - acc;
        }
    }
}

__global__ __launch_bounds__(kNnLanes) void nnls_affine_fit(const NnArgs g)
{
    extern __shared__ double nn_lds[];
    const int lane = threadIdx.x;
    const int s = blockIdx.x * kNnLanes + lane;
    if (s >= g.S) return;
    const int S = g.S, D = g.D, n = g.n;
    NnLds m;
    double *base = nn_lds + lane;
    m.G = base; m.A = m.G + kNnMax * kNnMax * kNnLanes; m.h = m.A + kNnMax * kNnMax * kNnLanes;
    m.b = m.h + kNnMax * kNnLanes; m.yv = m.b + kNnMax * kNnLanes; m.z = m.yv + kNnMax * kNnLanes;
    m.x = m.z + kNnMax * kNnLanes; m.w = m.x + kNnMax * kNnLanes; m.a = m.w + kNnMax * kNnLanes;
    m.coef = m.a + kNnMax * kNnLanes;
    int *ibase = (int *)(nn_lds + (size_t)kNnDoubles * kNnLanes) + lane;
    m.inP = ibase; m.idx = ibase + kNnMax * kNnLanes;

    // one pass over the region's block: G = X'X, column sums of |X| (norm(X,1)); every sum runs over t ascending
    for (int e = 0; e < n * n; e++) NN_AT(m.G, e) = 0.0;
    for (int j = 0; j < n; j++) NN_AT(m.w, j) = 0.0;      // |X| column sums (w is free until the first solve)
    for (int t = 0; t < D; t++) {
        double row[kNnMax];
#pragma unroll
        for (int j = 0; j < kNnMax; j++) row[j] = (j < n) ? g.X[((size_t)t * n + j) * S + s] : 0.0;
#pragma unroll
        for (int i = 0; i < kNnMax; i++) {
            if (i < n) {
                NN_AT(m.w, i) = NN_AT(m.w, i) + fabs(row[i]);
#pragma unroll
                for (int j = 0; j < kNnMax; j++)
                    if (j <= i) NN_AT(m.G, i + n * j) = fma(row[i], row[j], NN_AT(m.G, i + n * j));
            }
        }
    }
    double cmax = 0.0;
    for (int i = 0; i < n; i++) {
        if (i == 0 || NN_AT(m.w, i) > cmax) cmax = NN_AT(m.w, i);
        for (int j = 0; j < i; j++) NN_AT(m.G, j + n * i) = NN_AT(m.G, i + n * j);
    }
    const double tol = 10.0 * kEps * cmax * (double)(D > n ? D : n);   // 10*eps*norm(C,1)*length(C)

    auto build_h = [&](double shift) {                    // h = X'(y - shift)
        for (int j = 0; j < n; j++) NN_AT(m.h, j) = 0.0;
        for (int t = 0; t < D; t++) {
            const double r = g.y[(size_t)t * S + s] - shift;
#pragma unroll
            for (int j = 0; j < kNnMax; j++)
                if (j < n) NN_AT(m.h, j) = fma(g.X[((size_t)t * n + j) * S + s], r, NN_AT(m.h, j));
        }
    };
    auto residual = [&](int t) -> double {                // y(t) - X(t,:)*reg_coef_a
        double xa = g.X[((size_t)t * n) * S + s] * NN_AT(m.a, 0);
        for (int j = 1; j < n; j++) xa = fma(g.X[((size_t)t * n + j) * S + s], NN_AT(m.a, j), xa);
        return g.y[(size_t)t * S + s] - xa;
    };

    build_h(0.0);
    const int flag = nnls_gram(m, n, tol);                // :263
    for (int j = 0; j < n; j++) NN_AT(m.a, j) = NN_AT(m.x, j);
    double bb = 0.0, err = 0.0;
    for (int t = 0; t < D; t++) { const double r = residual(t); err = (t == 0) ? r * r : err + r * r; }   // :265
    int accepted = 0;
    for (int jj = 0; jj < g.max_iters; jj++) {            // :266-276
        build_h(bb);
        nnls_gram(m, n, tol);                             // lsqnonneg(X, y - reg_coef_b)
        for (int j = 0; j < n; j++) NN_AT(m.coef, j) = NN_AT(m.x, j);
        double sum = 0.0;
        for (int t = 0; t < D; t++) { const double r = residual(t); sum = (t == 0) ? r : sum + r; }
        const double c0 = sum / (double)D;                // mean(y - X*reg_coef_a) with the CURRENT reg_coef_a
        double e = 0.0;
        for (int t = 0; t < D; t++) { const double r = residual(t) - c0; e = (t == 0) ? r * r : e + r * r; }
        if (e < err) {
            for (int j = 0; j < n; j++) NN_AT(m.a, j) = NN_AT(m.coef, j);
            bb = c0; err = e; accepted++;
        } else {
            break;
        }
    }
    for (int j = 0; j < n; j++) g.a[(size_t)j * S + s] = NN_AT(m.a, j);
    if (g.b) g.b[s] = bb;
    if (g.min_err) g.min_err[s] = err;
    if (g.iters) g.iters[s] = accepted;
    if (g.flag) g.flag[s] = flag;
}
#undef NN_AT
