// ekf_pair.hpp -- cooperative forward / backward kernels: TWO lanes per chain (6-state generic models).
// Included from epiekf.hip inside namespace epi, after ekf_quad.hpp (whose exchange primitive, scalar-parameter struct and
// monitor kernel it shares).
//
// The middle shape between one lane per chain (ekf_sym.hpp) and four (ekf_quad.hpp): lane h of a pair (neighbouring lanes,
// h = lane & 1) owns ROWS 3h .. 3h+2 of every 6 x 6 matrix, all six columns -- 18 doubles per matrix.  A product costs
// each lane 108 fma (or 63 where the right factor is the Jacobian, whose structural zeros are the same for every lane here
// and are skipped) and the partner's three rows of the other factor arrive by `v_mov_b32_dpp quad_perm`.  Half as many
// wavefronts as the quad shape (32 chains per wave) at about the same per-day instruction stream: the shape for batches
// between 16 384 and 32 768 chains, where quad waves would no longer get a SIMD each.
// Arithmetic: as in ekf_quad.hpp -- an element's whole fma chain (k ascending) runs in one lane; bit-identical results.
#pragma once

constexpr int kPC = kWave / 2;                 // chains per wavefront
constexpr int PP_SWAP = EPI_QP(1, 0, 3, 2);    // my partner
constexpr int PP_LO = EPI_QP(0, 0, 2, 2);      // the lane of my pair that owns rows 0..2
constexpr int PP_HI = EPI_QP(1, 1, 3, 3);      // ... rows 3..5
typedef double rows3[3][6];
// value select.  (`c ? A[i] : A[j]` on two array ELEMENTS is a select of lvalues, i.e. of addresses: the compiler then
// indexes the array with a per-lane offset and the whole array goes to scratch -- measured: 160 B/lane, 2x slower.)
EPI_DEV double psel(bool c, double a, double b) { return c ? a : b; }

struct Pair { int h, lc; bool hi; };

template <int CTRL>
EPI_DEV void px(const rows3 &s, rows3 &d)
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 6; j++) d[r][j] = qx<CTRL>(s[r][j]);
}
// (F + F')/2.0 (GenericEKF.m:138,161,226) for my rows: F(j, i) is local where j is one of my rows, else the partner's
EPI_DEV void psym(const Pair &P, const rows3 &F, rows3 &S)
{
    double O[3][3], Ot[3][3];            // my rows x the OTHER half's columns, and the partner's such block, transposed
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) O[r][c] = psel(P.hi, F[r][c], F[r][3 + c]);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Ot[r][c] = qx<PP_SWAP>(O[c][r]);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double t_lo = psel(P.hi, Ot[r][c], F[c][r]);            // F(j, i), j = c     (rows 0..2)
            const double t_hi = psel(P.hi, F[c][3 + r], Ot[r][c]);        // F(j, i), j = 3 + c (rows 3..5)
            S[r][c] = (F[r][c] + t_lo) / 2.0;
            S[r][3 + c] = (F[r][3 + c] + t_hi) / 2.0;
        }
}
// C(i, j) = sum_k L(i, k) R(k, j), k = 0..5 ascending, for my rows i: Lrow = my rows of the left factor (held, or selected
// from a replicated matrix), R = [Rlo ; Rhi] = all six rows of the right factor (both lanes' rows, by broadcast)
EPI_DEV void pmul(const double (&Lrow)[3][6], const rows3 &Rlo, const rows3 &Rhi, rows3 &C)
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double acc = Lrow[r][0] * Rlo[0][j];
            acc = fma(Lrow[r][1], Rlo[1][j], acc);
            acc = fma(Lrow[r][2], Rlo[2][j], acc);
            acc = fma(Lrow[r][3], Rhi[0][j], acc);
            acc = fma(Lrow[r][4], Rhi[1][j], acc);
            acc = fma(Lrow[r][5], Rhi[2][j], acc);
            C[r][j] = acc;
        }
}
// C(i, j) = sum_q L(i, q) A(j, q) over the structural non-zeros of the Jacobian's row j (the first term a product, then
// fma, q ascending: predict_cov_sym / eks_bwd_sym) -- the pattern depends on j only, so it is the same for both lanes
EPI_DEV void pmul_at(const rows3 &L, const double (&A)[36], rows3 &C)
{
    constexpr int M = 6;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double acc = 0.0;
            bool first = true;
#pragma unroll
            for (int q = 0; q < 6; q++)
                if (a_nz<6>(j, q)) {
                    acc = first ? L[r][q] * A[IXM(j, q)] : fma(L[r][q], A[IXM(j, q)], acc);
                    first = false;
                }
            C[r][j] = acc;
        }
}

// --- addressing (chain-blocked arrays, see Lay / ekf_quad.hpp).  BLK > 0: compile-time lane_block (32) ----------------
// my three rows of a 6 x 6 array stored with all 36 rows (row e = i + 6 j): two lane bases so that every row offset stays
// inside the 12-bit immediate at a 256-byte row pitch
template <int BLK>
EPI_DEV void pstore_rows(double *__restrict__ dst, int t, const Lay &l, const Pair &P, const rows3 &R)
{
    if (!dst) return;
    unsigned voff, rowb;
    const rsrc_t r = qslice<BLK>(dst, t, 36, l, voff, rowb);
    const unsigned v0 = voff + (unsigned)(P.hi ? 3 : 0) * rowb, v1 = v0 + 18u * rowb;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            qst<BLK>(r, v0, (unsigned)(rr + 6 * j), rowb, R[rr][j]);
            qst<BLK>(r, v1, (unsigned)(rr + 6 * j), rowb, R[rr][3 + j]);
        }
}
template <int BLK>
EPI_DEV void pload_rows(const double *__restrict__ src, int t, const Lay &l, const Pair &P, rows3 &R)
{
    unsigned voff, rowb;
    const rsrc_t r = qslice<BLK>(src, t, 36, l, voff, rowb);
    const unsigned v0 = voff + (unsigned)(P.hi ? 3 : 0) * rowb, v1 = v0 + 18u * rowb;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            R[rr][j] = qld<BLK>(r, v0, (unsigned)(rr + 6 * j), rowb);
            R[rr][3 + j] = qld<BLK>(r, v1, (unsigned)(rr + 6 * j), rowb);
        }
}
// 6-vectors are replicated in the pair; lane h stores rows h, h + 2, h + 4
template <int BLK>
EPI_DEV void pstore_vec(double *__restrict__ dst, int t, const Lay &l, const Pair &P, const double (&v)[6])
{
    if (!dst) return;
    unsigned voff, rowb;
    const rsrc_t r = qslice<BLK>(dst, t, 6, l, voff, rowb);
    const unsigned vo = voff + (unsigned)P.h * rowb;
#pragma unroll
    for (int s = 0; s < 3; s++) qst<BLK>(r, vo, (unsigned)(2 * s), rowb, psel(P.hi, v[2 * s + 1], v[2 * s]));
}

// --- the model's NPI vectors, six NPIs per lane: lane h owns k = h, h + 2, ..., h + 10 (cf. QNpi) -----------------------
struct PNpi {
    double a[6], umin[6], umax[6], ew[6], term[6];
    double inv_sigma;
    const double *ga;        // LDS column of this chain: gamma * a(k), k = 0..11, stride kPC
};
EPI_DEV void pload_prm(QPrm &p, PNpi &n, const KArgs &a, int B, int c, const Pair &P, double *ga_col)
{
    auto g = [&](int f) { return a.prm[(size_t)f * B + c]; };
    p.dt = g(EPI_PRM_DT); p.beta = g(EPI_PRM_BETA); p.gamma = g(EPI_PRM_GAMMA);
    p.sigma = g(EPI_PRM_SIGMA); p.b = g(EPI_PRM_B); p.epsilon = g(EPI_PRM_EPSILON);
    p.slo = a.mf.lo_is_zero ? 0.0 : g(EPI_PRM_S_MIN);
    p.ilo = a.mf.lo_is_zero ? 0.0 : g(EPI_PRM_I_MIN);
    p.alpha_min = g(EPI_PRM_ALPHA_MIN); p.alpha_max = g(EPI_PRM_ALPHA_MAX);
    n.inv_sigma = 1.0 / p.sigma;
    n.ga = ga_col;
#pragma unroll
    for (int s = 0; s < 6; s++) {
        const int k = P.h + 2 * s;
        n.a[s] = g(EPI_PRM_A + k); n.umin[s] = g(EPI_PRM_U_MIN + k); n.umax[s] = g(EPI_PRM_U_MAX + k);
        n.ew[s] = p.epsilon * g(EPI_PRM_W_EFF + k);
        n.term[s] = p.gamma * p.dt * (p.sigma / 2.0) * n.a[s] * (n.umax[s] - n.umin[s]);   // as slope_term(): formed once
        ga_col[k * kPC] = p.gamma * n.a[s];
    }
}
EPI_DEV void pload_u(const KArgs &a, int t, int su, const Pair &P, double (&u6)[6])
{
    const unsigned rowb = (unsigned)a.Su * 8u, voff = (unsigned)su * 8u + (unsigned)P.h * rowb;
    const rsrc_t r = mk_rsrc(a.u + (size_t)t * a.n_npi * a.Su, (unsigned)a.n_npi * rowb);   // rows >= n_npi read 0.0 (load_u)
#pragma unroll
    for (int s = 0; s < 6; s++) u6[s] = bld(r, voff, (unsigned)(2 * s) * rowb);
}
template <int BLK>
EPI_DEV void pstore_u(double *__restrict__ dst, const KArgs &a, int t, const Lay &l, const Pair &P, const double (&u6)[6])
{
    if (!dst) return;
    unsigned voff, rowb;
    const rsrc_t r = qslice<BLK>(dst, t, (unsigned)a.n_npi, l, voff, rowb);
    const unsigned vo = voff + (unsigned)P.h * rowb;
    if (a.n_npi == kNpi) {
#pragma unroll
        for (int s = 0; s < 6; s++) qst<BLK>(r, vo, (unsigned)(2 * s), rowb, u6[s]);
        return;
    }
#pragma unroll
    for (int s = 0; s < 6; s++)
        if (P.h + 2 * s < a.n_npi) qst<BLK>(r, vo, (unsigned)(2 * s), rowb, u6[s]);
}
EPI_DEV void presolve(const QPrm &p, const PNpi &n, const ModelFlags &mf, double s6, const double (&u6)[6], double (&ur)[6],
                      double (&phi)[6])
{
    const double gs6 = p.gamma * s6;
#pragma unroll
    for (int s = 0; s < 6; s++) {
        phi[s] = n.ew[s] - gs6 * n.a[s];
        const bool lo = mf.phi_ge ? (phi[s] >= 0.0) : (phi[s] > 0.0);
        ur[s] = psel(is_nan(u6[s]), psel(lo, n.umin[s], n.umax[s]), u6[s]);
    }
}
EPI_DEV void pgather12(const double (&v6)[6], double (&v)[kNpi])
{
#pragma unroll
    for (int s = 0; s < 6; s++) { v[2 * s] = qx<PP_LO>(v6[s]); v[2 * s + 1] = qx<PP_HI>(v6[s]); }
}
EPI_DEV double pdot(const PNpi &n, const double (&ur)[6])
{
    double d6[6], d[kNpi];
#pragma unroll
    for (int s = 0; s < 6; s++) d6[s] = n.umax[s] - ur[s];
    pgather12(d6, d);
    double dot = n.ga[0] * d[0];
#pragma unroll
    for (int k = 1; k < kNpi; k++) dot = fma(n.ga[k * kPC], d[k], dot);
    return dot;
}
template <int FLIP>
EPI_DEV double pslope(const PNpi &n, const double (&u6)[6], const double (&phi)[6])
{
    bool any_free = false;
#pragma unroll
    for (int s = 0; s < 6; s++) any_free = any_free || is_nan(u6[s]);
    if (__builtin_amdgcn_ballot_w64(any_free) == 0ull) return 0.0;
    double tm6[6], tm[kNpi];
#pragma unroll
    for (int s = 0; s < 6; s++)
        tm6[s] = psel(is_nan(u6[s]) && phi[s] > -n.inv_sigma && phi[s] < n.inv_sigma, n.term[s], 0.0);
    pgather12(tm6, tm);
    double a36 = 0.0;
#pragma unroll
    for (int k = 0; k < kNpi; k++) a36 = FLIP ? (a36 + tm[k]) : (a36 - tm[k]);
    return a36;
}

// ---------------------------------------------------------------------------
// forward pass (monitor always hoisted: launched only when r_mode = 1; see ekf_monitor)
// ---------------------------------------------------------------------------
template <int FLIP, int BLK>
__global__ __launch_bounds__(kWave) void ekf_fwd_pair(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int M = 6;
    __shared__ double galds[kNpi * kPC];
    if (*dense_flag) return;
    Pair P;
    P.h = threadIdx.x & 1; P.lc = threadIdx.x >> 1; P.hi = P.h != 0;
    const int c = a.c0 + blockIdx.x * kPC + P.lc;
    if (c >= a.c0 + a.cn) return;               // whole pairs leave together
    const int B = a.B, T = a.T;
    const int sx = a.x_series ? a.x_series[c] : c;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    QPrm p;
    PNpi np;
    pload_prm(p, np, a, B, c, P, galds + P.lc);
    const double v_bar = a.prm[(size_t)EPI_PRM_V_BAR * B + c];
    const double gamma = a.prm[(size_t)EPI_PRM_GAMMA_EKF * B + c];

    double sk_minus[M], Qd[M];
    rows3 Pm;
#pragma unroll
    for (int i = 0; i < M; i++) { sk_minus[i] = a.s_init[(size_t)i * B + c]; Qd[i] = a.Q[(size_t)IXM(i, i) * B + c]; }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int i = (P.hi ? 3 : 0) + r;
            Pm[r][j] = a.Ps_init[(size_t)(i < j ? IXM(i, j) : IXM(j, i)) * B + c];     // bit-wise symmetric (ekf_precheck)
        }
    const int k_begin = a.k_begin, k_end = (a.k_end > 0 && a.k_end < T) ? a.k_end : T;   // time segments: see ekf_fwd_sym
    if (k_begin > 0) {
        qload_vec<BLK>(a.S_MINUS, tpos<FLIP>(k_begin, T), lay, sk_minus);
        pload_rows<BLK>(a.P_MINUS, tpos<FLIP>(k_begin, T), lay, P, Pm);
    }
    const unsigned voff_x = (unsigned)sx * 8u;
    double x_nxt = ldg(a.x + (size_t)tpos<FLIP>(k_begin, T) * a.Sx, voff_x);
    double r_nxt = ldg(a.R_series + (size_t)k_begin * a.Sx, voff_x);
    double u_nxt[6];
    pload_u(a, tpos<FLIP>(k_begin, T), su, P, u_nxt);

    for (int k = k_begin; k < k_end; k++) {
        const int t = tpos<FLIP>(k, T);
        const double Rk = r_nxt, xk = x_nxt;
        double u_in[6];
#pragma unroll
        for (int s = 0; s < 6; s++) u_in[s] = u_nxt[s];
        if (k + 1 < T) {
            const int tn = tpos<FLIP>(k + 1, T);
            x_nxt = ldg(a.x + (size_t)tn * a.Sx, voff_x);
            r_nxt = ldg(a.R_series + (size_t)(k + 1) * a.Sx, voff_x);
            pload_u(a, tn, su, P, u_nxt);
        }
        pstore_vec<BLK>(a.S_MINUS, t, lay, P, sk_minus);
        pstore_rows<BLK>(a.P_MINUS, t, lay, P, Pm);

        double C[M];
        obs_jacobian<M>(a.mf, sk_minus, C);                 // C(4:6) == 0
        const double xk_minus = predict_obs<M>(a.mf, sk_minus, v_bar);
        double innov, K[M], sk_plus[M];
        rows3 Pp;
        const bool valid = !is_nan(xk);
        if (valid) {
            innov = xk - xk_minus;
            double PCo[3], Ko[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                double acc = Pm[r][0] * C[0];
                acc = fma(Pm[r][1], C[1], acc);
                acc = fma(Pm[r][2], C[2], acc);
                PCo[r] = acc;                                // (P C')(3h + r)
            }
            double CPCt = qx<PP_LO>(PCo[0]) * C[0];
            CPCt = fma(qx<PP_LO>(PCo[1]), C[1], CPCt);
            CPCt = fma(qx<PP_LO>(PCo[2]), C[2], CPCt);
            const double den = CPCt + gamma * Rk;
#pragma unroll
            for (int r = 0; r < 3; r++) Ko[r] = PCo[r] / den;
#pragma unroll
            for (int r = 0; r < 3; r++) { K[r] = qx<PP_LO>(Ko[r]); K[3 + r] = qx<PP_HI>(Ko[r]); }
            // (I - K C), first three columns: all six rows (right factor) and my rows (left factor)
            double IK[6][3], IKo[3][3];
#pragma unroll
            for (int j = 0; j < 6; j++)
#pragma unroll
                for (int q = 0; q < 3; q++) IK[j][q] = ((j == q) ? 1.0 : 0.0) - K[j] * C[q];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int q = 0; q < 3; q++) IKo[r][q] = psel(P.hi, IK[3 + r][q], IK[r][q]);
            // Joseph form :127 (cf. ekf_fwd_sym): T1 = (I - K C) P, F = (T1 (I - K C)' + K R K') / gamma
            rows3 Plo, T1, F;
            px<PP_LO>(Pm, Plo);
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    double acc = IKo[r][0] * Plo[0][j];
                    acc = fma(IKo[r][1], Plo[1][j], acc);
                    acc = fma(IKo[r][2], Plo[2][j], acc);
                    const double plus = acc + Pm[r][j];              // + 1 * P(i, j) for i >= 3
                    T1[r][j] = psel(P.hi, plus, acc);
                }
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    double acc = T1[r][0] * IK[j][0];
                    acc = fma(T1[r][1], IK[j][1], acc);
                    acc = fma(T1[r][2], IK[j][2], acc);
                    if (j >= 3) acc = acc + T1[r][j];
                    F[r][j] = (acc + (Ko[r] * Rk) * K[j]) / gamma;
                }
            psym(P, F, Pp);
#pragma unroll
            for (int i = 0; i < M; i++) sk_plus[i] = sk_minus[i] + K[i] * innov;
        } else {
            innov = 0.0;
#pragma unroll
            for (int i = 0; i < M; i++) { K[i] = 0.0; sk_plus[i] = sk_minus[i]; }
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int j = 0; j < 6; j++) Pp[r][j] = Pm[r][j];
        }
        pstore_vec<BLK>(a.K_GAIN, t, lay, P, K);
        qstore_scalar(a.innovations, t, lay, innov);
        state_hard_margins<M>(p, sk_plus);
        pstore_vec<BLK>(a.S_PLUS, t, lay, P, sk_plus);
        pstore_rows<BLK>(a.P_PLUS, t, lay, P, Pp);

        double u_app[6], phi[6];
        presolve(p, np, a.mf, sk_plus[5], u_in, u_app, phi);
        state_map<M, FLIP>(p, pdot(np, u_app), sk_plus, sk_minus);
        pstore_u<BLK>(a.u_opt, a, t, lay, P, u_app);
        {
            // P(k+1|k) = sym(A P A' + Q)  :158-161 (predict_cov_sym): T1 = A P by my rows of A, G = T1 A' with the zeros of A skipped
            double A[M * M], Ar[3][6];
            jacobian_entries<M, FLIP>(p, sk_plus, pslope<FLIP>(np, u_in, phi), A);
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int q = 0; q < 6; q++) Ar[r][q] = psel(P.hi, A[IXM(3 + r, q)], A[IXM(r, q)]);
            rows3 Plo, Phi, T1, G;
            px<PP_LO>(Pp, Plo);
            px<PP_HI>(Pp, Phi);
            pmul(Ar, Plo, Phi, T1);
            pmul_at(T1, A, G);
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    // + Q_w(i, j): diagonal, i = 3h + r  (acc + ((i == j) ? Qd[i] : 0.0) in predict_cov_sym)
                    const double add = (j == r) ? psel(P.hi, 0.0, Qd[r]) : (j == r + 3) ? psel(P.hi, Qd[3 + r], 0.0) : 0.0;
                    G[r][j] = G[r][j] + add;
                }
            psym(P, G, Pm);
        }
        state_hard_margins<M>(p, sk_minus);
    }
    if (k_end < T) {       // hand-over to the next time segment
        pstore_vec<BLK>(a.S_MINUS, tpos<FLIP>(k_end, T), lay, P, sk_minus);
        pstore_rows<BLK>(a.P_MINUS, tpos<FLIP>(k_end, T), lay, P, Pm);
    }
}

// ---------------------------------------------------------------------------
// backward recursion (X = pinv(P_MINUS) from eks_pinv, stored with all 36 rows for this shape: KArgs.x_full)
// ---------------------------------------------------------------------------
template <int FLIP, int BLK>
__global__ __launch_bounds__(kWave) void eks_bwd_pair(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int M = 6;
    __shared__ double galds[kNpi * kPC];
    if (*dense_flag) return;
    Pair P;
    P.h = threadIdx.x & 1; P.lc = threadIdx.x >> 1; P.hi = P.h != 0;
    const int c = a.c0 + blockIdx.x * kPC + P.lc;
    if (c >= a.c0 + a.cn) return;
    const int B = a.B, T = a.T;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    QPrm p;
    PNpi np;
    pload_prm(p, np, a, B, c, P, galds + P.lc);

    // terminal conditions GenericEKF.m:189-202 (Ps_final symmetric in values and NaN pattern: ekf_precheck)
    double Ss[M];
    rows3 Ps;
    const int tT = tpos<FLIP>(T - 1, T);
    qload_vec<BLK>(a.S_PLUS, tT, lay, Ss);
#pragma unroll
    for (int i = 0; i < M; i++) {
        const double f = a.s_final[(size_t)i * B + c];
        if (!is_nan(f)) Ss[i] = f;
    }
    pload_rows<BLK>(a.P_PLUS, tT, lay, P, Ps);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int i = (P.hi ? 3 : 0) + r;
            const double f = a.Ps_final[(size_t)(i < j ? IXM(i, j) : IXM(j, i)) * B + c];
            if (!is_nan(f)) Ps[r][j] = f;
        }
    pstore_vec<BLK>(a.S_SMOOTH, tT, lay, P, Ss);
    pstore_rows<BLK>(a.P_SMOOTH, tT, lay, P, Ps);
    if (a.u_opt_smooth) {
        const double z[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        pstore_u<BLK>(a.u_opt_smooth, a, tT, lay, P, z);
    }
    qstore_scalar(a.pinv_rank, tT, lay, (int32_t)-1);

    int st_guard = 0, st_cap = 0, min_rank = M;
    for (int k = T - 2; k >= 0; k--) {
        const int t = tpos<FLIP>(k, T), t1 = tpos<FLIP>(k + 1, T);
        double Sp[M], Sm1[M], u_in[6];
        rows3 Pp, X, Pm1;
        qload_vec<BLK>(a.S_PLUS, t, lay, Sp);
        pload_u(a, t, su, P, u_in);
        const int rk = a.rankbuf[lay_scalar(t1, lay)];
        pload_rows<BLK>(a.P_PLUS, t, lay, P, Pp);
        pload_rows<BLK>(a.X, t1, lay, P, X);                  // (garbage where the :211 guard fired, rk < 0: unused)
        qload_vec<BLK>(a.S_MINUS, t1, lay, Sm1);
        pload_rows<BLK>(a.P_MINUS, t1, lay, P, Pm1);

        double A[M * M];
        {
            double ur_unused[6], phi[6];
            presolve(p, np, a.mf, Sp[5], u_in, ur_unused, phi);
            jacobian_entries<M, FLIP>(p, Sp, pslope<FLIP>(np, u_in, phi), A);   // :206
        }
        rows3 J;
        int rank = -1;
        if (rk < 0) {                                          // non-finite P_MINUS guard :211-213
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int j = 0; j < 6; j++) J[r][j] = 0.0;
            st_guard = 1;
        } else {
            rows3 PA, Xlo, Xhi;                                // J = (P+ A') X  :215
            pmul_at(Pp, A, PA);
            px<PP_LO>(X, Xlo);
            px<PP_HI>(X, Xhi);
            pmul(PA, Xlo, Xhi, J);
            rank = rk & 0xff;
            st_cap |= (rk >> 8) & 1;
            min_rank = rank < min_rank ? rank : min_rank;
        }
        double Sn[M];
        {
            double dv[M], So[3];
#pragma unroll
            for (int i = 0; i < M; i++) dv[i] = Ss[i] - Sm1[i];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                double acc = J[r][0] * dv[0];
#pragma unroll
                for (int j = 1; j < M; j++) acc = fma(J[r][j], dv[j], acc);
                So[r] = psel(P.hi, Sp[3 + r], Sp[r]) + acc;      // :218
            }
#pragma unroll
            for (int r = 0; r < 3; r++) { Sn[r] = qx<PP_LO>(So[r]); Sn[3 + r] = qx<PP_HI>(So[r]); }
        }
        state_hard_margins<M>(p, Sn);                          // :221
        {
            // P_SMOOTH(k) = sym(P+ - (J D) J'),  D = P_MINUS(k+1) - P_SMOOTH(k+1)   :223-226
            rows3 D, Dlo, Dhi, T1, Jlo, Jhi, F;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int j = 0; j < 6; j++) D[r][j] = Pm1[r][j] - Ps[r][j];
            px<PP_LO>(D, Dlo);
            px<PP_HI>(D, Dhi);
            pmul(J, Dlo, Dhi, T1);
            px<PP_LO>(J, Jlo);
            px<PP_HI>(J, Jhi);
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const rows3 &Jh = (j < 3) ? Jlo : Jhi;            // (j is a compile-time constant after unrolling)
                    constexpr int dummy = 0; (void)dummy;
                    const int jr = (j < 3) ? j : j - 3;
                    double acc = T1[r][0] * Jh[jr][0];
#pragma unroll
                    for (int q = 1; q < 6; q++) acc = fma(T1[r][q], Jh[jr][q], acc);
                    F[r][j] = Pp[r][j] - acc;
                }
            psym(P, F, Ps);
        }
#pragma unroll
        for (int i = 0; i < M; i++) Ss[i] = Sn[i];
        qstore_scalar(a.pinv_rank, t, lay, (int32_t)rank);
        pstore_vec<BLK>(a.S_SMOOTH, t, lay, P, Ss);
        pstore_rows<BLK>(a.P_SMOOTH, t, lay, P, Ps);
        if (a.u_opt_smooth) {                                  // :229 -- only the control NlinStateUpdate returns is kept
            double ur[6], phi_unused[6];
            presolve(p, np, a.mf, Ss[5], u_in, ur, phi_unused);
            pstore_u<BLK>(a.u_opt_smooth, a, t, lay, P, ur);
        }
    }
    if (a.status) a.status[c] = st_guard | (st_cap << 1) | (min_rank << 8);
}
