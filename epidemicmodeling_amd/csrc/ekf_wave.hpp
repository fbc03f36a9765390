// ekf_wave.hpp -- ONE WAVEFRONT PER CHAIN (epi_batch_desc.shape = 3): the lane mapping for batches of at most one
// chain per SIMD -- the unchanged reference caller's one call per cost weight (Tools/TrainPredictPrescribeNPI.m:421,460,
// B = 1), a region's 250 cost weights, the 300 regions of a forecast.  Included by epiekf.hip after ekf_quad.hpp.
//
// 6-state generic models (SIAlphaModelEKFOptControlled and its time-flipped wrapper), R_v a per-day series (the
// innovation monitor is then replayed by ekf_monitor), fixed Q_w, fp64 storage.
//
//   * lane e = i + 6 j (e < 36) owns element (i, j) of every 6 x 6 matrix of the chain -- MATLAB's column-major order,
//     so a wavefront stores a covariance with ONE instruction, 36 consecutive doubles when the layout keeps a chain's rows
//     together (lane_block = 1, what epi_ekf_preferred_lane_block returns for this shape);
//   * lane k < 12 also owns NPI k: its a(k), u_min(k), u_max(k), epsilon w(k), its control u(k, t), the bang-bang
//     substitution and its slope-term contribution; the two k-ascending reductions over the NPIs run in every lane on the
//     twelve operands fetched back from LDS;
//   * the state, the gain and C are held redundantly by all lanes (wave-uniform), so the model callbacks of ekf_device.hpp
//     (state_map, jacobian_entries, obs_jacobian, the clamps) run unchanged;
//   * a matrix product is 6 fma per lane: the row / column operands come through LDS (a 36-double tile written by its
//     owners and read back with ds_read2_b64 / ds_read_b128 -- the "wavefront shuffle" for operands that 36 lanes need
//     in 6 different arrangements); every element's fma chain runs k-ascending in ONE lane, dense (no structural zero
//     is skipped), which is the C oracle's and the dense kernels' rounding sequence term for term: results are bit for
//     bit the other shapes', non-finite values included;
//   * broadcasts of a single lane's value (C P, the gain, J d) go through v_readlane, transposed elements through
//     ds_bpermute -- no LDS write, no wait for it; the Jacobian's 21 non-zero entries are each evaluated by the lane that
//     owns them (w_jac_entry) instead of 45 wave-uniform multiplications;
//   * the step's inputs and, in the smoother, the stored forward quantities are requested one step ahead; every lane walks
//     its own element of each array with a 64-bit pointer in VGPRs.
#pragma once
// (included inside namespace epi, like ekf_sym.hpp / ekf_quad.hpp)

constexpr int kWE = 36;     // lanes that own a matrix element

struct WaveLane {
    int e, i, j;            // element, row, column (lanes >= 36 mirror element 0: their results are never stored)
    int k;                  // NPI owned (lanes >= 12 mirror NPI 0: never stored)
    bool own, npi, first;   // owns a matrix element / an NPI of this batch (k < n_npi) / is lane 0
};

EPI_DEV void w_row(const double *buf, int r, double (&o)[6])
{
#pragma unroll
    for (int q = 0; q < 6; q++) o[q] = buf[r + 6 * q];
}
EPI_DEV void w_col(const double *buf, int c, double (&o)[6])
{
#pragma unroll
    for (int q = 0; q < 6; q++) o[q] = buf[q + 6 * c];
}
EPI_DEV double w_dot6(const double (&x)[6], const double (&y)[6])     // x[0]*y[0], then fma k ascending
{
    double acc = x[0] * y[0];
#pragma unroll
    for (int q = 1; q < 6; q++) acc = fma(x[q], y[q], acc);
    return acc;
}
EPI_DEV WaveLane w_lane(const KArgs &a)
{
    WaveLane w;
    w.own = threadIdx.x < kWE;
    w.e = w.own ? (int)threadIdx.x : 0;
    w.j = w.e / 6; w.i = w.e - 6 * w.j;
    w.npi = (int)threadIdx.x < a.n_npi;
    w.k = threadIdx.x < kNpi ? (int)threadIdx.x : 0;
    w.first = threadIdx.x == 0;
    return w;
}
// Addressing: every lane walks its OWN element of each array with a 64-bit pointer held in VGPRs (one add per array and
// step); element (t, row, c) of an array with `rows` rows sits at ((t * nblk + cb) * rows + row) * blk + cr doubles, and one
// time step moves by rows * bp doubles (down the caller's time axis for the time-flipped wrappers).  No buffer descriptors:
// fourteen of them would not fit the scalar registers beside the lane masks of the Jacobian's entry kinds.
EPI_DEV size_t w_elem(const Lay &l, int t, unsigned rows, unsigned row)
{
    return ((size_t)t * l.nblk + l.cb) * rows * l.blk + (size_t)row * l.blk + l.cr;
}
// a wave-uniform 6-vector at `p` (the position of row 0; rows `blk` doubles apart): written by the calling lane
EPI_DEV void w_put_vec(double *p, unsigned blk, const double (&v)[6])
{
#pragma unroll
    for (int q = 0; q < 6; q++) p[(size_t)q * blk] = v[q];
}
EPI_DEV void w_get_vec(const double *p, unsigned blk, double (&v)[6])
{
#pragma unroll
    for (int q = 0; q < 6; q++) v[q] = p[(size_t)q * blk];
}
// keeps a wave-uniform value in a VGPR (the compiler would otherwise hold the ~25 model constants of a chain in scalar
// registers, and spill the lane masks that select the Jacobian's entry kinds)
EPI_DEV double w_vgpr(double v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// ---- cross-lane moves without an LDS write ---------------------------------------------------------------------------
// a value held by lane `src` (compile-time lane) to every lane, through two v_readlane_b32
EPI_DEV double w_bcast(double v, int src)
{
    const u32x2 b = __builtin_bit_cast(u32x2, v);
    u32x2 r;
    r.x = (unsigned)__builtin_amdgcn_readlane((int)b.x, src);
    r.y = (unsigned)__builtin_amdgcn_readlane((int)b.y, src);
    return __builtin_bit_cast(double, r);
}
// the value of lane `srcb / 4` (per-lane choice), through two ds_bpermute_b32 (the LDS crossbar, no LDS memory)
EPI_DEV double w_from(double v, int srcb)
{
    const u32x2 b = __builtin_bit_cast(u32x2, v);
    u32x2 r;
    r.x = (unsigned)__builtin_amdgcn_ds_bpermute(srcb, (int)b.x);
    r.y = (unsigned)__builtin_amdgcn_ds_bpermute(srcb, (int)b.y);
    return __builtin_bit_cast(double, r);
}

// ---- the Jacobian, every entry by the lane that owns it --------------------------------------------------------------
// StateJacobians (OptControlled.m:89-135, Backward...m:109-156) has four kinds of entries; each lane evaluates its own with
// the operation order of jacobian_entries() (ekf_device.hpp), which the dense kernels and the C oracle share:
//   S1     base + sgn ((dt * X) * Y)      X in {s1, s2, s3}, Y in {s1, s2, s3, rho}, base 0 or 1
//   S2     1 + sgn (dt * (s1 * s3 - beta))                                        (entries (2,2) and (5,5), 1-based)
//   CONST  1 -/+ dt * gamma ((3,3), (6,6)), or a structural zero
//   A36    the slope term of the bang-bang control ((3,6)): left 0.0 here, patched in by the lanes that read row 3
// `-(x)` and `1.0 + (-x)` are the bits of `-dt * ...` and `1.0 - ...` as written there.
struct WaveJac {
    bool x0, x1, y0, y1, y2, s2, neg, base1, cst;   // X = x0 ? s1 : (x1 ? s2 : s3);  Y = y0 ? s1 : (y1 ? s2 : (y2 ? s3 : rho))
    double c;                                       // CONST value
};
template <int FLIP>
EPI_DEV WaveJac w_jac_setup(const QPrm &p, const WaveLane &w)
{
    // per entry e = i + 6 j: kind (0 zero, 1 S1, 2 S2, 3 CONST, 4 A36), X, Y (3 = rho), sign (1 = minus), base
    struct E { unsigned char kind, x, y, neg, base1; };
    const E Z{0, 0, 0, 0, 0};
    E tab[36];
#pragma unroll
    for (int q = 0; q < 36; q++) tab[q] = Z;
    auto set = [&](int i, int j, E v) { tab[i + 6 * j] = v; };
    set(0, 0, E{1, 2, 1, 1, 1}); set(0, 1, E{1, 2, 0, 1, 0}); set(0, 2, E{1, 0, 1, 1, 0});
    set(1, 0, E{1, 1, 2, 0, 0}); set(1, 1, E{2, 0, 0, 0, 1}); set(1, 2, E{1, 0, 1, 0, 0});
    set(2, 2, E{3, 0, 0, 1, 1}); set(2, 5, E{4, 0, 0, 0, 0});
    set(3, 1, E{1, 2, 3, 0, 0}); set(3, 2, E{1, 1, 3, 0, 0}); set(3, 3, E{1, 1, 2, 0, 1}); set(3, 4, E{1, 1, 2, 1, 0});
    set(4, 0, E{1, 2, 3, 0, 0}); set(4, 2, E{1, 0, 3, 0, 0}); set(4, 3, E{1, 0, 2, 0, 0}); set(4, 4, E{2, 0, 0, 1, 1});
    set(5, 0, E{1, 1, 3, 0, 0}); set(5, 1, E{1, 0, 3, 0, 0}); set(5, 3, E{1, 0, 1, 0, 0}); set(5, 4, E{1, 0, 1, 1, 0});
    set(5, 5, E{3, 0, 0, 0, 1});
    E me = Z;
#pragma unroll
    for (int q = 0; q < 36; q++)
        if (w.e == q) me = tab[q];
    WaveJac j;
    j.x0 = me.x == 0; j.x1 = me.x == 1;
    j.y0 = me.y == 0; j.y1 = me.y == 1; j.y2 = me.y == 2;
    j.s2 = me.kind == 2;
    j.neg = (me.neg != 0) != (FLIP != 0);          // the time-flipped model reverses the sign of every dt term
    j.base1 = me.base1 != 0;
    j.cst = me.kind == 0 || me.kind == 3 || me.kind == 4;
    const double dg = p.dt * p.gamma;
    j.c = me.kind == 3 ? (j.neg ? 1.0 - dg : 1.0 + dg) : 0.0;
    return j;
}
EPI_DEV double w_jac_entry(const QPrm &p, const WaveJac &j, const double (&s)[6])
{
    const double rho = s[3] - s[4] - (1.0 - p.epsilon);
    const double w2 = p.dt * (s[0] * s[2] - p.beta);
    const double X = j.x0 ? s[0] : (j.x1 ? s[1] : s[2]);
    const double Y = j.y0 ? s[0] : (j.y1 ? s[1] : (j.y2 ? s[2] : rho));
    double v = p.dt * X * Y;
    v = j.s2 ? w2 : v;
    v = j.neg ? -v : v;
    const double r = j.base1 ? 1.0 + v : v;
    return j.cst ? j.c : r;
}
// rows i and j of the Jacobian as the owners left them in LDS, with the slope term patched into entry (3,6)
EPI_DEV void w_jac_rows(const double *sA, const WaveLane &w, double a36, double (&ri)[6], double (&rj)[6])
{
    w_row(sA, w.i, ri);
    w_row(sA, w.j, rj);
    ri[5] = (w.i == 2) ? a36 : ri[5];
    rj[5] = (w.j == 2) ? a36 : rj[5];
}

// ---- the NPI lanes ------------------------------------------------------------------------------------------------
struct WaveNpi {
    double a, umin, umax, ew, term;   // a(k), u_min(k), u_max(k), epsilon*w(k), gamma*dt*(sigma/2)*a(k)*(u_max(k)-u_min(k))
    double inv_sigma;
};
// sGa[12] = gamma * a(k): the constant factors of the fma chain of NlinStateUpdate (OptControlled.m:64)
EPI_DEV void w_load_prm(QPrm &p, WaveNpi &n, const KArgs &a, int B, int c, const WaveLane &w, double *sGa)
{
    auto g = [&](int f) { return a.prm[(size_t)f * B + c]; };
    p.dt = g(EPI_PRM_DT); p.beta = g(EPI_PRM_BETA); p.gamma = g(EPI_PRM_GAMMA);
    p.sigma = g(EPI_PRM_SIGMA); p.b = g(EPI_PRM_B); p.epsilon = g(EPI_PRM_EPSILON);
    p.slo = a.mf.lo_is_zero ? 0.0 : g(EPI_PRM_S_MIN);
    p.ilo = a.mf.lo_is_zero ? 0.0 : g(EPI_PRM_I_MIN);
    p.alpha_min = g(EPI_PRM_ALPHA_MIN); p.alpha_max = g(EPI_PRM_ALPHA_MAX);
    p.dt = w_vgpr(p.dt); p.beta = w_vgpr(p.beta); p.gamma = w_vgpr(p.gamma); p.sigma = w_vgpr(p.sigma); p.b = w_vgpr(p.b);
    p.epsilon = w_vgpr(p.epsilon); p.slo = w_vgpr(p.slo); p.ilo = w_vgpr(p.ilo);
    p.alpha_min = w_vgpr(p.alpha_min); p.alpha_max = w_vgpr(p.alpha_max);
    n.inv_sigma = 1.0 / p.sigma;
    n.a = g(EPI_PRM_A + w.k); n.umin = g(EPI_PRM_U_MIN + w.k); n.umax = g(EPI_PRM_U_MAX + w.k);
    n.ew = p.epsilon * g(EPI_PRM_W_EFF + w.k);
    // same products, same order as slope_term() / nlin_state_update(): constants of the chain, formed once
    n.term = p.gamma * p.dt * (p.sigma / 2.0) * n.a * (n.umax - n.umin);
    if (threadIdx.x < kNpi) sGa[w.k] = p.gamma * n.a;
}
// u(k, t) of my NPI: rows beyond n_npi read as 0.0, the padding value of load_u()
EPI_DEV double w_load_u(const double *pu, const WaveLane &w) { return w.npi ? *pu : 0.0; }
// my NPI at state s: phi (OptControlled.m:49), the control applied (:50-58, strict >) and my slope-term contribution
// (:107-114; 0.0 where the dense code adds nothing -- x - 0.0 == x, so the running sum keeps its bits)
EPI_DEV void w_npi(const QPrm &p, const WaveNpi &n, double u, double s6, double &uapp, double &tterm, bool phi_ge = false)
{
    const double gs6 = p.gamma * s6;
    const double phi = n.ew - gs6 * n.a;
    const bool free_u = is_nan(u);
    uapp = free_u ? ((phi_ge ? (phi >= 0.0) : (phi > 0.0)) ? n.umin : n.umax) : u;     // strict > (OptControlled.m:50), >= in NewCase...m:175
    tterm = (free_u && phi > -n.inv_sigma && phi < n.inv_sigma) ? n.term : 0.0;
}
// (gamma*a') * (u_max - u): `d` [12] = u_max(k) - u(k) as left in LDS by the NPI lanes
EPI_DEV double w_dot_npi(const double *sGa, const double *d)
{
    double dot = sGa[0] * d[0];
#pragma unroll
    for (int q = 1; q < kNpi; q++) dot = fma(sGa[q], d[q], dot);
    return dot;
}
template <int FLIP>
EPI_DEV double w_slope(const double *tt)
{
    double a36 = 0.0;
#pragma unroll
    for (int q = 0; q < kNpi; q++) a36 = FLIP ? (a36 + tt[q]) : (a36 - tt[q]);
    return a36;
}

// ---------------------------------------------------------------------------
// forward pass: GenericExtendedKalmanFilter.m:98-169 (the monitor :172-179 is replayed by ekf_monitor)
// ---------------------------------------------------------------------------
// GEN = 0 (round 5): Tools/NewCaseEKFEstimatorWithOptimalNPI.m, the older fused filter -- P+ = (I - K C) P- / gamma (:64), nothing
// symmetrised (so (C P)(j) and (P C')(j) differ: only the latter makes the gain), the running scalar R of :110-112, rho from
// cc / R without eps (:108), phi >= 0 (:175); any Ps_init and Q_w (every lane owns its element).  Always with MON = 1.
// MON = 1 (round 5): R_v is a scalar -- fixed, or adapted from the innovation statistics (GenericEKF.m:180-185, beta != 1:
// testScripts/testPrescribeXPRIZE01.m:211) -- so the innovation monitor :172-179 cannot be replayed afterwards: it runs inline.
// The chain's innovation is wave-uniform; the three L-sample windows are double-written rings in LDS (see qring_sum) that every
// lane sums newest -> oldest for itself (the reads broadcast).  The adaptive R(k+1) depends on day k's sum, so the ~60 dependent
// additions are on the day's critical path: 0.83 instead of 0.72 ms for one chain of 520 days -- against 1.25 in the quad shape,
// which such calls took until now.
template <int FLIP, int MON = 0, int LC = 0, int GEN = 1>
__global__ __launch_bounds__(kWave) void ekf_fwd_wave(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int M = 6;
    // tiles are 64 entries so that every lane writes its slot unconditionally (lanes >= 36 own padding)
    __shared__ double sP[kWave], sT[kWave], sA[kWave], sGa[kNpi], sD[kWave], sTt[kWave];
    extern __shared__ double wlds[];          // MON: three windows [3][2 L]
    if (dense_flag && *dense_flag) return;   // (the NewCase models have no packed / dense choice: no flag)
    const int c = a.c0 + (int)blockIdx.x;
    if (c >= a.c0 + a.cn) return;
    const int B = a.B, T = a.T;
    const int sx = a.x_series ? a.x_series[c] : c;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    const WaveLane w = w_lane(a);
    const unsigned lane = threadIdx.x;
    const int tr_b = (w.j + 6 * w.i) * 4;               // ds_bpermute address of the lane that owns the transposed element

    QPrm p;
    WaveNpi np;
    w_load_prm(p, np, a, B, c, w, sGa);
    const WaveJac jc = w_jac_setup<FLIP>(p, w);
    const double v_bar = w_vgpr(a.prm[(size_t)EPI_PRM_V_BAR * B + c]);
    const double gamma = w_vgpr(a.prm[(size_t)EPI_PRM_GAMMA_EKF * B + c]);
    const double beta = MON ? w_vgpr(a.prm[(size_t)EPI_PRM_BETA_EKF * B + c]) : 1.0;

    double sk_minus[M];
#pragma unroll
    for (int i = 0; i < M; i++) sk_minus[i] = a.s_init[(size_t)i * B + c];
    double Pm = a.Ps_init[(size_t)w.e * B + c];
    const double Qe = a.Q[(size_t)w.e * B + c];
    // the monitor's windows and the running R (see ekf_fwd_quad)
    const int L = LC ? LC : a.L;        // LC > 0: the window length is that compile-time constant (21 is what every caller passes)
    double *winMean = wlds, *winCov = wlds + 2 * L, *winCovN = wlds + 4 * L;
    if (MON) {
        for (int q = (int)lane; q < 6 * L; q += kWave) wlds[q] = 0.0;
        __syncthreads();
    }
    int pos = 0;
    const bool fixed_R = MON && a.r_mode == 0;
    const double R_v = fixed_R ? a.R_scalar[c] : 0.0;
    double R_next = R_v;

    const int k_begin = a.k_begin, k_end = (a.k_end > 0 && a.k_end < T) ? a.k_end : T;
    const int t0 = tpos<FLIP>(k_begin, T);
    const long dir = FLIP ? -1L : 1L;
    const long bp = (long)lay.bp;
    const unsigned blk = lay.blk;
    // this lane's element of time slice t0 of every array the kernel writes, and the per-step strides
    double *pSm = a.S_MINUS + w_elem(lay, t0, 6u, 0u), *pSp = a.S_PLUS + w_elem(lay, t0, 6u, 0u);
    double *pKg = a.K_GAIN ? a.K_GAIN + w_elem(lay, t0, 6u, 0u) : nullptr;
    double *pPm = a.P_MINUS + w_elem(lay, t0, 36u, (unsigned)w.e), *pPp = a.P_PLUS + w_elem(lay, t0, 36u, (unsigned)w.e);
    double *pUo = a.u_opt ? a.u_opt + w_elem(lay, t0, (unsigned)a.n_npi, (unsigned)w.k) : nullptr;
    double *pIn = a.innovations ? a.innovations + lay_scalar(t0, lay) : nullptr;
    const long d6 = dir * 6 * bp, d36 = dir * 36 * bp, dn = dir * (long)a.n_npi * bp, d1 = dir * bp;
    const double *pu = a.u + ((size_t)t0 * a.n_npi + (size_t)w.k) * a.Su + su;
    const long du = dir * (long)a.n_npi * a.Su;
    const long dx = dir * a.Sx;                           // x walks the caller's time axis, R_v the filter's (Backward*.m:27)
    const double *px = a.x + (size_t)t0 * a.Sx + sx, *pr = fixed_R ? px : a.R_series + (size_t)k_begin * a.Sx + sx;
    double *pRho = (MON && a.rho) ? a.rho + lay_scalar(k_begin, lay) : nullptr;    // filter-step order also when FLIP
    if (k_begin > 0) {           // a later time segment resumes from what the previous one stored
        w_get_vec(pSm, blk, sk_minus);
        Pm = *pPm;
    } else {                     // :100-101 for the first step (later steps: stored at the end of the step before)
        if (w.first) w_put_vec(pSm, blk, sk_minus);
        if (w.own) *pPm = Pm;
    }
    double x_nxt = *px, r_nxt = fixed_R ? 0.0 : *pr;
    double u_nxt = w_load_u(pu, w);

    for (int k = k_begin; k < k_end; k++) {
        const double Rk = fixed_R ? R_next : r_nxt, xk = x_nxt, u_in = u_nxt;
        if (k + 1 < T) {
            px += dx; pu += du;
            x_nxt = *px;
            if (!fixed_R) { pr += a.Sx; r_nxt = *pr; }
            u_nxt = w_load_u(pu, w);
        }

        double C[M];
        obs_jacobian<M>(a.mf, sk_minus, C);                              // :115
        const double xk_minus = predict_obs<M>(a.mf, sk_minus, v_bar);   // :116-119
        double innov, K[M], sk_plus[M], Pp;
        const bool valid = !is_nan(xk);                                  // :122 (wave-uniform)
        if (valid) {
            sP[lane] = Pm;
            __syncthreads();
            innov = xk - xk_minus;
            double prow[6], pcol[6];
            w_row(sP, w.i, prow);
            w_col(sP, w.j, pcol);
            const double PCt_i = w_dot6(prow, C);                        // (P C')(i): P(i,0)*C(0), fma ...
            double CP_j = C[0] * pcol[0];                                // (C P)(j): C(0)*P(0,j), fma ...
#pragma unroll
            for (int q = 1; q < M; q++) CP_j = fma(C[q], pcol[q], CP_j);
            double CPCt = w_bcast(CP_j, 0) * C[0];                       // lane 6 q holds (C P)(q)
#pragma unroll
            for (int q = 1; q < M; q++) CPCt = fma(w_bcast(CP_j, 6 * q), C[q], CPCt);
            const double den = CPCt + gamma * Rk;                        // :124 (D = 1, Hessian terms 0)
            // P(k|k-1) is bit-wise symmetric (symmetrised at :161; Ps_init by ekf_precheck), so (C P)(j) == (P C')(j)
            const double K_i = PCt_i / den, K_j = CP_j / den;
            double ikc_i[6], ikc_j[6];
#pragma unroll
            for (int q = 0; q < M; q++) {
                ikc_i[q] = ((w.i == q) ? 1.0 : 0.0) - K_i * C[q];
                ikc_j[q] = ((w.j == q) ? 1.0 : 0.0) - K_j * C[q];
            }
            const double T1 = w_dot6(ikc_i, pcol);                       // ((I - K C) P)(i,j)
            if (GEN) {
                sT[lane] = T1;
                __syncthreads();
            }
#pragma unroll
            for (int q = 0; q < M; q++) K[q] = w_bcast(K_i, q);          // lane q holds K(q)
#pragma unroll
            for (int q = 0; q < M; q++) sk_plus[q] = sk_minus[q] + K[q] * innov;   // :129
            if (GEN) {
                double t1row[6];
                w_row(sT, w.i, t1row);
                const double T2 = w_dot6(t1row, ikc_j);                  // Joseph form :127
                Pp = (T2 + (K_i * Rk) * K_j) / gamma;
            } else {
                Pp = T1 / gamma;                                         // NewCase...m:64
            }
        } else {                                                         // :130-135
            innov = 0.0;
#pragma unroll
            for (int q = 0; q < M; q++) { K[q] = 0.0; sk_plus[q] = sk_minus[q]; }
            Pp = Pm;
        }
        state_hard_margins<M>(p, sk_plus);                               // :141
        // my NPI at s(k|k): the control applied and the slope-term contribution; my entry of the Jacobian  :155-157
        double u_app, tterm;
        w_npi(p, np, u_in, sk_plus[5], u_app, tterm, a.mf.phi_ge != 0);
        const double myA = w_jac_entry(p, jc, sk_plus);
        if (GEN) Pp = (Pp + w_from(Pp, tr_b)) / 2.0;                     // :138  (P + P')/2.0
        __syncthreads();
        sP[lane] = Pp;                                                   // the symmetrised P(k|k)
        sA[lane] = myA;
        sD[lane] = np.umax - u_app;
        sTt[lane] = tterm;
        __syncthreads();

        // s(k+1|k), P(k+1|k)  :155-164
        double sk_next[M];
        {
            const double a36 = w_slope<FLIP>(sTt);
            double arow_i[6], arow_j[6], ppcol[6];
            w_jac_rows(sA, w, a36, arow_i, arow_j);
            w_col(sP, w.j, ppcol);
            const double T1 = w_dot6(arow_i, ppcol);                     // (A P+)(i,j)
            sT[lane] = T1;
            const double dot = w_dot_npi(sGa, sD);
            state_map<M, FLIP>(p, dot, sk_plus, sk_next);
            __syncthreads();
            double t1row[6];
            w_row(sT, w.i, t1row);
            const double T2 = w_dot6(t1row, arow_j);                     // (A P+ A')(i,j)
            double Pn = T2 + Qe;                                         // B = I
            if (GEN) Pn = (Pn + w_from(Pn, tr_b)) / 2.0;                 // :161
            Pm = Pn;
        }
        state_hard_margins<M>(p, sk_next);                               // :164

        // :167-169, and :100-101 of the next step (also the hand-over to a later time segment)
        const bool more = k + 1 < T;
        if (w.first) {
            w_put_vec(pSp, blk, sk_plus);
            if (pKg) w_put_vec(pKg, blk, K);
            if (pIn) *pIn = innov;
            if (more) w_put_vec(pSm + d6, blk, sk_next);
        }
        if (w.own) {
            *pPp = Pp;
            if (more) pPm[d36] = Pm;
        }
        if (pUo && w.npi) *pUo = u_app;
        pSm += d6; pSp += d6; pPm += d36; pPp += d36;
        if (pKg) pKg += d6;
        if (pUo) pUo += dn;
        if (pIn) pIn += d1;
#pragma unroll
        for (int q = 0; q < M; q++) sk_minus[q] = sk_next[q];

        if (!MON) continue;
        // innovation monitor :172-179 and the adaptive R :180-185 (the arithmetic of ekf_fwd_quad / ekf_fwd_sym; every lane holds
        // the same numbers, the LDS writes of the 64 lanes coincide)
        const int cnt = (k + 1 < L) ? (k + 1) : L;
        pos = (pos == 0) ? (L - 1) : (pos - 1);
        double *wm = winMean + pos;
        __syncthreads();
        wm[0] = innov; wm[L] = innov;
        __syncthreads();
        const double sum = qring_sum<LC, 1>(wm, L, innov);
        const double mu = sum / (double)cnt;
        const double cc2 = (innov - mu) * (innov - mu);
        const double ccn = GEN ? cc2 / (Rk + kEps) : cc2 / Rk;          // NewCase...m:108
        double *wc = winCov + pos, *wn = winCovN + pos;
        wc[0] = cc2; wc[L] = cc2;
        wn[0] = ccn; wn[L] = ccn;
        __syncthreads();
        const double sumN = qring_sum<LC, 1>(wn, L, ccn);
        if (pRho) {
            if (w.first) *pRho = sumN / (double)cnt;
            pRho += bp;
        }
        if (fixed_R) {
            if (GEN) {
                if (beta != 1.0 && valid && k < T - 1) {
                    const double sumC = qring_sum<LC, 1>(wc, L, cc2);
                    R_next = beta * Rk + (1.0 - beta) * (sumC / (double)cnt);      // :184
                } else {
                    R_next = R_v;
                }
            } else if (beta != 1.0 && valid) {                                     // NewCase...m:110-112: a running R
                const double sumC = qring_sum<LC, 1>(wc, L, cc2);
                R_next = beta * Rk + (1.0 - beta) * sumC / (double)cnt;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// backward pass: GenericExtendedKalmanFilter.m:189-230, X = pinv(P(k+1|k)) from the eks_pinv grid
// ---------------------------------------------------------------------------
struct WaveBwdIn {
    double Sp[6], Sm1[6];
    double u, Pp, Pm1, X;
    int rk;
};
// this lane's pointers at the array positions of filter step k (S+, P+, u) and k + 1 (S-, P-, X, rank word)
struct WaveBwdPtr {
    const double *Sp, *Sm1, *Pp, *Pm1, *X, *u;
    const int32_t *rk;
};
EPI_DEV void w_bwd_fetch(const WaveBwdPtr &q, unsigned blk, const WaveLane &w, WaveBwdIn &o)
{
    w_get_vec(q.Sp, blk, o.Sp);
    w_get_vec(q.Sm1, blk, o.Sm1);
    o.Pp = *q.Pp;
    o.Pm1 = *q.Pm1;
    o.X = *q.X;
    o.rk = *q.rk;
    o.u = w_load_u(q.u, w);
}

template <int FLIP>
__global__ __launch_bounds__(kWave) void eks_bwd_wave(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int M = 6;
    __shared__ double sP[kWave], sT[kWave], sA[kWave], sX[kWave], sJ[kWave], sGa[kNpi], sTt[kWave];
    if (*dense_flag) return;
    const int c = a.c0 + (int)blockIdx.x;
    if (c >= a.c0 + a.cn) return;
    const int B = a.B, T = a.T;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    const WaveLane w = w_lane(a);
    const unsigned lane = threadIdx.x;
    const int tr_b = (w.j + 6 * w.i) * 4;
    QPrm p;
    WaveNpi np;
    w_load_prm(p, np, a, B, c, w, sGa);
    const WaveJac jc = w_jac_setup<FLIP>(p, w);
    const unsigned blk = lay.blk;
    const long bp = (long)lay.bp;
    const long dir = FLIP ? 1L : -1L;                    // the smoother walks DOWN the filter's time axis
    const int lo = w.i < w.j ? w.i : w.j, hi = w.i < w.j ? w.j : w.i;

    // terminal condition :189-202
    const int tT = tpos<FLIP>(T - 1, T);
    double Ss[M], Ps;
    w_get_vec(a.S_PLUS + w_elem(lay, tT, 6u, 0u), blk, Ss);
#pragma unroll
    for (int i = 0; i < M; i++) {
        const double f = a.s_final[(size_t)i * B + c];
        if (!is_nan(f)) Ss[i] = f;
    }
    Ps = a.P_PLUS[w_elem(lay, tT, 36u, (unsigned)w.e)];
    {
        const double f = a.Ps_final[(size_t)w.e * B + c];
        if (!is_nan(f)) Ps = f;
    }
    double *pSs = a.S_SMOOTH ? a.S_SMOOTH + w_elem(lay, tT, 6u, 0u) : nullptr;
    double *pPs = a.P_SMOOTH ? a.P_SMOOTH + w_elem(lay, tT, 36u, (unsigned)w.e) : nullptr;
    double *pUs = a.u_opt_smooth ? a.u_opt_smooth + w_elem(lay, tT, (unsigned)a.n_npi, (unsigned)w.k) : nullptr;
    int32_t *pRk = a.pinv_rank ? a.pinv_rank + lay_scalar(tT, lay) : nullptr;
    if (w.first) {
        if (pSs) w_put_vec(pSs, blk, Ss);
        if (pRk) *pRk = -1;
    }
    if (pUs && w.npi) *pUs = 0.0;                        // column T is never written :95,204
    if (pPs && w.own) *pPs = Ps;
    const long d6 = dir * 6 * bp, d36 = dir * 36 * bp, d21 = dir * 21 * bp, dn = dir * (long)a.n_npi * bp, d1 = dir * bp;
    const long du = dir * (long)a.n_npi * a.Su;

    int st_guard = 0, st_cap = 0, min_rank = M;
    WaveBwdIn nxt;
    WaveBwdPtr q;
    if (T >= 2) {
        const int t = tpos<FLIP>(T - 2, T);              // step k = T - 2; step k + 1 sits at tT
        q.Sp = a.S_PLUS + w_elem(lay, t, 6u, 0u); q.Sm1 = a.S_MINUS + w_elem(lay, tT, 6u, 0u);
        q.Pp = a.P_PLUS + w_elem(lay, t, 36u, (unsigned)w.e); q.Pm1 = a.P_MINUS + w_elem(lay, tT, 36u, (unsigned)w.e);
        // X is stored packed (upper triangle, column by column): element (i, j), i <= j, at row i + j (j + 1) / 2
        q.X = a.X + w_elem(lay, tT, 21u, (unsigned)(lo + hi * (hi + 1) / 2));
        q.rk = a.rankbuf + lay_scalar(tT, lay);
        q.u = a.u + ((size_t)t * a.n_npi + (size_t)w.k) * a.Su + su;
        w_bwd_fetch(q, blk, w, nxt);
    }
    for (int k = T - 2; k >= 0; k--) {
        const WaveBwdIn cur = nxt;
        if (pSs) pSs += d6;
        if (pPs) pPs += d36;
        if (pUs) pUs += dn;
        if (pRk) pRk += d1;
        if (k > 0) {
            q.Sp += d6; q.Sm1 += d6; q.Pp += d36; q.Pm1 += d36; q.X += d21; q.rk += d1; q.u += du;
            w_bwd_fetch(q, blk, w, nxt);
        }

        // the Jacobian at S+(k) with the ORIGINAL control column :206 -- every entry by its owner
        double u_unused, tterm;
        w_npi(p, np, cur.u, cur.Sp[5], u_unused, tterm);
        const double myA = w_jac_entry(p, jc, cur.Sp);
        __syncthreads();
        sTt[lane] = tterm;
        sA[lane] = myA;
        sP[lane] = cur.Pp; sX[lane] = cur.X;
        __syncthreads();
        double J = 0.0;
        int rank = -1;
        if (cur.rk < 0) {                                                // non-finite P_MINUS guard :211-213
            st_guard = 1;
        } else {
            double pprow[6], arow_i[6], arow_j[6];
            w_row(sP, w.i, pprow);
            w_jac_rows(sA, w, w_slope<FLIP>(sTt), arow_i, arow_j);
            const double PAt = w_dot6(pprow, arow_j);                    // (P+ A')(i,j) = sum_q P+(i,q) A(j,q)
            sT[lane] = PAt;
            __syncthreads();
            double parow[6], xcol[6];
            w_row(sT, w.i, parow);
            w_col(sX, w.j, xcol);
            J = w_dot6(parow, xcol);                                     // :215
            rank = cur.rk & 0xff;
            st_cap |= (cur.rk >> 8) & 1;
            min_rank = rank < min_rank ? rank : min_rank;
        }
        // S_SMOOTH(k) = clamp(S+ + J (S_SMOOTH(k+1) - S-(k+1)))   :218-221
        double dv[M];
#pragma unroll
        for (int qq = 0; qq < M; qq++) dv[qq] = Ss[qq] - cur.Sm1[qq];
        __syncthreads();
        sJ[lane] = J;
        const double D = cur.Pm1 - Ps;                                   // D = P_MINUS(k+1) - P_SMOOTH(k+1)
        sT[lane] = D;
        __syncthreads();
        double jrow_i[6], jrow_j[6], dcol[6];
        w_row(sJ, w.i, jrow_i);
        w_row(sJ, w.j, jrow_j);
        w_col(sT, w.j, dcol);
        const double Jd_i = w_dot6(jrow_i, dv);
        const double T1 = w_dot6(jrow_i, dcol);                          // (J D)(i,j)
        sP[lane] = T1;
        __syncthreads();
        double Sn[M];
#pragma unroll
        for (int qq = 0; qq < M; qq++) Sn[qq] = cur.Sp[qq] + w_bcast(Jd_i, qq);   // lane q holds (J d)(q)
        state_hard_margins<M>(p, Sn);
        // u_opt_smooth(:, k) = the control NlinStateUpdate applies at S_SMOOTH(k)   :229
        double u_s, t_unused;
        w_npi(p, np, cur.u, Sn[5], u_s, t_unused);
        double t1row[6];
        w_row(sP, w.i, t1row);
        const double T2 = w_dot6(t1row, jrow_j);                         // (J D J')(i,j)
        double Pn = cur.Pp - T2;                                         // :223
        Pn = (Pn + w_from(Pn, tr_b)) / 2.0;                              // :226
        Ps = Pn;
#pragma unroll
        for (int qq = 0; qq < M; qq++) Ss[qq] = Sn[qq];
        if (w.first) {
            if (pSs) w_put_vec(pSs, blk, Ss);
            if (pRk) *pRk = rank;
        }
        if (pUs && w.npi) *pUs = u_s;
        if (pPs && w.own) *pPs = Ps;
    }
    if (w.first && a.status) a.status[c] = st_guard | (st_cap << 1) | (min_rank << 8);
}

// ---------------------------------------------------------------------------
// backward pass of NewCaseEKFEstimatorWithOptimalNPI (NewCase...m:115-139) in the wave shape (round 5)
// ---------------------------------------------------------------------------
// One row of X = Bm / A, i.e. of the solution of X A = Bm: A' x' = b' with b = Brow, by the operation order of mrdivide()
// (ekf_device.hpp: LAPACK dgetf2 + dgetrs -- first-max partial pivoting with predicated swaps, reciprocal scaling of the
// sub-column, column-oriented triangular solves).  The rows of X are independent given the factorisation, so lane c < 6 of the
// wavefront factors A' for itself and solves for row c: ~1/4 of the instructions of the whole mrdivide per lane.
template <int M>
EPI_DEV void mrdivide_row(const double (&Brow)[M], const double (&A)[M * M], double (&Xrow)[M])
{
    double Mt[M * M], Y[M];
#pragma unroll
    for (int j = 0; j < M; j++) {
#pragma unroll
        for (int i = 0; i < M; i++) Mt[IXM(i, j)] = A[IXM(j, i)];
        Y[j] = Brow[j];
    }
#pragma unroll
    for (int j = 0; j < M; j++) {
        int piv = j;
        double best = fabs(Mt[IXM(j, j)]);
#pragma unroll
        for (int i = j + 1; i < M; i++) {
            double vv = fabs(Mt[IXM(i, j)]);
            if (vv > best) { best = vv; piv = i; }
        }
        double pval = Mt[IXM(j, j)];
#pragma unroll
        for (int i = j + 1; i < M; i++) pval = (piv == i) ? Mt[IXM(i, j)] : pval;
        const bool nz = (pval != 0.0);
#pragma unroll
        for (int i = j + 1; i < M; i++) {
            const bool sw = (piv == i);
            if (__builtin_amdgcn_ballot_w64(sw) == 0ull) continue;
#pragma unroll
            for (int c = 0; c < M; c++) {
                double mj = Mt[IXM(j, c)], mi = Mt[IXM(i, c)];
                Mt[IXM(j, c)] = (sw && nz) ? mi : mj;
                Mt[IXM(i, c)] = (sw && nz) ? mj : mi;
            }
            double yj = Y[j], yi = Y[i];
            Y[j] = sw ? yi : yj;
            Y[i] = sw ? yj : yi;
        }
        if (nz) {
            if (fabs(Mt[IXM(j, j)]) >= 2.2250738585072014e-308) {
                const double r = 1.0 / Mt[IXM(j, j)];
#pragma unroll
                for (int i = j + 1; i < M; i++) Mt[IXM(i, j)] = Mt[IXM(i, j)] * r;
            } else {
#pragma unroll
                for (int i = j + 1; i < M; i++) Mt[IXM(i, j)] = Mt[IXM(i, j)] / Mt[IXM(j, j)];
            }
        }
#pragma unroll
        for (int c = j + 1; c < M; c++)
#pragma unroll
            for (int i = j + 1; i < M; i++) Mt[IXM(i, c)] = Mt[IXM(i, c)] - Mt[IXM(i, j)] * Mt[IXM(j, c)];
    }
#pragma unroll
    for (int k = 0; k < M; k++) {
        if (Y[k] != 0.0) {
#pragma unroll
            for (int i = k + 1; i < M; i++) Y[i] = Y[i] - Y[k] * Mt[IXM(i, k)];
        }
    }
#pragma unroll
    for (int k = M - 1; k >= 0; k--) {
        if (Y[k] != 0.0) {
            Y[k] = Y[k] / Mt[IXM(k, k)];
#pragma unroll
            for (int i = 0; i < k; i++) Y[i] = Y[i] - Y[k] * Mt[IXM(i, k)];
        }
    }
#pragma unroll
    for (int i = 0; i < M; i++) Xrow[i] = Y[i];
}

// Same lane mapping as eks_bwd_wave.  What differs (NewCase...m:115-139): the terminal condition's cross-product sub-assignment
// (:125-127), the gain J = (P+ A') / P- by mrdivide -- the six lanes c < 6 each solve for row c of J, see mrdivide_row --, no
// symmetrisation, no u_opt_smooth, no rank words.
__global__ __launch_bounds__(kWave) void eks_bwd_wave_nc(const KArgs a)
{
    constexpr int M = 6;
    constexpr int FLIP = 0;
    __shared__ double sP[kWave], sT[kWave], sA[kWave], sM[kWave], sJ[kWave], sGa[kNpi], sTt[kWave];
    const int c = a.c0 + (int)blockIdx.x;
    if (c >= a.c0 + a.cn) return;
    const int B = a.B, T = a.T;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    const WaveLane w = w_lane(a);
    const unsigned lane = threadIdx.x;
    QPrm p;
    WaveNpi np;
    w_load_prm(p, np, a, B, c, w, sGa);
    const WaveJac jc = w_jac_setup<FLIP>(p, w);
    const unsigned blk = lay.blk;
    const long bp = (long)lay.bp;
    const long dir = -1L;                                // the smoother walks DOWN the filter's time axis

    // terminal condition :117-127
    const int tT = T - 1;
    double Ss[M], Ps;
    w_get_vec(a.S_PLUS + w_elem(lay, tT, 6u, 0u), blk, Ss);
#pragma unroll
    for (int i = 0; i < M; i++) {
        const double f = a.s_final[(size_t)i * B + c];
        if (!is_nan(f)) Ss[i] = f;
    }
    Ps = a.P_PLUS[w_elem(lay, tT, 36u, (unsigned)w.e)];
    {
        // P_SMOOTH(row, col, T) = Ps_final(row, col) for the rows and columns that hold any non-NaN entry
        const double f = a.Ps_final[(size_t)w.e * B + c];
        sT[lane] = (w.own && !is_nan(f)) ? 1.0 : 0.0;
        __syncthreads();
        double fr[6], fc[6];
        w_row(sT, w.i, fr);
        w_col(sT, w.j, fc);
        bool rows = false, cols = false;
#pragma unroll
        for (int q = 0; q < M; q++) { rows = rows || fr[q] != 0.0; cols = cols || fc[q] != 0.0; }
        if (rows && cols) Ps = f;
        __syncthreads();
    }
    double *pSs = a.S_SMOOTH ? a.S_SMOOTH + w_elem(lay, tT, 6u, 0u) : nullptr;
    double *pPs = a.P_SMOOTH ? a.P_SMOOTH + w_elem(lay, tT, 36u, (unsigned)w.e) : nullptr;
    int32_t *pRk = a.pinv_rank ? a.pinv_rank + lay_scalar(tT, lay) : nullptr;
    if (w.first) {
        if (pSs) w_put_vec(pSs, blk, Ss);
        if (pRk) *pRk = -1;
    }
    if (pPs && w.own) *pPs = Ps;
    const long d6 = dir * 6 * bp, d36 = dir * 36 * bp, d1 = dir * bp;
    const long du = dir * (long)a.n_npi * a.Su;

    WaveBwdIn nxt;
    WaveBwdPtr q;
    auto fetch = [&](WaveBwdIn &o) __attribute__((always_inline)) {
        w_get_vec(q.Sp, blk, o.Sp);
        w_get_vec(q.Sm1, blk, o.Sm1);
        o.Pp = *q.Pp;
        o.Pm1 = *q.Pm1;
        o.u = w_load_u(q.u, w);
    };
    if (T >= 2) {
        const int t = T - 2;
        q.Sp = a.S_PLUS + w_elem(lay, t, 6u, 0u); q.Sm1 = a.S_MINUS + w_elem(lay, tT, 6u, 0u);
        q.Pp = a.P_PLUS + w_elem(lay, t, 36u, (unsigned)w.e); q.Pm1 = a.P_MINUS + w_elem(lay, tT, 36u, (unsigned)w.e);
        q.u = a.u + ((size_t)t * a.n_npi + (size_t)w.k) * a.Su + su;
        fetch(nxt);
    }
    for (int k = T - 2; k >= 0; k--) {
        const WaveBwdIn cur = nxt;
        if (pSs) pSs += d6;
        if (pPs) pPs += d36;
        if (pRk) pRk += d1;
        if (k > 0) {
            q.Sp += d6; q.Sm1 += d6; q.Pp += d36; q.Pm1 += d36; q.u += du;
            fetch(nxt);
        }
        double u_unused, tterm;
        w_npi(p, np, cur.u, cur.Sp[5], u_unused, tterm, a.mf.phi_ge != 0);
        const double myA = w_jac_entry(p, jc, cur.Sp);
        __syncthreads();
        sTt[lane] = tterm;
        sA[lane] = myA;
        sP[lane] = cur.Pp; sM[lane] = cur.Pm1;
        __syncthreads();
        {
            double pprow[6], arow_i[6], arow_j[6];
            w_row(sP, w.i, pprow);
            w_jac_rows(sA, w, w_slope<FLIP>(sTt), arow_i, arow_j);
            sT[lane] = w_dot6(pprow, arow_j);                            // (P+ A')(i,j)
        }
        __syncthreads();
        if (lane < M) {                                                  // J = (P+ A') / P-(k+1)  :132, row `lane`
            double Am[M * M], brow[M], xrow[M];
#pragma unroll
            for (int e = 0; e < M * M; e++) Am[e] = sM[e];
            w_row(sT, (int)lane, brow);
            mrdivide_row<M>(brow, Am, xrow);
#pragma unroll
            for (int qq = 0; qq < M; qq++) sJ[lane + 6 * qq] = xrow[qq];
        }
        // S_SMOOTH(k) = clamp(S+ + J (S_SMOOTH(k+1) - S-(k+1)))   :134-135
        double dv[M];
#pragma unroll
        for (int qq = 0; qq < M; qq++) dv[qq] = Ss[qq] - cur.Sm1[qq];
        sP[lane] = cur.Pm1 - Ps;                                         // D = P_MINUS(k+1) - P_SMOOTH(k+1)
        __syncthreads();
        double jrow_i[6], jrow_j[6], dcol[6];
        w_row(sJ, w.i, jrow_i);
        w_row(sJ, w.j, jrow_j);
        w_col(sP, w.j, dcol);
        const double Jd_i = w_dot6(jrow_i, dv);
        const double T1 = w_dot6(jrow_i, dcol);                          // (J D)(i,j)
        __syncthreads();
        sT[lane] = T1;
        __syncthreads();
        double Sn[M];
#pragma unroll
        for (int qq = 0; qq < M; qq++) Sn[qq] = cur.Sp[qq] + w_bcast(Jd_i, qq);   // lane q holds (J d)(q)
        state_hard_margins<M>(p, Sn);
        double t1row[6];
        w_row(sT, w.i, t1row);
        const double T2 = w_dot6(t1row, jrow_j);                         // (J D J')(i,j)
        Ps = cur.Pp - T2;                                                // :137, not symmetrised
#pragma unroll
        for (int qq = 0; qq < M; qq++) Ss[qq] = Sn[qq];
        if (w.first) {
            if (pSs) w_put_vec(pSs, blk, Ss);
            if (pRk) *pRk = -1;
        }
        if (pPs && w.own) *pPs = Ps;
    }
    if (w.first && a.status) a.status[c] = 0 | (0 << 1) | (M << 8);
}

// =====================================================================================================================
// 3-state generic models (SIAlphaModelEKF, SIAlphaModelBackwardEKF): SEVEN chains per wavefront, nine lanes each
// =====================================================================================================================
// Lane = 9 g + e: group g (0..6) is one chain, e = i + 3 j owns element (i, j) of its 3 x 3 matrices; lane 63 idles.  The
// chain's state, gain and C are replicated in the nine lanes of its group (no longer wave-uniform, so single-lane values
// travel by ds_bpermute from the owner instead of v_readlane); lane e also owns NPI e, and lanes e < 3 NPI 9 + e as well.
// The 3-state models have no free controls (SIAlphaModelEKF.m:39 returns u as it came), so the NPI lanes only form
// u_max(k) - u(k) for the fma chain of the alpha map.  Same dense, k-ascending products as above: bit for bit the one-lane
// kernels' and the oracle's results.
constexpr int kW3G = 7, kW3E = 9;

struct Wave3Lane {
    int g, e, i, j, c;         // group, element, row, column, chain
    int base;                  // first lane of the group
    bool own, lead, live;      // owns an element of a real chain / first lane of such a group / chain exists
};
EPI_DEV Wave3Lane w3_lane(const KArgs &a)
{
    Wave3Lane w;
    const int lane = (int)threadIdx.x;
    w.g = lane < kW3G * kW3E ? lane / kW3E : kW3G - 1;
    w.e = lane < kW3G * kW3E ? lane - w.g * kW3E : 0;
    w.j = w.e / 3; w.i = w.e - 3 * w.j;
    w.base = w.g * kW3E;
    const int c = a.c0 + (int)blockIdx.x * kW3G + w.g;
    w.live = c < a.c0 + a.cn;
    w.c = w.live ? c : a.c0 + a.cn - 1;          // idle groups mirror the last chain: everything they compute is dropped
    w.own = w.live && lane < kW3G * kW3E;
    w.lead = w.own && w.e == 0;
    return w;
}
EPI_DEV void w3_row(const double *t, int r, double (&o)[3]) { o[0] = t[r]; o[1] = t[r + 3]; o[2] = t[r + 6]; }
EPI_DEV void w3_col(const double *t, int c, double (&o)[3]) { o[0] = t[3 * c]; o[1] = t[3 * c + 1]; o[2] = t[3 * c + 2]; }
EPI_DEV double w3_dot(const double (&x)[3], const double (&y)[3]) { return fma(x[2], y[2], fma(x[1], y[1], x[0] * y[0])); }
EPI_DEV void w3_put_vec(double *p, unsigned blk, const double (&v)[3])
{
#pragma unroll
    for (int q = 0; q < 3; q++) p[(size_t)q * blk] = v[q];
}
EPI_DEV void w3_get_vec(const double *p, unsigned blk, double (&v)[3])
{
#pragma unroll
    for (int q = 0; q < 3; q++) v[q] = p[(size_t)q * blk];
}
// my entry of the 3 x 3 Jacobian (SIAlphaModelEKF.m:62-76, Backward...m:83-97): jacobian_entries<3>'s expressions
struct Wave3Jac { bool x0, x1, y0, y1, s2, neg, base1, cst; double c; };
template <int FLIP>
EPI_DEV Wave3Jac w3_jac_setup(const QPrm &p, const Wave3Lane &w)
{
    struct E { unsigned char kind, x, y, neg, base1; };       // kind 0 zero, 1 S1, 2 S2, 3 CONST (see WaveJac)
    const E tab[9] = {/*(0,0)*/ {1, 2, 1, 1, 1}, /*(1,0)*/ {1, 1, 2, 0, 0}, /*(2,0)*/ {0, 0, 0, 0, 0},
                      /*(0,1)*/ {1, 2, 0, 1, 0}, /*(1,1)*/ {2, 0, 0, 0, 1}, /*(2,1)*/ {0, 0, 0, 0, 0},
                      /*(0,2)*/ {1, 0, 1, 1, 0}, /*(1,2)*/ {1, 0, 1, 0, 0}, /*(2,2)*/ {3, 0, 0, 1, 1}};
    E me = tab[0];
#pragma unroll
    for (int q = 1; q < 9; q++)
        if (w.e == q) me = tab[q];
    Wave3Jac j;
    j.x0 = me.x == 0; j.x1 = me.x == 1; j.y0 = me.y == 0; j.y1 = me.y == 1;
    j.s2 = me.kind == 2;
    j.neg = (me.neg != 0) != (FLIP != 0);
    j.base1 = me.base1 != 0;
    j.cst = me.kind == 0 || me.kind == 3;
    const double dg = p.dt * p.gamma;
    j.c = me.kind == 3 ? (j.neg ? 1.0 - dg : 1.0 + dg) : 0.0;
    return j;
}
EPI_DEV double w3_jac_entry(const QPrm &p, const Wave3Jac &j, const double (&s)[3])
{
    const double w2 = p.dt * (s[0] * s[2] - p.beta);
    const double X = j.x0 ? s[0] : (j.x1 ? s[1] : s[2]);
    const double Y = j.y0 ? s[0] : (j.y1 ? s[1] : s[2]);
    double v = p.dt * X * Y;
    v = j.s2 ? w2 : v;
    v = j.neg ? -v : v;
    const double r = j.base1 ? 1.0 + v : v;
    return j.cst ? j.c : r;
}
struct Wave3Npi { double umax0, umax1; const double *pu0, *pu1; bool two; };   // NPI e, and NPI 9 + e for e < 3
EPI_DEV void w3_load_prm(QPrm &p, Wave3Npi &n, const KArgs &a, int B, const Wave3Lane &w, double *sGa)
{
    const int c = w.c;
    auto g = [&](int f) { return a.prm[(size_t)f * B + c]; };
    p.dt = g(EPI_PRM_DT); p.beta = g(EPI_PRM_BETA); p.gamma = g(EPI_PRM_GAMMA);
    p.sigma = g(EPI_PRM_SIGMA); p.b = g(EPI_PRM_B); p.epsilon = g(EPI_PRM_EPSILON);
    p.slo = a.mf.lo_is_zero ? 0.0 : g(EPI_PRM_S_MIN);
    p.ilo = a.mf.lo_is_zero ? 0.0 : g(EPI_PRM_I_MIN);
    p.alpha_min = g(EPI_PRM_ALPHA_MIN); p.alpha_max = g(EPI_PRM_ALPHA_MAX);
    n.two = w.e < 3;
    n.umax0 = g(EPI_PRM_U_MAX + w.e);
    n.umax1 = n.two ? g(EPI_PRM_U_MAX + 9 + w.e) : 0.0;
    sGa[w.g * kNpi + w.e] = p.gamma * g(EPI_PRM_A + w.e);           // gamma * a(k): the constant factors of the alpha map's fma chain
    if (n.two) sGa[w.g * kNpi + 9 + w.e] = p.gamma * g(EPI_PRM_A + 9 + w.e);
}
// u(k, t): rows beyond n_npi read as 0.0 (the padding of load_u)
EPI_DEV void w3_load_u(const KArgs &a, const Wave3Npi &n, const Wave3Lane &w, double &u0, double &u1)
{
    u0 = (w.e < a.n_npi) ? *n.pu0 : 0.0;
    u1 = (n.two && 9 + w.e < a.n_npi) ? *n.pu1 : 0.0;
}

template <int FLIP>
__global__ __launch_bounds__(kWave) void ekf_fwd_wave3(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int M = 3;
    __shared__ double sP[kWave], sT[kWave], sA[kWave], sGa[kW3G * kNpi], sD[kW3G * kNpi];
    if (*dense_flag) return;
    const Wave3Lane w = w3_lane(a);
    const int B = a.B, T = a.T, c = w.c;
    const int sx = a.x_series ? a.x_series[c] : c;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    const unsigned lane = threadIdx.x;
    double *tP = sP + w.base, *tT = sT + w.base, *tA = sA + w.base;
    const double *tGa = sGa + w.g * kNpi, *tD = sD + w.g * kNpi;
    const int tr_b = (w.base + w.j + 3 * w.i) * 4;

    QPrm p;
    Wave3Npi np;
    w3_load_prm(p, np, a, B, w, sGa);
    const Wave3Jac jc = w3_jac_setup<FLIP>(p, w);
    const double v_bar = a.prm[(size_t)EPI_PRM_V_BAR * B + c];
    const double gamma = a.prm[(size_t)EPI_PRM_GAMMA_EKF * B + c];

    double sk_minus[M];
#pragma unroll
    for (int i = 0; i < M; i++) sk_minus[i] = a.s_init[(size_t)i * B + c];
    double Pm = a.Ps_init[(size_t)w.e * B + c];
    const double Qe = a.Q[(size_t)w.e * B + c];

    const int k_begin = a.k_begin, k_end = (a.k_end > 0 && a.k_end < T) ? a.k_end : T;
    const int t0 = tpos<FLIP>(k_begin, T);
    const long dir = FLIP ? -1L : 1L;
    const long bp = (long)lay.bp;
    const unsigned blk = lay.blk;
    double *pSm = a.S_MINUS ? a.S_MINUS + w_elem(lay, t0, 3u, 0u) : nullptr, *pSp = a.S_PLUS + w_elem(lay, t0, 3u, 0u);
    double *pKg = a.K_GAIN ? a.K_GAIN + w_elem(lay, t0, 3u, 0u) : nullptr;
    double *pPm = a.P_MINUS + w_elem(lay, t0, 9u, (unsigned)w.e), *pPp = a.P_PLUS + w_elem(lay, t0, 9u, (unsigned)w.e);
    double *pUo0 = a.u_opt ? a.u_opt + w_elem(lay, t0, (unsigned)a.n_npi, (unsigned)w.e) : nullptr;
    double *pUo1 = a.u_opt ? a.u_opt + w_elem(lay, t0, (unsigned)a.n_npi, (unsigned)(9 + w.e)) : nullptr;
    double *pIn = a.innovations ? a.innovations + lay_scalar(t0, lay) : nullptr;
    const long d3 = dir * 3 * bp, d9 = dir * 9 * bp, dn = dir * (long)a.n_npi * bp, d1 = dir * bp;
    np.pu0 = a.u + ((size_t)t0 * a.n_npi + (size_t)(w.e < a.n_npi ? w.e : 0)) * a.Su + su;
    np.pu1 = a.u + ((size_t)t0 * a.n_npi + (size_t)(9 + w.e < a.n_npi ? 9 + w.e : 0)) * a.Su + su;
    const long du = dir * (long)a.n_npi * a.Su;
    const long dx = dir * a.Sx;
    const double *px = a.x + (size_t)t0 * a.Sx + sx, *pr = a.R_series + (size_t)k_begin * a.Sx + sx;
    const bool st0 = w.e < a.n_npi && w.own, st1 = np.two && 9 + w.e < a.n_npi && w.own;
    if (k_begin > 0) {
        w3_get_vec(pSm, blk, sk_minus);
        Pm = *pPm;
    } else {
        if (w.lead && pSm) w3_put_vec(pSm, blk, sk_minus);
        if (w.own) *pPm = Pm;
    }
    double x_nxt = *px, r_nxt = *pr, u0_nxt, u1_nxt;
    w3_load_u(a, np, w, u0_nxt, u1_nxt);

    for (int k = k_begin; k < k_end; k++) {
        const double Rk = r_nxt, xk = x_nxt, u0 = u0_nxt, u1 = u1_nxt;
        if (k + 1 < T) {
            px += dx; pr += a.Sx; np.pu0 += du; np.pu1 += du;
            x_nxt = *px; r_nxt = *pr;
            w3_load_u(a, np, w, u0_nxt, u1_nxt);
        }
        double C[M];
        obs_jacobian<M>(a.mf, sk_minus, C);                              // :115
        const double xk_minus = predict_obs<M>(a.mf, sk_minus, v_bar);   // :116-119
        double innov, K[M], sk_plus[M], Pp;
        const bool valid = !is_nan(xk);                                  // :122 (per chain: the groups of a wave may differ)
        // the LDS tiles are exchanged by all groups together, whatever their chains' observations say
        tP[w.e] = Pm;
        __syncthreads();
        {
            double prow[3], pcol[3];
            w3_row(tP, w.i, prow);
            w3_col(tP, w.j, pcol);
            const double PCt_i = w3_dot(prow, C);                        // (P C')(i)
            const double CP_j = fma(C[2], pcol[2], fma(C[1], pcol[1], C[0] * pcol[0]));   // (C P)(j)
            double CP[3];
#pragma unroll
            for (int q = 0; q < M; q++) CP[q] = w_from(CP_j, (w.base + 3 * q) * 4);     // lane (0, q) of my group holds (C P)(q)
            const double CPCt = fma(CP[2], C[2], fma(CP[1], C[1], CP[0] * C[0]));
            const double den = CPCt + gamma * Rk;                        // :124
            const double K_i = PCt_i / den, K_j = CP_j / den;            // P(k|k-1) bit-wise symmetric: (C P)(j) == (P C')(j)
            double ikc_i[3], ikc_j[3];
#pragma unroll
            for (int q = 0; q < M; q++) {
                ikc_i[q] = ((w.i == q) ? 1.0 : 0.0) - K_i * C[q];
                ikc_j[q] = ((w.j == q) ? 1.0 : 0.0) - K_j * C[q];
            }
            const double T1 = w3_dot(ikc_i, pcol);                       // ((I - K C) P)(i,j)
            tT[w.e] = T1;
            __syncthreads();
            double Kv[3];
#pragma unroll
            for (int q = 0; q < M; q++) Kv[q] = w_from(K_i, (w.base + q) * 4);          // lane (q, 0) holds K(q)
            double t1row[3];
            w3_row(tT, w.i, t1row);
            const double T2 = w3_dot(t1row, ikc_j);                      // Joseph form :127
            const double Pv = (T2 + (K_i * Rk) * K_j) / gamma;
            innov = valid ? xk - xk_minus : 0.0;
            Pp = valid ? Pv : Pm;                                        // :130-135 when the observation is missing
#pragma unroll
            for (int q = 0; q < M; q++) {
                K[q] = valid ? Kv[q] : 0.0;
                sk_plus[q] = valid ? sk_minus[q] + Kv[q] * innov : sk_minus[q];        // :129
            }
        }
        state_hard_margins<M>(p, sk_plus);                               // :141
        const double myA = w3_jac_entry(p, jc, sk_plus);
        Pp = (Pp + w_from(Pp, tr_b)) / 2.0;                              // :138
        __syncthreads();
        tP[w.e] = Pp;
        tA[w.e] = myA;
        sD[w.g * kNpi + w.e] = np.umax0 - u0;                            // u_opt == u for the 3-state models
        if (np.two) sD[w.g * kNpi + 9 + w.e] = np.umax1 - u1;
        __syncthreads();

        double sk_next[M];
        {
            double arow_i[3], arow_j[3], ppcol[3];
            w3_row(tA, w.i, arow_i);
            w3_row(tA, w.j, arow_j);
            w3_col(tP, w.j, ppcol);
            const double T1 = w3_dot(arow_i, ppcol);                     // (A P+)(i,j)
            tT[w.e] = T1;
            const double dot = w_dot_npi(tGa, tD);
            state_map<M, FLIP>(p, dot, sk_plus, sk_next);
            __syncthreads();
            double t1row[3];
            w3_row(tT, w.i, t1row);
            double Pn = w3_dot(t1row, arow_j) + Qe;                      // (A P+ A')(i,j) + Q
            Pn = (Pn + w_from(Pn, tr_b)) / 2.0;                          // :161
            Pm = Pn;
        }
        state_hard_margins<M>(p, sk_next);                               // :164

        const bool more = k + 1 < T;
        if (w.lead) {
            w3_put_vec(pSp, blk, sk_plus);
            if (pKg) w3_put_vec(pKg, blk, K);
            if (pIn) *pIn = innov;
            if (more && pSm) w3_put_vec(pSm + d3, blk, sk_next);
        }
        if (w.own) {
            *pPp = Pp;
            if (more) pPm[d9] = Pm;
        }
        if (pUo0) {
            if (st0) *pUo0 = u0;
            if (st1) *pUo1 = u1;
            pUo0 += dn; pUo1 += dn;
        }
        if (pSm) pSm += d3;
        pSp += d3; pPm += d9; pPp += d9;
        if (pKg) pKg += d3;
        if (pIn) pIn += d1;
#pragma unroll
        for (int q = 0; q < M; q++) sk_minus[q] = sk_next[q];
    }
}

struct Wave3BwdIn { double Sp[3], Sm1[3], u0, u1, Pp, Pm1, X; int rk; };
struct Wave3BwdPtr { const double *Sp, *Sm1, *Pp, *Pm1, *X; const int32_t *rk; };

template <int FLIP>
__global__ __launch_bounds__(kWave) void eks_bwd_wave3(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int M = 3;
    __shared__ double sP[kWave], sT[kWave], sA[kWave], sX[kWave], sJ[kWave], sGa[kW3G * kNpi];
    if (*dense_flag) return;
    const Wave3Lane w = w3_lane(a);
    const int B = a.B, T = a.T, c = w.c;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    double *tP = sP + w.base, *tT = sT + w.base, *tA = sA + w.base, *tX = sX + w.base, *tJ = sJ + w.base;
    const int tr_b = (w.base + w.j + 3 * w.i) * 4;
    QPrm p;
    Wave3Npi np;
    w3_load_prm(p, np, a, B, w, sGa);
    const Wave3Jac jc = w3_jac_setup<FLIP>(p, w);
    const unsigned blk = lay.blk;
    const long bp = (long)lay.bp;
    const long dir = FLIP ? 1L : -1L;
    const int lo = w.i < w.j ? w.i : w.j, hi = w.i < w.j ? w.j : w.i;

    const int tT_ = tpos<FLIP>(T - 1, T);
    double Ss[M], Ps;
    w3_get_vec(a.S_PLUS + w_elem(lay, tT_, 3u, 0u), blk, Ss);
#pragma unroll
    for (int i = 0; i < M; i++) {
        const double f = a.s_final[(size_t)i * B + c];
        if (!is_nan(f)) Ss[i] = f;
    }
    Ps = a.P_PLUS[w_elem(lay, tT_, 9u, (unsigned)w.e)];
    {
        const double f = a.Ps_final[(size_t)w.e * B + c];
        if (!is_nan(f)) Ps = f;
    }
    double *pSs = a.S_SMOOTH ? a.S_SMOOTH + w_elem(lay, tT_, 3u, 0u) : nullptr;
    double *pPs = a.P_SMOOTH ? a.P_SMOOTH + w_elem(lay, tT_, 9u, (unsigned)w.e) : nullptr;
    double *pUs0 = a.u_opt_smooth ? a.u_opt_smooth + w_elem(lay, tT_, (unsigned)a.n_npi, (unsigned)w.e) : nullptr;
    double *pUs1 = a.u_opt_smooth ? a.u_opt_smooth + w_elem(lay, tT_, (unsigned)a.n_npi, (unsigned)(9 + w.e)) : nullptr;
    int32_t *pRk = a.pinv_rank ? a.pinv_rank + lay_scalar(tT_, lay) : nullptr;
    const bool st0 = w.e < a.n_npi && w.own, st1 = np.two && 9 + w.e < a.n_npi && w.own;
    if (w.lead) {
        if (pSs) w3_put_vec(pSs, blk, Ss);
        if (pRk) *pRk = -1;
    }
    if (pUs0) {
        if (st0) *pUs0 = 0.0;                            // column T is never written :95,204
        if (st1) *pUs1 = 0.0;
    }
    if (pPs && w.own) *pPs = Ps;
    const long d3 = dir * 3 * bp, d9 = dir * 9 * bp, d6 = dir * 6 * bp, dn = dir * (long)a.n_npi * bp, d1 = dir * bp;
    const long du = dir * (long)a.n_npi * a.Su;

    int st_guard = 0, st_cap = 0, min_rank = M;
    Wave3BwdIn nxt;
    Wave3BwdPtr q;
    auto fetch = [&](Wave3BwdIn &o) {
        w3_get_vec(q.Sp, blk, o.Sp);
        w3_get_vec(q.Sm1, blk, o.Sm1);
        o.Pp = *q.Pp; o.Pm1 = *q.Pm1; o.X = *q.X; o.rk = *q.rk;
        w3_load_u(a, np, w, o.u0, o.u1);
    };
    if (T >= 2) {
        const int t = tpos<FLIP>(T - 2, T);
        q.Sp = a.S_PLUS + w_elem(lay, t, 3u, 0u);
        // fp32-storage runs keep no S_MINUS; this shape is fp64 only, S_MINUS always exists
        q.Sm1 = a.S_MINUS + w_elem(lay, tT_, 3u, 0u);
        q.Pp = a.P_PLUS + w_elem(lay, t, 9u, (unsigned)w.e); q.Pm1 = a.P_MINUS + w_elem(lay, tT_, 9u, (unsigned)w.e);
        q.X = a.X + w_elem(lay, tT_, 6u, (unsigned)(lo + hi * (hi + 1) / 2));
        q.rk = a.rankbuf + lay_scalar(tT_, lay);
        np.pu0 = a.u + ((size_t)t * a.n_npi + (size_t)(w.e < a.n_npi ? w.e : 0)) * a.Su + su;
        np.pu1 = a.u + ((size_t)t * a.n_npi + (size_t)(9 + w.e < a.n_npi ? 9 + w.e : 0)) * a.Su + su;
        fetch(nxt);
    }
    for (int k = T - 2; k >= 0; k--) {
        const Wave3BwdIn cur = nxt;
        if (pSs) pSs += d3;
        if (pPs) pPs += d9;
        if (pUs0) { pUs0 += dn; pUs1 += dn; }
        if (pRk) pRk += d1;
        if (k > 0) {
            q.Sp += d3; q.Sm1 += d3; q.Pp += d9; q.Pm1 += d9; q.X += d6; q.rk += d1; np.pu0 += du; np.pu1 += du;
            fetch(nxt);
        }
        const double myA = w3_jac_entry(p, jc, cur.Sp);                  // :206 (the 3 x 3 Jacobian does not depend on u)
        __syncthreads();
        tA[w.e] = myA;
        tP[w.e] = cur.Pp; tX[w.e] = cur.X;
        __syncthreads();
        const bool guard = cur.rk < 0;                                   // non-finite P_MINUS guard :211-213 (per chain)
        double J;
        {
            double pprow[3], arow_j[3];
            w3_row(tP, w.i, pprow);
            w3_row(tA, w.j, arow_j);
            const double PAt = w3_dot(pprow, arow_j);                    // (P+ A')(i,j)
            tT[w.e] = PAt;
            __syncthreads();
            double parow[3], xcol[3];
            w3_row(tT, w.i, parow);
            w3_col(tX, w.j, xcol);
            J = guard ? 0.0 : w3_dot(parow, xcol);                       // :215
        }
        int rank = -1;
        if (guard) st_guard = 1;
        else {
            rank = cur.rk & 0xff;
            st_cap |= (cur.rk >> 8) & 1;
            min_rank = rank < min_rank ? rank : min_rank;
        }
        double dv[M];
#pragma unroll
        for (int qq = 0; qq < M; qq++) dv[qq] = Ss[qq] - cur.Sm1[qq];
        __syncthreads();
        tJ[w.e] = J;
        tT[w.e] = cur.Pm1 - Ps;                                          // D = P_MINUS(k+1) - P_SMOOTH(k+1)
        __syncthreads();
        double jrow_i[3], jrow_j[3], dcol[3];
        w3_row(tJ, w.i, jrow_i);
        w3_row(tJ, w.j, jrow_j);
        w3_col(tT, w.j, dcol);
        const double Jd_i = w3_dot(jrow_i, dv);
        const double T1 = w3_dot(jrow_i, dcol);                          // (J D)(i,j)
        tP[w.e] = T1;
        __syncthreads();
        double Sn[M];
#pragma unroll
        for (int qq = 0; qq < M; qq++) Sn[qq] = cur.Sp[qq] + w_from(Jd_i, (w.base + qq) * 4);   // lane (q, 0) holds (J d)(q)
        state_hard_margins<M>(p, Sn);                                    // :218-221
        double t1row[3];
        w3_row(tP, w.i, t1row);
        double Pn = cur.Pp - w3_dot(t1row, jrow_j);                      // :223
        Pn = (Pn + w_from(Pn, tr_b)) / 2.0;                              // :226
        Ps = Pn;
#pragma unroll
        for (int qq = 0; qq < M; qq++) Ss[qq] = Sn[qq];
        if (w.lead) {
            if (pSs) w3_put_vec(pSs, blk, Ss);
            if (pRk) *pRk = rank;
        }
        if (pUs0) {                                                      // :229 -- the 3-state NlinStateUpdate returns u as it came
            if (st0) *pUs0 = cur.u0;
            if (st1) *pUs1 = cur.u1;
        }
        if (pPs && w.own) *pPs = Ps;
    }
    if (w.lead && a.status) a.status[c] = st_guard | (st_cap << 1) | (min_rank << 8);
}
