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
//   * the step's inputs and, in the smoother, the stored forward quantities are requested one step ahead; every array
//     slice is addressed through a buffer descriptor with one per-lane byte offset per array kind.
#pragma once
// (included inside namespace epi, like ekf_sym.hpp / ekf_quad.hpp)

constexpr int kWE = 36;     // lanes that own a matrix element

struct WaveLane {
    int e, i, j;            // element, row, column (lanes >= 36 mirror element 0: their results are never stored)
    int k;                  // NPI owned (lanes >= 12 mirror NPI 0: never stored)
    bool own, npi;
    unsigned v36, v6, vn, v21;   // byte offsets of this lane's row in a 36- / 6- / n_npi- / 21-row time slice (loads)
    unsigned v1i;                  // one-row int32 arrays (loads)
    unsigned s36, s6, sn, s1, s1i; // the same for stores: out of range for lanes that own nothing of the array
    unsigned rowb;
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
EPI_DEV rsrc_t w_slice(const void *p, int t, unsigned rows, const Lay &l, unsigned elem = 8u)
{
    return mk_rsrc((const char *)p + (size_t)t * rows * l.bp * elem, rows * l.bp * elem);
}
// The time slices of the arrays a kernel walks, as RUNNING byte offsets (one 64-bit add per kind and step instead of a
// 64-bit multiply per array and step): slice t of an array with `rows` rows starts at t * rows * bp * elem.
struct WaveWalk {
    long o36, o21, o6, on, o1, o1i, ou;      // offsets of the current slice for 36- / 21- / 6- / n_npi-row fp64 arrays, one-row fp64 / int32 arrays, the control series
    long d36, d21, d6, dn, d1, d1i, du;      // what one step adds (negative when the walk runs down the caller's time axis)
    unsigned z36, z21, z6, zn, z1, z1i, zu;  // slice sizes in bytes
};
EPI_DEV WaveWalk w_walk(const KArgs &a, const Lay &l, int t0, int dir)
{
    WaveWalk k;
    const long bp = (long)l.bp;
    k.z36 = (unsigned)(36 * bp * 8); k.z21 = (unsigned)(21 * bp * 8); k.z6 = (unsigned)(6 * bp * 8);
    k.zn = (unsigned)((long)a.n_npi * bp * 8); k.z1 = (unsigned)(bp * 8); k.z1i = (unsigned)(bp * 4);
    k.zu = (unsigned)((long)a.n_npi * a.Su * 8);
    k.o36 = (long)t0 * k.z36; k.o21 = (long)t0 * k.z21; k.o6 = (long)t0 * k.z6; k.on = (long)t0 * k.zn;
    k.o1 = (long)t0 * k.z1; k.o1i = (long)t0 * k.z1i; k.ou = (long)t0 * k.zu;
    k.d36 = dir * (long)k.z36; k.d21 = dir * (long)k.z21; k.d6 = dir * (long)k.z6; k.dn = dir * (long)k.zn;
    k.d1 = dir * (long)k.z1; k.d1i = dir * (long)k.z1i; k.du = dir * (long)k.zu;
    return k;
}
EPI_DEV void w_advance(WaveWalk &k)
{
    k.o36 += k.d36; k.o21 += k.d21; k.o6 += k.d6; k.on += k.dn; k.o1 += k.d1; k.o1i += k.d1i; k.ou += k.du;
}
EPI_DEV rsrc_t w_at(const void *p, long off, unsigned size) { return mk_rsrc((const char *)p + off, size); }
EPI_DEV WaveLane w_lane(const KArgs &a, const Lay &lay)
{
    WaveLane w;
    w.own = threadIdx.x < kWE;
    w.e = w.own ? (int)threadIdx.x : 0;
    w.j = w.e / 6; w.i = w.e - 6 * w.j;
    w.npi = (int)threadIdx.x < a.n_npi;
    w.k = threadIdx.x < kNpi ? (int)threadIdx.x : 0;
    w.rowb = lay.blk * 8u;
    // STORES are unconditional: a lane that owns nothing of an array carries an offset beyond the slice, and the buffer
    // descriptor's bounds check drops its store (no EXEC toggling around the ~10 stores of a step)
    constexpr unsigned OOB = 0x80000000u;     // slices stay below 2 GiB (B <= 2^20 in this shape), offset + row offset below 4 GiB
    w.v36 = (lay.cb * 36u * lay.blk + lay.cr) * 8u + (unsigned)w.e * w.rowb;
    w.s36 = w.own ? w.v36 : OOB;
    w.v6 = (lay.cb * 6u * lay.blk + lay.cr) * 8u;                                     // + row * rowb as the scalar offset
    w.s6 = threadIdx.x == 0 ? w.v6 : OOB;
    w.s1 = threadIdx.x == 0 ? lay.c * 8u : OOB;                                       // one-row arrays [T][nblk*blk]
    w.s1i = threadIdx.x == 0 ? lay.c * 4u : OOB;
    w.v1i = lay.c * 4u;
    w.vn = (lay.cb * (unsigned)a.n_npi * lay.blk + lay.cr) * 8u + (unsigned)w.k * w.rowb;
    w.sn = w.npi ? w.vn : OOB;
    const int lo = w.i < w.j ? w.i : w.j, hi = w.i < w.j ? w.j : w.i;
    w.v21 = (lay.cb * 21u * lay.blk + lay.cr) * 8u + (unsigned)(lo + hi * (hi + 1) / 2) * w.rowb;
    return w;
}
// a wave-uniform 6-vector: six stores with scalar row offsets, kept by lane 0 only (the others are out of range)
EPI_DEV void w_store_vec(double *dst, long off, unsigned size, const WaveLane &w, const double (&v)[6])
{
    if (!dst) return;
    const rsrc_t r = w_at(dst, off, size);
#pragma unroll
    for (int q = 0; q < 6; q++) bst(r, w.s6, (unsigned)q * w.rowb, v[q]);
}
EPI_DEV void w_load_vec(const double *src, long off, unsigned size, const WaveLane &w, double (&v)[6])
{
    const rsrc_t r = w_at(src, off, size);
#pragma unroll
    for (int q = 0; q < 6; q++) v[q] = bld(r, w.v6, (unsigned)q * w.rowb);
}

// the uniform Jacobian written to LDS for per-lane row reads.  All lanes hold the same values; structurally zero entries
// were zeroed once at kernel start and are never written again (jacobian_entries: the zero pattern is fixed per model).
EPI_DEV void w_put_jacobian(double *sA, const double (&A)[36])
{
    constexpr int M = 6;
    sA[IXM(0, 0)] = A[IXM(0, 0)]; sA[IXM(0, 1)] = A[IXM(0, 1)]; sA[IXM(0, 2)] = A[IXM(0, 2)];
    sA[IXM(1, 0)] = A[IXM(1, 0)]; sA[IXM(1, 1)] = A[IXM(1, 1)]; sA[IXM(1, 2)] = A[IXM(1, 2)];
    sA[IXM(2, 2)] = A[IXM(2, 2)]; sA[IXM(2, 5)] = A[IXM(2, 5)];
    sA[IXM(3, 1)] = A[IXM(3, 1)]; sA[IXM(3, 2)] = A[IXM(3, 2)]; sA[IXM(3, 3)] = A[IXM(3, 3)]; sA[IXM(3, 4)] = A[IXM(3, 4)];
    sA[IXM(4, 0)] = A[IXM(4, 0)]; sA[IXM(4, 2)] = A[IXM(4, 2)]; sA[IXM(4, 3)] = A[IXM(4, 3)]; sA[IXM(4, 4)] = A[IXM(4, 4)];
    sA[IXM(5, 0)] = A[IXM(5, 0)]; sA[IXM(5, 1)] = A[IXM(5, 1)]; sA[IXM(5, 3)] = A[IXM(5, 3)]; sA[IXM(5, 4)] = A[IXM(5, 4)];
    sA[IXM(5, 5)] = A[IXM(5, 5)];
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
    n.inv_sigma = 1.0 / p.sigma;
    n.a = g(EPI_PRM_A + w.k); n.umin = g(EPI_PRM_U_MIN + w.k); n.umax = g(EPI_PRM_U_MAX + w.k);
    n.ew = p.epsilon * g(EPI_PRM_W_EFF + w.k);
    // same products, same order as slope_term() / nlin_state_update(): constants of the chain, formed once
    n.term = p.gamma * p.dt * (p.sigma / 2.0) * n.a * (n.umax - n.umin);
    if (threadIdx.x < kNpi) sGa[w.k] = p.gamma * n.a;
}
// u(k, t) of my NPI (rows beyond n_npi read 0.0 through the descriptor's bounds check, see load_u); `off` = byte offset
// of time slice t of the control series, `vu` = this lane's offset in it
EPI_DEV double w_load_u(const KArgs &a, long off, unsigned size, unsigned vu)
{
    return bld(w_at(a.u, off, size), vu, 0u);
}
// my NPI at state s: phi (OptControlled.m:49), the control applied (:50-58, strict >) and my slope-term contribution
// (:107-114; 0.0 where the dense code adds nothing -- x - 0.0 == x, so the running sum keeps its bits)
EPI_DEV void w_npi(const QPrm &p, const WaveNpi &n, double u, double s6, double &uapp, double &tterm)
{
    const double gs6 = p.gamma * s6;
    const double phi = n.ew - gs6 * n.a;
    const bool free_u = is_nan(u);
    uapp = free_u ? ((phi > 0.0) ? n.umin : n.umax) : u;
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
template <int FLIP>
__global__ __launch_bounds__(kWave) void ekf_fwd_wave(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int M = 6;
    // tiles are 64 entries so that every lane writes its slot unconditionally (lanes >= 36 own padding)
    __shared__ double sP[kWave], sT[kWave], sA[kWE], sV[16], sGa[kNpi], sD[kWave], sTt[kWave];
    if (*dense_flag) return;
    const int c = a.c0 + (int)blockIdx.x;
    if (c >= a.c0 + a.cn) return;
    const int B = a.B, T = a.T;
    const int sx = a.x_series ? a.x_series[c] : c;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    const WaveLane w = w_lane(a, lay);
    const unsigned lane = threadIdx.x;

    QPrm p;
    WaveNpi np;
    w_load_prm(p, np, a, B, c, w, sGa);
    const double v_bar = a.prm[(size_t)EPI_PRM_V_BAR * B + c];
    const double gamma = a.prm[(size_t)EPI_PRM_GAMMA_EKF * B + c];

    double sk_minus[M];
#pragma unroll
    for (int i = 0; i < M; i++) sk_minus[i] = a.s_init[(size_t)i * B + c];
    double Pm = a.Ps_init[(size_t)w.e * B + c];
    const double Qe = a.Q[(size_t)w.e * B + c];
    if (w.own) sA[w.e] = 0.0;

    const int k_begin = a.k_begin, k_end = (a.k_end > 0 && a.k_end < T) ? a.k_end : T;
    WaveWalk wk = w_walk(a, lay, tpos<FLIP>(k_begin, T), FLIP ? -1 : 1);
    if (k_begin > 0) {           // a later time segment resumes from what the previous one stored
        w_load_vec(a.S_MINUS, wk.o6, wk.z6, w, sk_minus);
        Pm = bld(w_at(a.P_MINUS, wk.o36, wk.z36), w.v36, 0u);
    }
    const unsigned vu = (unsigned)su * 8u + (unsigned)w.k * (unsigned)a.Su * 8u;
    const long dx = (FLIP ? -1L : 1L) * a.Sx;             // x walks the caller's time axis, R_v the filter's (Backward*.m:27)
    const double *px = a.x + (size_t)tpos<FLIP>(k_begin, T) * a.Sx + sx, *pr = a.R_series + (size_t)k_begin * a.Sx + sx;
    double x_nxt = *px, r_nxt = *pr;
    double u_nxt = w_load_u(a, wk.ou, wk.zu, vu);

    for (int k = k_begin; k < k_end; k++) {
        const double Rk = r_nxt, xk = x_nxt, u_in = u_nxt;
        if (k + 1 < T) {
            px += dx; pr += a.Sx;
            x_nxt = *px; r_nxt = *pr;
            u_nxt = w_load_u(a, wk.ou + wk.du, wk.zu, vu);
        }

        // :100-101
        w_store_vec(a.S_MINUS, wk.o6, wk.z6, w, sk_minus);
        bst(w_at(a.P_MINUS, wk.o36, wk.z36), w.s36, 0u, Pm);
        sP[lane] = Pm;
        __syncthreads();

        double C[M];
        obs_jacobian<M>(a.mf, sk_minus, C);                              // :115
        const double xk_minus = predict_obs<M>(a.mf, sk_minus, v_bar);   // :116-119
        double innov, K[M], sk_plus[M], Pp;
        const bool valid = !is_nan(xk);                                  // :122 (wave-uniform)
        if (valid) {
            innov = xk - xk_minus;
            double prow[6], pcol[6];
            w_row(sP, w.i, prow);
            w_col(sP, w.j, pcol);
            const double PCt_i = w_dot6(prow, C);                        // (P C')(i): P(i,0)*C(0), fma ...
            double CP_j = C[0] * pcol[0];                                // (C P)(j): C(0)*P(0,j), fma ...
#pragma unroll
            for (int q = 1; q < M; q++) CP_j = fma(C[q], pcol[q], CP_j);
            if (w.i == 0) sV[w.j] = CP_j;                                // (mirror lanes rewrite entry 0 with lane 0's value)
            __syncthreads();
            double CP[M];
#pragma unroll
            for (int q = 0; q < M; q++) CP[q] = sV[q];
            double CPCt = CP[0] * C[0];
#pragma unroll
            for (int q = 1; q < M; q++) CPCt = fma(CP[q], C[q], CPCt);
            const double den = CPCt + gamma * Rk;                        // :124 (D = 1, Hessian terms 0)
            // P(k|k-1) is bit-wise symmetric (symmetrised at :161; Ps_init by ekf_precheck), so (C P)(j) == (P C')(j)
            const double K_i = PCt_i / den, K_j = CP_j / den;
            if (w.j == 0) sV[8 + w.i] = K_i;
            double ikc_i[6], ikc_j[6];
#pragma unroll
            for (int q = 0; q < M; q++) {
                ikc_i[q] = ((w.i == q) ? 1.0 : 0.0) - K_i * C[q];
                ikc_j[q] = ((w.j == q) ? 1.0 : 0.0) - K_j * C[q];
            }
            const double T1 = w_dot6(ikc_i, pcol);                       // ((I - K C) P)(i,j)
            sT[lane] = T1;
            __syncthreads();
            double t1row[6];
            w_row(sT, w.i, t1row);
            const double T2 = w_dot6(t1row, ikc_j);                      // Joseph form :127
            Pp = (T2 + (K_i * Rk) * K_j) / gamma;
#pragma unroll
            for (int q = 0; q < M; q++) K[q] = sV[8 + q];
#pragma unroll
            for (int q = 0; q < M; q++) sk_plus[q] = sk_minus[q] + K[q] * innov;   // :129
        } else {                                                         // :130-135
            innov = 0.0;
#pragma unroll
            for (int q = 0; q < M; q++) { K[q] = 0.0; sk_plus[q] = sk_minus[q]; }
            Pp = Pm;
        }
        state_hard_margins<M>(p, sk_plus);                               // :141
        // my NPI at s(k|k): the control applied and the slope-term contribution  :155-157
        double u_app, tterm;
        w_npi(p, np, u_in, sk_plus[5], u_app, tterm);
        // :138  (P + P')/2.0
        __syncthreads();
        sP[lane] = Pp;
        sD[lane] = np.umax - u_app;
        sTt[lane] = tterm;
        __syncthreads();
        Pp = (Pp + sP[w.j + 6 * w.i]) / 2.0;

        // s(k+1|k), P(k+1|k)  :155-164
        double sk_next[M];
        {
            const double dot = w_dot_npi(sGa, sD);
            const double a36 = w_slope<FLIP>(sTt);
            state_map<M, FLIP>(p, dot, sk_plus, sk_next);
            double A[M * M];
            jacobian_entries<M, FLIP>(p, sk_plus, a36, A);
            __syncthreads();
            w_put_jacobian(sA, A);
            sT[lane] = Pp;                                               // the symmetrised P(k|k)
            __syncthreads();
            double arow_i[6], arow_j[6], ppcol[6];
            w_row(sA, w.i, arow_i);
            w_row(sA, w.j, arow_j);
            w_col(sT, w.j, ppcol);
            const double T1 = w_dot6(arow_i, ppcol);                     // (A P+)(i,j)
            __syncthreads();
            sP[lane] = T1;
            __syncthreads();
            double t1row[6];
            w_row(sP, w.i, t1row);
            const double T2 = w_dot6(t1row, arow_j);                     // (A P+ A')(i,j)
            double Pn = T2 + Qe;                                         // B = I
            __syncthreads();
            sT[lane] = Pn;
            __syncthreads();
            Pn = (Pn + sT[w.j + 6 * w.i]) / 2.0;                         // :161
            Pm = Pn;
        }
        state_hard_margins<M>(p, sk_next);                               // :164

        // :167-169
        w_store_vec(a.S_PLUS, wk.o6, wk.z6, w, sk_plus);
        w_store_vec(a.K_GAIN, wk.o6, wk.z6, w, K);
        if (a.innovations) bst(w_at(a.innovations, wk.o1, wk.z1), w.s1, 0u, innov);
        if (a.u_opt) bst(w_at(a.u_opt, wk.on, wk.zn), w.sn, 0u, u_app);
        bst(w_at(a.P_PLUS, wk.o36, wk.z36), w.s36, 0u, Pp);
#pragma unroll
        for (int q = 0; q < M; q++) sk_minus[q] = sk_next[q];
        w_advance(wk);
        __syncthreads();
    }
    if (k_end < T) {             // hand-over to the next time segment (wk now points at filter step k_end)
        w_store_vec(a.S_MINUS, wk.o6, wk.z6, w, sk_minus);
        bst(w_at(a.P_MINUS, wk.o36, wk.z36), w.s36, 0u, Pm);
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
// inputs of smoother step k: `wk` points at the array position of filter step k, `w1` at that of step k + 1
EPI_DEV void w_bwd_fetch(const KArgs &a, const WaveLane &w, const WaveWalk &wk, long s1, unsigned vu, WaveBwdIn &o)
{
    // s1 = +1 / -1: where step k + 1 sits relative to step k on the caller's time axis
    w_load_vec(a.S_PLUS, wk.o6, wk.z6, w, o.Sp);
    w_load_vec(a.S_MINUS, wk.o6 + s1 * (long)wk.z6, wk.z6, w, o.Sm1);
    o.Pp = bld(w_at(a.P_PLUS, wk.o36, wk.z36), w.v36, 0u);
    o.Pm1 = bld(w_at(a.P_MINUS, wk.o36 + s1 * (long)wk.z36, wk.z36), w.v36, 0u);
    // X is stored packed (upper triangle, column by column): element (i, j), i <= j, at row i + j (j + 1) / 2
    o.X = bld(w_at(a.X, wk.o21 + s1 * (long)wk.z21, wk.z21), w.v21, 0u);
    o.rk = __builtin_amdgcn_raw_buffer_load_b32(w_at(a.rankbuf, wk.o1i + s1 * (long)wk.z1i, wk.z1i), w.v1i, 0u, 0);
    o.u = w_load_u(a, wk.ou, wk.zu, vu);
}

template <int FLIP>
__global__ __launch_bounds__(kWave) void eks_bwd_wave(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int M = 6;
    __shared__ double sP[kWave], sT[kWave], sA[kWE], sX[kWave], sJ[kWave], sV[8], sGa[kNpi], sTt[kWave];
    if (*dense_flag) return;
    const int c = a.c0 + (int)blockIdx.x;
    if (c >= a.c0 + a.cn) return;
    const int B = a.B, T = a.T;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    const WaveLane w = w_lane(a, lay);
    const unsigned lane = threadIdx.x;
    QPrm p;
    WaveNpi np;
    w_load_prm(p, np, a, B, c, w, sGa);
    if (w.own) sA[w.e] = 0.0;
    const unsigned vu = (unsigned)su * 8u + (unsigned)w.k * (unsigned)a.Su * 8u;

    // terminal condition :189-202
    WaveWalk wk = w_walk(a, lay, tpos<FLIP>(T - 1, T), FLIP ? 1 : -1);      // the walk runs DOWN the filter's time axis
    double Ss[M], Ps;
    w_load_vec(a.S_PLUS, wk.o6, wk.z6, w, Ss);
#pragma unroll
    for (int i = 0; i < M; i++) {
        const double f = a.s_final[(size_t)i * B + c];
        if (!is_nan(f)) Ss[i] = f;
    }
    Ps = bld(w_at(a.P_PLUS, wk.o36, wk.z36), w.v36, 0u);
    {
        const double f = a.Ps_final[(size_t)w.e * B + c];
        if (!is_nan(f)) Ps = f;
    }
    w_store_vec(a.S_SMOOTH, wk.o6, wk.z6, w, Ss);
    if (a.u_opt_smooth) bst(w_at(a.u_opt_smooth, wk.on, wk.zn), w.sn, 0u, 0.0);      // column T is never written :95,204
    if (a.pinv_rank) __builtin_amdgcn_raw_buffer_store_b32(-1, w_at(a.pinv_rank, wk.o1i, wk.z1i), w.s1i, 0u, 0);
    if (a.P_SMOOTH) bst(w_at(a.P_SMOOTH, wk.o36, wk.z36), w.s36, 0u, Ps);

    int st_guard = 0, st_cap = 0, min_rank = M;
    const long s1 = FLIP ? -1L : 1L;
    WaveBwdIn nxt;
    w_advance(wk);                                       // -> filter step T - 2
    if (T >= 2) w_bwd_fetch(a, w, wk, s1, vu, nxt);
    for (int k = T - 2; k >= 0; k--) {
        const WaveBwdIn cur = nxt;
        const WaveWalk here = wk;
        w_advance(wk);
        if (k > 0) w_bwd_fetch(a, w, wk, s1, vu, nxt);

        // the Jacobian at S+(k) with the ORIGINAL control column :206
        double u_unused, tterm;
        w_npi(p, np, cur.u, cur.Sp[5], u_unused, tterm);
        __syncthreads();
        sTt[lane] = tterm;
        sP[lane] = cur.Pp; sX[lane] = cur.X;
        __syncthreads();
        double A[M * M];
        jacobian_entries<M, FLIP>(p, cur.Sp, w_slope<FLIP>(sTt), A);
        w_put_jacobian(sA, A);
        __syncthreads();
        double J = 0.0;
        int rank = -1;
        if (cur.rk < 0) {                                                // non-finite P_MINUS guard :211-213
            st_guard = 1;
        } else {
            double pprow[6], arow_j[6];
            w_row(sP, w.i, pprow);
            w_row(sA, w.j, arow_j);
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
        for (int q = 0; q < M; q++) dv[q] = Ss[q] - cur.Sm1[q];
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
        if (w.j == 0) sV[w.i] = Jd_i;
        const double T1 = w_dot6(jrow_i, dcol);                          // (J D)(i,j)
        __syncthreads();
        sP[lane] = T1;
        __syncthreads();
        double Sn[M];
#pragma unroll
        for (int q = 0; q < M; q++) Sn[q] = cur.Sp[q] + sV[q];
        state_hard_margins<M>(p, Sn);
        // u_opt_smooth(:, k) = the control NlinStateUpdate applies at S_SMOOTH(k)   :229
        double u_s, t_unused;
        w_npi(p, np, cur.u, Sn[5], u_s, t_unused);
        double t1row[6];
        w_row(sP, w.i, t1row);
        const double T2 = w_dot6(t1row, jrow_j);                         // (J D J')(i,j)
        double Pn = cur.Pp - T2;                                         // :223
        __syncthreads();
        sT[lane] = Pn;
        __syncthreads();
        Pn = (Pn + sT[w.j + 6 * w.i]) / 2.0;                             // :226
        Ps = Pn;
#pragma unroll
        for (int q = 0; q < M; q++) Ss[q] = Sn[q];
        w_store_vec(a.S_SMOOTH, here.o6, here.z6, w, Ss);
        if (a.u_opt_smooth) bst(w_at(a.u_opt_smooth, here.on, here.zn), w.sn, 0u, u_s);
        if (a.pinv_rank) __builtin_amdgcn_raw_buffer_store_b32(rank, w_at(a.pinv_rank, here.o1i, here.z1i), w.s1i, 0u, 0);
        if (a.P_SMOOTH) bst(w_at(a.P_SMOOTH, here.o36, here.z36), w.s36, 0u, Ps);
    }
    if (threadIdx.x == 0 && a.status) a.status[c] = st_guard | (st_cap << 1) | (min_rank << 8);
}
