// ekf_sym.hpp -- register-lean forward / backward kernels for the GenericExtendedKalmanFilter models.
// Included from epiekf.hip inside namespace epi (after KArgs and the load/store helpers).
//
// Same arithmetic as the dense kernels (and the oracle), operation for operation, but
//  * every covariance the generic filter carries is bit-wise symmetric (it is symmetrised at
//    GenericEKF.m:138,161,226 and Ps_init is checked by ekf_precheck), so only the 21 (6) unique
//    entries live in registers, packed upper-triangular;
//  * terms that multiply a STRUCTURAL zero of the Jacobian A (SIAlphaModelEKFOptControlled.m:89-135:
//    21 of 36 entries can be non-zero) or of (I - K C) (C(4:6) = 0, :138-148) are skipped.  For finite
//    operands fma(a, 0, acc) == acc, so skipping them is exact; a chain whose covariance has overflowed
//    to Inf/NaN can differ from the dense evaluation in WHERE the non-finite values sit (both are
//    garbage there; status bit 0 / the J = 0 guard of :211 still fire) -- DESIGN.md "Arithmetic contract";
//  * the smoother keeps the four 12-vectors of `params` in an LDS column per lane instead of 96 VGPRs.
// Net effect for m = 6: about half the fp64 operations and 300-370 instead of 470-512 VGPRs (no scratch); still
// one wave per SIMD -- see DESIGN.md "Occupancy and the wave-count quantum" for what was tried to get to two.
#pragma once

template <int M> constexpr int nsym() { return M * (M + 1) / 2; }
constexpr int sidx(int i, int j) { return i <= j ? i + j * (j + 1) / 2 : j + i * (i + 1) / 2; }

// structural non-zero pattern of StateJacobians (same for the time-flipped twins)
template <int M> constexpr bool a_nz(int i, int k)
{
    if (M == 3) return (i < 2) ? true : (k == 2);
    return i == 0 ? (k < 3)
         : i == 1 ? (k < 3)
         : i == 2 ? (k == 2 || k == 5)
         : i == 3 ? (k >= 1 && k <= 4)
         : i == 4 ? (k == 0 || k == 2 || k == 3 || k == 4)
                  : (k == 0 || k == 1 || k == 3 || k == 4 || k == 5);
}

template <int M>
EPI_DEV void store_sym(double *__restrict__ dst, int t, const Lay &l, const double (&P)[nsym<M>()], bool upper_only = false)
{
    if (!dst) return;
    unsigned voff, rowb;
    const rsrc_t r = lay_slice(dst, t, M * M, l, voff, rowb);
    if (upper_only) {      // workspace read back through load_sym / eks_pinv only: the mirror entries are never read
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) bst(r, voff, (unsigned)IXM(i, j) * rowb, P[sidx(i, j)]);
        return;
    }
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = 0; i < M; i++) bst(r, voff, (unsigned)IXM(i, j) * rowb, P[sidx(i, j)]);
}
template <int M>
EPI_DEV void store_sym_f32(float *__restrict__ dst, int t, const Lay &l, const double (&P)[nsym<M>()])
{
    if (!dst) return;
    unsigned voff, rowb;
    const rsrc_t r = lay_slice_f32(dst, t, M * M, l, voff, rowb);
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = 0; i < M; i++) bst32(r, voff, (unsigned)IXM(i, j) * rowb, P[sidx(i, j)]);
}
template <int M>
EPI_DEV void load_sym(const double *__restrict__ src, int t, const Lay &l, double (&P)[nsym<M>()])
{
    unsigned voff, rowb;
    const rsrc_t r = lay_slice(src, t, M * M, l, voff, rowb);
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = 0; i <= j; i++) P[sidx(i, j)] = bld_s(r, voff, (unsigned)IXM(i, j) * rowb);
}

// ---------------------------------------------------------------------------
// pre-check: may this batch take the symmetric fast path?  (Ps_init and Ps_final bit-wise symmetric, Q_w diagonal,
// s_init / Ps_init / Q_w finite)
// ---------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(256) void ekf_precheck(const KArgs a, int *__restrict__ flag, int force_dense)
{
    if (force_dense) { if (blockIdx.x == 0 && threadIdx.x == 0) *flag = 1; return; }
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.B) return;
    const int B = a.B;
    bool dense = false;
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = 0; i < j; i++) {
            const double pu = a.Ps_init[(size_t)IXM(i, j) * B + c], pl = a.Ps_init[(size_t)IXM(j, i) * B + c];
            const bool same = (pu == pl) || (is_nan(pu) && is_nan(pl));
            const double qu = a.Q[(size_t)IXM(i, j) * B + c], ql = a.Q[(size_t)IXM(j, i) * B + c];
            // the end-point covariance overrides P_SMOOTH(:,:,T) entry by entry (GenericEKF.m:198-202): the
            // packed smoother needs the result symmetric, i.e. Ps_final symmetric in values and NaN pattern
            const double fu = a.Ps_final[(size_t)IXM(i, j) * B + c], fl = a.Ps_final[(size_t)IXM(j, i) * B + c];
            const bool fsame = (fu == fl) || (is_nan(fu) && is_nan(fl));
            dense = dense || !same || !fsame || !(qu == 0.0) || !(ql == 0.0);
        }
    // a non-finite entry in the initial state, the initial covariance or the process noise (e.g. the NaN "free end
    // point" markers handed to a time-flipped wrapper as its initial condition): the packed kernels skip products with
    // structural zeros, which is exact only for finite operands -- such a batch mirrors the oracle through the dense
    // kernels instead
#pragma unroll
    for (int i = 0; i < M; i++) {
        dense = dense || is_nonfinite(a.s_init[(size_t)i * B + c]) || is_nonfinite(a.Q[(size_t)IXM(i, i) * B + c]);
#pragma unroll
        for (int j = i; j < M; j++) dense = dense || is_nonfinite(a.Ps_init[(size_t)IXM(i, j) * B + c]);
    }
    if (dense) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------
// forward pass, symmetric-packed
// ---------------------------------------------------------------------------
// Launch bounds.  Capping at two waves per SIMD (kWave, 2 => 256 VGPRs) makes hipcc 7.2 spill ~430-680 B
// per lane to scratch, which measured slower (15.3 / 26.0 ms) than one spill-free wave per SIMD
// (11.8 / 14.1 ms) on the headline sweep; see DESIGN.md "Occupancy and the wave-count quantum".
#ifndef EPI_FWD_LB
#define EPI_FWD_LB kWave
#endif
#ifndef EPI_BWD_LB
#define EPI_BWD_LB kWave
#endif
#ifndef EPI_FWD3_WAVES
#define EPI_FWD3_WAVES 3          // waves per SIMD the 3-state forward variant with LDS-resident a / u_max is compiled for
#endif
// Two knobs of the packed smoother, settled by A/B runs on the headline sweep (profiles/ab_phase.py 4, medians):
//   EPI_BWD_PREFETCH  what is requested one step ahead (bit 0: state, controls, rank word; bit 1: P_PLUS; bit 2: X)
//   EPI_BWD_RECOMPUTE 1: s(k+1|k), P(k+1|k) are recomputed from the stored s(k|k), P(k|k), u with the forward kernel's
//                     own functions (bit-identical, 27 fewer loads per step, 10 GB less traffic per pass) -- but the
//                     kernel then needs all 512 registers, and with the prefetch and the blocked-layout addressing on
//                     top it spills (56 B of scratch per lane: 12.5 ms);  0: they are read back at their point of use.
//   chain-blocked outputs, blk = 8:  recompute 0 / prefetch 1: 8.0 ms (372 VGPRs)   0 / 5: 8.1   0 / 0: 8.4   1 / 0: 9.5
//   classic [T][rows][B]:            0 / 1: 8.6   0 / 0: 9.1   1 / 0: 9.6
#ifndef EPI_BWD_PREFETCH
#define EPI_BWD_PREFETCH 1
#endif
#ifndef EPI_BWD_RECOMPUTE
#define EPI_BWD_RECOMPUTE 0
#endif
constexpr int kPipeLanes = 40;   // lanes per workgroup of the LP = 1 forward variant
// where the forward kernel keeps the model constants (see ekf_fwd_sym)
template <int LP> struct PrmSelect { typedef ChainPrm type; };
template <> struct PrmSelect<1> { typedef LitePrm<VecLdsS> type; };
template <> struct PrmSelect<2> { typedef LitePrm<VecLds2> type; };     // 3-state: a, u_max in LDS (64-lane stride), see ekf_fwd_sym
template <int M>
EPI_DEV void init_prm(ChainPrm &p, const KArgs &a, int B, int c, double *, int) { load_prm<M>(p, a.prm, B, c, a.mf.lo_is_zero); }
template <int M>
EPI_DEV void init_prm(LitePrm<VecLdsS> &p, const KArgs &a, int B, int c, double *col, int stride)
{
    load_lite(p, a.prm, B, c, a.mf.lo_is_zero);
    p.v.base = col; p.v.stride = stride;
#pragma unroll
    for (int k = 0; k < kNpi; k++) {
        col[(0 * kNpi + k) * stride] = a.prm[(size_t)(EPI_PRM_A + k) * B + c];
        col[(1 * kNpi + k) * stride] = a.prm[(size_t)(EPI_PRM_U_MIN + k) * B + c];
        col[(2 * kNpi + k) * stride] = a.prm[(size_t)(EPI_PRM_U_MAX + k) * B + c];
        col[(3 * kNpi + k) * stride] = a.prm[(size_t)(EPI_PRM_W_EFF + k) * B + c];
    }
}

template <int M>
EPI_DEV void init_prm(LitePrm<VecLds2> &p, const KArgs &a, int B, int c, double *col, int)
{
    load_lite(p, a.prm, B, c, a.mf.lo_is_zero);
    p.v.base = col;
#pragma unroll
    for (int k = 0; k < kNpi; k++) {
        col[(0 * kNpi + k) * kWave] = a.prm[(size_t)(EPI_PRM_A + k) * B + c];
        col[(1 * kNpi + k) * kWave] = a.prm[(size_t)(EPI_PRM_U_MAX + k) * B + c];
    }
}

// P(k+1|k) = sym(A P(k|k) A' + Q), Q diagonal (GenericEKF.m:158-161), packed in and out.  One ROW at a time with
// the structural zeros of A skipped: row i of T1 = A P is formed, consumed into row i of the full result G, and as
// soon as both G(i,j) and G(j,i) exist they are averaged into the packed result.  Shared by ekf_fwd_sym and by
// eks_bwd_sym, which recomputes P(k+1|k) from the stored P(k|k) instead of reading it back: same function, same
// operands, same bits.
template <int M>
EPI_DEV void predict_cov_sym(const double (&A)[M * M], const double (&Pp)[M * (M + 1) / 2], const double (&Qd)[M],
                             double (&Pm)[M * (M + 1) / 2])
{
    double G[M * M];
#pragma unroll
    for (int i = 0; i < M; i++) {
        double T1r[M];
#pragma unroll
        for (int j = 0; j < M; j++) {
            double acc = 0.0;
            bool first = true;
#pragma unroll
            for (int q = 0; q < M; q++)
                if (a_nz<M>(i, q)) {
                    acc = first ? A[IXM(i, q)] * Pp[sidx(q, j)] : fma(A[IXM(i, q)], Pp[sidx(q, j)], acc);
                    first = false;
                }
            T1r[j] = acc;
        }
#pragma unroll
        for (int j = 0; j < M; j++) {
            double acc = 0.0;
            bool first = true;
#pragma unroll
            for (int q = 0; q < M; q++)
                if (a_nz<M>(j, q)) {
                    acc = first ? T1r[q] * A[IXM(j, q)] : fma(T1r[q], A[IXM(j, q)], acc);
                    first = false;
                }
            G[IXM(i, j)] = acc + ((i == j) ? Qd[i] : 0.0);
        }
#pragma unroll
        for (int j = 0; j < i; j++) Pm[sidx(j, i)] = (G[IXM(i, j)] + G[IXM(j, i)]) / 2.0;
        Pm[sidx(i, i)] = (G[IXM(i, i)] + G[IXM(i, i)]) / 2.0;
    }
}

// LP = 0: the model constants live in VGPRs (ChainPrm), the windows use a 64-lane stride -- 408 VGPRs, one wave per
//         SIMD, nothing else fits beside it.
// LP = 1: the four 12-vectors of `params` live in LDS next to the windows and every LDS array is sized by the lanes the
//         workgroup really uses (kPipeLanes = 40) -- 298 VGPRs and (3 L + 48) * 40 * 8 bytes of LDS, so that four such
//         waves fit a CU AND an eks_pinv wave (168 VGPRs, no LDS) fits beside each of them: the pipelined launch
//         (epi_batch_desc.chunks = -2) runs one half's eks_pinv grid in the issue slots the other half's forward waves
//         leave idle.
// MON = 0: the innovation monitor runs as a kernel of its own (ekf_monitor, ekf_quad.hpp: r_mode 1 only) -- no windows here
// USD = 1 ("u same day", round 5): the twelve controls of a day are requested at the top of THAT day, still ahead of its stores
//         -- they are first used ~700 instructions later (NlinStateUpdate, after the Joseph form) -- instead of a day ahead: 24
//         registers no longer live across the whole loop, and <6, FLIP, 1, 0, 0, 1> needs 255 registers and no accumulation
//         registers at all: TWO waves per SIMD (the verdict r04's item 2; 255 + 20 without).  Used where the kernel is bound by
//         the per-day latency of a lone wave, i.e. when the forward quantities are workspace (reduced outputs): forward stage
//         4.23 -> 3.86 ms; with all outputs, where it is bound by its 104 store rows per day, two resident waves per SIMD write
//         WORSE (5.7 -> 6.0 ms), and the 3-state kernels use the controls too early in the day (config 5: 8.7 -> 9.4 ms).
template <int M, int FLIP, int LP, int STOR = 0, int MON = 1, int USD = 0>
__global__ __launch_bounds__(EPI_FWD_LB, (M == 3 && LP == 2) ? EPI_FWD3_WAVES : 1) void ekf_fwd_sym(const KArgs a, const int *__restrict__ dense_flag)
{
    extern __shared__ double lds[];   // three sliding windows [3][L][stride], one column per lane (+ [48][stride], LP)
    if (*dense_flag) return;          // ekf_fwd (dense) runs instead
    constexpr int NS = nsym<M>();
    const int lane = threadIdx.x;
    const int c = a.c0 + blockIdx.x * a.lw + lane;
    if (lane >= a.lw || c >= a.c0 + a.cn) return;
    const int B = a.B, T = a.T, L = a.L;
    const int sx = a.x_series ? a.x_series[c] : c;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);

    constexpr int stride = (LP == 1) ? kPipeLanes : kWave;   // compile-time: LDS offsets stay immediates
    typename PrmSelect<LP>::type p;
    init_prm<M>(p, a, B, c, lds + (size_t)(MON ? 3 * L : 0) * stride + lane, stride);
    const double v_bar = a.prm[(size_t)EPI_PRM_V_BAR * B + c];
    const double beta = a.prm[(size_t)EPI_PRM_BETA_EKF * B + c];
    const double gamma = a.prm[(size_t)EPI_PRM_GAMMA_EKF * B + c];

    double sk_minus[M], Pm[NS], Qd[M];
#pragma unroll
    for (int i = 0; i < M; i++) {
        sk_minus[i] = a.s_init[(size_t)i * B + c];
        Qd[i] = a.Q[(size_t)IXM(i, i) * B + c];
    }
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = 0; i <= j; i++) Pm[sidx(i, j)] = a.Ps_init[(size_t)IXM(i, j) * B + c];
    // Time segments (launch_chain, "pipelined in time"): this launch runs filter steps [k_begin, k_end).  A later segment
    // resumes from s(k|k-1), P(k|k-1) exactly as the previous one left them in S_MINUS / P_MINUS (fp64, stored below when
    // the segment ends before T) -- the same values the single launch would carry in registers.  MON = 0 only (no windows).
    const int k_begin = a.k_begin, k_end = (a.k_end > 0 && a.k_end < T) ? a.k_end : T;
    if (k_begin > 0) {
        load_vec<M>(a.S_MINUS, tpos<FLIP>(k_begin, T), lay, sk_minus);
        load_sym<M>(a.P_MINUS, tpos<FLIP>(k_begin, T), lay, Pm);
    }

    double *winMean = lds + lane, *winCov = lds + (size_t)L * stride + lane, *winCovN = lds + (size_t)2 * L * stride + lane;
    if (MON)
        for (int j = 0; j < L; j++) { winMean[j * stride] = 0.0; winCov[j * stride] = 0.0; winCovN[j * stride] = 0.0; }
    int head = 0;
    const bool fixed_R = (a.r_mode == 0);
    const double R_v = fixed_R ? a.R_scalar[c] : 0.0;
    double R_next = R_v;

    // Software pipeline of the inputs.  Vector-memory operations retire in issue order (one vmcnt), so a
    // load issued after a step's ~100 stores would wait for all of them to drain; the inputs of step k+1 are
    // therefore requested at the top of step k, ahead of its stores, and consumed one iteration later.
    const unsigned voff = (unsigned)c * 8u, voff_x = (unsigned)sx * 8u;
    double x_nxt = ldg(a.x + (size_t)tpos<FLIP>(k_begin, T) * a.Sx, voff_x);
    double r_nxt = fixed_R ? 0.0 : ldg(a.R_series + (size_t)k_begin * a.Sx, voff_x);
    double u_nxt[kNpi];
    if (!USD) load_u(a, tpos<FLIP>(k_begin, T), su, u_nxt);
    // LP = 2 (3-state): NlinStateUpdate returns u as it came (SIAlphaModelEKF.m:39) and its fma chain (gamma a')(u_max - u)
    // does not depend on the state, so a day's controls are consumed when they ARRIVE -- at the end of the day before: u_opt
    // stored, the dot product formed -- and only that scalar crosses into the day (one control vector live instead of two)
    double dot_cur = 0.0;
    auto consume_u = [&](int tt, bool store) __attribute__((always_inline)) {
        if (store) {
            store_u(a.u_opt, a, tt, lay, u_nxt);
            if (STOR) store_rows_f32<kNpi>(a.f.u_opt, tt, (unsigned)a.n_npi, lay, u_nxt);
        }
        dot_cur = (p.gamma * p.A(0)) * (p.Umax(0) - u_nxt[0]);
#pragma unroll
        for (int q = 1; q < kNpi; q++) dot_cur = fma(p.gamma * p.A(q), p.Umax(q) - u_nxt[q], dot_cur);
    };
    if (LP == 2 && k_begin < k_end) consume_u(tpos<FLIP>(k_begin, T), true);

    for (int k = k_begin; k < k_end; k++) {
        const int t = tpos<FLIP>(k, T);
        const double Rk = fixed_R ? R_next : r_nxt;
        const double xk = x_nxt;
        if constexpr (LP == 2) {
            // the LDS column's address is made opaque once a day: hipcc otherwise hoists the 24 reads of a(k), u_max(k) out of
            // the loop -- back into the 48 registers the LDS copy is there to free
            const double *q = p.v.base;
            asm volatile("" : "+v"(q));
            p.v.base = q;
        }
        double u_in[kNpi];
        const double dot_day = dot_cur;
        if (USD) load_u(a, t, su, u_in);
        else if (LP != 2) {
#pragma unroll
            for (int q = 0; q < kNpi; q++) u_in[q] = u_nxt[q];
        }
        if (k + 1 < T) {
            const int tn = tpos<FLIP>(k + 1, T);
            x_nxt = ldg(a.x + (size_t)tn * a.Sx, voff_x);
            if (!fixed_R) r_nxt = ldg(a.R_series + (size_t)(k + 1) * a.Sx, voff_x);
            if (!USD) load_u(a, tn, su, u_nxt);
        }

        // (fp32 storage, three states: the fp64 S_MINUS is workspace that nobody reads -- eks_bwd_sym<3> recomputes s(k+1|k), the
        // pinv grid reads P_MINUS only, a later time segment resumes from the hand-over row stored below -- 24 of ~330 bytes a step)
        if (!(STOR && M == 3)) store_vec<M>(a.S_MINUS, t, lay, sk_minus);
        store_sym<M>(a.P_MINUS, t, lay, Pm, (a.ws_upper & 1) != 0);
        if (STOR) { store_rows_f32<M>(a.f.S_MINUS, t, M, lay, sk_minus); store_sym_f32<M>(a.f.P_MINUS, t, lay, Pm); }

        double C[M];
        obs_jacobian<M>(a.mf, sk_minus, C);                 // C(4:6) == 0 for m = 6
        const double xk_minus = predict_obs<M>(a.mf, sk_minus, v_bar);

        double innov, K[M], sk_plus[M], Pp[NS];
        const bool valid = !is_nan(xk);
        if (valid) {
            innov = xk - xk_minus;
            // P C' (== (C P)' bit for bit, P symmetric); only C(1:3) can be non-zero
            double PCt[M];
#pragma unroll
            for (int i = 0; i < M; i++) {
                double acc = Pm[sidx(i, 0)] * C[0];
                acc = fma(Pm[sidx(i, 1)], C[1], acc);
                acc = fma(Pm[sidx(i, 2)], C[2], acc);
                PCt[i] = acc;
            }
            double CPCt = PCt[0] * C[0];
            CPCt = fma(PCt[1], C[1], CPCt);
            CPCt = fma(PCt[2], C[2], CPCt);
            const double den = CPCt + gamma * Rk;
#pragma unroll
            for (int i = 0; i < M; i++) K[i] = PCt[i] / den;
            // (I - K C): columns 4..6 are exactly those of the identity
            double IKC[M][3];
#pragma unroll
            for (int i = 0; i < M; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) IKC[i][j] = ((i == j) ? 1.0 : 0.0) - K[i] * C[j];
            // P+ = sym(((I - K C) P (I - K C)' + K R K') / gamma), one ROW of the Joseph form at a time: row i of
            // T1 = (I - K C) P is formed, consumed into row i of the full result F, and as soon as both
            // F(i,j) and F(j,i) exist they are averaged into the packed P+ -- at most one 6-vector of T1 and
            // the not-yet-paired upper entries of F are live, instead of two full 6 x 6 temporaries
            double F[M * M];
#pragma unroll
            for (int i = 0; i < M; i++) {
                double T1r[M];
#pragma unroll
                for (int j = 0; j < M; j++) {
                    double acc = IKC[i][0] * Pm[sidx(0, j)];
                    acc = fma(IKC[i][1], Pm[sidx(1, j)], acc);
                    acc = fma(IKC[i][2], Pm[sidx(2, j)], acc);
                    if (i >= 3) acc = acc + Pm[sidx(i, j)];          // + 1 * P(i,j)
                    T1r[j] = acc;
                }
#pragma unroll
                for (int j = 0; j < M; j++) {
                    double acc = T1r[0] * IKC[j][0];
                    acc = fma(T1r[1], IKC[j][1], acc);
                    acc = fma(T1r[2], IKC[j][2], acc);
                    if (j >= 3) acc = acc + T1r[j];
                    F[IXM(i, j)] = (acc + (K[i] * Rk) * K[j]) / gamma;
                }
#pragma unroll
                for (int j = 0; j < i; j++) Pp[sidx(j, i)] = (F[IXM(i, j)] + F[IXM(j, i)]) / 2.0;
                Pp[sidx(i, i)] = (F[IXM(i, i)] + F[IXM(i, i)]) / 2.0;
            }
#pragma unroll
            for (int i = 0; i < M; i++) sk_plus[i] = sk_minus[i] + K[i] * innov;
        } else {
            innov = 0.0;
#pragma unroll
            for (int i = 0; i < M; i++) { K[i] = 0.0; sk_plus[i] = sk_minus[i]; }
#pragma unroll
            for (int e = 0; e < NS; e++) Pp[e] = Pm[e];
        }
        state_hard_margins<M>(p, sk_plus);

        if constexpr (LP == 2) {
            state_map<M, FLIP>(p, dot_day, sk_plus, sk_minus);
            double A[M * M];
            jacobian_entries<M, FLIP>(p, sk_plus, 0.0, A);         // no slope term for three states
            predict_cov_sym<M>(A, Pp, Qd, Pm);
            if (k + 1 < T) consume_u(tpos<FLIP>(k + 1, T), k + 1 < k_end);   // tomorrow's controls have arrived
        } else {
            double u_app[kNpi];
#pragma unroll
            for (int q = 0; q < kNpi; q++) u_app[q] = u_in[q];
            nlin_state_update<M, FLIP>(p, a.mf, u_app, sk_plus, sk_minus);
            store_u(a.u_opt, a, t, lay, u_app);
            if (STOR) store_rows_f32<kNpi>(a.f.u_opt, t, (unsigned)a.n_npi, lay, u_app);
            double A[M * M];
            state_jacobians<M, FLIP>(p, u_in, sk_plus, A);
            predict_cov_sym<M>(A, Pp, Qd, Pm);
        }
        state_hard_margins<M>(p, sk_minus);

        store_vec<M>(a.S_PLUS, t, lay, sk_plus);
        store_sym<M>(a.P_PLUS, t, lay, Pp, (a.ws_upper & 2) != 0);
        store_vec<M>(a.K_GAIN, t, lay, K);
        if (a.innovations) a.innovations[lay_scalar(t, lay)] = innov;
        if (STOR) {
            store_rows_f32<M>(a.f.S_PLUS, t, M, lay, sk_plus); store_sym_f32<M>(a.f.P_PLUS, t, lay, Pp);
            store_rows_f32<M>(a.f.K_GAIN, t, M, lay, K); store_scalar_f32(a.f.innovations, t, lay, innov);
        }

        if (!MON) continue;
        // innovation monitor (identical to ekf_fwd)
        const int cnt = (k + 1 < L) ? (k + 1) : L;
        head = (head == 0) ? (L - 1) : (head - 1);
        winMean[head * stride] = innov;
        const double sum = ring_sum(winMean, head, L, innov, stride);
        const double mu = sum / (double)cnt;
        const double cc = (innov - mu) * (innov - mu);
        const double ccn = cc / (Rk + kEps);
        winCov[head * stride] = cc;
        winCovN[head * stride] = ccn;
        const double sumN = ring_sum(winCovN, head, L, ccn, stride);
        // rho keeps FILTER-step order also for the time-flipped wrappers: GenericEKF.m:233 squeezes it to T x 1 and
        // Backward*.m:40 reverses a third dimension of size 1, i.e. nothing
        if (a.rho) a.rho[lay_scalar(k, lay)] = sumN / (double)cnt;
        if (STOR) store_scalar_f32(a.f.rho, k, lay, sumN / (double)cnt);
        if (fixed_R) {
            if (beta != 1.0 && valid && k < T - 1) {
                const double sumC = ring_sum(winCov, head, L, cc, stride);
                R_next = beta * Rk + (1.0 - beta) * (sumC / (double)cnt);
            } else {
                R_next = R_v;
            }
        }
    }
    if (k_end < T) {       // hand-over to the next time segment
        store_vec<M>(a.S_MINUS, tpos<FLIP>(k_end, T), lay, sk_minus);
        store_sym<M>(a.P_MINUS, tpos<FLIP>(k_end, T), lay, Pm, (a.ws_upper & 1) != 0);
    }
}

// ---------------------------------------------------------------------------
// backward recursion, symmetric-packed (X = pinv(P_MINUS) comes from eks_pinv)
// ---------------------------------------------------------------------------
template <int M>
struct BwdIn {   // everything smoother step k reads: forward quantities of step k and X = pinv(P(k+1|k))
    double Sp[M], Pp[M * (M + 1) / 2], u[kNpi], X[M * (M + 1) / 2];    // P_PLUS and X packed (both symmetric bit for bit)
    int rk;
};
template <int M, int FLIP, int STOR = 0>
__global__ __launch_bounds__(EPI_BWD_LB) void eks_bwd_sym(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int NV = (M == 6) ? 4 : 2;        // 6 states: a, u_min, u_max, w; 3 states: a, u_max (VecLds2)
    __shared__ double vlds[NV * kNpi * kWave];  // one column per lane
    if (*dense_flag) return;
    constexpr int NS = nsym<M>();
    const int lane = threadIdx.x;
    const int c = a.c0 + blockIdx.x * a.lw + lane;
    if (lane >= a.lw || c >= a.c0 + a.cn) return;
    const int B = a.B, T = a.T;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    LitePrm<typename std::conditional<M == 6, VecLds, VecLds2>::type> p;
    load_lite(p, a.prm, B, c, a.mf.lo_is_zero);
    p.v.base = vlds + lane;
#pragma unroll
    for (int k = 0; k < kNpi; k++) {
        vlds[(0 * kNpi + k) * kWave + lane] = a.prm[(size_t)(EPI_PRM_A + k) * B + c];
        if constexpr (M == 6) {
            vlds[(1 * kNpi + k) * kWave + lane] = a.prm[(size_t)(EPI_PRM_U_MIN + k) * B + c];
            vlds[(2 * kNpi + k) * kWave + lane] = a.prm[(size_t)(EPI_PRM_U_MAX + k) * B + c];
            vlds[(3 * kNpi + k) * kWave + lane] = a.prm[(size_t)(EPI_PRM_W_EFF + k) * B + c];
        } else {
            vlds[(1 * kNpi + k) * kWave + lane] = a.prm[(size_t)(EPI_PRM_U_MAX + k) * B + c];
        }
    }

    double Qd[M];
#pragma unroll
    for (int i = 0; i < M; i++) Qd[i] = a.Q[(size_t)IXM(i, i) * B + c];

    // Smoother steps of this launch: k = k_from down to k_to (launch_chain may cut the recursion in two so that the
    // sweep's scoring tail starts once the horizon days are final).  k_from = T - 2 starts from the terminal condition; a
    // later launch resumes from what the previous one left in the hand-over rows.
    const int k_from = a.bk_from, k_to = a.bk_to;
    const size_t hp = (size_t)a.hand_pitch;
    int st_guard = 0, st_cap = 0, min_rank = M;
    // terminal conditions GenericEKF.m:189-202.  Ps_final overrides entry by entry; ekf_precheck guarantees it
    // is symmetric (values and NaN pattern), so P_SMOOTH(:,:,T) is symmetric and stays packed.
    double Ss[M], Ps[NS];
    const int tT = tpos<FLIP>(T - 1, T);
    if (k_from < T - 2) {
#pragma unroll
        for (int i = 0; i < M; i++) Ss[i] = a.hand_s[(size_t)i * hp + c];
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) Ps[sidx(i, j)] = a.hand_p[(size_t)IXM(i, j) * hp + c];
        const int word = a.hand_i[c];
        st_guard = word & 1; st_cap = (word >> 1) & 1; min_rank = word >> 8;
    } else {
        load_vec<M>(a.S_PLUS, tT, lay, Ss);
#pragma unroll
        for (int i = 0; i < M; i++) {
            const double f = a.s_final[(size_t)i * B + c];
            if (!is_nan(f)) Ss[i] = f;
        }
        load_sym<M>(a.P_PLUS, tT, lay, Ps);
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) {
                const double f = a.Ps_final[(size_t)IXM(i, j) * B + c];
                if (!is_nan(f)) Ps[sidx(i, j)] = f;
            }
        store_vec<M>(a.S_SMOOTH, tT, lay, Ss);
        store_sym<M>(a.P_SMOOTH, tT, lay, Ps);
        if (STOR) { store_rows_f32<M>(a.f.S_SMOOTH, tT, M, lay, Ss); store_sym_f32<M>(a.f.P_SMOOTH, tT, lay, Ps); }
        if (a.u_opt_smooth || (STOR && a.f.u_opt_smooth)) {
            double z[kNpi];
#pragma unroll
            for (int k = 0; k < kNpi; k++) z[k] = 0.0;
            store_u(a.u_opt_smooth, a, tT, lay, z);
            if (STOR) store_rows_f32<kNpi>(a.f.u_opt_smooth, tT, (unsigned)a.n_npi, lay, z);
        }
        if (a.pinv_rank) a.pinv_rank[lay_scalar(tT, lay)] = -1;
    }

    // Software pipeline.  Vector-memory operations retire in issue order, so loads issued after a step's
    // stores would wait for those stores to drain.  The results of a step are therefore kept in registers
    // (they are the recursion state anyway) and stored at the top of the NEXT iteration, right after that
    // iteration's loads have been issued.
    BwdIn<M> cur;
    // the inputs of a step in two groups: the small ones that the step needs first (state, controls, rank word) and the
    // two packed 6 x 6 (P_PLUS, X).  EPI_BWD_PREFETCH selects what is requested one step ahead (bit 0: the small group,
    // bit 1: P_PLUS, bit 2: X); whatever is not prefetched is loaded at the top of its own step.
    auto fetch_small = [&](int k, BwdIn<M> &d) {
        const int t = tpos<FLIP>(k, T), t1 = tpos<FLIP>(k + 1, T);
        load_vec<M>(a.S_PLUS, t, lay, d.Sp);
        load_u(a, t, su, d.u);
        d.rk = a.rankbuf[lay_scalar(t1, lay)];
    };
    auto fetch_pp = [&](int k, BwdIn<M> &d) { load_sym<M>(a.P_PLUS, tpos<FLIP>(k, T), lay, d.Pp); };
    // (unused garbage where the :211 guard fired, rk < 0)
    auto fetch_x = [&](int k, BwdIn<M> &d) {
        unsigned voff, rowb;
        const rsrc_t r = lay_slice(a.X, tpos<FLIP>(k + 1, T), NS, lay, voff, rowb);
#pragma unroll
        for (int e = 0; e < NS; e++) d.X[e] = bld_s(r, voff, (unsigned)e * rowb);
    };
    int t_pend = -1, rank_pend = -1;
    double u_pend[kNpi];
#pragma unroll
    for (int q = 0; q < kNpi; q++) u_pend[q] = 0.0;
    auto flush = [&]() {          // store the previous step's results (Ss, Ps still hold them)
        if (t_pend < 0) return;
        if (a.pinv_rank) a.pinv_rank[lay_scalar(t_pend, lay)] = rank_pend;
        store_vec<M>(a.S_SMOOTH, t_pend, lay, Ss);
        store_sym<M>(a.P_SMOOTH, t_pend, lay, Ps);
        if (a.u_opt_smooth) store_u(a.u_opt_smooth, a, t_pend, lay, u_pend);
        if (STOR) {
            store_rows_f32<M>(a.f.S_SMOOTH, t_pend, M, lay, Ss); store_sym_f32<M>(a.f.P_SMOOTH, t_pend, lay, Ps);
            store_rows_f32<kNpi>(a.f.u_opt_smooth, t_pend, (unsigned)a.n_npi, lay, u_pend);
        }
    };

    // double buffering: the inputs of step k-1 are requested at the top of step k (ahead of the stores of step k+1's
    // results) and consumed one iteration later, so that a lone wave does not sit through a full memory round trip
    // at the start of every step
    constexpr int PF = (M == 6) ? EPI_BWD_PREFETCH : 0;   // the 3-state kernel would drop from two waves per SIMD to one
    constexpr bool RC = (M == 3) || EPI_BWD_RECOMPUTE;    // ... and has the registers to recompute s(k+1|k), P(k+1|k)
    BwdIn<M> nxt;
    auto step = [&](int k) {
        const int t = tpos<FLIP>(k, T);
        if (PF & 1) { if (k > k_to) fetch_small(k - 1, nxt); } else fetch_small(k, cur);
        if (PF & 2) { if (k > k_to) fetch_pp(k - 1, nxt); } else fetch_pp(k, cur);
        if (PF & 4) { if (k > k_to) fetch_x(k - 1, nxt); } else fetch_x(k, cur);
        flush();
        // s(k+1|k) and P(k+1|k): read back at their point of use, or (EPI_BWD_RECOMPUTE) recomputed from the stored
        // s(k|k), P(k|k), u(:,k) with the forward kernel's own functions (:155-164) -- bit-identical either way
        double A[M * M], Sm1[M];
        state_jacobians<M, FLIP>(p, cur.u, cur.Sp, A);         // :206 (and :157 of the forward pass)
        if (RC) {
            double u_app[kNpi];
#pragma unroll
            for (int q = 0; q < kNpi; q++) u_app[q] = cur.u[q];
            nlin_state_update<M, FLIP>(p, a.mf, u_app, cur.Sp, Sm1);
            state_hard_margins<M>(p, Sm1);
        } else {
            load_vec<M>(a.S_MINUS, tpos<FLIP>(k + 1, T), lay, Sm1);
        }
        double J[M * M];
        int rank = -1;
        if (cur.rk < 0) {                                      // non-finite P_MINUS guard :211-213
#pragma unroll
            for (int e = 0; e < M * M; e++) J[e] = 0.0;
            st_guard = 1;
        } else {
            // J = (P+ A') X  :215, one row at a time (zeros of A skipped): row i of P+ A' is consumed into
            // row i of J at once, so the 6 x 6 product P+ A' is never live as a whole
#pragma unroll
            for (int i = 0; i < M; i++) {
                double PAr[M];
#pragma unroll
                for (int j = 0; j < M; j++) {
                    double acc = 0.0;
                    bool first = true;
#pragma unroll
                    for (int q = 0; q < M; q++)
                        if (a_nz<M>(j, q)) {
                            acc = first ? cur.Pp[sidx(i, q)] * A[IXM(j, q)] : fma(cur.Pp[sidx(i, q)], A[IXM(j, q)], acc);
                            first = false;
                        }
                    PAr[j] = acc;
                }
#pragma unroll
                for (int j = 0; j < M; j++) {
                    double acc = PAr[0] * cur.X[sidx(0, j)];
#pragma unroll
                    for (int q = 1; q < M; q++) acc = fma(PAr[q], cur.X[sidx(q, j)], acc);
                    J[IXM(i, j)] = acc;
                }
            }
            rank = cur.rk & 0xff;
            st_cap |= (cur.rk >> 8) & 1;
            min_rank = rank < min_rank ? rank : min_rank;
        }
        double Sn[M];
        {
            double dv[M];
#pragma unroll
            for (int i = 0; i < M; i++) dv[i] = Ss[i] - Sm1[i];
#pragma unroll
            for (int i = 0; i < M; i++) {
                double acc = J[IXM(i, 0)] * dv[0];
#pragma unroll
                for (int j = 1; j < M; j++) acc = fma(J[IXM(i, j)], dv[j], acc);
                Sn[i] = cur.Sp[i] + acc;                       // :218
            }
        }
        state_hard_margins<M>(p, Sn);                          // :221
        {
            // P_SMOOTH(k) = sym(P+ - (J D) J'),  D = P_MINUS(k+1) - P_SMOOTH(k+1)   :223-226
            // rows of J D are consumed one at a time and paired entries are averaged as soon as both exist
            // (see ekf_fwd_sym)
            // (P(k+1|k) is formed only now, after X has been consumed by J: the two are never live together)
            double Dsym[NS];
            if (RC) predict_cov_sym<M>(A, cur.Pp, Qd, Dsym);
            else load_sym<M>(a.P_MINUS, tpos<FLIP>(k + 1, T), lay, Dsym);
#pragma unroll
            for (int e = 0; e < NS; e++) Dsym[e] = Dsym[e] - Ps[e];
            double F[M * M];
#pragma unroll
            for (int i = 0; i < M; i++) {
                double T1r[M];
#pragma unroll
                for (int j = 0; j < M; j++) {
                    double acc = J[IXM(i, 0)] * Dsym[sidx(0, j)];
#pragma unroll
                    for (int q = 1; q < M; q++) acc = fma(J[IXM(i, q)], Dsym[sidx(q, j)], acc);
                    T1r[j] = acc;
                }
#pragma unroll
                for (int j = 0; j < M; j++) {
                    double acc = T1r[0] * J[IXM(j, 0)];
#pragma unroll
                    for (int q = 1; q < M; q++) acc = fma(T1r[q], J[IXM(j, q)], acc);
                    F[IXM(i, j)] = cur.Pp[sidx(i, j)] - acc;
                }
#pragma unroll
                for (int j = 0; j < i; j++) Ps[sidx(j, i)] = (F[IXM(i, j)] + F[IXM(j, i)]) / 2.0;
                Ps[sidx(i, i)] = (F[IXM(i, i)] + F[IXM(i, i)]) / 2.0;
            }
        }
#pragma unroll
        for (int i = 0; i < M; i++) Ss[i] = Sn[i];
#pragma unroll
        for (int q = 0; q < kNpi; q++) u_pend[q] = cur.u[q];
        if (a.u_opt_smooth || (STOR && a.f.u_opt_smooth)) {    // :229
            double sn_unused[M];
            nlin_state_update<M, FLIP>(p, a.mf, u_pend, Ss, sn_unused);
        }
        t_pend = t;
        rank_pend = rank;
        if (PF & 1) {
#pragma unroll
            for (int i = 0; i < M; i++) cur.Sp[i] = nxt.Sp[i];
#pragma unroll
            for (int q = 0; q < kNpi; q++) cur.u[q] = nxt.u[q];
            cur.rk = nxt.rk;
        }
        if (PF & 2) {
#pragma unroll
            for (int e = 0; e < NS; e++) cur.Pp[e] = nxt.Pp[e];
        }
        if (PF & 4) {
#pragma unroll
            for (int e = 0; e < NS; e++) cur.X[e] = nxt.X[e];
        }
    };
    if ((PF & 1) && k_from >= k_to) fetch_small(k_from, cur);
    if ((PF & 2) && k_from >= k_to) fetch_pp(k_from, cur);
    if ((PF & 4) && k_from >= k_to) fetch_x(k_from, cur);
    for (int k = k_from; k >= k_to; k--) step(k);
    flush();
    if (k_to > 0) {        // hand-over to the launch that continues with step k_to - 1
#pragma unroll
        for (int i = 0; i < M; i++) a.hand_s[(size_t)i * hp + c] = Ss[i];
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) a.hand_p[(size_t)IXM(i, j) * hp + c] = Ps[sidx(i, j)];
        a.hand_i[c] = st_guard | (st_cap << 1) | (min_rank << 8);
    } else if (a.status) a.status[c] = st_guard | (st_cap << 1) | (min_rank << 8);
}

