// ekf_lane6.hpp -- the one-lane-per-chain kernels of the 6-state generic models with FIXED array descriptors (round 6).
// Included by epiekf.hip inside namespace epi, after ekf_hex.hpp.
//
// Same arithmetic as ekf_fwd_sym<6> / eks_bwd_sym<6> (ekf_sym.hpp), operation for operation -- what changes is how the
// kernels ADDRESS memory and how they branch, i.e. the instructions a lone wave pays for without computing anything:
//   * every array has ONE buffer descriptor per addressing window (as many days as fit 2 GiB: ~96 at 75 000 chains), built
//     when the window begins; the day travels in the buffer instructions' 32-bit scalar offset, a handful of SGPRs carried
//     from day to day by additions (ekf_hex.hpp took the hex kernels from 74 / 115 to 20 scalar instructions a day this way;
//     eks_bwd_sym<6> spent 277 scalar instructions, 46 v_readlane of spilled kernel arguments and 35 s_nop a day);
//   * the layout block is a compile-time constant BLK (= the lanes a workgroup uses, what epi_ekf_preferred_lane_block
//     returns), so a row's offset is part immediate (< 4 KiB), part a multiple of 4 KiB added to the scalar offset: no scalar
//     register holds a row offset and none is multiplied out every day;
//   * an output the caller did not select has an EMPTY descriptor (record count 0): its stores are dropped by the bounds
//     check, no store sits behind a branch;
//   * the bang-bang substitution and the slope term of A(3,6) (OptControlled.m:49-58,107-114) test every control for NaN in
//     EVERY lane: 36 exec-mask branches a day.  A wave now asks once a day whether ANY of its lanes has a free control (the sum
//     of the twelve is NaN iff one of them is) and skips all of them on historic days with one scalar branch;
//   * RC = 1: the smoother does NOT read s(k+1|k), P(k+1|k) back (27 of its 75 loads a step, 8.4 GB of the headline pass):
//     P+ A', which the gain needs anyway, is (A P+)' bit for bit (P+ is symmetric bit for bit and a product commutes), so
//     P(k+1|k) = sym((A P+) A' + Q) costs one more 6 x 6 product with A's 21 non-zeros, not two -- the forward kernel's own
//     sequence of operations, the same bits (eks_bwd_hex does the same across six lanes).
// Conditions (enqueue_fwd / enqueue_bwd check them): R_v a per-day series (monitor hoisted), fixed diagonal Q_w, fp64
// storage, lane_block == lanes per workgroup == BLK.  Everything else keeps ekf_sym.hpp.
#pragma once

// EPI_LANE6_BWD: 0 = eks_bwd_sym<6> as before, 1 = eks_bwd_lane6 reading s(k+1|k), P(k+1|k) back, 2 = recomputing them (RC)
#ifndef EPI_LANE6_BWD
#define EPI_LANE6_BWD 2
#endif
#ifndef EPI_PROBE_NOFREE
#define EPI_PROBE_NOFREE 0      // counting probe: no lane ever has a free control (static instruction count of the historic-day path)
#endif
#ifndef EPI_LANE6_LATE_PF
#define EPI_LANE6_LATE_PF 1
#endif
#ifndef EPI_LANE6_BIG_PF
#define EPI_LANE6_BIG_PF 0      // bit 0: P(k-1|k-1), bit 1: X of step k - 1 are requested during step k, into the registers step k has finished with
#endif
#ifndef EPI_LANE6D_FLUSH_TOP
#define EPI_LANE6D_FLUSH_TOP 1   // eks_bwd_lane6d: the previous step's stores right after the step's wait (0: after P(k+1|k) has been formed)
#endif
#ifndef EPI_LANE6_PS_LDS
#define EPI_LANE6_PS_LDS 0
#endif
#ifndef EPI_LANE6_XD
#define EPI_LANE6_XD 1           // X of the next step by LDS-DMA where every lane of the launch is alive (see eks_bwd_lane6)
#endif
#ifndef EPI_LANE6_X_LATE
#define EPI_LANE6_X_LATE 0      // 1: X is requested after P(k+1|k) has been formed (behind the previous step's stores) instead of at the top
#endif
#ifndef EPI_LANE6_PP_LDS
#define EPI_LANE6_PP_LDS 1      // RC: P(k|k) waits in LDS between P+ A' and :223 instead of in registers
#endif
#ifndef EPI_LANE6_ST_AUX
#define EPI_LANE6_ST_AUX EPI_ST_AUX
#endif
#ifndef EPI_LANE6_PHASES
#define EPI_LANE6_PHASES 1
#endif
// the lane blocks these kernels are instantiated for: the balanced waves (balanced_lanes) of batches beyond one round of 64-lane
// waves -- 40 is the headline's; not 64 (enqueue_bwd says why)
constexpr bool lane6_block(int blk) { return blk == 40 || blk == 48 || blk == 56; }
constexpr unsigned kLwRecords = 0x7FFFFFF8u;       // just below 2 GiB: the bounds check includes the scalar offset (ekf_hex.hpp)

EPI_DEV rsrc_t lw_rsrc(const void *p) { return mk_rsrc(p, p ? kLwRecords : 0u); }
template <class P> EPI_DEV P *lw_rebase(P *p, int tw, size_t elems_per_day) { return p ? p + (size_t)tw * elems_per_day : nullptr; }

// this lane's byte offset in one day of the arrays of each row count (see Lay: block cb holds rows x BLK elements)
struct LwLane { unsigned v1w, v6, vn, v21, v36; };
template <int BLK> EPI_DEV LwLane lw_lane(unsigned c, unsigned n_npi)
{
    const unsigned cb = c / (unsigned)BLK, cr = c - cb * (unsigned)BLK;
    LwLane l;
    l.v1w = c * 4u;
    l.v6 = (cb * 6u * BLK + cr) * 8u; l.vn = (cb * n_npi * BLK + cr) * 8u;
    l.v21 = (cb * 21u * BLK + cr) * 8u; l.v36 = (cb * 36u * BLK + cr) * 8u;
    return l;
}
// byte offsets of one day relative to the window's base: one-row arrays of 4-byte words (doubles: twice that), 6, n_npi, 21 and
// 36 rows; carried from day to day by additions
struct LwDay { unsigned o1w, o6, on, o21, o36; };
EPI_DEV LwDay lw_stride(const KArgs &a, unsigned bp)
{
    LwDay s;
    s.o1w = bp * 4u; s.o6 = bp * 48u; s.on = bp * 8u * (unsigned)a.n_npi; s.o21 = bp * 168u; s.o36 = bp * 288u;
    return s;
}
EPI_DEV LwDay lw_day(const LwDay &s, int t)
{
    const unsigned tt = (unsigned)t;
    LwDay d;
    d.o1w = tt * s.o1w; d.o6 = tt * s.o6; d.on = tt * s.on; d.o21 = tt * s.o21; d.o36 = tt * s.o36;
    return d;
}
template <int DIR> EPI_DEV LwDay lw_next(const LwDay &d, const LwDay &s)
{
    LwDay n;
    if (DIR > 0) { n.o1w = d.o1w + s.o1w; n.o6 = d.o6 + s.o6; n.on = d.on + s.on; n.o21 = d.o21 + s.o21; n.o36 = d.o36 + s.o36; }
    else { n.o1w = d.o1w - s.o1w; n.o6 = d.o6 - s.o6; n.on = d.on - s.on; n.o21 = d.o21 - s.o21; n.o36 = d.o36 - s.o36; }
    return n;
}
// days per addressing window: three days of the widest array short of the record count (a step touches its own day, the day
// after it in filter order and -- the prefetch -- the day before)
EPI_DEV int lw_window(const KArgs &a, unsigned bp)
{
    int w = (int)(kLwRecords / (bp * 288u)) - 3;
    if (a.hexw >= 2 && a.hexw < w) w = a.hexw;         // the test knob can only shorten the window
    return w < 2 ? 2 : w;
}
// row `row` (a compile-time constant once the loops are unrolled) of a BLK-blocked array: the part of its offset below 4 KiB
// goes into the instruction's immediate, the rest into the scalar offset
template <int BLK> EPI_DEV void lw_st(rsrc_t r, unsigned vo, unsigned so, int row, double v)
{
    const unsigned b = (unsigned)row * (unsigned)BLK * 8u;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, vo + (b & 4095u), so + (b & ~4095u), EPI_LANE6_ST_AUX);
}
template <int BLK> EPI_DEV double lw_ld(rsrc_t r, unsigned vo, unsigned so, int row)
{
    const unsigned b = (unsigned)row * (unsigned)BLK * 8u;
    return bld_s(r, vo + (b & 4095u), so + (b & ~4095u));
}

// the phases of a step (Jacobian and state map | P+ A' | P(k+1|k) | J | the recursion) are kept apart: hipcc otherwise hoists
// the later phases' operands above the earlier ones' arithmetic and the step's peak register demand rises by ~100
EPI_DEV void lw_phase()
{
#if EPI_LANE6_PHASES
    __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef EPI_LANE6_MARK
    asm volatile("; L6PHASE");
#endif
}
// `buffer_load_dwordx4 ... lds` (gfx950: 16 bytes per lane): global memory straight into LDS at a wave-uniform LDS address + 16 x lane,
// no register destination
typedef __attribute__((address_space(3))) void *lds_ptr_t;
// (the LDS destination is named by its 32-bit LDS address: a generic pointer cast back to LDS costs a null check per use)
EPI_DEV void lw_dma16(rsrc_t r, unsigned lds_addr, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(uintptr_t)lds_addr, 16, voff, soff, 0, EPI_LD_STREAM_AUX);
}
// an LDS address the compiler cannot see through: what was written there is READ BACK, not kept in registers beside it
// (hipcc forwards a store to a later load of the same LDS address -- and the value then stays live, which is what the
// LDS copy is there to avoid)
// (the LANE INDEX is hidden, not the pointer: a pointer that went through an asm statement has lost its address space and
// is read with flat_load instead of ds_read)
EPI_DEV int lw_opaque(int lane)
{
    asm volatile("" : "+v"(lane));
    return lane;
}
// the four 12-vectors of `params` in an LDS column per lane with a compile-time lane stride (see VecLds)
template <int STR> struct VecLdsT {
    const double *base;
    EPI_DEV double A(int k) const { return base[(0 * kNpi + k) * STR]; }
    EPI_DEV double Umin(int k) const { return base[(1 * kNpi + k) * STR]; }
    EPI_DEV double Umax(int k) const { return base[(2 * kNpi + k) * STR]; }
    EPI_DEV double W(int k) const { return base[(3 * kNpi + k) * STR]; }
};

// "does any lane of this wave hold a free (NaN) control today?"  x + NaN is NaN for every x, so the sum of the twelve is NaN
// whenever one of them is (Inf - Inf can only err on the side of the slow path, which is always right)
EPI_DEV bool lw_wave_has_free_control(const double (&u)[kNpi])
{
    const double s = ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7])) + ((u[8] + u[9]) + (u[10] + u[11]));
    return __builtin_amdgcn_ballot_w64(is_nan(s)) != 0ull;
}

// P(k+1|k) = sym(A P+ A' + Q) from PA = P+ A' (row-major: PA[6 q + i] = (P+ A')(q, i) == (A P+)(i, q)): predict_cov_sym's second
// product, its + Q and its symmetrisation, term for term
EPI_DEV void lw_predict_cov_from_pa(const double (&A)[36], const double (&PA)[36], const double (&Qd)[6], double (&Pm)[21])
{
    constexpr int M = 6;
    double G[36];
#pragma unroll
    for (int i = 0; i < M; i++) {
#pragma unroll
        for (int j = 0; j < M; j++) {
            double acc = 0.0;
            bool first = true;
#pragma unroll
            for (int q = 0; q < M; q++)
                if (a_nz<M>(j, q)) {
                    acc = first ? PA[6 * q + i] * A[IXM(j, q)] : fma(PA[6 * q + i], A[IXM(j, q)], acc);
                    first = false;
                }
            G[IXM(i, j)] = acc + ((i == j) ? Qd[i] : 0.0);
        }
#pragma unroll
        for (int j = 0; j < i; j++) Pm[sidx(j, i)] = (G[IXM(i, j)] + G[IXM(j, i)]) / 2.0;
        Pm[sidx(i, i)] = (G[IXM(i, i)] + G[IXM(i, i)]) / 2.0;
    }
}

// ---------------------------------------------------------------------------
// backward recursion (GenericExtendedKalmanFilter.m:204-230; X = pinv(P_MINUS) comes from eks_pinv)
// ---------------------------------------------------------------------------
// LATE_PF: the next step's small inputs are requested in the middle of the step instead of at its top
// XD = 1 (launches whose chain count is a multiple of BLK: every lane of every workgroup is alive): X of the NEXT step comes by
// LDS-DMA (lw_dma16) while this step runs -- 21 packed rows of BLK x 8 bytes, two rows per instruction of the wave's BLK lanes -- into
// an LDS image that the step reads when it forms J.  X is the one input that is requested long before its use (at the top, ahead of
// the stores; used after P+ A' and P(k+1|k)): in registers it is parked in accumulation registers and fetched back, 168 of the
// step's 373 such moves.
template <int FLIP, int BLK, int RC, int LATE_PF = EPI_LANE6_LATE_PF, int XD = 0>
__global__ __launch_bounds__(kWave) void eks_bwd_lane6(const KArgs a, const int *__restrict__ dense_flag)
{
    __shared__ __attribute__((aligned(16))) double s_xd[XD ? 21 * BLK : 2];      // DMA image of X (a separate object: the compiler waits for a DMA only where this array is read)
    constexpr int M = 6, NS = 21;
    // one column per lane, BLK lanes: a, u_min, u_max, w (48 rows); the pending u_opt_smooth (12) and -- RC -- P(k|k) between its
    // use in P+ A' and in :223 (21): values that would otherwise sit in accumulation registers and cost four moves each
    __shared__ double vlds[(4 * kNpi + kNpi + (RC ? 21 + 6 : 0) + (EPI_LANE6_PS_LDS ? 21 : 0)) * BLK];
    if (*dense_flag) return;
    const unsigned xd_base = (unsigned)(uintptr_t)(lds_ptr_t)s_xd;
    const int lane = threadIdx.x;
    const int c = a.c0 + blockIdx.x * BLK + lane;
    if (lane >= BLK || c >= a.c0 + a.cn) return;
    double *const l_upend = vlds + 4 * kNpi * BLK + lane, *const l_pp = vlds + 5 * kNpi * BLK + lane, *const l_q = l_pp + 21 * BLK;
    double *const l_ps = vlds + (5 * kNpi + (RC ? 27 : 0)) * BLK + lane;     // P_SMOOTH(k+1) between the step that forms it and the next step's stores / :223
    const int B = a.B, T = a.T;
    const int su = a.u_series ? a.u_series[c] : c;
    const unsigned bp = (unsigned)a.blk * (unsigned)a.nblk;
    const LwLane ll = lw_lane<BLK>((unsigned)c, (unsigned)a.n_npi);
    LitePrm<VecLdsT<BLK>> p;
    load_lite(p, a.prm, B, c, a.mf.lo_is_zero);
    p.v.base = vlds + lane;
#pragma unroll
    for (int k = 0; k < kNpi; k++) {
        vlds[(0 * kNpi + k) * BLK + lane] = a.prm[(size_t)(EPI_PRM_A + k) * B + c];
        vlds[(1 * kNpi + k) * BLK + lane] = a.prm[(size_t)(EPI_PRM_U_MIN + k) * B + c];
        vlds[(2 * kNpi + k) * BLK + lane] = a.prm[(size_t)(EPI_PRM_U_MAX + k) * B + c];
        vlds[(3 * kNpi + k) * BLK + lane] = a.prm[(size_t)(EPI_PRM_W_EFF + k) * B + c];
    }
    if (RC) {
#pragma unroll
        for (int i = 0; i < M; i++) l_q[i * BLK] = a.Q[(size_t)IXM(i, i) * B + c];      // Q_w diagonal (ekf_precheck)
    }

    const int k_from = a.bk_from, k_to = a.bk_to;      // smoother steps of this launch, see eks_bwd_sym
    const size_t hp = (size_t)a.hand_pitch;
    int st_guard = 0, st_cap = 0, min_rank = M;
    double Ss[M], Ps[NS];
    // EPI_LANE6_PS_LDS: the loop-carried P_SMOOTH is cold for most of a step (used by the stores and by :223 only); hipcc parks such
    // values in accumulation registers and pays four moves per double and use -- here it waits in LDS: a write, two reads
    auto ps_park = [&]() __attribute__((always_inline)) {
        if (EPI_LANE6_PS_LDS) {
#pragma unroll
            for (int e = 0; e < NS; e++) l_ps[e * BLK] = Ps[e];
        }
    };
    auto ps_fetch = [&]() __attribute__((always_inline)) {
        if (EPI_LANE6_PS_LDS) {
            const double *q = vlds + (5 * kNpi + (RC ? 27 : 0)) * BLK + lw_opaque(lane);
#pragma unroll
            for (int e = 0; e < NS; e++) Ps[e] = q[e * BLK];
        }
    };

    // the arrays of the current addressing window: descriptors with their bases at its first day `tw`
    rsrc_t rSp, rPp, rX, rSm, rPm, rRank, rSs, rPs, rUs, rPr;
    int tw = 0;
    auto rebase = [&](int t0) __attribute__((always_inline)) {
        tw = t0;
        const size_t bpl = bp;
        rSp = lw_rsrc(lw_rebase(a.S_PLUS, t0, 6 * bpl)); rPp = lw_rsrc(lw_rebase(a.P_PLUS, t0, 36 * bpl));
        rX = lw_rsrc(lw_rebase(a.X, t0, 21 * bpl)); rRank = lw_rsrc(lw_rebase(a.rankbuf, t0, bpl));
        if (!RC) { rSm = lw_rsrc(lw_rebase(a.S_MINUS, t0, 6 * bpl)); rPm = lw_rsrc(lw_rebase(a.P_MINUS, t0, 36 * bpl)); }
        rSs = lw_rsrc(lw_rebase(a.S_SMOOTH, t0, 6 * bpl)); rPs = lw_rsrc(lw_rebase(a.P_SMOOTH, t0, 36 * bpl));
        rUs = lw_rsrc(lw_rebase(a.u_opt_smooth, t0, (size_t)a.n_npi * bpl)); rPr = lw_rsrc(lw_rebase(a.pinv_rank, t0, bpl));
    };
    auto st_vec = [&](rsrc_t r, const LwDay &d, const double (&v)[M]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < M; i++) lw_st<BLK>(r, ll.v6, d.o6, i, v[i]);
    };
    auto st_sym = [&](rsrc_t r, const LwDay &d, const double (&P)[NS]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i < M; i++) lw_st<BLK>(r, ll.v36, d.o36, IXM(i, j), P[sidx(i, j)]);
    };
    auto st_u = [&](rsrc_t r, const LwDay &d, const double (&u)[kNpi]) __attribute__((always_inline)) {
        if (a.n_npi == kNpi) {
#pragma unroll
            for (int k = 0; k < kNpi; k++) lw_st<BLK>(r, ll.vn, d.on, k, u[k]);
        } else {
#pragma unroll
            for (int k = 0; k < kNpi; k++)
                if (k < a.n_npi) lw_st<BLK>(r, ll.vn, d.on, k, u[k]);
        }
    };
    auto st_word = [&](rsrc_t r, const LwDay &d, int v) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b32(v, r, ll.v1w, d.o1w, 0);
    };
    auto ld_sym = [&](rsrc_t r, const LwDay &d, double (&P)[NS]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) P[sidx(i, j)] = lw_ld<BLK>(r, ll.v36, d.o36, IXM(i, j));
    };
    auto ld_vec = [&](rsrc_t r, const LwDay &d, double (&v)[M]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < M; i++) v[i] = lw_ld<BLK>(r, ll.v6, d.o6, i);
    };

    // X of one step by DMA: this lane's 16-byte piece of two layout rows per instruction (the eleventh moves the last row alone)
    const unsigned vxd = ((unsigned)c / BLK) * 21u * BLK * 8u + (unsigned)lane * 16u;
    auto dma_x = [&](const LwDay &dd) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 10; i++) lw_dma16(rX, xd_base + (unsigned)(2 * i * BLK * 8), vxd, dd.o21 + (unsigned)(2 * i * BLK * 8));
        if (lane < BLK / 2) lw_dma16(rX, xd_base + (unsigned)(20 * BLK * 8), vxd, dd.o21 + (unsigned)(20 * BLK * 8));
    };
    if (k_from < T - 2) {      // resume from the hand-over rows
#pragma unroll
        for (int i = 0; i < M; i++) Ss[i] = a.hand_s[(size_t)i * hp + c];
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) Ps[sidx(i, j)] = a.hand_p[(size_t)IXM(i, j) * hp + c];
        const int word = a.hand_i[c];
        st_guard = word & 1; st_cap = (word >> 1) & 1; min_rank = word >> 8;
    }
    const LwDay ds = lw_stride(a, bp);
    // terminal conditions GenericEKF.m:189-202 (Ps_final symmetric in values and NaN pattern: ekf_precheck)
    auto terminal = [&]() __attribute__((always_inline)) {
        const LwDay dT = lw_day(ds, tpos<FLIP>(T - 1, T) - tw);
        ld_vec(rSp, dT, Ss);
#pragma unroll
        for (int i = 0; i < M; i++) {
            const double f = a.s_final[(size_t)i * B + c];
            if (!is_nan(f)) Ss[i] = f;
        }
        ld_sym(rPp, dT, Ps);
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) {
                const double f = a.Ps_final[(size_t)IXM(i, j) * B + c];
                if (!is_nan(f)) Ps[sidx(i, j)] = f;
            }
        st_vec(rSs, dT, Ss);
        st_sym(rPs, dT, Ps);
        double z[kNpi];
#pragma unroll
        for (int k = 0; k < kNpi; k++) z[k] = 0.0;
        st_u(rUs, dT, z);                                  // column T is never written :95,204
        st_word(rPr, dT, -1);
    };

    // what step k reads, in two groups: the small ones the step needs first, requested one step ahead (state, controls, rank
    // word), and the two packed 6 x 6 (P_PLUS, X), requested at the top of their own step, ahead of the previous step's stores
    struct Small { double Sp[M], u[kNpi]; int rk; };
    Small cur, nxt;
    double Pp[NS], X[NS];
    // (dt: the day of step k, tpos(k); dt1: the day after it in filter order, tpos(k + 1))
    auto fetch_small = [&](const LwDay &dt, const LwDay &dt1, int t_abs, Small &d) __attribute__((always_inline)) {
        ld_vec(rSp, dt, d.Sp);
        load_u(a, t_abs, su, d.u);
        d.rk = (int)__builtin_amdgcn_raw_buffer_load_b32(rRank, ll.v1w, dt1.o1w, 0);
    };
    LwDay d_pend = lw_day(ds, 0);
    bool have_pend = false;
    int rank_pend = -1;
    auto flush = [&]() __attribute__((always_inline)) {          // store the previous step's results (Ss, Ps still hold them)
        st_word(rPr, d_pend, rank_pend);
        st_vec(rSs, d_pend, Ss);
        ps_fetch();
        st_sym(rPs, d_pend, Ps);
        double u_pend[kNpi];
        const double *lu = vlds + 4 * kNpi * BLK + lw_opaque(lane);
#pragma unroll
        for (int q = 0; q < kNpi; q++) u_pend[q] = lu[q * BLK];
        st_u(rUs, d_pend, u_pend);
    };

    auto step = [&](int k, const LwDay &d0, const LwDay &d1) __attribute__((always_inline)) {
        if (!LATE_PF && k > k_to) fetch_small(lw_next<FLIP ? 1 : -1>(d0, ds), d0, tpos<FLIP>(k - 1, T), nxt);
        if (!(EPI_LANE6_BIG_PF & 1)) ld_sym(rPp, d0, Pp);
        if (!XD && !EPI_LANE6_X_LATE && !(EPI_LANE6_BIG_PF & 2)) {
#pragma unroll
            for (int e = 0; e < NS; e++) X[e] = lw_ld<BLK>(rX, ll.v21, d1.o21, e);      // (garbage where the :211 guard fired, rk < 0: unused)
        }
        double Sm1[M], Dsym[NS];
        if (!RC) {
            ld_vec(rSm, d1, Sm1);
            ld_sym(rPm, d1, Dsym);
        }
        if (have_pend) flush();
        // the controls as they came are what u_opt_smooth holds wherever they are not free (:229): parked now, so that they
        // need not stay in registers until the end of the step
#pragma unroll
        for (int q = 0; q < kNpi; q++) l_upend[q * BLK] = cur.u[q];

        // A = StateJacobians(u, s+) :206; the slope term and the bang-bang substitution only where some lane has a free control
        const bool wave_free = EPI_PROBE_NOFREE ? false : lw_wave_has_free_control(cur.u);
        double A[M * M];
        jacobian_entries<M, FLIP>(p, cur.Sp, wave_free ? slope_term<M, FLIP>(p, cur.u, cur.Sp) : 0.0, A);
        if (RC) {              // s(k+1|k) = StateHardMargins(NlinStateUpdate(u, s+)) :155,164
            double u_app[kNpi];
#pragma unroll
            for (int q = 0; q < kNpi; q++) u_app[q] = cur.u[q];
            if (wave_free) resolve_control<M>(p, a.mf, cur.Sp, u_app);
            double dot = (p.gamma * p.A(0)) * (p.Umax(0) - u_app[0]);
#pragma unroll
            for (int q = 1; q < kNpi; q++) dot = fma(p.gamma * p.A(q), p.Umax(q) - u_app[q], dot);
            state_map<M, FLIP>(p, dot, cur.Sp, Sm1);
            state_hard_margins<M>(p, Sm1);
        }
        lw_phase();
        // PA = P+ A' :215 (zeros of A skipped): row i of it is also column i of A P+
        double PA[M * M];
#pragma unroll
        for (int i = 0; i < M; i++)
#pragma unroll
            for (int j = 0; j < M; j++) {
                double acc = 0.0;
                bool first = true;
#pragma unroll
                for (int q = 0; q < M; q++)
                    if (a_nz<M>(j, q)) {
                        acc = first ? Pp[sidx(i, q)] * A[IXM(j, q)] : fma(Pp[sidx(i, q)], A[IXM(j, q)], acc);
                        first = false;
                    }
                PA[6 * i + j] = acc;
            }
        if (RC) {
            if (EPI_LANE6_PP_LDS) {
#pragma unroll
                for (int e = 0; e < NS; e++) l_pp[e * BLK] = Pp[e];
            }
            lw_phase();
            // P(k|k) is in LDS now: its registers take P(k-1|k-1) for the next step
            if ((EPI_LANE6_BIG_PF & 1) && k > k_to) ld_sym(rPp, lw_next<FLIP ? 1 : -1>(d0, ds), Pp);
            double Qd[M];
            const double *lq = vlds + (5 * kNpi + 21) * BLK + lw_opaque(lane);
#pragma unroll
            for (int i = 0; i < M; i++) Qd[i] = lq[i * BLK];
            lw_predict_cov_from_pa(A, PA, Qd, Dsym);
        }
        ps_fetch();
#pragma unroll
        for (int e = 0; e < NS; e++) Dsym[e] = Dsym[e] - Ps[e];        // D = P_MINUS(k+1) - P_SMOOTH(k+1)  :223
        lw_phase();
        // the next step's small inputs: requested only now (they return behind this step's X in any case, and are not needed
        // before the next step), so that they do not hold 37 registers through the phases above
        if (!XD && EPI_LANE6_X_LATE && !(EPI_LANE6_BIG_PF & 2)) {
#pragma unroll
            for (int e = 0; e < NS; e++) X[e] = lw_ld<BLK>(rX, ll.v21, d1.o21, e);
        }
        if (XD) {              // the image the previous step (or the prologue) requested.  It HAS landed: at least 99 vector-memory operations (this
            // step's 21 loads of P(k|k), the 78 stores of the previous step's results) were issued behind its DMA, a wave has at most 64 in
            // flight and they complete in order -- and the step loop ends with a full wait for the small inputs requested before it
#pragma unroll
            for (int e = 0; e < NS; e++) X[e] = s_xd[e * BLK + lane];
        }
        if (LATE_PF && k > k_to) fetch_small(lw_next<FLIP ? 1 : -1>(d0, ds), d0, tpos<FLIP>(k - 1, T), nxt);
        double J[M * M];
        int rank = -1;
        if (cur.rk < 0) {                                      // non-finite P_MINUS guard :211-213
#pragma unroll
            for (int e = 0; e < M * M; e++) J[e] = 0.0;
            st_guard = 1;
        } else {
#pragma unroll
            for (int i = 0; i < M; i++)
#pragma unroll
                for (int j = 0; j < M; j++) {
                    double acc = PA[6 * i] * X[sidx(0, j)];
#pragma unroll
                    for (int q = 1; q < M; q++) acc = fma(PA[6 * i + q], X[sidx(q, j)], acc);
                    J[IXM(i, j)] = acc;
                }
            rank = cur.rk & 0xff;
            st_cap |= (cur.rk >> 8) & 1;
            min_rank = rank < min_rank ? rank : min_rank;
        }
        lw_phase();
        if (XD && k > k_to) {                                  // X has been consumed: the image takes the next step's (X of step k - 1 lies at the position of step k)
            __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): the reads of the image are done before it is overwritten
            dma_x(d0);
        }
        if ((EPI_LANE6_BIG_PF & 2) && k > k_to) {             // X has been consumed: its registers take the next step's
#pragma unroll
            for (int e = 0; e < NS; e++) X[e] = lw_ld<BLK>(rX, ll.v21, d0.o21, e);
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
#pragma unroll
        for (int i = 0; i < M; i++) Ss[i] = Sn[i];
        {
            // P_SMOOTH(k) = sym(P+ - (J D) J')   :223-226, rows of J D consumed one at a time (see eks_bwd_sym)
            double F[M * M];
            const double *lp = vlds + 5 * kNpi * BLK + lw_opaque(lane);
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
                    F[IXM(i, j)] = ((RC && EPI_LANE6_PP_LDS) ? lp[sidx(i, j) * BLK] : Pp[sidx(i, j)]) - acc;
                }
#pragma unroll
                for (int j = 0; j < i; j++) Ps[sidx(j, i)] = (F[IXM(i, j)] + F[IXM(j, i)]) / 2.0;
                Ps[sidx(i, i)] = (F[IXM(i, i)] + F[IXM(i, i)]) / 2.0;
            }
        }
        ps_park();
        if (wave_free) {       // :229 -- only the control NlinStateUpdate returns is kept: the free ones, resolved with the smoothed costate
            double u_pend[kNpi];
            const double *lu = vlds + 4 * kNpi * BLK + lw_opaque(lane);
#pragma unroll
            for (int q = 0; q < kNpi; q++) u_pend[q] = lu[q * BLK];
            resolve_control<M>(p, a.mf, Ss, u_pend);
#pragma unroll
            for (int q = 0; q < kNpi; q++) l_upend[q * BLK] = u_pend[q];
        }
        d_pend = d0;
        have_pend = true;
        rank_pend = rank;
#pragma unroll
        for (int i = 0; i < M; i++) cur.Sp[i] = nxt.Sp[i];
#pragma unroll
        for (int q = 0; q < kNpi; q++) cur.u[q] = nxt.u[q];
        cur.rk = nxt.rk;
    };

    // One addressing window per pass of the outer loop: steps k = hi ... lo; it touches the days of steps lo - 1 (the prefetch)
    // ... hi + 1, and its base is the earliest of them.  Step k - 1 lies one day EARLIER in filter order (later in the array
    // when FLIP).
    constexpr int DIR = FLIP ? 1 : -1;
    const int W = lw_window(a, bp);
    int k = k_from;
    bool first = true;
    if (k_from >= T - 2 && k_from < k_to) {      // T == 1: no step, the terminal condition alone
        rebase(tpos<FLIP>(T - 1, T));
        terminal();
    }
    while (k >= k_to) {
        const int hi = k, lo = (hi - k_to >= W) ? hi - W + 1 : k_to;
        if (!first && have_pend) { flush(); have_pend = false; }      // the last step's results belong to the window that ends
        rebase(FLIP ? tpos<FLIP>(hi + 1, T) : (lo > 0 ? lo - 1 : 0));
        LwDay d1 = lw_day(ds, tpos<FLIP>(hi + 1, T) - tw), d0 = lw_day(ds, tpos<FLIP>(hi, T) - tw);
        if (first) {
            if (k_from >= T - 2) terminal();
            ps_park();
            fetch_small(d0, d1, tpos<FLIP>(hi, T), cur);
            if (XD) dma_x(d1);
            if (EPI_LANE6_BIG_PF & 1) ld_sym(rPp, d0, Pp);
            if (EPI_LANE6_BIG_PF & 2) {
#pragma unroll
                for (int e = 0; e < NS; e++) X[e] = lw_ld<BLK>(rX, ll.v21, d1.o21, e);
            }
            __builtin_amdgcn_s_waitcnt(0);       // the loop must not inherit the prologue's pending loads (see ekf_fwd_hex)
            first = false;
        }
        for (; k >= lo; k--) {
            step(k, d0, d1);
            d1 = d0;
            d0 = lw_next<DIR>(d0, ds);
        }
    }
    if (have_pend) flush();
    if (XD) __builtin_amdgcn_s_waitcnt(0x0070);      // (no DMA is in flight when the wave ends: the last step requests none)
    ps_fetch();
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

#if EPI_LANE6_BWD == 3      // measured and not adopted: built only on request
// ---------------------------------------------------------------------------
// backward recursion, inputs through LDS-DMA (round 6)
// ---------------------------------------------------------------------------
// eks_bwd_lane6 (RC) asks for P(k|k) and X at the top of step k and needs the first of them ~150 instructions later: a lone
// wave then sits through a memory round trip every step -- and through the completion of half of the 78 stores it issued in
// between, since loads and stores share ONE in-order counter of 6 bits (`s_waitcnt vmcnt(N)`: all but the N <= 63 youngest).
// Requesting them a step ahead into registers does not fit (512 registers are in use: spills inside the loop).  Here they
// never touch a register while in flight: `buffer_load_dwordx4 ... lds` (gfx950: 16 bytes per lane) copies the 21 packed rows
// of each straight into an LDS image [row][lane] -- a layout row of BLK = 40 chains is 320 contiguous bytes = 20 lanes' pieces,
// so one instruction of the wave's 40 lanes fills two rows -- and the step reads them with ds_read_b64 when it gets there.
// Per step:  wait (everything requested during the previous step)  |  Jacobian, state map  |  P+ A' from the LDS image (copied
// to a second LDS array for :223)  |  P(k+1|k)  |  the previous step's 78 STORES  |  requests for step k - 1: state, controls,
// rank word (registers), P(k-1|k-1) (LDS)  |  J from the X image  |  request X of step k - 1 (LDS)  |  :218-226.
// The loads are younger than the stores, so the one wait of a step is a plain vmcnt(0) -- about two thirds of a step after the
// stores and half a step after the loads were issued.
// packed row e = sidx(i, j) of a symmetric 6 x 6 array stored with all 36 rows: its row i + 6 j
constexpr int lw_src_row(int e)
{
    int j = 0;
    while ((j + 1) * (j + 2) / 2 <= e) j++;
    return (e - j * (j + 1) / 2) + 6 * j;
}

template <int FLIP, int BLK>
__global__ __launch_bounds__(kWave) void eks_bwd_lane6d(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int M = 6, NS = 21;
    static_assert(BLK % 2 == 0 && BLK <= 64 && (BLK * 8) % 16 == 0, "a layout row is BLK / 2 sixteen-byte pieces");
    constexpr int HALF = BLK / 2;               // lanes (16-byte pieces) per layout row
    // separate LDS objects, so that the compiler can tell a DMA target from the arrays it must not wait for
    __shared__ double s_prm[4 * kNpi * BLK];    // a, u_min, u_max, w: one column per lane
    __shared__ double s_upend[kNpi * BLK];      // the pending u_opt_smooth
    __shared__ double s_pp[NS * BLK];           // P(k|k) between P+ A' and :223
    __shared__ __attribute__((aligned(16))) double s_ppd[NS * BLK];   // DMA image of P(k|k), packed upper triangle
    __shared__ __attribute__((aligned(16))) double s_xd[NS * BLK];    // DMA image of X
    if (*dense_flag) return;
    const unsigned xd_base = (unsigned)(uintptr_t)(lds_ptr_t)s_xd, ppd_base = (unsigned)(uintptr_t)(lds_ptr_t)s_ppd;
    const int lane = threadIdx.x;
    const int c = a.c0 + blockIdx.x * BLK + lane;
    if (lane >= BLK) return;
    // (a lane beyond the batch's last chain keeps running: it moves its share of every DMA, computes on chain cn - 1 and stores nothing)
    const bool live = c < a.c0 + a.cn;
    const int cc = live ? c : a.c0 + a.cn - 1;
    const unsigned dead = live ? 0u : 0x80000000u;          // OR-ed into every store offset: beyond the record count
    const int B = a.B, T = a.T;
    const int su = a.u_series ? a.u_series[cc] : cc;
    const unsigned bp = (unsigned)a.blk * (unsigned)a.nblk;
    const LwLane ll = lw_lane<BLK>((unsigned)c, (unsigned)a.n_npi);
    // DMA source offsets: this lane's 16-byte piece of a layout row of its block; for P(k|k), whose 21 packed rows are not
    // contiguous among the 36 stored, the second row of an instruction lies D rows after the first (D = 1, 2, 5, 6)
    const unsigned cb = (unsigned)c / BLK, piece = (unsigned)(lane % HALF) * 16u, second = lane >= HALF ? 1u : 0u;
    const unsigned vxd = cb * 21u * BLK * 8u + (unsigned)lane * 16u;
    const unsigned vpd1 = cb * 36u * BLK * 8u + piece + second * (1u * BLK * 8u), vpd2 = cb * 36u * BLK * 8u + piece + second * (2u * BLK * 8u);
    const unsigned vpd5 = cb * 36u * BLK * 8u + piece + second * (5u * BLK * 8u), vpd6 = cb * 36u * BLK * 8u + piece + second * (6u * BLK * 8u);

    LitePrm<VecLdsT<BLK>> p;
    load_lite(p, a.prm, B, cc, a.mf.lo_is_zero);
    p.v.base = s_prm + lane;
#pragma unroll
    for (int k = 0; k < kNpi; k++) {
        s_prm[(0 * kNpi + k) * BLK + lane] = a.prm[(size_t)(EPI_PRM_A + k) * B + cc];
        s_prm[(1 * kNpi + k) * BLK + lane] = a.prm[(size_t)(EPI_PRM_U_MIN + k) * B + cc];
        s_prm[(2 * kNpi + k) * BLK + lane] = a.prm[(size_t)(EPI_PRM_U_MAX + k) * B + cc];
        s_prm[(3 * kNpi + k) * BLK + lane] = a.prm[(size_t)(EPI_PRM_W_EFF + k) * B + cc];
    }
    double Qd[M];
#pragma unroll
    for (int i = 0; i < M; i++) Qd[i] = a.Q[(size_t)IXM(i, i) * B + cc];      // Q_w diagonal (ekf_precheck)

    const int k_from = a.bk_from, k_to = a.bk_to;      // smoother steps of this launch, see eks_bwd_sym
    const size_t hp = (size_t)a.hand_pitch;
    int st_guard = 0, st_cap = 0, min_rank = M;
    double Ss[M], Ps[NS];

    rsrc_t rSp, rPp, rX, rRank, rSs, rPs, rUs, rPr;
    int tw = 0;
    auto rebase = [&](int t0) __attribute__((always_inline)) {
        tw = t0;
        const size_t bpl = bp;
        rSp = lw_rsrc(lw_rebase(a.S_PLUS, t0, 6 * bpl)); rPp = lw_rsrc(lw_rebase(a.P_PLUS, t0, 36 * bpl));
        rX = lw_rsrc(lw_rebase(a.X, t0, 21 * bpl)); rRank = lw_rsrc(lw_rebase(a.rankbuf, t0, bpl));
        rSs = lw_rsrc(lw_rebase(a.S_SMOOTH, t0, 6 * bpl)); rPs = lw_rsrc(lw_rebase(a.P_SMOOTH, t0, 36 * bpl));
        rUs = lw_rsrc(lw_rebase(a.u_opt_smooth, t0, (size_t)a.n_npi * bpl)); rPr = lw_rsrc(lw_rebase(a.pinv_rank, t0, bpl));
    };
    auto st_vec = [&](rsrc_t r, const LwDay &d, const double (&v)[M]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < M; i++) lw_st<BLK>(r, ll.v6 | dead, d.o6, i, v[i]);
    };
    auto st_sym = [&](rsrc_t r, const LwDay &d, const double (&P)[NS]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i < M; i++) lw_st<BLK>(r, ll.v36 | dead, d.o36, IXM(i, j), P[sidx(i, j)]);
    };
    auto st_u = [&](rsrc_t r, const LwDay &d, const double (&u)[kNpi]) __attribute__((always_inline)) {
        if (a.n_npi == kNpi) {
#pragma unroll
            for (int k = 0; k < kNpi; k++) lw_st<BLK>(r, ll.vn | dead, d.on, k, u[k]);
        } else {
#pragma unroll
            for (int k = 0; k < kNpi; k++)
                if (k < a.n_npi) lw_st<BLK>(r, ll.vn | dead, d.on, k, u[k]);
        }
    };
    auto st_word = [&](rsrc_t r, const LwDay &d, int v) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b32(v, r, ll.v1w | dead, d.o1w, 0);
    };
    // the two DMA requests: 21 rows each, two rows per instruction, the eleventh moves the last row alone (lanes of its first half)
    auto dma_x = [&](const LwDay &d1) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 10; i++) lw_dma16(rX, xd_base + (unsigned)(2 * i * BLK * 8), vxd, d1.o21 + (unsigned)(2 * i * BLK * 8));
        if (lane < HALF) lw_dma16(rX, xd_base + (unsigned)(20 * BLK * 8), vxd, d1.o21 + (unsigned)(20 * BLK * 8));
    };
    auto dma_pp = [&](const LwDay &d) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int ra = lw_src_row(2 * i), rb = lw_src_row(2 * i + 1), D = rb - ra;
            const unsigned vo = D == 1 ? vpd1 : D == 2 ? vpd2 : D == 5 ? vpd5 : vpd6;
            lw_dma16(rPp, ppd_base + (unsigned)(2 * i * BLK * 8), vo, d.o36 + (unsigned)(ra * BLK * 8));
        }
        if (lane < HALF) lw_dma16(rPp, ppd_base + (unsigned)(20 * BLK * 8), vpd1, d.o36 + (unsigned)(lw_src_row(20) * BLK * 8));
    };

    if (k_from < T - 2) {      // resume from the hand-over rows
#pragma unroll
        for (int i = 0; i < M; i++) Ss[i] = a.hand_s[(size_t)i * hp + cc];
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) Ps[sidx(i, j)] = a.hand_p[(size_t)IXM(i, j) * hp + cc];
        const int word = a.hand_i[cc];
        st_guard = word & 1; st_cap = (word >> 1) & 1; min_rank = word >> 8;
    }
    const LwDay ds = lw_stride(a, bp);
    // terminal conditions GenericEKF.m:189-202 (Ps_final symmetric in values and NaN pattern: ekf_precheck)
    auto terminal = [&]() __attribute__((always_inline)) {
        const LwDay dT = lw_day(ds, tpos<FLIP>(T - 1, T) - tw);
#pragma unroll
        for (int i = 0; i < M; i++) Ss[i] = lw_ld<BLK>(rSp, ll.v6 & ~dead, dT.o6, i);
#pragma unroll
        for (int i = 0; i < M; i++) {
            const double f = a.s_final[(size_t)i * B + cc];
            if (!is_nan(f)) Ss[i] = f;
        }
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) Ps[sidx(i, j)] = lw_ld<BLK>(rPp, ll.v36, dT.o36, IXM(i, j));
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) {
                const double f = a.Ps_final[(size_t)IXM(i, j) * B + cc];
                if (!is_nan(f)) Ps[sidx(i, j)] = f;
            }
        st_vec(rSs, dT, Ss);
        st_sym(rPs, dT, Ps);
        double z[kNpi];
#pragma unroll
        for (int k = 0; k < kNpi; k++) z[k] = 0.0;
        st_u(rUs, dT, z);                                  // column T is never written :95,204
        st_word(rPr, dT, -1);
    };

    struct Small { double Sp[M], u[kNpi]; int rk; };
    Small cur, nxt;
    // (dt: the day of step k, tpos(k); dt1: the day after it in filter order, tpos(k + 1))
    auto fetch_small = [&](const LwDay &dt, const LwDay &dt1, int t_abs, Small &d) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < M; i++) d.Sp[i] = lw_ld<BLK>(rSp, ll.v6, dt.o6, i);
        load_u(a, t_abs, su, d.u);
        d.rk = (int)__builtin_amdgcn_raw_buffer_load_b32(rRank, ll.v1w, dt1.o1w, 0);
    };
    LwDay d_pend = lw_day(ds, 0);
    bool have_pend = false;
    int rank_pend = -1;
    auto flush = [&]() __attribute__((always_inline)) {          // store the previous step's results (Ss, Ps still hold them)
        st_word(rPr, d_pend, rank_pend);
        st_vec(rSs, d_pend, Ss);
        st_sym(rPs, d_pend, Ps);
        double u_pend[kNpi];
#pragma unroll
        for (int q = 0; q < kNpi; q++) u_pend[q] = s_upend[q * BLK + lw_opaque(lane)];
        st_u(rUs, d_pend, u_pend);
    };

    auto step = [&](int k, const LwDay &d0, const LwDay &d1) __attribute__((always_inline)) {
        // everything requested during the previous step (or by the prologue) has landed
        __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0) (expcnt, lgkmcnt untouched)
#pragma unroll
        for (int i = 0; i < M; i++) cur.Sp[i] = nxt.Sp[i];
#pragma unroll
        for (int q = 0; q < kNpi; q++) cur.u[q] = nxt.u[q];
        cur.rk = nxt.rk;
        if (EPI_LANE6D_FLUSH_TOP) {
            if (have_pend) flush();
#pragma unroll
            for (int q = 0; q < kNpi; q++) s_upend[q * BLK + lane] = cur.u[q];
        }

        // A = StateJacobians(u, s+) :206; the slope term and the bang-bang substitution only where some lane has a free control
        const bool wave_free = EPI_PROBE_NOFREE ? false : lw_wave_has_free_control(cur.u);
        double A[M * M], Sm1[M];
        jacobian_entries<M, FLIP>(p, cur.Sp, wave_free ? slope_term<M, FLIP>(p, cur.u, cur.Sp) : 0.0, A);
        {                      // s(k+1|k) = StateHardMargins(NlinStateUpdate(u, s+)) :155,164
            double u_app[kNpi];
#pragma unroll
            for (int q = 0; q < kNpi; q++) u_app[q] = cur.u[q];
            if (wave_free) resolve_control<M>(p, a.mf, cur.Sp, u_app);
            double dot = (p.gamma * p.A(0)) * (p.Umax(0) - u_app[0]);
#pragma unroll
            for (int q = 1; q < kNpi; q++) dot = fma(p.gamma * p.A(q), p.Umax(q) - u_app[q], dot);
            state_map<M, FLIP>(p, dot, cur.Sp, Sm1);
            state_hard_margins<M>(p, Sm1);
        }
        lw_phase();
        // PA = P+ A' :215 (zeros of A skipped): row i of it is also column i of A P+
        double PA[M * M], Dsym[NS];
        {
            double Pp[NS];
#pragma unroll
            for (int e = 0; e < NS; e++) Pp[e] = s_ppd[e * BLK + lane];
#pragma unroll
            for (int i = 0; i < M; i++)
#pragma unroll
                for (int j = 0; j < M; j++) {
                    double acc = 0.0;
                    bool first = true;
#pragma unroll
                    for (int q = 0; q < M; q++)
                        if (a_nz<M>(j, q)) {
                            acc = first ? Pp[sidx(i, q)] * A[IXM(j, q)] : fma(Pp[sidx(i, q)], A[IXM(j, q)], acc);
                            first = false;
                        }
                    PA[6 * i + j] = acc;
                }
#pragma unroll
            for (int e = 0; e < NS; e++) s_pp[e * BLK + lane] = Pp[e];
        }
        lw_phase();
        lw_predict_cov_from_pa(A, PA, Qd, Dsym);
        lw_phase();
        if (!EPI_LANE6D_FLUSH_TOP) {
            if (have_pend) flush();
#pragma unroll
            for (int q = 0; q < kNpi; q++) s_upend[q * BLK + lane] = cur.u[q];
        }
#pragma unroll
        for (int e = 0; e < NS; e++) Dsym[e] = Dsym[e] - Ps[e];        // D = P_MINUS(k+1) - P_SMOOTH(k+1)  :223
        // requests for step k - 1 (younger than the stores above, so the next step's one wait covers both)
        if (k > k_to) {
            const LwDay dm1 = lw_next<FLIP ? 1 : -1>(d0, ds);
            fetch_small(dm1, d0, tpos<FLIP>(k - 1, T), nxt);
            __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): the reads of the P+ image above are done before it is overwritten
            dma_pp(dm1);
        }
        lw_phase();
        double J[M * M];
        int rank = -1;
        {
            double X[NS];
#pragma unroll
            for (int e = 0; e < NS; e++) X[e] = s_xd[e * BLK + lane];      // (garbage where the :211 guard fired, rk < 0: unused)
            if (cur.rk < 0) {                                  // non-finite P_MINUS guard :211-213
#pragma unroll
                for (int e = 0; e < M * M; e++) J[e] = 0.0;
                st_guard = 1;
            } else {
#pragma unroll
                for (int i = 0; i < M; i++)
#pragma unroll
                    for (int j = 0; j < M; j++) {
                        double acc = PA[6 * i] * X[sidx(0, j)];
#pragma unroll
                        for (int q = 1; q < M; q++) acc = fma(PA[6 * i + q], X[sidx(q, j)], acc);
                        J[IXM(i, j)] = acc;
                    }
                rank = cur.rk & 0xff;
                st_cap |= (cur.rk >> 8) & 1;
                min_rank = rank < min_rank ? rank : min_rank;
            }
        }
        lw_phase();
        if (k > k_to) {
            __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): X has been read
            dma_x(d0);                                 // X of step k - 1 lies at the position of step k
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
#pragma unroll
        for (int i = 0; i < M; i++) Ss[i] = Sn[i];
        {
            // P_SMOOTH(k) = sym(P+ - (J D) J')   :223-226, rows of J D consumed one at a time (see eks_bwd_sym)
            double F[M * M];
            const double *lp = s_pp + lw_opaque(lane);
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
                    F[IXM(i, j)] = lp[sidx(i, j) * BLK] - acc;
                }
#pragma unroll
                for (int j = 0; j < i; j++) Ps[sidx(j, i)] = (F[IXM(i, j)] + F[IXM(j, i)]) / 2.0;
                Ps[sidx(i, i)] = (F[IXM(i, i)] + F[IXM(i, i)]) / 2.0;
            }
        }
        if (wave_free) {       // :229 -- only the control NlinStateUpdate returns is kept: the free ones, resolved with the smoothed costate
            double u_pend[kNpi];
            const double *lu = s_upend + lw_opaque(lane);
#pragma unroll
            for (int q = 0; q < kNpi; q++) u_pend[q] = lu[q * BLK];
            resolve_control<M>(p, a.mf, Ss, u_pend);
#pragma unroll
            for (int q = 0; q < kNpi; q++) s_upend[q * BLK + lane] = u_pend[q];
        }
        d_pend = d0;
        have_pend = true;
        rank_pend = rank;
    };

    constexpr int DIR = FLIP ? 1 : -1;
    const int W = lw_window(a, bp);
    int k = k_from;
    bool first = true;
    if (k_from >= T - 2 && k_from < k_to) {      // T == 1: no step, the terminal condition alone
        rebase(tpos<FLIP>(T - 1, T));
        terminal();
    }
    while (k >= k_to) {
        const int hi = k, lo = (hi - k_to >= W) ? hi - W + 1 : k_to;
        if (!first && have_pend) { flush(); have_pend = false; }      // the last step's results belong to the window that ends
        rebase(FLIP ? tpos<FLIP>(hi + 1, T) : (lo > 0 ? lo - 1 : 0));
        LwDay d1 = lw_day(ds, tpos<FLIP>(hi + 1, T) - tw), d0 = lw_day(ds, tpos<FLIP>(hi, T) - tw);
        if (first) {
            if (k_from >= T - 2) terminal();
            fetch_small(d0, d1, tpos<FLIP>(hi, T), nxt);
            dma_pp(d0);
            dma_x(d1);
            first = false;
        }
        for (; k >= lo; k--) {
            step(k, d0, d1);
            d1 = d0;
            d0 = lw_next<DIR>(d0, ds);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);      // a DMA must not be in flight when the wave ends (none is: the last step requests nothing)
    if (have_pend) flush();
    if (!live) return;
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
#endif
