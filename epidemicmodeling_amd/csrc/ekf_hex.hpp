// ekf_hex.hpp -- SIX lanes per chain, TEN chains per wavefront (6-state generic models; epi_batch_desc.shape = 4).
// Included by epiekf.hip inside namespace epi, after ekf_wave.hpp.
//
// Why a fourth shape.  The shard of the headline sweep on one of 8 GPUs is 9 375 chains: with four lanes per chain
// (ekf_quad.hpp) that is 586 wavefronts on 1 024 SIMDs -- 438 SIMDs idle while every wave works through ~850 / ~730 vector
// instructions per day (a quarter of them DPP moves of the block exchange, a sixth v_cndmask of lanes that play two roles).
// The largest number of lanes a chain can have while every wavefront still owns a SIMD is 1 024 * 64 / 9 375 = 6.9: six.
//
//   * lane 6 g + j (g = 0..9) owns COLUMN j of every 6 x 6 matrix of chain g of the wavefront; lanes 60..63 idle (they mirror
//     the last chain into a tile of their own and never store);
//   * covariances are stored symmetrised (GenericEKF.m:138,161,226: both halves hold the same bits), so column j IS row j.
//     A product X = L P with the left factor known to every lane (the Jacobian, I - K C: functions of the state, which all
//     six lanes hold) is formed column by column with no exchange at all, and the second product Y = X L' ROW by row:
//     Y(j, i) = sum_k X(j, k) L(i, k) needs row j of X -- one transpose through LDS (three ds_write_b128, six ds_read_b64
//     per lane) -- and again only the replicated factor.  The symmetrisation (Y + Y') / 2 is a second transpose.  No lane
//     ever needs a whole matrix of another lane's making, except J in the smoother (one all-gather per step);
//   * since every lane multiplies with the WHOLE Jacobian, its 15 structural zeros are skipped without any per-lane select
//     (fma(0, x, acc) == acc for finite x: the packed kernels' rule) -- 21 fma per product and lane instead of
//     the quad shape's 54 + 36 DPP moves;
//   * every element's fma chain still runs k-ascending in ONE lane: the oracle's rounding sequence, bit for bit;
//   * the twelve NPIs are handled two per lane; vectors that all lanes need (P C', the gain, u_max - u, the slope terms,
//     S_SMOOTH) are gathered through eight-double LDS rows.
// One wavefront per workgroup; LDS exchanges need no barrier (a wave's LDS operations execute in order), only a
// wavefront-scope fence that keeps the compiler from moving a read above the write it depends on.
//
// Inputs: R_v a per-day series (the innovation monitor is replayed by ekf_monitor), fixed diagonal Q_w, fp64 storage --
// the conditions of the wave shape.  Measured: profiles/r05/batch_size_sweep.txt.
#pragma once

constexpr int kHL = 6;          // lanes per chain
constexpr int kHG = 10;         // chains per wavefront
constexpr int kHGp = 11;        // groups incl. the phantom one of lanes 60..63
constexpr int kHT = 38;         // doubles per chain in a matrix tile: 36 + 2, i.e. 304 B -- 16-byte aligned, and the b128 reads
                                // of different chains fall into different banks ((76 g + 4 m) mod 64 = 12 g + 4 m: distinct)
constexpr int kHV = 8;          // doubles per chain in a 6-vector row
constexpr int kHN = 12;         // doubles per chain in an NPI row

typedef double hx_d2 __attribute__((ext_vector_type(2)));

// Addressing windows.  Day offsets travel in the buffer instructions' 32-bit SCALAR offset: an array's descriptor then names a
// fixed base and does not change from day to day, where a descriptor per array and day costs the scalar unit ten instructions
// each (a 64-bit base sum, the mask, the moves into an aligned quad; the kernels also ran out of scalar registers over it and
// kept their pointers in VGPR lanes) -- and a lone wave pays four cycles for EVERY instruction it issues (9 375-chain shard:
// 2.72 -> 2.62 ms per pass with the empty-descriptor change that came with it).  The hardware's bounds check includes the scalar
// offset (out of range: offset >= num_records - soffset), so the record count cannot clip a day's slice any more: every live
// access is in range by construction, the count (just below 2 GiB, 0 for an output that was not selected) only has to drop
// the lanes that must not store, which carry bit 31 in their offset; rows beyond n_npi are predicated by hand.  A window is
// the run of days whose offsets fit: the kernels move every array's base at its start (HexWin; one window for 520 days of up
// to 14 000 chains).
constexpr unsigned kHexDead = 0x80000000u, kHexRecords = 0x7FFFFFF8u;
struct HexLane {
    int g, j, c;                // group (chain of the wavefront), column owned, chain
    bool live;                  // the chain exists: this lane stores
    unsigned dead;              // OR-ed into every store offset: 0, or bit 31 for a lane that must not store -- beyond the record
                                // count, so the descriptor's bounds check drops the store and no store needs an EXEC-mask
                                // branch around it
};
EPI_DEV HexLane hx_lane(const KArgs &a)
{
    HexLane h;
    const int lane = (int)threadIdx.x;
    h.g = lane / kHL;
    h.j = lane - h.g * kHL;                       // lanes 60..63: group 10, columns 0..3
    const int c = a.c0 + (int)blockIdx.x * kHG + h.g;
    h.live = h.g < kHG && c < a.c0 + a.cn;
    h.c = h.live ? c : a.c0 + a.cn - 1;           // idle groups mirror the last chain: what they compute is dropped
#ifdef EPI_HEX_NOSTORE           // timing probe: no lane stores anything (results are wrong)
    h.dead = kHexDead;
#else
    h.dead = h.live ? 0u : kHexDead;
#endif
    return h;
}
// Ordering of the LDS exchanges.  The hardware executes a wave's LDS operations in issue order, so all that is needed is that
// the compiler keeps a read behind the write it depends on -- and it must, without being told: every exchange has an LDS array of
// its own, and in each a lane's write (slot j, or element j of a row) and its reads (element j of every slot, or the whole row)
// overlap for some j, so they may alias and stay in program order; accesses to DIFFERENT exchanges' arrays are free to move,
// which is what lets a lone wave fill one exchange's latency with the next one's arithmetic.  EPI_HEX_FENCE=1 puts a
// wavefront-scope fence around every exchange instead (measured slower).
#ifndef EPI_HEX_FENCE
#define EPI_HEX_FENCE 0
#endif
#ifndef EPI_HEX_BRANCHLESS
#define EPI_HEX_BRANCHLESS 0
#endif
EPI_DEV void hx_fence()
{
#if EPI_HEX_FENCE
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#endif
}

// The stores of this shape are non-temporal but NOT written through (`sc1`, what the other shapes' stores add): a layout row of
// ten chains is 80 bytes, so a store instruction's 480 contiguous bytes begin and end inside cache lines that the neighbouring
// instruction completes -- written through, every such partial line costs a memory transaction of its own (forward kernel of the
// 9 375-chain shard 1.79 ms with `nt sc1`, 1.53 plain, 1.35 with `nt`, 1.32 with no stores at all).
EPI_DEV void hst(rsrc_t r, unsigned voff, unsigned soff, double v) { bst_nt(r, voff, soff, v); }

// my six values into slot j of my chain's tile (t = tile + kHT g + 6 j)
EPI_DEV void hx_put6(double *t, const double (&v)[6])
{
    hx_d2 *q = (hx_d2 *)t;
    q[0] = hx_d2{v[0], v[1]};
    q[1] = hx_d2{v[2], v[3]};
    q[2] = hx_d2{v[4], v[5]};
}
// element j of every slot (t = tile + kHT g + j): the transposed six
EPI_DEV void hx_get_tr(const double *t, double (&v)[6])
{
#pragma unroll
    for (int i = 0; i < 6; i++) v[i] = t[6 * i];
}
// transpose: lane j gives v (column j, or row j) and receives element j of the six lanes' vectors
EPI_DEV void hx_transpose(double *tile_g, int j, const double (&v)[6], double (&o)[6])
{
    hx_fence();
    hx_put6(tile_g + 6 * j, v);
    hx_fence();
    hx_get_tr(tile_g + j, o);
}
// all 36 values of my chain's tile, slot-major: m[6 s + e] = element e of slot s
EPI_DEV void hx_get_all(const double *tile_g, double (&m)[36])
{
    const hx_d2 *q = (const hx_d2 *)tile_g;
#pragma unroll
    for (int e = 0; e < 18; e++) {
        const hx_d2 v = q[e];
        m[2 * e] = v.x;
        m[2 * e + 1] = v.y;
    }
}
// v[j] of a replicated 6-vector (j is this lane's column): a select chain on lane predicates.  The operands are made opaque
// first: hipcc otherwise reads the chain as v[j], i.e. a dynamically indexed array, and moves the vector to scratch.
EPI_DEV double hx_opaque(double v)
{
    asm("" : "+v"(v));
    return v;
}
EPI_DEV double hx_pick(const double (&v)[6], int j)
{
    double w[6];
#pragma unroll
    for (int i = 0; i < 6; i++) w[i] = hx_opaque(v[i]);       // (inside the select's arm the asm would turn the select into a branch)
    double r = w[0];
#pragma unroll
    for (int i = 1; i < 6; i++) r = (j == i) ? w[i] : r;
    return r;
}

// ---- per-chain constants ------------------------------------------------------------------------------------------
struct HexNpi {
    double a[2], umin[2], umax[2], ew[2], term[2];   // NPIs k = j and j + 6 (see QNpi)
    double inv_sigma;
    double ga[kNpi];                                 // gamma * a(k), all twelve: the constant factors of NlinStateUpdate's fma chain
};
EPI_DEV void hx_load_prm(QPrm &p, HexNpi &n, const KArgs &a, int B, const HexLane &h)
{
    const int c = h.c;
    auto g = [&](int f) { return a.prm[(size_t)f * B + c]; };
    p.dt = g(EPI_PRM_DT); p.beta = g(EPI_PRM_BETA); p.gamma = g(EPI_PRM_GAMMA);
    p.sigma = g(EPI_PRM_SIGMA); p.b = g(EPI_PRM_B); p.epsilon = g(EPI_PRM_EPSILON);
    p.slo = a.mf.lo_is_zero ? 0.0 : g(EPI_PRM_S_MIN);
    p.ilo = a.mf.lo_is_zero ? 0.0 : g(EPI_PRM_I_MIN);
    p.alpha_min = g(EPI_PRM_ALPHA_MIN); p.alpha_max = g(EPI_PRM_ALPHA_MAX);
    n.inv_sigma = 1.0 / p.sigma;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int k = h.j + 6 * s;
        n.a[s] = g(EPI_PRM_A + k); n.umin[s] = g(EPI_PRM_U_MIN + k); n.umax[s] = g(EPI_PRM_U_MAX + k);
        n.ew[s] = p.epsilon * g(EPI_PRM_W_EFF + k);
        // same products, same order as slope_term() / nlin_state_update(): constants of the chain, formed once
        n.term[s] = p.gamma * p.dt * (p.sigma / 2.0) * n.a[s] * (n.umax[s] - n.umin[s]);
    }
#pragma unroll
    for (int k = 0; k < kNpi; k++) n.ga[k] = p.gamma * g(EPI_PRM_A + k);
}
// u(k, t) for my two NPIs (rows beyond n_npi read 0.0 through the descriptor's bounds check, see load_u)
EPI_DEV void hx_load_u_at(const double *u, const KArgs &a, unsigned byte_off, int su, int j, double (&u2)[2])
{
    const unsigned rowb = (unsigned)a.Su * 8u, voff = (unsigned)su * 8u + (unsigned)j * rowb;
    const rsrc_t r = mk_rsrc(u, kHexRecords);
    u2[0] = bld(r, (j < a.n_npi) ? voff : kHexDead, byte_off);
    u2[1] = bld(r, (j + 6 < a.n_npi) ? voff : kHexDead, byte_off + 6u * rowb);
}
EPI_DEV void hx_load_u(const double *u, const KArgs &a, int t, int su, int j, double (&u2)[2])
{
    hx_load_u_at(u, a, (unsigned)t * a.n_npi * a.Su * 8u, su, j, u2);
}
// Addressing of the chain-blocked arrays (see Lay).  BLK > 0: the layout's lane_block is that compile-time constant (10 = one
// block per wavefront, what the host chooses for this shape): the row pitch is 80 bytes, every row offset folds into the
// instruction's 12-bit immediate and no scalar register holds one (the kernels run out of them otherwise and re-load their
// arguments every day).  BLK = 0: any layout, row offsets in SGPRs.
// An output the caller did not select (dst == NULL) gets an EMPTY descriptor: every store through it is dropped by the bounds
// check, and no store sits behind a branch (the waits the compiler places for the loads of a step count the stores issued
// since, which it can only do in straight-line code).  Only the RECORD COUNT depends on dst: a selected base address as well
// makes the compiler branch around the address arithmetic of every array every day.
EPI_DEV rsrc_t hx_rsrc(const void *dst) { return mk_rsrc(dst, dst ? kHexRecords : 0u); }
// The day of an array -- counted from the window's first day -- is named either by its index t or (the smoother's loop, see
// HexDay) by its byte offset.
struct HexAt { unsigned off; };
template <int BLK>
EPI_DEV rsrc_t hx_slice(const double *dst, HexAt at, unsigned rows, const Lay &l, unsigned &voff, unsigned &rowb)
{
    const unsigned blk = BLK ? (unsigned)BLK : l.blk;
    rowb = blk * 8u;
    voff = (l.cb * rows * blk + l.cr) * 8u;
    return hx_rsrc(dst);
}
EPI_DEV unsigned hx_soff(HexAt at, unsigned, const Lay &) { return at.off; }
EPI_DEV unsigned hx_soff(int t, unsigned rows, const Lay &l) { return (unsigned)t * rows * l.bp * 8u; }
template <int BLK>
EPI_DEV rsrc_t hx_slice(const double *dst, int t, unsigned rows, const Lay &l, unsigned &voff, unsigned &rowb)
{
    return hx_slice<BLK>(dst, HexAt{0u}, rows, l, voff, rowb);
}
// Byte offsets of ONE day (relative to the window's base) in the arrays of each row count -- one-row arrays of doubles (words:
// half of it), 6, 21 (packed X), 36 and n_npi rows, and the classic [T][n_npi][Su] control series -- carried from day to day by
// additions: the smoother's loop, whose day indices the compiler does not strength-reduce (its prefetch index may be -1).
struct HexDay { unsigned o1, o6, o21, o36, on, ou; };
struct HexDayStride { unsigned s1, s6, s21, s36, sn, su; };
EPI_DEV HexDayStride hx_day_stride(const KArgs &a, const Lay &l)
{
    HexDayStride s;
    s.s1 = l.bp * 8u; s.s6 = l.bp * 48u; s.s21 = l.bp * 168u; s.s36 = l.bp * 288u;
    s.sn = l.bp * 8u * (unsigned)a.n_npi; s.su = (unsigned)a.n_npi * (unsigned)a.Su * 8u;
    return s;
}
// days per addressing window: even (the loops alternate two input sets and a window must end on the second), and three days of
// the widest array short of the record count (the prefetch reaches one day beyond the window on either side)
EPI_DEV int hx_window(const KArgs &a, const Lay &l)
{
    const unsigned s36 = l.bp * 288u, su = (unsigned)a.n_npi * (unsigned)a.Su * 8u;
    int w = (int)(kHexRecords / (s36 > su ? s36 : su)) - 3;
    if (a.hexw >= 2 && a.hexw < w) w = a.hexw;         // the test knob can only shorten the window
    return w < 2 ? 2 : (w & ~1);
}
template <class P> EPI_DEV P *hx_rebase(P *p, int tw, size_t elems_per_day) { return p ? p + (size_t)tw * elems_per_day : nullptr; }
EPI_DEV HexDay hx_day(const HexDayStride &s, int t)
{
    HexDay d;
    const unsigned tt = (unsigned)t;
    d.o1 = tt * s.s1; d.o6 = tt * s.s6; d.o21 = tt * s.s21; d.o36 = tt * s.s36; d.on = tt * s.sn; d.ou = tt * s.su;
    return d;
}
template <int DIR> EPI_DEV HexDay hx_day_next(const HexDay &d, const HexDayStride &s)      // day t + DIR
{
    HexDay n;
    if (DIR > 0) { n.o1 = d.o1 + s.s1; n.o6 = d.o6 + s.s6; n.o21 = d.o21 + s.s21; n.o36 = d.o36 + s.s36; n.on = d.on + s.sn; n.ou = d.ou + s.su; }
    else { n.o1 = d.o1 - s.s1; n.o6 = d.o6 - s.s6; n.o21 = d.o21 - s.s21; n.o36 = d.o36 - s.s36; n.on = d.on - s.sn; n.ou = d.ou - s.su; }
    return n;
}
template <int BLK> EPI_DEV void hx_st(rsrc_t r, unsigned vo, unsigned row, unsigned rowb, double v, unsigned so)
{
    if (BLK) hst(r, vo + row * ((unsigned)BLK * 8u), so, v); else hst(r, vo, so + row * rowb, v);
}
template <int BLK> EPI_DEV double hx_ld(rsrc_t r, unsigned vo, unsigned row, unsigned rowb, unsigned so)
{
    return BLK ? bld_s(r, vo + row * ((unsigned)BLK * 8u), so) : bld_s(r, vo, so + row * rowb);
}
template <int BLK, class TT>
EPI_DEV void hx_store_u(double *__restrict__ dst, const KArgs &a, TT t, const Lay &l, const HexLane &h, const double (&u2)[2])
{
    unsigned voff, rowb;
    const rsrc_t r = hx_slice<BLK>(dst, t, (unsigned)a.n_npi, l, voff, rowb);
    const unsigned so = hx_soff(t, (unsigned)a.n_npi, l);
    const unsigned vo = (voff + (unsigned)h.j * rowb) | h.dead;
    // rows beyond n_npi lie beyond the slice (its last block's rows at the latest): dropped by the bounds check -- but a row
    // k >= n_npi of an EARLIER block would land in the next block's rows, so those lanes are sent out of range as well
    hx_st<BLK>(r, (h.j < a.n_npi) ? vo : kHexDead, 0u, rowb, u2[0], so);
    hx_st<BLK>(r, (h.j + 6 < a.n_npi) ? vo : kHexDead, 6u, rowb, u2[1], so);
}
// bang-bang substitution of my NaN controls (OptControlled.m:49-58) and my slope-term contributions (:107-114)
EPI_DEV void hx_resolve(const QPrm &p, const HexNpi &n, const ModelFlags &mf, double s6, const double (&u2)[2], double (&ur)[2],
                        double (&tm)[2])
{
    const double gs6 = p.gamma * s6;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const double phi = n.ew[s] - gs6 * n.a[s];
        const bool lo = mf.phi_ge ? (phi >= 0.0) : (phi > 0.0);
        const bool fr = is_nan(u2[s]);
        ur[s] = fr ? (lo ? n.umin[s] : n.umax[s]) : u2[s];
        tm[s] = (fr && phi > -n.inv_sigma && phi < n.inv_sigma) ? n.term[s] : 0.0;
    }
}
// my two values of a 12-vector to the chain's NPI row, then all twelve in NPI order
EPI_DEV void hx_gather12(double *row_g, int j, const double (&v2)[2], double (&v)[kNpi])
{
    hx_fence();
    row_g[j] = v2[0];
    row_g[j + 6] = v2[1];
    hx_fence();
    const hx_d2 *q = (const hx_d2 *)row_g;
#pragma unroll
    for (int e = 0; e < 6; e++) {
        const hx_d2 x = q[e];
        v[2 * e] = x.x;
        v[2 * e + 1] = x.y;
    }
}
// my value of a 6-vector to the chain's vector row, then all six
EPI_DEV void hx_gather6(double *row_g, int j, double mine, double (&v)[6])
{
    hx_fence();
    row_g[j] = mine;
    hx_fence();
    const hx_d2 *q = (const hx_d2 *)row_g;
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const hx_d2 x = q[e];
        v[2 * e] = x.x;
        v[2 * e + 1] = x.y;
    }
}
// slope_term(): a36 -= term(k) (+= for the time-flipped models), k ascending; 0.0 where a control does not qualify
// (a36 never is -0.0, so adding or subtracting +0.0 leaves its bits)
template <int FLIP>
EPI_DEV double hx_slope(double *row_g, int j, const double (&u2)[2], const double (&tm2)[2])
{
    const bool any_free = is_nan(u2[0]) || is_nan(u2[1]);
    if (__builtin_amdgcn_ballot_w64(any_free) == 0ull) return 0.0;      // historic days: no lane of the wave has a free control
    double tm[kNpi];
    hx_gather12(row_g, j, tm2, tm);
    double a36 = 0.0;
#pragma unroll
    for (int k = 0; k < kNpi; k++) a36 = FLIP ? (a36 + tm[k]) : (a36 - tm[k]);
    return a36;
}
// X(:, j) = A * v for the Jacobian's non-zero pattern (a_nz, ekf_sym.hpp): k ascending, structural zeros skipped
EPI_DEV void hx_mul_A(const double (&A)[36], const double (&v)[6], double (&o)[6])
{
    constexpr int M = 6;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double acc = 0.0;
        bool first = true;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            if (!a_nz<6>(i, k)) continue;
            acc = first ? A[IXM(i, k)] * v[k] : fma(A[IXM(i, k)], v[k], acc);
            first = false;
        }
        o[i] = acc;
    }
}

// vector arrays (6 rows): lane j stores row j
template <int BLK, class TT>
EPI_DEV void hx_store_elem(double *__restrict__ dst, TT t, const Lay &l, const HexLane &h, double v)
{
    unsigned voff, rowb;
    const rsrc_t r = hx_slice<BLK>(dst, t, 6, l, voff, rowb);
    hst(r, (voff + (unsigned)h.j * rowb) | h.dead, hx_soff(t, 6, l), v);
}
// one-row arrays ([T][nblk * blk] doubles: innovations): the six lanes of a chain store the same word
EPI_DEV void hx_store_scalar(double *__restrict__ dst, int t, const Lay &l, const HexLane &h, double v)
{
    hst(hx_rsrc(dst), (l.c * 8u) | h.dead, (unsigned)t * l.bp * 8u, v);
}
EPI_DEV void hx_store_word(int32_t *__restrict__ dst, HexAt at, const Lay &l, const HexLane &h, int32_t v)
{
    __builtin_amdgcn_raw_buffer_store_b32(v, hx_rsrc(dst), (l.c * 4u) | h.dead, at.off, 0);
}
EPI_DEV void hx_store_word(int32_t *__restrict__ dst, int t, const Lay &l, const HexLane &h, int32_t v)
{
    hx_store_word(dst, HexAt{(unsigned)t * l.bp * 4u}, l, h, v);
}
EPI_DEV int32_t hx_load_word(const int32_t *__restrict__ src, HexAt at, const Lay &l)
{
    return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(mk_rsrc(src, kHexRecords), l.c * 4u, at.off, 0);
}
template <int BLK, class TT>
EPI_DEV void hx_load_vec(const double *__restrict__ src, TT t, const Lay &l, double (&v)[6])
{
    unsigned voff, rowb;
    const rsrc_t r = hx_slice<BLK>(src, t, 6, l, voff, rowb);
#pragma unroll
    for (int i = 0; i < 6; i++) v[i] = hx_ld<BLK>(r, voff, (unsigned)i, rowb, hx_soff(t, 6, l));
}
// my element of a 6-row vector array
template <int BLK, class TT>
EPI_DEV double hx_load_elem(const double *__restrict__ src, TT t, const Lay &l, int j)
{
    unsigned voff, rowb;
    const rsrc_t r = hx_slice<BLK>(src, t, 6, l, voff, rowb);
    return bld_s(r, voff + (unsigned)j * rowb, hx_soff(t, 6, l));
}
// my column of a SYMMETRIC 6 x 6 array stored with all 36 rows (row e = i + 6 j): element (i, j) of my column goes to the
// position of element (j, i) -- the same bits (the lane that owns column i stores the same value at (i, j)) -- so that one
// store instruction covers rows 6 i + (0..5) of the wavefront's ten chains: 480 contiguous bytes when the layout block is
// the wavefront's ten chains, instead of six 80-byte pieces 480 bytes apart (partial cache lines: measured 2.9 instead of
// 1.2 ms for the forward kernel of the 9 375-chain shard)
template <int BLK, class TT>
EPI_DEV void hx_store_col(double *__restrict__ dst, TT t, const Lay &l, const HexLane &h, const double (&v)[6])
{
    unsigned voff, rowb;
    const rsrc_t r = hx_slice<BLK>(dst, t, 36, l, voff, rowb);
    const unsigned vo = (voff + (unsigned)h.j * rowb) | h.dead;
#pragma unroll
    for (int i = 0; i < 6; i++) hx_st<BLK>(r, vo, (unsigned)(6 * i), rowb, v[i], hx_soff(t, 36, l));
}
// my column of a symmetric 6 x 6 array that hx_store_col wrote (all 36 rows): element (i, j) from the position of (j, i)
template <int BLK, class TT>
EPI_DEV void hx_load_col(const double *__restrict__ src, TT t, const Lay &l, int j, double (&v)[6])
{
    unsigned voff, rowb;
    const rsrc_t r = hx_slice<BLK>(src, t, 36, l, voff, rowb);
    const unsigned vo = voff + (unsigned)j * rowb;
#pragma unroll
    for (int i = 0; i < 6; i++) v[i] = hx_ld<BLK>(r, vo, (unsigned)(6 * i), rowb, hx_soff(t, 36, l));
}

// ---------------------------------------------------------------------------
// forward pass: GenericExtendedKalmanFilter.m:98-169 (the monitor :172-179 is replayed by ekf_monitor)
// ---------------------------------------------------------------------------
struct HexFwdIn { double x, r, u[2]; };

// SOLO = 1: the kernel claims more than half of the register file (a clobbered accumulation register), so that two of its waves
// never share a SIMD while other SIMDs idle -- with its 235 registers the dispatcher otherwise packs them two to a SIMD here and
// there, and the launch ends with the slowest pair (forward stage of the 9 375-chain shard 1.23-1.28 -> 1.02-1.03 ms; the first
// version of the kernel, bound by its stores, did not care).  SOLO = 0 for grids beyond one wave per SIMD.
template <int FLIP, int BLK, int SOLO = 0>
__global__ __launch_bounds__(kWave, SOLO ? 1 : 2) void ekf_fwd_hex(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int M = 6;
    __shared__ __attribute__((aligned(16))) double tA[kHGp * kHT], tB[kHGp * kHT], tC[kHGp * kHT], tD[kHGp * kHT];
    __shared__ __attribute__((aligned(16))) double vPC[kHGp * kHV], vK[kHGp * kHV], vD[kHGp * kHN], vTm[kHGp * kHN];
    if (SOLO) asm volatile("" ::: "a60");
    if (*dense_flag) return;
    const HexLane h = hx_lane(a);
    const int B = a.B, T = a.T, c = h.c, j = h.j;
    const int sx = a.x_series ? a.x_series[c] : c;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    double *tAg = tA + kHT * h.g, *tBg = tB + kHT * h.g, *tCg = tC + kHT * h.g, *tDg = tD + kHT * h.g;
    double *vPCg = vPC + kHV * h.g, *vKg = vK + kHV * h.g, *vDg = vD + kHN * h.g, *vTmg = vTm + kHN * h.g;

    QPrm p;
    HexNpi np;
    hx_load_prm(p, np, a, B, h);
    const double v_bar = a.prm[(size_t)EPI_PRM_V_BAR * B + c];
    const double gamma = a.prm[(size_t)EPI_PRM_GAMMA_EKF * B + c];

    double sk_minus[M], Pc[M], Qv[M];
#pragma unroll
    for (int i = 0; i < M; i++) sk_minus[i] = a.s_init[(size_t)i * B + c];
    // Ps_init is bit-wise symmetric (ekf_precheck); the packed kernels read its upper triangle, so do we
#pragma unroll
    for (int i = 0; i < M; i++) {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        Pc[i] = a.Ps_init[(size_t)IXM(lo, hi) * B + c];
        Qv[i] = (i == j) ? a.Q[(size_t)IXM(j, j) * B + c] : 0.0;       // Q_w diagonal (ekf_precheck): row j of it
    }
    const double dk[3] = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0};

    // time segments: see ekf_fwd_sym
    const int k_begin = a.k_begin, k_end = (a.k_end > 0 && a.k_end < T) ? a.k_end : T;
    // the arrays of the current addressing window (see kHexDead): bases at its first day `tw`, days counted from there
    struct { double *S_MINUS, *P_MINUS, *S_PLUS, *P_PLUS, *K_GAIN, *innovations, *u_opt; const double *u; } w;
    int tw = 0;
    auto rebase = [&](int t0) __attribute__((always_inline)) {
        tw = t0;
        const size_t bp = lay.bp;
        w.S_MINUS = hx_rebase(a.S_MINUS, t0, 6 * bp); w.S_PLUS = hx_rebase(a.S_PLUS, t0, 6 * bp); w.K_GAIN = hx_rebase(a.K_GAIN, t0, 6 * bp);
        w.P_MINUS = hx_rebase(a.P_MINUS, t0, 36 * bp); w.P_PLUS = hx_rebase(a.P_PLUS, t0, 36 * bp);
        w.innovations = hx_rebase(a.innovations, t0, bp); w.u_opt = hx_rebase(a.u_opt, t0, (size_t)a.n_npi * bp);
        w.u = hx_rebase(a.u, t0, (size_t)a.n_npi * a.Su);
    };
    const unsigned voff_x = (unsigned)sx * 8u;
    // the inputs of a day are requested one day ahead into one of two register sets used alternately (the loop body exists
    // twice): nothing is copied at the end of a day, so no day ends by waiting for its own stores to drain
    auto fetch = [&](int k, HexFwdIn &d) __attribute__((always_inline)) {
        const int tn = tpos<FLIP>(k, T);
        d.x = ldg(a.x + (size_t)tn * a.Sx, voff_x);
        d.r = ldg(a.R_series + (size_t)k * a.Sx, voff_x);                 // R_v is not time-flipped (Backward*.m:27)
        hx_load_u(w.u, a, tn - tw, su, j, d.u);
    };
    auto day = [&](int k, const HexFwdIn &cur) __attribute__((always_inline)) {
        const int t = tpos<FLIP>(k, T) - tw;                                 // day of the window
        const double Rk = cur.r, xk = cur.x;
        const double (&u_in)[2] = cur.u;

        hx_store_elem<BLK>(w.S_MINUS, t, lay, h, hx_pick(sk_minus, j));          // :100-101
        hx_store_col<BLK>(w.P_MINUS, t, lay, h, Pc);

        double C[M];
        obs_jacobian<M>(a.mf, sk_minus, C);                                 // :115, C(4:6) == 0
        const double xk_minus = predict_obs<M>(a.mf, sk_minus, v_bar);      // :116-119

        double innov, Kj, sk_plus[M], Ppc[M];
        const bool valid = !is_nan(xk);                                     // :122 (per chain)
        // EPI_HEX_BRANCHLESS: every chain goes through the update and one without an observation keeps its values by select
        // (:130-135) -- the day is then ONE basic block, and the scheduler can fill the update's LDS round trips with the
        // arithmetic of the state map and the Jacobian, which only need the gain
        const bool upd = EPI_HEX_BRANCHLESS ? true : valid;
        if (upd) {
            innov = valid ? xk - xk_minus : 0.0;
            // (P C')(j) = P(j, 0:2) C(0:2)' -- P(j, k) == P(k, j): my column.  (C P)(k) holds the same bits.
            double PCj = Pc[0] * C[0];
            PCj = fma(Pc[1], C[1], PCj);
            PCj = fma(Pc[2], C[2], PCj);
            double PC[M];
            hx_gather6(vPCg, j, PCj, PC);
            double CPCt = PC[0] * C[0];
            CPCt = fma(PC[1], C[1], CPCt);
            CPCt = fma(PC[2], C[2], CPCt);
            const double den = CPCt + gamma * Rk;                           // :124 (D = 1, Hessian terms 0)
            const double Kraw = PCj / den;
            Kj = valid ? Kraw : 0.0;
            double Kg[M], K[M];
            hx_gather6(vKg, j, Kraw, Kg);
#pragma unroll
            for (int i = 0; i < M; i++) K[i] = Kg[i];
#pragma unroll
            for (int i = 0; i < M; i++) sk_plus[i] = valid ? sk_minus[i] + K[i] * innov : sk_minus[i];   // :129
            // I - K C: its first three columns (C(4:6) = 0: the others are those of the identity)
            double IK[M][3];
#pragma unroll
            for (int i = 0; i < M; i++)
#pragma unroll
                for (int q = 0; q < 3; q++) IK[i][q] = ((i == q) ? 1.0 : 0.0) - K[i] * C[q];
            // Joseph form :127.  T1 = (I - K C) P: my column
            double T1c[M], T1r[M], Fr[M], Fc[M];
#pragma unroll
            for (int i = 0; i < M; i++) {
                double acc = IK[i][0] * Pc[0];
                acc = fma(IK[i][1], Pc[1], acc);
                acc = fma(IK[i][2], Pc[2], acc);
                T1c[i] = (i >= 3) ? acc + Pc[i] : acc;                      // + 1 * P(i, j) for i >= 3
            }
            hx_transpose(tAg, j, T1c, T1r);
            // F = (T1 (I - K C)' + K R K') / gamma: my ROW, F(j, i) = sum_q T1(j, q) IK(i, q)
            const double KjR = Kraw * Rk;
#pragma unroll
            for (int i = 0; i < M; i++) {
                double acc = T1r[0] * IK[i][0];
                acc = fma(T1r[1], IK[i][1], acc);
                acc = fma(T1r[2], IK[i][2], acc);
                acc = (i >= 3) ? acc + T1r[i] : acc;                        // + T1(j, i) * 1 for i >= 3
                Fr[i] = (acc + KjR * K[i]) / gamma;
            }
            hx_transpose(tBg, j, Fr, Fc);
#pragma unroll
            for (int i = 0; i < M; i++) Ppc[i] = valid ? (Fc[i] + Fr[i]) / 2.0 : Pc[i];     // :138
        } else {                                                            // :130-135
            innov = 0.0;
            Kj = 0.0;
#pragma unroll
            for (int i = 0; i < M; i++) { sk_plus[i] = sk_minus[i]; Ppc[i] = Pc[i]; }
        }
        hx_store_elem<BLK>(w.K_GAIN, t, lay, h, Kj);
        hx_store_scalar(w.innovations, t, lay, h, innov);
        state_hard_margins<M>(p, sk_plus);                                  // :141
        hx_store_elem<BLK>(w.S_PLUS, t, lay, h, hx_pick(sk_plus, j));            // :167-169
        hx_store_col<BLK>(w.P_PLUS, t, lay, h, Ppc);

        // s(k+1|k) = NlinStateUpdate(u, s+), A = StateJacobians(u, s+)  :155-157
        double u_app[2], tm[2], d2[2], d[kNpi];
        hx_resolve(p, np, a.mf, sk_plus[5], u_in, u_app, tm);
        hx_store_u<BLK>(w.u_opt, a, t, lay, h, u_app);
        d2[0] = np.umax[0] - u_app[0];
        d2[1] = np.umax[1] - u_app[1];
        hx_gather12(vDg, j, d2, d);
        double dot = np.ga[0] * d[0];
#pragma unroll
        for (int q = 1; q < kNpi; q++) dot = fma(np.ga[q], d[q], dot);
        double sk_next[M];
        state_map<M, FLIP>(p, dot, sk_plus, sk_next);
        {
            // P(k+1|k) = sym(A P A' + Q)  :158-161: (A P)(:, j) mine, then row j of (A P) A', then the transposed half
            double A[M * M], Tc[M], Tr[M], Gr[M], Gc[M];
            jacobian_entries<M, FLIP>(p, sk_plus, hx_slope<FLIP>(vTmg, j, u_in, tm), A);
            hx_mul_A(A, Ppc, Tc);
            hx_transpose(tCg, j, Tc, Tr);
            hx_mul_A(A, Tr, Gr);                                            // G(j, i) = sum_q (A P)(j, q) A(i, q)
#pragma unroll
            for (int i = 0; i < M; i++) Gr[i] = Gr[i] + Qv[i];
            hx_transpose(tDg, j, Gr, Gc);
#pragma unroll
            for (int i = 0; i < M; i++) Pc[i] = (Gc[i] + Gr[i]) / 2.0;      // :161
        }
        state_hard_margins<M>(p, sk_next);                                  // :164
#pragma unroll
        for (int i = 0; i < M; i++) sk_minus[i] = sk_next[i];
    };
    {
        HexFwdIn bufA, bufB;
        const int W = hx_window(a, lay);
        int k = k_begin;
        bool first = true;
        while (k < k_end) {                      // one addressing window per pass: filter steps [k, w1)
            const int w1 = (k_end - k > W) ? k + W : k_end;
            // its earliest day -- the day after its last step included: the inputs of that one are requested from here
            rebase(FLIP ? tpos<FLIP>(w1 < T ? w1 : T - 1, T) : k);
            if (first) {
                if (k_begin > 0) {
                    hx_load_vec<BLK>(w.S_MINUS, tpos<FLIP>(k_begin, T) - tw, lay, sk_minus);
                    hx_load_col<BLK>(w.P_MINUS, tpos<FLIP>(k_begin, T) - tw, lay, j, Pc);
                }
                fetch(k, bufA);
                // Everything loaded so far has landed before the loop is entered.  Otherwise the loop inherits the prologue's
                // pending loads: the compiler's wait-count pass merges "requested just now, nothing issued since" (from here)
                // with "requested a day ago, eighteen stores issued since" (from the back edge) into the stricter of the two,
                // and EVERY day then begins by waiting until all but its own four new requests have drained -- i.e. for the
                // previous day's stores.
                __builtin_amdgcn_s_waitcnt(0);
                first = false;
            }
            // the inputs of a day are requested one day ahead into one of two register sets used alternately (the loop body
            // exists twice); a window that is not the last has an even number of days and hands bufA to the next
            while (k < w1) {
                if (k + 1 < T) fetch(k + 1, bufB);
                day(k, bufA);
                if (++k >= w1) break;
                if (k + 1 < T) fetch(k + 1, bufA);
                day(k, bufB);
                ++k;
            }
        }
    }
    if (k_end < T) {       // hand-over to the next time segment
        rebase(tpos<FLIP>(k_end, T));
        hx_store_elem<BLK>(w.S_MINUS, 0, lay, h, hx_pick(sk_minus, j));
        hx_store_col<BLK>(w.P_MINUS, 0, lay, h, Pc);
    }
}

// ---------------------------------------------------------------------------
// backward recursion: GenericExtendedKalmanFilter.m:189-230 (X = pinv(P_MINUS) comes from eks_pinv, packed)
// ---------------------------------------------------------------------------
// EPI_HEX_SHARE_LOADS: what all six lanes of a chain need of the stored forward quantities (S+, S-, the 21 packed entries of X)
// is loaded ONCE -- lane j its element of the vectors and the packed entries j, j + 6, j + 12, j + 18 of X -- and handed round
// through LDS at the start of the step: 21 instead of 48 vector-memory instructions per step and lane (the six lanes'
// replicated loads hit the same cache lines, but every one of them passes through the CU's one address unit, which four such
// waves keep busy 57 % of the time)
#ifndef EPI_HEX_SHARE_LOADS
#define EPI_HEX_SHARE_LOADS 1
#endif
// EPI_HEX_BWD_RECOMPUTE: P(k+1|k) is NOT read back: it is formed again from the P(k|k), S+(k), u(k) the step loads anyway, by
// the forward kernel's own instruction sequence (hx_mul_A, the two transposes, + Q, the symmetrisation) -- the same bits -- and
// its first product A P(k|k) is the (P+ A') row the gain needs in any case.  The smoother of this shape moves 6.2 GB in 1.30 ms
// at the 9 375-chain shard, the box's copy rate, with its vector unit busy 29 % of the time: 288 of its 1 272 bytes per step
// for 27 fma and two LDS transposes.  (The one-lane kernels lost with this in rounds 1-2: they pay every instruction.)
#ifndef EPI_HEX_BWD_RECOMPUTE
#define EPI_HEX_BWD_RECOMPUTE 1
#endif

#if EPI_HEX_SHARE_LOADS
struct HexBwdIn { double Spj, Sm1j, u[2], Ppc[6], Pm1c[6], Xp[4]; int rk; };
#else
struct HexBwdIn { double Sp[6], Sm1[6], u[2], Ppc[6], Pm1c[6], X[21]; int rk; };
#endif

// PF = 1: a step's inputs are requested one step ahead into one of two register sets (256 + 22 registers: one wave per SIMD) --
// for launches of at most one wave per SIMD; PF = 0: requested at the start of the step (242 registers, no accumulation
// registers: TWO waves per SIMD, which hide each other's memory latency) -- for larger launches.
// PF = 2 (round 6, BLK = 10 only): one step ahead as with PF = 1, but by LDS-DMA (`buffer_load_dwordx4 ... lds`).  With ten chains
// per layout block a day of a wave's S_PLUS / S_MINUS / P_PLUS / X is ONE contiguous run of 480 / 480 / 2 880 / 1 680 bytes, which
// seven instructions of 64 x 16 bytes copy into an LDS image; the lanes then pick their entries out of the image with ds_read_b64.
// The twelve 8-byte loads per lane and step of PF = 1 touch the same bytes in 60 x 12 scattered pieces: it was the CU's address
// unit that those kept busy (round 5: "0.77 ms without its loads" of 1.13), not the memory.
constexpr int kHxImgSp = 0, kHxImgSm = 60, kHxImgP = 120, kHxImgX = 480, kHxImg = 692;      // doubles; 5 536 bytes per image
typedef __attribute__((address_space(3))) void *hx_lds_ptr_t;
EPI_DEV void hx_dma16(rsrc_t r, unsigned lds_addr, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (hx_lds_ptr_t)(uintptr_t)lds_addr, 16, voff, soff, 0, EPI_LD_STREAM_AUX);
}
template <int FLIP, int BLK, int PF = 1>
__global__ __launch_bounds__(kWave, PF ? 1 : 2) void eks_bwd_hex(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int M = 6;
    static_assert(PF != 2 || BLK == kHG, "the DMA images assume one layout block per wavefront");
    __shared__ __attribute__((aligned(16))) double imgA[PF == 2 ? kHxImg : 2], imgB[PF == 2 ? kHxImg : 2];
    __shared__ __attribute__((aligned(16))) double tJ[kHGp * kHT], tB[kHGp * kHT], tC[kHGp * kHT];
    __shared__ __attribute__((aligned(16))) double vS[kHGp * kHV], vTm[kHGp * kHN];

#if EPI_HEX_SHARE_LOADS
    __shared__ __attribute__((aligned(16))) double vSp[kHGp * kHV], vSm[kHGp * kHV], tXs[kHGp * 24];
#endif
#ifdef EPI_HEX_BWD_PRIO
    __builtin_amdgcn_s_setprio(3);             // probe: the smoother's waves ahead of the pinv waves beside them
#endif
    if (*dense_flag) return;
    const HexLane h = hx_lane(a);
    const int B = a.B, T = a.T, c = h.c, j = h.j;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    double *tJg = tJ + kHT * h.g, *tBg = tB + kHT * h.g, *tCg = tC + kHT * h.g;
    double *vSg = vS + kHV * h.g, *vTmg = vTm + kHN * h.g;
#if EPI_HEX_BWD_RECOMPUTE
    double *tDg = tBg, *tEg = tCg;               // the recomputation's two transposes come before the recursion's: same tiles
    double Qv[M];
#pragma unroll
    for (int i = 0; i < M; i++) Qv[i] = (i == j) ? a.Q[(size_t)IXM(j, j) * B + c] : 0.0;   // Q_w diagonal (ekf_precheck): row j of it
#endif
#if EPI_HEX_SHARE_LOADS
    double *vSpg = vSp + kHV * h.g, *vSmg = vSm + kHV * h.g, *tXg = tXs + 24 * h.g;
#endif
    QPrm p;
    HexNpi np;
    hx_load_prm(p, np, a, B, h);
    // the clamp of MY state element (StateHardMargins: s, i, alpha; the costates are free)
    const double my_lo = (j == 0) ? p.slo : (j == 1) ? p.ilo : p.alpha_min;
    const double my_hi = (j == 2) ? p.alpha_max : 1.0;

    // smoother steps of this launch: k = k_from down to k_to (see eks_bwd_sym); k_from = T - 2 starts from the terminal
    // condition, a later launch resumes from the hand-over rows
    const int k_from = a.bk_from, k_to = a.bk_to;
    const size_t hp = (size_t)a.hand_pitch;
    int st_guard = 0, st_cap = 0, min_rank = M;
    double Ss[M], Psc[M];
    // the arrays of the current addressing window (see kHexDead): bases at its first day `tw`, days counted from there
    struct {
        const double *S_PLUS, *P_PLUS, *X, *S_MINUS, *P_MINUS, *u;
        const int32_t *rankbuf;
        double *S_SMOOTH, *P_SMOOTH, *u_opt_smooth;
        int32_t *pinv_rank;
    } w;
    int tw = 0;
    auto rebase = [&](int t0) __attribute__((always_inline)) {
        tw = t0;
        const size_t bp = lay.bp;
        w.S_PLUS = hx_rebase(a.S_PLUS, t0, 6 * bp); w.S_MINUS = hx_rebase(a.S_MINUS, t0, 6 * bp); w.S_SMOOTH = hx_rebase(a.S_SMOOTH, t0, 6 * bp);
        w.P_PLUS = hx_rebase(a.P_PLUS, t0, 36 * bp); w.P_MINUS = hx_rebase(a.P_MINUS, t0, 36 * bp); w.P_SMOOTH = hx_rebase(a.P_SMOOTH, t0, 36 * bp);
        w.X = hx_rebase(a.X, t0, 21 * bp);
        w.rankbuf = hx_rebase(a.rankbuf, t0, bp); w.pinv_rank = hx_rebase(a.pinv_rank, t0, bp);
        w.u_opt_smooth = hx_rebase(a.u_opt_smooth, t0, (size_t)a.n_npi * bp);
        w.u = hx_rebase(a.u, t0, (size_t)a.n_npi * a.Su);
    };
    if (k_from < T - 2) {
#pragma unroll
        for (int i = 0; i < M; i++) {
            Ss[i] = a.hand_s[(size_t)i * hp + c];
            Psc[i] = a.hand_p[(size_t)IXM(i, j) * hp + c];
        }
        const int word = a.hand_i[c];
        st_guard = word & 1; st_cap = (word >> 1) & 1; min_rank = word >> 8;
    }
    // terminal conditions GenericEKF.m:189-202 (Ps_final symmetric in values and NaN pattern: ekf_precheck); the last day lies
    // in the first window
    auto terminal = [&]() __attribute__((always_inline)) {
        const int tT = tpos<FLIP>(T - 1, T) - tw;
        hx_load_vec<BLK>(w.S_PLUS, tT, lay, Ss);
#pragma unroll
        for (int i = 0; i < M; i++) {
            const double f = a.s_final[(size_t)i * B + c];
            if (!is_nan(f)) Ss[i] = f;
        }
        hx_load_col<BLK>(w.P_PLUS, tT, lay, j, Psc);
#pragma unroll
        for (int i = 0; i < M; i++) {
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            const double f = a.Ps_final[(size_t)IXM(lo, hi) * B + c];
            if (!is_nan(f)) Psc[i] = f;
        }
        hx_store_elem<BLK>(w.S_SMOOTH, tT, lay, h, hx_pick(Ss, j));
        hx_store_col<BLK>(w.P_SMOOTH, tT, lay, h, Psc);
        if (a.u_opt_smooth) {
            const double z[2] = {0.0, 0.0};
            hx_store_u<BLK>(w.u_opt_smooth, a, tT, lay, h, z);                   // column T is never written :95,204
        }
        hx_store_word(w.pinv_rank, tT, lay, h, -1);
    };

    // everything step k reads is requested one iteration ahead: the stored forward quantities come from HBM
    // (dt: the day of step k, tpos(k); dt1: the day after it in filter order, tpos(k + 1) -- see HexDay)
    auto fetch = [&](const HexDay &dt, const HexDay &dt1, HexBwdIn &d) __attribute__((always_inline)) {
#if EPI_HEX_SHARE_LOADS
        d.Spj = hx_load_elem<BLK>(w.S_PLUS, HexAt{dt.o6}, lay, j);
        hx_load_u_at(w.u, a, dt.ou, su, j, d.u);
        d.rk = hx_load_word(w.rankbuf, HexAt{dt1.o1 >> 1}, lay);
        hx_load_col<BLK>(w.P_PLUS, HexAt{dt.o36}, lay, j, d.Ppc);
        {
            unsigned voff, rowb;
            const rsrc_t r = hx_slice<BLK>(w.X, HexAt{dt1.o21}, 21, lay, voff, rowb);  // (garbage where the :211 guard fired, rk < 0: unused)
            const unsigned vo = voff + (unsigned)j * rowb;
#pragma unroll
            for (int m = 0; m < 3; m++) d.Xp[m] = hx_ld<BLK>(r, vo, (unsigned)(6 * m), rowb, dt1.o21);
            d.Xp[3] = hx_ld<BLK>(r, voff + (unsigned)(j < 3 ? j : 2) * rowb, 18u, rowb, dt1.o21);     // packed entries 18..20 exist for j < 3
        }
        d.Sm1j = hx_load_elem<BLK>(w.S_MINUS, HexAt{dt1.o6}, lay, j);
#if !EPI_HEX_BWD_RECOMPUTE
        hx_load_col<BLK>(w.P_MINUS, HexAt{dt1.o36}, lay, j, d.Pm1c);
#endif
#else
        hx_load_vec<BLK>(w.S_PLUS, HexAt{dt.o6}, lay, d.Sp);
        hx_load_u_at(w.u, a, dt.ou, su, j, d.u);
        d.rk = hx_load_word(w.rankbuf, HexAt{dt1.o1 >> 1}, lay);
        hx_load_col<BLK>(w.P_PLUS, HexAt{dt.o36}, lay, j, d.Ppc);
        {
            unsigned voff, rowb;
            const rsrc_t r = hx_slice<BLK>(w.X, HexAt{dt1.o21}, 21, lay, voff, rowb);  // (garbage where the :211 guard fired, rk < 0: unused)
#pragma unroll
            for (int e = 0; e < 21; e++) d.X[e] = hx_ld<BLK>(r, voff, (unsigned)e, rowb, dt1.o21);
        }
        hx_load_vec<BLK>(w.S_MINUS, HexAt{dt1.o6}, lay, d.Sm1);
#if !EPI_HEX_BWD_RECOMPUTE
        hx_load_col<BLK>(w.P_MINUS, HexAt{dt1.o36}, lay, j, d.Pm1c);
#endif
#endif
    };
    // PF = 2: the same inputs by DMA into an image, the two small ones (controls, rank word) into registers
    struct HexSmall { double u[2]; int rk; };
    const unsigned lane16 = (unsigned)threadIdx.x * 16u;
    // (the WAVE's layout block, not the lane's: idle lanes mirror the batch's last chain, which lies in another block)
    const unsigned wblk = (unsigned)a.c0 / (unsigned)kHG + blockIdx.x;
    const unsigned vo6 = wblk * (6u * 80u) + lane16, vo36 = wblk * (36u * 80u) + lane16, vo21 = wblk * (21u * 80u) + lane16;
    auto fetch_dma = [&](const HexDay &dt, const HexDay &dt1, unsigned img, HexSmall &sm) __attribute__((always_inline)) {
        const rsrc_t rSp = hx_rsrc(w.S_PLUS), rSm = hx_rsrc(w.S_MINUS), rP = hx_rsrc(w.P_PLUS), rX = hx_rsrc(w.X);
        const unsigned lane = threadIdx.x;
        if (lane < 30u) hx_dma16(rSp, img + kHxImgSp * 8u, vo6, dt.o6);
        if (lane < 30u) hx_dma16(rSm, img + kHxImgSm * 8u, vo6, dt1.o6);
        hx_dma16(rP, img + kHxImgP * 8u, vo36, dt.o36);
        hx_dma16(rP, img + kHxImgP * 8u + 1024u, vo36, dt.o36 + 1024u);
        if (lane < 52u) hx_dma16(rP, img + kHxImgP * 8u + 2048u, vo36, dt.o36 + 2048u);
        hx_dma16(rX, img + kHxImgX * 8u, vo21, dt1.o21);
        if (lane < 41u) hx_dma16(rX, img + kHxImgX * 8u + 1024u, vo21, dt1.o21 + 1024u);
        hx_load_u_at(w.u, a, dt.ou, su, j, sm.u);
        sm.rk = hx_load_word(w.rankbuf, HexAt{dt1.o1 >> 1}, lay);
    };
    // my entries out of an image: element (row r, chain g) of a run lies at r * 10 + g
    auto unpack = [&](const double *img, const HexSmall &sm, HexBwdIn &d) __attribute__((always_inline)) {
#if EPI_HEX_SHARE_LOADS
        const int g = h.g;
        d.Spj = img[kHxImgSp + j * 10 + g];
        d.Sm1j = img[kHxImgSm + j * 10 + g];
#pragma unroll
        for (int i = 0; i < 6; i++) d.Ppc[i] = img[kHxImgP + (j + 6 * i) * 10 + g];
#pragma unroll
        for (int m = 0; m < 3; m++) d.Xp[m] = img[kHxImgX + (j + 6 * m) * 10 + g];
        d.Xp[3] = img[kHxImgX + (18 + (j < 3 ? j : 2)) * 10 + g];
        d.u[0] = sm.u[0]; d.u[1] = sm.u[1]; d.rk = sm.rk;
#endif
    };
    auto step = [&](const HexDay &dt, const HexBwdIn &cur) __attribute__((always_inline)) {
#if EPI_HEX_SHARE_LOADS
        double Sp[M], Sm1[M], X[24];
        hx_gather6(vSpg, j, cur.Spj, Sp);
        hx_gather6(vSmg, j, cur.Sm1j, Sm1);
        {
            tXg[j] = cur.Xp[0]; tXg[j + 6] = cur.Xp[1]; tXg[j + 12] = cur.Xp[2];
            tXg[18 + j] = cur.Xp[3];                                       // lanes j >= 3 leave their duplicate in the padding entries 21..23
            const hx_d2 *q = (const hx_d2 *)tXg;
#pragma unroll
            for (int e = 0; e < 11; e++) {
                const hx_d2 x = q[e];
                X[2 * e] = x.x;
                X[2 * e + 1] = x.y;
            }
        }
#else
        const double (&Sp)[M] = cur.Sp;
        const double (&Sm1)[M] = cur.Sm1;
        const double (&X)[21] = cur.X;
#endif

        double A[M * M];
        {
            double ur_unused[2], tm[2];
            hx_resolve(p, np, a.mf, Sp[5], cur.u, ur_unused, tm);
            jacobian_entries<M, FLIP>(p, Sp, hx_slope<FLIP>(vTmg, j, cur.u, tm), A);   // :206
        }
        // J = (P+ A') X  :215.  Row j of P+ A' is A P+(:, j) (P+ symmetric bit for bit); row j of J needs all of X
        double Jr[M];
        int rank = -1;
        const bool guard = cur.rk < 0;                                      // non-finite P_MINUS guard :211-213 (per chain)
        double Pm1c[M];
        {
            double PAr[M];
            hx_mul_A(A, cur.Ppc, PAr);                                      // (P+ A')(j, i) = sum_q P+(j, q) A(i, q)
#pragma unroll
            for (int i = 0; i < M; i++) {
                double acc = PAr[0] * X[sidx(0, i)];
#pragma unroll
                for (int q = 1; q < M; q++) acc = fma(PAr[q], X[sidx(q, i)], acc);
                Jr[i] = guard ? 0.0 : acc;
            }
#if EPI_HEX_BWD_RECOMPUTE
            // P(k+1|k) = sym(A P+ A' + Q) exactly as ekf_fwd_hex formed it: (A P+)(:, j) is PAr
            double Tr[M], Gr[M], Gc[M];
            hx_transpose(tDg, j, PAr, Tr);
            hx_mul_A(A, Tr, Gr);
#pragma unroll
            for (int i = 0; i < M; i++) Gr[i] = Gr[i] + Qv[i];
            hx_transpose(tEg, j, Gr, Gc);
#pragma unroll
            for (int i = 0; i < M; i++) Pm1c[i] = (Gc[i] + Gr[i]) / 2.0;
#else
#pragma unroll
            for (int i = 0; i < M; i++) Pm1c[i] = cur.Pm1c[i];
#endif
        }
        if (guard) st_guard = 1;
        else {
            rank = cur.rk & 0xff;
            st_cap |= (cur.rk >> 8) & 1;
            min_rank = rank < min_rank ? rank : min_rank;
        }
        // S_SMOOTH(k) = clamp(s+ + J (s_s(k+1) - s-(k+1)))  :218-221: my element, then all six for the next step
        double Sn[M];
        {
            double acc = Jr[0] * (Ss[0] - Sm1[0]);
#pragma unroll
            for (int q = 1; q < M; q++) acc = fma(Jr[q], Ss[q] - Sm1[q], acc);
#if EPI_HEX_SHARE_LOADS
            const double mine = cur.Spj + acc;
#else
            const double mine = hx_pick(Sp, j) + acc;
#endif
            const double clamped = fmin(my_hi, fmax(my_lo, mine));
            hx_gather6(vSg, j, (j < 3) ? clamped : mine, Sn);
        }
        // P_SMOOTH(k) = sym(P+ - (J D) J'),  D = P_MINUS(k+1) - P_SMOOTH(k+1)   :223-226
        {
            double Jall[36], Dc[M], JDc[M], JDr[M], Fr[M], Fc[M];
            hx_fence();
            hx_put6(tJg + 6 * j, Jr);                                       // slot j = row j of J
            hx_fence();
            hx_get_all(tJg, Jall);                                          // Jall[6 i + q] = J(i, q)
#pragma unroll
            for (int i = 0; i < M; i++) Dc[i] = Pm1c[i] - Psc[i];           // D(:, j)
#pragma unroll
            for (int i = 0; i < M; i++) {                                   // (J D)(:, j)
                double acc = Jall[6 * i] * Dc[0];
#pragma unroll
                for (int q = 1; q < M; q++) acc = fma(Jall[6 * i + q], Dc[q], acc);
                JDc[i] = acc;
            }
            hx_transpose(tBg, j, JDc, JDr);
#pragma unroll
            for (int i = 0; i < M; i++) {                                   // F(j, i) = P+(j, i) - sum_q (J D)(j, q) J(i, q)
                double acc = JDr[0] * Jall[6 * i];
#pragma unroll
                for (int q = 1; q < M; q++) acc = fma(JDr[q], Jall[6 * i + q], acc);
                Fr[i] = cur.Ppc[i] - acc;
            }
            hx_transpose(tCg, j, Fr, Fc);
#pragma unroll
            for (int i = 0; i < M; i++) Psc[i] = (Fc[i] + Fr[i]) / 2.0;     // :226
        }
#pragma unroll
        for (int i = 0; i < M; i++) Ss[i] = Sn[i];
        hx_store_word(w.pinv_rank, HexAt{dt.o1 >> 1}, lay, h, (int32_t)rank);
        hx_store_elem<BLK>(w.S_SMOOTH, HexAt{dt.o6}, lay, h, hx_pick(Ss, j));
        hx_store_col<BLK>(w.P_SMOOTH, HexAt{dt.o36}, lay, h, Psc);
        if (a.u_opt_smooth) {                                               // :229 -- only the control NlinStateUpdate returns is kept
            double ur[2], tm_unused[2];
            hx_resolve(p, np, a.mf, Ss[5], cur.u, ur, tm_unused);
            hx_store_u<BLK>(w.u_opt_smooth, a, HexAt{dt.on}, lay, h, ur);
        }
    };
    // The days of the steps, walked with k: step k - 1 lies one day EARLIER in filter order (later in the array when FLIP).
    // One addressing window per pass of the outer loop: steps k = hi ... lo; it touches the days of steps lo - 1 (the inputs of
    // the next window's first step are requested from here) ... hi + 1, and its base is the earliest of them.
    constexpr int DIR = FLIP ? 1 : -1;
    const HexDayStride ds = hx_day_stride(a, lay);
    const int W = hx_window(a, lay);
    HexBwdIn bufA, bufB;
    HexSmall smA, smB;
    int k = k_from;
    bool first = true;
    if (k_from >= T - 2 && k_from < k_to) {      // T == 1: no step, the terminal condition alone
        rebase(tpos<FLIP>(T - 1, T));
        terminal();
    }
    while (k >= k_to) {
        const int hi = k, lo = (hi - k_to >= W) ? hi - W + 1 : k_to;
        rebase(FLIP ? tpos<FLIP>(hi + 1, T) : (lo > 0 ? lo - 1 : 0));
        HexDay d1 = hx_day(ds, tpos<FLIP>(hi + 1, T) - tw), d0 = hx_day(ds, tpos<FLIP>(hi, T) - tw);      // days of steps k + 1, k
        if (first && k_from >= T - 2) terminal();
        if constexpr (!PF) {
            for (; k >= lo; k--) {
                fetch(d0, d1, bufA);
                step(d0, bufA);
                d1 = d0;
                d0 = hx_day_next<DIR>(d0, ds);
            }
        } else if constexpr (PF == 2) {
            // two images used alternately (the loop body exists twice, each naming its image: the compiler waits for the DMA into the
            // image a step reads and for no other); a window that is not the last has an even number of steps and hands image A on
            const unsigned aA = (unsigned)(uintptr_t)(hx_lds_ptr_t)imgA, aB = (unsigned)(uintptr_t)(hx_lds_ptr_t)imgB;
            if (first) {
                fetch_dma(d0, d1, aA, smA);
                __builtin_amdgcn_s_waitcnt(0);
            }
            // The image a step reads was requested a step ago; the wait is explicit -- all but the ten vector-memory operations of the
            // group just issued (seven DMA, the two control loads, the rank word) have completed, or all when nothing was issued --
            // because nothing tells the compiler that the DMA (its LDS address is an integer) wrote THIS array: without it a step
            // read a stale image once in a while, when the pinv grids beside the smoother made memory slow (found by the shard test).
            while (k >= lo) {
                const HexDay dm1 = hx_day_next<DIR>(d0, ds);
                if (k > k_to) { fetch_dma(dm1, d0, aB, smB); __builtin_amdgcn_s_waitcnt(0x0F7A); } else __builtin_amdgcn_s_waitcnt(0x0F70);
                unpack(imgA, smA, bufA);
                step(d0, bufA);
                if (--k < lo) break;
                const HexDay dm2 = hx_day_next<DIR>(dm1, ds);
                if (k > k_to) { fetch_dma(dm2, dm1, aA, smA); __builtin_amdgcn_s_waitcnt(0x0F7A); } else __builtin_amdgcn_s_waitcnt(0x0F70);
                unpack(imgB, smB, bufB);
                step(dm1, bufB);
                --k;
                d0 = dm2;
            }
        } else {
            // two input sets used alternately (the loop body exists twice): the prefetched values are consumed where they landed; a
            // window that is not the last has an even number of steps and hands bufA to the next
            if (first) {
                fetch(d0, d1, bufA);
                __builtin_amdgcn_s_waitcnt(0);       // see ekf_fwd_hex: the loop must not inherit the prologue's pending loads
            }
            while (k >= lo) {
                const HexDay dm1 = hx_day_next<DIR>(d0, ds);                    // day of step k - 1 (never used when there is none)
                if (k > k_to) fetch(dm1, d0, bufB);
                step(d0, bufA);
                if (--k < lo) break;
                const HexDay dm2 = hx_day_next<DIR>(dm1, ds);
                if (k > k_to) fetch(dm2, dm1, bufA);
                step(dm1, bufB);
                --k;
                d0 = dm2;
            }
        }
        first = false;
    }
    if (PF == 2) __builtin_amdgcn_s_waitcnt(0x0070);      // (no DMA is in flight when the wave ends: the last step requests none)
    if (!h.live) return;
    if (k_to > 0) {        // hand-over to the launch that continues with step k_to - 1
        a.hand_s[(size_t)j * hp + c] = hx_pick(Ss, j);
        if (j == 0) a.hand_i[c] = st_guard | (st_cap << 1) | (min_rank << 8);
#pragma unroll
        for (int i = 0; i < M; i++) a.hand_p[(size_t)IXM(i, j) * hp + c] = Psc[i];
    } else if (a.status && j == 0) a.status[c] = st_guard | (st_cap << 1) | (min_rank << 8);
}
