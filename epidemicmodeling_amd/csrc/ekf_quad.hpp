// ekf_quad.hpp -- cooperative forward / backward kernels: FOUR lanes per chain (6-state generic models).
// Included from epiekf.hip inside namespace epi, after ekf_sym.hpp.
//
// Why a second shape.  With one lane per chain (ekf_sym.hpp) a wavefront carries ~1500 (forward) / ~1300 (backward)
// dependent fp64 instructions per day and needs 300-410 VGPRs, i.e. one wave per SIMD: a batch that does not fill the
// chip (the 9 375-chain shard of the headline sweep on one of 8 GPUs = 147 waves for 1024 SIMDs) runs at the latency
// of a lone wave, ~5 us per day.  Here the four lanes of a DPP quad share one chain: every 6 x 6 matrix is cut into a
// 2 x 2 grid of 3 x 3 blocks and lane (bi, bj) = (quad lane >> 1, quad lane & 1) owns block (bi, bj), so a matrix
// product costs each lane 54 instead of 216 fma, the blocks it needs from its neighbours arrive by `v_mov_b32_dpp
// quad_perm` (two per double, no LDS), and a wavefront holds 16 chains in a quarter of the registers.  Four times as
// many waves per batch, a per-day instruction stream ~2x shorter: what a small batch needs.  At 75 000 chains the chip
// is full either way and the one-lane-per-chain kernels do less total work; the host picks the shape by batch size
// (epi_batch_desc.shape).
//
// Arithmetic: every output element is produced by the same sequence of roundings as in ekf_sym.hpp / the oracle -- a
// reduction is per element, and an element's whole fma chain (k ascending) runs in ONE lane; where a chain spans two
// blocks (the smoother's J * (s - s^-)) the partial sum is handed from the left lane to the right lane and continued.
// Products with structural zeros of the Jacobian are NOT skipped here (the union of the two block rows' patterns is
// nearly dense): fma(0, x, acc) == acc for finite operands, the same caveat as in ekf_sym.hpp.
#pragma once

constexpr int kQC = kWave / 4;   // chains per wavefront
// One wavefront per workgroup.  (Workgroups of four wavefronts -- one per SIMD of a CU -- were measured: every wave ran
// 30-45 % slower, 1.65 instead of 1.13 ms for the forward kernel at 300 chains; waves that share a CU get in each other's
// way, so the fewer per CU the better, and single-wave workgroups spread over all CUs first.)
#ifndef EPI_QUAD_BWD_PF
#define EPI_QUAD_BWD_PF 0         // smoother: 1 = request a step's inputs one iteration ahead (two register sets); measured
                                  // level with 0 (1.49 vs 1.46 ms, 9 375 chains): the quad kernels are issue-bound, not latency-bound
#endif
#ifndef EPI_QUAD_WAVES
#define EPI_QUAD_WAVES 1          // minimum waves per SIMD the quad kernels are compiled for (register cap 512 / n)
#endif

#define EPI_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
constexpr int QP_ROW_L = EPI_QP(0, 0, 2, 2);   // block (bi, 0) of my block row
constexpr int QP_ROW_R = EPI_QP(1, 1, 3, 3);   // block (bi, 1)
constexpr int QP_COL_T = EPI_QP(0, 1, 0, 1);   // block (0, bj) of my block column
constexpr int QP_COL_B = EPI_QP(2, 3, 2, 3);   // block (1, bj)
constexpr int QP_TRN = EPI_QP(0, 2, 1, 3);     // lane (bj, bi): the transposed position
constexpr int QP_CJ_L = EPI_QP(0, 2, 0, 2);    // block (bj, 0): block row bj, left
constexpr int QP_CJ_R = EPI_QP(1, 3, 1, 3);    // block (bj, 1)

template <int CTRL>
EPI_DEV double qx(double v)
{
    const u32x2 w = __builtin_bit_cast(u32x2, v);
    u32x2 r;
    r.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w.x, CTRL, 0xf, 0xf, true);
    r.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w.y, CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, r);
}
typedef double blk3[3][3];
template <int CTRL>
EPI_DEV void qx3(const blk3 &s, blk3 &d)
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) d[r][c] = qx<CTRL>(s[r][c]);
}
// d = (block held by the transposed lane)' : d[r][c] is element (j, i) of the matrix when mine is (i, j)
EPI_DEV void qx3_transposed(const blk3 &s, blk3 &d)
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) d[r][c] = qx<QP_TRN>(s[c][r]);
}
EPI_DEV void qsym(const blk3 &F, blk3 &P)      // (F + F')/2.0   GenericEKF.m:138,161,226
{
    blk3 Ft;
    qx3_transposed(F, Ft);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) P[r][c] = (F[r][c] + Ft[r][c]) / 2.0;
}

struct Quad {
    int q, lc;          // lane in the quad, chain in the wavefront
    bool bi, bj;        // block row / block column owned
};

// --- addressing of the chain-blocked arrays (see Lay): element (row e, chain c) of slice t -------------------------
// BLK > 0: the layout's lane_block is that compile-time constant (16 = one block per quad wavefront, what the host
// chooses for this shape): the row pitch is BLK * 8 bytes, every row offset folds into the instruction's 12-bit immediate
// and no SGPR holds one.  BLK = 0: any layout, row offsets in SGPRs (as in ekf_sym.hpp).
template <int BLK>
EPI_DEV rsrc_t qslice(const double *p, int t, unsigned rows, const Lay &l, unsigned &voff, unsigned &rowb)
{
    if (BLK) {
        rowb = (unsigned)BLK * 8u;
        voff = (l.cb * rows * (unsigned)BLK + l.cr) * 8u;
        return mk_rsrc(p + (size_t)t * rows * l.bp, rows * l.bp * 8u);
    }
    return lay_slice(p, t, rows, l, voff, rowb);
}
template <int BLK> EPI_DEV double qld(rsrc_t r, unsigned vo, unsigned row, unsigned rowb)
{
    return BLK ? bld(r, vo + row * ((unsigned)BLK * 8u), 0u) : bld(r, vo, row * rowb);
}
template <int BLK> EPI_DEV void qst(rsrc_t r, unsigned vo, unsigned row, unsigned rowb, double v)
{
    if (BLK) bst(r, vo + row * ((unsigned)BLK * 8u), 0u, v); else bst(r, vo, row * rowb, v);
}
// vector arrays (6 rows): quad lane q stores rows q and 4 + (q & 1) -- lanes 2, 3 repeat what lanes 0, 1 store in the
// second instruction (same value, same address), so no lane is ever masked off
template <int BLK>
EPI_DEV void qstore_vec(double *__restrict__ dst, int t, const Lay &l, const Quad &Q, const double (&v)[6])
{
    if (!dst) return;
    unsigned voff, rowb;
    const rsrc_t r = qslice<BLK>(dst, t, 6, l, voff, rowb);
    const double lo = (Q.q == 0) ? v[0] : (Q.q == 1) ? v[1] : (Q.q == 2) ? v[2] : v[3];
    const double hi = (Q.q & 1) ? v[5] : v[4];
    qst<BLK>(r, voff + (unsigned)Q.q * rowb, 0u, rowb, lo);
    qst<BLK>(r, voff + (unsigned)(Q.q & 1) * rowb, 4u, rowb, hi);
}
template <int BLK>
EPI_DEV void qload_vec(const double *__restrict__ src, int t, const Lay &l, double (&v)[6])
{
    unsigned voff, rowb;
    const rsrc_t r = qslice<BLK>(src, t, 6, l, voff, rowb);
#pragma unroll
    for (int i = 0; i < 6; i++) v[i] = qld<BLK>(r, voff, (unsigned)i, rowb);
}
// one-row arrays ([T][nblk*blk]: innovations, rho, rank words): the four lanes of a quad store the same word
template <class TV>
EPI_DEV void qstore_scalar(TV *__restrict__ dst, int t, const Lay &l, TV v)
{
    if (dst) dst[lay_scalar(t, l)] = v;
}
// my 3 x 3 block of a 6 x 6 array stored with all 36 rows (row e = i + 6 j).  Always the full matrix, also where the
// one-lane kernels store the upper triangle only (workspace): this shape runs batches that do not fill the chip, where
// an unmasked store costs less than the exec-mask bookkeeping around a masked one
template <int BLK>
EPI_DEV void qstore_blk(double *__restrict__ dst, int t, const Lay &l, const Quad &Q, const blk3 &Bk)
{
    if (!dst) return;
    unsigned voff, rowb;
    const rsrc_t r = qslice<BLK>(dst, t, 36, l, voff, rowb);
    const unsigned vo = voff + (unsigned)((Q.bi ? 3 : 0) + (Q.bj ? 18 : 0)) * rowb;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) qst<BLK>(r, vo, (unsigned)(rr + 6 * c), rowb, Bk[rr][c]);
}
// my block of a SYMMETRIC 6 x 6 array of which only the upper triangle may be stored: element (i, j) is read from
// row min + 6 max.  rows36 = false: the packed 21-row form (row i + j (j + 1) / 2, i <= j) of eks_pinv's X.
// The nine per-lane byte offsets (chain part included) are formed once per kernel.
struct QOff { unsigned o[3][3]; };
template <int BLK>
EPI_DEV QOff qoffsets(const Quad &Q, const Lay &l, bool rows36)
{
    QOff f;
    const unsigned rows = rows36 ? 36u : 21u, blk = BLK ? (unsigned)BLK : l.blk;
    const unsigned base = (l.cb * rows * blk + l.cr) * 8u, rowb = blk * 8u;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int i = (Q.bi ? 3 : 0) + r, j = (Q.bj ? 3 : 0) + c;
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            f.o[r][c] = base + (unsigned)(rows36 ? lo + 6 * hi : lo + hi * (hi + 1) / 2) * rowb;
        }
    return f;
}
EPI_DEV void qload_sym_blk(const double *__restrict__ src, int t, const Lay &l, const QOff &f, unsigned rows, blk3 &Bk)
{
    const rsrc_t r = mk_rsrc(src + (size_t)t * rows * l.bp, rows * l.bp * 8u);
#pragma unroll
    for (int rr = 0; rr < 3; rr++)
#pragma unroll
        for (int c = 0; c < 3; c++) Bk[rr][c] = bld(r, f.o[rr][c], 0u);
}

// --- the innovation monitor's windows (GenericEKF.m:172-179) --------------------------------------------------------
// A window of L samples is a ring in LDS written TWICE (at pos and pos + L of a 2L-long column): the L samples newest ->
// oldest are then the contiguous run pos .. pos + L - 1, read with immediate offsets and no wrap arithmetic.
// LC > 0: L is that compile-time constant (21 is what every caller of the reference passes) and the sum is fully unrolled.
template <int LC, int STRIDE = kQC>
EPI_DEV double qring_sum(const double *newest_ptr, int L, double newest)
{
    double sum = newest;
    if (LC) {
        double v[LC > 1 ? LC - 1 : 1];
#pragma unroll
        for (int j = 1; j < LC; j++) v[j - 1] = newest_ptr[j * STRIDE];
#pragma unroll
        for (int j = 1; j < LC; j++) sum = sum + v[j - 1];
        return sum;
    }
    int j = 1;
    for (; j + 10 <= L; j += 10) {
        double v[10];
#pragma unroll
        for (int q = 0; q < 10; q++) v[q] = newest_ptr[(j + q) * STRIDE];
#pragma unroll
        for (int q = 0; q < 10; q++) sum = sum + v[q];
    }
    for (; j < L; j++) sum = sum + newest_ptr[j * STRIDE];
    return sum;
}

// ---------------------------------------------------------------------------
// innovation monitor as a kernel of its own
// ---------------------------------------------------------------------------
// GenericEKF.m:172-179.  rho(k) depends on the innovations and on R(k) only.  When R_v is a per-day series (r_mode 1: what
// TrainPredictPrescribeNPI.m:240 passes) nothing of the monitor feeds back into the filter -- the adaptive R of :180-185
// needs a scalar R_v -- so the three L-sample window sums (~150 dependent instructions and 2-3 LDS windows per chain and
// day) leave the sequential forward kernels: they store the innovations, and this kernel, one lane per chain, replays the
// monitor over them with the very same arithmetic (same window order, same sums).  Used by both lane mappings.
template <int FLIP, int LC>
__global__ __launch_bounds__(kWave) void ekf_monitor(const KArgs a, const int *__restrict__ dense_flag)
{
    extern __shared__ double lds[];   // two windows [2][2 L][64], double-written (see qring_sum)
    if (*dense_flag) return;          // the dense kernels keep the monitor inline
    const int lane = threadIdx.x;
    const int c = a.c0 + blockIdx.x * kWave + lane;
    if (c >= a.c0 + a.cn) return;
    const int T = a.T, L = LC ? LC : a.L;
    const int sx = a.x_series ? a.x_series[c] : c;
    const Lay lay = make_lay(a, c);
    double *winMean = lds + lane, *winCovN = lds + (size_t)2 * L * kWave + lane;
    for (int j = 0; j < 2 * L; j++) { winMean[j * kWave] = 0.0; winCovN[j * kWave] = 0.0; }
    int pos = 0;
    const unsigned voff_x = (unsigned)sx * 8u;
    // rho(k) is a function of the innovations of steps k-2L+2 .. k only (the normalised window holds L values, each made
    // from an L-sample mean window), so the time axis is cut into gridDim.y segments: a lane replays the 2L-2 steps before
    // its segment to refill the two windows (zeros before step 0, exactly as at the start of the filter), then emits
    const int seg = (T + (int)gridDim.y - 1) / (int)gridDim.y;
    const int k_lo = (int)blockIdx.y * seg, k_hi = (k_lo + seg < T) ? (k_lo + seg) : T;
    const int k_begin = (k_lo - (2 * L - 2) > 0) ? (k_lo - (2 * L - 2)) : 0;
    if (k_lo >= k_hi) return;
    double in_nxt = a.innovations[lay_scalar(tpos<FLIP>(k_begin, T), lay)];
    double r_nxt = ldg(a.R_series + (size_t)k_begin * a.Sx, voff_x);
    for (int k = k_begin; k < k_hi; k++) {
        const double innov = in_nxt, Rk = r_nxt;
        if (k + 1 < k_hi) {
            in_nxt = a.innovations[lay_scalar(tpos<FLIP>(k + 1, T), lay)];
            r_nxt = ldg(a.R_series + (size_t)(k + 1) * a.Sx, voff_x);     // R_v is not time-flipped (Backward*.m:27)
        }
        const int cnt = (k + 1 < L) ? (k + 1) : L;
        pos = (pos == 0) ? (L - 1) : (pos - 1);
        double *wm = winMean + pos * kWave;
        wm[0] = innov; wm[L * kWave] = innov;
        const double sum = qring_sum<LC, kWave>(wm, L, innov);
        const double mu = sum / (double)cnt;
        const double cc = (innov - mu) * (innov - mu);
        const double ccn = cc / (Rk + kEps);
        double *wn = winCovN + pos * kWave;
        wn[0] = ccn; wn[L * kWave] = ccn;
        const double sumN = qring_sum<LC, kWave>(wn, L, ccn);
        if (k < k_lo) continue;                                             // still refilling the windows
        const double rho = sumN / (double)cnt;
        if (a.rho) a.rho[lay_scalar(k, lay)] = rho;                         // filter-step order also when FLIP
        if (a.f.rho) a.f.rho[lay_scalar(k, lay)] = (float)rho;
    }
}

// The same monitor with NO sequential scan (round 5): one lane per (chain, block of D days), everything in registers.
// rho(k) is a function of the 2L - 1 innovations up to day k: the lane loads the D + 2L - 2 innovations its D days see, forms
// the D + L - 1 window means and normalised squares they need (each window summed newest -> oldest, exactly the ring's order,
// with the zeros the rings hold before step 0), then the D window sums of those.  L - 1 of every D + L - 1 means are formed
// again by the neighbouring lane -- ~210 instructions per (chain, day) instead of ~100 -- but nothing waits on anything: the
// scan kernel's lanes walk ~100 dependent steps each (0.24 ms alone at 9 375 chains, 0.7 ms beside the pinv grid, and the
// smoother's first launch beside it runs 0.1 ms longer); this grid is done in ~0.05 ms.  L = 21 (what every caller passes).
template <int FLIP, int LC, int D>
__global__ __launch_bounds__(kWave) void ekf_monitor_par(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int L = LC, NI = D + 2 * L - 2, NM = D + L - 1;
    if (*dense_flag) return;          // the dense kernels keep the monitor inline
    const int c = a.c0 + blockIdx.x * kWave + (int)threadIdx.x;
    if (c >= a.c0 + a.cn) return;
    const int T = a.T;
    const int k0 = (int)blockIdx.y * D;
    if (k0 >= T) return;
    const int sx = a.x_series ? a.x_series[c] : c;
    const Lay lay = make_lay(a, c);
    double in[NI], ccn[NM];
#pragma unroll
    for (int q = 0; q < NI; q++) {
        const int kq = k0 - (2 * L - 2) + q;
        in[q] = (kq >= 0 && kq < T) ? a.innovations[lay_scalar(tpos<FLIP>(kq, T), lay)] : 0.0;
    }
#pragma unroll
    for (int m = 0; m < NM; m++) {
        const int kp = k0 - (L - 1) + m;                 // the day whose normalised square this is
        const int q = m + L - 1;                         // its innovation
        double sum = in[q];
#pragma unroll
        for (int jj = 1; jj < L; jj++) sum = sum + in[q - jj];
        const int cnt = (kp + 1 < L) ? (kp + 1) : L;
        const bool live = kp >= 0 && kp < T;
        const double Rk = live ? a.R_series[(size_t)kp * a.Sx + sx] : 1.0;     // R_v is not time-flipped (Backward*.m:27)
        const double mu = sum / (double)(live ? cnt : 1);
        const double cc = (in[q] - mu) * (in[q] - mu);
        ccn[m] = live ? cc / (Rk + kEps) : 0.0;          // before step 0 the ring holds zeros
    }
#pragma unroll
    for (int d = 0; d < D; d++) {
        const int k = k0 + d;
        if (k >= T) break;
        const int m = d + L - 1;
        double sumN = ccn[m];
#pragma unroll
        for (int jj = 1; jj < L; jj++) sumN = sumN + ccn[m - jj];
        const int cnt = (k + 1 < L) ? (k + 1) : L;
        const double rho = sumN / (double)cnt;
        if (a.rho) a.rho[lay_scalar(k, lay)] = rho;                          // filter-step order also when FLIP
        if (a.f.rho) a.f.rho[lay_scalar(k, lay)] = (float)rho;
    }
}

// rows of the Jacobian a lane multiplies with: its block row's (bi) and its block column's (bj)
EPI_DEV void qrows(const double (&A)[36], bool hi, double (&R)[3][6])
{
    constexpr int M = 6;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 6; k++) R[r][k] = hi ? A[IXM(3 + r, k)] : A[IXM(r, k)];
}
// C(i, j) = sum_k L(i, k) * Rr(j, k), k = 0..5 ascending, for my block: Lfull = [Ll | Lr] holds rows i (all six
// columns), Rr[c][k] row j = 3 bj + c of the right factor
EPI_DEV void qmul_bt(const blk3 &Ll, const blk3 &Lr, const double (&Rr)[3][6], blk3 &Cb)
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double acc = Ll[r][0] * Rr[c][0];
            acc = fma(Ll[r][1], Rr[c][1], acc);
            acc = fma(Ll[r][2], Rr[c][2], acc);
            acc = fma(Lr[r][0], Rr[c][3], acc);
            acc = fma(Lr[r][1], Rr[c][4], acc);
            acc = fma(Lr[r][2], Rr[c][5], acc);
            Cb[r][c] = acc;
        }
}
// C(i, j) = sum_k Lr(i, k) * R(k, j): Lrow[r][k] row i = 3 bi + r of the left factor (all six columns),
// R = [Rt ; Rb] column block bj (all six rows)
EPI_DEV void qmul_rows(const double (&Lrow)[3][6], const blk3 &Rt, const blk3 &Rb, blk3 &Cb)
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double acc = Lrow[r][0] * Rt[0][c];
            acc = fma(Lrow[r][1], Rt[1][c], acc);
            acc = fma(Lrow[r][2], Rt[2][c], acc);
            acc = fma(Lrow[r][3], Rb[0][c], acc);
            acc = fma(Lrow[r][4], Rb[1][c], acc);
            acc = fma(Lrow[r][5], Rb[2][c], acc);
            Cb[r][c] = acc;
        }
}
// C = L * R with both factors held as blocks: L(i, :) = [Ll | Lr], R(:, j) = [Rt ; Rb]
EPI_DEV void qmul(const blk3 &Ll, const blk3 &Lr, const blk3 &Rt, const blk3 &Rb, blk3 &Cb)
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double acc = Ll[r][0] * Rt[0][c];
            acc = fma(Ll[r][1], Rt[1][c], acc);
            acc = fma(Ll[r][2], Rt[2][c], acc);
            acc = fma(Lr[r][0], Rb[0][c], acc);
            acc = fma(Lr[r][1], Rb[1][c], acc);
            acc = fma(Lr[r][2], Rb[2][c], acc);
            Cb[r][c] = acc;
        }
}

// P(k+1|k) = sym(A P(k|k) A' + Q), Q diagonal (GenericEKF.m:158-161): blocks in, blocks out (cf. predict_cov_sym)
// Qadd[r]: Q_w(i, i) of my block's diagonal entries for the lanes (0,0), (1,1), 0.0 for the off-diagonal blocks
EPI_DEV void qpredict_cov(const Quad &Q, const double (&A)[36], const blk3 &Pp, const double (&Qadd)[3], blk3 &Pm)
{
    double Ar[3][6], Ac[3][6];
    qrows(A, Q.bi, Ar);
    qrows(A, Q.bj, Ac);
    blk3 Pt, Pb, T1, T1l, T1r, G;
    qx3<QP_COL_T>(Pp, Pt);
    qx3<QP_COL_B>(Pp, Pb);
    qmul_rows(Ar, Pt, Pb, T1);                 // T1 = A P
    qx3<QP_ROW_L>(T1, T1l);
    qx3<QP_ROW_R>(T1, T1r);
    qmul_bt(T1l, T1r, Ac, G);                  // G = T1 A'
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) G[r][c] = G[r][c] + ((r == c) ? Qadd[r] : 0.0);
    qsym(G, Pm);
}

// --- the model's NPI vectors, three NPIs per lane ------------------------------------------------------------------
// Lane q of a quad owns NPIs k = q, q + 4, q + 8: it loads u(k), resolves a free control (NaN) and forms u_max(k) - u(k)
// and the slope-term contribution for them; the two sequential reductions over k (the fma chain of (gamma a')(u_max - u)
// and the running a36 -= term) are then run by every lane on operands fetched from their owners by DPP broadcast.
struct QPrm { double dt, beta, gamma, sigma, b, epsilon, slo, ilo, alpha_min, alpha_max; };
struct QNpi {
    double a[3], umin[3], umax[3], ew[3], term[3];   // a(k), u_min(k), u_max(k), epsilon*w(k), gamma*dt*(sigma/2)*a(k)*(u_max(k)-u_min(k))
    double inv_sigma;
    const double *ga;                                // LDS column of this chain: gamma * a(k), k = 0..11, stride kQC
};
EPI_DEV void qload_prm(QPrm &p, QNpi &n, const KArgs &a, int B, int c, const Quad &Q, double *ga_col)
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
    for (int s = 0; s < 3; s++) {
        const int k = Q.q + 4 * s;
        n.a[s] = g(EPI_PRM_A + k); n.umin[s] = g(EPI_PRM_U_MIN + k); n.umax[s] = g(EPI_PRM_U_MAX + k);
        n.ew[s] = p.epsilon * g(EPI_PRM_W_EFF + k);
        // same products, same order as slope_term() / nlin_state_update(): constants of the chain, formed once
        n.term[s] = p.gamma * p.dt * (p.sigma / 2.0) * n.a[s] * (n.umax[s] - n.umin[s]);
        ga_col[k * kQC] = p.gamma * n.a[s];
    }
}
// u(k, t) for my three NPIs (rows beyond n_npi read 0.0 through the descriptor's bounds check, see load_u)
EPI_DEV void qload_u(const KArgs &a, int t, int su, const Quad &Q, double (&u3)[3])
{
    const unsigned rowb = (unsigned)a.Su * 8u, voff = (unsigned)su * 8u + (unsigned)Q.q * rowb;
    const rsrc_t r = mk_rsrc(a.u + (size_t)t * a.n_npi * a.Su, (unsigned)a.n_npi * rowb);
#pragma unroll
    for (int s = 0; s < 3; s++) u3[s] = bld(r, voff, (unsigned)(4 * s) * rowb);
}
template <int BLK>
EPI_DEV void qstore_u(double *__restrict__ dst, const KArgs &a, int t, const Lay &l, const Quad &Q, const double (&u3)[3])
{
    if (!dst) return;
    unsigned voff, rowb;
    const rsrc_t r = qslice<BLK>(dst, t, (unsigned)a.n_npi, l, voff, rowb);
    const unsigned vo = voff + (unsigned)Q.q * rowb;
    if (a.n_npi == kNpi) {
#pragma unroll
        for (int s = 0; s < 3; s++) qst<BLK>(r, vo, (unsigned)(4 * s), rowb, u3[s]);
        return;
    }
#pragma unroll
    for (int s = 0; s < 3; s++)
        if (Q.q + 4 * s < a.n_npi) qst<BLK>(r, vo, (unsigned)(4 * s), rowb, u3[s]);
}
// bang-bang substitution of my NaN controls, OptControlled.m:49-58 (resolve_control); phi(k) is kept for the slope term
EPI_DEV void qresolve(const QPrm &p, const QNpi &n, const ModelFlags &mf, double s6, const double (&u3)[3], double (&ur)[3],
                      double (&phi)[3])
{
    const double gs6 = p.gamma * s6;
#pragma unroll
    for (int s = 0; s < 3; s++) {
        phi[s] = n.ew[s] - gs6 * n.a[s];
        const bool lo = mf.phi_ge ? (phi[s] >= 0.0) : (phi[s] > 0.0);
        ur[s] = is_nan(u3[s]) ? (lo ? n.umin[s] : n.umax[s]) : u3[s];
    }
}
template <int J> EPI_DEV double qbc(double v) { return qx<EPI_QP(J, J, J, J)>(v); }
// all twelve values of a three-per-lane vector, in NPI order k = 4 s + owner lane
EPI_DEV void qgather12(const double (&v3)[3], double (&v)[kNpi])
{
#pragma unroll
    for (int s = 0; s < 3; s++) {
        v[4 * s + 0] = qbc<0>(v3[s]); v[4 * s + 1] = qbc<1>(v3[s]);
        v[4 * s + 2] = qbc<2>(v3[s]); v[4 * s + 3] = qbc<3>(v3[s]);
    }
}
// (gamma a') (u_max - u): the k-ascending fma chain of nlin_state_update()
EPI_DEV double qdot(const QNpi &n, const double (&ur)[3])
{
    double d3[3], d[kNpi];
#pragma unroll
    for (int s = 0; s < 3; s++) d3[s] = n.umax[s] - ur[s];
    qgather12(d3, d);
    double dot = n.ga[0] * d[0];
#pragma unroll
    for (int k = 1; k < kNpi; k++) dot = fma(n.ga[k * kQC], d[k], dot);
    return dot;
}
// slope_term(): a36 -= term(k) (+= for the time-flipped models) for every free control with |phi(k)| < 1/sigma, k ascending.
// A control that does not qualify contributes -(+0.0): a36 never is -0.0, so that leaves it bit-wise unchanged.
template <int FLIP>
EPI_DEV double qslope(const QNpi &n, const double (&u3)[3], const double (&phi)[3])
{
    const bool any_free = is_nan(u3[0]) || is_nan(u3[1]) || is_nan(u3[2]);
    if (__builtin_amdgcn_ballot_w64(any_free) == 0ull) return 0.0;      // historic days: no lane of the wave has one
    double tm3[3], tm[kNpi];
#pragma unroll
    for (int s = 0; s < 3; s++)
        tm3[s] = (is_nan(u3[s]) && phi[s] > -n.inv_sigma && phi[s] < n.inv_sigma) ? n.term[s] : 0.0;
    qgather12(tm3, tm);
    double a36 = 0.0;
#pragma unroll
    for (int k = 0; k < kNpi; k++) a36 = FLIP ? (a36 + tm[k]) : (a36 - tm[k]);
    return a36;
}

// ---------------------------------------------------------------------------
// forward pass
// ---------------------------------------------------------------------------
// MON = 0: the innovation monitor runs as a kernel of its own (ekf_monitor: r_mode 1 only) -- no windows here
// SOLO = 1: the kernel claims more than half of the register file (a clobbered high accumulation register), so that two of
// its waves never share a SIMD.  With its ~230 registers the MON = 0 kernel would otherwise be packed two to a SIMD while
// other SIMDs idle -- measured 1.89 instead of ~1.2 ms at 586 waves.  SOLO = 0 for grids beyond one wave per SIMD.
template <int FLIP, int BLK, int LC, int MON, int SOLO>
__global__ __launch_bounds__(kWave) void ekf_fwd_quad(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int M = 6;
    extern __shared__ double lds[];   // MON: windows [3][2 L][kQC]; then gamma*a [12][kQC], one column per chain
    if (SOLO) asm volatile("" ::: "a130");     // 228 + 131 registers: not even an eks_pinv wave (168) fits beside it
    if (*dense_flag) return;
    Quad Q;
    Q.q = threadIdx.x & 3; Q.lc = threadIdx.x >> 2; Q.bi = (Q.q >> 1) != 0; Q.bj = (Q.q & 1) != 0;
    const int c = a.c0 + blockIdx.x * kQC + Q.lc;
    if (c >= a.c0 + a.cn) return;               // whole quads leave together
    const int B = a.B, T = a.T, L = LC ? LC : a.L;
    const int sx = a.x_series ? a.x_series[c] : c;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    QPrm p;
    QNpi np;
    qload_prm(p, np, a, B, c, Q, lds + (size_t)(MON ? 6 * L : 0) * kQC + Q.lc);
    const double v_bar = a.prm[(size_t)EPI_PRM_V_BAR * B + c];
    const double beta = a.prm[(size_t)EPI_PRM_BETA_EKF * B + c];
    const double gamma = a.prm[(size_t)EPI_PRM_GAMMA_EKF * B + c];

    double sk_minus[M], Qadd[3];
    blk3 Pm;
#pragma unroll
    for (int i = 0; i < M; i++) sk_minus[i] = a.s_init[(size_t)i * B + c];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int i = (Q.bi ? 3 : 0) + r;
        const double qd = a.Q[(size_t)IXM(i, i) * B + c];
        Qadd[r] = (Q.bi == Q.bj) ? qd : 0.0;
    }
    // Ps_init is bit-wise symmetric (ekf_precheck); the packed kernel reads its upper triangle, so do we
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            const int i = (Q.bi ? 3 : 0) + r, j = (Q.bj ? 3 : 0) + cc;
            Pm[r][cc] = a.Ps_init[(size_t)(i < j ? IXM(i, j) : IXM(j, i)) * B + c];
        }

    // time segments: see ekf_fwd_sym
    const int k_begin = a.k_begin, k_end = (a.k_end > 0 && a.k_end < T) ? a.k_end : T;
    if (k_begin > 0) {
        qload_vec<BLK>(a.S_MINUS, tpos<FLIP>(k_begin, T), lay, sk_minus);
        const QOff o36 = qoffsets<BLK>(Q, lay, true);
        qload_sym_blk(a.P_MINUS, tpos<FLIP>(k_begin, T), lay, o36, 36, Pm);
    }
    // three windows, each a 2L-long column per chain (see qring_sum); `pos` = where the newest sample sits
    double *winMean = lds + Q.lc, *winCov = lds + (size_t)2 * L * kQC + Q.lc, *winCovN = lds + (size_t)4 * L * kQC + Q.lc;
    if (MON)
        for (int j = 0; j < 2 * L; j++) { winMean[j * kQC] = 0.0; winCov[j * kQC] = 0.0; winCovN[j * kQC] = 0.0; }
    int pos = 0;
    const bool fixed_R = (a.r_mode == 0);
    const double R_v = fixed_R ? a.R_scalar[c] : 0.0;
    double R_next = R_v;

    const unsigned voff_x = (unsigned)sx * 8u;
    double x_nxt = ldg(a.x + (size_t)tpos<FLIP>(k_begin, T) * a.Sx, voff_x);
    double r_nxt = fixed_R ? 0.0 : ldg(a.R_series + (size_t)k_begin * a.Sx, voff_x);
    double u_nxt[3];
    qload_u(a, tpos<FLIP>(k_begin, T), su, Q, u_nxt);

    for (int k = k_begin; k < k_end; k++) {
        const int t = tpos<FLIP>(k, T);
        const double Rk = fixed_R ? R_next : r_nxt;
        const double xk = x_nxt;
        double u_in[3];
#pragma unroll
        for (int qq = 0; qq < 3; qq++) u_in[qq] = u_nxt[qq];
        if (k + 1 < T) {
            const int tn = tpos<FLIP>(k + 1, T);
            x_nxt = ldg(a.x + (size_t)tn * a.Sx, voff_x);
            if (!fixed_R) r_nxt = ldg(a.R_series + (size_t)(k + 1) * a.Sx, voff_x);
            qload_u(a, tn, su, Q, u_nxt);
        }

        qstore_vec<BLK>(a.S_MINUS, t, lay, Q, sk_minus);
        qstore_blk<BLK>(a.P_MINUS, t, lay, Q, Pm);

        double C[M];
        obs_jacobian<M>(a.mf, sk_minus, C);                 // C(4:6) == 0
        const double xk_minus = predict_obs<M>(a.mf, sk_minus, v_bar);

        double innov, K[M], sk_plus[M];
        blk3 Pp;
        const bool valid = !is_nan(xk);
        if (valid) {
            innov = xk - xk_minus;
            // P C' for my block row: formed by the lane that owns block (bi, 0) -- C(4:6) = 0 -- and handed to the right
            double PCr[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                double acc = Pm[r][0] * C[0];
                acc = fma(Pm[r][1], C[1], acc);
                acc = fma(Pm[r][2], C[2], acc);
                PCr[r] = qx<QP_ROW_L>(acc);
            }
            double PC0[3];
#pragma unroll
            for (int r = 0; r < 3; r++) PC0[r] = qbc<0>(PCr[r]);
            double CPCt = PC0[0] * C[0];
            CPCt = fma(PC0[1], C[1], CPCt);
            CPCt = fma(PC0[2], C[2], CPCt);
            const double den = CPCt + gamma * Rk;
            double Kr[3], Kc[3];
#pragma unroll
            for (int r = 0; r < 3; r++) Kr[r] = PCr[r] / den;
#pragma unroll
            for (int r = 0; r < 3; r++) {
                K[r] = qbc<0>(Kr[r]);
                K[3 + r] = qbc<2>(Kr[r]);
                Kc[r] = Q.bj ? K[3 + r] : K[r];
            }
            // (I - K C): my block row's and my block column's rows of its first three columns
            double IKr[3][3], IKc[3][3];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    IKr[r][j] = ((r == j && !Q.bi) ? 1.0 : 0.0) - Kr[r] * C[j];
                    IKc[r][j] = ((r == j && !Q.bj) ? 1.0 : 0.0) - Kc[r] * C[j];
                }
            // Joseph form :127, cf. ekf_fwd_sym: T1 = (I - K C) P, F = (T1 (I - K C)' + K R K') / gamma
            blk3 P0, T1, T1l, F;
            qx3<QP_COL_T>(Pm, P0);
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int cc = 0; cc < 3; cc++) {
                    double acc = IKr[r][0] * P0[0][cc];
                    acc = fma(IKr[r][1], P0[1][cc], acc);
                    acc = fma(IKr[r][2], P0[2][cc], acc);
                    const double plus = acc + Pm[r][cc];            // + 1 * P(i, j) for i >= 3
                    T1[r][cc] = Q.bi ? plus : acc;
                }
            qx3<QP_ROW_L>(T1, T1l);
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int cc = 0; cc < 3; cc++) {
                    double acc = T1l[r][0] * IKc[cc][0];
                    acc = fma(T1l[r][1], IKc[cc][1], acc);
                    acc = fma(T1l[r][2], IKc[cc][2], acc);
                    const double plus = acc + T1[r][cc];            // + T1(i, j) for j >= 3
                    acc = Q.bj ? plus : acc;
                    F[r][cc] = (acc + (Kr[r] * Rk) * Kc[cc]) / gamma;
                }
            qsym(F, Pp);
#pragma unroll
            for (int i = 0; i < M; i++) sk_plus[i] = sk_minus[i] + K[i] * innov;
        } else {
            innov = 0.0;
#pragma unroll
            for (int i = 0; i < M; i++) { K[i] = 0.0; sk_plus[i] = sk_minus[i]; }
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int cc = 0; cc < 3; cc++) Pp[r][cc] = Pm[r][cc];
        }
        qstore_vec<BLK>(a.K_GAIN, t, lay, Q, K);
        qstore_scalar(a.innovations, t, lay, innov);
        state_hard_margins<M>(p, sk_plus);
        qstore_vec<BLK>(a.S_PLUS, t, lay, Q, sk_plus);
        qstore_blk<BLK>(a.P_PLUS, t, lay, Q, Pp);

        // s(k+1|k) = NlinStateUpdate(u, s+), A = StateJacobians(u, s+)  :155-157 (phi(k) serves both)
        double u_app[3], phi[3];
        qresolve(p, np, a.mf, sk_plus[5], u_in, u_app, phi);
        state_map<M, FLIP>(p, qdot(np, u_app), sk_plus, sk_minus);
        qstore_u<BLK>(a.u_opt, a, t, lay, Q, u_app);
        {
            double A[M * M];
            jacobian_entries<M, FLIP>(p, sk_plus, qslope<FLIP>(np, u_in, phi), A);
            qpredict_cov(Q, A, Pp, Qadd, Pm);
        }
        state_hard_margins<M>(p, sk_minus);

        if (!MON) continue;
        // innovation monitor :172-179 (identical arithmetic to ekf_fwd_sym; the four lanes of a quad hold the same numbers)
        const int cnt = (k + 1 < L) ? (k + 1) : L;
        pos = (pos == 0) ? (L - 1) : (pos - 1);
        double *wm = winMean + pos * kQC;
        wm[0] = innov; wm[L * kQC] = innov;
        const double sum = qring_sum<LC>(wm, L, innov);
        const double mu = sum / (double)cnt;
        const double cc2 = (innov - mu) * (innov - mu);
        const double ccn = cc2 / (Rk + kEps);
        double *wc = winCov + pos * kQC, *wn = winCovN + pos * kQC;
        wc[0] = cc2; wc[L * kQC] = cc2;
        wn[0] = ccn; wn[L * kQC] = ccn;
        // the two remaining window sums are independent: even lanes of the quad add up the normalised window, odd
        // lanes the plain one (needed for the adaptive R only), each strictly newest -> oldest
        const bool odd = fixed_R && (Q.q & 1);
        const double s2 = qring_sum<LC>(odd ? wc : wn, L, odd ? cc2 : ccn);
        const double sumN = qbc<0>(s2);
        qstore_scalar(a.rho, k, lay, sumN / (double)cnt);     // filter-step order also when FLIP (see ekf_fwd_sym)
        if (fixed_R) {
            const double sumC = qbc<1>(s2);
            if (beta != 1.0 && valid && k < T - 1) {
                R_next = beta * Rk + (1.0 - beta) * (sumC / (double)cnt);
            } else {
                R_next = R_v;
            }
        }
    }
    if (k_end < T) {       // hand-over to the next time segment
        qstore_vec<BLK>(a.S_MINUS, tpos<FLIP>(k_end, T), lay, Q, sk_minus);
        qstore_blk<BLK>(a.P_MINUS, tpos<FLIP>(k_end, T), lay, Q, Pm);
    }
}

// ---------------------------------------------------------------------------
// backward recursion (X = pinv(P_MINUS) comes from eks_pinv, packed)
// ---------------------------------------------------------------------------
template <int FLIP, int BLK>
__global__ __launch_bounds__(kWave) void eks_bwd_quad(const KArgs a, const int *__restrict__ dense_flag)
{
    constexpr int M = 6;
    __shared__ double galds[kNpi * kQC];   // gamma * a(k), one column per chain
    if (*dense_flag) return;
    Quad Q;
    Q.q = threadIdx.x & 3; Q.lc = threadIdx.x >> 2; Q.bi = (Q.q >> 1) != 0; Q.bj = (Q.q & 1) != 0;
    const int c = a.c0 + blockIdx.x * kQC + Q.lc;
    if (c >= a.c0 + a.cn) return;
    const int B = a.B, T = a.T;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    QPrm p;
    QNpi np;
    qload_prm(p, np, a, B, c, Q, galds + Q.lc);
    const QOff o36 = qoffsets<BLK>(Q, lay, true), o21 = qoffsets<BLK>(Q, lay, false);

    // smoother steps of this launch: k = k_from down to k_to (see eks_bwd_sym); k_from = T - 2 starts from the terminal
    // condition, a later launch resumes from the hand-over rows (every lane of the quad its own 3 x 3 block)
    const int k_from = a.bk_from, k_to = a.bk_to;
    const size_t hp = (size_t)a.hand_pitch;
    int st_guard = 0, st_cap = 0, min_rank = M;
    double Ss[M];
    blk3 Ps;
    const int tT = tpos<FLIP>(T - 1, T);
    if (k_from < T - 2) {
#pragma unroll
        for (int i = 0; i < M; i++) Ss[i] = a.hand_s[(size_t)i * hp + c];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                const int i = (Q.bi ? 3 : 0) + r, j = (Q.bj ? 3 : 0) + cc;
                Ps[r][cc] = a.hand_p[(size_t)IXM(i, j) * hp + c];
            }
        const int word = a.hand_i[c];
        st_guard = word & 1; st_cap = (word >> 1) & 1; min_rank = word >> 8;
    } else {
        // terminal conditions GenericEKF.m:189-202 (Ps_final symmetric in values and NaN pattern: ekf_precheck)
        qload_vec<BLK>(a.S_PLUS, tT, lay, Ss);
#pragma unroll
        for (int i = 0; i < M; i++) {
            const double f = a.s_final[(size_t)i * B + c];
            if (!is_nan(f)) Ss[i] = f;
        }
        qload_sym_blk(a.P_PLUS, tT, lay, o36, 36, Ps);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                const int i = (Q.bi ? 3 : 0) + r, j = (Q.bj ? 3 : 0) + cc;
                const double f = a.Ps_final[(size_t)(i < j ? IXM(i, j) : IXM(j, i)) * B + c];
                if (!is_nan(f)) Ps[r][cc] = f;
            }
        qstore_vec<BLK>(a.S_SMOOTH, tT, lay, Q, Ss);
        qstore_blk<BLK>(a.P_SMOOTH, tT, lay, Q, Ps);
        if (a.u_opt_smooth) {
            const double z[3] = {0.0, 0.0, 0.0};
            qstore_u<BLK>(a.u_opt_smooth, a, tT, lay, Q, z);
        }
        qstore_scalar(a.pinv_rank, tT, lay, (int32_t)-1);
    }

    // Everything step k reads is requested one iteration ahead (EPI_QUAD_BWD_PF): a lone wave then never sits through a
    // memory round trip at the top of a step.  Two register sets used alternately (the loop body exists twice), so the
    // prefetched values are consumed where they landed -- copying them costs more than the latency (measured).
    struct In { double Sp[M], Sm1[M], u[3]; blk3 Pp, X, Pm1; int rk; };
    auto fetch = [&](int k, In &d) {
        const int t = tpos<FLIP>(k, T), t1 = tpos<FLIP>(k + 1, T);
        qload_vec<BLK>(a.S_PLUS, t, lay, d.Sp);
        qload_u(a, t, su, Q, d.u);
        d.rk = a.rankbuf[lay_scalar(t1, lay)];
        qload_sym_blk(a.P_PLUS, t, lay, o36, 36, d.Pp);
        qload_sym_blk(a.X, t1, lay, o21, 21, d.X);           // (garbage where the :211 guard fired, rk < 0: unused)
        qload_vec<BLK>(a.S_MINUS, t1, lay, d.Sm1);
        qload_sym_blk(a.P_MINUS, t1, lay, o36, 36, d.Pm1);
    };
    auto step = [&](int k, In &cur) {
        const int t = tpos<FLIP>(k, T);
        const double (&Sp)[M] = cur.Sp;
        const double (&Sm1)[M] = cur.Sm1;
        const double (&u_in)[3] = cur.u;
        const blk3 &Pp = cur.Pp, &X = cur.X, &Pm1 = cur.Pm1;
        const int rk = cur.rk;

        double A[M * M];
        {
            double ur_unused[3], phi[3];
            qresolve(p, np, a.mf, Sp[5], u_in, ur_unused, phi);
            jacobian_entries<M, FLIP>(p, Sp, qslope<FLIP>(np, u_in, phi), A);   // :206
        }
        blk3 J;
        int rank = -1;
        if (rk < 0) {                                          // non-finite P_MINUS guard :211-213
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int cc = 0; cc < 3; cc++) J[r][cc] = 0.0;
            st_guard = 1;
        } else {
            // J = (P+ A') X  :215
            double Ac[3][6];
            qrows(A, Q.bj, Ac);
            blk3 Pl, Pr, PA, PAl, PAr, Xt, Xb;
            qx3<QP_ROW_L>(Pp, Pl);
            qx3<QP_ROW_R>(Pp, Pr);
            qmul_bt(Pl, Pr, Ac, PA);
            qx3<QP_ROW_L>(PA, PAl);
            qx3<QP_ROW_R>(PA, PAr);
            qx3<QP_COL_T>(X, Xt);
            qx3<QP_COL_B>(X, Xb);
            qmul(PAl, PAr, Xt, Xb, J);
            rank = rk & 0xff;
            st_cap |= (rk >> 8) & 1;
            min_rank = rank < min_rank ? rank : min_rank;
        }
        // S_SMOOTH(k) = clamp(s+ + J (s_s(k+1) - s-(k+1)))  :218-221.  The chain over j = 0..5 of row i starts in the
        // lane that owns J(i, 0:2) and is continued by the lane that owns J(i, 3:5)
        double Sn[M];
        {
            double dv[M];
#pragma unroll
            for (int i = 0; i < M; i++) dv[i] = Ss[i] - Sm1[i];
            double accr[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                double accl = J[r][0] * dv[0];
                accl = fma(J[r][1], dv[1], accl);
                accl = fma(J[r][2], dv[2], accl);
                const double from_left = qx<QP_ROW_L>(accl);
                double acc = fma(J[r][0], dv[3], from_left);
                acc = fma(J[r][1], dv[4], acc);
                acc = fma(J[r][2], dv[5], acc);
                accr[r] = acc;                                 // complete in the lanes with bj = 1
            }
#pragma unroll
            for (int r = 0; r < 3; r++) {
                Sn[r] = Sp[r] + qx<EPI_QP(1, 1, 1, 1)>(accr[r]);
                Sn[3 + r] = Sp[3 + r] + qx<EPI_QP(3, 3, 3, 3)>(accr[r]);
            }
        }
        state_hard_margins<M>(p, Sn);
        {
            // P_SMOOTH(k) = sym(P+ - (J D) J'),  D = P_MINUS(k+1) - P_SMOOTH(k+1)   :223-226
            blk3 D, Jl, Jr, Dt, Db, T1, T1l, T1r, Jcl, Jcr, F;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int cc = 0; cc < 3; cc++) D[r][cc] = Pm1[r][cc] - Ps[r][cc];
            qx3<QP_ROW_L>(J, Jl);
            qx3<QP_ROW_R>(J, Jr);
            qx3<QP_COL_T>(D, Dt);
            qx3<QP_COL_B>(D, Db);
            qmul(Jl, Jr, Dt, Db, T1);
            qx3<QP_ROW_L>(T1, T1l);
            qx3<QP_ROW_R>(T1, T1r);
            qx3<QP_CJ_L>(J, Jcl);
            qx3<QP_CJ_R>(J, Jcr);
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int cc = 0; cc < 3; cc++) {
                    double acc = T1l[r][0] * Jcl[cc][0];
                    acc = fma(T1l[r][1], Jcl[cc][1], acc);
                    acc = fma(T1l[r][2], Jcl[cc][2], acc);
                    acc = fma(T1r[r][0], Jcr[cc][0], acc);
                    acc = fma(T1r[r][1], Jcr[cc][1], acc);
                    acc = fma(T1r[r][2], Jcr[cc][2], acc);
                    F[r][cc] = Pp[r][cc] - acc;
                }
            qsym(F, Ps);
        }
#pragma unroll
        for (int i = 0; i < M; i++) Ss[i] = Sn[i];
        qstore_scalar(a.pinv_rank, t, lay, (int32_t)rank);
        qstore_vec<BLK>(a.S_SMOOTH, t, lay, Q, Ss);
        qstore_blk<BLK>(a.P_SMOOTH, t, lay, Q, Ps);
        if (a.u_opt_smooth) {                                  // :229 -- only the control NlinStateUpdate returns is kept
            double ur[3], phi_unused[3];
            qresolve(p, np, a.mf, Ss[5], u_in, ur, phi_unused);
            qstore_u<BLK>(a.u_opt_smooth, a, t, lay, Q, ur);
        }
    };
#if EPI_QUAD_BWD_PF
    {
        In bufA, bufB;
        int k = k_from;
        if (k >= k_to) fetch(k, bufA);
        while (k >= k_to) {
            if (k > k_to) fetch(k - 1, bufB);
            step(k, bufA);
            if (--k < k_to) break;
            if (k > k_to) fetch(k - 1, bufA);
            step(k, bufB);
            --k;
        }
    }
#else
    for (int k = k_from; k >= k_to; k--) {
        In cur;
        fetch(k, cur);
        step(k, cur);
    }
#endif
    if (k_to > 0) {        // hand-over to the launch that continues with step k_to - 1
        if (Q.q == 0) {
#pragma unroll
            for (int i = 0; i < M; i++) a.hand_s[(size_t)i * hp + c] = Ss[i];
            a.hand_i[c] = st_guard | (st_cap << 1) | (min_rank << 8);
        }
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                const int i = (Q.bi ? 3 : 0) + r, j = (Q.bj ? 3 : 0) + cc;
                a.hand_p[(size_t)IXM(i, j) * hp + c] = Ps[r][cc];
            }
    } else if (a.status) a.status[c] = st_guard | (st_cap << 1) | (min_rank << 8);
}
