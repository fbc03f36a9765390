// epiekf.hip -- kernels and C ABI of libepiekf.so (see include/epiekf.h).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared epiekf.hip -o libepiekf.so
//
// Kernels
//   ekf_fwd<M,FLIP,GENERIC>  forward EKF loop, Tools/GenericExtendedKalmanFilter.m:98-186
//                            (GENERIC=0: Tools/NewCaseEKFEstimatorWithOptimalNPI.m:37-113)
//   eks_bwd<M,FLIP,GENERIC>  backward smoother loop, GenericEKF.m:189-230 (NewCase...m:115-139)
// Both keep one chain per lane (ekf_device.hpp); 64-thread workgroups so that the
// B/64 waves spread over the 1024 SIMDs of the chip as evenly as the batch allows.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>
#include "ekf_device.hpp"

namespace epi {

struct ModelInfo { int m, flipped, lo_is_zero, phi_ge, obs_clamp, obs_fixed, generic; };
static const ModelInfo MODEL_TABLE[6] = {
    /* SIA3             */ {3, 0, 0, 0, 1, 0, 1},
    /* SIA6             */ {6, 0, 1, 0, 1, 0, 1},
    /* SIA3_BWD         */ {3, 1, 1, 0, 1, 0, 1},
    /* SIA6_BWD         */ {6, 1, 1, 0, 1, 0, 1},
    /* NEWCASE6         */ {6, 0, 1, 1, 1, 0, 0},
    /* NEWCASE6_CODEGEN */ {6, 0, 1, 1, 0, 1, 0},
};

struct KArgs {
    int B, T, Sx, Su, n_npi, L, r_mode;
    int q_mode;   // 1: Q is [T][m*m][B], Q(:,:,k) of filter step k (GenericEKF.m:63-73); dense kernels only
    ModelFlags mf;
    const int32_t *x_series, *u_series;
    const double *x, *u, *R_series, *R_scalar, *prm;
    const double *s_init, *Ps_init, *s_final, *Ps_final, *Q;   // already swapped for flipped models
    // forward quantities (outputs or workspace; never NULL)
    double *S_MINUS, *S_PLUS, *P_MINUS, *P_PLUS;
    // bit 0 / bit 1: P_MINUS / P_PLUS is workspace, not a caller's output -- the packed kernels then store its upper
    // triangle only (what eks_pinv and eks_bwd_sym read back): 15 of 36 rows less to write per array and step
    int ws_upper;
    // optional outputs (NULL = not stored)
    double *u_opt, *u_opt_smooth, *S_SMOOTH, *P_SMOOTH, *K_GAIN, *innovations, *rho;
    int32_t *pinv_rank, *status;
    // smoother intermediates (workspace): X = pinv(P_MINUS(:,:,k+1)) stored at the position of step k+1,
    // rankbuf = kept rank | sweep-cap flag << 8 (or -1 where the non-finite guard of :211 fired)
    double *X;
    int32_t *rankbuf;
    int pinv_pos0;
    // generic models: word set by ekf_precheck when the batch must take the dense kernels (non-symmetric
    // Ps_init or non-diagonal Q_w); NULL = dense kernels always run (NewCase models)
    int *dense_flag;
    // chain range [c0, c0 + cn) this launch covers (a call may be split into chunks on two streams)
    int c0, cn;
    // chain-blocked layout of the output / workspace arrays (see Lay): lanes per block and blocks per time slice
    int blk, nblk;
    // lanes used per 64-thread workgroup (<= 64).  When the batch needs more than one round of resident waves,
    // the host narrows the waves so that the rounds are equally full (launch_chain): the sequential kernels are
    // bound by HBM / per-CU memory throughput, which scales with active lanes, not by wave count.
    int lw;
    int quad;   // 6-state generic models: four lanes per chain (ekf_quad.hpp) instead of one
    int wave;   // 6-state generic models: one WAVEFRONT per chain (ekf_wave.hpp)
    int hex;    // 6-state generic models: six lanes per chain, ten chains per wavefront (ekf_hex.hpp)
    int hexw;   // hex kernels: days per addressing window (0: as many as 2 GiB hold, see HexWin; the tests set a few days)
    // epi_batch_desc.storage = 1: the caller's outputs are fp32 arrays (same layouts, 4-byte elements); each selected
    // one is the fp64 result rounded once.  The four forward quantities the smoother reads back are then always fp64
    // workspace (S_MINUS ... P_PLUS above) and their fp32 copies are extra stores.  Packed (sym) kernels only.
    int stor;   // 1: fp32 storage (see F32 below)
    int k_begin, k_end;   // filter steps [k_begin, k_end) of this forward launch (k_end <= 0: up to T); eks_pinv: steps from pinv_step0
    int pinv_step0;
    // smoother steps of this backward launch (packed / quad kernels): k = bk_from, bk_from - 1, ..., bk_to.  bk_from = T - 2
    // starts from the terminal condition (:189-202); a later launch resumes from the S_SMOOTH / P_SMOOTH / status word the
    // previous one left in the hand-over workspace (hand_s [m][Bp], hand_p [m*m][Bp], hand_i [Bp]) -- the same bits
    int bk_from, bk_to;
    double *hand_s, *hand_p;
    int32_t *hand_i;
    int hand_pitch;       // Bp: chains the hand-over rows are sized for
    // epi_batch_desc.exact_nonfinite: after the packed kernels, chains whose covariance went non-finite (status bit 0) are
    // run again by the dense kernels, in place.  `only` [B] (workspace): the dense fwd / pinv / bwd launches of that second
    // pass return at once for every chain whose word is 0; only[B] counts the marked chains
    const int32_t *only;
    int32_t *only_buf;
    int mon_defer;   // 1: this launch does not enqueue ekf_monitor itself (the caller does, later)
    int mon_hoist;   // 1: the packed / quad forward kernels skip the innovation monitor, ekf_monitor replays it (r_mode 1)
    int mon_scan;    // test hook (epi_batch_desc.test_flags bit 1): the scan kernel ekf_monitor whatever the batch size
    struct F32 { float *u_opt, *u_opt_smooth, *S_MINUS, *S_PLUS, *S_SMOOTH, *P_MINUS, *P_PLUS, *P_SMOOTH, *K_GAIN, *innovations, *rho; } f;
};

// position in the caller's time axis of filter step k (flipped wrappers run the
// generic filter on time-reversed u/x and reverse every output: Backward*.m:19-40)
template <int FLIP> EPI_DEV int tpos(int k, int T) { return FLIP ? (T - 1 - k) : k; }

// Addressing.  Every array is [T][rows][B] (chain-minor), so an access is
//     slice base (wave-uniform: kernel argument + uniform time index)  +  row * B * 8 (uniform)  +  8 * chain (per lane).
// The per-step slice of an array is addressed through a buffer resource descriptor held in SGPRs
// (cdna_hip_programming.md T8/T20): `buffer_load/store_dwordx2 v_data, v_off, s[rsrc], s_rowoff offen` with ONE shared
// 32-bit per-lane byte offset, the row offset in an SGPR and hardware bounds checking -- no 64-bit per-lane address
// arithmetic and no VGPR pairs per access.  A slice must stay below 4 GiB: rows * B * 8 < 2^32  =>  B <= 2^23.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
EPI_DEV rsrc_t mk_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, /*stride*/ 0, bytes, 0x00020000);
}
EPI_DEV double bld(rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
// Cache policy of the big streams (aux bit 1 = `nt` on gfx950).  Every output is written once and read back -- if at all --
// tens of GB later, so the stores are NON-TEMPORAL: they do not push the shared input series, the model constants and the
// smoother's own read stream out of L2 / Infinity Cache.  Measured on the headline sweep, alternating builds on one box
// (round 4): 17.4-18.0 -> 16.3-17.0 ms per pass (smoother 8.0 -> 7.3 ms, forward kernel -0.2 ms), reduced outputs 13.7 -> 13.4,
// nothing slower.  Non-temporal LOADS on every load: 20.3 ms (the shared series stop being cached) -- so only the streams a launch
// reads exactly once (the stored forward quantities and X in the smoothers, P_MINUS in the pinv grid) are loaded `nt` (bld_s):
// another 16.2-17.0 -> 15.8-16.1 ms, the next pass's forward kernel finding its shared inputs still in cache.
// Round 4, later: `nt` + `sc1` (aux 18: non-temporal AND written through at system scope, so the line does not stay dirty
// in L2) -- level with `nt` alone on boxes / placements where the pass runs at its best (15.5-15.9 ms) and 0.4-0.8 ms faster
// where the L2's tag pipeline stalls on the concurrently streamed arrays ("where the arrays lie"): forward kernel
// 7.05 -> 6.67 ms on one box, smoother 7.8 -> 7.0 ms on another, three boxes 16.1-16.9 -> 15.6-15.9 ms per pass.  `sc1` on the
// stream LOADS as well (aux 18) brings the slow mode back; `sc0 | sc1 | nt` (19) on them is level with `nt` alone.
#ifndef EPI_ST_AUX
#define EPI_ST_AUX 18
#endif
#ifndef EPI_ST32_AUX
#define EPI_ST32_AUX 18           // the fp32-storage twins of the stores (BASELINE config 5)
#endif
// (`sc0 | sc1 | nt` = 19 on the stream loads: -0.24 +- 0.10 ms per headline pass against `nt` alone over 20 paired runs on four
// boxes, smoother -0.13, slow outliers 16.6-17.2 -> at most 16.8 ms; `sc1 | nt` = 18 brings the smoother's slow mode back)
#ifndef EPI_LD_STREAM_AUX
#define EPI_LD_STREAM_AUX 19
#endif
EPI_DEV void bst(rsrc_t r, unsigned voff, unsigned soff, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, EPI_ST_AUX);
}
// non-temporal only: for layouts whose rows are not whole cache lines (the hex shape's ten-chain blocks, ekf_hex.hpp), where
// writing through (`sc1`) turns every partial line into a memory transaction of its own
EPI_DEV void bst_nt(rsrc_t r, unsigned voff, unsigned soff, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, 2);
}
// a load of data that this launch reads exactly once (stored forward quantities, X)
EPI_DEV double bld_s(rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, EPI_LD_STREAM_AUX));
}
EPI_DEV double ldg(const double *__restrict__ row, unsigned voff)
{
    return *(const double *)((const char *)row + voff);
}
EPI_DEV void stg(double *__restrict__ row, unsigned voff, double v) { *(double *)((char *)row + voff) = v; }
EPI_DEV int ldg_i(const int32_t *__restrict__ row, unsigned voff4) { return *(const int32_t *)((const char *)row + voff4); }
EPI_DEV void stg_i(int32_t *__restrict__ row, unsigned voff4, int v) { *(int32_t *)((char *)row + voff4) = v; }

// Chain-blocked layout of the big per-(time, row, chain) arrays (the 11 outputs and the workspace).  Element
// (t, row, c) of an array with `rows` rows lives at double index
//     ((t * nblk + c / blk) * rows + row) * blk + c % blk,        nblk = ceil(B / blk).
// blk = B (nblk = 1) is the classic [T][rows][B].  blk = 8 makes the rows of eight neighbouring chains one contiguous
// rows * 64-byte block, so that the ~100 stores a wave issues per step fill whole DRAM pages instead of feeding ~100
// concurrent row streams (profiles/layout_probe: 5.5 ms instead of 7.7-8.5 ms for the forward kernel's 32 GB of
// stores).  Arrays with one row ([T][B]: innovations, rho, rank words) are simply [T][nblk * blk].
struct Lay { unsigned blk, nblk, cb, cr, c, bp; };
EPI_DEV Lay make_lay(const KArgs &a, int c)
{
    Lay l;
    l.blk = (unsigned)a.blk; l.nblk = (unsigned)a.nblk; l.c = (unsigned)c;
    l.cb = (unsigned)c / l.blk; l.cr = (unsigned)c - l.cb * l.blk; l.bp = l.blk * l.nblk;
    return l;
}
// buffer descriptor of time slice t of an array with `rows` rows, this lane's byte offset in it, and the row pitch
EPI_DEV rsrc_t lay_slice(const double *p, int t, unsigned rows, const Lay &l, unsigned &voff, unsigned &rowb)
{
    rowb = l.blk * 8u;
    voff = (l.cb * rows * l.blk + l.cr) * 8u;
    return mk_rsrc(p + (size_t)t * rows * l.bp, rows * l.bp * 8u);
}
// inputs that are per (time, row, chain) -- a time-varying Q_w -- keep the classic [T][rows][B]
EPI_DEV Lay lay_classic(int B, int c)
{
    Lay l;
    l.blk = (unsigned)B; l.nblk = 1u; l.c = (unsigned)c; l.cb = 0u; l.cr = (unsigned)c; l.bp = (unsigned)B;
    return l;
}
EPI_DEV Lay make_lay_classic(const KArgs &a, int c) { return lay_classic(a.B, c); }
EPI_DEV size_t lay_scalar(int t, const Lay &l) { return (size_t)t * l.bp + l.c; }   // index into a [T][nblk*blk] array

template <int M>
EPI_DEV void store_vec(double *__restrict__ dst, int t, const Lay &l, const double (&v)[M])
{
    if (!dst) return;
    unsigned voff, rowb;
    const rsrc_t r = lay_slice(dst, t, M, l, voff, rowb);
#pragma unroll
    for (int i = 0; i < M; i++) bst(r, voff, (unsigned)i * rowb, v[i]);
}
template <int M>
EPI_DEV void store_mat(double *__restrict__ dst, int t, const Lay &l, const double (&P)[M * M])
{
    if (!dst) return;
    unsigned voff, rowb;
    const rsrc_t r = lay_slice(dst, t, M * M, l, voff, rowb);
#pragma unroll
    for (int e = 0; e < M * M; e++) bst(r, voff, (unsigned)e * rowb, P[e]);
}
template <int M>
EPI_DEV void load_vec(const double *__restrict__ src, int t, const Lay &l, double (&v)[M])
{
    unsigned voff, rowb;
    const rsrc_t r = lay_slice(src, t, M, l, voff, rowb);
#pragma unroll
    for (int i = 0; i < M; i++) v[i] = bld_s(r, voff, (unsigned)i * rowb);
}
template <int M>
EPI_DEV void load_mat(const double *__restrict__ src, int t, const Lay &l, double (&P)[M * M])
{
    unsigned voff, rowb;
    const rsrc_t r = lay_slice(src, t, M * M, l, voff, rowb);
#pragma unroll
    for (int e = 0; e < M * M; e++) P[e] = bld_s(r, voff, (unsigned)e * rowb);
}
// full symmetric matrix from a packed upper-triangle array (M(M+1)/2 rows)
template <int M>
EPI_DEV void load_packed(const double *__restrict__ src, int t, const Lay &l, double (&P)[M * M])
{
    constexpr int NSX = M * (M + 1) / 2;
    unsigned voff, rowb;
    const rsrc_t r = lay_slice(src, t, NSX, l, voff, rowb);
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = 0; i <= j; i++) {
            const double v = bld_s(r, voff, (unsigned)(i + j * (j + 1) / 2) * rowb);
            P[IXM(i, j)] = v;
            P[IXM(j, i)] = v;
        }
}
// the control series stay [T][n_npi][Su] (inputs, shared by the chains of a region)
EPI_DEV void load_u(const KArgs &a, int t, int su, double (&u)[kNpi])
{
    const unsigned voff = (unsigned)su * 8u, rowb = (unsigned)a.Su * 8u;
    const rsrc_t r = mk_rsrc(a.u + (size_t)t * a.n_npi * a.Su, (unsigned)a.n_npi * rowb);
    // rows k >= n_npi lie beyond the descriptor's range: the hardware bounds check returns 0.0 for them, which is the
    // padding value -- twelve unconditional loads instead of twelve scalar branches (the check includes the SGPR row
    // offset on gfx950: profiles/bounds_probe/probe.hip)
#pragma unroll
    for (int k = 0; k < kNpi; k++) u[k] = bld(r, voff, (unsigned)k * rowb);
}
EPI_DEV void store_u(double *__restrict__ dst, const KArgs &a, int t, const Lay &l, const double (&u)[kNpi])
{
    if (!dst) return;
    unsigned voff, rowb;
    const rsrc_t r = lay_slice(dst, t, (unsigned)a.n_npi, l, voff, rowb);
    if (a.n_npi == kNpi) {                    // the usual case: one scalar branch instead of twelve
#pragma unroll
        for (int k = 0; k < kNpi; k++) bst(r, voff, (unsigned)k * rowb, u[k]);
        return;
    }
#pragma unroll
    for (int k = 0; k < kNpi; k++)
        if (k < a.n_npi) bst(r, voff, (unsigned)k * rowb, u[k]);
}

// fp32-storage twins of the stores above (epi_batch_desc.storage = 1): same layout formulas with 4-byte elements, the
// value rounded once from the fp64 register
EPI_DEV rsrc_t lay_slice_f32(const float *p, int t, unsigned rows, const Lay &l, unsigned &voff, unsigned &rowb)
{
    rowb = l.blk * 4u;
    voff = (l.cb * rows * l.blk + l.cr) * 4u;
    return mk_rsrc(p + (size_t)t * rows * l.bp, rows * l.bp * 4u);
}
EPI_DEV void bst32(rsrc_t r, unsigned voff, unsigned soff, double v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v), r, voff, soff, EPI_ST32_AUX);
}
template <int N>
EPI_DEV void store_rows_f32(float *__restrict__ dst, int t, unsigned rows, const Lay &l, const double (&v)[N])
{
    if (!dst) return;
    unsigned voff, rowb;
    const rsrc_t r = lay_slice_f32(dst, t, rows, l, voff, rowb);
#pragma unroll
    for (int i = 0; i < N; i++)
        if ((unsigned)i < rows) bst32(r, voff, (unsigned)i * rowb, v[i]);
}
EPI_DEV void store_scalar_f32(float *__restrict__ dst, int t, const Lay &l, double v)
{
    if (dst) dst[lay_scalar(t, l)] = (float)v;
}

// ---------------------------------------------------------------------------
// forward pass
// ---------------------------------------------------------------------------
template <int M, int FLIP, int GENERIC>
__global__ __launch_bounds__(kWave) void ekf_fwd(const KArgs a)
{
    extern __shared__ double lds[];   // three sliding windows [3][L][64], one column per lane
    if (a.dense_flag && !*a.dense_flag) return;   // the symmetric fast path (ekf_fwd_sym) handles this batch
    const int lane = threadIdx.x;
    const int c = a.c0 + blockIdx.x * a.lw + lane;
    if (lane >= a.lw || c >= a.c0 + a.cn) return;
    if (a.only && !a.only[c]) return;             // second pass over the chains whose covariance went non-finite
    const int B = a.B, T = a.T, L = a.L;
    const int sx = a.x_series ? a.x_series[c] : c;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);

    ChainPrm p;
    load_prm<M>(p, a.prm, B, c, a.mf.lo_is_zero);
    const double v_bar = a.prm[(size_t)EPI_PRM_V_BAR * B + c];
    const double beta = a.prm[(size_t)EPI_PRM_BETA_EKF * B + c];
    const double gamma = a.prm[(size_t)EPI_PRM_GAMMA_EKF * B + c];

    double sk_minus[M], Pk_minus[M * M], Q[M * M];
#pragma unroll
    for (int i = 0; i < M; i++) sk_minus[i] = a.s_init[(size_t)i * B + c];
#pragma unroll
    for (int e = 0; e < M * M; e++) {
        Pk_minus[e] = a.Ps_init[(size_t)e * B + c];
        Q[e] = a.Q[(size_t)e * B + c];
    }
    double *winMean = lds + lane, *winCov = lds + (size_t)L * kWave + lane, *winCovN = lds + (size_t)2 * L * kWave + lane;
    for (int j = 0; j < L; j++) { winMean[j * kWave] = 0.0; winCov[j * kWave] = 0.0; winCovN[j * kWave] = 0.0; }
    int head = 0;

    const bool fixed_R = (a.r_mode == 0);            // GenericEKF.m:79-85
    const double R_v = fixed_R ? a.R_scalar[c] : 0.0;
    double R_next = R_v;                              // R(:,:,k+1) as left by step k (GENERIC) / running R (NewCase)

    for (int k = 0; k < T; k++) {
        const int t = tpos<FLIP>(k, T);
        // R_v is NOT time-flipped by the backward wrappers (Backward*.m:27 passes it through)
        const double Rk = fixed_R ? R_next : a.R_series[(size_t)k * a.Sx + sx];
        const double xk = a.x[(size_t)t * a.Sx + sx];
        double u_in[kNpi];
        load_u(a, t, su, u_in);
        if (GENERIC && a.q_mode)   // Q(:,:,k): like R_v, Q_w is not time-flipped by the backward wrappers
            load_mat<M>(a.Q, k, make_lay_classic(a, c), Q);

        store_vec<M>(a.S_MINUS, t, lay, sk_minus);       // :100-101
        store_mat<M>(a.P_MINUS, t, lay, Pk_minus);

        double C[M];
        obs_jacobian<M>(a.mf, sk_minus, C);               // :115
        const double xk_minus = predict_obs<M>(a.mf, sk_minus, v_bar);  // :116-119

        double innov, K[M], sk_plus[M], Pk_plus[M * M];
        const bool valid = !is_nan(xk);                   // :122
        if (valid) {
            innov = xk - xk_minus;
            double PCt[M], CP[M];
#pragma unroll
            for (int i = 0; i < M; i++) {
                double acc = Pk_minus[IXM(i, 0)] * C[0];
#pragma unroll
                for (int j = 1; j < M; j++) acc = fma(Pk_minus[IXM(i, j)], C[j], acc);
                PCt[i] = acc;
            }
#pragma unroll
            for (int j = 0; j < M; j++) {
                double acc = C[0] * Pk_minus[IXM(0, j)];
#pragma unroll
                for (int i = 1; i < M; i++) acc = fma(C[i], Pk_minus[IXM(i, j)], acc);
                CP[j] = acc;
            }
            double CPCt = CP[0] * C[0];
#pragma unroll
            for (int j = 1; j < M; j++) CPCt = fma(CP[j], C[j], CPCt);
            const double den = CPCt + gamma * Rk;          // :124 (D = 1, Hessian terms 0)
#pragma unroll
            for (int i = 0; i < M; i++) K[i] = PCt[i] / den;
            double IKC[M * M], T1[M * M];
#pragma unroll
            for (int j = 0; j < M; j++)
#pragma unroll
                for (int i = 0; i < M; i++) IKC[IXM(i, j)] = ((i == j) ? 1.0 : 0.0) - K[i] * C[j];
            mat_mul<M>(IKC, Pk_minus, T1);
            if (GENERIC) {
                double T2[M * M];
                mat_mul_bt<M>(T1, IKC, T2);                // Joseph form :127
#pragma unroll
                for (int j = 0; j < M; j++)
#pragma unroll
                    for (int i = 0; i < M; i++)
                        Pk_plus[IXM(i, j)] = (T2[IXM(i, j)] + (K[i] * Rk) * K[j]) / gamma;
            } else {
#pragma unroll
                for (int e = 0; e < M * M; e++) Pk_plus[e] = T1[e] / gamma;   // NewCase...m:64
            }
#pragma unroll
            for (int i = 0; i < M; i++) sk_plus[i] = sk_minus[i] + K[i] * innov;   // :129
        } else {                                           // :130-135
            innov = 0.0;
#pragma unroll
            for (int i = 0; i < M; i++) { K[i] = 0.0; sk_plus[i] = sk_minus[i]; }
#pragma unroll
            for (int e = 0; e < M * M; e++) Pk_plus[e] = Pk_minus[e];
        }
        if (GENERIC) symmetrize<M>(Pk_plus);               // :138
        state_hard_margins<M>(p, sk_plus);                 // :141

        // s(k+1|k), P(k+1|k)  :155-164
        double u_app[kNpi];
#pragma unroll
        for (int q = 0; q < kNpi; q++) u_app[q] = u_in[q];
        nlin_state_update<M, FLIP>(p, a.mf, u_app, sk_plus, sk_minus);
        store_u(a.u_opt, a, t, lay, u_app);
        {
            double A[M * M], T1[M * M], T2[M * M];
            state_jacobians<M, FLIP>(p, u_in, sk_plus, A);
            mat_mul<M>(A, Pk_plus, T1);
            mat_mul_bt<M>(T1, A, T2);
#pragma unroll
            for (int e = 0; e < M * M; e++) Pk_minus[e] = T2[e] + Q[e];   // B = I
        }
        if (GENERIC) symmetrize<M>(Pk_minus);              // :161
        state_hard_margins<M>(p, sk_minus);                // :164

        store_vec<M>(a.S_PLUS, t, lay, sk_plus);          // :167-169
        store_mat<M>(a.P_PLUS, t, lay, Pk_plus);
        store_vec<M>(a.K_GAIN, t, lay, K);
        if (a.innovations) a.innovations[lay_scalar(t, lay)] = innov;

        // innovation monitor :172-185 -- windows are newest-first and summed front to back
        const int cnt = (k + 1 < L) ? (k + 1) : L;
        head = (head == 0) ? (L - 1) : (head - 1);
        winMean[head * kWave] = innov;
        const double sum = ring_sum(winMean, head, L, innov);
        const double mu = sum / (double)cnt;
        const double cc = (innov - mu) * (innov - mu);
        const double ccn = GENERIC ? cc / (Rk + kEps) : cc / Rk;
        winCov[head * kWave] = cc;
        winCovN[head * kWave] = ccn;
        const double sumN = ring_sum(winCovN, head, L, ccn);
        // rho keeps FILTER-step order also for the time-flipped wrappers: GenericEKF.m:233 squeezes it to T x 1 and
        // Backward*.m:40 reverses a third dimension of size 1, i.e. nothing
        if (a.rho) a.rho[lay_scalar(k, lay)] = sumN / (double)cnt;
        if (fixed_R) {
            const bool adapt = GENERIC ? (beta != 1.0 && valid && k < T - 1) : (beta != 1.0 && valid);
            if (adapt) {
                const double sumC = ring_sum(winCov, head, L, cc);
                if (GENERIC) R_next = beta * Rk + (1.0 - beta) * (sumC / (double)cnt);   // :184
                else R_next = beta * Rk + (1.0 - beta) * sumC / (double)cnt;             // NewCase...m:111
            } else if (GENERIC) {
                R_next = R_v;   // R(k+1) keeps its initial value when step k does not write it
            }
        }
    }
}

// ---------------------------------------------------------------------------
// smoother gain, part 1: X = pinv(P_MINUS(:,:,k+1)) for every (chain, step) pair
// ---------------------------------------------------------------------------
// GenericEKF.m:209-215.  pinv of P(k+1|k) depends only on forward quantities, not on the backward
// recursion, so it is hoisted out of the sequential smoother loop: one lane per (chain, step) pair,
// (T-1)*B independent items.  This is where ~80 % of the path's flops are (cyclic Jacobi on a 6 x 6),
// and as a flat grid it load-balances over all 1024 SIMDs instead of B/64 long-lived waves.
// 6 x 6: one wavefront per workgroup.  The Jacobi iteration count is data dependent, and with 256-thread workgroups a
// SIMD's wave slot stays empty until all four waves of a workgroup have finished (measured 1.67 resident waves per SIMD
// instead of 1.9; 7.25 -> 6.6 ms on the headline sweep, 128 threads: 6.9 ms).  3 x 3: the waves are short and nearly
// uniform, and four times as many workgroups cost more than the slots they free (5.2 -> 5.9 ms on the 307 200-chain
// ensemble), so they keep 256 threads.
template <int M>
constexpr int pinv_wg() { return M >= 6 ? 64 : 256; }
// Three waves per SIMD (168 VGPRs): the third wave fills VALU issue slots two dependent fp64 chains leave idle (6.2 ->
// 5.9 ms on the headline sweep with 76 B/lane of scratch; 5.8 ms and no scratch once the Jacobi's b/z accumulators
// moved to LDS, 6 KB per wavefront).
// X = pinv(P(t1|t1-1)) of chain c: the work of one lane of the grid below
template <int M>
EPI_DEV void pinv_one(const KArgs &a, int c, int t1, double *plds)
{
    constexpr int WG = pinv_wg<M>();
    const Lay lay = make_lay(a, c);
    constexpr int NSX = M * (M + 1) / 2;
    double Pu[NSX];
    // P_MINUS is stored symmetrised (:161): only its upper triangle is read
    unsigned voff_p, rowb_p;
    const rsrc_t rp = lay_slice(a.P_MINUS, t1, M * M, lay, voff_p, rowb_p);
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = 0; i <= j; i++) Pu[i + j * (j + 1) / 2] = bld_s(rp, voff_p, (unsigned)IXM(i, j) * rowb_p);
    bool bad = false;                                      // :211
#pragma unroll
    for (int i = 0; i < NSX; i++) bad = bad || is_nonfinite(Pu[i]);
    int32_t *rword = a.rankbuf + lay_scalar(t1, lay);
    if (bad) {
        *rword = -1;
        return;
    }
    // X is symmetric bit for bit: the workspace holds its packed upper triangle (M(M+1)/2 rows)
    double Xu[NSX];
    bool capped, indef;
    // one LDS column per lane: scratch of the full-rank route, and later the b/z accumulators of the two-sided fall-back
    int rank = sym_pinv_psd<M, WG>(Pu, Xu, &capped, &indef, plds + threadIdx.x);              // :215
    unsigned voff_x, rowb_x;
    const rsrc_t rx = lay_slice(a.X, t1, NSX, lay, voff_x, rowb_x);
    if (a.hex) {       // 80-byte rows: see bst_nt (eks_pinv of the 9 375-chain shard 0.84 ms written through, 0.49 ms not)
#pragma unroll
        for (int i = 0; i < NSX; i++) bst_nt(rx, voff_x, (unsigned)i * rowb_x, Xu[i]);
    } else {
#pragma unroll
        for (int i = 0; i < NSX; i++) bst(rx, voff_x, (unsigned)i * rowb_x, Xu[i]);
    }
    *rword = rank | (capped ? 0x100 : 0);
    if (__builtin_amdgcn_ballot_w64(indef) != 0ull) {
        // Not positive semi-definite up to rounding (never the case for a covariance the filter produced from a positive
        // semi-definite Ps_init): the two-sided Jacobi route on the matrix read again; the lanes concerned overwrite what
        // they stored above (nothing of the first route is live across this block).
        constexpr int BZS = (M >= 6) ? WG : 0;              // its b/z accumulators (one column per lane), see jacobi_eig
        double *bzs = plds;
        double P[M * M], X[M * M];
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) {
                const double v = bld(rp, voff_p, (unsigned)IXM(i, j) * rowb_p);
                P[IXM(i, j)] = v;
                P[IXM(j, i)] = v;
            }
        bool capped2;
        const int rank2 = sym_pinv_two_sided<M, BZS>(P, X, &capped2, bzs + threadIdx.x);
        if (indef) {
#pragma unroll
            for (int j = 0; j < M; j++)
#pragma unroll
                for (int i = 0; i <= j; i++) bst(rx, voff_x, (unsigned)(i + j * (j + 1) / 2) * rowb_x, X[IXM(i, j)]);
            *rword = rank2 | (capped2 ? 0x100 : 0);
        }
    }
}
template <int M, int LIST = 0>
__global__ __launch_bounds__(pinv_wg<M>(), 3) void eks_pinv(const KArgs a)
{
    // grid: x = pinv_wg<M>()-chain tiles of the chain range (rounded up to a multiple of 8), y = step; a workgroup shares one
    // step => uniform row bases
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2) in linear-id order, so
    // XCD c gets ids c, c + 8, ...  Within EVERY step give each XCD a contiguous eighth of the tiles: neighbouring tiles --
    // whose 64 chains share a partial cache line where they straddle a 40-chain layout block -- then meet in one L2, and all
    // XCDs still walk the steps together (the early, full-rank steps cost three times the late ones).
    const unsigned gx = gridDim.x;
    const unsigned long long lin = (unsigned long long)blockIdx.y * gx + blockIdx.x;      // (2^17 tiles x 2^16 steps exceed 32 bits)
    const unsigned per = gx >> 3;                              // tiles of a step per XCD (the grid's x extent is a multiple of 8)
    const unsigned xcd = (unsigned)(lin & 7ull);
    const unsigned long long k = lin >> 3;
    unsigned by = blockIdx.y, bx = blockIdx.x;                 // fewer than 8 tiles: the launch keeps the plain order
    if ((gx & 7u) == 0u) {
        by = (unsigned)(k / per);
        bx = xcd * per + (unsigned)(k - (unsigned long long)by * per);
    }
    constexpr int WG = pinv_wg<M>();
    constexpr int NSX = M * (M + 1) / 2;
    constexpr int LROWS = NSX > 2 * M ? NSX : 2 * M;
    // one LDS column per lane: scratch of the full-rank route, and later the b/z accumulators of the two-sided fall-back
    __shared__ double plds[LROWS * WG];
    // array positions of filter steps 2..T: 1..T-1, or 0..T-2 for the time-flipped models (pinv_pos0 = 0)
    const int t1 = a.pinv_pos0 + a.pinv_step0 + (int)by;
    if (LIST) {
        // Second pass over the chains whose covariance went non-finite (epi_batch_desc.exact_nonfinite, a.only): the launch is a
        // few tiles wide and walks the list mark_nonfinite left -- nothing to do, and next to nothing dispatched, when no chain
        // is marked.  (A kernel of its own: the loop around the 60 KB body costs the main grid 16 B of scratch per lane.)
        const int32_t *list = a.only + a.B + 1;
        const int count = a.only[a.B];
        for (int cl = (int)(bx * blockDim.x + threadIdx.x); cl < count; cl += (int)(gx * blockDim.x)) pinv_one<M>(a, list[cl], t1, plds);
        return;
    }
    const int cl = (int)(bx * blockDim.x + threadIdx.x);
    if (cl >= a.cn) return;
    pinv_one<M>(a, a.c0 + cl, t1, plds);
}

// ---------------------------------------------------------------------------
// backward pass (fixed-interval smoother)
// ---------------------------------------------------------------------------
template <int M, int FLIP, int GENERIC>
__global__ __launch_bounds__(kWave) void eks_bwd(const KArgs a)
{
    if (a.dense_flag && !*a.dense_flag) return;   // eks_bwd_sym handles this batch
    const int c = a.c0 + blockIdx.x * a.lw + threadIdx.x;
    if ((int)threadIdx.x >= a.lw || c >= a.c0 + a.cn) return;
    if (a.only && !a.only[c]) return;
    const int B = a.B, T = a.T;
    const int su = a.u_series ? a.u_series[c] : c;
    const Lay lay = make_lay(a, c);
    // the four 12-vectors of `params` in an LDS column per lane instead of 96 VGPRs (without this the 6-state kernel
    // spills: 512 registers + 84 B of scratch)
    // (three states: a and u_max only, VecLds2 -- 12 KB instead of 24 KB per workgroup, so that LDS does not cap the kernel
    // below the two waves per SIMD its registers allow)
    constexpr int NV = (M == 6) ? 4 : 2;
    __shared__ double vlds[NV * kNpi * kWave];
    LitePrm<typename std::conditional<M == 6, VecLds, VecLds2>::type> p;
    load_lite(p, a.prm, B, c, a.mf.lo_is_zero);
    p.v.base = vlds + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kNpi; k++) {
        vlds[(0 * kNpi + k) * kWave + threadIdx.x] = a.prm[(size_t)(EPI_PRM_A + k) * B + c];
        if constexpr (M == 6) {
            vlds[(1 * kNpi + k) * kWave + threadIdx.x] = a.prm[(size_t)(EPI_PRM_U_MIN + k) * B + c];
            vlds[(2 * kNpi + k) * kWave + threadIdx.x] = a.prm[(size_t)(EPI_PRM_U_MAX + k) * B + c];
            vlds[(3 * kNpi + k) * kWave + threadIdx.x] = a.prm[(size_t)(EPI_PRM_W_EFF + k) * B + c];
        } else {
            vlds[(1 * kNpi + k) * kWave + threadIdx.x] = a.prm[(size_t)(EPI_PRM_U_MAX + k) * B + c];
        }
    }

    // terminal conditions :189-202
    double Ss[M], Ps[M * M];
    const int tT = tpos<FLIP>(T - 1, T);
    load_vec<M>(a.S_PLUS, tT, lay, Ss);
    load_mat<M>(a.P_PLUS, tT, lay, Ps);
#pragma unroll
    for (int i = 0; i < M; i++) {
        double f = a.s_final[(size_t)i * B + c];
        if (!is_nan(f)) Ss[i] = f;
    }
    {
        double Pf[M * M];
#pragma unroll
        for (int e = 0; e < M * M; e++) Pf[e] = a.Ps_final[(size_t)e * B + c];
        if (GENERIC) {
#pragma unroll
            for (int e = 0; e < M * M; e++)
                if (!is_nan(Pf[e])) Ps[e] = Pf[e];
        } else {
            // P_SMOOTH(row, col, T) = Ps_final(row, col): cross-product sub-assignment, NewCase...m:125-127
            bool rows[M], cols[M];
#pragma unroll
            for (int i = 0; i < M; i++) { rows[i] = false; cols[i] = false; }
#pragma unroll
            for (int j = 0; j < M; j++)
#pragma unroll
                for (int i = 0; i < M; i++)
                    if (!is_nan(Pf[IXM(i, j)])) { rows[i] = true; cols[j] = true; }
#pragma unroll
            for (int j = 0; j < M; j++)
#pragma unroll
                for (int i = 0; i < M; i++)
                    if (rows[i] && cols[j]) Ps[IXM(i, j)] = Pf[IXM(i, j)];
        }
    }
    store_vec<M>(a.S_SMOOTH, tT, lay, Ss);
    store_mat<M>(a.P_SMOOTH, tT, lay, Ps);
    if (GENERIC && a.u_opt_smooth) {
        double z[kNpi];
#pragma unroll
        for (int k = 0; k < kNpi; k++) z[k] = 0.0;
        store_u(a.u_opt_smooth, a, tT, lay, z);      // column T is never written :95,204
    }
    if (a.pinv_rank) a.pinv_rank[lay_scalar(tT, lay)] = -1;

    int st_guard = 0, st_cap = 0, min_rank = M;
    for (int k = T - 2; k >= 0; k--) {
        const int t = tpos<FLIP>(k, T), t1 = tpos<FLIP>(k + 1, T);
        double Sp[M], Pp[M * M], Sm1[M], Pm1[M * M], u_in[kNpi];
        load_vec<M>(a.S_PLUS, t, lay, Sp);
        load_mat<M>(a.P_PLUS, t, lay, Pp);
        load_vec<M>(a.S_MINUS, t1, lay, Sm1);
        load_mat<M>(a.P_MINUS, t1, lay, Pm1);
        load_u(a, t, su, u_in);

        double J[M * M];
        int rank = -1;
        {
            double A[M * M], PAt[M * M];
            state_jacobians<M, FLIP>(p, u_in, Sp, A);          // :206
            mat_mul_bt<M>(Pp, A, PAt);                         // P_PLUS * A'
            if (GENERIC) {
                // pinv(P_MINUS(:,:,k+1)) was computed by eks_pinv (one lane per (chain, step) pair)
                const int rk = a.rankbuf[lay_scalar(t1, lay)];
                if (rk < 0) {                                  // non-finite P_MINUS guard :211-213
#pragma unroll
                    for (int e = 0; e < M * M; e++) J[e] = 0.0;
                    st_guard = 1;
                } else {
                    double X[M * M];
                    load_packed<M>(a.X, t1, lay, X);
                    mat_mul<M>(PAt, X, J);                     // :215
                    rank = rk & 0xff;
                    st_cap |= (rk >> 8) & 1;
                    min_rank = rank < min_rank ? rank : min_rank;
                }
            } else {
                mrdivide<M>(PAt, Pm1, J);                      // NewCase...m:132
            }
        }
        if (a.pinv_rank) a.pinv_rank[lay_scalar(t, lay)] = rank;

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
                Sn[i] = Sp[i] + acc;                           // :218
            }
        }
        state_hard_margins<M>(p, Sn);                          // :221
        {
            double D[M * M], T1[M * M], T2[M * M];
#pragma unroll
            for (int e = 0; e < M * M; e++) D[e] = Pm1[e] - Ps[e];
            mat_mul<M>(J, D, T1);
            mat_mul_bt<M>(T1, J, T2);
#pragma unroll
            for (int e = 0; e < M * M; e++) Ps[e] = Pp[e] - T2[e];   // :223
        }
        if (GENERIC) symmetrize<M>(Ps);                        // :226
#pragma unroll
        for (int i = 0; i < M; i++) Ss[i] = Sn[i];
        store_vec<M>(a.S_SMOOTH, t, lay, Ss);
        store_mat<M>(a.P_SMOOTH, t, lay, Ps);
        if (GENERIC && a.u_opt_smooth) {                       // :229
            double sn_unused[M];
            nlin_state_update<M, FLIP>(p, a.mf, u_in, Ss, sn_unused);
            store_u(a.u_opt_smooth, a, t, lay, u_in);
        }
    }
    if (a.status) a.status[c] = st_guard | (st_cap << 1) | (min_rank << 8);
}

#include "ekf_sym.hpp"
#include "ekf_quad.hpp"
#include "ekf_wave.hpp"
#include "ekf_hex.hpp"
#include "ekf_lane6.hpp"

// ---------------------------------------------------------------------------
// forward simulators
// ---------------------------------------------------------------------------
// Tools/SIalpha_Controlled.m:24-28 (+ NPICost.m:6-10 fused when J0/J1 are requested)
struct SimPrm {
    double alpha_min, alpha_max, gamma, b, beta, dt, s_std, i_std, a_std;
    double ga[kNpi], um[kNpi], w[kNpi];   // gamma*a(k), u_max(k), NPICost weights(k)
};
// column `c` of an [EPI_SIM_PRM_COUNT][n] parameter block
EPI_DEV void load_sim_prm(SimPrm &p, const double *__restrict__ sp, int n, int c, double &s, double &i, double &al)
{
    auto g = [&](int f) { return sp[(size_t)f * n + c]; };
    s = g(EPI_SIM_S0); i = g(EPI_SIM_I0); al = g(EPI_SIM_ALPHA0);
    p.alpha_min = g(EPI_SIM_ALPHA_MIN); p.alpha_max = g(EPI_SIM_ALPHA_MAX); p.gamma = g(EPI_SIM_GAMMA);
    p.b = g(EPI_SIM_B); p.beta = g(EPI_SIM_BETA); p.dt = g(EPI_SIM_DT);
    p.s_std = g(EPI_SIM_S_STD); p.i_std = g(EPI_SIM_I_STD); p.a_std = g(EPI_SIM_ALPHA_STD);
#pragma unroll
    for (int k = 0; k < kNpi; k++) { p.ga[k] = p.gamma * g(EPI_SIM_A + k); p.um[k] = g(EPI_SIM_U_MAX + k); p.w[k] = g(EPI_SIM_W + k); }
}
// one day of SIalpha_Controlled.m:25-27 (entries of uk beyond n_npi are 0 and a(k) there is 0)
EPI_DEV void sialpha_step(const SimPrm &p, const double (&uk)[kNpi], double z1, double z2, double z3, double &s, double &i,
                          double &al)
{
    double dot = p.ga[0] * (p.um[0] - uk[0]);
#pragma unroll
    for (int k = 1; k < kNpi; k++) dot = fma(p.ga[k], p.um[k] - uk[k], dot);
    const double sn = fmax(0.0, fmin(1.0, s - p.dt * (al * s * i + z1 * p.s_std)));
    const double in = fmax(0.0, fmin(1.0, i + p.dt * (al * s * i - p.beta * i + z2 * p.i_std)));
    const double an = fmax(p.alpha_min, fmin(p.alpha_max, al + p.dt * (-p.gamma * al + p.gamma * p.b + dot + z3 * p.a_std)));
    s = sn; i = in; al = an;
}
// running sums of NPICost.m:6-10 over one more day: mean(newcases) and mean(weights(:).*inputs(:)) in column-major
// order (NPI index fastest); `first` = this is the very first day of the span (sum starts with the term itself)
EPI_DEV void npicost_accumulate(const SimPrm &p, const double (&uk)[kNpi], int n_npi, bool first, double s, double i, double al,
                                double &acc0, double &acc1)
{
    const double nc = s * i * al;
    acc0 = first ? nc : acc0 + nc;
#pragma unroll
    for (int k = 0; k < kNpi; k++)
        if (k < n_npi) {
            const double term = p.w[k] * uk[k];
            acc1 = (first && k == 0) ? term : acc1 + term;
        }
}

__global__ __launch_bounds__(256) void sialpha_sim(const epi_sim_desc d, const int32_t *__restrict__ u_series,
                                                   const double *__restrict__ u, const double *__restrict__ sp,
                                                   const double *__restrict__ z, double *__restrict__ so,
                                                   double *__restrict__ io, double *__restrict__ ao,
                                                   double *__restrict__ J0, double *__restrict__ J1,
                                                   const double *__restrict__ J0_prefix,
                                                   const double *__restrict__ J1_prefix,
                                                   const int32_t *__restrict__ gate)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.B) return;
    if (gate && *gate == 0) return;       // second scoring pass after the dense re-run: nothing was re-run
    const int B = d.B;
    const int su = u_series ? u_series[c] : c;
    SimPrm p;
    double s, i, al;
    load_sim_prm(p, sp, B, c, s, i, al);
    // NPICost over [historic days, simulated days] (TrainPredictPrescribeNPI.m:481-493): the historic part of
    // the two sequential sums arrives as a per-chain prefix and the simulated days are added in order
    const bool pre = (d.prefix_days > 0) && J0_prefix && J1_prefix;
    double acc0 = pre ? J0_prefix[c] : 0.0, acc1 = pre ? J1_prefix[c] : 0.0;
    // u: [K][n_npi][Su], or chain-blocked ((t*nblk + su/blk)*n_npi + k)*blk + su%blk
    const int ublk = (d.u_block <= 0 || d.u_block >= d.Su) ? d.Su : d.u_block;
    const size_t unb = (size_t)(d.Su + ublk - 1) / ublk, ucb = (size_t)su / ublk, ucr = (size_t)su % ublk;
    for (int t = 0; t < d.K; t++) {
        double uk[kNpi];
#pragma unroll
        for (int k = 0; k < kNpi; k++) uk[k] = (k < d.n_npi) ? u[(((size_t)t * unb + ucb) * d.n_npi + k) * ublk + ucr] : 0.0;
        double z1 = 0.0, z2 = 0.0, z3 = 0.0;
        if (d.noise) {
            z1 = z[((size_t)t * 3 + 0) * B + c]; z2 = z[((size_t)t * 3 + 1) * B + c]; z3 = z[((size_t)t * 3 + 2) * B + c];
        }
        sialpha_step(p, uk, z1, z2, z3, s, i, al);
        if (so) so[(size_t)t * B + c] = s;
        if (io) io[(size_t)t * B + c] = i;
        if (ao) ao[(size_t)t * B + c] = al;
        if (d.with_cost) npicost_accumulate(p, uk, d.n_npi, t == 0 && !pre, s, i, al, acc0, acc1);
    }
    if (d.with_cost) {
        const size_t days = (size_t)d.K + (size_t)(pre ? d.prefix_days : 0);
        if (J0) J0[c] = acc0 / (double)days;
        if (J1) J1[c] = acc1 / (double)((size_t)d.n_npi * days);
    }
}

#include "scenario_kernels.hpp"
#include "rt_expfit.hpp"
#include "preprocess.hpp"
#include "nnls.hpp"

struct SeirpRates { double ae, ai, kappa, rho, beta, mu, gamma; };
EPI_DEV void seirp_rhs(const SeirpRates &r, const double (&y)[5], double (&f)[5])
{
    // SEIRP.m:27-31
    f[0] = -r.ae * y[0] * y[1] - r.ai * y[0] * y[2] + r.gamma * y[3];
    f[1] = r.ae * y[0] * y[1] + r.ai * y[0] * y[2] - r.kappa * y[1] - r.rho * y[1];
    f[2] = r.kappa * y[1] - r.beta * y[2] - r.mu * y[2];
    f[3] = r.beta * y[2] + r.rho * y[1] - r.gamma * y[3];
    f[4] = r.mu * y[2];
}
__global__ __launch_bounds__(256) void seirp_sim(int B, int K, int par_steps, double dt, int saturated, int integrator,
                                                 const double *__restrict__ par, const double *__restrict__ init,
                                                 const double *__restrict__ sat, double *__restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    double y[5];
#pragma unroll
    for (int q = 0; q < 5; q++) { y[q] = init[(size_t)q * B + c]; out[(size_t)q * B + c] = y[q]; }
    double b0 = 0, bs = 0, m0 = 0, ms = 0, sg = 1, i_0 = 0;
    if (saturated) {
        b0 = sat[(size_t)0 * B + c]; bs = sat[(size_t)1 * B + c]; m0 = sat[(size_t)2 * B + c];
        ms = sat[(size_t)3 * B + c]; sg = sat[(size_t)4 * B + c]; i_0 = sat[(size_t)5 * B + c];
    }
    for (int t = 0; t < K - 1; t++) {
        const size_t pt = (par_steps == 1) ? 0 : (size_t)t;
        SeirpRates r;
        r.ae = par[(pt * 7 + 0) * B + c]; r.ai = par[(pt * 7 + 1) * B + c]; r.kappa = par[(pt * 7 + 2) * B + c];
        r.rho = par[(pt * 7 + 3) * B + c]; r.beta = par[(pt * 7 + 4) * B + c]; r.mu = par[(pt * 7 + 5) * B + c];
        r.gamma = par[(pt * 7 + 6) * B + c];
        if (saturated) {   // SEIRPSaturatedResource.m:27-29
            const double h = (epi_tanh((y[2] - i_0) / sg) + 1.0) / 2.0;
            r.beta = (bs - b0) * h + b0;
            r.mu = (ms - m0) * h + m0;
        }
        double yn[5];
        if (integrator == 0) {      // explicit Euler, the reference's integrator: (rhs)*dt + y
            double f[5];
            seirp_rhs(r, y, f);
#pragma unroll
            for (int q = 0; q < 5; q++) yn[q] = f[q] * dt + y[q];
        } else {                    // classical RK4 with the step's rates frozen (extension)
            double k1[5], k2[5], k3[5], k4[5], yt[5];
            seirp_rhs(r, y, k1);
#pragma unroll
            for (int q = 0; q < 5; q++) yt[q] = y[q] + 0.5 * dt * k1[q];
            seirp_rhs(r, yt, k2);
#pragma unroll
            for (int q = 0; q < 5; q++) yt[q] = y[q] + 0.5 * dt * k2[q];
            seirp_rhs(r, yt, k3);
#pragma unroll
            for (int q = 0; q < 5; q++) yt[q] = y[q] + dt * k3[q];
            seirp_rhs(r, yt, k4);
#pragma unroll
            for (int q = 0; q < 5; q++) yn[q] = y[q] + (dt / 6.0) * (k1[q] + 2.0 * k2[q] + 2.0 * k3[q] + k4[q]);
        }
#pragma unroll
        for (int q = 0; q < 5; q++) { y[q] = yn[q]; out[((size_t)(t + 1) * 5 + q) * B + c] = y[q]; }
    }
}

// Measurement utility: streams n doubles src -> dst with the SAME access shape as the filter kernels (one
// 8-byte element per lane, 512 B per wave instruction, grid-stride).  profiles/traffic_probe.py uses it to
// calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on a known byte count (MI355X_MICROARCH.md, HBM section).
__global__ __launch_bounds__(256) void calib_copy_f64(const double *__restrict__ src, double *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// epi_batch_desc.exact_nonfinite: only[c] = status bit 0 of chain c (the non-finite guard of GenericEKF.m:211 fired: its
// covariance overflowed at some day), only[B] = how many chains that is, only[B + 1 ...] = which (any order): the dense
// second pass's pinv grid walks that list instead of testing every (chain, step) pair
__global__ __launch_bounds__(256) void mark_nonfinite(const int32_t *__restrict__ status, int32_t *__restrict__ only, int B)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    const int v = status[c] & 1;
    only[c] = v;
    if (v) only[B + 1 + atomicAdd(only + B, 1)] = c;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static void set_err(char *err, const char *msg)
{
    if (err) { strncpy(err, msg, 255); err[255] = 0; }
}
static int hip_fail(char *err, hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "HIP error in %s: %s", what, hipGetErrorString(e));
    set_err(err, buf);
    return EPI_ERR_HIP;
}

struct WsLayout { size_t s_minus, s_plus, p_minus, p_plus, x, rank, flag, innov, hand_s, hand_p, hand_i, only, status, total; };
static int lane_block_of(const epi_batch_desc *d) { return (d->lane_block <= 0 || d->lane_block >= d->B) ? d->B : d->lane_block; }
static size_t padded_chains(const epi_batch_desc *d)
{
    const int blk = lane_block_of(d);
    return (size_t)((d->B + blk - 1) / blk) * blk;
}

// ---------------------------------------------------------------------------
// per-device state (the only state the library keeps besides the host entry points' context pool; both are freed by
// epi_host_pool_release): the SIMD count of each device, and idle helper streams
// ---------------------------------------------------------------------------
constexpr int kMaxDevices = 64;
constexpr int kHelperEvents = 16;
// A helper stream of the LOWEST priority (its workgroups are placed after the caller's stream's) with the events one call
// needs to fork work onto it and join it back.  A call leases one for the time it takes to ENQUEUE its kernels; work of
// consecutive lessees simply queues up on the stream.
struct Helper { hipStream_t stream = nullptr; hipEvent_t ev[kHelperEvents] = {}; };
static std::mutex g_dev_mu;
static std::atomic<int> g_simds[kMaxDevices];
static std::vector<Helper *> g_helpers[kMaxDevices];
// helpers that were forked into a caller's stream CAPTURE: such a stream stays in capture mode until the caller ends the
// capture, which the library cannot see, so it is never leased again (another thread's hipEventRecord on it would fail or
// invalidate that capture); kept here until epi_host_pool_release
static std::vector<Helper *> g_retired_helpers[kMaxDevices];

static int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return dev;
}
static int simd_count(int dev)
{
    const int v = g_simds[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        return 1024;                       // MI355X; not cached, so a later call asks the runtime again
    }
    g_simds[dev].store(cus * 4, std::memory_order_relaxed);
    return cus * 4;
}
static void helper_destroy(Helper *h)
{
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    for (auto &ev : h->ev)
        if (ev) (void)hipEventDestroy(ev);
    delete h;
}
static hipError_t helper_acquire(int dev, Helper **out)
{
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        if (!g_helpers[dev].empty()) { *out = g_helpers[dev].back(); g_helpers[dev].pop_back(); return hipSuccess; }
    }
    Helper *h = new Helper();
    int lo = 0, hi = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);      // lo = numerically largest = lowest priority
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, lo);
    for (auto &ev : h->ev)
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) { helper_destroy(h); return e; }
    *out = h;
    return hipSuccess;
}
struct HelperLease {       // returns the helper to its device's idle list when the enqueueing call ends
    int dev; Helper *h = nullptr;
    bool captured = false;  // the caller's stream is being captured: the helper joined that capture and is retired instead
    explicit HelperLease(int d) : dev(d) {}
    ~HelperLease()
    {
        if (!h) return;
        std::lock_guard<std::mutex> lk(g_dev_mu);
        (captured ? g_retired_helpers[dev] : g_helpers[dev]).push_back(h);
    }
};
static void helpers_release_all()
{
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    for (int dev = 0; dev < kMaxDevices; dev++) {
        std::vector<Helper *> mine;
        {
            std::lock_guard<std::mutex> lk(g_dev_mu);
            mine.swap(g_helpers[dev]);
            mine.insert(mine.end(), g_retired_helpers[dev].begin(), g_retired_helpers[dev].end());
            g_retired_helpers[dev].clear();
        }
        if (mine.empty()) continue;
        (void)hipSetDevice(dev);
        for (Helper *h : mine) helper_destroy(h);
    }
    if (have_prev) (void)hipSetDevice(prev);
}

// The innovation monitor leaves the forward kernels (ekf_monitor, ekf_quad.hpp) when R_v is a per-day series -- then rho
// feeds nothing back -- and two double-written L-sample windows per lane fit the default dynamic-LDS limit.
static bool monitor_hoisted(const epi_batch_desc *d)
{
    return MODEL_TABLE[d->model].generic && d->r_mode == 1 && d->q_mode == 0 && d->path_hint != 2 &&
           (size_t)4 * d->L * kWave * sizeof(double) <= 64u * 1024u;
}
// Which lane mapping runs the 6-state generic models (epi_batch_desc.shape).  Auto: four lanes per chain while every
// quad wavefront (16 chains) still gets a SIMD of its own, i.e. up to 16 384 chains on MI355X -- there the chains'
// per-day latency is what counts and the quad kernels' instruction stream is half as long (9 375 chains: 3.5 instead of
// 6.1 ms per pass); beyond that the quad waves (one per SIMD at ~290 registers) would run in rounds and one lane per
// chain, the shape with the least total work, wins (18 750 chains: 6.1 against 8.4 ms).  profiles/r02/batch_size_sweep.txt.
// Round 4: one WAVEFRONT per chain (ekf_wave.hpp) while every chain can have a SIMD of its own (B <= 1024 on MI355X): a lone
// wave's day costs ~0.6 us there against 2.1-2.2 us for a quad wave, at ~4 x the quad shape's total VALU work.  It needs the
// innovation monitor as its own kernel (R_v a per-day series) and a fixed Q_w; otherwise four lanes per chain.
static int shape_of(const epi_batch_desc *d, int dev)
{
    const ModelInfo &mi = MODEL_TABLE[d->model];
    if (!mi.generic) {
        // NewCaseEKFEstimatorWithOptimalNPI (round 5): one wavefront per chain (ekf_fwd_wave<0, 1, LC, 0>, eks_bwd_wave_nc) while every
        // chain can have a SIMD of its own -- the reference's own callers make ONE call per chain
        // (testScripts/testSIModelOptimalControl04EKS.m:168,302): 2.1 instead of 8.4 ms -- otherwise the dense one-lane kernels
        const bool ok = mi.m == 6 && !d->storage && d->q_mode == 0 && (size_t)6 * d->L * sizeof(double) <= 48u * 1024u;
        if (d->shape == EPI_SHAPE_WAVE) return ok ? EPI_SHAPE_WAVE : EPI_SHAPE_LANE;
        if (d->shape != EPI_SHAPE_AUTO) return EPI_SHAPE_LANE;
        return (ok && (long)d->B <= (long)simd_count(dev)) ? EPI_SHAPE_WAVE : EPI_SHAPE_LANE;
    }
    if (d->storage) return EPI_SHAPE_LANE;
    if (mi.m == 3) {
        // 3-state models: seven chains per wavefront, nine lanes each (ekf_fwd_wave3), for small batches; otherwise one lane
        // per chain.  There is no four-lane shape for them.  Measured (BASELINE config 3's chains x 400 days, ms per pass,
        // wave / lane): 300 chains 0.91 / 1.04, 2 000 1.09 / 1.20, 7 000 1.47 / 1.26, 20 000 3.08 / 1.80 -- a 3-state step is
        // bound by its scalar chain (state map, the 12-term NPI sum, two divisions: ~1.4 us per day in EITHER shape), the
        // 3 x 3 algebra the wave shape spreads over nine lanes is a small part of it.  Chosen up to 2 048 chains.
        const bool ok = monitor_hoisted(d);
        if (d->shape == EPI_SHAPE_WAVE) return ok ? EPI_SHAPE_WAVE : EPI_SHAPE_LANE;
        if (d->shape == EPI_SHAPE_LANE || d->shape == EPI_SHAPE_QUAD || d->shape == EPI_SHAPE_HEX) return EPI_SHAPE_LANE;
        return (ok && d->B <= 2048) ? EPI_SHAPE_WAVE : EPI_SHAPE_LANE;
    }
    // the hex shape needs the monitor as its own kernel (R_v a per-day series) and a fixed Q_w -- and at least a few days of
    // every array within the 2 GiB its addressing windows span (ekf_hex.hpp, hx_window): 288 B and 8 n_npi B per chain and day
    const bool hex_ok = monitor_hoisted(d) && (long)d->B <= (1L << 20) && (long)d->n_npi * d->Su <= (1L << 25);
    // the wave shape also runs the monitor inline (a scalar, possibly adaptive R_v: ekf_fwd_wave<FLIP, 1>, round 5)
    const bool wave_ok = hex_ok || (mi.generic && d->r_mode == 0 && d->q_mode == 0 && d->path_hint != 2 &&
                                    (size_t)6 * d->L * sizeof(double) <= 48u * 1024u);
    if (d->shape == EPI_SHAPE_WAVE) return wave_ok ? EPI_SHAPE_WAVE : EPI_SHAPE_QUAD;
    if (d->shape == EPI_SHAPE_HEX) return hex_ok ? EPI_SHAPE_HEX : EPI_SHAPE_QUAD;
    if (d->shape == EPI_SHAPE_QUAD || d->shape == EPI_SHAPE_LANE) return d->shape;
    // (round 5: with the hex shape there, one wavefront per chain wins only up to ~600 chains: 300 chains 1.59 against 1.81 ms per
    // call, 1 024 chains 2.15 against 1.84 -- profiles/r05/shape_latency.json; where the hex shape cannot run, up to one chain per SIMD)
    if (wave_ok && (long)d->B <= (hex_ok ? (long)simd_count(dev) * 5 / 8 : (long)simd_count(dev))) return EPI_SHAPE_WAVE;
    // round 5: six lanes per chain, ten chains per wavefront (ekf_hex.hpp) up to TWO such wavefronts per SIMD (20 480 chains on
    // MI355X; beyond one per SIMD its kernels run two waves per SIMD, a third does not fit their registers): 9 375 chains -- the
    // shard of the headline sweep on one of 8 GPUs -- 2.6 ms against 3.2 (quad) and 4.9 (lane); 18 750 chains (one of 4 GPUs)
    // 5.1 against 5.6 (lane), 20 480 chains 5.5 against 5.7, 20 500 chains 6.5 against 5.8: profiles/r05/ab_hex_threshold.txt
    // (with the kernels' day offsets in the scalar offset; before that the shapes were level at 18 750)
    if (hex_ok && ((long)d->B + kHG - 1) / kHG <= 2 * (long)simd_count(dev)) return EPI_SHAPE_HEX;
    return ((long)d->B + kQC - 1) / kQC <= (long)simd_count(dev) ? EPI_SHAPE_QUAD : EPI_SHAPE_LANE;
}

static WsLayout ws_layout(const epi_batch_desc *d)
{
    const int m = MODEL_TABLE[d->model].m;
    const size_t Bp = padded_chains(d);
    const size_t nS = (size_t)d->T * m * Bp * sizeof(double), nP = (size_t)d->T * m * m * Bp * sizeof(double);
    WsLayout w{};
    size_t off = 0;
    auto take = [&](bool need, size_t n) { size_t o = off; if (need) off += (n + 255) & ~(size_t)255; return o; };
    // fp32 storage: the forward quantities the smoother reads back are fp64 workspace whatever the caller selected
    const uint32_t om = d->storage ? (d->out_mask & ~(uint32_t)(EPI_OUT_S_MINUS | EPI_OUT_S_PLUS | EPI_OUT_P_MINUS | EPI_OUT_P_PLUS)) : d->out_mask;
    w.s_minus = take(!(om & EPI_OUT_S_MINUS), nS);
    w.s_plus = take(!(om & EPI_OUT_S_PLUS), nS);
    w.p_minus = take(!(om & EPI_OUT_P_MINUS), nP);
    w.p_plus = take(!(om & EPI_OUT_P_PLUS), nP);
    // smoother intermediates of the generic models: X = pinv(P_MINUS), packed upper triangle, and its rank word per (step, chain)
    const bool generic = MODEL_TABLE[d->model].generic;
    w.x = take(generic, (size_t)d->T * (m * (m + 1) / 2) * Bp * sizeof(double));
    w.rank = take(generic, (size_t)d->T * Bp * sizeof(int32_t));
    w.flag = take(generic, 256);
    // ekf_monitor reads the fp64 innovations: workspace when the caller does not take them as an fp64 output
    w.innov = take(monitor_hoisted(d) && (d->storage || !(d->out_mask & EPI_OUT_INNOVATIONS)), (size_t)d->T * Bp * sizeof(double));
    // hand-over rows between two backward launches (epi_sweep_run_device cuts the smoother where the horizon ends)
    w.hand_s = take(generic, (size_t)m * Bp * sizeof(double));
    w.hand_p = take(generic, (size_t)m * m * Bp * sizeof(double));
    w.hand_i = take(generic, Bp * sizeof(int32_t));
    // exact_nonfinite: the per-chain mask of the dense second pass (+ its counter) and a status array of the library's own
    // (the caller need not pass one)
    w.only = take(generic && d->exact_nonfinite >= 0, (2 * Bp + 1) * sizeof(int32_t));
    w.status = take(generic && d->exact_nonfinite >= 0, Bp * sizeof(int32_t));
    w.total = off;
    return w;
}

// ---------------------------------------------------------------------------
// launch logic
// ---------------------------------------------------------------------------
// Lanes per wave for the one-chain-per-lane kernels.  `waves_per_simd` waves of such a kernel fit a SIMD (1 for the
// 6-state kernels, 2 for the 3-state ones).  6 states: if cn/64 waves exceed what is resident at once, the launch would run
// in rounds and the last round would leave most SIMDs idle while its few waves are limited by what ONE compute unit
// can pull from memory; narrower waves, a multiple of 8 lanes (64-byte segments), make every round equally full.
// 3 states: always full 64-lane waves.  Their arrays have 3 and 9 rows, so a wave writes 1.3 / 4 KB per array and step, and
// chunks that small only reach the HBM's rate when every row is a whole number of cache lines (64 lanes = 512 B):
// profiles/layout_probe/run_rows.py measures 4.7 / 5.2 TB/s for 3- / 9-row chunks of 56 lanes against 6.2 / 6.3 TB/s with 64
// (with 12 rows and more the width stops mattering), and BASELINE config 5 runs 20.6 -> 17.5 ms with full waves although its
// last round is a third full (profiles/r03/README.md).
static int balanced_lanes(int cn, int waves_per_simd, int dev)
{
    if (waves_per_simd >= 2) return kWave;
    const long cap = (long)simd_count(dev) * waves_per_simd;
    const long w64 = (cn + kWave - 1) / kWave;
    if (w64 <= cap) return kWave;
    const long rounds = (w64 + cap - 1) / cap;
    long lw = (cn + rounds * cap - 1) / (rounds * cap);
    lw = (lw + 7) / 8 * 8;
    return (int)(lw > kWave ? kWave : lw);
}

// the scoring tail of the Pareto sweep (epi_sweep_run_device): device pointers, validated by the entry point
struct Tail {
    int t_hist, R, P;
    const double *sp, *J0_prefix, *J1_prefix;
    double *J0, *J1;
    int32_t *on_front, *i_opt;
};
// what one call enqueues
struct Launch {
    int dev, phase, hint, time_pipe;
    int test_flags;  // epi_batch_desc.test_flags (0 in production)
    bool smooth;
    const Tail *tail;
    bool rerun;      // epi_batch_desc.exact_nonfinite: dense second pass over the chains whose covariance went non-finite
};

// the innovation monitor as its own launch (after the forward kernel that wrote the innovations)
template <int FLIP>
static hipError_t launch_monitor(const KArgs &ka, int dev, hipStream_t st)
{
    if (!ka.mon_hoist || !(ka.rho || ka.f.rho)) return hipSuccess;
#ifdef EPI_PROBE_NO_MONITOR        // timing probe: rho is not computed
    return hipSuccess;
#endif
    const size_t shm = (size_t)4 * ka.L * kWave * sizeof(double);
    const int mb = (ka.B + kWave - 1) / kWave;
#ifndef EPI_MONITOR_PAR_MAX_WAVES
#define EPI_MONITOR_PAR_MAX_WAVES 2       // use the scan-free grid (ekf_monitor_par) while the batch has at most TWO 64-chain waves per SIMD (131 072 chains;
                                          // round 6: the headline's 75 000 chains included -- forward stage + monitor 5.45-5.6 against 5.8-6.3 ms, the pass
                                          // 0.3-0.5 ms shorter in each of four alternating runs, profiles/r06/ab_monitor_par.txt; round 5 had it at one)
#endif
    if (ka.L == 21 && !ka.mon_scan && (long)mb <= (long)EPI_MONITOR_PAR_MAX_WAVES * simd_count(dev)) {
        constexpr int D = 8;
        hipLaunchKernelGGL((ekf_monitor_par<FLIP, 21, D>), dim3(mb, (ka.T + D - 1) / D), dim3(kWave), 0, st, ka, ka.dense_flag);
        return hipGetLastError();
    }
    // time segments (each pays a 2L-2-step warm-up): enough of them that the grid gives every SIMD about eight of these
    // light waves (a lane's 100-step scan is latency-bound: 0.70 ms with two segments, 0.25 ms with eight at 75 000 chains),
    // none shorter than ~64 steps
    int nseg = (8 * simd_count(dev) + mb - 1) / mb;
    const int max_seg = ka.T / 64 > 1 ? ka.T / 64 : 1;
    nseg = nseg < 1 ? 1 : (nseg > max_seg ? max_seg : nseg);
    if (ka.L == 21) hipLaunchKernelGGL((ekf_monitor<FLIP, 21>), dim3(mb, nseg), dim3(kWave), shm, st, ka, ka.dense_flag);
    else hipLaunchKernelGGL((ekf_monitor<FLIP, 0>), dim3(mb, nseg), dim3(kWave), shm, st, ka, ka.dense_flag);
    return hipGetLastError();
}

#ifndef EPI_FWD3_LEAN
#define EPI_FWD3_LEAN 0
#endif
#ifndef EPI_FWD3_LDS_PAD
#define EPI_FWD3_LDS_PAD 0          // probe: extra dynamic LDS per workgroup of the 3-state LP = 2 forward variant (caps its waves per SIMD)
#endif
// forward kernel(s) over filter steps [ka.k_begin, ka.k_end) of all chains
template <int M, int FLIP, int GENERIC>
static hipError_t enqueue_fwd(KArgs ka, const Launch &L, hipStream_t st, int c0 = 0, int cn = -1)
{
    ka.c0 = 0; ka.cn = ka.B;
    ka.lw = balanced_lanes(ka.B, M == 6 ? 1 : 2, L.dev);
    if (cn >= 0) { ka.c0 = c0; ka.cn = cn; }      // a chain range (a multiple of the wave's lanes; the one-lane kernels only)
    const int blocks = (ka.cn + ka.lw - 1) / ka.lw;
    const size_t shmem = (size_t)3 * ka.L * kWave * sizeof(double);
    // hint 0: both variants are enqueued, the one ekf_precheck did not select returns at once
    const bool run_sym = GENERIC && L.hint != 2, run_dense = !GENERIC || L.hint != 1;
    hipError_t e = hipSuccess;
    if constexpr (M == 6 && !GENERIC) {
        if (ka.wave && !ka.only) {  // NewCaseEKFEstimatorWithOptimalNPI, one wavefront per chain (monitor always inline)
            const size_t wshm = (size_t)6 * ka.L * sizeof(double);
            if (ka.L == 21) hipLaunchKernelGGL((ekf_fwd_wave<0, 1, 21, 0>), dim3((unsigned)ka.cn), dim3(kWave), wshm, st, ka, (const int *)nullptr);
            else hipLaunchKernelGGL((ekf_fwd_wave<0, 1, 0, 0>), dim3((unsigned)ka.cn), dim3(kWave), wshm, st, ka, (const int *)nullptr);
            return hipGetLastError();
        }
    }
    if (run_sym) {
        bool done = false;
        if constexpr (M == 6 && GENERIC) {
            if (ka.wave) {          // one wavefront per chain (ekf_wave.hpp); with a scalar R_v the monitor runs inline
                if (ka.mon_hoist) hipLaunchKernelGGL((ekf_fwd_wave<FLIP, 0>), dim3((unsigned)ka.cn), dim3(kWave), 0, st, ka, ka.dense_flag);
                else if (ka.L == 21) hipLaunchKernelGGL((ekf_fwd_wave<FLIP, 1, 21>), dim3((unsigned)ka.cn), dim3(kWave), (size_t)6 * ka.L * sizeof(double), st, ka, ka.dense_flag);
                else hipLaunchKernelGGL((ekf_fwd_wave<FLIP, 1, 0>), dim3((unsigned)ka.cn), dim3(kWave), (size_t)6 * ka.L * sizeof(double), st, ka, ka.dense_flag);
                if ((e = hipGetLastError()) != hipSuccess) return e;
                done = true;
            }
        }
        if constexpr (M == 3 && GENERIC) {
            if (ka.wave) {          // seven chains per wavefront, nine lanes each
                hipLaunchKernelGGL((ekf_fwd_wave3<FLIP>), dim3((unsigned)((ka.cn + kW3G - 1) / kW3G)), dim3(kWave), 0, st, ka, ka.dense_flag);
                if ((e = hipGetLastError()) != hipSuccess) return e;
                done = true;
            }
        }
        if constexpr (M == 6 && GENERIC) {
            if (ka.hex && !done) {  // six lanes per chain, ten chains per wavefront (ekf_hex.hpp)
                const unsigned hblocks = (unsigned)((ka.cn + kHG - 1) / kHG);
                const bool solo = (long)hblocks <= (long)simd_count(L.dev);   // every wave can have a SIMD of its own: keep it that way
                if (ka.blk == kHG) {
                    if (solo) hipLaunchKernelGGL((ekf_fwd_hex<FLIP, kHG, 1>), dim3(hblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                    else hipLaunchKernelGGL((ekf_fwd_hex<FLIP, kHG, 0>), dim3(hblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                } else {
                    if (solo) hipLaunchKernelGGL((ekf_fwd_hex<FLIP, 0, 1>), dim3(hblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                    else hipLaunchKernelGGL((ekf_fwd_hex<FLIP, 0, 0>), dim3(hblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                }
                if ((e = hipGetLastError()) != hipSuccess) return e;
                done = true;
            }
        }
        if constexpr (M == 6 && GENERIC) {
            if (ka.quad && !done) {
                // four lanes per chain, 16 chains per wavefront (ekf_quad.hpp): the specialisation for the layout and the
                // window length this shape is meant for, or the general one
                const int qblocks = (ka.cn + kQC - 1) / kQC;
                const size_t qshm = ((size_t)(ka.mon_hoist ? 0 : 6 * ka.L) + kNpi) * kQC * sizeof(double);
                const bool fast = ka.blk == kQC && ka.L == 21;
                const bool solo = qblocks <= simd_count(L.dev);   // every wave can have a SIMD of its own: keep it that way
                auto go = [&](auto kern) -> hipError_t {
                    if (qshm > 64u * 1024u) {      // inline monitor with a long window: above the default dynamic-LDS limit
                        const hipError_t ea = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qshm);
                        if (ea != hipSuccess) return ea;
                    }
                    hipLaunchKernelGGL(kern, dim3(qblocks), dim3(kWave), qshm, st, ka, ka.dense_flag);
                    return hipGetLastError();
                };
                if (ka.mon_hoist) {
                    if (solo) e = fast ? go(ekf_fwd_quad<FLIP, kQC, 21, 0, 1>) : go(ekf_fwd_quad<FLIP, 0, 0, 0, 1>);
                    else e = fast ? go(ekf_fwd_quad<FLIP, kQC, 21, 0, 0>) : go(ekf_fwd_quad<FLIP, 0, 0, 0, 0>);
                } else {
                    e = fast ? go(ekf_fwd_quad<FLIP, kQC, 21, 1, 0>) : go(ekf_fwd_quad<FLIP, 0, 0, 1, 0>);
                }
                if (e != hipSuccess) return e;
                done = true;
            }
        }
        if (!done) {
            // narrow (balanced) waves: the variant with LDS-resident model constants and LDS sized by 40 lanes -- 298
            // instead of 408 VGPRs (a third of the AGPR traffic), still four workgroups per CU
            const size_t per_lane = ((size_t)3 * ka.L + 4 * kNpi) * sizeof(double);
            const bool lp = M == 6 && ka.lw <= kPipeLanes && per_lane * kPipeLanes * 4 <= 160u * 1024u;
            if (ka.mon_hoist) {         // no windows in LDS: only the LP variant's model vectors
                const size_t lp_shm = (size_t)4 * kNpi * kPipeLanes * sizeof(double);
#if EPI_FWD3_LEAN
                // Measured and NOT adopted (round 5, verdict r04 item 3): the 3-state forward kernel with a, u_max in LDS and a day's
                // controls consumed on arrival (LP = 2) needs 166 registers, no scratch -- THREE clean waves per SIMD -- and takes
                // 11.4-11.8 ms on BASELINE config 5 against 8.8-9.5 for the 224-register kernel at two (capped at two waves by an LDS
                // pad: 12.3): the 24 LDS reads a day on the alpha map's dependent chain cost more than the third wave returns
                // (profiles/r05/ab_cfg5_three_waves.txt).  -DEPI_FWD3_LEAN=1 builds it.
                if (ka.stor && M == 3 && ka.lw == kWave)
                    hipLaunchKernelGGL((ekf_fwd_sym<M, FLIP, (M == 3 ? 2 : 0), 1, 0>), dim3(blocks), dim3(kWave), (size_t)2 * kNpi * kWave * sizeof(double) + EPI_FWD3_LDS_PAD, st, ka, ka.dense_flag);
                else
#endif
                if (ka.stor) hipLaunchKernelGGL((ekf_fwd_sym<M, FLIP, 0, 1, 0>), dim3(blocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                else if (lp && M == 6 && ka.ws_upper == 3)     // reduced outputs: the variant that fits two waves per SIMD (USD)
                    hipLaunchKernelGGL((ekf_fwd_sym<M, FLIP, 1, 0, 0, 1>), dim3(blocks), dim3(kWave), lp_shm, st, ka, ka.dense_flag);
                else if (lp) hipLaunchKernelGGL((ekf_fwd_sym<M, FLIP, 1, 0, 0>), dim3(blocks), dim3(kWave), lp_shm, st, ka, ka.dense_flag);
                else hipLaunchKernelGGL((ekf_fwd_sym<M, FLIP, 0, 0, 0>), dim3(blocks), dim3(kWave), 0, st, ka, ka.dense_flag);
            } else if (ka.stor) hipLaunchKernelGGL((ekf_fwd_sym<M, FLIP, 0, 1>), dim3(blocks), dim3(kWave), shmem, st, ka, ka.dense_flag);
            else if (lp) {
                const size_t shm = per_lane * kPipeLanes;
                if (shm > 64u * 1024u &&
                    (e = hipFuncSetAttribute((const void *)ekf_fwd_sym<M, FLIP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)) != hipSuccess)
                    return e;
                hipLaunchKernelGGL((ekf_fwd_sym<M, FLIP, 1>), dim3(blocks), dim3(kWave), shm, st, ka, ka.dense_flag);
            } else hipLaunchKernelGGL((ekf_fwd_sym<M, FLIP, 0>), dim3(blocks), dim3(kWave), shmem, st, ka, ka.dense_flag);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
    }
    if (run_dense) {
        hipLaunchKernelGGL((ekf_fwd<M, FLIP, GENERIC>), dim3(blocks), dim3(kWave), shmem, st, ka);
        e = hipGetLastError();
    }
    return e;
}

// X = pinv(P(j|j-1)) for the `nsteps` array positions from pinv_pos0 + step0 on, all chains
template <int M>
static hipError_t enqueue_pinv(KArgs ka, int step0, int nsteps, hipStream_t st, int c0 = 0, int cn = -1)
{
    if (nsteps <= 0) return hipSuccess;
    ka.c0 = 0; ka.cn = ka.B; ka.pinv_step0 = step0;
    if (cn >= 0) { ka.c0 = c0; ka.cn = cn; }
    // x extent rounded up to a multiple of 8: the kernel re-orders its tiles so that every XCD gets a contiguous eighth per step
    unsigned tiles = (unsigned)((ka.cn + pinv_wg<M>() - 1) / pinv_wg<M>());
    if (ka.only && tiles > 8u) tiles = 8u;      // the second pass of exact_nonfinite walks the list of marked chains (see eks_pinv)
    if (ka.only) hipLaunchKernelGGL((eks_pinv<M, 1>), dim3(tiles >= 8u ? ((tiles + 7u) & ~7u) : tiles, (unsigned)nsteps), dim3(pinv_wg<M>()), 0, st, ka);
    else hipLaunchKernelGGL((eks_pinv<M, 0>), dim3(tiles >= 8u ? ((tiles + 7u) & ~7u) : tiles, (unsigned)nsteps), dim3(pinv_wg<M>()), 0, st, ka);
    return hipGetLastError();
}

// backward recursion over smoother steps ka.bk_from ... ka.bk_to (the dense kernels only know the full range)
// (c0, cn: a chain range -- a multiple of the chains one wave of the shape holds: the layout block of the fixed-descriptor one-lane
// smoother, ten chains in the hex shape)
template <int M, int FLIP, int GENERIC>
static hipError_t enqueue_bwd(KArgs ka, const Launch &L, hipStream_t st, int c0 = 0, int cn = -1)
{
    ka.c0 = 0; ka.cn = ka.B;
    ka.lw = balanced_lanes(ka.B, M == 6 ? 1 : 2, L.dev);
    if (cn >= 0) { ka.c0 = c0; ka.cn = cn; }      // a chain range: a multiple of the chains a wave of the shape holds
    const int blocks = (ka.cn + ka.lw - 1) / ka.lw;
    const bool run_sym = GENERIC && L.hint != 2, run_dense = !GENERIC || L.hint != 1;
    hipError_t e = hipSuccess;
    if constexpr (M == 6 && !GENERIC) {
        if (ka.wave && !ka.only) {
            hipLaunchKernelGGL(eks_bwd_wave_nc, dim3((unsigned)ka.cn), dim3(kWave), 0, st, ka);
            return hipGetLastError();
        }
    }
    if (run_sym) {
        bool done = false;
        if constexpr (M == 6 && GENERIC) {
            if (ka.wave) {
                hipLaunchKernelGGL((eks_bwd_wave<FLIP>), dim3((unsigned)ka.cn), dim3(kWave), 0, st, ka, ka.dense_flag);
                done = true;
            }
        }
        if constexpr (M == 3 && GENERIC) {
            if (ka.wave) {
                hipLaunchKernelGGL((eks_bwd_wave3<FLIP>), dim3((unsigned)((ka.cn + kW3G - 1) / kW3G)), dim3(kWave), 0, st, ka, ka.dense_flag);
                done = true;
            }
        }
        if constexpr (M == 6 && GENERIC) {
            if (ka.hex && !done) {
                const unsigned hblocks = (unsigned)((ka.cn + kHG - 1) / kHG);
                const bool pf = (long)hblocks <= (long)simd_count(L.dev);      // one wave per SIMD: prefetch; beyond: two waves per SIMD
                if (ka.blk == kHG) {
#ifndef EPI_HEX_BWD_DMA
#define EPI_HEX_BWD_DMA 1
#endif
                    if (pf && EPI_HEX_BWD_DMA) hipLaunchKernelGGL((eks_bwd_hex<FLIP, kHG, 2>), dim3(hblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                    else if (pf) hipLaunchKernelGGL((eks_bwd_hex<FLIP, kHG, 1>), dim3(hblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                    else hipLaunchKernelGGL((eks_bwd_hex<FLIP, kHG, 0>), dim3(hblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                } else {
                    if (pf) hipLaunchKernelGGL((eks_bwd_hex<FLIP, 0, 1>), dim3(hblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                    else hipLaunchKernelGGL((eks_bwd_hex<FLIP, 0, 0>), dim3(hblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                }
                done = true;
            }
        }
        if constexpr (M == 6 && GENERIC) {
            if (ka.quad && !done) {
                const int qblocks = (ka.cn + kQC - 1) / kQC;
                if (ka.blk == kQC) hipLaunchKernelGGL((eks_bwd_quad<FLIP, kQC>), dim3(qblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                else hipLaunchKernelGGL((eks_bwd_quad<FLIP, 0>), dim3(qblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                done = true;
            }
        }
        if constexpr (M == 6 && GENERIC) {
            // one lane per chain with fixed descriptors per addressing window (ekf_lane6.hpp): the layout block must be the
            // lanes a workgroup uses, and a compile-time constant
            // (not for 64-chain blocks: four workgroups' LDS columns would not fit a CU, and where three suffice -- one round of 64-lane
            // waves, the N = 2 shard of the headline sweep: 37 500 chains -- the kernel measured 3.84 against eks_bwd_sym's 3.5 ms:
            // such a batch is not issue-bound, profiles/r06/ab_n2_shard.txt)
            const bool l6_ok = lane6_block(ka.blk);
            if (!done && EPI_LANE6_BWD && !ka.stor && l6_ok && (long)ka.blk * ka.nblk <= (1L << 20)) {
                const int lblocks = (ka.cn + ka.blk - 1) / ka.blk;
#if EPI_LANE6_BWD == 3
                if (ka.blk == 40) hipLaunchKernelGGL((eks_bwd_lane6d<FLIP, 40>), dim3(lblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
#else
                if (ka.blk == 40 && EPI_LANE6_XD && ka.cn % 40 == 0) hipLaunchKernelGGL((eks_bwd_lane6<FLIP, 40, (EPI_LANE6_BWD > 1), EPI_LANE6_LATE_PF, 1>), dim3(lblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                else if (ka.blk == 40) hipLaunchKernelGGL((eks_bwd_lane6<FLIP, 40, (EPI_LANE6_BWD > 1)>), dim3(lblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
#endif
                else if (ka.blk == 48) hipLaunchKernelGGL((eks_bwd_lane6<FLIP, 48, (EPI_LANE6_BWD > 1)>), dim3(lblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                else hipLaunchKernelGGL((eks_bwd_lane6<FLIP, 56, (EPI_LANE6_BWD > 1)>), dim3(lblocks), dim3(kWave), 0, st, ka, ka.dense_flag);
                done = true;
            }
        }
        if (!done) {
            if (ka.stor) hipLaunchKernelGGL((eks_bwd_sym<M, FLIP, 1>), dim3(blocks), dim3(kWave), 0, st, ka, ka.dense_flag);
            else hipLaunchKernelGGL((eks_bwd_sym<M, FLIP>), dim3(blocks), dim3(kWave), 0, st, ka, ka.dense_flag);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (run_dense) {
        hipLaunchKernelGGL((eks_bwd<M, FLIP, GENERIC>), dim3(blocks), dim3(kWave), 0, st, ka);
        e = hipGetLastError();
    }
    return e;
}

// scoring of the sweep's horizon (TrainPredictPrescribeNPI.m:481-493) from the u_opt_smooth the smoother just wrote,
// then the Pareto filter and optimum per region (:624-633) when the call holds whole regions
static hipError_t enqueue_tail(const KArgs &ka, const Tail &t, hipStream_t st, const int32_t *gate = nullptr)
{
    epi_sim_desc sd{};
    sd.abi_version = EPIEKF_ABI_VERSION; sd.B = ka.B; sd.K = ka.T - t.t_hist; sd.Su = ka.B; sd.n_npi = ka.n_npi;
    sd.noise = 0; sd.with_cost = 1; sd.prefix_days = t.t_hist; sd.u_block = ka.blk >= ka.B ? 0 : ka.blk;
    // day t_hist of u_opt_smooth: a time slice is n_npi rows of nblk * blk chains in either layout
    const double *u_h = ka.u_opt_smooth + (size_t)t.t_hist * ka.n_npi * ((size_t)ka.nblk * ka.blk);
    hipLaunchKernelGGL(sialpha_sim, dim3((ka.B + 255) / 256), dim3(256), 0, st, sd, (const int32_t *)nullptr, u_h, t.sp,
                       (const double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr, t.J0, t.J1,
                       t.J0_prefix, t.J1_prefix, gate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || (!t.on_front && !t.i_opt)) return e;
    hipLaunchKernelGGL(pareto_front, dim3(t.R), dim3(256), (size_t)2 * t.P * sizeof(double), st, t.P, t.J0, t.J1, t.on_front, t.i_opt, gate);
    return hipGetLastError();
}

// Forward pass in time segments (percent of T where each ends).  pinv(P(k|k-1)) needs nothing but the forward pass's
// output of day k and the smoother consumes X in REVERSE time order, so only the pinv grid of the last segment stands
// between the end of the forward pass and the start of the smoother: the last segments are short.
// (re-measured in round 3 with the cheaper pinv: four segments ending at 50 / 85 / 97 / 100 % give 3.06-3.11 ms against
// 3.12-3.16 for five ending at 40 / 70 / 90 / 98 / 100 % at 9 375 chains, level at 18 750: profiles/r03/time_cuts.txt)
#ifndef EPI_TIME_CUTS
#define EPI_TIME_CUTS 0, 50, 85, 97, 100
#endif
constexpr int kTimeCuts[] = {EPI_TIME_CUTS};
constexpr int kTimeSeg = (int)(sizeof(kTimeCuts) / sizeof(kTimeCuts[0])) - 1;

// epi_batch_desc.exact_nonfinite.  The packed / quad kernels skip products with structural zeros, which is exact for finite
// operands; once a chain's covariance has overflowed, a skipped `Inf * 0` leaves a finite number where the dense evaluation
// (the reference's, and MATLAB's BLAS) has NaN.  Up to the first overflow both evaluations hold the same bits, so the
// non-finite guard of :211 fires for such a chain in either (status bit 0).  Those chains -- and only those -- are run again
// from the start by the dense kernels, which mirror the reference term for term, writing over their outputs in place; then
// the scoring tail is repeated if any chain was re-run.  Costs three launches that return at once when no chain is marked
// (the pinv grid's ~(B/64)(T-1) empty workgroups: ~0.15 ms at 75 000 x 520).
template <int M, int FLIP>
static hipError_t rerun_nonfinite_dense(const KArgs &ka, const Launch &L, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(ka.only_buf + ka.B, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mark_nonfinite, dim3((ka.B + 255) / 256), dim3(256), 0, st, (const int32_t *)ka.status, ka.only_buf, ka.B);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    KArgs kd = ka;
    kd.only = ka.only_buf; kd.dense_flag = nullptr; kd.quad = 0; kd.hex = 0; kd.mon_hoist = 0;
    kd.k_begin = 0; kd.k_end = 0; kd.bk_from = ka.T - 2; kd.bk_to = 0;
    Launch Ld = L;
    Ld.hint = 2; Ld.tail = nullptr;
    if ((e = enqueue_fwd<M, FLIP, 1>(kd, Ld, st)) != hipSuccess) return e;
    if ((e = enqueue_pinv<M>(kd, 0, ka.T - 1, st)) != hipSuccess) return e;
    if ((e = enqueue_bwd<M, FLIP, 1>(kd, Ld, st)) != hipSuccess) return e;
    if (L.tail) e = enqueue_tail(ka, *L.tail, st, ka.only_buf + ka.B);
    return e;
}

template <int M, int FLIP, int GENERIC>
static hipError_t launch_chain(const KArgs &ka, const Launch &L, hipStream_t st)
{
    const size_t shmem = (size_t)3 * ka.L * kWave * sizeof(double);
    const int T = ka.T, phase = L.phase;
    hipError_t e = hipSuccess;
    if (phase == 0 || phase == 1) {
        if (shmem > 64u * 1024u) {   // above the default dynamic-LDS limit
            e = hipFuncSetAttribute((const void *)ekf_fwd<M, FLIP, GENERIC>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            if (e != hipSuccess) return e;
            if (GENERIC) {
                e = hipFuncSetAttribute((const void *)ekf_fwd_sym<M, FLIP, 0>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
                if (e != hipSuccess) return e;
                e = hipFuncSetAttribute((const void *)ekf_fwd_sym<M, FLIP, 0, 1>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
                if (e != hipSuccess) return e;
            }
        }
        if (GENERIC) {
            // hint 0: fast path unless ekf_precheck finds a non-symmetric Ps_init / non-diagonal Q_w in the batch;
            // hint 1 / 2: the caller decided; the flag the kernels test is set accordingly
            if ((e = hipMemsetAsync(ka.dense_flag, 0, sizeof(int), st)) != hipSuccess) return e;
            if (L.hint == 0) {
                hipLaunchKernelGGL((ekf_precheck<M>), dim3((ka.B + 255) / 256), dim3(256), 0, st, ka, ka.dense_flag, 0);
                if ((e = hipGetLastError()) != hipSuccess) return e;
            } else if (L.hint == 2) {
                hipLaunchKernelGGL((ekf_precheck<M>), dim3(1), dim3(64), 0, st, ka, ka.dense_flag, 1);
                if ((e = hipGetLastError()) != hipSuccess) return e;
            }
        }
    }
    // smoother positions of filter steps 2..T (array positions 1..T-1, or 0..T-2 for the time-flipped models)
    const bool overlap = GENERIC && phase == 0 && L.smooth && L.hint == 1;
    if (!overlap) {
        // one stage after the other on the caller's stream (single stages for per-kernel timing, the forward pass alone,
        // batches that may take the dense kernels, the NewCase models)
        if (phase == 0 || phase == 1) {
            if ((e = enqueue_fwd<M, FLIP, GENERIC>(ka, L, st)) != hipSuccess) return e;
            if (GENERIC && L.hint != 2 && (e = launch_monitor<FLIP>(ka, L.dev, st)) != hipSuccess) return e;
        }
        if (!L.smooth) return e;
        if (GENERIC && (phase == 0 || phase == 2 || phase == 3) && (e = enqueue_pinv<M>(ka, 0, T - 1, st)) != hipSuccess) return e;
        if (phase == 0 || phase == 2 || phase == 4) {
            if ((e = enqueue_bwd<M, FLIP, GENERIC>(ka, L, st)) != hipSuccess) return e;
        }
        if (phase == 0 && L.tail) e = enqueue_tail(ka, *L.tail, st);
        if constexpr (GENERIC) {
            if (e == hipSuccess && L.rerun && phase == 0 && L.hint != 2) e = rerun_nonfinite_dense<M, FLIP>(ka, L, st);
        }
        return e;
    }

    // A full call on the packed kernels.  What does not lie on the critical path runs on a helper stream:
    //   * the innovation monitor (needs the forward pass only) beside the pinv grid and the smoother;
    //   * pipelined in TIME: a batch that does not fill the chip (the shards of the sweep on 2, 4, 8 GPUs) leaves SIMDs idle
    //     while its sequential forward waves crawl through the T days.  The forward kernel then runs in kTimeSeg launches (a
    //     later segment resumes from the S_MINUS / P_MINUS the previous one stored: same bits), and after each of them the
    //     eks_pinv grid of ITS days starts on the helper stream, beside the next forward segment.  Not for a batch that
    //     fills the chip (nothing idles: measured level, round 1);
    //   * the sweep's scoring tail: the horizon's u_opt_smooth is final after the smoother's first T - 1 - t_hist steps,
    //     so the recursion is cut there and scoring + Pareto filter run beside the rest of it.
    HelperLease lease(L.dev);
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();    // the legacy default stream cannot be queried while another stream captures
        lease.captured = cs != hipStreamCaptureStatusNone;
    }
    if ((e = helper_acquire(L.dev, &lease.h)) != hipSuccess) return e;
    Helper *h = lease.h;
    int ne = 0;
    auto fork = [&](hipStream_t from, hipStream_t to) -> hipError_t {     // `to` continues after what `from` holds now
        hipError_t ee = hipEventRecord(h->ev[ne], from);
        if (ee == hipSuccess) ee = hipStreamWaitEvent(to, h->ev[ne], 0);
        ne++;
        return ee;
    };
    const long fwd_waves = ka.hex ? ((long)ka.B + kHG - 1) / kHG : ka.quad ? ((long)ka.B + kQC - 1) / kQC : ((long)ka.B + kWave - 1) / kWave;
    // (one wavefront per chain: the pinv grid of so few chains takes ~30 us, nothing to pipeline -- unless asked for)
    const bool force_split = (L.test_flags >> 2) & 1;                                                      // test hook
    const bool force_rp = (L.test_flags & 1) && ka.hex && ka.mon_hoist && T >= 4 && L.time_pipe >= 0;      // test hook
    const bool tp = !force_rp && ka.mon_hoist && !ka.stor && T >= 128 && L.time_pipe >= 0 &&
                    (L.time_pipe == 1 || (!ka.wave && fwd_waves * 4 <= (long)simd_count(L.dev) * 3));
    bool helper_busy = false;
    // Pipelined in REVERSE time (round 5, the hex shape): the smoother consumes X = pinv(P(k+1|k)) from the last day backwards, so
    // only the pinv grid of the LAST days has to stand between the forward kernel and the smoother; the grids of the earlier days
    // run on the helper stream beside the smoother's first launches, and the smoother is cut where they end (the hand-over rows of
    // the horizon / observed-days split).  Beside the FORWARD kernel a pinv grid takes exactly what it saves (both want the vector
    // unit: forward segments 1.07 -> 1.50 ms); the smoother of this shape waits on memory most of the time.
#ifndef EPI_REVERSE_PIPE
#define EPI_REVERSE_PIPE 1
#endif
    const bool rp = force_rp || (EPI_REVERSE_PIPE && !tp && ka.hex && ka.mon_hoist && T >= 128 && L.time_pipe >= 0);
    // (Round 6, measured and not adopted: the hex shape beyond one wavefront per SIMD -- 10 241 .. 20 480 chains, the shard of the headline
    // sweep on one of 4 GPUs -- as TWO chain ranges through the one-wave-per-SIMD kernels, pipelined across the ranges (the pinv grid of
    // the first beside the forward kernel of the second, the grid of the second beside the smoother of the first) instead of the
    // two-waves-per-SIMD kernels pipelined in reverse time: 5.20-5.28 against 4.58-4.80 ms at 18 750 chains, 3.75-3.80 against 3.06-3.09 at
    // 10 375 -- profiles/r06/ab_hex_rounds.txt.)
    if (rp) {
        if ((e = enqueue_fwd<M, FLIP, GENERIC>(ka, L, st)) != hipSuccess) return e;
        if ((e = fork(st, h->stream)) != hipSuccess) return e;
        helper_busy = true;
        // smoother steps k = T-2 ... 0 in segments [hi, lo], descending: the horizon first when there is a scoring tail, the rest in
        // three parts; segment s needs X of filter steps lo+1 ... hi+1
        int seg_hi[8], seg_lo[8], ns = 0, tail_seg = -1;
        int top = T - 2;
#ifndef EPI_RP_PARTS
#define EPI_RP_PARTS 3
#endif
#ifndef EPI_RP_FIRST
#define EPI_RP_FIRST 0            // > 0: a first segment of that many days ahead of everything (probe)
#endif
        if (EPI_RP_FIRST > 0 && L.tail && L.tail->t_hist >= 1 && T - 2 - EPI_RP_FIRST > L.tail->t_hist) {
            seg_hi[ns] = top; seg_lo[ns] = top - EPI_RP_FIRST + 1; ns++;
            top -= EPI_RP_FIRST;
        }
        if (L.tail && L.tail->t_hist >= 1 && L.tail->t_hist <= T - 2) {
            seg_hi[ns] = top; seg_lo[ns] = L.tail->t_hist; tail_seg = ns; ns++;
            top = L.tail->t_hist - 1;
        }
        const int parts = (top + 1 >= 96 || (force_rp && top + 1 >= 6)) ? EPI_RP_PARTS : 1;
        for (int q = 0; q < parts; q++) {
            const int lo = (int)((long)(top + 1) * (parts - 1 - q) / parts);
            seg_hi[ns] = top; seg_lo[ns] = lo; ns++;
            top = lo - 1;
        }
        hipEvent_t ev_pinv[8];
        for (int sgm = 0; sgm < ns; sgm++) {
            const int j_lo = seg_lo[sgm] + 1, j_hi = seg_hi[sgm] + 1;
            hipStream_t ps = sgm == 0 ? st : h->stream;
            if ((e = enqueue_pinv<M>(ka, FLIP ? (T - 1 - j_hi) : (j_lo - 1), j_hi - j_lo + 1, ps)) != hipSuccess) return e;
            if (sgm > 0) {
                ev_pinv[sgm] = h->ev[ne++];
                if ((e = hipEventRecord(ev_pinv[sgm], h->stream)) != hipSuccess) return e;
            }
        }
        for (int sgm = 0; sgm < ns; sgm++) {
            if (sgm > 0 && (e = hipStreamWaitEvent(st, ev_pinv[sgm], 0)) != hipSuccess) return e;
            KArgs kb = ka;
            kb.bk_from = seg_hi[sgm]; kb.bk_to = seg_lo[sgm];
            if ((e = enqueue_bwd<M, FLIP, GENERIC>(kb, L, st)) != hipSuccess) return e;
            if (sgm == tail_seg) {              // the horizon's u_opt_smooth is final: scoring + Pareto filter behind the helper's pinv grids
                if ((e = fork(st, h->stream)) != hipSuccess) return e;
                if ((e = enqueue_tail(ka, *L.tail, h->stream)) != hipSuccess) return e;
            }
        }
        if (L.tail && tail_seg < 0 && (e = enqueue_tail(ka, *L.tail, st)) != hipSuccess) return e;
        if (ka.rho || ka.f.rho) {
            if ((e = launch_monitor<FLIP>(ka, L.dev, h->stream)) != hipSuccess) return e;
        }
    } else {
    if (tp) {
        for (int sg = 0; sg < kTimeSeg; sg++) {
            KArgs kc = ka;
            kc.k_begin = (int)((long)T * kTimeCuts[sg] / 100);
            kc.k_end = (sg == kTimeSeg - 1) ? T : (int)((long)T * kTimeCuts[sg + 1] / 100);
            if (kc.k_end <= kc.k_begin) continue;
            if ((e = enqueue_fwd<M, FLIP, GENERIC>(kc, L, st)) != hipSuccess) return e;
            if ((e = fork(st, h->stream)) != hipSuccess) return e;
            // filter steps whose P(j|j-1) exists now and has not been inverted yet: j_lo .. j_hi (the smoother needs
            // j = 1 .. T-1); array position of step j: j, or T-1-j for the time-flipped models
            const int j_lo = kc.k_begin + 1, j_hi = (kc.k_end < T) ? kc.k_end : T - 1;
            if ((e = enqueue_pinv<M>(ka, FLIP ? (T - 1 - j_hi) : (j_lo - 1), j_hi - j_lo + 1, h->stream)) != hipSuccess) return e;
        }
        if ((e = fork(h->stream, st)) != hipSuccess) return e;       // the smoother needs every pinv grid
        helper_busy = true;
    } else {
        // A one-lane batch that runs in ROUNDS of resident waves (more waves than SIMDs: the headline sweep): the chains of the forward
        // kernel's first round and the rest are two launches, and the pinv grid of the first range runs on the helper stream beside
        // the forward kernel of the second (whose 851 waves leave SIMDs and registers free at 75 000 chains; the forward kernel is
        // bound by its stores, the grid by the vector unit): 0.15 ms of the grid's 3.0 hidden (profiles/r06/ab_pinv_beside_fwd.txt).
        // epi_batch_desc.test_flags bit 2 cuts ANY one-lane batch of two waves or more in the middle the same way (test hook).
#ifndef EPI_PINV_BESIDE_FWD
#define EPI_PINV_BESIDE_FWD 1
#endif
        const int lwf = balanced_lanes(ka.B, M == 6 ? 1 : 2, L.dev);
        const long wavesf = ((long)ka.B + lwf - 1) / lwf;
        const int cA = force_split ? (int)(wavesf / 2) * lwf : simd_count(L.dev) * lwf;
        // (not when neither P(k|k-1) nor P(k|k) is an output: that forward kernel fits TWO waves per SIMD and the headline's 1 875 are
        // resident at once -- cut in two launches it took 1.1 ms longer, bench_cfg4_reduced 13.6 -> 14.7 ms)
        if (EPI_PINV_BESIDE_FWD && M == 6 && GENERIC && !ka.hex && !ka.quad && !ka.wave && !ka.only && L.hint != 2 && cA > 0 && cA < ka.B &&
            (force_split || (lwf < kWave && ka.ws_upper != 3))) {
            if ((e = enqueue_fwd<M, FLIP, GENERIC>(ka, L, st, 0, cA)) != hipSuccess) return e;
            if ((e = fork(st, h->stream)) != hipSuccess) return e;
            if ((e = enqueue_pinv<M>(ka, 0, T - 1, h->stream, 0, cA)) != hipSuccess) return e;
            if ((e = enqueue_fwd<M, FLIP, GENERIC>(ka, L, st, cA, ka.B - cA)) != hipSuccess) return e;
            if (ka.mon_hoist && (ka.rho || ka.f.rho) && (e = fork(st, h->stream)) != hipSuccess) return e;
            if ((e = enqueue_pinv<M>(ka, 0, T - 1, st, cA, ka.B - cA)) != hipSuccess) return e;     // beside the tail of the first range's grid
            if ((e = fork(h->stream, st)) != hipSuccess) return e;
        } else {
        if ((e = enqueue_fwd<M, FLIP, GENERIC>(ka, L, st)) != hipSuccess) return e;
        if (ka.mon_hoist && (ka.rho || ka.f.rho) && (e = fork(st, h->stream)) != hipSuccess) return e;
        if ((e = enqueue_pinv<M>(ka, 0, T - 1, st)) != hipSuccess) return e;
        }
    }
    // A one-lane batch that runs in ROUNDS (more 64-chain waves than SIMDs: the headline sweep) on the fixed-descriptor smoother:
    // the chains of the smoother's first round of resident waves and the rest are two launches, and the monitor starts between
    // them -- beside the second launch, whose waves leave SIMDs free (851 waves on 1 024 SIMDs at 75 000 chains), instead of
    // beside the pinv grid, which is bound by the vector unit and simply takes 0.44 ms longer with the monitor next to it.
    const bool in_rounds = !ka.hex && !ka.quad && !ka.wave && fwd_waves > (long)simd_count(L.dev);
    bool mon_late = false;
    int first_round = 0;
    if constexpr (M == 6 && GENERIC) {
#ifndef EPI_MONITOR_LATE
#define EPI_MONITOR_LATE 1
#endif
        const bool cut = force_split && !ka.hex && !ka.quad && !ka.wave && ka.nblk >= 2;        // test hook: the two launches at any size
        // (a first launch of 1 004 / 984 / 964 / 944 waves instead of the 1 024 that are resident at once: level, 14.41-14.88 ms per pass
        // whatever the cut; the monitor ahead of the first launch instead of between the two: +0.15 ms -- profiles/r06/ab_bwd_balance*.txt)
        first_round = cut ? (ka.nblk / 2) * ka.blk : simd_count(L.dev) * ka.blk;
        mon_late = EPI_MONITOR_LATE && !tp && (in_rounds || cut) && EPI_LANE6_BWD && !ka.stor && lane6_block(ka.blk) && (long)ka.blk * ka.nblk <= (1L << 20) &&
                   ka.mon_hoist && (ka.rho || ka.f.rho) && first_round < ka.B;
    }
    if (!mon_late && ka.mon_hoist && (ka.rho || ka.f.rho)) {
        if ((e = launch_monitor<FLIP>(ka, L.dev, h->stream)) != hipSuccess) return e;
        helper_busy = true;
    }
    // (Not for a one-lane batch that runs in ROUNDS of resident waves -- more 64-chain waves than SIMDs, the headline sweep: every
    // SIMD is taken by a 512-register wave, the tail finds no room beside the smoother and a second launch costs the smoother a
    // second pair of rounds: 15.1-15.9 against 15.6-16.2 ms per pass over four alternating runs, profiles/r06/ab_monitor_par.txt.)
    if (!in_rounds && !mon_late && L.tail && L.tail->t_hist >= 1 && L.tail->t_hist <= T - 2 && !ka.wave) {
        KArgs kb = ka;
        kb.bk_from = T - 2; kb.bk_to = L.tail->t_hist;              // the horizon days
        if ((e = enqueue_bwd<M, FLIP, GENERIC>(kb, L, st)) != hipSuccess) return e;
        if ((e = fork(st, h->stream)) != hipSuccess) return e;
        if ((e = enqueue_tail(ka, *L.tail, h->stream)) != hipSuccess) return e;
        helper_busy = true;
        kb.bk_from = L.tail->t_hist - 1; kb.bk_to = 0;              // the observed days
        if ((e = enqueue_bwd<M, FLIP, GENERIC>(kb, L, st)) != hipSuccess) return e;
    } else if (mon_late) {
        if ((e = enqueue_bwd<M, FLIP, GENERIC>(ka, L, st, 0, first_round)) != hipSuccess) return e;
        if ((e = fork(st, h->stream)) != hipSuccess) return e;
        if ((e = launch_monitor<FLIP>(ka, L.dev, h->stream)) != hipSuccess) return e;
        helper_busy = true;
        if ((e = enqueue_bwd<M, FLIP, GENERIC>(ka, L, st, first_round, ka.B - first_round)) != hipSuccess) return e;
        if (L.tail && (e = enqueue_tail(ka, *L.tail, st)) != hipSuccess) return e;
    } else {
        if ((e = enqueue_bwd<M, FLIP, GENERIC>(ka, L, st)) != hipSuccess) return e;
        if (L.tail && (e = enqueue_tail(ka, *L.tail, st)) != hipSuccess) return e;
    }
    }
    if (helper_busy) e = fork(h->stream, st);
    if constexpr (GENERIC) {
        if (e == hipSuccess && L.rerun) e = rerun_nonfinite_dense<M, FLIP>(ka, L, st);
    }
    return e;
}

}  // namespace epi

using namespace epi;

extern "C" {

int epi_abi_version(void) { return EPIEKF_ABI_VERSION; }

const char *epi_status_string(int status)
{
    switch (status) {
    case EPI_OK: return "ok";
    case EPI_ERR_UNDEFINED_ORDER: return "Undefined order";
    case EPI_ERR_Q_MISMATCH: return "Process noise covariance noise mismatch";
    case EPI_ERR_R_MISMATCH: return "Observation noise covariance noise mismatch";
    case EPI_ERR_OBS_TYPE: return "unknown observation type";
    case EPI_ERR_BAD_ARG: return "bad argument";
    case EPI_ERR_WORKSPACE: return "workspace too small";
    case EPI_ERR_HIP: return "HIP runtime error";
    case EPI_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
    }
}

int epi_model_dim(int model) { return (model >= 0 && model < 6) ? MODEL_TABLE[model].m : -1; }

int epi_ekf_validate(const epi_batch_desc *d, char *err)
{
    if (!d) { set_err(err, "NULL descriptor"); return EPI_ERR_BAD_ARG; }
    if (d->abi_version != EPIEKF_ABI_VERSION) { set_err(err, "ABI version mismatch"); return EPI_ERR_BAD_ARG; }
    if (d->model < 0 || d->model > 5) { set_err(err, "unknown model"); return EPI_ERR_BAD_ARG; }
    if (d->B < 1 || d->T < 1 || d->Sx < 1 || d->Su < 1 || d->L < 1) { set_err(err, "B, T, Sx, Su, L must be >= 1"); return EPI_ERR_BAD_ARG; }
    if (d->T > 65536) { set_err(err, "T is limited to 65536 (grid y dimension of eks_pinv)"); return EPI_ERR_BAD_ARG; }
    if (d->B > (1 << 23) || d->Sx > (1 << 23) || d->Su > (1 << 23)) { set_err(err, "B, Sx, Su are limited to 2^23 (a per-step array slice is addressed through one 4 GiB buffer descriptor)"); return EPI_ERR_BAD_ARG; }
    if (d->n_npi < 1 || d->n_npi > EPI_MAX_NPI) { set_err(err, "n_npi out of range 1..12"); return EPI_ERR_BAD_ARG; }
    if (d->order != 1 && d->order != 2) { set_err(err, epi_status_string(EPI_ERR_UNDEFINED_ORDER)); return EPI_ERR_UNDEFINED_ORDER; }
    const ModelInfo &mi = MODEL_TABLE[d->model];
    if (!mi.obs_fixed && d->obs_type != EPI_OBS_NEWCASES && d->obs_type != EPI_OBS_TOTALCASES) {
        set_err(err, epi_status_string(EPI_ERR_OBS_TYPE)); return EPI_ERR_OBS_TYPE;
    }
    if (d->q_mode != 0 && d->q_mode != 1) { set_err(err, epi_status_string(EPI_ERR_Q_MISMATCH)); return EPI_ERR_Q_MISMATCH; }
    if (!mi.generic && d->q_mode != 0) {   // NewCase...m:30  Q = Q_w is added to an m x m matrix as it is
        set_err(err, epi_status_string(EPI_ERR_Q_MISMATCH)); return EPI_ERR_Q_MISMATCH;
    }
    if (d->r_mode != 0 && d->r_mode != 1) { set_err(err, epi_status_string(EPI_ERR_R_MISMATCH)); return EPI_ERR_R_MISMATCH; }
    if (!mi.generic && d->r_mode != 0) {   // NewCase...m:31  R = R_v is used as a scalar
        set_err(err, epi_status_string(EPI_ERR_R_MISMATCH)); return EPI_ERR_R_MISMATCH;
    }
    // three fp64 windows of L samples per lane must fit the CU's 160 KiB LDS
    if (d->phase < 0 || d->phase > 4) { set_err(err, "phase must be 0..4"); return EPI_ERR_BAD_ARG; }
    if (d->path_hint < 0 || d->path_hint > 2) { set_err(err, "path_hint must be 0, 1 or 2"); return EPI_ERR_BAD_ARG; }
    if (d->time_pipe < -1 || d->time_pipe > 1) { set_err(err, "time_pipe must be -1 (off), 0 (auto) or 1 (on)"); return EPI_ERR_BAD_ARG; }
    if (d->lane_block < 0) { set_err(err, "lane_block must be >= 0"); return EPI_ERR_BAD_ARG; }
    if (d->shape < 0 || d->shape > 4) { set_err(err, "shape must be 0 (auto), 1 (one lane per chain), 2 (four lanes per chain), 3 (one wavefront per chain) or 4 (six lanes per chain)"); return EPI_ERR_BAD_ARG; }
    if (d->storage < 0 || d->storage > 1) { set_err(err, "storage must be 0 (fp64) or 1 (fp32)"); return EPI_ERR_BAD_ARG; }
    if (d->exact_nonfinite < -1 || d->exact_nonfinite > 1) { set_err(err, "exact_nonfinite must be 0 (default: on), 1 (on) or -1 (off)"); return EPI_ERR_BAD_ARG; }
    if (d->placement_tries < 0 || d->placement_tries > EPI_PLACEMENT_MAX_TRIES) { set_err(err, "placement_tries must be 0 .. EPI_PLACEMENT_MAX_TRIES"); return EPI_ERR_BAD_ARG; }
    if (d->test_window < 0 || d->test_window == 1) { set_err(err, "test_window must be 0 (production) or >= 2"); return EPI_ERR_BAD_ARG; }
    if (d->test_flags < 0 || d->test_flags > 7) { set_err(err, "test_flags must be 0 (production) .. 7"); return EPI_ERR_BAD_ARG; }
    if (padded_chains(d) > ((size_t)1 << 23)) { set_err(err, "B rounded up to lane_block exceeds 2^23"); return EPI_ERR_BAD_ARG; }
    if ((size_t)3 * d->L * kWave * sizeof(double) > 160u * 1024u) { set_err(err, "inv_monitor_len too large for LDS (max 106)"); return EPI_ERR_UNSUPPORTED; }
    return EPI_OK;
}

size_t epi_ekf_workspace_bytes(const epi_batch_desc *d)
{
    if (epi_ekf_validate(d, nullptr) != EPI_OK) return 0;
    return ws_layout(d).total;
}

int epi_ekf_preferred_lane_block(const epi_batch_desc *d)
{
    epi_batch_desc probe;
    if (!d) return 0;
    probe = *d; probe.lane_block = 0;
    if (epi_ekf_validate(&probe, nullptr) != EPI_OK) return 0;
    const ModelInfo &mi = MODEL_TABLE[d->model];
    const int dev = current_device();
    const int sh = shape_of(d, dev);
    const int lw = sh == EPI_SHAPE_WAVE ? 1 : (sh == EPI_SHAPE_HEX ? kHG : (sh == EPI_SHAPE_QUAD ? kQC : balanced_lanes(d->B, mi.m == 6 ? 1 : 2, dev)));
    return lw < d->B ? lw : d->B;
}

static int run_device_impl(const epi_batch_desc *d, const epi_inputs *in, const epi_outputs *out,
                           void *workspace, size_t workspace_bytes, const Tail *tail, void *stream, char *err)
{
    int rc = epi_ekf_validate(d, err);
    if (rc != EPI_OK) return rc;
    if (!in || !out) { set_err(err, "NULL inputs/outputs"); return EPI_ERR_BAD_ARG; }
    const ModelInfo &mi = MODEL_TABLE[d->model];
    if (!in->x || !in->u || !in->prm || !in->s_init || !in->Ps_init || !in->s_final || !in->Ps_final || !in->Q) {
        set_err(err, "NULL input array"); return EPI_ERR_BAD_ARG;
    }
    if (d->r_mode == 0 && !in->R_scalar) { set_err(err, "r_mode 0 needs R_scalar"); return EPI_ERR_BAD_ARG; }
    if (d->r_mode == 1 && !in->R_series) { set_err(err, "r_mode 1 needs R_series"); return EPI_ERR_BAD_ARG; }
    if (!in->x_series && d->Sx != d->B) { set_err(err, "identity x_series needs Sx == B"); return EPI_ERR_BAD_ARG; }
    if (!in->u_series && d->Su != d->B) { set_err(err, "identity u_series needs Su == B"); return EPI_ERR_BAD_ARG; }
    const WsLayout wl = ws_layout(d);
    if (wl.total > 0 && (!workspace || workspace_bytes < wl.total)) {
        set_err(err, epi_status_string(EPI_ERR_WORKSPACE)); return EPI_ERR_WORKSPACE;
    }
    const bool f32 = d->storage == 1;
    if (f32 && (!mi.generic || d->q_mode != 0 || d->path_hint != 1)) {
        set_err(err, "storage = 1 (fp32) runs the packed kernels of the generic models only: fixed Q_w and path_hint = 1 "
                     "(epi_ekf_precheck_device said the batch qualifies)");
        return EPI_ERR_UNSUPPORTED;
    }
    // fp32 storage: every fp64 output pointer the kernels see is either workspace (forward quantities) or NULL
    const uint32_t om = f32 ? 0u : d->out_mask;
    const uint32_t om32 = f32 ? d->out_mask : 0u;
    auto sel = [&](uint32_t bit, double *p) -> double * { return (om & bit) ? p : nullptr; };
    auto sel32 = [&](uint32_t bit, double *p) -> float * { return (om32 & bit) ? (float *)p : nullptr; };
    char *ws = (char *)workspace;
    KArgs ka{};
    ka.B = d->B; ka.T = d->T; ka.Sx = d->Sx; ka.Su = d->Su; ka.n_npi = d->n_npi; ka.L = d->L; ka.r_mode = d->r_mode; ka.q_mode = d->q_mode;
    ka.blk = lane_block_of(d); ka.nblk = (d->B + ka.blk - 1) / ka.blk;
    const int dev = current_device();
    ka.quad = shape_of(d, dev) == EPI_SHAPE_QUAD ? 1 : 0;
    ka.wave = shape_of(d, dev) == EPI_SHAPE_WAVE ? 1 : 0;
    ka.hex = shape_of(d, dev) == EPI_SHAPE_HEX ? 1 : 0;
    ka.mon_scan = (d->test_flags >> 1) & 1;
    ka.hexw = d->test_window;      // test hook (0 in production): days per addressing window, see epi_batch_desc.test_window
    ka.stor = f32 ? 1 : 0;
    ka.bk_from = d->T - 2; ka.bk_to = 0;
    ka.c0 = 0; ka.cn = d->B;
    ka.mf.lo_is_zero = mi.lo_is_zero; ka.mf.phi_ge = mi.phi_ge; ka.mf.obs_clamp = mi.obs_clamp;
    ka.mf.obs_type = mi.obs_fixed ? EPI_OBS_NEWCASES : d->obs_type;
    ka.x_series = in->x_series; ka.u_series = in->u_series;
    ka.x = in->x; ka.u = in->u; ka.R_series = in->R_series; ka.R_scalar = in->R_scalar; ka.prm = in->prm; ka.Q = in->Q;
    if (!mi.flipped) {
        ka.s_init = in->s_init; ka.Ps_init = in->Ps_init; ka.s_final = in->s_final; ka.Ps_final = in->Ps_final;
    } else {   // Backward*.m:21-24
        ka.s_init = in->s_final; ka.Ps_init = in->Ps_final; ka.s_final = in->s_init; ka.Ps_final = in->Ps_init;
    }
    const bool has_uos = mi.generic;
    ka.S_MINUS = (om & EPI_OUT_S_MINUS) ? out->S_MINUS : (double *)(ws + wl.s_minus);
    ka.S_PLUS = (om & EPI_OUT_S_PLUS) ? out->S_PLUS : (double *)(ws + wl.s_plus);
    ka.P_MINUS = (om & EPI_OUT_P_MINUS) ? out->P_MINUS : (double *)(ws + wl.p_minus);
    ka.P_PLUS = (om & EPI_OUT_P_PLUS) ? out->P_PLUS : (double *)(ws + wl.p_plus);
    ka.ws_upper = ((om & EPI_OUT_P_MINUS) ? 0 : 1) | ((om & EPI_OUT_P_PLUS) ? 0 : 2);
    // fp32 storage, 3 states: the packed smoother recomputes s(k+1|k) from s(k|k) (ekf_sym.hpp, RC), nothing reads an
    // fp64 S_MINUS back -- the forward kernel then stores the caller's fp32 copy only
    if (f32 && mi.m == 3) ka.S_MINUS = nullptr;
    ka.u_opt = sel(EPI_OUT_U_OPT, out->u_opt);
    ka.u_opt_smooth = has_uos ? sel(EPI_OUT_U_OPT_SMOOTH, out->u_opt_smooth) : nullptr;
    ka.S_SMOOTH = sel(EPI_OUT_S_SMOOTH, out->S_SMOOTH);
    ka.P_SMOOTH = sel(EPI_OUT_P_SMOOTH, out->P_SMOOTH);
    ka.K_GAIN = sel(EPI_OUT_K_GAIN, out->K_GAIN);
    ka.innovations = sel(EPI_OUT_INNOVATIONS, out->innovations);
    ka.rho = sel(EPI_OUT_RHO, out->rho);
    ka.f.u_opt = sel32(EPI_OUT_U_OPT, out->u_opt);
    ka.f.u_opt_smooth = has_uos ? sel32(EPI_OUT_U_OPT_SMOOTH, out->u_opt_smooth) : nullptr;
    ka.f.S_MINUS = sel32(EPI_OUT_S_MINUS, out->S_MINUS); ka.f.S_PLUS = sel32(EPI_OUT_S_PLUS, out->S_PLUS);
    ka.f.S_SMOOTH = sel32(EPI_OUT_S_SMOOTH, out->S_SMOOTH);
    ka.f.P_MINUS = sel32(EPI_OUT_P_MINUS, out->P_MINUS); ka.f.P_PLUS = sel32(EPI_OUT_P_PLUS, out->P_PLUS);
    ka.f.P_SMOOTH = sel32(EPI_OUT_P_SMOOTH, out->P_SMOOTH);
    ka.f.K_GAIN = sel32(EPI_OUT_K_GAIN, out->K_GAIN); ka.f.innovations = sel32(EPI_OUT_INNOVATIONS, out->innovations);
    ka.f.rho = sel32(EPI_OUT_RHO, out->rho);
    ka.pinv_rank = out->pinv_rank; ka.status = out->status;
    // exact_nonfinite: 1 = always (runs the smoother to find the chains, whatever is selected); 0, the default = whenever the call
    // runs the smoother anyway; -1 = never
    const bool smooth_sel = ((om | om32) & (EPI_OUT_S_SMOOTH | EPI_OUT_P_SMOOTH)) || (has_uos && ((om | om32) & EPI_OUT_U_OPT_SMOOTH)) ||
                            out->pinv_rank || out->status;
    const bool rerun = mi.generic && (d->exact_nonfinite > 0 || (d->exact_nonfinite == 0 && smooth_sel)) && d->q_mode == 0 && d->phase == 0 && !f32;
    if (mi.generic && d->exact_nonfinite >= 0) {
        ka.only_buf = (int32_t *)(ws + wl.only);
        if (!ka.status && rerun) ka.status = (int32_t *)(ws + wl.status);
    }
    ka.mon_hoist = monitor_hoisted(d) ? 1 : 0;
    if (ka.mon_hoist && !ka.innovations && (ka.rho || ka.f.rho)) ka.innovations = (double *)(ws + wl.innov);
    ka.X = mi.generic ? (double *)(ws + wl.x) : nullptr;
    ka.rankbuf = mi.generic ? (int32_t *)(ws + wl.rank) : nullptr;
    ka.pinv_pos0 = mi.flipped ? 0 : 1;
    ka.dense_flag = mi.generic ? (int *)(ws + wl.flag) : nullptr;
    if (mi.generic) {
        ka.hand_s = (double *)(ws + wl.hand_s); ka.hand_p = (double *)(ws + wl.hand_p); ka.hand_i = (int32_t *)(ws + wl.hand_i);
        ka.hand_pitch = (int)padded_chains(d);
    }
    {
        struct { uint32_t bit; const void *p; const char *n; } chk[] = {
            {EPI_OUT_U_OPT, out->u_opt, "u_opt"}, {EPI_OUT_S_MINUS, out->S_MINUS, "S_MINUS"},
            {EPI_OUT_S_PLUS, out->S_PLUS, "S_PLUS"}, {EPI_OUT_S_SMOOTH, out->S_SMOOTH, "S_SMOOTH"},
            {EPI_OUT_P_MINUS, out->P_MINUS, "P_MINUS"}, {EPI_OUT_P_PLUS, out->P_PLUS, "P_PLUS"},
            {EPI_OUT_P_SMOOTH, out->P_SMOOTH, "P_SMOOTH"}, {EPI_OUT_K_GAIN, out->K_GAIN, "K_GAIN"},
            {EPI_OUT_INNOVATIONS, out->innovations, "innovations"}, {EPI_OUT_RHO, out->rho, "rho"}};
        for (auto &q : chk)
            if (((om | om32) & q.bit) && !q.p) { char b[128]; snprintf(b, sizeof b, "output %s selected but NULL", q.n); set_err(err, b); return EPI_ERR_BAD_ARG; }
        if (has_uos && ((om | om32) & EPI_OUT_U_OPT_SMOOTH) && !out->u_opt_smooth) { set_err(err, "output u_opt_smooth selected but NULL"); return EPI_ERR_BAD_ARG; }
    }
    // (exact_nonfinite needs the smoother's guard to find the chains: it runs the smoother whatever is selected)
    const bool smooth = smooth_sel || rerun;
    hipStream_t st = (hipStream_t)stream;
    Launch L{};
    L.dev = dev; L.phase = d->phase; L.time_pipe = d->time_pipe; L.test_flags = d->test_flags; L.smooth = smooth; L.tail = tail; L.rerun = rerun;
    // a time-varying Q_w is read per step by the dense kernels only
    L.hint = (mi.generic && d->q_mode == 0) ? d->path_hint : 2;
    if (tail && (!ka.u_opt_smooth || d->phase != 0)) { set_err(err, "the sweep's scoring tail needs a full call (phase 0) with fp64 u_opt_smooth selected"); return EPI_ERR_BAD_ARG; }
    hipError_t e;
    switch (d->model) {
    case EPI_MODEL_SIA3: e = launch_chain<3, 0, 1>(ka, L, st); break;
    case EPI_MODEL_SIA6: e = launch_chain<6, 0, 1>(ka, L, st); break;
    case EPI_MODEL_SIA3_BWD: e = launch_chain<3, 1, 1>(ka, L, st); break;
    case EPI_MODEL_SIA6_BWD: e = launch_chain<6, 1, 1>(ka, L, st); break;
    default: e = launch_chain<6, 0, 0>(ka, L, st); break;
    }
    if (e != hipSuccess) return hip_fail(err, e, "kernel launch");
    return EPI_OK;
}

int epi_ekf_run_device(const epi_batch_desc *d, const epi_inputs *in, const epi_outputs *out,
                       void *workspace, size_t workspace_bytes, void *stream, char *err)
{
    return run_device_impl(d, in, out, workspace, workspace_bytes, nullptr, stream, err);
}

static int sweep_args_ok(const epi_batch_desc *d, const epi_sweep_desc *sd, const double *sp, const double *J0_prefix,
                         const double *J1_prefix, double *J0, double *J1, int32_t *on_front, int32_t *i_opt, char *err)
{
    if (!d || !sd || sd->abi_version != EPIEKF_ABI_VERSION) { set_err(err, "bad sweep descriptor"); return EPI_ERR_BAD_ARG; }
    if (d->model != EPI_MODEL_SIA6) { set_err(err, "the cost-weight sweep runs SIAlphaModelEKFOptControlled (EPI_MODEL_SIA6)"); return EPI_ERR_UNSUPPORTED; }
    if (sd->t_hist < 1 || sd->t_hist >= d->T) { set_err(err, "t_hist must leave at least one horizon day: 1 <= t_hist < T"); return EPI_ERR_BAD_ARG; }
    if (!sp || !J0_prefix || !J1_prefix || !J0 || !J1) { set_err(err, "NULL scoring array"); return EPI_ERR_BAD_ARG; }
    if (on_front || i_opt) {
        if (sd->R < 1 || sd->P < 1 || (int64_t)sd->R * sd->P != (int64_t)d->B) { set_err(err, "the Pareto filter needs whole regions: B == R * P"); return EPI_ERR_BAD_ARG; }
        if (sd->P > 8192) { set_err(err, "more than 8192 points per region"); return EPI_ERR_UNSUPPORTED; }
    }
    if (d->storage != 0 || !(d->out_mask & EPI_OUT_U_OPT_SMOOTH)) { set_err(err, "the sweep scores the fp64 u_opt_smooth output: select it (storage 0)"); return EPI_ERR_BAD_ARG; }
    return EPI_OK;
}

int epi_sweep_run_device(const epi_batch_desc *d, const epi_inputs *in, const epi_outputs *out, void *workspace,
                         size_t workspace_bytes, const epi_sweep_desc *sd, const double *sp, const double *J0_prefix,
                         const double *J1_prefix, double *J0, double *J1, int32_t *on_front, int32_t *i_opt, void *stream,
                         char *err)
{
    int rc = sweep_args_ok(d, sd, sp, J0_prefix, J1_prefix, J0, J1, on_front, i_opt, err);
    if (rc != EPI_OK) return rc;
    Tail t{};
    t.t_hist = sd->t_hist; t.R = sd->R; t.P = sd->P;
    t.sp = sp; t.J0_prefix = J0_prefix; t.J1_prefix = J1_prefix; t.J0 = J0; t.J1 = J1; t.on_front = on_front; t.i_opt = i_opt;
    if ((on_front || i_opt) && (size_t)2 * sd->P * sizeof(double) > 64u * 1024u) {
        hipError_t e = hipFuncSetAttribute((const void *)pareto_front, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)2 * sd->P * sizeof(double)));
        if (e != hipSuccess) return hip_fail(err, e, "hipFuncSetAttribute");
    }
    epi_batch_desc full = *d;
    full.phase = 0;
    return run_device_impl(&full, in, out, workspace, workspace_bytes, &t, stream, err);
}

// (forward + monitor, pinv grid, smoother) milliseconds of this call on THESE arrays, enqueued stage by stage (phases 1, 3, 4)
// between HIP events on `stream`: the mean over as many rounds as fill `min_ms` of device time (at least one), after one untimed
// round.  Synchronous.  What a device-pointer caller compares allocations with ("Placement"): the same arrays give the
// same times run after run, another allocation of the same arrays may be 5-15 % off.
int epi_ekf_time_stages_device(const epi_batch_desc *d, const epi_inputs *in, const epi_outputs *out, void *workspace,
                               size_t workspace_bytes, void *stream, double min_ms, double *ms, char *err)
{
    if (!d || !ms) { set_err(err, "NULL descriptor / result array"); return EPI_ERR_BAD_ARG; }
    epi_batch_desc s = *d;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipEventCreate(&ev[i]);
    auto done = [&](int rc) { for (auto &x : ev) if (x) (void)hipEventDestroy(x); return rc; };
    if (e != hipSuccess) return done(hip_fail(err, e, "hipEventCreate"));
    const int phases[3] = {1, 3, 4};
    double acc[3] = {0.0, 0.0, 0.0}, total = 0.0;
    int rounds = 0;
    for (int r = -1; r < 4096 && (r < 1 || total < min_ms); r++) {       // r = -1: the untimed round
        (void)hipEventRecord(ev[0], st);
        for (int q = 0; q < 3; q++) {
            s.phase = phases[q];
            const int rc = epi_ekf_run_device(&s, in, out, workspace, workspace_bytes, stream, err);
            if (rc != EPI_OK) { (void)hipStreamSynchronize(st); return done(rc); }
            (void)hipEventRecord(ev[q + 1], st);
        }
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return done(hip_fail(err, e, "kernel execution"));
        if (r < 0) continue;
        for (int q = 0; q < 3; q++) {
            float t = 0.0f;
            (void)hipEventElapsedTime(&t, ev[q], ev[q + 1]);
            acc[q] += t; total += t;
        }
        rounds++;
    }
    for (int q = 0; q < 3; q++) ms[q] = acc[q] / (double)rounds;
    return done(EPI_OK);
}

int epi_ekf_precheck_device(const epi_batch_desc *d, const epi_inputs *in, void *stream, int *fast_ok, char *err)
{
    int rc = epi_ekf_validate(d, err);
    if (rc != EPI_OK) return rc;
    if (!in || !fast_ok || !in->Ps_init || !in->Ps_final || !in->Q || !in->s_init || !in->s_final) { set_err(err, "NULL argument"); return EPI_ERR_BAD_ARG; }
    const ModelInfo &mi = MODEL_TABLE[d->model];
    *fast_ok = 0;
    if (!mi.generic || d->q_mode != 0) return EPI_OK;   // NewCase models never symmetrise; Q(:,:,k): dense kernels only
    KArgs ka{};
    ka.B = d->B;
    ka.Ps_init = mi.flipped ? in->Ps_final : in->Ps_init;
    ka.Ps_final = mi.flipped ? in->Ps_init : in->Ps_final;
    ka.s_init = mi.flipped ? in->s_final : in->s_init;
    ka.Q = in->Q;
    int *flag = nullptr;
    hipError_t e = hipMalloc((void **)&flag, sizeof(int));
    if (e != hipSuccess) return hip_fail(err, e, "hipMalloc");
    hipStream_t st = (hipStream_t)stream;
    int host_flag = 1;
    if ((e = hipMemsetAsync(flag, 0, sizeof(int), st)) == hipSuccess) {
        if (mi.m == 3) hipLaunchKernelGGL((ekf_precheck<3>), dim3((d->B + 255) / 256), dim3(256), 0, st, ka, flag, 0);
        else hipLaunchKernelGGL((ekf_precheck<6>), dim3((d->B + 255) / 256), dim3(256), 0, st, ka, flag, 0);
        if ((e = hipGetLastError()) == hipSuccess)
            e = hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFree(flag);
    if (e != hipSuccess) return hip_fail(err, e, "precheck");
    *fast_ok = host_flag ? 0 : 1;
    return EPI_OK;
}

// ---------------------------------------------------------------------------
// host-pointer entry points: pooled contexts
// ---------------------------------------------------------------------------
// A context owns what a host call needs on one device -- a stream, a device arena and a pinned staging buffer -- and
// outlives the call (idle contexts wait in a per-device pool), so that the 250 calls per region of the unchanged
// reference caller (TrainPredictPrescribeNPI.m:421-460) do not pay 50 hipMalloc/hipFree and 50 synchronous copies each.
}   // extern "C"
namespace epi {
constexpr size_t kStageBytes = (size_t)64 << 20;
constexpr size_t kStageSmallBytes = (size_t)8 << 20;   // dense calls above this go straight to / from the caller's arrays (HostIO::staged)
constexpr size_t kArenaKeepBytes = (size_t)2 << 30;   // an idle context keeps at most this much device memory
constexpr int kPoolPerDevice = 4;                      // idle contexts kept per device (64 MiB of pinned memory each)
struct HostCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    char *arena = nullptr; size_t arena_bytes = 0;
    bool tuned = false;        // the arena is the fastest of several candidates (place_and_run): kept whatever its size
    char *pinned = nullptr;
    ~HostCtx()
    {
        if (device < 0) return;
        (void)hipSetDevice(device);
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        if (arena) (void)hipFree(arena);
        if (pinned) (void)hipHostFree(pinned);
    }
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= arena_bytes) return hipSuccess;
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        if (arena) { (void)hipFree(arena); arena = nullptr; arena_bytes = 0; tuned = false; }
        size_t want = bytes + bytes / 4;
        e = hipMalloc((void **)&arena, want);
        if (e != hipSuccess) { (void)hipGetLastError(); want = bytes; e = hipMalloc((void **)&arena, want); }
        if (e != hipSuccess) { arena = nullptr; return e; }
        arena_bytes = want;
        return hipSuccess;
    }
};
static std::mutex g_pool_mu;
static std::vector<HostCtx *> g_pool;     // idle contexts of all devices
static HostCtx *ctx_acquire(int device, hipError_t *e)
{
    if (device < 0 || device >= kMaxDevices) { *e = hipErrorInvalidDevice; return nullptr; }
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t pick = g_pool.size();               // the idle context of this device with the largest arena
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i]->device == device && (pick == g_pool.size() || g_pool[i]->arena_bytes > g_pool[pick]->arena_bytes)) pick = i;
        if (pick < g_pool.size()) { HostCtx *c = g_pool[pick]; g_pool.erase(g_pool.begin() + (long)pick); *e = hipSetDevice(device); return c; }
    }
    if ((*e = hipSetDevice(device)) != hipSuccess) return nullptr;
    HostCtx *c = new HostCtx();
    c->device = device;
    if ((*e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { c->device = -1; delete c; return nullptr; }
    if ((*e = hipHostMalloc((void **)&c->pinned, kStageBytes, hipHostMallocDefault)) != hipSuccess) { delete c; return nullptr; }
    return c;
}
// back to the pool: a device keeps at most kPoolPerDevice idle contexts, and only ONE of them an arena above kArenaKeepBytes (the
// others hand theirs back to the device first; epi_host_pool_release frees everything).  Until round 6 every large arena was
// returned at once -- but hipFree takes ~30 ms per GiB on these boxes (24 GiB: 730 ms; hipMalloc 0.4 ms) and the next hipMalloc
// sometimes waits behind it for seconds: the headline sweep's host-pointer call took 16 ms or 0.7-6 s, the 9 375-chain shard with all
// outputs 118 or 200 ms, depending on whether the previous call's arena was still being returned (profiles/r06/host_calls.json).
static void ctx_release(HostCtx *c)
{
    bool big_kept = false;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (HostCtx *o : g_pool) big_kept = big_kept || (o->device == c->device && o->arena_bytes > kArenaKeepBytes);
    }
    if (c->arena_bytes > kArenaKeepBytes && !c->tuned && big_kept) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(c->arena);
        c->arena = nullptr; c->arena_bytes = 0;
    }
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        int same = 0;
        for (HostCtx *o : g_pool) same += o->device == c->device;
        if (same < kPoolPerDevice) { g_pool.push_back(c); return; }
    }
    delete c;
}

// Placement of a host call's arena (epi_batch_desc.placement_tries / epi_prescribe_desc.placement_tries, ABI 6).  Where the
// allocator puts the ~14 arrays a pass streams concurrently changes the forward kernel's and the smoother's time by 5-15 %
// (which PHYSICAL pages the allocation got: DESIGN.md 4, "Placement"); it is a property of the allocation and a host-pointer
// caller never sees the allocation.  When a call has to allocate a NEW arena and asks for `tries` > 1: the call's own kernels
// are run once untimed (clocks, code objects), then timed on up to `tries` candidate arenas, each allocated while the earlier
// ones are held (so that other memory is handed out) and each for at least ~15 ms of kernels; the fastest is kept -- with the
// complete results of its last run in it, nothing is computed again -- and stays with the pooled context whatever its size
// (epi_host_pool_release frees it).  compute(base, ev0, ev1) enqueues upload + kernels for the arena at `base` on the
// context's stream and records the two events (when given) around the kernels.
template <class F>
static int place_and_run(HostCtx *cx, size_t need, int tries, epi_placement_report *rep, F &&compute, char *err)
{
    if (rep) memset(rep, 0, sizeof *rep);
    const bool fresh = need > cx->arena_bytes;
    hipError_t e = cx->reserve(need);
    if (e != hipSuccess) return hip_fail(err, e, "device arena");
    if (tries <= 1 || !fresh) return compute(cx->arena, nullptr, nullptr);
    if (tries > EPI_PLACEMENT_MAX_TRIES) tries = EPI_PLACEMENT_MAX_TRIES;
    int rc = compute(cx->arena, nullptr, nullptr);
    if (rc != EPI_OK) return rc;
    if ((e = hipStreamSynchronize(cx->stream)) != hipSuccess) return hip_fail(err, e, "kernel execution (placement warm-up)");
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if ((e = hipEventCreate(&ev0)) != hipSuccess || (e = hipEventCreate(&ev1)) != hipSuccess) {
        if (ev0) (void)hipEventDestroy(ev0);
        return hip_fail(err, e, "hipEventCreate");
    }
    struct Cand { char *p; size_t bytes; float ms; };
    std::vector<Cand> cands;
    for (int i = 0; i < tries && rc == EPI_OK; i++) {
        Cand c{cx->arena, cx->arena_bytes, 0.0f};
        if (i > 0) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need + need / 16) { (void)hipGetLastError(); break; }
            if (hipMalloc((void **)&c.p, need) != hipSuccess) { (void)hipGetLastError(); break; }
            c.bytes = need;
        }
        float acc = 0.0f;
        int n = 0;
        do {
            rc = compute(c.p, ev0, ev1);
            if (rc != EPI_OK) break;
            if ((e = hipStreamSynchronize(cx->stream)) != hipSuccess) { rc = hip_fail(err, e, "kernel execution (placement try)"); break; }
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, ev0, ev1);
            acc += ms; n++;
        } while (acc < 15.0f && n < 64);
        c.ms = n ? acc / (float)n : 0.0f;
        cands.push_back(c);
    }
    (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
    size_t best = 0;
    for (size_t i = 1; i < cands.size(); i++)
        if (rc == EPI_OK && cands[i].ms < cands[best].ms) best = i;
    if (rc != EPI_OK) best = 0;                       // an error: back to the first arena, the others are freed
    (void)hipStreamSynchronize(cx->stream);
    for (size_t i = 0; i < cands.size(); i++)
        if (i != best) (void)hipFree(cands[i].p);
    if (!cands.empty()) { cx->arena = cands[best].p; cx->arena_bytes = cands[best].bytes; }
    cx->tuned = rc == EPI_OK && cands.size() > 1;
    if (rep && rc == EPI_OK) {
        rep->tries = (int32_t)cands.size(); rep->chosen = (int32_t)best;
        for (size_t i = 0; i < cands.size(); i++) rep->ms[i] = cands[i].ms;
    }
    return rc;
}

// The arrays of one host call on one context: every array is `rows` rows of which this call moves a strided piece
// (columns [col0, col0 + cols) of a row of cols_full elements) to / from a contiguous device copy.  Calls whose arrays fit
// the pinned buffer are packed there and moved by ONE copy each way; larger ones go row block by row block.
struct HostIO {
    struct Piece { const char *src; char *dst; size_t rows, width, pitch, off; };
    std::vector<Piece> ins, outs;
    size_t off = 0, in_bytes = 0;
    size_t align = 256;
    size_t add_in(const void *host, size_t rows, size_t elem, size_t cols_full, size_t col0, size_t cols)
    {
        ins.push_back(Piece{host ? (const char *)host + col0 * elem : nullptr, nullptr, rows, cols * elem, cols_full * elem, off});
        const size_t o = off;
        off += (rows * cols * elem + align - 1) / align * align;
        in_bytes = off;
        return o;
    }
    size_t add_out(void *host, size_t rows, size_t elem, size_t cols_full, size_t col0, size_t cols)
    {
        // host == NULL: the device copy exists (kernels write it) but nothing is copied back
        outs.push_back(Piece{nullptr, host ? (char *)host + col0 * elem : nullptr, rows, cols * elem, cols_full * elem, off});
        const size_t o = off;
        off += (rows * cols * elem + align - 1) / align * align;
        return o;
    }
    size_t reserve(size_t bytes)       // device-only scratch inside the same arena
    {
        off = (off + 255) & ~(size_t)255;
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    }
    bool inputs_present() const
    {
        for (auto &p : ins) if (!p.src) return false;
        return true;
    }
    // (outputs must all have been added before the first reserve() for the staged download to be one copy; the code
    // below copies [in_bytes, out_end) where out_end is the end of the last output piece)
    size_t out_end() const { return outs.empty() ? in_bytes : outs.back().off + outs.back().rows * outs.back().width; }
    // A piece that covers whole rows (a call over all chains of the caller's arrays) is one contiguous range on both sides.
    static bool dense(const Piece &p) { return p.width == p.pitch || p.rows <= 1; }
    bool all_dense() const
    {
        for (auto &p : ins) if (!dense(p)) return false;
        for (auto &p : outs) if (p.dst && !dense(p)) return false;
        return true;
    }
    // Through the pinned buffer (ONE copy each way + the host's memcpy per row) or straight between the caller's arrays and
    // the device?  Measured on the pool's boxes (profiles/pcie_probe): a copy from / to pageable memory runs at 13 GB/s for 1
    // MiB and 54-56 GB/s from 16 MiB on, the pinned buffer at 37 / 55-57 GB/s, the host's memcpy out of it at 25 GB/s beyond
    // the caches -- so small calls (the reference's one-chain call: 0.66 MB in 11 arrays) are packed, large dense ones are
    // not, and strided pieces (a chain block of a multi-device call) are packed while they fit.
    bool staged() const
    {
        const size_t end = out_end();
        if (end > kStageBytes || in_bytes > kStageBytes) return false;
        return end <= kStageSmallBytes || !all_dense();
    }
    static hipError_t move(char *dev, const Piece &p, bool to_device, hipStream_t st)
    {
        if (dense(p))
            return to_device ? hipMemcpyAsync(dev, p.src, p.rows * p.width, hipMemcpyHostToDevice, st)
                             : hipMemcpyAsync(p.dst, dev, p.rows * p.width, hipMemcpyDeviceToHost, st);
        return to_device ? hipMemcpy2DAsync(dev, p.width, p.src, p.pitch, p.width, p.rows, hipMemcpyHostToDevice, st)
                         : hipMemcpy2DAsync(p.dst, p.pitch, dev, p.width, p.width, p.rows, hipMemcpyDeviceToHost, st);
    }
    hipError_t upload(HostCtx *cx, char *base) const
    {
        if (staged()) {
            for (auto &p : ins)
                for (size_t r = 0; r < p.rows; r++) memcpy(cx->pinned + p.off + r * p.width, p.src + r * p.pitch, p.width);
            return in_bytes ? hipMemcpyAsync(base, cx->pinned, in_bytes, hipMemcpyHostToDevice, cx->stream) : hipSuccess;
        }
        for (auto &p : ins) {
            const hipError_t e = move(base + p.off, p, true, cx->stream);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    // enqueues the copies back, waits for the stream, and (staged) scatters the rows into the caller's arrays
    hipError_t download(HostCtx *cx, const char *base) const
    {
        hipError_t e;
        if (staged()) {
            const size_t end = out_end();
            if (end > in_bytes && (e = hipMemcpyAsync(cx->pinned + in_bytes, base + in_bytes, end - in_bytes, hipMemcpyDeviceToHost, cx->stream)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(cx->stream)) != hipSuccess) return e;
            for (auto &p : outs)
                if (p.dst)
                    for (size_t r = 0; r < p.rows; r++) memcpy(p.dst + r * p.pitch, cx->pinned + p.off + r * p.width, p.width);
            return hipSuccess;
        }
        // Large pieces into memory the caller has never touched (a MEX gateway's freshly created outputs, np.empty) would be
        // faulted in page by page under the copy, by ONE thread inside the driver's pinning call: 14-17 GB/s instead of the
        // 33-52 GB/s resident pages reach (profiles/r06/host_calls.json).  A helper thread therefore populates the destination
        // of piece k + 1 on several threads (populate_pages) while piece k is on the wire; the copy of a piece is issued when
        // its pages are there.  Resident pages cost a page-table walk.
        std::vector<const Piece *> todo;
        size_t big = 0;
        for (auto &p : outs)
            if (p.dst) { todo.push_back(&p); if (span_bytes(p) >= kPopulateMinBytes) big++; }
        if (big == 0) {
            for (const Piece *p : todo)
                if ((e = move((char *)base + p->off, *p, false, cx->stream)) != hipSuccess) return e;
            return hipStreamSynchronize(cx->stream);
        }
        std::mutex mu;
        std::condition_variable cv;
        size_t ready = 0;                      // pieces [0, ready) are populated
        std::thread helper([&] {
            for (size_t k = 0; k < todo.size(); k++) {
                if (span_bytes(*todo[k]) >= kPopulateMinBytes) populate_pages(todo[k]->dst, span_bytes(*todo[k]), dense(*todo[k]));
                { std::lock_guard<std::mutex> lk(mu); ready = k + 1; }
                cv.notify_one();
            }
        });
        e = hipSuccess;
        for (size_t k = 0; k < todo.size() && e == hipSuccess; k++) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return ready > k; }); }
            e = move((char *)base + todo[k]->off, *todo[k], false, cx->stream);
        }
        helper.join();
        if (e != hipSuccess) return e;
        return hipStreamSynchronize(cx->stream);
    }
    // bytes of the caller's array a piece spans (a strided piece: first row's start to last row's end, gaps included -- they
    // belong to the same array)
    static size_t span_bytes(const Piece &p) { return p.rows ? (p.rows - 1) * p.pitch + p.width : 0; }
    static constexpr size_t kPopulateMinBytes = (size_t)8 << 20;
    // Make [p, p + bytes) resident and writable WITHOUT changing its contents: madvise(MADV_POPULATE_WRITE) per slice on up to
    // eight threads (page-table population scales with threads; one thread zeroes fresh pages at ~10 GB/s), falling back to
    // writing a byte of every page back to itself where the kernel does not know the advice (< 5.14; dense pieces only).
    static void populate_pages(char *p, size_t bytes, bool whole)
    {
        const size_t page = 4096;
        const uintptr_t a0 = (uintptr_t)p & ~(uintptr_t)(page - 1), a1 = ((uintptr_t)p + bytes + page - 1) & ~(uintptr_t)(page - 1);
        const size_t pages = (a1 - a0) / page;
        unsigned hw = std::thread::hardware_concurrency();
        size_t nt = bytes / ((size_t)16 << 20) + 1;
        const size_t cap = hw >= 16 ? 12 : (hw >= 4 ? hw / 2 : 1);
        if (nt > cap) nt = cap;
        // huge pages where the system grants them on request: 512 times fewer faults, the population is then bound by zeroing
        if (a1 - a0 >= ((size_t)4 << 20)) (void)madvise((void *)a0, a1 - a0, MADV_HUGEPAGE);
        auto slice = [=](size_t i) {
            const uintptr_t b = a0 + pages * i / nt * page, e = a0 + pages * (i + 1) / nt * page;
            if (e <= b) return;
#ifdef MADV_POPULATE_WRITE
            if (madvise((void *)b, e - b, MADV_POPULATE_WRITE) == 0) return;
#else
            if (madvise((void *)b, e - b, 23) == 0) return;
#endif
            // (a strided piece's gaps are other blocks' columns, which another device's copy may be writing right now: no
            // write-back there.)  The pages at the two ends may hold bytes outside [p, p + bytes): the byte touched is inside
            if (!whole) return;
            for (uintptr_t q = b; q < e; q += page) {
                uintptr_t t = q < (uintptr_t)p ? (uintptr_t)p : q;
                if (t >= (uintptr_t)p + bytes) break;
                volatile char *c = (volatile char *)t;
                *c = *c;
            }
        };
        std::vector<std::thread> th;
        for (size_t i = 1; i < nt; i++) th.emplace_back(slice, i);
        slice(0);
        for (auto &t : th) t.join();
    }
};

// ekf_precheck's rules on HOST arrays ([rows][pitch] doubles, columns [lo, lo + n)): may these chains take the packed
// kernels?  Ps_init and Ps_final bit-wise symmetric (values and NaN pattern), Q_w diagonal, s_init / Ps_init / diag(Q_w)
// finite.  The host entry points have the arrays in host memory, so the check costs no device round trip.
static bool host_precheck(int m, bool flipped, const epi_inputs *in, size_t pitch, size_t lo, size_t n)
{
    const double *Pi = flipped ? in->Ps_final : in->Ps_init, *Pf = flipped ? in->Ps_init : in->Ps_final;
    const double *si = flipped ? in->s_final : in->s_init, *Q = in->Q;
    auto nonfinite = [](double v) { return !(v - v == 0.0); };
    auto same = [](double a, double b) { return a == b || (a != a && b != b); };
    for (int j = 0; j < m; j++)
        for (int i = 0; i <= j; i++) {
            const double *pu = Pi + (size_t)(i + m * j) * pitch + lo, *pl = Pi + (size_t)(j + m * i) * pitch + lo;
            const double *fu = Pf + (size_t)(i + m * j) * pitch + lo, *fl = Pf + (size_t)(j + m * i) * pitch + lo;
            const double *qu = Q + (size_t)(i + m * j) * pitch + lo, *ql = Q + (size_t)(j + m * i) * pitch + lo;
            for (size_t c = 0; c < n; c++) {
                if (nonfinite(pu[c])) return false;
                if (i == j) { if (nonfinite(qu[c])) return false; continue; }
                if (!same(pu[c], pl[c]) || !same(fu[c], fl[c]) || !(qu[c] == 0.0) || !(ql[c] == 0.0)) return false;
            }
        }
    for (int i = 0; i < m; i++)
        for (size_t c = 0; c < n; c++)
            if (nonfinite(si[(size_t)i * pitch + lo + c])) return false;
    return true;
}

// chains [lo, lo + n) of a host call on one context.  Per-chain arrays are [rows][B] in the caller's memory: the block is a
// strided piece of each row.
static int run_host_block(HostCtx *cx, const epi_batch_desc *d0, const epi_inputs *in, const epi_outputs *out, int lo, int n, char *err)
{
    if (n <= 0) return EPI_OK;
    epi_batch_desc d = *d0;
    const size_t Bfull = (size_t)d0->B;
    d.B = n;
    // A host call returns what the dense evaluation returns, overflowed chains included (exact_nonfinite) -- but the call ends
    // with a synchronising download anyway, so the per-chain status words come back with it and the dense second pass is
    // enqueued only when one of them has bit 0 set: the common call pays nothing for it (the device-side variant costs five
    // launches that return at once, ~40 us of a 1.5 ms one-chain call).
    d.exact_nonfinite = -1;
    const bool id_x = !in->x_series, id_u = !in->u_series;      // identity series: one series per chain, sliced with the chains
    if (id_x) d.Sx = n;
    if (id_u) d.Su = n;
    const ModelInfo &mi = MODEL_TABLE[d.model];
    const int m = mi.m, mm = m * m;
    const size_t T = (size_t)d.T;
    HostIO io;
    epi_inputs din{};
    epi_outputs dout{};
    // inputs
    const size_t o_xs = in->x_series ? io.add_in(in->x_series, 1, 4, Bfull, lo, n) : 0;
    const size_t o_us = in->u_series ? io.add_in(in->u_series, 1, 4, Bfull, lo, n) : 0;
    const size_t o_x = id_x ? io.add_in(in->x, T, 8, Bfull, lo, n) : io.add_in(in->x, T, 8, d0->Sx, 0, d0->Sx);
    const size_t o_u = id_u ? io.add_in(in->u, T * d.n_npi, 8, Bfull, lo, n) : io.add_in(in->u, T * d.n_npi, 8, d0->Su, 0, d0->Su);
    size_t o_rs = 0, o_rc = 0;
    if (d.r_mode == 1) o_rs = id_x ? io.add_in(in->R_series, T, 8, Bfull, lo, n) : io.add_in(in->R_series, T, 8, d0->Sx, 0, d0->Sx);
    else o_rc = io.add_in(in->R_scalar, 1, 8, Bfull, lo, n);
    const size_t o_prm = io.add_in(in->prm, EPI_PRM_COUNT, 8, Bfull, lo, n);
    const size_t o_si = io.add_in(in->s_init, m, 8, Bfull, lo, n), o_pi = io.add_in(in->Ps_init, mm, 8, Bfull, lo, n);
    const size_t o_sf = io.add_in(in->s_final, m, 8, Bfull, lo, n), o_pf = io.add_in(in->Ps_final, mm, 8, Bfull, lo, n);
    const size_t o_q = io.add_in(in->Q, (d.q_mode ? T : (size_t)1) * mm, 8, Bfull, lo, n);
    if (!io.inputs_present()) { set_err(err, "NULL input array"); return EPI_ERR_BAD_ARG; }
    // the caller's arrays are in host memory: decide here which kernels run (what epi_ekf_precheck_device answers for
    // device arrays), so that a host call enqueues one variant only and gets the overlapped launch
    if (d.path_hint == 0 && mi.generic && d.q_mode == 0)
        d.path_hint = host_precheck(m, mi.flipped != 0, in, Bfull, (size_t)lo, (size_t)n) ? 1 : 2;
    // outputs
    struct O { uint32_t bit; double *host; double **dev; size_t rows; };
    const size_t rU = T * d.n_npi, rS = T * m, rP = T * mm, r1 = T;
    O olist[] = {{EPI_OUT_U_OPT, out->u_opt, &dout.u_opt, rU}, {EPI_OUT_U_OPT_SMOOTH, out->u_opt_smooth, &dout.u_opt_smooth, rU},
                 {EPI_OUT_S_MINUS, out->S_MINUS, &dout.S_MINUS, rS}, {EPI_OUT_S_PLUS, out->S_PLUS, &dout.S_PLUS, rS},
                 {EPI_OUT_S_SMOOTH, out->S_SMOOTH, &dout.S_SMOOTH, rS}, {EPI_OUT_P_MINUS, out->P_MINUS, &dout.P_MINUS, rP},
                 {EPI_OUT_P_PLUS, out->P_PLUS, &dout.P_PLUS, rP}, {EPI_OUT_P_SMOOTH, out->P_SMOOTH, &dout.P_SMOOTH, rP},
                 {EPI_OUT_K_GAIN, out->K_GAIN, &dout.K_GAIN, rS}, {EPI_OUT_INNOVATIONS, out->innovations, &dout.innovations, r1},
                 {EPI_OUT_RHO, out->rho, &dout.rho, r1}};
    std::vector<size_t> o_out;
    for (auto &o : olist)
        o_out.push_back(((d.out_mask & o.bit) && o.host) ? io.add_out(o.host, o.rows, 8, Bfull, lo, n) : (size_t)-1);
    const size_t o_rank = out->pinv_rank ? io.add_out(out->pinv_rank, T, 4, Bfull, lo, n) : (size_t)-1;
    // does this call run the smoother (which is where an overflowed chain is recognised)?
    const uint32_t smooth_bits = EPI_OUT_S_SMOOTH | EPI_OUT_P_SMOOTH | (mi.generic ? (uint32_t)EPI_OUT_U_OPT_SMOOTH : 0u);
    bool smooths = out->pinv_rank != nullptr || out->status != nullptr;
    for (auto &o : olist) smooths = smooths || ((d.out_mask & o.bit & smooth_bits) && o.host);
    const bool watch = smooths && mi.generic && d.q_mode == 0 && d.storage == 0 && d.phase == 0;
    std::vector<int32_t> own_status;
    int32_t *hstat = out->status ? out->status + lo : nullptr;
    size_t o_stat = (size_t)-1;
    if (out->status) o_stat = io.add_out(out->status, 1, 4, Bfull, lo, n);
    else if (watch) { own_status.assign((size_t)n, 0); hstat = own_status.data(); o_stat = io.add_out(own_status.data(), 1, 4, (size_t)n, 0, n); }
    epi_batch_desc dmax = d;
    dmax.exact_nonfinite = watch ? 1 : -1;           // the workspace is sized for the second pass
    const size_t wsb = epi_ekf_workspace_bytes(&dmax);
    const size_t o_ws = io.reserve(wsb);
    hipError_t e = hipSuccess;
    // the device-side view of the call for the arena at `base`
    auto bind = [&](char *base) {
        din.x_series = in->x_series ? (const int32_t *)(base + o_xs) : nullptr;
        din.u_series = in->u_series ? (const int32_t *)(base + o_us) : nullptr;
        din.x = (const double *)(base + o_x); din.u = (const double *)(base + o_u);
        din.R_series = d.r_mode == 1 ? (const double *)(base + o_rs) : nullptr;
        din.R_scalar = d.r_mode == 1 ? nullptr : (const double *)(base + o_rc);
        din.prm = (const double *)(base + o_prm);
        din.s_init = (const double *)(base + o_si); din.Ps_init = (const double *)(base + o_pi);
        din.s_final = (const double *)(base + o_sf); din.Ps_final = (const double *)(base + o_pf);
        din.Q = (const double *)(base + o_q);
        size_t k = 0;
        for (auto &o : olist) { if (o_out[k] != (size_t)-1) *o.dev = (double *)(base + o_out[k]); k++; }
        if (o_rank != (size_t)-1) dout.pinv_rank = (int32_t *)(base + o_rank);
        if (o_stat != (size_t)-1) dout.status = (int32_t *)(base + o_stat);
    };
    // upload + kernels (place_and_run may call this for several candidate arenas; ev0 / ev1 bracket the kernels).  From the
    // upload on, copies that read the caller's arrays / the pinned buffer may be in flight: every error return waits for the
    // stream first, so that neither is touched after the call has returned
    auto compute = [&](char *base, hipEvent_t ev0, hipEvent_t ev1) -> int {
        bind(base);
        if ((e = io.upload(cx, base)) != hipSuccess) { (void)hipStreamSynchronize(cx->stream); return hip_fail(err, e, "upload"); }
        if (ev0) (void)hipEventRecord(ev0, cx->stream);
        const int rc = epi_ekf_run_device(&d, &din, &dout, wsb ? base + o_ws : nullptr, wsb, cx->stream, err);
        if (rc != EPI_OK) { (void)hipStreamSynchronize(cx->stream); return rc; }
        if (ev1) (void)hipEventRecord(ev1, cx->stream);
        return EPI_OK;
    };
    {
        const int rc = place_and_run(cx, io.off + 256, d0->placement_tries, lo == 0 ? out->placement : nullptr, compute, err);
        if (rc != EPI_OK) return rc;
    }
    char *base = cx->arena;
    bind(base);
    if ((e = io.download(cx, base)) != hipSuccess) return hip_fail(err, e, "kernel execution / download");
    if (watch) {
        bool any = false;
        for (int c = 0; c < n; c++) any = any || (hstat[c] & 1);
        if (any) {                                    // rare: a covariance overflowed -- the dense second pass, then the outputs again
            const int rc2 = epi_ekf_run_device(&dmax, &din, &dout, wsb ? base + o_ws : nullptr, wsb, cx->stream, err);
            if (rc2 != EPI_OK) { (void)hipStreamSynchronize(cx->stream); return rc2; }
            if ((e = io.download(cx, base)) != hipSuccess) return hip_fail(err, e, "kernel execution / download (dense second pass)");
        }
    }
    return EPI_OK;
}
static int host_args_ok(const epi_batch_desc *d, const epi_inputs *in, const epi_outputs *out, char *err)
{
    int rc = epi_ekf_validate(d, err);
    if (rc != EPI_OK) return rc;
    if (!in || !out) { set_err(err, "NULL inputs/outputs"); return EPI_ERR_BAD_ARG; }
    if (lane_block_of(d) != d->B) { set_err(err, "the host entry points take the classic layout only (lane_block = 0)"); return EPI_ERR_UNSUPPORTED; }
    if (d->storage != 0) { set_err(err, "the host entry points return fp64 arrays (storage = 0)"); return EPI_ERR_UNSUPPORTED; }
    if (!in->x_series && d->Sx != d->B) { set_err(err, "identity x_series needs Sx == B"); return EPI_ERR_BAD_ARG; }
    if (!in->u_series && d->Su != d->B) { set_err(err, "identity u_series needs Su == B"); return EPI_ERR_BAD_ARG; }
    return EPI_OK;
}

// One worker thread per device for the *_multi entry points (created on first use, parked on a condition variable between
// calls, joined by epi_host_pool_release): a call hands every block of chains / regions to the worker of its device and
// waits; blocks that name the same device run one after the other.
struct DevWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false;
    void loop()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                job = std::move(q.front());
                q.pop_front();
            }
            job();
        }
    }
};
static std::mutex g_worker_mu;
static DevWorker *g_workers[kMaxDevices];
static void worker_submit(int device, std::function<void()> job)
{
    DevWorker *w;
    {
        std::lock_guard<std::mutex> lk(g_worker_mu);
        if (!g_workers[device]) {
            g_workers[device] = new DevWorker();
            g_workers[device]->th = std::thread([p = g_workers[device]] { p->loop(); });
        }
        w = g_workers[device];
    }
    { std::lock_guard<std::mutex> lk(w->mu); w->q.push_back(std::move(job)); }
    w->cv.notify_one();
}
static void workers_release_all()
{
    for (int dev = 0; dev < kMaxDevices; dev++) {
        DevWorker *w;
        { std::lock_guard<std::mutex> lk(g_worker_mu); w = g_workers[dev]; g_workers[dev] = nullptr; }
        if (!w) continue;
        { std::lock_guard<std::mutex> lk(w->mu); w->stop = true; }
        w->cv.notify_one();
        w->th.join();
        delete w;
    }
}
// runs job(r) for r = 0 .. n-1 on the workers of devices dev_of(r) and waits for all of them
static void run_on_devices(int n, const std::function<int(int)> &dev_of, const std::function<void(int)> &job)
{
    if (n == 1) {                            // one block: the calling thread does it, and keeps its current device
        int prev = 0;
        const bool have_prev = hipGetDevice(&prev) == hipSuccess;
        job(0);
        if (have_prev) (void)hipSetDevice(prev);
        return;
    }
    std::mutex mu;
    std::condition_variable cv;
    int left = n;
    for (int r = 0; r < n; r++)
        worker_submit(dev_of(r), [&, r] {
            job(r);
            std::lock_guard<std::mutex> lk(mu);
            if (--left == 0) cv.notify_one();
        });
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return left == 0; });
}
}   // namespace epi
extern "C" {

int epi_ekf_run_host(const epi_batch_desc *d, const epi_inputs *in, const epi_outputs *out, int device, char *err)
{
    int rc = host_args_ok(d, in, out, err);
    if (rc != EPI_OK) return rc;
    hipError_t e = hipSuccess;
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;       // the calling thread keeps its current device
    HostCtx *cx = ctx_acquire(device, &e);
    if (!cx || e != hipSuccess) {
        if (cx) ctx_release(cx);
        if (have_prev) (void)hipSetDevice(prev);
        return hip_fail(err, e, "hipSetDevice / context");
    }
    rc = run_host_block(cx, d, in, out, 0, d->B, err);
    ctx_release(cx);
    if (have_prev) (void)hipSetDevice(prev);
    return rc;
}

static int multi_devices_ok(int n_devices, const int *device_ids, char *err)
{
    if (n_devices < 1 || n_devices > kMaxDevices) { set_err(err, "n_devices must be 1..64"); return EPI_ERR_BAD_ARG; }
    for (int r = 0; r < n_devices; r++) {
        const int dev = device_ids ? device_ids[r] : r;
        if (dev < 0 || dev >= kMaxDevices) { set_err(err, "device id out of range 0..63"); return EPI_ERR_BAD_ARG; }
    }
    return EPI_OK;
}

int epi_ekf_run_host_multi(const epi_batch_desc *d, const epi_inputs *in, const epi_outputs *out, int n_devices,
                           const int *device_ids, char *err)
{
    int rc = host_args_ok(d, in, out, err);
    if (rc != EPI_OK) return rc;
    if ((rc = multi_devices_ok(n_devices, device_ids, err)) != EPI_OK) return rc;
    const int per = (d->B + n_devices - 1) / n_devices;
    std::vector<int> rcs((size_t)n_devices, EPI_OK);
    std::vector<std::vector<char>> errs((size_t)n_devices, std::vector<char>(256, 0));
    auto dev_of = [&](int r) { return device_ids ? device_ids[r] : r; };
    run_on_devices(n_devices, dev_of, [&](int r) {
        const int lo = r * per < d->B ? r * per : d->B, n = (lo + per <= d->B ? per : d->B - lo);
        if (n <= 0) return;
        hipError_t e = hipSuccess;
        HostCtx *cx = ctx_acquire(dev_of(r), &e);
        if (!cx || e != hipSuccess) { if (cx) ctx_release(cx); rcs[(size_t)r] = hip_fail(errs[(size_t)r].data(), e, "hipSetDevice / context"); return; }
        rcs[(size_t)r] = run_host_block(cx, d, in, out, lo, n, errs[(size_t)r].data());
        ctx_release(cx);
    });
    for (int r = 0; r < n_devices; r++)
        if (rcs[(size_t)r] != EPI_OK) { set_err(err, errs[(size_t)r].data()); return rcs[(size_t)r]; }
    return EPI_OK;
}

// regions [r0, r0 + Rd) of the sweep on one context (see epi_sweep_prescribe_host)
static int prescribe_block(HostCtx *cx, const epi_prescribe_desc *pd, const epi_prescribe_inputs *in, const epi_prescribe_outputs *out,
                           int r0, int Rd, char *err)
{
    if (Rd <= 0) return EPI_OK;
    const size_t R = (size_t)pd->R, P = (size_t)pd->P, T = (size_t)pd->T, n = (size_t)pd->n_npi;
    const size_t Bd = (size_t)Rd * P, Bfull = R * P, c0 = (size_t)r0 * P;
    const bool extras = pd->out_mask != 0;
    // the filter's descriptor for this block: chain c = region * P + ll reads series `region`
    epi_batch_desc d{};
    d.abi_version = EPIEKF_ABI_VERSION; d.model = EPI_MODEL_SIA6; d.B = (int32_t)Bd; d.T = pd->T; d.Sx = Rd; d.Su = Rd;
    d.n_npi = pd->n_npi; d.L = pd->L; d.order = pd->order; d.obs_type = pd->obs_type; d.r_mode = 1; d.q_mode = 0;
    d.out_mask = pd->out_mask | EPI_OUT_U_OPT_SMOOTH | (out->S_opt ? EPI_OUT_S_SMOOTH : 0u);
    d.shape = pd->shape; d.time_pipe = pd->time_pipe;
    d.exact_nonfinite = 1;
    {
        epi_inputs hin{};
        hin.s_init = in->s_init; hin.Ps_init = in->Ps_init; hin.s_final = in->s_final; hin.Ps_final = in->Ps_final; hin.Q = in->Q;
        d.path_hint = host_precheck(6, false, &hin, R, (size_t)r0, (size_t)Rd) ? 1 : 2;
    }
    // per-chain extras come back in the classic layout; without them the filter writes the layout it runs fastest in
    d.lane_block = extras ? 0 : epi_ekf_preferred_lane_block(&d);
    if (d.lane_block >= d.B) d.lane_block = 0;
    int rc = epi_ekf_validate(&d, err);
    if (rc != EPI_OK) return rc;
    const size_t blk = d.lane_block ? (size_t)d.lane_block : Bd, nblk = (Bd + blk - 1) / blk, Bp = nblk * blk;

    HostIO io;
    const size_t o_x = io.add_in(in->x, T, 8, R, r0, Rd), o_u = io.add_in(in->u, T * n, 8, R, r0, Rd);
    const size_t o_rs = io.add_in(in->R_series, T, 8, R, r0, Rd), o_eps = io.add_in(in->eps, 1, 8, P, 0, P);
    // the per-region rows, contiguous and in the order of the per-chain rows sweep_expand writes
    io.off = (io.off + 255) & ~(size_t)255;
    io.align = 8;
    struct Row { const double *host; size_t rows; };
    const Row rows[] = {{in->prm, EPI_PRM_COUNT}, {in->s_init, 6}, {in->Ps_init, 36}, {in->s_final, 6}, {in->Ps_final, 36},
                        {in->Q, 36}, {in->sp, EPI_SIM_PRM_COUNT}, {in->J0_prefix, 1}, {in->J1_prefix, 1}};
    size_t o_reg = 0, nrows = 0;
    for (auto &rw : rows) {
        const size_t o = io.add_in(rw.host, rw.rows, 8, R, r0, Rd);
        if (nrows == 0) o_reg = o;
        nrows += rw.rows;
    }
    io.align = 256;
    io.off = (io.off + 255) & ~(size_t)255; io.in_bytes = io.off;
    if (!io.inputs_present()) { set_err(err, "NULL input array"); return EPI_ERR_BAD_ARG; }
    // outputs, in one run of the arena so that a small call downloads them with one copy
    const size_t o_j0 = io.add_out(out->J0, 1, 8, Bfull, c0, Bd), o_j1 = io.add_out(out->J1, 1, 8, Bfull, c0, Bd);
    const size_t o_of = io.add_out(out->on_front, 1, 4, Bfull, c0, Bd), o_io = io.add_out(out->i_opt, 1, 4, R, r0, Rd);
    const size_t o_uo = io.add_out(out->u_opt, T * n, 8, R, r0, Rd), o_so = io.add_out(out->S_opt, T * 6, 8, R, r0, Rd);
    struct O { uint32_t bit; double *host; double *epi_outputs::*dev; size_t rows; };
    const epi_outputs &ex = out->extras;
    O olist[] = {{EPI_OUT_U_OPT, ex.u_opt, &epi_outputs::u_opt, T * n}, {EPI_OUT_U_OPT_SMOOTH, ex.u_opt_smooth, &epi_outputs::u_opt_smooth, T * n},
                 {EPI_OUT_S_MINUS, ex.S_MINUS, &epi_outputs::S_MINUS, T * 6}, {EPI_OUT_S_PLUS, ex.S_PLUS, &epi_outputs::S_PLUS, T * 6},
                 {EPI_OUT_S_SMOOTH, ex.S_SMOOTH, &epi_outputs::S_SMOOTH, T * 6}, {EPI_OUT_P_MINUS, ex.P_MINUS, &epi_outputs::P_MINUS, T * 36},
                 {EPI_OUT_P_PLUS, ex.P_PLUS, &epi_outputs::P_PLUS, T * 36}, {EPI_OUT_P_SMOOTH, ex.P_SMOOTH, &epi_outputs::P_SMOOTH, T * 36},
                 {EPI_OUT_K_GAIN, ex.K_GAIN, &epi_outputs::K_GAIN, T * 6}, {EPI_OUT_INNOVATIONS, ex.innovations, &epi_outputs::innovations, T},
                 {EPI_OUT_RHO, ex.rho, &epi_outputs::rho, T}};
    std::vector<size_t> o_ex;
    for (auto &o : olist) {
        if ((pd->out_mask & o.bit) && !o.host) { set_err(err, "extras: output selected but NULL"); return EPI_ERR_BAD_ARG; }
        o_ex.push_back((pd->out_mask & o.bit) ? io.add_out(o.host, o.rows, 8, Bfull, c0, Bd) : (size_t)-1);
    }
    // device-only: expanded per-chain rows, the series map, filter outputs that do not leave the device, workspace
    const size_t o_chain = io.reserve(nrows * Bd * 8), o_ser = io.reserve(Bd * 4);
    const size_t o_uos = (pd->out_mask & EPI_OUT_U_OPT_SMOOTH) ? (size_t)-1 : io.reserve(T * n * Bp * 8);
    const size_t o_ss = ((pd->out_mask & EPI_OUT_S_SMOOTH) || !out->S_opt) ? (size_t)-1 : io.reserve(T * 6 * Bp * 8);
    const size_t wsb = epi_ekf_workspace_bytes(&d);
    const size_t o_ws = io.reserve(wsb);
    hipError_t e = hipSuccess;
    // upload + kernels for the arena at `base` (place_and_run may call this for several candidate arenas: every call leaves the
    // complete results in its arena); ev0 / ev1 bracket the kernels
    auto compute = [&](char *base, hipEvent_t ev0, hipEvent_t ev1) -> int {
    // (as in run_host_block: after upload() every error return waits for the stream)
    if ((e = io.upload(cx, base)) != hipSuccess) { (void)hipStreamSynchronize(cx->stream); return hip_fail(err, e, "upload"); }
    if (ev0) (void)hipEventRecord(ev0, cx->stream);
    double *chain = (double *)(base + o_chain);
    int32_t *series = (int32_t *)(base + o_ser);
    hipLaunchKernelGGL(sweep_expand, dim3((unsigned)((Bd + 255) / 256), 8), dim3(256), 0, cx->stream, (int)nrows, Rd, (int)P,
                       (int)EPI_PRM_EPSILON, (const double *)(base + o_reg), (const double *)(base + o_eps), chain, series);
    if ((e = hipGetLastError()) != hipSuccess) { (void)hipStreamSynchronize(cx->stream); return hip_fail(err, e, "sweep_expand launch"); }
    epi_inputs din{};
    din.x_series = series; din.u_series = series;
    din.x = (const double *)(base + o_x); din.u = (const double *)(base + o_u); din.R_series = (const double *)(base + o_rs);
    size_t row = 0;
    auto take = [&](size_t nr) { const double *p = chain + row * Bd; row += nr; return p; };
    din.prm = take(EPI_PRM_COUNT); din.s_init = take(6); din.Ps_init = take(36); din.s_final = take(6); din.Ps_final = take(36);
    din.Q = take(36);
    const double *sp = take(EPI_SIM_PRM_COUNT), *j0p = take(1), *j1p = take(1);
    epi_outputs dd{};
    {
        size_t k = 0;
        for (auto &o : olist) { if (o_ex[k] != (size_t)-1) dd.*(o.dev) = (double *)(base + o_ex[k]); k++; }
    }
    if (!dd.u_opt_smooth) dd.u_opt_smooth = (double *)(base + o_uos);
    if (!dd.S_SMOOTH && out->S_opt) dd.S_SMOOTH = (double *)(base + o_ss);
    epi_sweep_desc sd{};
    sd.abi_version = EPIEKF_ABI_VERSION; sd.R = Rd; sd.P = (int32_t)P; sd.t_hist = pd->t_hist;
    rc = epi_sweep_run_device(&d, &din, &dd, wsb ? base + o_ws : nullptr, wsb, &sd, sp, j0p, j1p, (double *)(base + o_j0),
                              (double *)(base + o_j1), (int32_t *)(base + o_of), (int32_t *)(base + o_io), cx->stream, err);
    if (rc != EPI_OK) { (void)hipStreamSynchronize(cx->stream); return rc; }
    struct G { double *host; const double *src; size_t off; int rows; };
    const G gs[] = {{out->u_opt, dd.u_opt_smooth, o_uo, (int)n}, {out->S_opt, dd.S_SMOOTH, o_so, 6}};
    for (auto &g : gs) {
        if (!g.host) continue;
        const size_t cnt = T * (size_t)g.rows * (size_t)Rd;
        hipLaunchKernelGGL(sweep_gather_opt, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, cx->stream, (int)T, g.rows, Rd, (int)P,
                           (int)blk, (int)nblk, (const int32_t *)(base + o_io), g.src, (double *)(base + g.off));
        if ((e = hipGetLastError()) != hipSuccess) { (void)hipStreamSynchronize(cx->stream); return hip_fail(err, e, "sweep_gather_opt launch"); }
    }
    if (ev1) (void)hipEventRecord(ev1, cx->stream);
    return EPI_OK;
    };
    rc = place_and_run(cx, io.off + 256, pd->placement_tries, r0 == 0 ? out->placement : nullptr, compute, err);
    if (rc != EPI_OK) return rc;
    char *base = cx->arena;
    if ((e = io.download(cx, base)) != hipSuccess) return hip_fail(err, e, "kernel execution / download");
    return EPI_OK;
}

int epi_sweep_prescribe_host(const epi_prescribe_desc *d, const epi_prescribe_inputs *in, const epi_prescribe_outputs *out,
                             int n_devices, const int *device_ids, char *err)
{
    if (!d || !in || !out || d->abi_version != EPIEKF_ABI_VERSION) { set_err(err, "bad prescribe descriptor"); return EPI_ERR_BAD_ARG; }
    if (d->R < 1 || d->P < 1 || (int64_t)d->R * d->P > ((int64_t)1 << 23)) { set_err(err, "R, P must be >= 1 and R * P <= 2^23"); return EPI_ERR_BAD_ARG; }
    if (d->P > 8192) { set_err(err, "more than 8192 points per region"); return EPI_ERR_UNSUPPORTED; }
    if (d->t_hist < 1 || d->t_hist >= d->T) { set_err(err, "t_hist must leave at least one horizon day: 1 <= t_hist < T"); return EPI_ERR_BAD_ARG; }
    if (d->out_mask & ~(uint32_t)EPI_OUT_ALL) { set_err(err, "unknown bits in out_mask"); return EPI_ERR_BAD_ARG; }
    if (d->placement_tries < 0 || d->placement_tries > EPI_PLACEMENT_MAX_TRIES) { set_err(err, "placement_tries must be 0 .. EPI_PLACEMENT_MAX_TRIES"); return EPI_ERR_BAD_ARG; }
    int rc = multi_devices_ok(n_devices, device_ids, err);
    if (rc != EPI_OK) return rc;
    const int per = (d->R + n_devices - 1) / n_devices;
    std::vector<int> rcs((size_t)n_devices, EPI_OK);
    std::vector<std::vector<char>> errs((size_t)n_devices, std::vector<char>(256, 0));
    auto dev_of = [&](int r) { return device_ids ? device_ids[r] : r; };
    run_on_devices(n_devices, dev_of, [&](int r) {
        const int lo = r * per < d->R ? r * per : d->R, n = (lo + per <= d->R ? per : d->R - lo);
        if (n <= 0) return;
        hipError_t e = hipSuccess;
        HostCtx *cx = ctx_acquire(dev_of(r), &e);
        if (!cx || e != hipSuccess) { if (cx) ctx_release(cx); rcs[(size_t)r] = hip_fail(errs[(size_t)r].data(), e, "hipSetDevice / context"); return; }
        rcs[(size_t)r] = prescribe_block(cx, d, in, out, lo, n, errs[(size_t)r].data());
        ctx_release(cx);
    });
    for (int r = 0; r < n_devices; r++)
        if (rcs[(size_t)r] != EPI_OK) { set_err(err, errs[(size_t)r].data()); return rcs[(size_t)r]; }
    return EPI_OK;
}

void epi_host_pool_release(void)
{
    workers_release_all();
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (HostCtx *c : g_pool) delete c;
        g_pool.clear();
    }
    helpers_release_all();
}

int epi_calib_copy_f64_device(const double *src, double *dst, size_t n, void *stream, char *err)
{
    if (!src || !dst || n == 0) { set_err(err, "bad calibration arguments"); return EPI_ERR_BAD_ARG; }
    hipLaunchKernelGGL(calib_copy_f64, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(err, e, "calib_copy_f64 launch");
    return EPI_OK;
}

static int sialpha_launch(const epi_sim_desc *d, const int32_t *u_series, const double *u, const double *sp,
                          const double *z, double *s, double *i, double *alpha, double *J0, double *J1,
                          const double *J0_prefix, const double *J1_prefix, void *stream, char *err)
{
    if (!d || d->abi_version != EPIEKF_ABI_VERSION || d->B < 1 || d->K < 1 || d->Su < 1 || d->n_npi < 1 ||
        d->n_npi > EPI_MAX_NPI || !u || !sp || (d->noise && !z) || (!u_series && d->Su != d->B) || d->prefix_days < 0 ||
        (d->prefix_days > 0 && (!J0_prefix || !J1_prefix))) {
        set_err(err, "bad simulator descriptor"); return EPI_ERR_BAD_ARG;
    }
    const int blocks = (d->B + 255) / 256;
    hipLaunchKernelGGL(sialpha_sim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *d, u_series, u, sp, z, s, i, alpha,
                       J0, J1, J0_prefix, J1_prefix, (const int32_t *)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(err, e, "sialpha_sim launch");
    return EPI_OK;
}

int epi_sialpha_sim_device(const epi_sim_desc *d, const int32_t *u_series, const double *u, const double *sp,
                           const double *z, double *s, double *i, double *alpha, double *J0, double *J1,
                           void *stream, char *err)
{
    if (d && d->prefix_days != 0) { set_err(err, "prefix_days needs epi_sialpha_score_device"); return EPI_ERR_BAD_ARG; }
    return sialpha_launch(d, u_series, u, sp, z, s, i, alpha, J0, J1, nullptr, nullptr, stream, err);
}

int epi_sialpha_score_device(const epi_sim_desc *d, const int32_t *u_series, const double *u, const double *sp,
                             const double *z, const double *J0_prefix, const double *J1_prefix, double *s, double *i,
                             double *alpha, double *J0, double *J1, void *stream, char *err)
{
    return sialpha_launch(d, u_series, u, sp, z, s, i, alpha, J0, J1, J0_prefix, J1_prefix, stream, err);
}

int epi_random_npi_mc_device(const epi_mc_desc *d, const double *sp, const double *u_min, const double *z,
                             const double *J0_prefix, const double *J1_prefix, double *u_out, double *J0, double *J1,
                             void *stream, char *err)
{
    if (!d || d->abi_version != EPIEKF_ABI_VERSION || d->R < 1 || d->n_scen < 1 || d->K < 1 || d->n_npi < 1 ||
        d->n_npi > EPI_MAX_NPI || !sp || !u_min || !J0 || !J1 || (d->noise && !z) || d->prefix_days < 0 ||
        (d->prefix_days > 0 && (!J0_prefix || !J1_prefix)) || (int64_t)d->R * d->n_scen > (int64_t)1 << 30) {
        set_err(err, "bad Monte-Carlo scenario descriptor"); return EPI_ERR_BAD_ARG;
    }
    const int B = d->R * d->n_scen;
    hipLaunchKernelGGL(random_npi_mc, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d, sp, u_min, z,
                       J0_prefix, J1_prefix, u_out, J0, J1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(err, e, "random_npi_mc launch");
    return EPI_OK;
}

int epi_pareto_front_device(int32_t R, int32_t P, const double *J0, const double *J1, int32_t *on_front,
                            int32_t *i_opt, void *stream, char *err)
{
    if (R < 1 || P < 1 || !J0 || !J1 || (!on_front && !i_opt)) { set_err(err, "bad Pareto-front arguments"); return EPI_ERR_BAD_ARG; }
    if (P > 8192) { set_err(err, "more than 8192 points per region"); return EPI_ERR_UNSUPPORTED; }
    const size_t shmem = (size_t)2 * P * sizeof(double);
    if (shmem > 64u * 1024u) {
        hipError_t e = hipFuncSetAttribute((const void *)pareto_front, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return hip_fail(err, e, "hipFuncSetAttribute");
    }
    hipLaunchKernelGGL(pareto_front, dim3(R), dim3(256), shmem, (hipStream_t)stream, P, J0, J1, on_front, i_opt, (const int32_t *)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(err, e, "pareto_front launch");
    return EPI_OK;
}

int epi_npi_cost_device(int32_t B, int32_t T, int32_t n_npi, int32_t Su, int32_t weights_per_day,
                        const int32_t *u_series, const double *newcases, const double *inputs, const double *weights,
                        double *J0, double *J1, void *stream, char *err)
{
    if (B < 1 || T < 1 || n_npi < 1 || Su < 1 || !newcases || !inputs || !weights || !J0 || !J1 || (!u_series && Su != B)) {
        set_err(err, "bad NPICost arguments"); return EPI_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL(npi_cost, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, T, n_npi, Su,
                       weights_per_day, u_series, newcases, inputs, weights, J0, J1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(err, e, "npi_cost launch");
    return EPI_OK;
}

int epi_si_controlled_device(int32_t B, int32_t K, int32_t Sa, double dt, const int32_t *alpha_series, const double *alpha,
                             const double *prm, double *s, double *i, void *stream, char *err)
{
    if (B < 1 || K < 1 || Sa < 1 || !prm || !s || !i || (K > 1 && !alpha) || (!alpha_series && Sa != B)) {
        set_err(err, "bad SI_Controlled arguments"); return EPI_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL(si_controlled, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, K, Sa, dt, alpha_series,
                       alpha, prm, s, i);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(err, e, "si_controlled launch");
    return EPI_OK;
}

int epi_sir_sim_device(int32_t B, int32_t K, double dt, const double *prm, double *out, void *stream, char *err)
{
    if (B < 1 || K < 1 || !prm || !out) { set_err(err, "bad SIR arguments"); return EPI_ERR_BAD_ARG; }
    hipLaunchKernelGGL(sir_sim, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, K, dt, prm, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(err, e, "sir_sim launch");
    return EPI_OK;
}

// ---- host-pointer variants of the simulators and the cost (what a MEX gateway binds: matlab/epiekf_sim_mex.cpp) ----
namespace {
struct HostStage {   // device copies of host arrays for one call; frees everything on destruction
    std::vector<void *> allocs;
    std::vector<std::tuple<void *, void *, size_t>> downloads;
    hipError_t e = hipSuccess;
    ~HostStage() { for (void *p : allocs) (void)hipFree(p); }
    void *in(const void *host, size_t bytes)
    {
        if (!host || e != hipSuccess) return nullptr;
        void *p = nullptr;
        if ((e = hipMalloc(&p, bytes)) != hipSuccess) return nullptr;
        allocs.push_back(p);
        e = hipMemcpy(p, host, bytes, hipMemcpyHostToDevice);
        return p;
    }
    void *out(void *host, size_t bytes)
    {
        if (!host || e != hipSuccess) return nullptr;
        void *p = nullptr;
        if ((e = hipMalloc(&p, bytes)) != hipSuccess) return nullptr;
        allocs.push_back(p);
        downloads.emplace_back(host, p, bytes);
        return p;
    }
    int finish(int rc, char *err)
    {
        if (rc != EPI_OK) return rc;
        if (e == hipSuccess) e = hipDeviceSynchronize();
        for (auto &d : downloads)
            if (e == hipSuccess) e = hipMemcpy(std::get<0>(d), std::get<1>(d), std::get<2>(d), hipMemcpyDeviceToHost);
        return e == hipSuccess ? EPI_OK : hip_fail(err, e, "host staging");
    }
};
}  // namespace

int epi_sialpha_sim_host(const epi_sim_desc *d, const int32_t *u_series, const double *u, const double *sp,
                         const double *z, double *s, double *i, double *alpha, double *J0, double *J1, int device,
                         char *err)
{
    if (!d || d->B < 1 || d->K < 1 || d->Su < 1 || d->n_npi < 1 || !u || !sp) { set_err(err, "bad simulator descriptor"); return EPI_ERR_BAD_ARG; }
    if (d->u_block != 0) { set_err(err, "u_block is a device-side layout"); return EPI_ERR_BAD_ARG; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(err, e, "hipSetDevice");
    const size_t B = d->B, K = d->K;
    HostStage h;
    const void *dus = h.in(u_series, B * 4), *du = h.in(u, K * d->n_npi * (size_t)d->Su * 8);
    const void *dsp = h.in(sp, (size_t)EPI_SIM_PRM_COUNT * B * 8), *dz = h.in(z, K * 3 * B * 8);
    void *ds = h.out(s, K * B * 8), *di = h.out(i, K * B * 8), *da = h.out(alpha, K * B * 8);
    void *d0 = h.out(J0, B * 8), *d1 = h.out(J1, B * 8);
    if (h.e != hipSuccess) return hip_fail(err, h.e, "host staging");
    return h.finish(epi_sialpha_sim_device(d, (const int32_t *)dus, (const double *)du, (const double *)dsp, (const double *)dz,
                                           (double *)ds, (double *)di, (double *)da, (double *)d0, (double *)d1, nullptr, err), err);
}

int epi_seirp_sim_host(int32_t B, int32_t K, int32_t par_steps, double dt, int32_t saturated, int32_t integrator,
                       const double *par, const double *init, const double *sat, double *out, int device, char *err)
{
    if (B < 1 || K < 1 || par_steps < 1 || !par || !init || !out) { set_err(err, "bad SEIRP arguments"); return EPI_ERR_BAD_ARG; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(err, e, "hipSetDevice");
    HostStage h;
    const void *dp = h.in(par, (size_t)par_steps * 7 * B * 8), *di = h.in(init, (size_t)5 * B * 8), *ds = h.in(sat, (size_t)6 * B * 8);
    void *dout = h.out(out, (size_t)K * 5 * B * 8);
    if (h.e != hipSuccess) return hip_fail(err, h.e, "host staging");
    return h.finish(epi_seirp_sim_device(B, K, par_steps, dt, saturated, integrator, (const double *)dp, (const double *)di,
                                         (const double *)ds, (double *)dout, nullptr, err), err);
}

int epi_si_controlled_host(int32_t B, int32_t K, int32_t Sa, double dt, const int32_t *alpha_series, const double *alpha,
                           const double *prm, double *s, double *i, int device, char *err)
{
    if (B < 1 || K < 1 || Sa < 1) { set_err(err, "bad SI_Controlled arguments"); return EPI_ERR_BAD_ARG; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(err, e, "hipSetDevice");
    HostStage h;
    const void *das = h.in(alpha_series, (size_t)B * 4), *da = h.in(alpha, (size_t)(K > 1 ? K - 1 : 1) * Sa * 8);
    const void *dp = h.in(prm, (size_t)3 * B * 8);
    void *ds = h.out(s, (size_t)K * B * 8), *di = h.out(i, (size_t)K * B * 8);
    if (h.e != hipSuccess) return hip_fail(err, h.e, "host staging");
    return h.finish(epi_si_controlled_device(B, K, Sa, dt, (const int32_t *)das, (const double *)da, (const double *)dp,
                                             (double *)ds, (double *)di, nullptr, err), err);
}

int epi_npi_cost_host(int32_t B, int32_t T, int32_t n_npi, int32_t Su, int32_t weights_per_day, const int32_t *u_series,
                      const double *newcases, const double *inputs, const double *weights, double *J0, double *J1,
                      int device, char *err)
{
    if (B < 1 || T < 1 || n_npi < 1 || Su < 1) { set_err(err, "bad NPICost arguments"); return EPI_ERR_BAD_ARG; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(err, e, "hipSetDevice");
    HostStage h;
    const void *dus = h.in(u_series, (size_t)B * 4), *dn = h.in(newcases, (size_t)T * B * 8);
    const void *du = h.in(inputs, (size_t)T * n_npi * Su * 8);
    const void *dw = h.in(weights, (size_t)(weights_per_day ? T : 1) * n_npi * B * 8);
    void *d0 = h.out(J0, (size_t)B * 8), *d1 = h.out(J1, (size_t)B * 8);
    if (h.e != hipSuccess) return hip_fail(err, h.e, "host staging");
    return h.finish(epi_npi_cost_device(B, T, n_npi, Su, weights_per_day, (const int32_t *)dus, (const double *)dn,
                                        (const double *)du, (const double *)dw, (double *)d0, (double *)d1, nullptr, err), err);
}

int epi_sir_sim_host(int32_t B, int32_t K, double dt, const double *prm, double *out, int device, char *err)
{
    if (B < 1 || K < 1) { set_err(err, "bad SIR arguments"); return EPI_ERR_BAD_ARG; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(err, e, "hipSetDevice");
    HostStage h;
    const void *dp = h.in(prm, (size_t)6 * B * 8);
    void *dout = h.out(out, (size_t)K * 3 * B * 8);
    if (h.e != hipSuccess) return hip_fail(err, h.e, "host staging");
    return h.finish(epi_sir_sim_device(B, K, dt, (const double *)dp, (double *)dout, nullptr, err), err);
}

int epi_preprocess_host(const epi_pre_desc *d, const double *cases, const double *deaths, const double *population,
                        const double *ip, const epi_pre_outputs *out, int device, char *err)
{
    if (!d || !out || d->S < 1 || d->T < 1 || d->n_npi < 0 || d->n_npi > EPI_MAX_NPI) { set_err(err, "bad preprocessing descriptor"); return EPI_ERR_BAD_ARG; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(err, e, "hipSetDevice");
    const size_t TS = (size_t)d->T * d->S * 8;
    HostStage h;
    const void *dc = h.in(cases, TS), *dd = h.in(deaths, TS), *dp = h.in(population, (size_t)d->S * 8);
    const void *dip = h.in(ip, TS * (size_t)d->n_npi);
    epi_pre_outputs dout{};
    dout.new_refined = (double *)h.out(out->new_refined, TS); dout.new_smoothed = (double *)h.out(out->new_smoothed, TS);
    dout.zero_lag = (double *)h.out(out->zero_lag, TS); dout.x_new = (double *)h.out(out->x_new, TS);
    dout.x_total = (double *)h.out(out->x_total, TS); dout.R_v = (double *)h.out(out->R_v, TS);
    dout.fatality = (double *)h.out(out->fatality, TS); dout.I0 = (double *)h.out(out->I0, (size_t)d->S * 8);
    dout.ip_filled = (double *)h.out(out->ip_filled, TS * (size_t)d->n_npi);
    const size_t wsb = epi_preprocess_workspace_bytes(d);
    void *ws = nullptr;
    if (wsb && h.e == hipSuccess && (h.e = hipMalloc(&ws, wsb)) == hipSuccess) h.allocs.push_back(ws);
    if (h.e != hipSuccess) return hip_fail(err, h.e, "host staging");
    return h.finish(epi_preprocess_device(d, (const double *)dc, (const double *)dd, (const double *)dp, (const double *)dip, &dout,
                                          ws, wsb, nullptr, err), err);
}

int epi_nnls_affine_fit_host(const epi_nnls_desc *d, const double *X, const double *y, double *a, double *b,
                             double *min_err, int32_t *iters, int32_t *flag, int device, char *err)
{
    if (!d || d->S < 1 || d->D < 1 || d->n < 1 || d->n > EPI_MAX_NPI) { set_err(err, "bad NNLS descriptor"); return EPI_ERR_BAD_ARG; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(err, e, "hipSetDevice");
    const size_t S = d->S;
    HostStage h;
    const void *dX = h.in(X, (size_t)d->D * d->n * S * 8), *dy = h.in(y, (size_t)d->D * S * 8);
    void *da = h.out(a, (size_t)d->n * S * 8), *db = h.out(b, S * 8), *dm = h.out(min_err, S * 8);
    void *di = h.out(iters, S * 4), *df = h.out(flag, S * 4);
    if (h.e != hipSuccess) return hip_fail(err, h.e, "host staging");
    return h.finish(epi_nnls_affine_fit_device(d, (const double *)dX, (const double *)dy, (double *)da, (double *)db, (double *)dm,
                                               (int32_t *)di, (int32_t *)df, nullptr, err), err);
}

int epi_random_npi_mc_host(const epi_mc_desc *d, const double *sp, const double *u_min, const double *z,
                           const double *J0_prefix, const double *J1_prefix, double *u_out, double *J0, double *J1,
                           int device, char *err)
{
    if (!d || d->R < 1 || d->n_scen < 1 || d->K < 1 || d->n_npi < 1 || d->n_npi > EPI_MAX_NPI ||
        (int64_t)d->R * d->n_scen > (int64_t)1 << 30) { set_err(err, "bad Monte-Carlo scenario descriptor"); return EPI_ERR_BAD_ARG; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(err, e, "hipSetDevice");
    const size_t R = d->R, B = R * (size_t)d->n_scen, K = d->K;
    HostStage h;
    const void *dsp = h.in(sp, (size_t)EPI_SIM_PRM_COUNT * R * 8), *dum = h.in(u_min, (size_t)d->n_npi * R * 8);
    const void *dz = h.in(z, K * 3 * B * 8), *dj0 = h.in(J0_prefix, R * 8), *dj1 = h.in(J1_prefix, R * 8);
    void *du = h.out(u_out, K * (size_t)d->n_npi * B * 8), *d0 = h.out(J0, B * 8), *d1 = h.out(J1, B * 8);
    if (h.e != hipSuccess) return hip_fail(err, h.e, "host staging");
    return h.finish(epi_random_npi_mc_device(d, (const double *)dsp, (const double *)dum, (const double *)dz, (const double *)dj0,
                                             (const double *)dj1, (double *)du, (double *)d0, (double *)d1, nullptr, err), err);
}

int epi_rt_expfit_validate(const epi_rt_desc *d, char *err)
{
    if (!d) { set_err(err, "NULL descriptor"); return EPI_ERR_BAD_ARG; }
    if (d->abi_version != EPIEKF_ABI_VERSION) { set_err(err, "ABI version mismatch"); return EPI_ERR_BAD_ARG; }
    if (d->B < 1 || d->T < 1 || d->Sx < 1 || d->L < 1) { set_err(err, "B, T, Sx, L must be >= 1"); return EPI_ERR_BAD_ARG; }
    if (d->B > (1 << 23) || d->Sx > (1 << 23)) { set_err(err, "B, Sx are limited to 2^23"); return EPI_ERR_BAD_ARG; }
    if (d->order != 1 && d->order != 2) { set_err(err, epi_status_string(EPI_ERR_UNDEFINED_ORDER)); return EPI_ERR_UNDEFINED_ORDER; }
    if ((size_t)3 * d->L * kWave * sizeof(double) > 160u * 1024u) { set_err(err, "inv_monitor_len too large for LDS (max 106)"); return EPI_ERR_UNSUPPORTED; }
    return EPI_OK;
}

int epi_rt_expfit_run_device(const epi_rt_desc *d, const int32_t *x_series, const double *x, const double *rp,
                             const epi_rt_outputs *out, void *stream, char *err)
{
    int rc = epi_rt_expfit_validate(d, err);
    if (rc != EPI_OK) return rc;
    if (!x || !rp || !out) { set_err(err, "NULL input array"); return EPI_ERR_BAD_ARG; }
    if (!x_series && d->Sx != d->B) { set_err(err, "identity x_series needs Sx == B"); return EPI_ERR_BAD_ARG; }
    if (!out->S_MINUS || !out->S_PLUS || !out->P_MINUS || !out->P_PLUS) {
        set_err(err, "S_MINUS, S_PLUS, P_MINUS, P_PLUS are required outputs"); return EPI_ERR_BAD_ARG;
    }
    RtArgs a{};
    a.B = d->B; a.T = d->T; a.Sx = d->Sx; a.L = d->L; a.order = d->order;
    a.x_series = x_series; a.x = x; a.rp = rp;
    a.S_MINUS = out->S_MINUS; a.S_PLUS = out->S_PLUS; a.P_MINUS = out->P_MINUS; a.P_PLUS = out->P_PLUS;
    a.K_GAIN = out->K_GAIN; a.S_SMOOTH = out->S_SMOOTH; a.P_SMOOTH = out->P_SMOOTH;
    a.innovations = out->innovations; a.rho = out->rho;
    const size_t shmem = (size_t)3 * d->L * kWave * sizeof(double);
    hipError_t e;
    if (shmem > 64u * 1024u &&
        (e = hipFuncSetAttribute((const void *)rt_expfit_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)) != hipSuccess)
        return hip_fail(err, e, "hipFuncSetAttribute");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rt_expfit_fwd, dim3((d->B + kWave - 1) / kWave), dim3(kWave), shmem, st, a);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(err, e, "rt_expfit_fwd launch");
    if (a.S_SMOOTH || a.P_SMOOTH) {
        hipLaunchKernelGGL(rt_expfit_bwd, dim3((d->B + 255) / 256), dim3(256), 0, st, a);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(err, e, "rt_expfit_bwd launch");
    }
    return EPI_OK;
}

int epi_rt_expfit_run_host(const epi_rt_desc *d, const int32_t *x_series, const double *x, const double *rp,
                           const epi_rt_outputs *out, int device, char *err)
{
    int rc = epi_rt_expfit_validate(d, err);
    if (rc != EPI_OK) return rc;
    if (!x || !rp || !out) { set_err(err, "NULL input array"); return EPI_ERR_BAD_ARG; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(err, e, "hipSetDevice");
    const size_t B = d->B, T = d->T;
    std::vector<void *> allocs;
    auto fail = [&](hipError_t ee, const char *what) { for (void *p : allocs) (void)hipFree(p); return hip_fail(err, ee, what); };
    auto dev_alloc = [&](size_t bytes, void **p) -> hipError_t {
        hipError_t ee = hipMalloc(p, bytes);
        if (ee == hipSuccess) allocs.push_back(*p);
        return ee;
    };
    void *dxs = nullptr, *dx = nullptr, *drp = nullptr;
    if (x_series) {
        if ((e = dev_alloc(B * 4, &dxs)) != hipSuccess || (e = hipMemcpy(dxs, x_series, B * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload x_series");
    }
    if ((e = dev_alloc(T * d->Sx * 8, &dx)) != hipSuccess || (e = hipMemcpy(dx, x, T * d->Sx * 8, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload x");
    if ((e = dev_alloc((size_t)EPI_RT_PRM_COUNT * B * 8, &drp)) != hipSuccess || (e = hipMemcpy(drp, rp, (size_t)EPI_RT_PRM_COUNT * B * 8, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload rp");
    struct O { double *host; double **dev; size_t bytes; bool required; };
    epi_rt_outputs dout{};
    const size_t n2 = T * 2 * B * 8, n4 = T * 4 * B * 8, n1 = T * B * 8;
    O outs[] = {{out->S_MINUS, &dout.S_MINUS, n2, true}, {out->S_PLUS, &dout.S_PLUS, n2, true}, {out->P_MINUS, &dout.P_MINUS, n4, true},
                {out->P_PLUS, &dout.P_PLUS, n4, true}, {out->K_GAIN, &dout.K_GAIN, n2, false}, {out->S_SMOOTH, &dout.S_SMOOTH, n2, false},
                {out->P_SMOOTH, &dout.P_SMOOTH, n4, false}, {out->innovations, &dout.innovations, n1, false}, {out->rho, &dout.rho, n1, false}};
    for (auto &o : outs)
        if (o.host || o.required) {   // forward quantities the caller does not want still feed the smoother
            void *p; if ((e = dev_alloc(o.bytes, &p)) != hipSuccess) return fail(e, "hipMalloc output");
            *o.dev = (double *)p;
        }
    rc = epi_rt_expfit_run_device(d, (const int32_t *)dxs, (const double *)dx, (const double *)drp, &dout, nullptr, err);
    if (rc != EPI_OK) { for (void *p : allocs) (void)hipFree(p); return rc; }
    if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(e, "kernel execution");
    for (auto &o : outs)
        if (o.host && (e = hipMemcpy(o.host, *o.dev, o.bytes, hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "download");
    for (void *p : allocs) (void)hipFree(p);
    return EPI_OK;
}

static int pre_validate(const epi_pre_desc *d, int *W2, int *nfact, char *err)
{
    if (!d || d->abi_version != EPIEKF_ABI_VERSION || d->S < 1 || d->n_npi < 0 || d->n_npi > EPI_MAX_NPI || d->first_num_days < 0) {
        set_err(err, "bad preprocessing descriptor"); return EPI_ERR_BAD_ARG;
    }
    if (d->W < 1 || d->W > kPreMaxTaps) { set_err(err, "SmoothingWinLen out of range 1..32"); return EPI_ERR_BAD_ARG; }
    if (d->T < 2) { set_err(err, "Insufficient data"); return EPI_ERR_BAD_ARG; }            // :168
    int w2 = (int)floor((double)d->W / 2 + 0.5);                                            // MATLAB round()
    if (w2 < 1) w2 = 1;
    const int nf = (3 * (w2 - 1) > 1) ? 3 * (w2 - 1) : 1;
    if (d->T <= nf) {
        char b[96]; snprintf(b, sizeof b, "Data length must be larger than %d samples.", nf);
        set_err(err, b); return EPI_ERR_BAD_ARG;
    }
    *W2 = w2; *nfact = nf;
    return EPI_OK;
}

size_t epi_preprocess_workspace_bytes(const epi_pre_desc *d)
{
    int W2, nfact;
    if (pre_validate(d, &W2, &nfact, nullptr) != EPI_OK) return 0;
    return ((size_t)3 * d->T + 2 * (size_t)nfact) * (size_t)d->S * sizeof(double);
}

int epi_preprocess_device(const epi_pre_desc *d, const double *cases, const double *deaths, const double *population,
                          const double *ip, const epi_pre_outputs *out, void *workspace, size_t workspace_bytes,
                          void *stream, char *err)
{
    int W2 = 0, nfact = 0;
    int rc = pre_validate(d, &W2, &nfact, err);
    if (rc != EPI_OK) return rc;
    if (!cases || !population || !out) { set_err(err, "NULL input array"); return EPI_ERR_BAD_ARG; }
    if (out->ip_filled && (!ip || d->n_npi < 1)) { set_err(err, "ip_filled selected without ip"); return EPI_ERR_BAD_ARG; }
    if (out->fatality && !deaths) { set_err(err, "fatality selected without deaths"); return EPI_ERR_BAD_ARG; }
    if (!workspace || workspace_bytes < epi_preprocess_workspace_bytes(d)) {
        set_err(err, epi_status_string(EPI_ERR_WORKSPACE)); return EPI_ERR_WORKSPACE;
    }
    PreArgs a{};
    a.T = d->T; a.S = d->S; a.W = d->W; a.W2 = W2; a.nfact = nfact; a.first_num_days = d->first_num_days;
    a.min_cases = d->min_cases;
    a.cases = cases; a.deaths = deaths; a.population = population;
    a.new_refined = out->new_refined; a.new_smoothed = out->new_smoothed; a.zero_lag = out->zero_lag;
    a.x_new = out->x_new; a.x_total = out->x_total; a.R_v = out->R_v; a.fatality = out->fatality; a.I0 = out->I0;
    a.ws = (double *)workspace;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(preprocess_regions, dim3((d->S + 63) / 64), dim3(64), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(err, e, "preprocess_regions launch");
    if (out->ip_filled) {
        const int cols = d->n_npi * d->S;
        hipLaunchKernelGGL(npi_fill, dim3((cols + 255) / 256), dim3(256), 0, st, d->T, cols, ip, out->ip_filled);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(err, e, "npi_fill launch");
    }
    return EPI_OK;
}

int epi_nnls_affine_fit_device(const epi_nnls_desc *d, const double *X, const double *y, double *a, double *b,
                               double *min_err, int32_t *iters, int32_t *flag, void *stream, char *err)
{
    if (!d || d->abi_version != EPIEKF_ABI_VERSION || d->S < 1 || d->D < 1 || d->n < 1 || d->n > kNnMax || d->max_iters < 0 ||
        !X || !y || !a) {
        set_err(err, "bad NNLS descriptor"); return EPI_ERR_BAD_ARG;
    }
    const size_t shmem = ((size_t)kNnDoubles * sizeof(double) + (size_t)kNnInts * sizeof(int)) * kNnLanes;
    hipError_t e = hipFuncSetAttribute((const void *)nnls_affine_fit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return hip_fail(err, e, "hipFuncSetAttribute");
    NnArgs g{};
    g.S = d->S; g.D = d->D; g.n = d->n; g.max_iters = d->max_iters;
    g.X = X; g.y = y; g.a = a; g.b = b; g.min_err = min_err; g.iters = iters; g.flag = flag;
    hipLaunchKernelGGL(nnls_affine_fit, dim3((d->S + kNnLanes - 1) / kNnLanes), dim3(kNnLanes), shmem, (hipStream_t)stream, g);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(err, e, "nnls_affine_fit launch");
    return EPI_OK;
}

int epi_seirp_sim_device(int32_t B, int32_t K, int32_t par_steps, double dt, int32_t saturated, int32_t integrator,
                         const double *par, const double *init, const double *sat, double *out, void *stream, char *err)
{
    if (B < 1 || K < 1 || !par || !init || !out || (saturated && !sat) || (par_steps != 1 && par_steps < K - 1) ||
        (integrator != 0 && integrator != 1)) {
        set_err(err, "bad SEIRP arguments"); return EPI_ERR_BAD_ARG;
    }
    const int blocks = (B + 255) / 256;
    hipLaunchKernelGGL(seirp_sim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, B, K, par_steps, dt, saturated, integrator, par, init, sat, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(err, e, "seirp_sim launch");
    return EPI_OK;
}

}  // extern "C"
