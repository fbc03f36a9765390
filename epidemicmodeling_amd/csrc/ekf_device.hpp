// ekf_device.hpp -- device-side building blocks of the EKF/EKS kernels (gfx950).
//
// One LANE owns one filter chain: the state vector, the m x m covariances and
// every temporary live in that lane's VGPRs (fully unrolled, compile-time
// indices only), so a wavefront advances 64 chains in lock-step with no
// cross-lane traffic, and every global access is one coalesced 512-byte row
// of a time-major / chain-minor array.
//
// Arithmetic contract: IEEE fp64, built with -ffp-contract=off so the compiler
// never fuses on its own.  Scalar / element-wise MATLAB expressions are
// evaluated as written, one rounding per * + - /.  BLAS-class operations
// (matrix-matrix, matrix-vector, dot products -- what MATLAB hands to its BLAS)
// accumulate with an explicit fma():  acc = a0*b0; acc = fma(a_k, b_k, acc),
// k ascending.  The Jacobi eigen-solver behind pinv uses fma() in its rotations.
// The CPU oracle uses the same fma() in the same places => bit-identical results.
// Citations: Tools/*.m of the reference, file:line.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/epiekf.h"

namespace epi {

constexpr int kNpi = EPI_MAX_NPI;
constexpr int kWave = 64;
constexpr double kEps = 2.220446049250313e-16;  // MATLAB eps

#define EPI_DEV __device__ __forceinline__
#define IXM(i, j) ((i) + M * (j))

// per-chain model constants: the reference's `params` struct
struct ChainPrm {
    double dt, beta, gamma, sigma, b, epsilon, slo, ilo, alpha_min, alpha_max;
    double a[kNpi], u_min[kNpi], u_max[kNpi], w[kNpi];
    EPI_DEV double A(int k) const { return a[k]; }
    EPI_DEV double Umin(int k) const { return u_min[k]; }
    EPI_DEV double Umax(int k) const { return u_max[k]; }
    EPI_DEV double W(int k) const { return w[k]; }
};

// the same constants with the four 12-vectors (a, u_min, u_max, w) kept in an LDS column per lane, so that
// they do not occupy 96 VGPRs for the whole life of the chain
struct VecLds {
    const double *base;   // this lane's column of a [4][12][64] block: a, u_min, u_max, w
    EPI_DEV double A(int k) const { return base[(0 * kNpi + k) * kWave]; }
    EPI_DEV double Umin(int k) const { return base[(1 * kNpi + k) * kWave]; }
    EPI_DEV double Umax(int k) const { return base[(2 * kNpi + k) * kWave]; }
    EPI_DEV double W(int k) const { return base[(3 * kNpi + k) * kWave]; }
};
// The three-state models read a and u_max only (SIAlphaModelEKF.m:39-48: no bang-bang substitution, no slope term), so their
// packed smoother keeps two 12-vectors per lane in LDS instead of four: 12 KB per 64-lane workgroup instead of 24 KB, which
// had capped it at six workgroups per CU (1.5 waves per SIMD) where its registers allow two.
struct VecLds2 {
    const double *base;   // this lane's column of a [2][12][64] block: a, u_max
    EPI_DEV double A(int k) const { return base[(0 * kNpi + k) * kWave]; }
    EPI_DEV double Umax(int k) const { return base[(1 * kNpi + k) * kWave]; }
    EPI_DEV double Umin(int) const { return 0.0; }     // six-state code only (resolve_control, the slope term)
    EPI_DEV double W(int) const { return 0.0; }
};
struct VecLdsS {          // the same with a run-time lane stride (LDS sized by the lanes a workgroup actually uses)
    const double *base;
    int stride;
    EPI_DEV double A(int k) const { return base[(0 * kNpi + k) * stride]; }
    EPI_DEV double Umin(int k) const { return base[(1 * kNpi + k) * stride]; }
    EPI_DEV double Umax(int k) const { return base[(2 * kNpi + k) * stride]; }
    EPI_DEV double W(int k) const { return base[(3 * kNpi + k) * stride]; }
};
template <class V>
struct LitePrm {
    double dt, beta, gamma, sigma, b, epsilon, slo, ilo, alpha_min, alpha_max;
    V v;
    EPI_DEV double A(int k) const { return v.A(k); }
    EPI_DEV double Umin(int k) const { return v.Umin(k); }
    EPI_DEV double Umax(int k) const { return v.Umax(k); }
    EPI_DEV double W(int k) const { return v.W(k); }
};
template <class V>
EPI_DEV void load_lite(LitePrm<V> &p, const double *__restrict__ prm, int B, int c, int lo_is_zero)
{
    auto g = [&](int f) { return prm[(size_t)f * B + c]; };
    p.dt = g(EPI_PRM_DT); p.beta = g(EPI_PRM_BETA); p.gamma = g(EPI_PRM_GAMMA);
    p.sigma = g(EPI_PRM_SIGMA); p.b = g(EPI_PRM_B); p.epsilon = g(EPI_PRM_EPSILON);
    p.slo = lo_is_zero ? 0.0 : g(EPI_PRM_S_MIN);
    p.ilo = lo_is_zero ? 0.0 : g(EPI_PRM_I_MIN);
    p.alpha_min = g(EPI_PRM_ALPHA_MIN); p.alpha_max = g(EPI_PRM_ALPHA_MAX);
}

// wave-uniform model switches (see MODEL_TABLE in epiekf.hip)
struct ModelFlags {
    int lo_is_zero;   // s,i clamps at 0 (OptControlled.m:28-29) instead of s_min/i_min (SIAlphaModelEKF.m:28-29)
    int phi_ge;       // phi >= 0 (NewCase...m:175) instead of phi > 0 (OptControlled.m:52)
    int obs_clamp;    // ObsHardMargins = max(0, .) ; identity in MatlabCodeGenerator/ObsHardMargins.m
    int obs_type;     // resolved observation type
};

EPI_DEV bool is_nan(double v) { return v != v; }

// ---- exp and tanh with a fixed operation order -----------------------------------------------------------------
// Rt_ExpFitEKF.m and SEIRPSaturatedResource.m call exp / tanh.  libm (CPU oracle) and the device math library round
// them differently, and a nearly singular 2 x 2 smoother gain amplifies that last-bit difference (seen: 1e-9 on
// P_SMOOTH).  Both sides therefore evaluate the SAME sequence: k = rint(x/ln2), two-part Cody-Waite reduction,
// degree-13 Taylor polynomial of expm1 in Horner form with fma, exact scaling by 2^k.  Error < 1 ulp for exp, a few
// ulp for tanh -- against MATLAB's own exp/tanh that is far inside the 1e-6 bar, and the parity tests become bit for
// bit.  oracle/ekf_oracle.c holds the same text.
EPI_DEV double epi_expm1_reduced(double r)
{
    // sum_{n>=2} r^(n-2)/n!, Horner with fma; the constants are correctly rounded quotients on every IEEE compiler
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
EPI_DEV double epi_reduce_ln2(double y, double *k)
{
    *k = rint(y * 1.44269504088896338700e+00);
    double r = fma(-*k, 6.93147180369123816490e-01, y);      // ln2 high part: 32 significant bits, k*hi exact
    return fma(-*k, 1.90821492927058770002e-10, r);          // ln2 low part
}
EPI_DEV double epi_exp(double x)
{
    if (x != x) return x;
    if (x > 709.78271289338397) return (double)INFINITY;
    if (x < -745.13321910194122) return 0.0;
    double k;
    const double r = epi_reduce_ln2(x, &k);
    return ldexp(1.0 + epi_expm1_reduced(r), (int)k);
}
EPI_DEV double epi_tanh(double x)
{
    if (x != x) return x;
    const double ax = fabs(x);
    double res = 1.0;                                         // |x| > 22: 1 - 2e-19 rounds to 1
    if (ax <= 22.0) {
        double k;
        const double r = epi_reduce_ln2(ax + ax, &k);
        const double q = epi_expm1_reduced(r);                // e^(2|x|) = 2^k (1 + q)
        const double s = ldexp(1.0, (int)k);                  // tanh = (e - 1)/(e + 1), both formed with one rounding
        res = fma(s, q, s - 1.0) / fma(s, q, s + 1.0);
    }
    return copysign(res, x);
}
EPI_DEV bool is_nonfinite(double v) { return !(fabs(v) <= 1.7976931348623157e308); }

template <int M>
EPI_DEV void load_prm(ChainPrm &p, const double *__restrict__ prm, int B, int c, int lo_is_zero)
{
    auto g = [&](int f) { return prm[(size_t)f * B + c]; };
    p.dt = g(EPI_PRM_DT); p.beta = g(EPI_PRM_BETA); p.gamma = g(EPI_PRM_GAMMA);
    p.sigma = g(EPI_PRM_SIGMA); p.b = g(EPI_PRM_B); p.epsilon = g(EPI_PRM_EPSILON);
    p.slo = lo_is_zero ? 0.0 : g(EPI_PRM_S_MIN);
    p.ilo = lo_is_zero ? 0.0 : g(EPI_PRM_I_MIN);
    p.alpha_min = g(EPI_PRM_ALPHA_MIN); p.alpha_max = g(EPI_PRM_ALPHA_MAX);
#pragma unroll
    for (int k = 0; k < kNpi; k++) {
        p.a[k] = g(EPI_PRM_A + k);
        p.u_max[k] = g(EPI_PRM_U_MAX + k);
        if (M == 6) {
            p.u_min[k] = g(EPI_PRM_U_MIN + k);
            p.w[k] = g(EPI_PRM_W_EFF + k);
        } else {
            p.u_min[k] = 0.0; p.w[k] = 0.0;
        }
    }
}

// ---- sliding windows of the innovation monitor (GenericEKF.m:172-179) -------
// `win` is this lane's column of an LDS ring buffer (stride kWave doubles); the newest sample sits at `head`
// and has already been written.  Returns the sum newest -> oldest, added strictly in that order (the
// reference's cat/sum order).  The LDS reads are issued ten at a time so that their latencies overlap; only
// the additions are serial.
EPI_DEV double ring_sum(const double *win, int head, int L, double newest, int stride = kWave)
{
    constexpr int BLK = 10;
    double sum = newest;
    int idx = head, j = 1;
    for (; j + BLK <= L; j += BLK) {
        double v[BLK];
#pragma unroll
        for (int q = 0; q < BLK; q++) { idx = (idx + 1 == L) ? 0 : idx + 1; v[q] = win[idx * stride]; }
#pragma unroll
        for (int q = 0; q < BLK; q++) sum = sum + v[q];
    }
    for (; j < L; j++) { idx = (idx + 1 == L) ? 0 : idx + 1; sum = sum + win[idx * stride]; }
    return sum;
}

// ---- dense helpers -------------------------------------------------------
template <int M>
EPI_DEV void mat_mul(const double (&A)[M * M], const double (&B)[M * M], double (&C)[M * M])
{
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = 0; i < M; i++) {
            double acc = A[IXM(i, 0)] * B[IXM(0, j)];
#pragma unroll
            for (int k = 1; k < M; k++) acc = fma(A[IXM(i, k)], B[IXM(k, j)], acc);
            C[IXM(i, j)] = acc;
        }
}
template <int M>
EPI_DEV void mat_mul_bt(const double (&A)[M * M], const double (&B)[M * M], double (&C)[M * M])
{
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = 0; i < M; i++) {
            double acc = A[IXM(i, 0)] * B[IXM(j, 0)];
#pragma unroll
            for (int k = 1; k < M; k++) acc = fma(A[IXM(i, k)], B[IXM(j, k)], acc);
            C[IXM(i, j)] = acc;
        }
}
template <int M>
EPI_DEV void symmetrize(double (&P)[M * M])  // (P + P')/2.0   GenericEKF.m:138,161,226
{
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = j + 1; i < M; i++) {
            double v = (P[IXM(i, j)] + P[IXM(j, i)]) / 2.0;
            P[IXM(i, j)] = v;
            P[IXM(j, i)] = v;
        }
#pragma unroll
    for (int i = 0; i < M; i++) P[IXM(i, i)] = (P[IXM(i, i)] + P[IXM(i, i)]) / 2.0;
}

// ---- model callbacks -----------------------------------------------------
// StateHardMargins: SIAlphaModelEKF.m:27-31, OptControlled.m:27-31, Backward*.m:48-52
template <int M, class PRM>
EPI_DEV void state_hard_margins(const PRM &p, double (&s)[M])
{
    s[0] = fmin(1.0, fmax(p.slo, s[0]));
    s[1] = fmin(1.0, fmax(p.ilo, s[1]));
    s[2] = fmin(p.alpha_max, fmax(p.alpha_min, s[2]));
}

// bang-bang substitution of NaN controls: OptControlled.m:49-58 (strict >), NewCase...m:172-181 (>=)
template <int M, class PRM>
EPI_DEV void resolve_control(const PRM &p, const ModelFlags &mf, const double (&s)[M], double (&u)[kNpi])
{
    if (M == 6) {
        const double gs6 = p.gamma * s[M == 6 ? 5 : 0];
#pragma unroll
        for (int k = 0; k < kNpi; k++) {
            if (is_nan(u[k])) {
                double phi = p.epsilon * p.W(k) - gs6 * p.A(k);
                bool lo = mf.phi_ge ? (phi >= 0.0) : (phi > 0.0);
                u[k] = lo ? p.Umin(k) : p.Umax(k);
            }
        }
    }
}

template <int M, int FLIP, class PRM>
EPI_DEV void state_map(const PRM &p, double dot, const double (&s)[M], double (&sn)[M]);

// NlinStateUpdate: SIAlphaModelEKF.m:39-48, OptControlled.m:39-74, Backward*.m:60-95.
// `u` comes in with NaNs and leaves as the control actually applied (u_opt).
template <int M, int FLIP, class PRM>
EPI_DEV void nlin_state_update(const PRM &p, const ModelFlags &mf, double (&u)[kNpi],
                               const double (&s)[M], double (&sn)[M])
{
    resolve_control<M>(p, mf, s, u);
    // params.gamma * params.a' * (params.u_max - u): row vector (gamma*a') times column
    double dot = (p.gamma * p.A(0)) * (p.Umax(0) - u[0]);
#pragma unroll
    for (int k = 1; k < kNpi; k++) dot = fma(p.gamma * p.A(k), p.Umax(k) - u[k], dot);
    state_map<M, FLIP>(p, dot, s, sn);
}

// the Euler maps themselves, given dot = (gamma*a') * (u_max - u)   (shared by the one-lane and the four-lane kernels)
template <int M, int FLIP, class PRM>
EPI_DEV void state_map(const PRM &p, double dot, const double (&s)[M], double (&sn)[M])
{
    const double asi = s[2] * s[0] * s[1];
    const double f3 = -p.gamma * s[2] + p.gamma * p.b + dot;
    if (!FLIP) {
        sn[0] = fmax(p.slo, fmin(1.0, s[0] - p.dt * s[2] * s[0] * s[1]));
        sn[1] = fmax(p.ilo, fmin(1.0, s[1] + p.dt * (asi - p.beta * s[1])));
        sn[2] = fmax(p.alpha_min, fmin(p.alpha_max, s[2] + p.dt * f3));
    } else {
        sn[0] = fmax(p.slo, fmin(1.0, s[0] + p.dt * s[2] * s[0] * s[1]));
        sn[1] = fmax(p.ilo, fmin(1.0, s[1] - p.dt * (asi - p.beta * s[1])));
        sn[2] = fmax(p.alpha_min, fmin(p.alpha_max, s[2] - p.dt * f3));
    }
    if (M == 6) {
        constexpr int i3 = (M == 6) ? 3 : 0, i4 = (M == 6) ? 4 : 0, i5 = (M == 6) ? 5 : 0;
        const double rho = s[i3] - s[i4] - (1.0 - p.epsilon);
        const double g4 = p.dt * rho * s[2] * s[1];
        const double g5 = p.dt * (rho * s[2] * s[0] + p.beta * s[i4]);
        const double g6 = p.dt * (rho * s[0] * s[1] + p.gamma * s[i5]);
        if (!FLIP) { sn[i3] = s[i3] + g4; sn[i4] = s[i4] + g5; sn[i5] = s[i5] + g6; }
        else       { sn[i3] = s[i3] - g4; sn[i4] = s[i4] - g5; sn[i5] = s[i5] - g6; }
    }
}

// NlinObsUpdate + ObsHardMargins: SIAlphaModelEKF.m:34-36,51-59
template <int M>
EPI_DEV double predict_obs(const ModelFlags &mf, const double (&s)[M], double v_bar)
{
    double v = (mf.obs_type == EPI_OBS_NEWCASES) ? (s[0] * s[1] * s[2] + v_bar) : (1.0 - s[0] + v_bar);
    return mf.obs_clamp ? fmax(0.0, v) : v;
}

// ObsJacobian: SIAlphaModelEKF.m:79-89, OptControlled.m:138-148
template <int M>
EPI_DEV void obs_jacobian(const ModelFlags &mf, const double (&s)[M], double (&C)[M])
{
#pragma unroll
    for (int i = 0; i < M; i++) C[i] = 0.0;
    if (mf.obs_type == EPI_OBS_NEWCASES) {
        C[0] = s[1] * s[2];
        C[1] = s[0] * s[2];
        C[2] = s[0] * s[1];
    } else {
        C[0] = -1.0;
    }
}

// StateJacobians: SIAlphaModelEKF.m:62-76, OptControlled.m:89-135, Backward*.m:83-97 / :109-156.
// `u` is the ORIGINAL control column (NaNs kept), GenericEKF.m:157,206.
// linear-slope term A(3,6) of the bang-bang control, OptControlled.m:107-114 (`u` with its NaNs)
template <int M, int FLIP, class PRM>
EPI_DEV double slope_term(const PRM &p, const double (&u)[kNpi], const double (&s)[M])
{
    double a36 = 0.0;
    if (M == 6) {
        const double dt = p.dt;
        const double gs6 = p.gamma * s[M == 6 ? 5 : 0];
        const double inv_sigma = 1.0 / p.sigma;
#pragma unroll
        for (int k = 0; k < kNpi; k++) {
            if (is_nan(u[k])) {
                double phi = p.epsilon * p.W(k) - gs6 * p.A(k);
                if (phi > -inv_sigma && phi < inv_sigma) {
                    double term = p.gamma * dt * (p.sigma / 2.0) * p.A(k) * (p.Umax(k) - p.Umin(k));
                    a36 = FLIP ? (a36 + term) : (a36 - term);
                }
            }
        }
    }
    return a36;
}
template <int M, int FLIP, class PRM>
EPI_DEV void jacobian_entries(const PRM &p, const double (&s)[M], double a36, double (&A)[M * M]);
template <int M, int FLIP, class PRM>
EPI_DEV void state_jacobians(const PRM &p, const double (&u)[kNpi], const double (&s)[M], double (&A)[M * M])
{
    jacobian_entries<M, FLIP>(p, s, slope_term<M, FLIP>(p, u, s), A);
}
// the entries of the Jacobian given the slope term (shared by the one-lane and the four-lane kernels)
template <int M, int FLIP, class PRM>
EPI_DEV void jacobian_entries(const PRM &p, const double (&s)[M], double a36, double (&A)[M * M])
{
    const double dt = p.dt;
#pragma unroll
    for (int i = 0; i < M * M; i++) A[i] = 0.0;
    if (!FLIP) {
        A[IXM(0, 0)] = 1.0 - dt * s[2] * s[1];
        A[IXM(0, 1)] = -dt * s[2] * s[0];
        A[IXM(0, 2)] = -dt * s[0] * s[1];
        A[IXM(1, 0)] = dt * s[1] * s[2];
        A[IXM(1, 1)] = 1.0 + dt * (s[0] * s[2] - p.beta);
        A[IXM(1, 2)] = dt * s[0] * s[1];
        A[IXM(2, 2)] = 1.0 - dt * p.gamma;
    } else {
        A[IXM(0, 0)] = 1.0 + dt * s[2] * s[1];
        A[IXM(0, 1)] = dt * s[2] * s[0];
        A[IXM(0, 2)] = dt * s[0] * s[1];
        A[IXM(1, 0)] = -dt * s[1] * s[2];
        A[IXM(1, 1)] = 1.0 - dt * (s[0] * s[2] - p.beta);
        A[IXM(1, 2)] = -dt * s[0] * s[1];
        A[IXM(2, 2)] = 1.0 + dt * p.gamma;
    }
    if (M == 6) {
        constexpr int i3 = (M == 6) ? 3 : 0, i4 = (M == 6) ? 4 : 0, i5 = (M == 6) ? 5 : 0;
        A[IXM(2, i5)] = a36;
        const double rho = s[i3] - s[i4] - (1.0 - p.epsilon);
        if (!FLIP) {
            A[IXM(i3, 1)] = dt * s[2] * rho;
            A[IXM(i3, 2)] = dt * s[1] * rho;
            A[IXM(i3, i3)] = 1.0 + dt * s[1] * s[2];
            A[IXM(i3, i4)] = -dt * s[1] * s[2];
            A[IXM(i4, 0)] = dt * s[2] * rho;
            A[IXM(i4, 2)] = dt * s[0] * rho;
            A[IXM(i4, i3)] = dt * s[0] * s[2];
            A[IXM(i4, i4)] = 1.0 - dt * (s[0] * s[2] - p.beta);
            A[IXM(i5, 0)] = dt * s[1] * rho;
            A[IXM(i5, 1)] = dt * s[0] * rho;
            A[IXM(i5, i3)] = dt * s[0] * s[1];
            A[IXM(i5, i4)] = -dt * s[0] * s[1];
            A[IXM(i5, i5)] = 1.0 + dt * p.gamma;
        } else {
            A[IXM(i3, 1)] = -dt * s[2] * rho;
            A[IXM(i3, 2)] = -dt * s[1] * rho;
            A[IXM(i3, i3)] = 1.0 - dt * s[1] * s[2];
            A[IXM(i3, i4)] = dt * s[1] * s[2];
            A[IXM(i4, 0)] = -dt * s[2] * rho;
            A[IXM(i4, 2)] = -dt * s[0] * rho;
            A[IXM(i4, i3)] = -dt * s[0] * s[2];
            A[IXM(i4, i4)] = 1.0 + dt * (s[0] * s[2] - p.beta);
            A[IXM(i5, 0)] = -dt * s[1] * rho;
            A[IXM(i5, 1)] = -dt * s[0] * rho;
            A[IXM(i5, i3)] = -dt * s[0] * s[1];
            A[IXM(i5, i4)] = dt * s[0] * s[1];
            A[IXM(i5, i5)] = 1.0 - dt * p.gamma;
        }
    }
}

// ---- MATLAB pinv of a symmetric matrix ------------------------------------
// pinv.m: svd, tol = max(size(A))*eps(norm(s,inf)), keep s > tol.  Symmetric
// argument => singular triplets from the eigen-decomposition; cyclic Jacobi
// (Rutishauser's formulation with b/z accumulators) on A scaled by a power of 2.
EPI_DEV double eps_of(double x)
{
    if (x == 0.0) return 4.9406564584124654e-324;
    int e = ilogb(x);
    if (e < -1022) return 4.9406564584124654e-324;
    return ldexp(1.0, e - 52);
}

constexpr int kJacobiMaxSweeps = 50;

template <int M>
EPI_DEV void jacobi_rot(double &x, double &y, double s, double tau)
{
    double g = x, h = y;
    x = fma(-s, fma(g, tau, h), g);
    y = fma(s, fma(-h, tau, g), h);
}

EPI_DEV bool jacobi_left_alone(bool dead_p, bool dead_q, double dp, double dq, double apq)
{
    const double dl = fabs(dead_p ? dq : dp);   // the live entry of a mixed pair
    return (dead_p && dead_q) || (dead_p != dead_q && (dl + 100.0 * fabs(apq)) == dl);
}

// a: symmetric, only the upper triangle (i <= j) is read/updated.  Returns true if the sweep cap was hit.
// BZS > 0: the accumulators b and z (Rutishauser) live in LDS, one column per lane with a stride of BZS doubles (bz
// points at this lane's column: b(i) = bz[i*BZS], z(i) = bz[(M+i)*BZS]) -- they are touched twice per rotation and once
// per sweep, and 24 registers less let three 6 x 6 waves share a SIMD without scratch.
template <int M, int BZS = 0>
EPI_DEV bool jacobi_eig(double (&a)[M * M], double (&d)[M], double (&v)[M * M], double *bz = nullptr)
{
    double b[BZS ? 1 : M], z[BZS ? 1 : M];
    auto getb = [&](int i) { return BZS ? bz[i * BZS] : b[BZS ? 0 : i]; };
    auto setb = [&](int i, double x) { if (BZS) bz[i * BZS] = x; else b[BZS ? 0 : i] = x; };
    auto getz = [&](int i) { return BZS ? bz[(M + i) * BZS] : z[BZS ? 0 : i]; };
    auto setz = [&](int i, double x) { if (BZS) bz[(M + i) * BZS] = x; else z[BZS ? 0 : i] = x; };
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = 0; i < M; i++) v[IXM(i, j)] = (i == j) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < M; i++) { d[i] = a[IXM(i, i)]; setb(i, d[i]); setz(i, 0.0); }
    bool capped = true;
    for (int sweep = 1; sweep <= kJacobiMaxSweeps; sweep++) {
        // sym_pinv discards every eigenpair below tol = M*eps(max|d|); an index is "dead" when its diagonal entry is
        // below 2^-10 of that cut-off.  Left alone for this sweep (not rotated, zeroed or counted in the convergence sum;
        // same rule as the oracle): pairs of two dead indices, and dead/live pairs whose 100|a_pq| vanishes against the
        // live diagonal entry -- both judged on the values at the start of the sweep.  X changes by <1e-18 relative,
        // the rotations by -30 %, the pair tests by another -28 %.
        double dmax = 0.0;
#pragma unroll
        for (int i = 0; i < M; i++) dmax = fmax(dmax, fabs(d[i]));
        const double cut = ((double)M * eps_of(dmax)) * 0x1p-10;
        // (6 x 6 only: the 3 x 3 covariances of the SI-alpha filter are practically never that rank deficient, and there
        // the bookkeeping costs more than it saves -- 5.2 vs 5.9 ms on the 307 200-chain ensemble.  Same switch in the oracle.)
        constexpr bool kSkipDiscarded = (M > 3);
        bool dead[M];
#pragma unroll
        for (int i = 0; i < M; i++) dead[i] = kSkipDiscarded && (fabs(d[i]) < cut);
        bool la[M * M];
        double sm = 0.0;
#pragma unroll
        for (int p = 0; p < M - 1; p++)
#pragma unroll
            for (int q = p + 1; q < M; q++) {
                la[IXM(p, q)] = jacobi_left_alone(dead[p], dead[q], d[p], d[q], a[IXM(p, q)]);
                sm = la[IXM(p, q)] ? sm : sm + fabs(a[IXM(p, q)]);
            }
        if (sm == 0.0) { capped = false; break; }
        const double tresh = (sweep < 4) ? 0.2 * sm / (double)(M * M) : 0.0;
#pragma unroll
        for (int p = 0; p < M - 1; p++) {
#pragma unroll
            for (int q = p + 1; q < M; q++) {
                // Same decisions and the same arithmetic as the scalar formulation (oracle/ekf_oracle.c), arranged
                // for a wavefront: the pair is skipped only if NO lane rotates it (wave-uniform branch, no exec-mask
                // bookkeeping); inside, every lane applies a rotation, the identity (t = 0 => c = 1, s = tau = 0,
                // which leaves every operand bit-wise unchanged) for the lanes that do not rotate.
                if (kSkipDiscarded && __builtin_amdgcn_ballot_w64(!la[IXM(p, q)]) == 0ull) continue;   // in play in no lane this sweep
                const double apq = a[IXM(p, q)];
                const double g = 100.0 * fabs(apq);
                const bool live = !la[IXM(p, q)];
                const bool negl = live && sweep > 4 && (fabs(d[p]) + g) == fabs(d[p]) && (fabs(d[q]) + g) == fabs(d[q]);
                const bool rot = live && !negl && (fabs(apq) > tresh);
                if (__builtin_amdgcn_ballot_w64(rot) != 0ull) {
                    const double hd = d[q] - d[p];
                    const bool small = (fabs(hd) + g) == fabs(hd);
                    // t = apq/h, or sgn(theta)/(|theta| + sqrt(theta^2+1)) with theta = h/(2 apq) scaled by |2 apq|
                    const double two_apq = 2.0 * apq;
                    const double w = sqrt(fma(hd, hd, two_apq * two_apq));
                    const double num = small ? apq : two_apq;
                    const double den = small ? hd : (fabs(hd) + w);
                    double t = num / den;
                    if (!small && hd < 0.0) t = -t;
                    t = rot ? t : 0.0;
                    // c = 1/r, tau = s/(1+c) = t/(1+r), r = sqrt(1+t^2): one division serves both
                    const double r = sqrt(fma(t, t, 1.0));
                    const double ir = 1.0 / fma(r, r, r);
                    const double c = (1.0 + r) * ir;
                    const double s = t * c;
                    const double tau = (t * r) * ir;
                    const double h = t * apq;
                    setz(p, getz(p) - h);
                    setz(q, getz(q) + h);
                    d[p] = d[p] - h;
                    d[q] = d[q] + h;
#pragma unroll
                    for (int j = 0; j < p; j++) jacobi_rot<M>(a[IXM(j, p)], a[IXM(j, q)], s, tau);
#pragma unroll
                    for (int j = p + 1; j < q; j++) jacobi_rot<M>(a[IXM(p, j)], a[IXM(j, q)], s, tau);
#pragma unroll
                    for (int j = q + 1; j < M; j++) jacobi_rot<M>(a[IXM(p, j)], a[IXM(q, j)], s, tau);
#pragma unroll
                    for (int j = 0; j < M; j++) jacobi_rot<M>(v[IXM(j, p)], v[IXM(j, q)], s, tau);
                }
                a[IXM(p, q)] = (negl || rot) ? 0.0 : apq;
            }
        }
#pragma unroll
        for (int i = 0; i < M; i++) { const double bi = getb(i) + getz(i); setb(i, bi); d[i] = bi; setz(i, 0.0); }
    }
    return capped;
}

// X = pinv(A) for ANY symmetric A through the two-sided Jacobi eigen-decomposition above; returns the rank kept.
// *capped: Jacobi hit the sweep cap.  sym_pinv (below) hands it the matrices that are not positive semi-definite.
template <int M, int BZS = 0>
EPI_DEV int sym_pinv_two_sided(const double (&A)[M * M], double (&X)[M * M], bool *capped, double *bz = nullptr)
{
    double a[M * M], d[M], v[M * M];
    double amax = 0.0;
#pragma unroll
    for (int i = 0; i < M * M; i++) amax = fmax(amax, fabs(A[i]));
#pragma unroll
    for (int i = 0; i < M * M; i++) X[i] = 0.0;
    *capped = false;
    if (amax == 0.0) return 0;
    const int e = ilogb(amax);
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = 0; i < M; i++) a[IXM(i, j)] = ldexp(A[IXM(i <= j ? i : j, i <= j ? j : i)], -e);
    *capped = jacobi_eig<M, BZS>(a, d, v, bz);
    double smax = 0.0;
#pragma unroll
    for (int i = 0; i < M; i++) smax = fmax(smax, fabs(d[i]));
    const double tol = (double)M * eps_of(smax);
    int rank = 0;
#pragma unroll
    for (int i = 0; i < M; i++) {
        const double sv = fabs(d[i]);
        if (sv > tol) {
            rank++;
            const double inv = 1.0 / sv;
            const double sg = (d[i] < 0.0) ? -1.0 : 1.0;
            // symmetric by construction: accumulate the upper triangle, mirror afterwards
#pragma unroll
            for (int c = 0; c < M; c++)
#pragma unroll
                for (int r = 0; r <= c; r++)
                    X[IXM(r, c)] = fma(v[IXM(r, i)] * inv, sg * v[IXM(c, i)], X[IXM(r, c)]);
        }
    }
#pragma unroll
    for (int c = 0; c < M; c++)
#pragma unroll
        for (int r = 0; r <= c; r++) {
            X[IXM(r, c)] = ldexp(X[IXM(r, c)], -e);
            X[IXM(c, r)] = X[IXM(r, c)];
        }
    return rank;
}

// X = pinv(A), A symmetric -- GenericExtendedKalmanFilter.m:215 applies it to the covariance P(k+1|k).  MATLAB's rule
// (singular values s, tol = max(size(A)) * eps(max(s)), keep s > tol) evaluated for a positive semi-definite argument
// without computing what the rule throws away (45 % of the singular values on the headline sweep; a wavefront's 64
// same-day matrices have the same rank almost everywhere):
//   1. A * 2^-e = G G' + S by a Cholesky factorisation with diagonal pivoting.  The pivot is SELECTED (predicated moves:
//      every index stays compile-time), rows stay in place, column k of G belongs to the k-th pivot.  It ends when the trace
//      of what is left is below 2^-20 of MATLAB's cut-off, or when the pivot column violates a_ip^2 <= a_pp a_ii beyond
//      rounding (rounding noise has taken over).
//   2. one-sided Jacobi rotations orthogonalise the columns of G: the non-zero eigenvalues are the squared column norms,
//      the eigenvectors the normalised columns; no eigenvector matrix is accumulated, 2 - 3.5 sweeps.
//   3. X = sum over the kept columns of g g' / (g'g)^2.
// Operation for operation the oracle's orc_sym_pinv; a lane whose matrix is not positive semi-definite up to rounding is
// flagged and takes sym_pinv_two_sided in the caller (eks_pinv re-reads the matrix for it: rare).  Every loop is wave-uniform:
// a pair / a factorisation step is skipped only if NO lane needs it, lanes that do not rotate apply the identity.
constexpr int kPinvMaxSweeps = 30;
// Au: upper triangle of A, packed (entry (i, j), i <= j, at i + j (j + 1) / 2); Xu: upper triangle of X, packed the same way.
// Returns the rank kept; *indef: this lane's matrix is not positive semi-definite up to rounding and Xu / the rank are NOT
// valid -- the caller runs sym_pinv_two_sided for it.
// lds: this lane's column of an LDS block of NS doubles per lane with a stride of LSTR doubles (scratch of the full-rank
// route: the inverse of the triangular factor waits there while the other lanes of the wavefront iterate, and the result is
// brought back from pivot order through it -- a per-lane permutation is one ds_write with a computed address).
template <int M, int LSTR>
EPI_DEV int sym_pinv_psd(const double (&Au)[M * (M + 1) / 2], double (&Xu)[M * (M + 1) / 2], bool *capped, bool *indef_out,
                         double *lds)
{
    constexpr int NS = M * (M + 1) / 2;
    auto sx = [](int i, int j) constexpr { return i <= j ? i + j * (j + 1) / 2 : j + i * (i + 1) / 2; };
    double amax = 0.0;
#pragma unroll
    for (int i = 0; i < NS; i++) amax = fmax(amax, fabs(Au[i]));
#pragma unroll
    for (int i = 0; i < NS; i++) Xu[i] = 0.0;
    *capped = false;
    *indef_out = false;
    if (amax == 0.0) return 0;
    const int e = ilogb(amax);
    double a[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) a[i] = ldexp(Au[i], -e);
    double dmax0 = 0.0;
#pragma unroll
    for (int i = 0; i < M; i++) dmax0 = fmax(dmax0, a[sx(i, i)]);
    const double noise = (double)M * eps_of(dmax0);
    const double stop = noise * 0x1p-20;
    bool indef = false;
#pragma unroll
    for (int i = 0; i < M; i++) indef = indef || (a[sx(i, i)] < -0.25 * noise);
    double tr0 = 0.0;                   // trace of the scaled matrix: an upper bound of its largest eigenvalue
#pragma unroll
    for (int i = 0; i < M; i++) tr0 = tr0 + a[sx(i, i)];
    double G[M * M];
#pragma unroll
    for (int i = 0; i < M * M; i++) G[i] = 0.0;
    unsigned used = 0u;
    int r = 0;
    unsigned ordw = 0u;                 // the pivot of factorisation step k in bits 3k .. 3k+2
    auto ord = [&](int k) { return (int)((ordw >> (3 * k)) & 7u); };
    bool active = !indef;
#pragma unroll
    for (int k = 0; k < M; k++) {
        if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
        // pivot: the first of the largest diagonal entries among the indices not used yet
        int p = -1;
        double d = 0.0;
#pragma unroll
        for (int i = 0; i < M; i++) {
            const bool take = !((used >> i) & 1u) && (p < 0 || a[sx(i, i)] > d);
            p = take ? i : p;
            d = take ? a[sx(i, i)] : d;
        }
        // its column a(:, p).  The 64 same-day matrices of a wavefront nearly always pick the same pivot: then the column
        // is named at compile time (six register copies); otherwise it is selected entry by entry.
        double colraw[M];
        const int p0 = __builtin_amdgcn_readfirstlane(p);
        if (__builtin_amdgcn_ballot_w64(active && p != p0) == 0ull) {
            bool hit = false;
#pragma unroll
            for (int q = 0; q < M; q++)
                if (!hit && p0 == q) {
                    hit = true;
#pragma unroll
                    for (int i = 0; i < M; i++) colraw[i] = a[sx(i, q)];
                }
            if (!hit) {
#pragma unroll
                for (int i = 0; i < M; i++) colraw[i] = 0.0;      // no lane is active with p0 < 0
            }
        } else {
#pragma unroll
            for (int i = 0; i < M; i++) {
                double v = a[sx(i, 0)];
#pragma unroll
                for (int q = 1; q < M; q++) v = (p == q) ? a[sx(i, q)] : v;
                colraw[i] = v;
            }
        }
        bool quit = !(d * (double)(M - k) > stop);
        if (__builtin_amdgcn_ballot_w64(active && d <= noise) != 0ull) {    // only pivots at the noise scale can be noise
            bool viol = false;
#pragma unroll
            for (int i = 0; i < M; i++) {
                const bool other = !((used >> i) & 1u) && i != p;
                viol = viol || (other && (colraw[i] * colraw[i] > (4.0 * d) * fabs(a[sx(i, i)])));
            }
            quit = quit || (d <= noise && viol);
        }
        if (__builtin_amdgcn_ballot_w64(active && quit) != 0ull) {
            double rest = 0.0;
#pragma unroll
            for (int j = 0; j < M; j++)
#pragma unroll
                for (int i = 0; i <= j; i++) {
                    const bool free_ij = !((used >> i) & 1u) && !((used >> j) & 1u);
                    rest = free_ij ? fmax(rest, fabs(a[sx(i, j)])) : rest;
                }
            indef = indef || (active && quit && rest > 0.25 * noise);
        }
        const bool go = active && !quit;
        const double l = sqrt(d), il = 1.0 / l;
        double col[M];
#pragma unroll
        for (int i = 0; i < M; i++) {
            col[i] = ((used >> i) & 1u) ? 0.0 : colraw[i] * il;
            col[i] = (i == p) ? l : col[i];
            G[IXM(i, k)] = go ? col[i] : 0.0;
        }
        used = go ? (used | (1u << (p & 31))) : used;
        ordw = go ? (ordw | ((unsigned)(p & 7) << (3 * k))) : ordw;
        if (go) lds[k * LSTR] = il;         // 1 / G(ord[k], k): waits in LDS for the full-rank route (12 registers less)
        // Schur complement of the free part.  (Entries of used rows are never read again, and neither is anything of a
        // lane that has stopped: both may take any value, so the update is not predicated.)
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) a[sx(i, j)] = fma(-col[i], col[j], a[sx(i, j)]);
        bool negd = false;
#pragma unroll
        for (int i = 0; i < M; i++) negd = negd || (!((used >> i) & 1u) && a[sx(i, i)] < -0.25 * noise);
        indef = indef || (go && negd);
        active = go && !negd;
        r = go ? k + 1 : r;
    }
    // Full rank, certified: every index has been a pivot and 1 / trace(inv(A)) -- a lower bound of the smallest eigenvalue;
    // trace(inv(A)) is the squared Frobenius norm of the inverse of the triangular factor -- lies four cut-offs above the
    // cut-off.  Then pinv(A) = inv(A) = P' (inv(Lp)' inv(Lp)) P with Lp the rows of G in pivot order (see the oracle):
    // no iteration.  inv(Lp) is formed here, while G is at hand, and parked in LDS; the product follows at the end.
    constexpr auto lt = [](int i, int j) constexpr { return i * (i + 1) / 2 + j; };      // lower triangle, packed by rows
    bool cert = false;
    {
        const bool full = !indef && r == M;
        if (__builtin_amdgcn_ballot_w64(full) != 0ull) {
            // row by row: row i of Lp is gathered (a wavefront's matrices nearly always share the pivot order: one of the six
            // candidate rows is in play, the others are skipped) and consumed into row i of the inverse at once
            double Li[NS], ilv[M];
#pragma unroll
            for (int k = 0; k < M; k++) ilv[k] = lds[k * LSTR];      // (garbage in the lanes that are not `full`: unused)
#pragma unroll
            for (int i = 0; i < M; i++) {
                double Lr[M];
#pragma unroll
                for (int c = 0; c < M; c++) Lr[c] = 0.0;
#pragma unroll
                for (int q = 0; q < M; q++) {
                    const bool hit = full && ord(i) == q;
                    if (__builtin_amdgcn_ballot_w64(hit) == 0ull) continue;
#pragma unroll
                    for (int c = 0; c < i; c++) Lr[c] = hit ? G[IXM(q, c)] : Lr[c];
                }
#pragma unroll
                for (int j = 0; j < i; j++) {
                    double acc = Lr[j] * Li[lt(j, j)];
#pragma unroll
                    for (int k = j + 1; k < i; k++) acc = fma(Lr[k], Li[lt(k, j)], acc);
                    Li[lt(i, j)] = -(acc * ilv[i]);
                }
                Li[lt(i, i)] = ilv[i];
            }
            double fro = 0.0;
#pragma unroll
            for (int j = 0; j < M; j++)
#pragma unroll
                for (int i = j; i < M; i++) fro = fma(Li[lt(i, j)], Li[lt(i, j)], fro);
            cert = full && (4.0 * ((double)M * eps_of(tr0)) * fro < 1.0);
            if (__builtin_amdgcn_ballot_w64(cert) != 0ull) {
#pragma unroll
                for (int i = 0; i < NS; i++) lds[i * LSTR] = Li[i];
            }
        }
    }
    // one-sided Jacobi on the r columns of G.  The squared column norms are kept beside G and formed anew (same fma chain
    // the oracle runs at every test) for the two columns a rotation has touched.
    double nrm[M];
#pragma unroll
    for (int k = 0; k < M; k++) {
        double s2 = 0.0;
#pragma unroll
        for (int i = 0; i < M; i++) s2 = fma(G[IXM(i, k)], G[IXM(i, k)], s2);
        nrm[k] = s2;
    }
    bool done = indef || cert || r < 2;
    bool cap = false;
    for (int sweep = 1; sweep <= kPinvMaxSweeps; sweep++) {
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < M - 1; p++) {
#pragma unroll
            for (int q = p + 1; q < M; q++) {
                const bool in = !done && q < r;
                if (__builtin_amdgcn_ballot_w64(in) == 0ull) continue;
                const double al = nrm[p], be = nrm[q];
                double ga = 0.0;
#pragma unroll
                for (int i = 0; i < M; i++) ga = fma(G[IXM(i, p)], G[IXM(i, q)], ga);
                const bool rot = in && (ga * ga > 0x1p-106 * (al * be));
                if (__builtin_amdgcn_ballot_w64(rot) == 0ull) continue;
                rotated = rotated || rot;
                // cos = (|h| + w) / D, sin = 2 ga / D, D = sqrt(2 w (|h| + w)) (see the oracle); the identity (cos = 1,
                // sin = 0) for the lanes that do not rotate
                const double h = be - al, two = 2.0 * ga;
                const double w = sqrt(fma(h, h, two * two)), sum = fabs(h) + w;
                const double iD = 1.0 / sqrt((2.0 * w) * sum);
                const double c = rot ? sum * iD : 1.0;
                double s = two * iD;
                s = (h < 0.0) ? -s : s;
                s = rot ? s : 0.0;
                double np = 0.0, nq = 0.0;
#pragma unroll
                for (int i = 0; i < M; i++) {
                    const double gp = G[IXM(i, p)], gq = G[IXM(i, q)];
                    const double xp = fma(c, gp, -(s * gq)), xq = fma(s, gp, c * gq);
                    G[IXM(i, p)] = xp;
                    G[IXM(i, q)] = xq;
                    np = fma(xp, xp, np);
                    nq = fma(xq, xq, nq);
                }
                nrm[p] = np;
                nrm[q] = nq;
            }
        }
        cap = cap || (!done && rotated && sweep == kPinvMaxSweeps);
        done = done || !rotated;
    }
    double lam[M], lmax = 0.0;
#pragma unroll
    for (int k = 0; k < M; k++) {
        lam[k] = (k < r) ? nrm[k] : 0.0;
        lmax = fmax(lmax, lam[k]);
    }
    const double tol = (double)M * eps_of(lmax);
    int rank = 0;
#pragma unroll
    for (int k = 0; k < M; k++) {
        const bool keep = !cert && k < r && lam[k] > tol;
        if (__builtin_amdgcn_ballot_w64(keep) == 0ull) continue;
        rank += keep ? 1 : 0;
        const double w = keep ? 1.0 / (lam[k] * lam[k]) : 0.0;
        // (w = 0 for the lanes that do not keep column k: G w = 0 and fma(0, g, x) = x, finite operands)
#pragma unroll
        for (int c = 0; c < M; c++)
#pragma unroll
            for (int rr = 0; rr <= c; rr++) Xu[sx(rr, c)] = fma(G[IXM(rr, k)] * w, G[IXM(c, k)], Xu[sx(rr, c)]);
    }
    if (__builtin_amdgcn_ballot_w64(cert) != 0ull) {
        // the certified lanes: Xp = inv(Lp)' inv(Lp) in pivot order, then X(ord[i], ord[j]) = Xp(i, j) through LDS
        double Li[NS], Xp[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) Li[i] = lds[i * LSTR];
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) {
                double acc = Li[lt(j, i)] * Li[lt(j, j)];
#pragma unroll
                for (int k = j + 1; k < M; k++) acc = fma(Li[lt(k, i)], Li[lt(k, j)], acc);
                Xp[sx(i, j)] = acc;
            }
        if (cert) {
#pragma unroll
            for (int j = 0; j < M; j++)
#pragma unroll
                for (int i = 0; i <= j; i++) {
                    const int oi = ord(i), oj = ord(j);
                    const int oa = oi < oj ? oi : oj, ob = oi < oj ? oj : oi;
                    lds[(oa + ob * (ob + 1) / 2) * LSTR] = Xp[sx(i, j)];
                }
#pragma unroll
            for (int i = 0; i < NS; i++) Xu[i] = lds[i * LSTR];
            rank = M;
        }
    }
#pragma unroll
    for (int i = 0; i < NS; i++) Xu[i] = ldexp(Xu[i], -e);
    *capped = cap;
    *indef_out = indef;
    return rank;
}

// ---- MATLAB mrdivide, square right operand: X = Bm / A = (A' \ Bm')' -------
// LAPACK dgetf2 + dgetrs operation order (first-max partial pivoting, reciprocal
// scaling of the sub-column, column-oriented triangular solves).  All indices are
// compile-time; the pivot row is chosen with predicated swaps.
template <int M>
EPI_DEV void mrdivide(const double (&Bm)[M * M], const double (&A)[M * M], double (&X)[M * M])
{
    double Mt[M * M], Y[M * M];
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = 0; i < M; i++) { Mt[IXM(i, j)] = A[IXM(j, i)]; Y[IXM(i, j)] = Bm[IXM(j, i)]; }
#pragma unroll
    for (int j = 0; j < M; j++) {
        // idamax over rows j..M-1 of column j: first index of the maximum |.|
        int piv = j;
        double best = fabs(Mt[IXM(j, j)]);
#pragma unroll
        for (int i = j + 1; i < M; i++) {
            double vv = fabs(Mt[IXM(i, j)]);
            if (vv > best) { best = vv; piv = i; }
        }
        // pivot value (selected without dynamic indexing)
        double pval = Mt[IXM(j, j)];
#pragma unroll
        for (int i = j + 1; i < M; i++) pval = (piv == i) ? Mt[IXM(i, j)] : pval;
        const bool nz = (pval != 0.0);
        // swap rows j <-> piv of the matrix (all columns) and of the right-hand sides (dlaswp)
#pragma unroll
        for (int i = j + 1; i < M; i++) {
            const bool sw = (piv == i);
            // the chains of a wave (one region, neighbouring cost weights) mostly agree on the pivot row: a candidate
            // row that no lane picks costs one ballot instead of 4*M selects
            if (__builtin_amdgcn_ballot_w64(sw) == 0ull) continue;
#pragma unroll
            for (int c = 0; c < M; c++) {
                // dgetf2 swaps only when the pivot is non-zero; dgetrs applies ipiv regardless
                double mj = Mt[IXM(j, c)], mi = Mt[IXM(i, c)];
                Mt[IXM(j, c)] = (sw && nz) ? mi : mj;
                Mt[IXM(i, c)] = (sw && nz) ? mj : mi;
                double yj = Y[IXM(j, c)], yi = Y[IXM(i, c)];
                Y[IXM(j, c)] = sw ? yi : yj;
                Y[IXM(i, c)] = sw ? yj : yi;
            }
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
    for (int c = 0; c < M; c++) {
#pragma unroll
        for (int k = 0; k < M; k++) {
            if (Y[IXM(k, c)] != 0.0) {
#pragma unroll
                for (int i = k + 1; i < M; i++) Y[IXM(i, c)] = Y[IXM(i, c)] - Y[IXM(k, c)] * Mt[IXM(i, k)];
            }
        }
#pragma unroll
        for (int k = M - 1; k >= 0; k--) {
            if (Y[IXM(k, c)] != 0.0) {
                Y[IXM(k, c)] = Y[IXM(k, c)] / Mt[IXM(k, k)];
#pragma unroll
                for (int i = 0; i < k; i++) Y[IXM(i, c)] = Y[IXM(i, c)] - Y[IXM(k, c)] * Mt[IXM(i, k)];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < M; j++)
#pragma unroll
        for (int i = 0; i < M; i++) X[IXM(i, j)] = Y[IXM(j, i)];
}

}  // namespace epi
