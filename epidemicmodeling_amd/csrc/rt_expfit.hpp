// Tools/Rt_ExpFitEKF.m:1-227 -- 2-state exponential-fit EKF / fixed-interval smoother with the second-order
// (Hessian trace) terms of `order == 2`; included by epiekf.hip.
//
// Same execution shape as the SI-alpha kernels: one lane per chain, arrays [T][rows][B] chain-minor, the three
// innovation windows as LDS ring buffers.  State and covariance are 2 + 4 doubles, so the kernels are bound by
// their stores (fwd 128 B/step) and loads (bwd 96 B/step); occupancy is limited by the LDS windows only.
// exp / tanh are epi_exp / epi_tanh (ekf_device.hpp; the oracle evaluates the same sequence), so parity with the CPU
// oracle is bit for bit here too.
#pragma once

struct RtArgs {
    int B, T, Sx, L, order;
    const int32_t *x_series;
    const double *x, *rp;
    double *S_MINUS, *S_PLUS, *P_MINUS, *P_PLUS;          // never NULL (the smoother reads them back)
    double *K_GAIN, *S_SMOOTH, *P_SMOOTH, *innovations, *rho;   // NULL = not stored
};

struct RtModel { double ts, alpha, sigma, w1, w2; };

// NlinStateUpdate :133-140 pieces and StateJacobians :143-160 at s
EPI_DEV void rt_jacobian(const RtModel &m, const double (&s)[2], double (&A)[4], double &E, double &tnh, double &omt)
{
    constexpr int M = 2;
    E = epi_exp(m.ts * s[1]);
    tnh = epi_tanh((m.alpha * s[1] + m.w2) / m.sigma);
    omt = 1.0 - tnh * tnh;
    A[IXM(0, 0)] = E; A[IXM(0, 1)] = (m.ts * s[0]) * E; A[IXM(1, 0)] = 0.0; A[IXM(1, 1)] = m.alpha * omt;
}

// trace terms of StateHessianTerms :176-199 for one list {F1, F2}:  f(ii) = trace(Pk*F{ii})/2,
// Cm(ii,jj) = trace(Pk*F{ii}*Pk*F{jj})/2 with the products taken left to right
EPI_DEV void rt_hessian_terms(const double (&Pk)[4], const double (&F1)[4], const double (&F2)[4], double (&f)[2],
                              double (&Cm)[4])
{
    constexpr int M = 2;
    double T1[4], T2[4], T3[4];
    mat_mul<2>(Pk, F1, T1);
    f[0] = (T1[0] + T1[3]) / 2.0;
    mat_mul<2>(T1, Pk, T2);
    mat_mul<2>(T2, F1, T3); Cm[IXM(0, 0)] = (T3[0] + T3[3]) / 2.0;
    mat_mul<2>(T2, F2, T3); Cm[IXM(0, 1)] = (T3[0] + T3[3]) / 2.0;
    mat_mul<2>(Pk, F2, T1);
    f[1] = (T1[0] + T1[3]) / 2.0;
    mat_mul<2>(T1, Pk, T2);
    mat_mul<2>(T2, F1, T3); Cm[IXM(1, 0)] = (T3[0] + T3[3]) / 2.0;
    mat_mul<2>(T2, F2, T3); Cm[IXM(1, 1)] = (T3[0] + T3[3]) / 2.0;
}

__global__ __launch_bounds__(kWave) void rt_expfit_fwd(const RtArgs a)
{
    constexpr int M = 2;
    extern __shared__ double lds[];   // three sliding windows [3][L][64], one column per lane
    const int lane = threadIdx.x;
    const int c = blockIdx.x * kWave + lane;
    if (c >= a.B) return;
    const int B = a.B, T = a.T, L = a.L;
    const int sx = a.x_series ? a.x_series[c] : c;
    const Lay lay = lay_classic(B, c);   // this function's arrays are plain [T][rows][B]
    auto g = [&](int f) { return a.rp[(size_t)f * B + c]; };
    const RtModel mo = {g(EPI_RT_TIME_SCALE), g(EPI_RT_ALPHA), g(EPI_RT_SIGMA), g(EPI_RT_W_BAR), g(EPI_RT_W_BAR + 1)};
    const double v_bar = g(EPI_RT_V_BAR), beta = g(EPI_RT_BETA_EKF), gamma = g(EPI_RT_GAMMA_EKF);
    double R = g(EPI_RT_R_V);                                  // :32  (running scalar, :99-101)
    double sm[2] = {g(EPI_RT_S_INIT), g(EPI_RT_S_INIT + 1)}, Pm[4], Q[4];
#pragma unroll
    for (int e = 0; e < 4; e++) { Pm[e] = g(EPI_RT_PS_INIT + e); Q[e] = g(EPI_RT_Q_W + e); }
    double *winMean = lds + lane, *winCov = lds + (size_t)L * kWave + lane, *winCovN = lds + (size_t)2 * L * kWave + lane;
    for (int j = 0; j < L; j++) { winMean[j * kWave] = 0.0; winCov[j * kWave] = 0.0; winCovN[j * kWave] = 0.0; }
    int head = 0;
    const double Cj[2] = {1.0, 0.0}, Dj = 1.0;                  // ObsJacobian :151-154

    double xk = a.x[sx];
    for (int k = 0; k < T; k++) {
        const double xn = (k + 1 < T) ? a.x[(size_t)(k + 1) * a.Sx + sx] : 0.0;   // next step's input, ahead of the stores
        store_vec<2>(a.S_MINUS, k, lay, sm);                   // :37-38
        store_mat<2>(a.P_MINUS, k, lay, Pm);
        // ObsHessianTerms :202-227: Gs = Gv = {0}, so gs, Gsp, gv, Gvp are 0 for either order
        const double xk_minus = ((sm[0] + v_bar) + 0.0) + 0.0;  // :52
        double innov, K[2], sp[2], Pp[4];
        const bool valid = !is_nan(xk);
        if (valid) {                                            // :55-59
            innov = xk - xk_minus;
            double PCt[2], CP[2];
#pragma unroll
            for (int i = 0; i < 2; i++) PCt[i] = fma(Pm[IXM(i, 1)], Cj[1], Pm[IXM(i, 0)] * Cj[0]);
#pragma unroll
            for (int j = 0; j < 2; j++) CP[j] = fma(Cj[1], Pm[IXM(1, j)], Cj[0] * Pm[IXM(0, j)]);
            const double CPCt = fma(CP[1], Cj[1], CP[0] * Cj[0]);
            const double den = ((CPCt + gamma * ((Dj * R) * Dj)) + 0.0) + 0.0;
            K[0] = PCt[0] / den; K[1] = PCt[1] / den;
            double IKC[4], T1[4];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 2; i++) IKC[IXM(i, j)] = ((i == j) ? 1.0 : 0.0) - K[i] * Cj[j];
            mat_mul<2>(IKC, Pm, T1);
#pragma unroll
            for (int e = 0; e < 4; e++) Pp[e] = T1[e] / gamma;
            sp[0] = sm[0] + K[0] * innov; sp[1] = sm[1] + K[1] * innov;
        } else {                                                // :60-65
            innov = 0.0; K[0] = 0.0; K[1] = 0.0;
#pragma unroll
            for (int e = 0; e < 4; e++) Pp[e] = Pm[e];
            sp[0] = sm[0]; sp[1] = sm[1];
        }
        double A[4], E, tnh, omt;
        rt_jacobian(mo, sp, A, E, tnh, omt);
        double fs[2] = {0.0, 0.0}, fw[2] = {0.0, 0.0}, Fsp[4] = {0.0, 0.0, 0.0, 0.0}, Fwp[4] = {0.0, 0.0, 0.0, 0.0};
        if (a.order == 2) {                                     // StateHessianTerms :163-199
            double Fs1[4] = {0.0, 0.0, 0.0, 0.0}, Fs2[4] = {0.0, 0.0, 0.0, 0.0};
            const double Fw1[4] = {0.0, 0.0, 0.0, 0.0};
            double Fw2[4] = {0.0, 0.0, 0.0, 0.0};
            Fs1[IXM(0, 1)] = mo.ts * E; Fs1[IXM(1, 0)] = Fs1[IXM(0, 1)];
            Fs1[IXM(1, 1)] = ((mo.ts * mo.ts) * sp[0]) * E;
            Fs2[IXM(1, 1)] = (((-2.0 * (mo.alpha * mo.alpha)) / mo.sigma) * tnh) * omt;
            Fw2[IXM(1, 1)] = ((-2.0 / mo.sigma) * tnh) * omt;
            rt_hessian_terms(Pp, Fs1, Fs2, fs, Fsp);
            rt_hessian_terms(Q, Fw1, Fw2, fw, Fwp);
        }
        sm[0] = ((sp[0] * E + mo.w1) + fs[0]) + fw[0];          // :81
        sm[1] = ((mo.sigma * tnh) + fs[1]) + fw[1];
        {
            const double Bm[4] = {1.0, 0.0, 0.0, omt};
            double T1[4], T2[4], T3[4];
            mat_mul<2>(A, Pp, T1); mat_mul_bt<2>(T1, A, T2);
            mat_mul<2>(Bm, Q, T1); mat_mul_bt<2>(T1, Bm, T3);
#pragma unroll
            for (int e = 0; e < 4; e++) Pm[e] = ((T2[e] + T3[e]) + Fsp[e]) + Fwp[e];   // :83
        }
        store_vec<2>(a.S_PLUS, k, lay, sp);                    // :86-88
        store_mat<2>(a.P_PLUS, k, lay, Pp);
        store_vec<2>(a.K_GAIN, k, lay, K);
        if (a.innovations) a.innovations[(size_t)k * B + c] = innov;
        // :91-101  windows are newest-first and summed front to back
        const int cnt = (k + 1 < L) ? (k + 1) : L;
        head = (head == 0) ? (L - 1) : (head - 1);
        winMean[head * kWave] = innov;
        const double mu = ring_sum(winMean, head, L, innov) / (double)cnt;
        const double cc = (innov - mu) * (innov - mu);
        const double ccn = cc / R;
        winCov[head * kWave] = cc;
        winCovN[head * kWave] = ccn;
        const double sumN = ring_sum(winCovN, head, L, ccn);
        if (a.rho) a.rho[(size_t)k * B + c] = sumN / (double)cnt;
        if (beta != 1.0 && valid) R = beta * R + (1.0 - beta) * ring_sum(winCov, head, L, cc) / (double)cnt;
        xk = xn;
    }
}

// :104-116  (no end-point constraints, no clamps, mrdivide on the 2 x 2)
__global__ __launch_bounds__(256) void rt_expfit_bwd(const RtArgs a)
{
    constexpr int M = 2;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.B) return;
    const int B = a.B, T = a.T;
    const Lay lay = lay_classic(B, c);
    auto g = [&](int f) { return a.rp[(size_t)f * B + c]; };
    const RtModel mo = {g(EPI_RT_TIME_SCALE), g(EPI_RT_ALPHA), g(EPI_RT_SIGMA), g(EPI_RT_W_BAR), g(EPI_RT_W_BAR + 1)};
    double Ss[2], Ps[4];
    load_vec<2>(a.S_PLUS, T - 1, lay, Ss);
    load_mat<2>(a.P_PLUS, T - 1, lay, Ps);
    store_vec<2>(a.S_SMOOTH, T - 1, lay, Ss);
    store_mat<2>(a.P_SMOOTH, T - 1, lay, Ps);
    for (int k = T - 2; k >= 0; k--) {
        double sp[2], Pp[4], Sm1[2], Pm1[4];
        load_vec<2>(a.S_PLUS, k, lay, sp);
        load_mat<2>(a.P_PLUS, k, lay, Pp);
        load_vec<2>(a.S_MINUS, k + 1, lay, Sm1);
        load_mat<2>(a.P_MINUS, k + 1, lay, Pm1);
        double A[4], E, tnh, omt, T1[4], J[4], D[4], T2[4];
        rt_jacobian(mo, sp, A, E, tnh, omt);
        mat_mul_bt<2>(Pp, A, T1);
        mrdivide<2>(T1, Pm1, J);                                // :112
        const double d0 = Ss[0] - Sm1[0], d1 = Ss[1] - Sm1[1];
        Ss[0] = sp[0] + fma(J[IXM(0, 1)], d1, J[IXM(0, 0)] * d0);
        Ss[1] = sp[1] + fma(J[IXM(1, 1)], d1, J[IXM(1, 0)] * d0);
#pragma unroll
        for (int e = 0; e < 4; e++) D[e] = Pm1[e] - Ps[e];
        mat_mul<2>(J, D, T1); mat_mul_bt<2>(T1, J, T2);
#pragma unroll
        for (int e = 0; e < 4; e++) Ps[e] = Pp[e] - T2[e];
        store_vec<2>(a.S_SMOOTH, k, lay, Ss);
        store_mat<2>(a.P_SMOOTH, k, lay, Ps);
    }
}
