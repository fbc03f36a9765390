/* selftest.c -- sanitizer harness for the CPU oracle (TEST INFRASTRUCTURE).  Built with
 * -fsanitize=address,undefined by `make selftest_asan` and run by tests/test_oracle.py: every model variant,
 * the batched driver, pinv / mrdivide and the simulators on small deterministic inputs; any out-of-bounds
 * access, use of uninitialised heap, signed overflow ... aborts the run.  (GPU AddressSanitizer is not
 * available on the pool, so the sanitizers cover the CPU build only.) */
#include "ekf_oracle.h"
#include "../include/epiekf_layout.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

static double lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (double)(*s >> 8) / 16777216.0; }

int main(void)
{
    unsigned seed = 12345u;
    const int T = 37, B = 5, S = 2, n = 12, L = 21;
    for (int model = 0; model < 6; model++) {
        const int m = orc_model_dim(model), mm = m * m;
        double *x = malloc(sizeof(double) * T * S), *u = malloc(sizeof(double) * T * n * S), *R = malloc(sizeof(double) * T * S);
        double *prm = calloc((size_t)EPI_PRM_COUNT * B, sizeof(double));
        double *si = malloc(sizeof(double) * m * B), *sf = malloc(sizeof(double) * m * B);
        double *Pi = calloc((size_t)mm * B, sizeof(double)), *Pf = malloc(sizeof(double) * mm * B), *Q = calloc((size_t)mm * B, sizeof(double));
        double *Rs = malloc(sizeof(double) * B);
        int *xs = malloc(sizeof(int) * B);
        for (int i = 0; i < T * S; i++) { x[i] = (i % 11 == 7) ? NAN : 1e-5 * lcg(&seed); R[i] = 1e-12 * (1 + lcg(&seed)); }
        for (int i = 0; i < T * n * S; i++) u[i] = (i > T * n * S * 2 / 3 && m == 6) ? NAN : floor(3 * lcg(&seed));
        for (int c = 0; c < B; c++) {
            xs[c] = c % S; Rs[c] = 1e-10;
            double *p = prm; 
#define P_(f) p[(size_t)(f) * B + c]
            P_(EPI_PRM_DT) = 1; P_(EPI_PRM_BETA) = 0.2; P_(EPI_PRM_GAMMA) = 1.0 / 7; P_(EPI_PRM_SIGMA) = 1e4; P_(EPI_PRM_B) = 0.01;
            P_(EPI_PRM_EPSILON) = 0.1 * (c + 1); P_(EPI_PRM_S_MIN) = 1e-6; P_(EPI_PRM_I_MIN) = 1e-6; P_(EPI_PRM_ALPHA_MIN) = 1e-8;
            P_(EPI_PRM_ALPHA_MAX) = 100; P_(EPI_PRM_V_BAR) = 0; P_(EPI_PRM_BETA_EKF) = (model >= 4) ? 0.9 : 1.0; P_(EPI_PRM_GAMMA_EKF) = 0.995;
            for (int k = 0; k < n; k++) { P_(EPI_PRM_A + k) = 0.01 * lcg(&seed); P_(EPI_PRM_U_MAX + k) = 3; P_(EPI_PRM_W_EFF + k) = 1; }
#undef P_
            for (int i = 0; i < m; i++) { si[(size_t)i * B + c] = (i == 0) ? 0.99 : (i == 1) ? 0.01 : (i == 2) ? 1.1 : 0.0; sf[(size_t)i * B + c] = (i >= 3) ? 0.0 : NAN; }
            for (int i = 0; i < mm; i++) Pf[(size_t)i * B + c] = NAN;
            for (int d = 0; d < m; d++) { Pi[(size_t)(d * m + d) * B + c] = 1e-4; Q[(size_t)(d * m + d) * B + c] = 1e-8; }
            if (model == 2 || model == 3) for (int i = 0; i < m; i++) { sf[(size_t)i * B + c] = si[(size_t)i * B + c]; for (int j = 0; j < m; j++) Pf[(size_t)(i + m * j) * B + c] = (i == j) ? 1e-4 : 0.0; }
        }
        orc_batch bt = {0};
        bt.model = model; bt.B = B; bt.T = T; bt.Sx = S; bt.Su = S; bt.n_npi = n; bt.L = L; bt.order = 1; bt.obs_type = 0;
        bt.r_mode = (model >= 4) ? 0 : 1;
        bt.x_series_of_chain = xs; bt.u_series_of_chain = xs; bt.x = x; bt.u = u; bt.R_series = R; bt.R_scalar = Rs; bt.prm = prm;
        bt.s_init = si; bt.Ps_init = Pi; bt.s_final = sf; bt.Ps_final = Pf; bt.Q = Q;
        bt.u_opt = malloc(sizeof(double) * T * n * B); bt.u_opt_smooth = malloc(sizeof(double) * T * n * B);
        bt.S_MINUS = malloc(sizeof(double) * T * m * B); bt.S_PLUS = malloc(sizeof(double) * T * m * B); bt.S_SMOOTH = malloc(sizeof(double) * T * m * B);
        bt.P_MINUS = malloc(sizeof(double) * T * mm * B); bt.P_PLUS = malloc(sizeof(double) * T * mm * B); bt.P_SMOOTH = malloc(sizeof(double) * T * mm * B);
        bt.K_GAIN = malloc(sizeof(double) * T * m * B); bt.innovations = malloc(sizeof(double) * T * B); bt.rho = malloc(sizeof(double) * T * B);
        bt.pinv_rank = malloc(sizeof(int) * T * B);
        int rc = orc_ekf_run_batch(&bt, 1);
        if (rc != 0) { printf("model %d failed rc=%d\n", model, rc); return 1; }
        double chk = 0;
        for (int i = 0; i < T * m * B; i++) if (isfinite(bt.S_SMOOTH[i])) chk += bt.S_SMOOTH[i];
        printf("model %d ok, checksum %.6e\n", model, chk);
        free(x); free(u); free(R); free(prm); free(si); free(sf); free(Pi); free(Pf); free(Q); free(Rs); free(xs);
        free(bt.u_opt); free(bt.u_opt_smooth); free(bt.S_MINUS); free(bt.S_PLUS); free(bt.S_SMOOTH); free(bt.P_MINUS);
        free(bt.P_PLUS); free(bt.P_SMOOTH); free(bt.K_GAIN); free(bt.innovations); free(bt.rho); free(bt.pinv_rank);
    }
    /* simulators */
    {
        const int K = 50;
        double u[12 * 50], a[12], um[12], s[50], i[50], al[50], z[150];
        for (int k = 0; k < 12 * K; k++) u[k] = floor(3 * lcg(&seed));
        for (int k = 0; k < 12; k++) { a[k] = 0.01; um[k] = 3; }
        for (int k = 0; k < 3 * K; k++) z[k] = lcg(&seed) - 0.5;
        orc_sialpha_controlled(u, 12, 0.99, 0.01, 1.1, um, 1e-8, 100, 1.0 / 7, a, 0.0, 0.2, 1e-4, 1e-4, 1e-3, K, 1.0, z, s, i, al);
        double J0, J1, w[12 * 50];
        for (int k = 0; k < 12 * K; k++) w[k] = 1.0;
        double nc[50];
        for (int k = 0; k < K; k++) nc[k] = s[k] * i[k] * al[k];
        orc_npi_cost(nc, u, w, 12, K, &J0, &J1);
        double par[7][60], out[5][60];
        for (int q = 0; q < 7; q++) for (int t = 0; t < 60; t++) par[q][t] = 0.05 * (q + 1);
        orc_seirp(par[0], par[1], par[2], par[3], par[4], par[5], par[6], 0.99, 0.01, 0, 0, 0, 60, 0.1, out[0], out[1], out[2], out[3], out[4]);
        orc_seirp_saturated(par[0], par[1], par[2], par[3], par[6], 0.99, 0.01, 0, 0, 0, 60, 0.1, 0.1, 0.02, 0.02, 0.08, 0.01, 0.05, out[0], out[1], out[2], out[3], out[4]);
        printf("sims ok J0=%.3e J1=%.3e\n", J0, J1);
    }
    /* scenario generation / selection and Rt_ExpFitEKF */
    {
        const int K = 30, P = 40;
        double lo[12], hi[12], plan[12 * 30], J0[40], J1[40];
        int on[40], io;
        for (int k = 0; k < 12; k++) { lo[k] = 0; hi[k] = 2 + (k % 3); }
        orc_random_npi_plan(7u, 9u, 3, 0, 10, 12, K, lo, hi, plan);
        orc_random_npi_plan(7u, 9u, 3, 9, 10, 12, K, lo, hi, plan);
        for (int q = 0; q < P; q++) { J0[q] = lcg(&seed); J1[q] = lcg(&seed); }
        J0[5] = NAN;
        orc_pareto_front(P, J0, J1, on, &io);
        orc_pareto_front(1, J0, J1, on, &io);
        const int Tr = 45, Br = 3;
        double xr[45 * 2], rp[19 * 3], o2[3][45 * 2 * 3], o4[3][45 * 4 * 3], kg[45 * 2 * 3], inn[45 * 3], rho[45 * 3];
        int xsr[3] = {0, 1, 0};
        for (int t = 0; t < Tr * 2; t++) xr[t] = (t % 13 == 5) ? NAN : 500 + 100 * lcg(&seed);
        for (int c = 0; c < Br; c++) {
            const double col[19] = {1, 0.9, 0.1, 0.1, 1e-4, 0, 100, 0.9, 0.995, 500, 0.01, 6.25e6, 0, 0, 9e-4, 62500, 0, 0, 9e-6};
            for (int f = 0; f < 19; f++) rp[f * Br + c] = col[f];
        }
        for (int order = 1; order <= 2; order++) {
            int rc = orc_rt_expfit_batch(Br, Tr, 2, xsr, xr, rp, 21, order, o2[0], o2[1], o4[0], o4[1], kg, o2[2], o4[2], inn, rho, 1);
            if (rc != 0) { printf("rt_expfit order %d failed rc=%d\n", order, rc); return 1; }
        }
        if (orc_rt_expfit_batch(Br, Tr, 2, xsr, xr, rp, 21, 3, o2[0], o2[1], o4[0], o4[1], kg, o2[2], o4[2], inn, rho, 1) != ORC_ERR_UNDEFINED_ORDER) return 1;
        printf("scenarios + rt_expfit ok, i_opt=%d S_SMOOTH[0]=%.6e\n", io, o2[2][0]);
    }
    return 0;
}
