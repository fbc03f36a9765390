/* c_abi_demo.c -- the drop-in boundary from plain C, no Python / PyTorch involved.
 *
 * One call of the reference's SIAlphaModelEKFOptControlled (Tools/SIAlphaModelEKFOptControlled.m:1) through
 * epi_ekf_run_host with B = 1: every array has MATLAB's column-major layout (u: 12 x T, S: 6 x T, P: 6 x 6 x T),
 * exactly what a MEX gateway passes (matlab/epiekf_mex.cpp).  Inputs are read from a small binary file written
 * by tests/test_gpu_parity.py so that the test can compare the outputs with the Python path bit for bit.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -o examples/c_abi_demo -Lepidemicmodeling_amd -lepiekf \
 *       -Wl,-rpath,'$ORIGIN/../epidemicmodeling_amd'
 *   examples/c_abi_demo in.bin out.bin
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "epiekf.h"

static double *rd(FILE *f, size_t n)
{
    double *p = (double *)malloc(n * sizeof(double));
    if (fread(p, sizeof(double), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int hdr[4];                                   /* T, n_npi, L, order */
    if (fread(hdr, sizeof(int), 4, f) != 4) return 2;
    const int T = hdr[0], n = hdr[1], m = 6;
    double *u = rd(f, (size_t)n * T), *x = rd(f, T), *R = rd(f, T), *prm = rd(f, EPI_PRM_COUNT);
    double *s_init = rd(f, m), *Ps_init = rd(f, m * m), *s_final = rd(f, m), *Ps_final = rd(f, m * m), *Q = rd(f, m * m);
    fclose(f);

    epi_batch_desc d;
    memset(&d, 0, sizeof d);
    d.abi_version = EPIEKF_ABI_VERSION; d.model = EPI_MODEL_SIA6; d.B = 1; d.T = T; d.Sx = 1; d.Su = 1; d.n_npi = n;
    d.L = hdr[2]; d.order = hdr[3]; d.obs_type = EPI_OBS_NEWCASES; d.r_mode = 1; d.out_mask = EPI_OUT_ALL;
    epi_inputs in;
    memset(&in, 0, sizeof in);
    in.x = x; in.u = u; in.R_series = R; in.prm = prm; in.s_init = s_init; in.Ps_init = Ps_init;
    in.s_final = s_final; in.Ps_final = Ps_final; in.Q = Q;
    epi_outputs out;
    memset(&out, 0, sizeof out);
    const size_t nS = (size_t)m * T, nP = (size_t)m * m * T, nU = (size_t)n * T;
    out.u_opt = calloc(nU, 8); out.u_opt_smooth = calloc(nU, 8);
    out.S_MINUS = calloc(nS, 8); out.S_PLUS = calloc(nS, 8); out.S_SMOOTH = calloc(nS, 8);
    out.P_MINUS = calloc(nP, 8); out.P_PLUS = calloc(nP, 8); out.P_SMOOTH = calloc(nP, 8);
    out.K_GAIN = calloc(nS, 8); out.innovations = calloc(T, 8); out.rho = calloc(T, 8);
    char err[256] = {0};
    /* placement (ABI 6): the first call of a size allocates the device arena -- let it be the fastest of three candidates;
       the report says what was tried, the next call finds the arena with the pooled context */
    epi_placement_report rep;
    d.placement_tries = 3; out.placement = &rep;
    int rc = epi_ekf_run_host(&d, &in, &out, 0, err);
    if (rc != EPI_OK) { fprintf(stderr, "epi_ekf_run_host: %d (%s) %s\n", rc, epi_status_string(rc), err); return 1; }
    printf("placement: %d arenas tried, kept #%d:", rep.tries, rep.chosen);
    for (int i = 0; i < rep.tries; i++) printf(" %.3f ms", rep.ms[i]);
    printf("\n");
    if (rep.tries != 3 || rep.chosen < 0 || rep.chosen >= 3) return 1;
    rc = epi_ekf_run_host(&d, &in, &out, 0, err);
    if (rc != EPI_OK || rep.tries != 0) { fprintf(stderr, "second call: rc %d, %d tries (expected none: pooled arena)\n", rc, rep.tries); return 1; }

    /* error behaviour mirrors the reference: order = 3 -> 'Undefined order' */
    d.order = 3;
    rc = epi_ekf_run_host(&d, &in, &out, 0, err);
    printf("order=3 -> rc %d, message '%s'\n", rc, err);
    if (rc != EPI_ERR_UNDEFINED_ORDER || strcmp(err, "Undefined order") != 0) return 1;

    f = fopen(argv[2], "wb");
    fwrite(out.S_PLUS, 8, nS, f); fwrite(out.S_SMOOTH, 8, nS, f); fwrite(out.P_SMOOTH, 8, nP, f);
    fwrite(out.u_opt_smooth, 8, nU, f); fwrite(out.rho, 8, T, f);
    fclose(f);
    printf("ok: T=%d, S_SMOOTH(1,T)=%.17g\n", T, out.S_SMOOTH[(size_t)m * (T - 1)]);
    return 0;
}
