// Scenario generation and selection around the Pareto sweep; included by epiekf.hip.
//
//   random_npi_mc : Tools/TrainPredictPrescribeNPI.m:496-521 -- per region, n_scen random NPI plans on the forecast
//                   horizon (the first half constant over time, the rest random over NPI and time), each simulated
//                   with SIalpha_Controlled from the end-of-history state and scored with NPICost over
//                   [historic days, horizon days].  One lane per (scenario, region); the plan is never stored
//                   unless the caller asks for it.
//   pareto_front  : :624-633 -- non-dominated filter over the sweep's (J0, J1) points of each region and the
//                   normalised-distance optimum I_opt.  One 256-thread workgroup per region, points staged in LDS.
#pragma once

// Philox4x32-10 (Salmon et al., SC'11): counter-based, so the draw of (region, scenario, NPI, day) does not depend
// on launch geometry and the CPU oracle reproduces it exactly.  MATLAB's randi stream itself cannot be matched
// (and is not part of the function's contract); the mapping plan -> (J0, J1) is what parity is checked on.
EPI_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// chain c = scenario * R + region (region-minor: lanes of a wave read consecutive columns of the per-region blocks)
__global__ __launch_bounds__(256) void random_npi_mc(const epi_mc_desc d, const double *__restrict__ sp,
                                                     const double *__restrict__ u_min, const double *__restrict__ z,
                                                     const double *__restrict__ J0_prefix,
                                                     const double *__restrict__ J1_prefix, double *__restrict__ u_out,
                                                     double *__restrict__ J0, double *__restrict__ J1)
{
    const int B = d.R * d.n_scen;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    const int r = c % d.R, j = c / d.R;
    SimPrm p;
    double s, i, al;
    load_sim_prm(p, sp, d.R, r, s, i, al);
    double lo[kNpi];
    uint32_t span[kNpi];   // randi([lo, hi]): hi - lo + 1 equally likely integers
#pragma unroll
    for (int k = 0; k < kNpi; k++) {
        lo[k] = (k < d.n_npi) ? u_min[(size_t)k * d.R + r] : 0.0;
        const double w = (k < d.n_npi) ? (p.um[k] - lo[k]) : 0.0;
        span[k] = (w >= 0.0 && w < 4294967295.0) ? (uint32_t)w + 1u : 1u;
    }
    // :502  `scenario < num_random_input_monte_carlo_runs/2` with a 1-based scenario index
    const bool constant_plan = 2 * (j + 1) < d.n_scen;
    const bool pre = d.prefix_days > 0;
    double acc0 = pre ? J0_prefix[r] : 0.0, acc1 = pre ? J1_prefix[r] : 0.0;
    double uk[kNpi];
#pragma unroll
    for (int k = 0; k < kNpi; k++) uk[k] = 0.0;
    for (int t = 0; t < d.K; t++) {
        if (!constant_plan || t == 0) {
            // counter = (region, scenario, NPI block of four, day); day 0 is the draw of a constant plan
            const uint32_t day = constant_plan ? 0u : (uint32_t)t + 1u;
#pragma unroll
            for (int blk = 0; blk < kNpi / 4; blk++) {
                uint32_t x[4];
                philox4x32_10((uint32_t)r, (uint32_t)j, (uint32_t)blk, day, d.seed_lo, d.seed_hi, x);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int k = blk * 4 + q;
                    if (k < d.n_npi) uk[k] = lo[k] + (double)__umulhi(x[q], span[k]);
                }
            }
        }
        if (u_out) {
#pragma unroll
            for (int k = 0; k < kNpi; k++)
                if (k < d.n_npi) u_out[((size_t)t * d.n_npi + k) * B + c] = uk[k];
        }
        double z1 = 0.0, z2 = 0.0, z3 = 0.0;
        if (d.noise) {
            z1 = z[((size_t)t * 3 + 0) * B + c]; z2 = z[((size_t)t * 3 + 1) * B + c]; z3 = z[((size_t)t * 3 + 2) * B + c];
        }
        sialpha_step(p, uk, z1, z2, z3, s, i, al);
        npicost_accumulate(p, uk, d.n_npi, t == 0 && !pre, s, i, al, acc0, acc1);
    }
    const size_t days = (size_t)d.K + (size_t)d.prefix_days;
    J0[c] = acc0 / (double)days;
    J1[c] = acc1 / (double)((size_t)d.n_npi * days);
}

// (value, index) pairs ordered like MATLAB's [~, I] = min(v): NaNs ignored, first index among equal minima
EPI_DEV bool argmin_better(double v, int i, double bv, int bi)
{
    if (is_nan(v)) return false;
    if (is_nan(bv)) return true;
    return v < bv || (v == bv && i < bi);
}

// J0, J1: [R][P] (the sweep's chain order: region-major, cost weight fastest).  on_front [R][P]; i_opt [R] (0-based).
__global__ __launch_bounds__(256) void pareto_front(int P, const double *__restrict__ J0, const double *__restrict__ J1,
                                                    int32_t *__restrict__ on_front, int32_t *__restrict__ i_opt,
                                                    const int32_t *__restrict__ gate)
{
    if (gate && *gate == 0) return;   // second pass after the dense re-run of non-finite chains: nothing was re-run
    extern __shared__ double pts[];   // [2][P]
    __shared__ double red_a[256], red_b[256];
    __shared__ int red_i[256];
    double *a = pts, *b = pts + P;
    const int r = blockIdx.x, tid = threadIdx.x;
    const double nan = __builtin_nan("");
    double ma = nan, mb = nan;        // max() ignores NaN; all-NaN stays NaN
    for (int q = tid; q < P; q += 256) {
        const double va = J0[(size_t)r * P + q], vb = J1[(size_t)r * P + q];
        a[q] = va; b[q] = vb;
        ma = fmax(ma, va); mb = fmax(mb, vb);
    }
    red_a[tid] = ma; red_b[tid] = mb;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) { red_a[tid] = fmax(red_a[tid], red_a[tid + w]); red_b[tid] = fmax(red_b[tid], red_b[tid + w]); }
        __syncthreads();
    }
    ma = red_a[0]; mb = red_b[0];
    __syncthreads();
    double bv = nan;
    int bi = 0x7fffffff;
    for (int q = tid; q < P; q += 256) {
        const double va = a[q], vb = b[q];
        // :626  sum(J0 < J0(ii) & J1 < J1(ii)) == 0
        int dominated = 0;
        for (int o = 0; o < P; o++) dominated |= (a[o] < va) & (b[o] < vb);
        if (on_front) on_front[(size_t)r * P + q] = dominated ? 0 : 1;
        // :633  (J0/max(J0)).^2 + (J1/max(J1)).^2
        const double na = va / ma, nb = vb / mb;
        const double sc = na * na + nb * nb;
        if (argmin_better(sc, q, bv, bi)) { bv = sc; bi = q; }
    }
    red_a[tid] = bv; red_i[tid] = bi;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w && argmin_better(red_a[tid + w], red_i[tid + w], red_a[tid], red_i[tid])) {
            red_a[tid] = red_a[tid + w]; red_i[tid] = red_i[tid + w];
        }
        __syncthreads();
    }
    if (tid == 0 && i_opt) i_opt[r] = is_nan(red_a[0]) ? 0 : red_i[0];   // min of all-NaN returns index 1
}

// Tools/NPICost.m:1-10, one lane per chain: both means as sequential sums in column-major element order
__global__ __launch_bounds__(256) void npi_cost(int B, int T, int n_npi, int Su, int per_day, const int32_t *__restrict__ u_series,
                                                const double *__restrict__ newcases, const double *__restrict__ inputs,
                                                const double *__restrict__ weights, double *__restrict__ J0,
                                                double *__restrict__ J1)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    const int su = u_series ? u_series[c] : c;
    double a0 = 0.0, a1 = 0.0;
    for (int t = 0; t < T; t++) {
        const double nc = newcases[(size_t)t * B + c];
        a0 = (t == 0) ? nc : a0 + nc;                                                  // :6
        for (int k = 0; k < n_npi; k++) {
            const double w = weights[((size_t)(per_day ? t : 0) * n_npi + k) * B + c];
            const double term = w * inputs[((size_t)t * n_npi + k) * Su + su];         // :9
            a1 = (t == 0 && k == 0) ? term : a1 + term;                                // :10
        }
    }
    J0[c] = a0 / (double)T;
    J1[c] = a1 / (double)((size_t)n_npi * (size_t)T);
}

// Tools/SI_Controlled.m:12-23, one lane per chain: 2-state forward Euler with a time-dependent infection rate; the
// first sample is the initial condition (:15-16)
__global__ __launch_bounds__(256) void si_controlled(int B, int K, int Sa, double dt, const int32_t *__restrict__ a_series,
                                                     const double *__restrict__ alpha, const double *__restrict__ prm,
                                                     double *__restrict__ s_out, double *__restrict__ i_out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    const int sa = a_series ? a_series[c] : c;
    const double beta = prm[c];
    double s = prm[(size_t)B + c], i = prm[(size_t)2 * B + c];
    s_out[c] = s; i_out[c] = i;
    for (int t = 0; t < K - 1; t++) {
        const double al = alpha[(size_t)t * Sa + sa];
        const double sn = fmax(0.0, fmin(1.0, s - dt * al * s * i));                       // :21
        const double in = fmax(0.0, fmin(1.0, i + dt * (al * s * i - beta * i)));          // :22
        s = sn; i = in;
        s_out[(size_t)(t + 1) * B + c] = s; i_out[(size_t)(t + 1) * B + c] = i;
    }
}

// testScripts/testSIR01.m:28-36 (BASELINE config 1), one lane per parameter set: the 3-compartment SIR with return flow
// r -> s, forward Euler, no clamps; prm [6][B] = alpha, beta, gamma, s0, i0, r0; out [K][3][B], first sample = initial state
__global__ __launch_bounds__(256) void sir_sim(int B, int K, double dt, const double *__restrict__ prm, double *__restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    const double alpha = prm[c], beta = prm[(size_t)B + c], gamma = prm[(size_t)2 * B + c];
    double s = prm[(size_t)3 * B + c], i = prm[(size_t)4 * B + c], r = prm[(size_t)5 * B + c];
    out[c] = s; out[(size_t)B + c] = i; out[(size_t)2 * B + c] = r;
    for (int t = 0; t < K - 1; t++) {
        const double sn = (-alpha * s * i + gamma * r) * dt + s;      // :33
        const double in = (alpha * s * i - beta * i) * dt + i;        // :34
        const double rn = (beta * i - gamma * r) * dt + r;            // :35
        s = sn; i = in; r = rn;
        double *o = out + (size_t)(t + 1) * 3 * B + c;
        o[0] = s; o[(size_t)B] = i; o[(size_t)2 * B] = r;
    }
}

// ---------------------------------------------------------------------------
// the cost-weight sweep from per-region inputs (epi_sweep_prescribe_host)
// ---------------------------------------------------------------------------
// Tools/TrainPredictPrescribeNPI.m:421-460 runs the SAME region inputs with params.epsilon = human_npi_cost_factor(ll):
// chain c = region * P + ll.  Per-region rows [nrows][R] become per-chain rows [nrows][R * P] on the device (a host caller
// sends R columns, not R * P); row `eps_row` is the cost-weight grid instead, and `series` [R * P] receives the region
// of every chain (its x / u / R_v series).
__global__ __launch_bounds__(256) void sweep_expand(int nrows, int R, int P, int eps_row, const double *__restrict__ src,
                                                    const double *__restrict__ eps, double *__restrict__ dst,
                                                    int32_t *__restrict__ series)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int B = R * P;
    if (c >= B) return;
    const int r = c / P;
    if (blockIdx.y == 0 && series) series[c] = r;
    for (int row = blockIdx.y; row < nrows; row += gridDim.y)
        dst[(size_t)row * B + c] = (row == eps_row) ? eps[c - r * P] : src[(size_t)row * R + r];
}

// The prescription (:624-633 picks I_opt; the plan of that cost weight is what the caller keeps): column
// c = r * P + i_opt[r] of a filter output [T][rows][chains] (classic or chain-blocked, see Lay) -> dst [T][rows][R]
__global__ __launch_bounds__(256) void sweep_gather_opt(int T, int rows, int R, int P, int blk, int nblk,
                                                        const int32_t *__restrict__ i_opt, const double *__restrict__ src,
                                                        double *__restrict__ dst)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)T * rows * R) return;
    const int r = (int)(idx % (size_t)R);
    const int row = (int)((idx / (size_t)R) % (size_t)rows);
    const size_t t = idx / ((size_t)R * rows);
    const int c = r * P + i_opt[r];
    const int cb = c / blk, cr = c - cb * blk;
    dst[idx] = src[((t * (size_t)nblk + cb) * rows + row) * blk + cr];
}
