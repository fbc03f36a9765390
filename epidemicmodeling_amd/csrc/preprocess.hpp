// Per-region preprocessing that turns the data set's columns into the filters' inputs on the device
// (SURVEY.md 8(f2)): Tools/TrainPredictPrescribeNPI.m:142-198 (cleaning, smoothing), :201-202 (I0), :240 (R_v).
// Included by epiekf.hip.
//
//   preprocess_regions : one lane per region.  cumulative counts -> daily counts (diff, negatives clamped, a missing
//       last day filled with the last valid one, other gaps 0) -> 7-day causal moving average `filter(ones(1,W),W,.)`,
//       its cumulative sum, the zero-phase `filtfilt(ones(1,W2),W2,.)`, W2 = round(W/2), the normalised observation
//       series x (NEWCASES / TOTALCASES), the per-day observation-noise variance R_v, the case-fatality series and I0.
//   npi_fill           : one lane per (NPI, region).  N/A levels take the previous day's level; leading N/A are 0.
//
// Every array is [T][S] (day-major, region-minor) -- exactly the x / R_series / u layout the filter kernels read, so
// the outputs feed epi_ekf_run_device without a transpose.  `filter` and `filtfilt` are restated from their published
// definitions (direct form II transposed; Gustafsson's initial conditions with 3 (nb - 1) samples of odd reflection),
// operation for operation like the oracle: results are bit-identical to it.
#pragma once

constexpr int kPreMaxTaps = 32;

struct PreArgs {
    int T, S, W, W2, nfact, first_num_days;
    double min_cases;
    const double *cases, *deaths, *population;
    double *new_refined, *new_smoothed, *zero_lag, *x_new, *x_total, *R_v, *fatality, *I0;
    double *ws;   // [T] refined | [T] smoothed | [T + 2 nfact] first filtfilt pass, each x S
};

// diff([c(1); c]), negatives -> 0, NaN last day <- last valid day, remaining NaN -> 0   (:166-178), into dst [T][S]
EPI_DEV void pre_refine(const double *__restrict__ cum, double *__restrict__ dst, int T, int S, int s)
{
    double prev = cum[s], last_val = 0.0;
    bool any_valid = false;
    for (int t = 0; t < T; t++) {
        const double cur = cum[(size_t)t * S + s];
        double d = cur - prev;
        prev = cur;
        if (d < 0.0) d = 0.0;                 // NaN < 0 is false
        const bool ok = !is_nan(d);
        if (ok) { last_val = d; any_valid = true; }
        if (!ok) d = (t == T - 1 && any_valid) ? last_val : 0.0;
        dst[(size_t)t * S + s] = d;
    }
}

// y(n) = filter(ones(1,W), W, x)(n): c x(n) + (c x(n-1) + (... + c x(n-W+1))), summed from the oldest sample inwards
EPI_DEV double pre_causal_ma(const double *__restrict__ x, int n, int W, double c, int S, int s)
{
    double acc = 0.0;
    for (int k = W - 1; k >= 1; k--) {
        const double p = (n - k >= 0) ? c * x[(size_t)(n - k) * S + s] : 0.0;
        acc = p + acc;
    }
    return c * x[(size_t)n * S + s] + acc;
}

// one DF-II-T step of an nb-tap FIR with equal taps c; state z[0 .. nb-2]
EPI_DEV double pre_df2t_step(double (&z)[kPreMaxTaps], int nb, double c, double x)
{
    const double y = (nb > 1) ? c * x + z[0] : c * x;
#pragma unroll
    for (int i = 0; i < kPreMaxTaps - 1; i++) {
        if (i < nb - 2) z[i] = c * x + z[i + 1];
        else if (i == nb - 2) z[i] = c * x;
    }
    return y;
}

__global__ __launch_bounds__(64) void preprocess_regions(const PreArgs a)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.S) return;
    const int T = a.T, S = a.S, W = a.W, W2 = a.W2, nfact = a.nfact;
    const double N = a.population[s];
    double *ref = a.ws, *sm = a.ws + (size_t)T * S, *f1 = a.ws + (size_t)2 * T * S;
    pre_refine(a.cases, ref, T, S, s);
    // causal smoothing, cumulative sum, normalisation, I0  (:173-180, :201-202)
    {
        const double c = 1.0 / (double)W;
        double run = 0.0, i0_sum = 0.0;
        int i0_cnt = 0;
        for (int t = 0; t < T; t++) {
            const double y = pre_causal_ma(ref, t, W, c, S, s);
            run = (t == 0) ? y : run + y;
            sm[(size_t)t * S + s] = y;
            if (a.new_refined) a.new_refined[(size_t)t * S + s] = ref[(size_t)t * S + s];
            if (a.new_smoothed) a.new_smoothed[(size_t)t * S + s] = y;
            if (a.x_new) a.x_new[(size_t)t * S + s] = y / N;
            if (a.x_total) a.x_total[(size_t)t * S + s] = run / N;
            if (y > 0.0 && i0_cnt < a.first_num_days) { i0_sum = (i0_cnt == 0) ? y : i0_sum + y; i0_cnt++; }
        }
        if (a.I0) {
            const double mean = i0_cnt ? i0_sum / (double)i0_cnt : __builtin_nan("");
            a.I0[s] = fmax(a.min_cases, mean);           // max() ignores NaN
        }
    }
    // zero-phase smoothing and the observation-noise variance  (:174, :240)
    if (a.zero_lag || a.R_v) {
        const double c = 1.0 / (double)W2;
        double zi[kPreMaxTaps], z[kPreMaxTaps];
#pragma unroll
        for (int i = 0; i < kPreMaxTaps; i++) { zi[i] = 0.0; z[i] = 0.0; }
        // zi(nb-1) = b(nb), zi(i) = b(i+1) + zi(i+1): steady state of the DF-II-T delays for a unit step
#pragma unroll
        for (int i = kPreMaxTaps - 2; i >= 0; i--) {
            if (i == W2 - 2) zi[i] = c;
            else if (i < W2 - 2) zi[i] = c + zi[i + 1];
        }
        const int n = T + 2 * nfact;
        const double x0 = ref[s], xl = ref[(size_t)(T - 1) * S + s];
        auto padded = [&](int i) -> double {          // odd reflection of nfact samples at both ends
            if (i < nfact) return 2.0 * x0 - ref[(size_t)(nfact - i) * S + s];
            if (i < nfact + T) return ref[(size_t)(i - nfact) * S + s];
            return 2.0 * xl - ref[(size_t)(T - 2 - (i - nfact - T)) * S + s];
        };
        const double e0 = padded(0);
#pragma unroll
        for (int i = 0; i < kPreMaxTaps; i++) z[i] = zi[i] * e0;
        for (int i = 0; i < n; i++) f1[(size_t)i * S + s] = pre_df2t_step(z, W2, c, padded(i));
        const double r0 = f1[(size_t)(n - 1) * S + s];
#pragma unroll
        for (int i = 0; i < kPreMaxTaps; i++) z[i] = zi[i] * r0;
        for (int i = 0; i < n; i++) {                  // second pass over the time-reversed first pass
            const double y = pre_df2t_step(z, W2, c, f1[(size_t)(n - 1 - i) * S + s]);
            const int t = n - 1 - i - nfact;           // position after the final reversal, padding removed
            if (t >= 0 && t < T) {
                if (a.zero_lag) a.zero_lag[(size_t)t * S + s] = y;
                if (a.R_v) {
                    const double d = (y - ref[(size_t)t * S + s]) / N;
                    a.R_v[(size_t)t * S + s] = 0.1 * (d * d);
                }
            }
        }
    }
    // case-fatality series  (:183-197); `ref` is free again
    if (a.fatality && a.deaths) {
        pre_refine(a.deaths, ref, T, S, s);
        const double c = 1.0 / (double)W;
        double drun = 0.0, crun = 0.0;
        for (int t = 0; t < T; t++) {
            const double y = pre_causal_ma(ref, t, W, c, S, s);
            const double v = sm[(size_t)t * S + s];
            drun = (t == 0) ? y : drun + y;
            crun = (t == 0) ? v : crun + v;
            const double fr = drun / crun;
            a.fatality[(size_t)t * S + s] = is_nan(fr) ? 0.0 : fr;
        }
    }
}

// ip, out: [T][n_npi][S]; one lane per (NPI, region) column  (:142-150)
__global__ __launch_bounds__(256) void npi_fill(int T, int cols, const double *__restrict__ ip, double *__restrict__ out)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= cols) return;
    double prev = 0.0;                  // a leading N/A ends up 0 either way
    for (int t = 0; t < T; t++) {
        double v = ip[(size_t)t * cols + q];
        if (is_nan(v)) v = prev;
        out[(size_t)t * cols + q] = v;
        prev = v;
    }
}
