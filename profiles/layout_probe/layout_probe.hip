// Measurement only: does the forward kernel's store pattern -- ~104 row streams written concurrently, 320 B per wave and
// row -- reach HBM write bandwidth, and would a blocked layout (one wave's rows of a step contiguous) do better?
//   mode 0: out[t][row][B]            (the ABI's layout: row stride B*8 bytes)
//   mode 1: out[t][B/blk][row][blk]   (a wave's ROWS rows of one step are one contiguous ROWS*blk*8-byte block)
//   mode 2: out[t][B/lb][row][lb]     with a layout block lb (8, 16, ...) smaller than the lanes a wave uses
//   mode 3: out[t][B/blk][row/2][blk][2]  row PAIRS interleaved per chain: 16 B per lane and store, half the stores
// One lane per chain, blk lanes per 64-thread workgroup, T sequential steps, `work` dependent FMAs per step to mimic
// the latency-bound compute between stores.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(64) void probe(double *out, int B, int T, int rows, int blk, int mode, int work, int lb)
{
    const int lane = threadIdx.x, c = blockIdx.x * blk + lane;
    if (lane >= blk || c >= B) return;
    double v = c * 1e-9;
    for (int t = 0; t < T; t++) {
        for (int w = 0; w < work; w++) v = fma(v, 1.0000001, 1e-12);
        if (mode == 0) {
            double *p = out + (size_t)t * rows * B + c;
            for (int r = 0; r < rows; r++) p[(size_t)r * B] = v + r;
        } else if (mode == 1) {
            double *p = out + ((size_t)t * gridDim.x + blockIdx.x) * rows * blk + lane;
            for (int r = 0; r < rows; r++) p[(size_t)r * blk] = v + r;
        } else if (mode == 4) {     // [t][B/blk][row][lb]: blk lanes per wave in rows PADDED to lb doubles (rows start on a cache line)
            double *p = out + ((size_t)t * gridDim.x + blockIdx.x) * rows * lb + lane;
            for (int r = 0; r < rows; r++) p[(size_t)r * lb] = v + r;
        } else if (mode == 3) {
            double2 *p = (double2 *)out + ((size_t)t * gridDim.x + blockIdx.x) * (rows / 2) * blk + lane;
            for (int r = 0; r < rows / 2; r++) p[(size_t)r * blk] = make_double2(v + r, v - r);
        } else {
            const size_t nlb = (size_t)(B + lb - 1) / lb;
            double *p = out + (((size_t)t * nlb + c / lb) * rows) * lb + c % lb;
            for (int r = 0; r < rows; r++) p[(size_t)r * lb] = v + r;
        }
    }
}
extern "C" int run_probe(double *out, int B, int T, int rows, int blk, int mode, int work, int lb, void *stream)
{
    hipLaunchKernelGGL(probe, dim3((B + blk - 1) / blk), dim3(64), 0, (hipStream_t)stream, out, B, T, rows, blk, mode, work, lb);
    return (int)hipGetLastError();
}

// Round 3: the smoother's side.  eks_bwd_sym reads ~76 doubles per chain and step (S+ 6, P+ 21, S- 6, P- 21, X 21, rank word)
// and writes 54 (S_SMOOTH 6, P_SMOOTH 36, u_opt_smooth 12) as one lone 370-VGPR wave per SIMD.  Would 16 B per lane on
// the reads (row PAIRS interleaved per chain, buffer_load_b128) move the 24.5 GB faster than 8 B per lane?
//   rmode / wmode 0: [t][B/blk][row][blk] 8 B per lane      1: [t][B/blk][row/2][blk][2] 16 B per lane
// `work` FMAs in four independent chains sit between a step's loads and its stores; 40 KB of LDS per workgroup keeps the
// occupancy at one wave per SIMD like the real kernel.
extern "C" __global__ __launch_bounds__(64) void probe_rw(const double *in, double *out, int B, int T, int rrows, int wrows, int blk,
                                                           int rmode, int wmode, int work)
{
    extern __shared__ double pad[];
    const int lane = threadIdx.x, c = blockIdx.x * blk + lane;
    if (lane >= blk || c >= B) return;
    if (work < 0) pad[lane] = 1.0;
    double acc0 = 0.0, acc1 = 1.0, acc2 = 2.0, acc3 = 3.0;
    for (int t = T - 1; t >= 0; t--) {
        const size_t slice = (size_t)t * gridDim.x + blockIdx.x;
        // every load of the step is issued before the first one is consumed (as the real kernel does): 76 rows in flight
        constexpr int RR = 76;
        double s = 0.0;
        if (rmode == 0) {
            const double *p = in + slice * RR * blk + lane;
            double v[RR];
#pragma unroll
            for (int r = 0; r < RR; r++) v[r] = p[(size_t)r * blk];
#pragma unroll
            for (int r = 0; r < RR; r++) s += v[r];
        } else {
            const double2 *p = (const double2 *)in + slice * (RR / 2) * blk + lane;
            double2 v[RR / 2];
#pragma unroll
            for (int r = 0; r < RR / 2; r++) v[r] = p[(size_t)r * blk];
#pragma unroll
            for (int r = 0; r < RR / 2; r++) s += v[r].x + v[r].y;
        }
        acc0 += s;
        for (int w = 0; w < work; w += 4) {
            acc0 = fma(acc0, 1.0000001, 1e-12); acc1 = fma(acc1, 1.0000001, 1e-12);
            acc2 = fma(acc2, 1.0000001, 1e-12); acc3 = fma(acc3, 1.0000001, 1e-12);
        }
        const double v = (acc0 + acc1) + (acc2 + acc3);
        if (wmode == 0) {
            double *p = out + slice * wrows * blk + lane;
            for (int r = 0; r < wrows; r++) p[(size_t)r * blk] = v + r;
        } else {
            double2 *p = (double2 *)out + slice * (wrows / 2) * blk + lane;
            for (int r = 0; r < wrows / 2; r++) p[(size_t)r * blk] = make_double2(v + r, v - r);
        }
    }
}
extern "C" int run_probe_rw(const double *in, double *out, int B, int T, int rrows, int wrows, int blk, int rmode, int wmode, int work,
                            void *stream)
{
    hipLaunchKernelGGL(probe_rw, dim3((B + blk - 1) / blk), dim3(64), 40 * 1024, (hipStream_t)stream, in, out, B, T, rrows, wrows, blk,
                       rmode, wmode, work);
    return (int)hipGetLastError();
}

// Round 4: ONE record per (step, block) against the eight separate arrays the forward kernel writes today (rows 6, 6, 36,
// 36, 6, 1, 1, 12 of a 40-chain block): the same 104 rows of 320 B per wave and step, plain or non-temporal stores.
// Run over fresh allocations to see whether the spread between placements (profiles/alloc_probe.py) comes with the number
// of concurrently written arrays.
struct Ptr8 { double *p[8]; };
extern "C" __global__ __launch_bounds__(64) void probe8(Ptr8 a, int B, int T, int blk, int nt)
{
    const int rows[8] = {6, 6, 36, 36, 6, 1, 1, 12};
    const int lane = threadIdx.x, c = blockIdx.x * blk + lane;
    if (lane >= blk || c >= B) return;
    double v = c * 1e-9;
    for (int t = 0; t < T; t++) {
        v = fma(v, 1.0000001, 1e-12);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            double *p = a.p[k] + ((size_t)t * gridDim.x + blockIdx.x) * rows[k] * blk + lane;
            if (nt) for (int r = 0; r < rows[k]; r++) __builtin_nontemporal_store(v + r, p + (size_t)r * blk);
            else for (int r = 0; r < rows[k]; r++) p[(size_t)r * blk] = v + r;
        }
    }
}
extern "C" int run_probe8(double **ptrs, int B, int T, int blk, int nt, void *stream)
{
    Ptr8 a;
    for (int k = 0; k < 8; k++) a.p[k] = ptrs[k];
    hipLaunchKernelGGL(probe8, dim3((B + blk - 1) / blk), dim3(64), 0, (hipStream_t)stream, a, B, T, blk, nt);
    return (int)hipGetLastError();
}

// Round 4, BASELINE config 5's forward store pattern (fp32 storage, 3 states, 64-chain blocks): `n` arrays, array k with rows[k]
// rows of 64 elements of es[k] bytes (4 or 8) per block and step -- today's eight fp32 outputs + three fp64 workspace arrays
// against variants that fuse arrays into records.  nt stores.
struct PtrN { void *p[12]; int rows[12]; int es[12]; int n; };
extern "C" __global__ __launch_bounds__(64) void probeN(PtrN a, int B, int T)
{
    const int lane = threadIdx.x, c = blockIdx.x * 64 + lane;
    if (c >= B) return;
    double v = c * 1e-9;
    for (int t = 0; t < T; t++) {
        v = fma(v, 1.0000001, 1e-12);
        for (int k = 0; k < a.n; k++) {
            const size_t base = ((size_t)t * gridDim.x + blockIdx.x) * a.rows[k] * 64 + lane;
            if (a.es[k] == 8) { double *p = (double *)a.p[k] + base; for (int r = 0; r < a.rows[k]; r++) __builtin_nontemporal_store(v + r, p + (size_t)r * 64); }
            else { float *p = (float *)a.p[k] + base; for (int r = 0; r < a.rows[k]; r++) __builtin_nontemporal_store((float)(v + r), p + (size_t)r * 64); }
        }
    }
}
extern "C" int run_probeN(void **ptrs, const int *rows, const int *es, int n, int B, int T, void *stream)
{
    PtrN a;
    a.n = n;
    for (int k = 0; k < n; k++) { a.p[k] = ptrs[k]; a.rows[k] = rows[k]; a.es[k] = es[k]; }
    hipLaunchKernelGGL(probeN, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, a, B, T);
    return (int)hipGetLastError();
}

// probeN with the real kernel's constraints: `lds_bytes` of dynamic LDS per workgroup caps the resident waves (20 KB -> 8 per
// CU = 2 per SIMD), `work` dependent FMAs per step stand for the filter's arithmetic, and `wait` makes every step read one
// value per lane (from `in`, [T][B]) and use it -- a vector load the wave must wait for once per step, as the filter does.
extern "C" __global__ __launch_bounds__(64) void probeNW(PtrN a, const double *in, int B, int T, int work, int wait)
{
    extern __shared__ double pad[];
    const int lane = threadIdx.x, c = blockIdx.x * 64 + lane;
    if (c >= B) return;
    if (work < 0) pad[lane] = 1.0;
    double v = c * 1e-9;
    double nxt = wait ? in[c] : 0.0;
    for (int t = 0; t < T; t++) {
        const double cur = nxt;
        if (wait && t + 1 < T) nxt = in[(size_t)(t + 1) * B + c];
        v += cur;
        for (int w = 0; w < work; w++) v = fma(v, 1.0000001, 1e-12);
        for (int k = 0; k < a.n; k++) {
            const size_t base = ((size_t)t * gridDim.x + blockIdx.x) * a.rows[k] * 64 + lane;
            if (a.es[k] == 8) { double *p = (double *)a.p[k] + base; for (int r = 0; r < a.rows[k]; r++) __builtin_nontemporal_store(v + r, p + (size_t)r * 64); }
            else { float *p = (float *)a.p[k] + base; for (int r = 0; r < a.rows[k]; r++) __builtin_nontemporal_store((float)(v + r), p + (size_t)r * 64); }
        }
    }
}
extern "C" int run_probeNW(void **ptrs, const int *rows, const int *es, int n, const double *in, int B, int T, int work, int wait,
                           int lds_bytes, void *stream)
{
    PtrN a;
    a.n = n;
    for (int k = 0; k < n; k++) { a.p[k] = ptrs[k]; a.rows[k] = rows[k]; a.es[k] = es[k]; }
    hipLaunchKernelGGL(probeNW, dim3((B + 63) / 64), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, a, in, B, T, work, wait);
    return (int)hipGetLastError();
}

// The headline's forward pattern (104 fp64 rows per 40-chain block and step in eight arrays) with arithmetic between the
// stores and one awaited (cached) load per step, one wave per SIMD (40 KB of LDS per workgroup): 8 B per lane and store (104
// operations per step, as shipped) against 16 B per lane (row PAIRS interleaved per chain: 52 operations).  A wave may have 63
// vector-memory operations outstanding; does halving the operations let the arithmetic of the next step overlap the drain?
extern "C" __global__ __launch_bounds__(64) void probeP(Ptr8 a, const double *in, int B, int T, int blk, int work, int pair)
{
    extern __shared__ double pad[];
    const int rows[8] = {6, 6, 36, 36, 6, 2, 2, 12};      // (the two one-row arrays counted as one two-row array each way)
    const int lane = threadIdx.x, c = blockIdx.x * blk + lane;
    if (lane >= blk || c >= B) return;
    if (work < 0) pad[lane] = 1.0;
    double v0 = c * 1e-9, v1 = 1.0, v2 = 2.0, v3 = 3.0;
    double nxt = in[blockIdx.x & 255];
    for (int t = 0; t < T; t++) {
        const double cur = nxt;
        nxt = in[(t + blockIdx.x) & 255];
        v0 += cur;
        for (int w = 0; w < work; w += 4) {
            v0 = fma(v0, 1.0000001, 1e-12); v1 = fma(v1, 1.0000001, 1e-12); v2 = fma(v2, 1.0000001, 1e-12); v3 = fma(v3, 1.0000001, 1e-12);
        }
        const double v = (v0 + v1) + (v2 + v3);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (pair) {
                typedef double d2v __attribute__((ext_vector_type(2)));
                d2v *p = (d2v *)a.p[k] + ((size_t)t * gridDim.x + blockIdx.x) * (rows[k] / 2) * blk + lane;
                for (int r = 0; r < rows[k] / 2; r++) { d2v x = {v + r, v - r}; __builtin_nontemporal_store(x, p + (size_t)r * blk); }
            } else {
                double *p = a.p[k] + ((size_t)t * gridDim.x + blockIdx.x) * rows[k] * blk + lane;
                for (int r = 0; r < rows[k]; r++) __builtin_nontemporal_store(v + r, p + (size_t)r * blk);
            }
        }
    }
}
extern "C" int run_probeP(double **ptrs, const double *in, int B, int T, int blk, int work, int pair, int lds_bytes, void *stream)
{
    Ptr8 a;
    for (int k = 0; k < 8; k++) a.p[k] = ptrs[k];
    hipLaunchKernelGGL(probeP, dim3((B + blk - 1) / blk), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, a, in, B, T, blk, work, pair);
    return (int)hipGetLastError();
}

// pair = 2 of the idea above WITHOUT another layout: the wave writes its rows to LDS ([row][lane]) and reads them back
// transposed -- lanes 0..blk/2-1 take row 2q, lanes blk/2..blk-1 row 2q+1, each TWO neighbouring chains (16 B) -- so that one
// store covers two rows of the shipped [t][block][row][blk] layout: 53 stores of 16 B per lane instead of 106 of 8 B, the
// same bytes at the same addresses.
extern "C" __global__ __launch_bounds__(64) void probeT(Ptr8 a, const double *in, int B, int T, int blk, int work)
{
    extern __shared__ double lds[];          // [36][blk] transposition tile (+ padding to cap the occupancy)
    typedef double d2v __attribute__((ext_vector_type(2)));
    const int rows[8] = {6, 6, 36, 36, 6, 2, 2, 12};
    const int lane = threadIdx.x, c = blockIdx.x * blk + lane;
    if (lane >= blk || c >= B) return;       // (a ragged last block would need the per-row path; not in this probe)
    const int half = blk / 2, hi = lane >= half ? 1 : 0, pl = lane - hi * half;      // my row of the pair, my chain pair
    double v0 = c * 1e-9, v1 = 1.0, v2 = 2.0, v3 = 3.0;
    double nxt = in[blockIdx.x & 255];
    for (int t = 0; t < T; t++) {
        const double cur = nxt;
        nxt = in[(t + blockIdx.x) & 255];
        v0 += cur;
        for (int w = 0; w < work; w += 4) {
            v0 = fma(v0, 1.0000001, 1e-12); v1 = fma(v1, 1.0000001, 1e-12); v2 = fma(v2, 1.0000001, 1e-12); v3 = fma(v3, 1.0000001, 1e-12);
        }
        const double v = (v0 + v1) + (v2 + v3);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            for (int r = 0; r < rows[k]; r++) lds[r * blk + lane] = v + r;
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the wave's own LDS writes have landed (one wave per workgroup)
            char *base = (char *)(a.p[k] + ((size_t)t * gridDim.x + blockIdx.x) * rows[k] * blk);
            for (int q = 0; q < rows[k] / 2; q++) {
                const int r = 2 * q + hi;
                const d2v x = *(const d2v *)&lds[r * blk + 2 * pl];
                __builtin_nontemporal_store(x, (d2v *)(base + ((size_t)r * blk + 2 * pl) * 8));
            }
        }
    }
}
extern "C" int run_probeT(double **ptrs, const double *in, int B, int T, int blk, int work, int lds_bytes, void *stream)
{
    Ptr8 a;
    for (int k = 0; k < 8; k++) a.p[k] = ptrs[k];
    hipLaunchKernelGGL(probeT, dim3((B + blk - 1) / blk), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, a, in, B, T, blk, work);
    return (int)hipGetLastError();
}
