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
