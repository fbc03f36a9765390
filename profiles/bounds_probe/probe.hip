// Does the raw-buffer bounds check of gfx950 include the SGPR offset (soffset)?  load_u() relies on it: rows beyond
// n_npi are addressed through soffset and must read as 0.0.  Prints what a lane gets for in-range and out-of-range rows.
//   hipcc --offload-arch=gfx950 -O2 probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ void k(const double *src, unsigned bytes, unsigned rowb, double *out)
{
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, bytes, 0x00020000);
    const unsigned voff = threadIdx.x * 8u;
    for (int row = 0; row < 6; row++) {
        unsigned soff = __builtin_amdgcn_readfirstlane(row * rowb);
        out[row * 64 + threadIdx.x] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    }
}
int main()
{
    double *src, *out, h[6 * 64];
    hipMalloc(&src, 6 * 64 * 8); hipMalloc(&out, 6 * 64 * 8);
    for (int i = 0; i < 6 * 64; i++) h[i] = 1000.0 + i;
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    // descriptor covers 3 rows of 64 doubles; rows 3..5 exist in memory but are beyond num_records
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, 3u * 64u * 8u, 64u * 8u, out);
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int row = 0; row < 6; row++) {
        printf("row %d lane 0 -> %g, lane 63 -> %g\n", row, h[row * 64], h[row * 64 + 63]);
        for (int l = 0; l < 64; l++) ok &= (row < 3) ? (h[row * 64 + l] == 1000.0 + row * 64 + l) : (h[row * 64 + l] == 0.0);
    }
    printf(ok ? "SOFFSET IS BOUNDS-CHECKED: out-of-range rows read 0\n" : "SOFFSET IS NOT BOUNDS-CHECKED\n");
    return ok ? 0 : 1;
}
