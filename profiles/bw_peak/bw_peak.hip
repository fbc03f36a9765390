// Measurement only: what this box's HBM delivers to the simplest possible streams -- write-only, read-only and copy, 8 or
// 16 bytes per lane, default or non-temporal cache policy -- as a ceiling for the filter kernels' own streams
// (profiles/bw_peak/run.py).  Grid-stride over n elements, 256-thread workgroups, 8 workgroups per CU.
#include <hip/hip_runtime.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int W, int NT>
__global__ __launch_bounds__(256) void k_fill(char *dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (W == 8) { u32x2 v = {(unsigned)i, 1u}; if (NT) __builtin_nontemporal_store(v, (u32x2 *)dst + i); else ((u32x2 *)dst)[i] = v; }
        else { u32x4 v = {(unsigned)i, 1u, 2u, 3u}; if (NT) __builtin_nontemporal_store(v, (u32x4 *)dst + i); else ((u32x4 *)dst)[i] = v; }
    }
}
template <int W, int NT>
__global__ __launch_bounds__(256) void k_read(const char *src, size_t n, unsigned *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (W == 8) { u32x2 v = NT ? __builtin_nontemporal_load((const u32x2 *)src + i) : ((const u32x2 *)src)[i]; acc += v.x ^ v.y; }
        else { u32x4 v = NT ? __builtin_nontemporal_load((const u32x4 *)src + i) : ((const u32x4 *)src)[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) *sink = acc;
}
template <int W, int NT>
__global__ __launch_bounds__(256) void k_copy(const char *src, char *dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (W == 8) { u32x2 v = NT ? __builtin_nontemporal_load((const u32x2 *)src + i) : ((const u32x2 *)src)[i]; if (NT) __builtin_nontemporal_store(v, (u32x2 *)dst + i); else ((u32x2 *)dst)[i] = v; }
        else { u32x4 v = NT ? __builtin_nontemporal_load((const u32x4 *)src + i) : ((const u32x4 *)src)[i]; if (NT) __builtin_nontemporal_store(v, (u32x4 *)dst + i); else ((u32x4 *)dst)[i] = v; }
    }
}
// The same streams as ONE-SHOT grids: every workgroup moves one contiguous tile of U * 256 elements and ends (what a
// vectorised elementwise library kernel does), instead of 2 048 persistent workgroups striding through the buffer.
template <int W, int NT, int U>
__global__ __launch_bounds__(256) void k_fill_tile(char *dst, size_t n)
{
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t i = ((size_t)blockIdx.x * U + u) * 256 + threadIdx.x;
        if (i >= n) return;
        if (W == 8) { u32x2 v = {(unsigned)i, 1u}; if (NT) __builtin_nontemporal_store(v, (u32x2 *)dst + i); else ((u32x2 *)dst)[i] = v; }
        else { u32x4 v = {(unsigned)i, 1u, 2u, 3u}; if (NT) __builtin_nontemporal_store(v, (u32x4 *)dst + i); else ((u32x4 *)dst)[i] = v; }
    }
}
template <int W, int NT, int U>
__global__ __launch_bounds__(256) void k_read_tile(const char *src, size_t n, unsigned *sink)
{
    unsigned acc = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t i = ((size_t)blockIdx.x * U + u) * 256 + threadIdx.x;
        if (i >= n) break;
        if (W == 8) { u32x2 v = NT ? __builtin_nontemporal_load((const u32x2 *)src + i) : ((const u32x2 *)src)[i]; acc += v.x ^ v.y; }
        else { u32x4 v = NT ? __builtin_nontemporal_load((const u32x4 *)src + i) : ((const u32x4 *)src)[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) *sink = acc;
}
template <int W, int NT, int U>
__global__ __launch_bounds__(256) void k_copy_tile(const char *src, char *dst, size_t n)
{
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t i = ((size_t)blockIdx.x * U + u) * 256 + threadIdx.x;
        if (i >= n) return;
        if (W == 8) { u32x2 v = NT ? __builtin_nontemporal_load((const u32x2 *)src + i) : ((const u32x2 *)src)[i]; if (NT) __builtin_nontemporal_store(v, (u32x2 *)dst + i); else ((u32x2 *)dst)[i] = v; }
        else { u32x4 v = NT ? __builtin_nontemporal_load((const u32x4 *)src + i) : ((const u32x4 *)src)[i]; if (NT) __builtin_nontemporal_store(v, (u32x4 *)dst + i); else ((u32x4 *)dst)[i] = v; }
    }
}
extern "C" int bw_run_tile(int kind, int width, int nt, const void *src, void *dst, size_t bytes, void *sink, void *stream)
{
    constexpr int U = 4;
    const size_t n = bytes / (size_t)width;
    const dim3 g((unsigned)((n + 256 * U - 1) / (256 * U))), b(256);
    hipStream_t st = (hipStream_t)stream;
#define GOT(K, ...) \
    if (width == 8 && !nt) hipLaunchKernelGGL((K<8, 0, U>), g, b, 0, st, __VA_ARGS__); \
    else if (width == 8) hipLaunchKernelGGL((K<8, 1, U>), g, b, 0, st, __VA_ARGS__); \
    else if (!nt) hipLaunchKernelGGL((K<16, 0, U>), g, b, 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((K<16, 1, U>), g, b, 0, st, __VA_ARGS__);
    if (kind == 0) { GOT(k_fill_tile, (char *)dst, n) }
    else if (kind == 1) { GOT(k_read_tile, (const char *)src, n, (unsigned *)sink) }
    else { GOT(k_copy_tile, (const char *)src, (char *)dst, n) }
    return (int)hipGetLastError();
}
extern "C" int bw_run(int kind, int width, int nt, const void *src, void *dst, size_t bytes, void *sink, void *stream)
{
    const size_t n = bytes / (size_t)width;
    const dim3 g(256 * 8), b(256);
    hipStream_t st = (hipStream_t)stream;
#define GO(K, ...) \
    if (width == 8 && !nt) hipLaunchKernelGGL((K<8, 0>), g, b, 0, st, __VA_ARGS__); \
    else if (width == 8) hipLaunchKernelGGL((K<8, 1>), g, b, 0, st, __VA_ARGS__); \
    else if (!nt) hipLaunchKernelGGL((K<16, 0>), g, b, 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((K<16, 1>), g, b, 0, st, __VA_ARGS__);
    if (kind == 0) { GO(k_fill, (char *)dst, n) }
    else if (kind == 1) { GO(k_read, (const char *)src, n, (unsigned *)sink) }
    else { GO(k_copy, (const char *)src, (char *)dst, n) }
    return (int)hipGetLastError();
}
