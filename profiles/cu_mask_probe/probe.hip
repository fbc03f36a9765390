// Where do the bits of a queue's CU mask land on an MI355X?  (measurement only, not part of the product)
//
//   hipcc --offload-arch=gfx950 -O2 probe.hip -o probe && ./probe
//
// Each workgroup of a spinning kernel records the XCD, shader engine and compute unit it runs on (s_getreg XCC_ID, HW_ID);
// the host prints the set of places a masked queue's workgroups were seen on.  Every mask used here leaves each XCD at least
// one compute unit under BOTH readings of the bit order (bits dealt round-robin over the XCDs, or XCD by XCD), so no
// workgroup can be left without a place to run.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where(uint32_t *out, int spin)
{
    // HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20; operand = (size-1) << 11 | offset << 6 | id
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(const char *label, const std::vector<uint32_t> &mask, int cus)
{
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    const int wgs = 8192;
    uint32_t *d;
    CK(hipMalloc(&d, wgs * 2 * sizeof(uint32_t)));
    hipLaunchKernelGGL(where, dim3(wgs), dim3(64), 0, st, d, 2000);      // 2000 ticks of the 100 MHz counter = 20 us
    CK(hipGetLastError());
    CK(hipStreamSynchronize(st));
    std::vector<uint32_t> h(wgs * 2);
    CK(hipMemcpy(h.data(), d, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::map<int, std::map<int, std::set<int>>> seen;     // xcd -> se -> cu
    for (int i = 0; i < wgs; i++) {
        const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        seen[xcc][(hw >> 13) & 7].insert((hw >> 8) & 0x1f);          // se_id [15:13], sh_id [12] + cu_id [11:8]
    }
    int bits = 0;
    for (uint32_t w : mask) bits += __builtin_popcount(w);
    int total = 0;
    printf("%s  (%d of %d bits set)\n", label, bits, cus);
    for (auto &x : seen) {
        printf("   XCD %d:", x.first);
        for (auto &s : x.second) { printf("  SE%d %zu CUs", s.first, s.second.size()); total += (int)s.second.size(); }
        printf("\n");
    }
    printf("   -> %d compute units on %zu XCDs\n", total, seen.size());
    CK(hipFree(d));
    CK(hipStreamDestroy(st));
}

int main()
{
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("%d compute units\n", cus);
    const int words = (cus + 31) / 32;
    std::vector<uint32_t> all(words, 0u);
    for (int i = 0; i < cus; i++) all[i / 32] |= 1u << (i % 32);
    run("all bits", all, cus);
    // bits 0..31 and every 32nd bit: XCD by XCD that is all of XCD 0 and one CU on each other die; round-robin it is four
    // CUs on every die (and a few more on die 0)
    std::vector<uint32_t> m1(words, 0u);
    m1[0] = 0xffffffffu;
    for (int k = 1; k < words; k++) m1[k] |= 1u;
    run("bits 0-31 + every 32nd", m1, cus);
    // 5/8 of every die under both readings, but not the same share of every SE (the forward waves of the time-pipelined
    // launch doubled up on the short engines with this one: 4.7 instead of 3.2 ms per pass at 9 375 chains)
    std::vector<uint32_t> m2(words, 0u), m2c(words, 0u);
    for (int i = 0; i < cus; i++) ((((i % 8) + (i / 8)) % 8 < 5) ? m2 : m2c)[i / 32] |= 1u << (i % 32);
    run("((i%8)+(i/8))%8 < 5", m2, cus);
    run("its complement", m2c, cus);
    // the first 160 bits + one bit in every later word, and the complement: only meaningful (and only balanced) if the
    // bits are dealt round-robin over XCDs, then SEs
    std::vector<uint32_t> m3(words, 0u), m3c(words, 0u);
    for (int i = 0; i < cus; i++) {
        const bool on = i < 160 ? (i % 32 != 31) : (i % 32 == 31);
        (on ? m3 : m3c)[i / 32] |= 1u << (i % 32);
    }
    run("first 160 bits (one bit per word swapped)", m3, cus);
    run("its complement", m3c, cus);
    return 0;
}
