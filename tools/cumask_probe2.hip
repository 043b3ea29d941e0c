// Do complementary CU masks partition the chip?  Mask bit k = CU (k / 8) of XCD (k % 8) (tools/cumask_probe.hip: a mask that leaves
// an XCD without a CU falls back to a default set there, so every mask here keeps CUs in every XCD).  For a split s (CUs per XCD
// for set T, the other 32 - s for set S): two streams, a 4096-workgroup kernel on each, the physical CUs each one reached.
// Build: hipcc -O2 --offload-arch=gfx950 tools/cumask_probe2.hip -o tools/bin/cumask_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void where(unsigned *out) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
    for (int i = 0; i < 200; i++) __builtin_amdgcn_s_sleep(10);
}
static std::set<unsigned> run(const std::vector<uint32_t> &mask, unsigned *d, int G) {
    hipStream_t s;
    std::set<unsigned> seen;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("stream creation failed\n"); return seen; }
    hipLaunchKernelGGL(where, dim3(G), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * G);
    hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < G; b++) seen.insert(((h[2 * b + 1] & 0xf) << 12) | (((h[2 * b] >> 13) & 7) << 8) | (((h[2 * b] >> 12) & 1) << 4) | ((h[2 * b] >> 8) & 0xf));
    hipStreamDestroy(s);
    return seen;
}
int main() {
    const int G = 8192;
    unsigned *d;
    hipMalloc(&d, G * 8);
    for (int s : {4, 5, 6, 8}) {
        std::vector<uint32_t> T(8, 0), S(8, 0);
        for (int k = 0; k < 256; k++) ((k / 8 >= 32 - s) ? T : S)[k / 32] |= 1u << (k % 32);
        std::set<unsigned> t = run(T, d, G), sc = run(S, d, G);
        int common = 0, per_xcd_t[8] = {0}, per_xcd_s[8] = {0};
        for (unsigned v : t) { common += sc.count(v); per_xcd_t[v >> 12]++; }
        for (unsigned v : sc) per_xcd_s[v >> 12]++;
        printf("split %d/%d per XCD: set T reached %zu CUs, set S %zu, in common %d; per XCD T:", s, 32 - s, t.size(), sc.size(), common);
        for (int x = 0; x < 8; x++) printf(" %d", per_xcd_t[x]);
        printf("  S:");
        for (int x = 0; x < 8; x++) printf(" %d", per_xcd_s[x]);
        printf("\n");
    }
    return 0;
}
