// Which physical CU does bit k of a hipExtStreamCreateWithCUMask mask select on this GPU?  For every bit k: a stream with only
// that bit set, a kernel of 64 one-wave workgroups, each reporting (XCC_ID, HW_ID).  Prints bit -> (xcd, se, cu) and whether
// the mask was honoured (all workgroups on one CU).  Build: hipcc -O2 --offload-arch=gfx950 tools/cumask_probe.hip -o tools/bin/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void where(unsigned *out) {
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    for (int i = 0; i < 2000; i++) __builtin_amdgcn_s_sleep(10);  // stay resident so that later workgroups must find their own CU
}
int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount, words = (n_cu + 31) / 32;
    printf("%s: %d CUs\n", prop.name, n_cu);
    unsigned *d;
    const int G = 64;
    hipMalloc(&d, G * 8);
    std::vector<unsigned> h(2 * G);
    for (int k = 0; k < n_cu; k++) {
        std::vector<uint32_t> mask(words, 0);
        mask[k / 32] = 1u << (k % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, words, mask.data()) != hipSuccess) { printf("bit %d: stream creation failed\n", k); return 1; }
        hipMemsetAsync(d, 0xff, G * 8, s);
        hipLaunchKernelGGL(where, dim3(G), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
        std::set<unsigned> seen;
        for (int b = 0; b < G; b++) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            seen.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
        }
        printf("bit %3d ->", k);
        for (unsigned v : seen) printf(" (xcd %u se %u sh %u cu %u)", v >> 12, (v >> 8) & 7, (v >> 4) & 1, v & 0xf);
        printf("%s\n", seen.size() == 1 ? "" : "   <-- more than one CU");
        hipStreamDestroy(s);
    }
    return 0;
}
