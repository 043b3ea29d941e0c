// Does the int8 MFMA rate of a power-limited chip depend on the operand VALUES?  The prefilter's window operand is (ink - 128): blank paper
// is 0x80 in every byte.  The loop of tools/mfma_rate.hip (2 K-steps, 4 M-tiles, LDS re-loads) over a window operand that is all 0x80,
// all 0x00, 85 % 0x80 / 15 % random, 85 % 0x00 / 15 % random (0..127), and random; templates random in +-40.  ~1 s per case, whole chip.
// With arguments (files of 65 536 int8, e.g. a real quantised bank): the TEMPLATE operand's values instead, over page-like and random windows.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_power.hip -o tools/bin/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024, 4) void k(const v4i *__restrict__ src, const v4i *__restrict__ bsrc, int *__restrict__ out, int iters, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4i *bank = (v4i *)smem;
    for (int i = threadIdx.x; i < ntiles * 2 * 64; i += 1024) bank[i] = bsrc[i % 4096];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    v4i a[4][2];
    for (int mt = 0; mt < 4; mt++)
        for (int ks = 0; ks < 2; ks++) a[mt][ks] = src[(blockIdx.x * 131 + threadIdx.x + 64 * (mt * 2 + ks)) % 4096];
    int found = 0;
    for (int it = 0; it < iters; it++) {
        v4i bf[2];
        for (int ks = 0; ks < 2; ks++) bf[ks] = bank[ks * 64 + lane];
        for (int nt = 0; nt < ntiles; nt++) {
            v4i acc[4];
            for (int mt = 0; mt < 4; mt++) acc[mt] = v4i{-1000000000, -1000000000, -1000000000, -1000000000};
            const int nxt = nt + 1 < ntiles ? nt + 1 : nt;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
#pragma unroll
                for (int mt = 0; mt < 4; mt++) acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bf[ks], a[mt][ks], acc[mt], 0, 0, 0);
                bf[ks] = bank[(nxt * 2 + ks) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
            }
            int m = acc[0][0];
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                m = max(m, max(acc[mt][0], acc[mt][1]));
                m = max(m, max(acc[mt][2], acc[mt][3]));
            }
            if (__builtin_amdgcn_ballot_w64(m > 0)) found++;
        }
    }
    if (found == 12345) out[threadIdx.x] = found;
}
int main(int argc, char **argv) {
    const int ntiles = 24, n = 4096 * 16;
    std::vector<signed char> tmpl(n), win(n);
    srand(7);
    for (int i = 0; i < n; i++) tmpl[i] = (signed char)(rand() % 81 - 40);
    v4i *dsrc, *dbank;
    int *dout;
    hipMalloc(&dsrc, n);
    hipMalloc(&dbank, n);
    hipMalloc(&dout, 4096);
    hipMemcpy(dbank, tmpl.data(), n, hipMemcpyHostToDevice);
    const size_t lds_b = (size_t)24 * 2 * 1024;
    if (argc > 1) {  // template operand VALUES: files of 65 536 int8 each, over page-like and random windows, alternating, three rounds
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        for (int round = 0; round < 3; round++)
            for (int c = 0; c < 2; c++) {
                srand(11 + c);
                for (int i = 0; i < n; i++) win[i] = c == 0 ? (rand() % 100 < 15 ? (signed char)(rand() & 255) : (signed char)0x80) : (signed char)(rand() & 255);
                hipMemcpy(dsrc, win.data(), n, hipMemcpyHostToDevice);
                for (int f = 1; f <= argc; f++) {
                    if (f < argc) {
                        FILE *fp = fopen(argv[f], "rb");
                        if (!fp || fread(tmpl.data(), 1, n, fp) != (size_t)n) return 1;
                        fclose(fp);
                    } else {
                        srand(7);
                        for (int i = 0; i < n; i++) tmpl[i] = (signed char)(rand() % 81 - 40);
                    }
                    hipMemcpy(dbank, tmpl.data(), n, hipMemcpyHostToDevice);
                    hipEvent_t e0, e1;
                    hipEventCreate(&e0);
                    hipEventCreate(&e1);
                    const int iters = 6000;
                    hipLaunchKernelGGL(k, dim3(256), dim3(1024), lds_b, 0, dsrc, dbank, dout, 400, 24);
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k, dim3(256), dim3(1024), lds_b, 0, dsrc, dbank, dout, iters, 24);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms = 0;
                    hipEventElapsedTime(&ms, e0, e1);
                    printf("windows %-10s templates %-28s %8.1f ms  %7.1f TMAC/s\n", c == 0 ? "page-like" : "random", f < argc ? argv[f] : "random in +-40", ms,
                           256.0 * 16 * iters * 24 * 2 * 4 * 16384 / ms / 1e9);
                }
            }
        return 0;
    }
    const size_t lds = (size_t)ntiles * 2 * 1024;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const char *names[] = {"all 0x80 (blank paper as ink - 128)", "all 0x00", "85% 0x80 + 15% random", "85% 0x00 + 15% random 0..127", "random bytes"};
    for (int round = 0; round < 2; round++)
        for (int c = 0; c < 5; c++) {
            for (int i = 0; i < n; i++) {
                const bool ink = rand() % 100 < 15;
                win[i] = c == 0 ? (signed char)0x80 : c == 1 ? 0 : c == 2 ? (ink ? (signed char)(rand() & 255) : (signed char)0x80) : c == 3 ? (ink ? (signed char)(rand() & 127) : 0) : (signed char)(rand() & 255);
            }
            hipMemcpy(dsrc, win.data(), n, hipMemcpyHostToDevice);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            const int iters = 12000;
            hipLaunchKernelGGL(k, dim3(256), dim3(1024), lds, 0, dsrc, dbank, dout, 400, ntiles);  // warm
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(1024), lds, 0, dsrc, dbank, dout, iters, ntiles);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double macs = 256.0 * 16 * iters * ntiles * 2 * 4 * 16384;
            printf("%-40s %8.1f ms  %7.1f TMAC/s\n", names[c], ms, macs / ms / 1e9);
        }
    return 0;
}
