// microbenchmark: sustained v_mfma_i32_16x16x64_i8 issue rate in the prefilter's loop shape (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int MODE, int MT, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void k(const v4i* __restrict__ src, const v4i* __restrict__ bsrc, int* __restrict__ out, int iters, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4i* bank = (v4i*)smem;
    for (int i = threadIdx.x; i < ntiles * 3 * 64; i += NW * 64) bank[i] = bsrc[i % 4096];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    v4i a[MT][3], nl[MT];
    for (int mt = 0; mt < MT; mt++) { for (int ks = 0; ks < 3; ks++) a[mt][ks] = src[(threadIdx.x + 64 * (mt * 3 + ks)) % 4096]; nl[mt] = v4i{-1000000000, -1000000000, -1000000000, -1000000000}; }
    int found = 0;
    for (int it = 0; it < iters; it++) {
        v4i bf[3];
        for (int ks = 0; ks < 3; ks++) bf[ks] = bank[ks * 64 + lane];
        for (int nt = 0; nt < ntiles; nt++) {
            v4i acc[MT];
            for (int mt = 0; mt < MT; mt++) acc[mt] = nl[mt];
            const int nxt = nt + 1 < ntiles ? nt + 1 : nt;
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[mt][ks], bf[ks], acc[mt], 0, 0, 0);
                if (MODE >= 1) { bf[ks] = bank[(nxt * 3 + ks) * 64 + lane]; __builtin_amdgcn_sched_barrier(0); }
            }
            if (MODE >= 2) {
                int m = acc[0][0];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) { m = max(m, max(acc[mt][0], acc[mt][1])); m = max(m, max(acc[mt][2], acc[mt][3])); }
                if (__builtin_amdgcn_ballot_w64(m > 0)) found++;
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) asm volatile("" ::"v"(acc[mt]));
            }
        }
    }
    if (found == 12345) out[threadIdx.x] = found;
}
// the same loop with v_mfma_i32_32x32x32_i8: 2 M-tiles of 32 windows, N-tiles of 32 templates, 6 K-steps of 32 bytes; C-in = 0 and an
// explicit per-lane threshold (a 16-register C-in splat per M-tile does not fit the register budget)
typedef int v16i __attribute__((ext_vector_type(16)));
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void k32(const v4i* __restrict__ src, const v4i* __restrict__ bsrc, int* __restrict__ out, int iters, int ntiles32) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4i* bank = (v4i*)smem;
    for (int i = threadIdx.x; i < ntiles32 * 6 * 64; i += NW * 64) bank[i] = bsrc[i % 4096];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    v4i a[2][6];
    for (int m = 0; m < 2; m++) for (int s = 0; s < 6; s++) a[m][s] = src[(threadIdx.x + 64 * (m * 6 + s)) % 4096];
    const int thr0 = 1000000000 + lane, thr1 = 1000000001 + lane;
    int found = 0;
    for (int it = 0; it < iters; it++) {
        v4i bf[6];
        for (int s = 0; s < 6; s++) bf[s] = bank[s * 64 + lane];
        for (int nt = 0; nt < ntiles32; nt++) {
            v16i acc[2];
            for (int m = 0; m < 2; m++) for (int i = 0; i < 16; i++) acc[m][i] = 0;
            const int nxt = nt + 1 < ntiles32 ? nt + 1 : nt;
#pragma unroll
            for (int s = 0; s < 6; s++) {
#pragma unroll
                for (int m = 0; m < 2; m++) acc[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[s], a[m][s], acc[m], 0, 0, 0);
                if (MODE >= 1) { bf[s] = bank[(nxt * 6 + s) * 64 + lane]; __builtin_amdgcn_sched_barrier(0); }
            }
            if (MODE >= 2) {
                int m0 = acc[0][0], m1 = acc[1][0];
#pragma unroll
                for (int i = 1; i < 16; i += 2) { m0 = max(m0, max(acc[0][i], acc[0][(i + 1) & 15])); m1 = max(m1, max(acc[1][i], acc[1][(i + 1) & 15])); }
                if (__builtin_amdgcn_ballot_w64(m0 > thr0 || m1 > thr1)) found++;
            } else {
#pragma unroll
                for (int m = 0; m < 2; m++) asm volatile("" ::"v"(acc[m]));
            }
        }
    }
    if (found == 12345) out[threadIdx.x] = found;
}
template <int MODE, int NW>
void run32(const v4i* d, const v4i* db, int* o, const char* name) {
    const int ntiles32 = 12, iters = 200;
    size_t lds = ntiles32 * 6 * 1024;
    auto kern = k32<MODE, NW>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(NW * 64), lds, 0, d, db, o, iters, ntiles32);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double mf = 256.0 * NW * iters * ntiles32 * 6 * 2;
        if (rep == 2) printf("%s: %.3f ms, %.1f G MFMA/s, %.1f TMAC/s\n", name, ms, mf / ms / 1e6, mf * 32768 / ms / 1e9);
    }
}
template <int MODE, int MT, int NW>
void run(const v4i* d, const v4i* db, int* o, const char* name) {
    const int ntiles = 24, iters = 200;
    size_t lds = ntiles * 3 * 1024;
    auto kern = k<MODE, MT, NW>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(NW * 64), lds, 0, d, db, o, iters, ntiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double mf = 256.0 * NW * iters * ntiles * 3 * MT;
        if (rep == 2) printf("%s: %.3f ms, %.1f G MFMA/s, %.2f cycles/MFMA/SIMD @2.4GHz, %.1f TMAC/s\n", name, ms, mf / ms / 1e6, 1024 * 2.4e9 / (mf / (ms * 1e-3)), mf * 16384 / ms / 1e9);
    }
}
int main(int argc, char** argv) {
    v4i* d; int* o; hipMalloc(&d, 4096 * 16); hipMalloc(&o, 4096);
    unsigned* h = (unsigned*)malloc(4096 * 16); srand(1); for (int i = 0; i < 4096 * 4; i++) h[i] = rand() * 2654435761u;
    // optional: argv[1] = hex byte pattern for the A operands (first 1536 v4i = a[][] sources), e.g. 80 or 00
    if (argc > 1) { unsigned b = strtoul(argv[1], 0, 16) & 0xff; unsigned w = b * 0x01010101u; int frac = argc > 2 ? atoi(argv[2]) : 100;
        for (int i = 0; i < 4096 * 4; i++) if ((rand() % 100) < frac) h[i] = w; printf("A/B words set to 0x%08x with probability %d%%\n", w, frac); }
    v4i* db; hipMalloc(&db, 4096 * 16);
    { unsigned* hb = (unsigned*)malloc(4096 * 16); srand(7); for (int i = 0; i < 4096 * 4; i++) { int v[4]; for (int q = 0; q < 4; q++) v[q] = (rand() % 3 == 0) ? 0 : (rand() % 61 - 30); hb[i] = (v[0] & 0xff) | ((v[1] & 0xff) << 8) | ((v[2] & 0xff) << 16) | ((unsigned)(v[3] & 0xff) << 24); }
      hipMemcpy(db, hb, 4096 * 16, hipMemcpyHostToDevice); }  // B: template-like small ints, a third zeros
    hipMemcpy(d, h, 4096 * 16, hipMemcpyHostToDevice);
    run<0, 8, 8>(d, db, o, "mfma only      MT8 NW8 ");
    run<1, 8, 8>(d, db, o, "+ B reload     MT8 NW8 ");
    run<2, 8, 8>(d, db, o, "+ max/ballot   MT8 NW8 ");
    run<0, 4, 16>(d, db, o, "mfma only      MT4 NW16");
    run<2, 4, 16>(d, db, o, "+ all          MT4 NW16");
    run<2, 6, 12>(d, db, o, "+ all          MT6 NW12");
    run32<0, 16>(d, db, o, "32x32x32 mfma only NW16");
    run32<1, 16>(d, db, o, "32x32x32 + reload  NW16");
    run32<2, 16>(d, db, o, "32x32x32 + all     NW16");
    run<0, 4, 16>(d, db, o, "mfma only      MT4 NW16");
    run<2, 4, 16>(d, db, o, "+ all          MT4 NW16");
    run32<2, 16>(d, db, o, "32x32x32 + all     NW16");
    run<0, 8, 4>(d, db, o, "mfma only      MT8 NW4 ");
    run<2, 8, 4>(d, db, o, "+ all          MT8 NW4 ");
    return 0;
}
