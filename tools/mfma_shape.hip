// microbenchmark (tools only): the prefilter's N-tile loop at 2 K-steps of 64 bytes in its two MFMA shapes, with the shader clock
// measured inside the kernel (s_memtime against the 100 MHz wall clock), so that "slower" can be told from "clocked lower":
//   form 16: v_mfma_i32_16x16x64_i8, 4 M-tiles of 16 windows, 24 N-tiles of 16 templates, 2 K-steps (the product kernel's loop)
//   form 32: v_mfma_i32_32x32x32_i8, 2 M-tiles of 32 windows, 12 N-tiles of 32 templates, 4 K-steps of 32 bytes; C-in 0 (inline
//            constant) and the lane's threshold compared after the max (a lane's 16 outputs are all its own window's)
// MODE 0 = MFMAs only, 1 = + bank re-load from LDS, 2 = + the "any output above the threshold" test.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_shape.hip -o tools/bin/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct Clk {
    unsigned long long cyc, wall;
};

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void k16(const v4i *__restrict__ src, const v4i *__restrict__ bsrc, int *__restrict__ out, int iters, Clk *clk) {
    constexpr int MT = 4, KS = 2, NT = 24;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4i *bank = (v4i *)smem;
    for (int i = threadIdx.x; i < NT * KS * 64; i += NW * 64) bank[i] = bsrc[i % 4096];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    v4i a[MT][KS];
    int ci[MT];
    for (int mt = 0; mt < MT; mt++) {
        for (int ks = 0; ks < KS; ks++) a[mt][ks] = src[(threadIdx.x + 64 * (mt * KS + ks)) % 4096];
        ci[mt] = -1000000000 - lane - mt;
    }
    int found = 0;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        v4i bf[KS];
        for (int ks = 0; ks < KS; ks++) bf[ks] = bank[ks * 64 + lane];
        for (int nt = 0; nt < NT; nt++) {
            v4i acc[MT];
            for (int mt = 0; mt < MT; mt++) acc[mt] = v4i{ci[mt], ci[mt], ci[mt], ci[mt]};
            const int nxt = nt + 1 < NT ? nt + 1 : nt;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bf[ks], a[mt][ks], acc[mt], 0, 0, 0);
                if (MODE >= 1) {
                    bf[ks] = bank[(nxt * KS + ks) * 64 + lane];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE >= 2) {
                int m = max(max(acc[0][0], acc[0][1]), max(acc[0][2], acc[0][3]));
#pragma unroll
                for (int mt = 1; mt < MT; mt++) {
                    m = max(max(m, acc[mt][0]), acc[mt][1]);
                    m = max(max(m, acc[mt][2]), acc[mt][3]);
                }
                if (__builtin_amdgcn_ballot_w64(m > 0)) found++;
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) asm volatile("" ::"v"(acc[mt]));
            }
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = Clk{c1 - c0, w1 - w0};
    if (found == 12345) out[threadIdx.x] = found;
}

template <int MODE, int NW, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void k32(const v4i *__restrict__ src, const v4i *__restrict__ bsrc, int *__restrict__ out, int iters, Clk *clk) {
    constexpr int MT = 2, KS = 4, NT = 12;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4i *bank = (v4i *)smem;
    for (int i = threadIdx.x; i < NT * KS * 64; i += NW * 64) bank[i] = bsrc[i % 4096];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    v4i a[MT][KS];
    int thr[MT];
    for (int mt = 0; mt < MT; mt++) {
        for (int ks = 0; ks < KS; ks++) a[mt][ks] = src[(threadIdx.x + 64 * (mt * KS + ks)) % 4096];
        thr[mt] = 1000000000 + lane + mt;
    }
    int found = 0;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        v4i bf[KS];
        for (int ks = 0; ks < KS; ks++) bf[ks] = bank[ks * 64 + lane];
        for (int nt = 0; nt < NT; nt++) {
            v16i acc[MT];
            const int nxt = nt + 1 < NT ? nt + 1 : nt;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    if (ks == 0) {
                        const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[ks], a[mt][ks], z, 0, 0, 0);
                    } else {
                        acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[ks], a[mt][ks], acc[mt], 0, 0, 0);
                    }
                }
                if (MODE >= 1) {
                    bf[ks] = bank[(nxt * KS + ks) * 64 + lane];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE >= 2) {
                bool any = false;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    int m = max(max(acc[mt][0], acc[mt][1]), max(acc[mt][2], acc[mt][3]));
#pragma unroll
                    for (int i = 4; i < 16; i += 2) m = max(max(m, acc[mt][i]), acc[mt][i + 1]);
                    any |= m > thr[mt];
                }
                if (__builtin_amdgcn_ballot_w64(any)) found++;
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) asm volatile("" ::"v"(acc[mt]));
            }
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = Clk{c1 - c0, w1 - w0};
    if (found == 12345) out[threadIdx.x] = found;
}

template <typename K>
static void run(K kern, int nw, const v4i *d, const v4i *db, int *o, Clk *dclk, const char *name) {
    const int iters = 300;
    const size_t lds = 24 * 2 * 1024;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(nw * 64), lds, 0, d, db, o, iters, dclk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        Clk c;
        hipMemcpy(&c, dclk, sizeof c, hipMemcpyDeviceToHost);
        const double macs = 256.0 * nw * iters * 24 * 2 * 4 * 16384;  // both forms: 64 windows x 384 templates x 128 K per wave and iteration
        const double ghz = (double)c.cyc / (double)c.wall * 0.1;
        const double pipe_cycles = (double)iters * 24 * 2 * 4 * 16 * (nw / 4);  // MFMA pipe cycles per SIMD
        if (rep == 2)
            printf("%-34s %.3f ms  %.2f PMAC/s  s_memtime %.2f GHz-equivalent  wave-0 loop %.0f memtime ticks, MFMA pipe needs %.0f cycles per SIMD\n", name, ms, macs / ms / 1e12,
                   ghz, (double)c.cyc, pipe_cycles);
    }
}

int main() {
    v4i *d, *db;
    int *o;
    Clk *dclk;
    hipMalloc(&d, 4096 * 16);
    hipMalloc(&db, 4096 * 16);
    hipMalloc(&o, 4096 * 4);
    hipMalloc(&dclk, sizeof(Clk));
    unsigned *h = (unsigned *)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 4; i++) h[i] = rand() * 2654435761u;
    hipMemcpy(d, h, 4096 * 16, hipMemcpyHostToDevice);
    srand(7);
    for (int i = 0; i < 4096 * 4; i++) {
        int v[4];
        for (int q = 0; q < 4; q++) v[q] = (rand() % 3 == 0) ? 0 : (rand() % 61 - 30);
        h[i] = (v[0] & 0xff) | ((v[1] & 0xff) << 8) | ((v[2] & 0xff) << 16) | ((unsigned)(v[3] & 0xff) << 24);
    }
    hipMemcpy(db, h, 4096 * 16, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; round++) {
        run(k16<0, 16>, 16, d, db, o, dclk, "16x16x64 MFMA only        16 waves");
        run(k16<1, 16>, 16, d, db, o, dclk, "16x16x64 + re-load        16 waves");
        run(k16<2, 16>, 16, d, db, o, dclk, "16x16x64 + re-load + test 16 waves");
        run(k32<0, 16, 4>, 16, d, db, o, dclk, "32x32x32 MFMA only        16 waves");
        run(k32<1, 16, 4>, 16, d, db, o, dclk, "32x32x32 + re-load        16 waves");
        run(k32<2, 16, 4>, 16, d, db, o, dclk, "32x32x32 + re-load + test 16 waves");
        run(k32<2, 12, 3>, 12, d, db, o, dclk, "32x32x32 + re-load + test 12 waves");
        run(k32<2, 8, 2>, 8, d, db, o, dclk, "32x32x32 + re-load + test  8 waves");
    }
    return 0;
}
