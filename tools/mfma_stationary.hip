// What would a TEMPLATE-STATIONARY prefilter loop sustain?  Two bare loops on the same operands, alternating, whole chip:
//   A  the product kernel's form (tools/mfma_power.hip): the bank in LDS, a wave holds 4 M-tiles of windows in registers and walks the
//      N-tiles: per N-tile 2 LDS reads (B), 8 MFMAs, a test over 16 accumulator registers              (2.1 vector instr. per MFMA with the prologue)
//   B  each wave keeps NTW N-tiles of the bank in registers for the whole launch; the workgroup stages SUPER M-tiles of windows (and their
//      C-in, replicated x4) in LDS, double-buffered, one barrier per super-item; per M-tile a wave reads 2 A fragments + C-in from LDS,
//      issues 2 * NTW MFMAs and tests 4 * NTW accumulator registers
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_stationary.hip -o tools/bin/mfma_stationary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024, 4) void loop_a(const v4i *__restrict__ src, const v4i *__restrict__ bsrc, int *__restrict__ out, int iters, int ntiles) {
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

// B: NTW N-tiles per wave in registers; SUPER M-tiles per super-item in LDS (2 KB of A fragments + 1 KB of C-in each), two buffers
template <int NTW, int SUPER>
__global__ __launch_bounds__(1024, 4) void loop_b(const v4i *__restrict__ src, const v4i *__restrict__ bsrc, int *__restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4i *stage = (v4i *)smem;  // [2][SUPER][3][64]: A k-step 0, A k-step 1, C-in
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4i b[NTW][2];
#pragma unroll
    for (int nt = 0; nt < NTW; nt++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) b[nt][ks] = bsrc[(((wave * NTW + nt) * 2 + ks) * 64 + lane) % 4096];
    constexpr int PER_BUF = SUPER * 3 * 64;  // v4i per buffer
    auto fill = [&](int buf, int it) {
        for (int i = threadIdx.x; i < PER_BUF; i += 1024) {
            const int part = (i / 64) % 3;
            stage[buf * PER_BUF + i] = part == 2 ? v4i{-1000000000, -1000000000, -1000000000, -1000000000} : src[(blockIdx.x * 131 + it * 977 + i) % 4096];
        }
    };
    fill(0, 0);
    __syncthreads();
    int found = 0;
    for (int it = 0; it < iters; it++) {
        const int buf = it & 1;
        fill(buf ^ 1, it + 1);  // the next super-item's windows on their way while this one is multiplied
        const v4i *st = stage + buf * PER_BUF;
        v4i a0 = st[lane], a1 = st[64 + lane], cin = st[128 + lane];
#pragma unroll 2
        for (int mt = 0; mt < SUPER; mt++) {
            v4i acc[NTW];
            const int nx = mt + 1 < SUPER ? mt + 1 : mt;
#pragma unroll
            for (int nt = 0; nt < NTW; nt++) acc[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b[nt][0], a0, cin, 0, 0, 0);
            const v4i n0 = st[nx * 192 + lane], ncin = st[nx * 192 + 128 + lane];
#pragma unroll
            for (int nt = 0; nt < NTW; nt++) acc[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b[nt][1], a1, acc[nt], 0, 0, 0);
            const v4i n1 = st[nx * 192 + 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
            int m = acc[0][0];
#pragma unroll
            for (int nt = 0; nt < NTW; nt++) {
                m = max(m, max(acc[nt][0], acc[nt][1]));
                m = max(m, max(acc[nt][2], acc[nt][3]));
            }
            if (__builtin_amdgcn_ballot_w64(m > 0)) found++;
            a0 = n0;
            a1 = n1;
            cin = ncin;
        }
        __syncthreads();
    }
    if (found == 12345) out[threadIdx.x] = found;
}

template <typename F>
static float timed(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int n = 4096 * 16;
    std::vector<signed char> tmpl(n), win(n);
    srand(7);
    for (int i = 0; i < n; i++) tmpl[i] = (signed char)(rand() % 81 - 40);
    v4i *dsrc, *dbank;
    int *dout;
    hipMalloc(&dsrc, n);
    hipMalloc(&dbank, n);
    hipMalloc(&dout, 4096);
    hipMemcpy(dbank, tmpl.data(), n, hipMemcpyHostToDevice);
    constexpr int NT_A = 24;
    const size_t lds_a = (size_t)NT_A * 2 * 1024;
    hipFuncSetAttribute((const void *)loop_a, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a);
    constexpr int S8 = 8, S16 = 16;
    const size_t lds_b8 = 2 * S8 * 3 * 1024, lds_b16 = 2 * S16 * 3 * 1024;
    hipFuncSetAttribute((const void *)loop_b<6, S8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b8);
    hipFuncSetAttribute((const void *)loop_b<6, S16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b16);
    hipFuncSetAttribute((const void *)loop_b<2, S16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b16);
    hipFuncSetAttribute((const void *)loop_b<3, S16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b16);
    hipFuncSetAttribute((const void *)loop_b<4, S16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b16);
    const char *names[] = {"85% 0x80 + 15% random (page-like)", "random bytes"};
    for (int round = 0; round < 2; round++)
        for (int c = 0; c < 2; c++) {
            for (int i = 0; i < n; i++) {
                const bool ink = rand() % 100 < 15;
                win[i] = c == 0 ? (ink ? (signed char)(rand() & 255) : (signed char)0x80) : (signed char)(rand() & 255);
            }
            hipMemcpy(dsrc, win.data(), n, hipMemcpyHostToDevice);
            printf("%s\n", names[c]);
            {
                const int iters = 6000;
                hipLaunchKernelGGL(loop_a, dim3(256), dim3(1024), lds_a, 0, dsrc, dbank, dout, 300, NT_A);
                const float ms = timed([&] { hipLaunchKernelGGL(loop_a, dim3(256), dim3(1024), lds_a, 0, dsrc, dbank, dout, iters, NT_A); });
                printf("  A  bank in LDS, 4 M-tiles in registers, 24 N-tiles          %8.1f ms  %7.1f TMAC/s\n", ms, 256.0 * 16 * iters * NT_A * 2 * 4 * 16384 / ms / 1e9);
            }
#define RUN_B(NTW, SUPER, LDS, TXT)                                                                                                       \
    {                                                                                                                                      \
        const int iters = 6000 * 24 * 4 / (NTW * SUPER);                                                                                   \
        hipLaunchKernelGGL((loop_b<NTW, SUPER>), dim3(256), dim3(1024), LDS, 0, dsrc, dbank, dout, 300);                                   \
        const float ms = timed([&] { hipLaunchKernelGGL((loop_b<NTW, SUPER>), dim3(256), dim3(1024), LDS, 0, dsrc, dbank, dout, iters); }); \
        printf("  B  %-58s %8.1f ms  %7.1f TMAC/s\n", TXT, ms, 256.0 * 16 * iters * NTW * 2 * SUPER * 16384 / ms / 1e9);                    \
    }
            RUN_B(6, S16, lds_b16, "6 N-tiles per wave in registers, 16 M-tiles per barrier");
            RUN_B(6, S8, lds_b8, "6 N-tiles per wave in registers, 8 M-tiles per barrier");
            RUN_B(4, S16, lds_b16, "4 N-tiles per wave in registers, 16 M-tiles per barrier");
            RUN_B(3, S16, lds_b16, "3 N-tiles per wave in registers, 16 M-tiles per barrier");
            RUN_B(2, S16, lds_b16, "2 N-tiles per wave in registers, 16 M-tiles per barrier");
        }
    return 0;
}
