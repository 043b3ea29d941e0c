// BASELINE configs[4] ("template-bank-as-GEMM variant, 256-glyph bank, bf16 MFMA windows x templates; compare rocprof MFMA util
// vs LDS-NCC kernel"), settled with a number (tools only; not part of the product): the scan's item loop — 4 M-tiles of 16
// windows per wave and item, the bank streamed from LDS one 16-template N-tile at a time, window fragments read straight from
// the page with the 8-byte-row K layout (mfma_common.h, LAYOUT_W8), one "any accumulator above the threshold" test per N-tile —
// instantiated twice over the same pages and the same 256 templates of 8x15 taps:
//   i8    v_mfma_i32_16x16x64_i8:   2 K-steps per (M-tile, N-tile), operands as they lie in memory (one v_xor per dword)
//   bf16  v_mfma_f32_16x16x32_bf16: 4 K-steps per (M-tile, N-tile), every window byte converted to bf16 once per item
//         (u8 is exact in bf16; f32 accumulation is exact: |sum| <= 255 * 127 * 120 < 2^24), templates twice the LDS bytes
// The templates sum to zero, so sum (a - 128) b = sum a b and both forms see the same integers: the candidate counts must be
// equal (checked).  Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/c5_forms tools/c5_forms.hip ; run under
// rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES for the pipe utilisation (tools/r3_profiles.sh).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef v2i v2i_u __attribute__((aligned(1)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

constexpr int R_W = 608, R_H = 720, PITCH = 704, ROWS = R_H + 48, N_W = 8, N_H = 15, NT = 16, MT = 4, NW = 16;
constexpr int MTX = (R_W - N_W + 1 + 15) / 16, NROWS = R_H - N_H;

// two u8 -> two bf16 in one dword: the high halves of their f32 forms (exact: 8 significant bits)
__device__ __forceinline__ int pack_bf16(float lo, float hi) { return (int)__builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u); }

template <bool BF16>
__global__ __launch_bounds__(NW * 64, 4) void scan_form(const uint8_t *__restrict__ pages, int n_pages, const v4i *__restrict__ bank, int thr,
                                                       unsigned long long *__restrict__ n_cand) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    v4i *lb = reinterpret_cast<v4i *>(smem);
    constexpr int KS = BF16 ? 4 : 2;  // K-steps per N-tile
    for (int i = threadIdx.x; i < NT * KS * 64; i += NW * 64) lb[i] = bank[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const long n_items = ((long)n_pages * NROWS * MTX + MT - 1) / MT;
    unsigned long long found = 0;
    for (long item = (long)blockIdx.x * NW + (threadIdx.x >> 6); item < n_items; item += (long)gridDim.x * NW) {
        v4i frag[MT][KS];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const long m = min(item * MT + mt, (long)n_pages * NROWS * MTX - 1);
            const int col = (int)(m % MTX), row = (int)((m / MTX) % NROWS), pg = (int)(m / ((long)MTX * NROWS));
            const uint8_t *base = pages + ((size_t)pg * ROWS + 1 + row) * PITCH + 16 * col + r;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const uint8_t *p0 = base + (size_t)(2 * (4 * ks + g)) * PITCH;
                const v2i lo = *reinterpret_cast<const v2i_u *>(p0), hi = *reinterpret_cast<const v2i_u *>(p0 + PITCH);
                if (!BF16) {
                    frag[mt][ks] = v4i{lo[0], lo[1], hi[0], hi[1]} ^ (int)0x80808080;
                } else {  // 16 bytes -> 16 bf16 = two operands of 8: bytes 0..7 (row 2q) and 8..15 (row 2q + 1)
                    const int w[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        v4i o;
#pragma unroll
                        for (int d = 0; d < 2; d++) {
                            const unsigned x = (unsigned)w[2 * h + d];
                            o[2 * d] = pack_bf16((float)(x & 0xff), (float)((x >> 8) & 0xff));
                            o[2 * d + 1] = pack_bf16((float)((x >> 16) & 0xff), (float)(x >> 24));
                        }
                        frag[mt][2 * ks + h] = o;
                    }
                }
            }
        }
        v4i bf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) bf[ks] = lb[ks * 64 + lane];
        for (int nt = 0; nt < NT; nt++) {
            const int nxt = nt + 1 < NT ? nt + 1 : nt;
            bool any;
            if (!BF16) {
                v4i acc[MT];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) acc[mt] = v4i{-thr, -thr, -thr, -thr};
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bf[ks], frag[mt][ks], acc[mt], 0, 0, 0);
                    bf[ks] = lb[(nxt * KS + ks) * 64 + lane];
                    __builtin_amdgcn_sched_barrier(0);
                }
                int m = max(max(acc[0][0], acc[0][1]), max(acc[0][2], acc[0][3]));
#pragma unroll
                for (int mt = 1; mt < MT; mt++) m = max(max(max(m, acc[mt][0]), acc[mt][1]), max(acc[mt][2], acc[mt][3]));
                any = __builtin_amdgcn_ballot_w64(m > 0) != 0;
                if (any)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
                        if (item * MT + mt < (long)n_pages * NROWS * MTX)
#pragma unroll
                            for (int i = 0; i < 4; i++) found += acc[mt][i] > 0;
            } else {
                v4f acc[MT];
                const float c0 = -(float)thr;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) acc[mt] = v4f{c0, c0, c0, c0};
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, bf[ks]), __builtin_bit_cast(v8bf, frag[mt][ks]), acc[mt], 0, 0, 0);
                    bf[ks] = lb[(nxt * KS + ks) * 64 + lane];
                    __builtin_amdgcn_sched_barrier(0);
                }
                float m = fmaxf(fmaxf(acc[0][0], acc[0][1]), fmaxf(acc[0][2], acc[0][3]));
#pragma unroll
                for (int mt = 1; mt < MT; mt++) m = fmaxf(fmaxf(fmaxf(m, acc[mt][0]), acc[mt][1]), fmaxf(acc[mt][2], acc[mt][3]));
                any = __builtin_amdgcn_ballot_w64(m > 0.f) != 0;
                if (any)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
                        if (item * MT + mt < (long)n_pages * NROWS * MTX)
#pragma unroll
                            for (int i = 0; i < 4; i++) found += acc[mt][i] > 0.f;
            }
        }
    }
    if (found) atomicAdd(n_cand, found);
}

static uint16_t bf16_of(int v) {
    float f = (float)v;
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)(u >> 16);  // |v| < 256: exact
}

int main(int argc, char **argv) {
    const int n_pages = argc > 1 ? atoi(argv[1]) : 64, reps = 6;
    std::vector<uint8_t> pages((size_t)n_pages * ROWS * PITCH, 0);
    srand(11);
    for (int p = 0; p < n_pages; p++)
        for (int y = 0; y < R_H; y++)
            for (int x = 0; x < R_W; x++) pages[((size_t)p * ROWS + y) * PITCH + x] = (uint8_t)(rand() & 0xff);
    // 256 templates, 120 taps in [-30, 30], zero sum; tap k of template (nt, n): K-step ks = k / 64, lane group g = (k % 64) / 16, byte k % 16
    std::vector<int8_t> q8((size_t)NT * 2 * 64 * 16, 0);
    std::vector<uint16_t> qb((size_t)NT * 4 * 64 * 8, 0);
    for (int t = 0; t < NT * 16; t++) {
        int tap[128] = {0}, sum = 0;
        for (int k = 0; k < 120; k++) {  // taps of rows 0..14 (8 bytes each): k = 8 * row + x -> K position 16 * (row / 2) + 8 * (row % 2) + x = k
            tap[k] = rand() % 61 - 30;
            sum += tap[k];
        }
        for (int k = 0; sum != 0; k = (k + 1) % 120) {  // zero the sum without leaving [-30, 30]
            if (sum > 0 && tap[k] > -30) tap[k]--, sum--;
            else if (sum < 0 && tap[k] < 30) tap[k]++, sum++;
        }
        const int nt = t / 16, n = t % 16;
        for (int k = 0; k < 128; k++) {
            const int ks = k / 64, g = (k % 64) / 16, b = k % 16;
            q8[(((size_t)nt * 2 + ks) * 64 + g * 16 + n) * 16 + b] = (int8_t)tap[k];
            qb[(((size_t)nt * 4 + 2 * ks + b / 8) * 64 + g * 16 + n) * 8 + b % 8] = bf16_of(tap[k]);
        }
    }
    uint8_t *d_pages;
    v4i *d_q8, *d_qb;
    unsigned long long *d_n;
    hipMalloc(&d_pages, pages.size());
    hipMalloc(&d_q8, q8.size());
    hipMalloc(&d_qb, qb.size() * 2);
    hipMalloc(&d_n, 8);
    hipMemcpy(d_pages, pages.data(), pages.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_q8, q8.data(), q8.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_qb, qb.data(), qb.size() * 2, hipMemcpyHostToDevice);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int thr = 60000;  // ~4 sigma of the random sums: a few candidates per 10^5 pairs, as in the real scan
    const double macs = (double)n_pages * NROWS * MTX * 16 * 256 * 120;  // true taps, searched windows incl. the M-tile padding
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double ms_of[2] = {0, 0};
    unsigned long long cand[2] = {0, 0};
    for (int form = 0; form < 2; form++) {
        const size_t lds = (size_t)NT * (form ? 4 : 2) * 1024;
        auto kern = form ? scan_form<true> : scan_form<false>;
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int rep = 0; rep < reps; rep++) {
            hipMemset(d_n, 0, 8);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(prop.multiProcessorCount), dim3(NW * 64), lds, 0, d_pages, n_pages, form ? d_qb : d_q8, thr, d_n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2) ms_of[form] += ms / (reps - 2);
        }
        hipMemcpy(&cand[form], d_n, 8, hipMemcpyDeviceToHost);
    }
    printf("{\"workload\": \"%d noise pages 608x720, 256 templates 8x15, the scan's item loop\", \"i8_mfma_16x16x64\": {\"ms\": %.3f, \"TMACs\": %.1f, \"candidates\": %llu}, "
           "\"bf16_mfma_16x16x32\": {\"ms\": %.3f, \"TMACs\": %.1f, \"candidates\": %llu}, \"bf16_over_i8_time\": %.2f, \"same_candidates\": %s}\n",
           n_pages, ms_of[0], macs / ms_of[0] / 1e9, cand[0], ms_of[1], macs / ms_of[1] / 1e9, cand[1], ms_of[1] / ms_of[0], cand[0] == cand[1] ? "true" : "false");
    return cand[0] == cand[1] ? 0 : 1;
}
