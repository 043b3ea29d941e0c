// scan_mfma.hip — the fast scan: i8 MFMA conservative prefilter + exact verify.
//
// The reference evaluates, for every window w and template t (src/ncc.cpp:302-392),
//     sim = num / (norm_n * norm_p),  num = sum_k a_k b_k - s_n s_p / n = sum_k a_k (b_k - mean_t)
// and emits iff sim > thr.  Almost no (w, t) pair passes, so the device splits the work:
//
//  1. window statistics (stats_kernel): per size class and window, the exact integer sums s_p, s2_p
//     and  negL(w) = -floor(kappa * norm_p(w)) , or -REJECT for windows the reference never emits
//     (x = 0, y = 0, out of range, zero variance => rnorm = inf/NaN, src/ncc.rs:309-311).
//  2. MFMA prefilter (scan_mfma_kernel): every template is mean-centred, scaled by a bank-wide
//     constant c/norm_n(t) and rounded to int8 with the rounding chosen so that sum_k bq_k = 0.
//     G(w,t) = sum_k (a_k - 128) bq_k  (= sum_k a_k bq_k) is one v_mfma_i32_16x16x64_i8 chain over
//     the window's bytes (M = 16 windows, N = 16 templates, K = 64 bytes per instruction) with
//     C-in = negL(w), so "D > 0" <=> G > kappa*norm_p.  Cauchy-Schwarz bounds the rounding error:
//         | c*num/norm_n - G | = | sum_k (a_k - mean_w) e_k | <= norm_p * ||e_t||_2
//     hence sim > thr  ==>  G > (c*thr - max_t ||e_t||) * norm_p =: kappa * norm_p.  The filter
//     has no false negatives; kappa carries an extra relative margin for the f64 roundings of
//     the exact formula.  Survivors (a few per 10^5 pairs) go to a candidate list.
//  3. exact verify (verify_kernel): the reference formula, operation for operation (common.h),
//     on every candidate -> unordered hit list -> order.hip.
//
// Layout: A operand = windows.  Lane (r = lane&15, g = lane>>4) of a K-step holds 16 bytes of
// window x0+r: the 16-byte k-group q = 4*kstep + g, which is image row y+q, columns x..x+15
// (rows_per_group = 1, n_w in 9..16) or rows y+2q, y+2q+1, columns x..x+7 each (rows_per_group =
// 2, n_w <= 8).  B operand = the quantised bank, staged once per block in LDS in exactly the
// per-lane order the MFMA wants (1 KiB contiguous per ds_read_b128 wave-instruction, no bank
// conflicts).  Every A fragment is built once per (tile, class) and reused for all templates.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "mfma_common.h"

namespace focr {

int ensure_hit_capacity(focr_ctx *c, size_t want);

constexpr int MWAVES = 8;                 // window rows (= waves) per block tile
constexpr int MPITCH = 96;                // LDS image-tile pitch in bytes (24 dwords: conflict-free A reads)
constexpr size_t BANK_LDS_BUDGET = 72 << 10;  // bytes of LDS per block for the bank chunk (2 blocks / CU)

// ---------------------------------------------------------------------------------------------
// 1. window statistics -> negL table
constexpr int STX = 64, STY = 4, SLDW = 21;

template <int NDW, int MAXH>
__global__ __launch_bounds__(256) void stats_kernel(const uint8_t *__restrict__ pages, uint32_t pitch, uint32_t rows_alloc,
                                                    uint32_t r_w, uint32_t r_h, uint32_t n_w, uint32_t n_h, double kappa,
                                                    int32_t *__restrict__ negL, uint32_t Lpitch, uint32_t Lrows) {
    constexpr int LROWS = STY + MAXH - 1;
    __shared__ uint32_t tile[LROWS][SLDW];
    const uint32_t page = blockIdx.z, x0 = blockIdx.x * STX, y0 = blockIdx.y * STY;
    const uint8_t *pg = pages + (size_t)page * rows_alloc * pitch;
    for (uint32_t i = threadIdx.x; i < LROWS * SLDW; i += 256) {
        uint32_t r = i / SLDW, cdw = i % SLDW;
        uint32_t gy = y0 + r, gx = x0 + cdw * 4;
        uint32_t v = 0;
        if (gy < rows_alloc && gx + 4 <= pitch) v = *reinterpret_cast<const uint32_t *>(pg + (size_t)gy * pitch + gx);
        tile[r][cdw] = v;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const uint32_t x = x0 + lane, y = y0 + wy;
    const uint32_t cb = lane >> 2, sh = lane & 3;
    uint32_t s_p = 0, s2_p = 0;
#pragma unroll
    for (int j = 0; j < MAXH; j++) {
#pragma unroll
        for (int k = 0; k < NDW; k++) {
            uint32_t lo = tile[wy + j][cb + k], hi = tile[wy + j][cb + k + 1];
            uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, sh);
            uint32_t keep = n_w >= (uint32_t)(4 * k + 4) ? 0xffffffffu
                            : (n_w <= (uint32_t)(4 * k) ? 0u : ((1u << (8 * (n_w - 4 * k))) - 1u));
            w = ((uint32_t)j < n_h) ? (w & keep) : 0u;
            s_p = __builtin_amdgcn_udot4(w, 0x01010101u, s_p, false);
            s2_p = __builtin_amdgcn_udot4(w, w, s2_p, false);
        }
    }
    if (x >= Lpitch || y >= Lrows) return;
    // searched windows: x in [1, r_w - n_w], y in [1, r_h - n_h]  (src/ncc.rs:279-282, src/ncc.cpp:302)
    const bool in_range = x >= 1 && y >= 1 && x + n_w <= r_w && y + n_h <= r_h;
    // same expression as patch_rnorm's argument (src/ncc.rs:309): norm2 <= 0 or NaN => rnorm = inf/NaN => never emitted
    const double norm2 = (double)s2_p - ((double)((uint64_t)s_p * (uint64_t)s_p)) / (double)(n_w * n_h);
    int32_t out = -REJECT;
    if (in_range && norm2 > 0.0) {
        double Lf = __builtin_floor(kappa * __builtin_sqrt(norm2)) - 2.0;
        Lf = __builtin_fmin(__builtin_fmax(Lf, -(double)(REJECT - 1)), (double)REJECT);
        out = -(int32_t)Lf;
    }
    negL[((size_t)page * Lrows + y) * Lpitch + x] = out;
}

// ---------------------------------------------------------------------------------------------
// 2. MFMA prefilter
template <int KSTEPS, int RPG, int MT>
__global__ __launch_bounds__(512, 4) void scan_mfma_kernel(
    const uint8_t *__restrict__ pages, uint32_t pitch, uint32_t rows_alloc, uint32_t tiles_x, uint32_t tiles_y,
    uint32_t n_pages, const v4i *__restrict__ qbank, uint32_t n_tiles16, uint32_t n_chunk,
    const int32_t *__restrict__ negL, uint32_t Lpitch, uint32_t Lrows, const uint32_t *__restrict__ tglobal,
    uint32_t n_total, uint64_t *__restrict__ cand, unsigned long long *__restrict__ cand_counter,
    unsigned long long cand_cap) {
    constexpr int TROWS = MWAVES + 4 * KSTEPS * RPG - 1;  // image rows a tile touches
    constexpr int TWID = 16 * MT;                         // windows per wave
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    v4i *bank = reinterpret_cast<v4i *>(smem);
    const uint32_t bank_vec = n_tiles16 * KSTEPS * 64;
    uint8_t *tile = smem + (size_t)bank_vec * 16;

    for (uint32_t i = threadIdx.x; i < bank_vec; i += 512) bank[i] = qbank[i];

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    uint64_t *wbuf = reinterpret_cast<uint64_t *>(tile + ((TROWS * MPITCH + 15) & ~15)) + w * WBUF;
    uint32_t wcount = 0;  // wave-uniform number of staged candidates
    const uint32_t total_tiles = n_pages * tiles_y * tiles_x;

    for (uint32_t tid = blockIdx.x; tid < total_tiles; tid += gridDim.x) {
        const uint32_t tx = tid % tiles_x, ty = (tid / tiles_x) % tiles_y, page = tid / (tiles_x * tiles_y);
        const uint32_t x0 = tx * TWID, y0 = ty * MWAVES;
        const uint8_t *pg = pages + (size_t)page * rows_alloc * pitch;
        __syncthreads();  // previous tile's fragments are built (and the bank is staged on the first pass)
        for (uint32_t i = threadIdx.x; i < TROWS * (MPITCH / 4); i += 512) {
            uint32_t rr = i / (MPITCH / 4), cdw = i % (MPITCH / 4);
            uint32_t gy = y0 + rr, gx = x0 + cdw * 4;
            uint32_t v = 0;
            if (gy < rows_alloc && gx + 4 <= pitch) v = *reinterpret_cast<const uint32_t *>(pg + (size_t)gy * pitch + gx);
            // u8 -> i8: a - 128 (the templates sum to zero, so the bias cancels exactly)
            reinterpret_cast<uint32_t *>(tile)[rr * (MPITCH / 4) + cdw] = v ^ 0x80808080u;
        }
        __syncthreads();

        // A fragments: 16 bytes per lane per (M-tile, K-step)
        v4i afrag[MT][KSTEPS];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const uint32_t col = 16 * mt + r, cb = col & ~3u, sh = col & 3u;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) {
                const int q = 4 * ks + g;
                if (RPG == 1) {
                    const uint32_t *p = reinterpret_cast<const uint32_t *>(tile + (w + q) * MPITCH + cb);
                    uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
                    afrag[mt][ks] = v4i{(int)__builtin_amdgcn_alignbyte(d1, d0, sh), (int)__builtin_amdgcn_alignbyte(d2, d1, sh),
                                        (int)__builtin_amdgcn_alignbyte(d3, d2, sh), (int)__builtin_amdgcn_alignbyte(d4, d3, sh)};
                } else {
                    const uint32_t *p0 = reinterpret_cast<const uint32_t *>(tile + (w + 2 * q) * MPITCH + cb);
                    const uint32_t *p1 = reinterpret_cast<const uint32_t *>(tile + (w + 2 * q + 1) * MPITCH + cb);
                    uint32_t a0 = p0[0], a1 = p0[1], a2 = p0[2], b0 = p1[0], b1 = p1[1], b2 = p1[2];
                    afrag[mt][ks] = v4i{(int)__builtin_amdgcn_alignbyte(a1, a0, sh), (int)__builtin_amdgcn_alignbyte(a2, a1, sh),
                                        (int)__builtin_amdgcn_alignbyte(b1, b0, sh), (int)__builtin_amdgcn_alignbyte(b2, b1, sh)};
                }
            }
        }
        // C-in: lane (r, g) owns output rows 4g..4g+3 (= windows x0+16mt+4g+i) of every M-tile
        const uint32_t y = y0 + w;
        v4i nl[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
            nl[mt] = *reinterpret_cast<const v4i *>(negL + ((size_t)page * Lrows + y) * Lpitch + x0 + 16 * mt + 4 * g);

        for (uint32_t nt = 0; nt < n_tiles16; nt++) {
            v4i acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = nl[mt];
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) {
                const v4i b = bank[(nt * KSTEPS + ks) * 64 + lane];
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[mt][ks], b, acc[mt], 0, 0, 0);
            }
            int m = acc[0][0];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                m = max(m, max(acc[mt][0], acc[mt][1]));
                m = max(m, max(acc[mt][2], acc[mt][3]));
            }
            if (__builtin_amdgcn_ballot_w64(m > 0) != 0) {  // wave-uniform, rare
                const uint32_t tl = nt * 16 + r;
                const uint32_t tg = tl < n_chunk ? tglobal[tl] : 0xffffffffu;  // dead / padding templates never emit
                const bool lane_ok = tg != 0xffffffffu;
                const uint64_t key_hi = ((uint64_t)(page * n_total + tg) << 32) | ((uint64_t)y << 16);
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const bool f = lane_ok && acc[mt][i] > 0;
                        const uint64_t mask = __builtin_amdgcn_ballot_w64(f);
                        if (mask) {  // wave-uniform
                            const uint32_t cnt = (uint32_t)__builtin_popcountll(mask);
                            if (wcount + cnt > WBUF) {
                                flush_wave_candidates(wbuf, wcount, lane, cand, cand_counter, cand_cap);
                                wcount = 0;
                            }
                            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                            if (f) wbuf[wcount + pos] = key_hi | (uint64_t)(x0 + 16 * mt + 4 * g + i);
                            wcount += cnt;
                        }
                    }
            }
        }
    }
    if (wcount) flush_wave_candidates(wbuf, wcount, lane, cand, cand_counter, cand_cap);
}

// ---------------------------------------------------------------------------------------------
// 3. exact verify: the reference arithmetic on every candidate
__global__ __launch_bounds__(256) void verify_kernel(const uint64_t *__restrict__ cand, unsigned long long n_cand,
                                                     const uint8_t *__restrict__ pages, uint32_t pitch, uint32_t rows_alloc,
                                                     uint32_t n_total, const uint32_t *__restrict__ order_of,
                                                     const TemplateConst *__restrict__ tc, const uint8_t *__restrict__ needles,
                                                     const uint32_t *__restrict__ needle_off, double thr_d,
                                                     uint64_t *__restrict__ hit_keys, float *__restrict__ hit_sims,
                                                     unsigned long long *__restrict__ counter, unsigned long long capacity) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cand) return;
    const uint64_t key = cand[i];
    const uint32_t seg = (uint32_t)(key >> 32), page = seg / n_total, t = seg % n_total;
    const uint32_t x = (uint32_t)(key & 0xffff), y = (uint32_t)((key >> 16) & 0xffff);
    const uint32_t ci = order_of[t];
    const TemplateConst c = tc[ci];
    const uint8_t *nd = needles + needle_off[ci];
    const uint8_t *pg = pages + ((size_t)page * rows_alloc + y) * pitch + x;
    uint32_t acc = 0, s_p = 0, s2_p = 0;
    for (uint32_t j = 0; j < c.n_h; j++)
        for (uint32_t k = 0; k < c.n_w; k++) {
            uint32_t a = pg[(size_t)j * pitch + k], b = nd[j * c.n_w + k];
            acc += a * b;    // src/ncc.cpp:316-321
            s_p += a;        // patch_sum,  src/ncc.rs:307, 310
            s2_p += a * a;   // window sum of squares, src/ncc.rs:308
        }
    const double rnorm_p = window_rnorm(s_p, (uint64_t)s2_p, (double)(c.n_w * c.n_h));
    const double sim = ncc_similarity(acc, s_p, c.s_n, c.n_recip, c.rnorm_n, rnorm_p);
    if (ncc_emits(sim, thr_d)) {
        unsigned long long idx = atomicAdd(counter, 1ull);
        if (idx < capacity) {
            hit_keys[idx] = key;
            hit_sims[idx] = (float)sim;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side

// Quantise the bank (see the header comment).  `dense` holds the class-ordered dense needles.
int build_mfma_bank(focr_ctx *c, const uint8_t *dense) {
    std::vector<int8_t> qbank;
    std::vector<uint32_t> tglobal(c->h_tconst.size(), 0xffffffffu), order_of(c->n_templates, 0);
    for (size_t k = 0; k < c->classes.size(); k++) {
        SizeClass &sc = c->classes[k];
        sc.rows_per_group = sc.n_w <= 8 ? 2 : 1;
        const uint32_t groups = (sc.n_h + sc.rows_per_group - 1) / sc.rows_per_group;
        sc.k_groups = (groups + 3) / 4 * 4;
        sc.n_tiles16 = (sc.n_templates + 15) / 16;
        sc.q_offset = (uint32_t)qbank.size();
        const uint32_t n = sc.n_w * sc.n_h, ksteps = sc.k_groups / 4;
        // bank-wide scale: 126 / max |b - mean| / norm_n over live templates
        double max_ratio = 0.0;
        std::vector<double> norm_n(sc.n_templates, 0.0), mean(sc.n_templates, 0.0);
        for (uint32_t i = 0; i < sc.n_templates; i++) {
            const TemplateConst &tc = c->h_tconst[sc.first + i];
            order_of[tc.index] = sc.first + i;
            const uint8_t *nd = dense + c->h_needle_off[sc.first + i];
            double s = 0, s2 = 0;
            for (uint32_t p = 0; p < n; p++) {
                s += nd[p];
                s2 += (double)nd[p] * nd[p];
            }
            mean[i] = s / n;
            double n2 = s2 - s * s / n;
            if (!(n2 > 0.0) || !std::isfinite(tc.rnorm_n)) continue;  // constant needle: rnorm_n = inf, never emits
            norm_n[i] = std::sqrt(n2);
            tglobal[sc.first + i] = tc.index;
            for (uint32_t p = 0; p < n; p++) max_ratio = std::max(max_ratio, std::fabs(nd[p] - mean[i]) / norm_n[i]);
        }
        const double c_scale = max_ratio > 0 ? 126.0 / max_ratio : 1.0;
        double e_max = 0.0;
        const size_t class_bytes = (size_t)sc.n_tiles16 * ksteps * 1024;
        qbank.resize(sc.q_offset + class_bytes, 0);
        std::vector<double> rk(n);
        std::vector<int> bq(n);
        std::vector<uint32_t> idx(n);
        for (uint32_t i = 0; i < sc.n_templates; i++) {
            if (norm_n[i] == 0.0) continue;
            const uint8_t *nd = dense + c->h_needle_off[sc.first + i];
            const double q = c_scale / norm_n[i];
            long sum = 0;
            for (uint32_t p = 0; p < n; p++) {
                rk[p] = q * (nd[p] - mean[i]);
                bq[p] = (int)std::floor(rk[p]);
                sum += bq[p];
                idx[p] = p;
            }
            // largest-remainder rounding so that the int8 template sums to exactly zero
            const long deficit = -sum;  // sum(rk) = 0 in exact arithmetic, so 0 <= deficit <= n
            if (deficit < 0 || deficit > (long)n) return fail(c, FOCR_ERR_INVALID, "mfma bank: rounding deficit out of range");
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return rk[a] - bq[a] > rk[b] - bq[b]; });
            for (long d = 0; d < deficit; d++) bq[idx[d]] += 1;
            double e2 = 0;
            long check = 0;
            for (uint32_t p = 0; p < n; p++) {
                if (bq[p] > 127 || bq[p] < -127) return fail(c, FOCR_ERR_INVALID, "mfma bank: quantised template out of int8 range");
                double e = rk[p] - bq[p];
                e2 += e * e;
                check += bq[p];
            }
            if (check != 0) return fail(c, FOCR_ERR_INVALID, "mfma bank: quantised template does not sum to zero");
            e_max = std::max(e_max, std::sqrt(e2));
            // scatter into the per-lane MFMA B layout: [n-tile][k-step][g][n][16 bytes]
            const uint32_t nt = i / 16, nn = i % 16;
            for (uint32_t j = 0; j < sc.n_h; j++)
                for (uint32_t x = 0; x < sc.n_w; x++) {
                    uint32_t qg, byte;
                    if (sc.rows_per_group == 1) {
                        qg = j;
                        byte = x;
                    } else {
                        qg = j / 2;
                        byte = 8 * (j % 2) + x;
                    }
                    const uint32_t ks = qg / 4, g = qg % 4;
                    qbank[sc.q_offset + ((size_t)(nt * ksteps + ks) * 64 + g * 16 + nn) * 16 + byte] = (int8_t)bq[j * sc.n_w + x];
                }
        }
        sc.kappa = 0.f;  // per-scan (depends on the threshold); keep the two ingredients
        c->mfma_c_scale.push_back(c_scale);
        c->mfma_e_max.push_back(e_max);
    }
    FOCR_HIP(c, hipMalloc((void **)&c->d_qbank, qbank.size() ? qbank.size() : 16));
    FOCR_HIP(c, hipMemcpy(c->d_qbank, qbank.data(), qbank.size(), hipMemcpyHostToDevice));
    FOCR_HIP(c, hipMalloc((void **)&c->d_tglobal, tglobal.size() * 4));
    FOCR_HIP(c, hipMemcpy(c->d_tglobal, tglobal.data(), tglobal.size() * 4, hipMemcpyHostToDevice));
    FOCR_HIP(c, hipMalloc((void **)&c->d_order_of, order_of.size() * 4));
    FOCR_HIP(c, hipMemcpy(c->d_order_of, order_of.data(), order_of.size() * 4, hipMemcpyHostToDevice));
    return FOCR_OK;
}

template <int NDW, int MAXH>
static void launch_stats(focr_ctx *c, const SizeClass &sc, double kappa, int32_t *negL, uint32_t Lpitch, uint32_t Lrows) {
    dim3 grid(Lpitch / STX, (Lrows + STY - 1) / STY, (unsigned)c->n_pages);
    hipLaunchKernelGGL((stats_kernel<NDW, MAXH>), grid, dim3(256), 0, c->stream, c->d_pages, (uint32_t)c->pitch,
                       (uint32_t)c->rows_alloc, (uint32_t)c->r_w, (uint32_t)c->r_h, sc.n_w, sc.n_h, kappa, negL, Lpitch, Lrows);
}

template <int KSTEPS, int RPG, int MT>
static void launch_mfma(focr_ctx *c, const MfmaLaunch &L, unsigned n_blocks) {
    const SizeClass &sc = *L.sc;
    const uint32_t twid = 16 * MT;
    const uint32_t tiles_x = (uint32_t)((c->r_w - sc.n_w + 1 + twid - 1) / twid);  // windows x in [0, r_w - n_w]
    const uint32_t tiles_y = (uint32_t)((c->r_h - sc.n_h + 1 + MWAVES - 1) / MWAVES);
    const uint32_t n_tiles16 = (L.chunk_n + 15) / 16;
    const size_t lds = (size_t)n_tiles16 * KSTEPS * 1024 + (((size_t)(MWAVES + 4 * KSTEPS * RPG - 1) * MPITCH + 15) & ~(size_t)15) +
                       (size_t)MWAVES * WBUF * 8;
    const uint64_t total_tiles = (uint64_t)tiles_x * tiles_y * c->n_pages;
    unsigned grid = (unsigned)std::min<uint64_t>(n_blocks, total_tiles);
    auto kern = scan_mfma_kernel<KSTEPS, RPG, MT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const v4i *qb = reinterpret_cast<const v4i *>(c->d_qbank + sc.q_offset + (size_t)(L.chunk_first / 16) * KSTEPS * 1024);
    const uint64_t issued = total_tiles * MWAVES * twid * (uint64_t)n_tiles16 * 16 * KSTEPS * 64;
    const uint64_t alg = (uint64_t)(c->r_w - sc.n_w) * (c->r_h - sc.n_h) * sc.n_w * sc.n_h * L.chunk_n * c->n_pages;
    char name[64];
    snprintf(name, sizeof name, "scan_mfma_kernel<%d,%d,%d>", KSTEPS, RPG, MT);
    c->launch_begin(name, L.chunk_n, alg, issued);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, c->stream, c->d_pages, (uint32_t)c->pitch, (uint32_t)c->rows_alloc,
                       tiles_x, tiles_y, (uint32_t)c->n_pages, qb, n_tiles16, L.chunk_n, L.negL, L.Lpitch, L.Lrows,
                       c->d_tglobal + sc.first + L.chunk_first, (uint32_t)c->n_templates, c->d_cand,
                       (unsigned long long *)c->d_counter + 1, (unsigned long long)c->cand_capacity);
    c->launch_end();
    c->counters[3] += issued;
}

static int dispatch_mfma(focr_ctx *c, const MfmaLaunch &L, unsigned n_blocks) {
    const uint32_t ks = L.sc->k_groups / 4, rpg = L.sc->rows_per_group;
#define CASE(K, R, M)                 \
    case (K) * 10 + (R):              \
        launch_mfma<K, R, M>(c, L, n_blocks); \
        break;
    switch (ks * 10 + rpg) {
        CASE(1, 1, 4) CASE(2, 1, 4) CASE(3, 1, 4) CASE(4, 1, 4) CASE(5, 1, 2) CASE(6, 1, 2) CASE(7, 1, 2) CASE(8, 1, 2)
        CASE(1, 2, 4) CASE(2, 2, 4) CASE(3, 2, 4) CASE(4, 2, 4)
        default: return fail(c, FOCR_ERR_INVALID, "scan_mfma: unsupported size class");
    }
#undef CASE
    FOCR_HIP(c, hipGetLastError());
    return FOCR_OK;
}

int launch_scan_mfma(focr_ctx *c, float threshold) {
    const double thr_d = (double)threshold;  // src/ncc.cpp:83, 288
    const uint32_t Lpitch = (uint32_t)((c->r_w + 63) / 64 * 64 + 64), Lrows = (uint32_t)((c->r_h + 7) / 8 * 8 + 8);
    const size_t L_per_class = c->n_pages * (size_t)Lrows * Lpitch;
    const size_t L_bytes = L_per_class * c->classes.size() * sizeof(int32_t);
    if (c->L_bytes < L_bytes) {
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_L) (void)hipFree(c->d_L);
        c->d_L = nullptr;
        c->L_bytes = 0;
        if (hipMalloc(&c->d_L, L_bytes) != hipSuccess) return fail(c, FOCR_ERR_NOMEM, "scan_mfma: hipMalloc(negL) failed");
        c->L_bytes = L_bytes;
    }
    int rc = ensure_hit_capacity(c, std::max<size_t>(c->hit_capacity, std::max<size_t>(1u << 20, c->n_pages * 65536)));
    if (rc) return rc;
    size_t want_cand = std::max<size_t>(c->cand_capacity, std::max<size_t>(1u << 21, c->n_pages * 131072));
    hipDeviceProp_t prop;
    FOCR_HIP(c, hipGetDeviceProperties(&prop, c->device));
    const unsigned n_blocks = 2u * (unsigned)prop.multiProcessorCount;
    int variant = 2;  // FOCR_MFMA_VARIANT=1 selects the LDS-tiled, barrier-per-tile kernel (kept for A/B)
    if (const char *e = getenv("FOCR_MFMA_VARIANT")) variant = atoi(e) == 1 ? 1 : 2;

    for (int attempt = 0; attempt < 4; attempt++) {
        if (c->cand_capacity < want_cand) {
            FOCR_HIP(c, hipStreamSynchronize(c->stream));
            if (c->d_cand) (void)hipFree(c->d_cand);
            c->d_cand = nullptr;
            c->cand_capacity = 0;
            if (hipMalloc(&c->d_cand, want_cand * 8) != hipSuccess) return fail(c, FOCR_ERR_NOMEM, "scan_mfma: hipMalloc(cand) failed");
            c->cand_capacity = want_cand;
        }
        c->counters[3] = 0;
        c->launches_reset();
        FOCR_HIP(c, hipMemsetAsync(c->d_counter, 0, 64 * sizeof(uint32_t), c->stream));
        FOCR_HIP(c, hipEventRecord(c->ev[0], c->stream));
        // 1. statistics
        for (size_t k = 0; k < c->classes.size(); k++) {
            const SizeClass &sc = c->classes[k];
            if (sc.n_w >= c->r_w || sc.n_h >= c->r_h) continue;
            const double cs = c->mfma_c_scale[k], em = c->mfma_e_max[k];
            const double kappa = cs * thr_d - em - 1e-4 * (cs * (1.0 + std::fabs(thr_d)) + em);
            int32_t *negL = c->d_L + k * L_per_class;
            switch (sc.ndw * 100 + sc.maxh) {
                case 116: launch_stats<1, 16>(c, sc, kappa, negL, Lpitch, Lrows); break;
                case 216: launch_stats<2, 16>(c, sc, kappa, negL, Lpitch, Lrows); break;
                case 316: launch_stats<3, 16>(c, sc, kappa, negL, Lpitch, Lrows); break;
                case 416: launch_stats<4, 16>(c, sc, kappa, negL, Lpitch, Lrows); break;
                case 132: launch_stats<1, 32>(c, sc, kappa, negL, Lpitch, Lrows); break;
                case 232: launch_stats<2, 32>(c, sc, kappa, negL, Lpitch, Lrows); break;
                case 332: launch_stats<3, 32>(c, sc, kappa, negL, Lpitch, Lrows); break;
                case 432: launch_stats<4, 32>(c, sc, kappa, negL, Lpitch, Lrows); break;
                default: return fail(c, FOCR_ERR_INVALID, "scan_mfma: unsupported size class");
            }
            FOCR_HIP(c, hipGetLastError());
        }
        FOCR_HIP(c, hipEventRecord(c->ev[1], c->stream));
        // 2. MFMA prefilter, one launch per (class, bank chunk that fits the LDS budget)
        for (size_t k = 0; k < c->classes.size(); k++) {
            const SizeClass &sc = c->classes[k];
            if (sc.n_w >= c->r_w || sc.n_h >= c->r_h) continue;
            const uint32_t ksteps = sc.k_groups / 4;
            const size_t budget = variant == 1 ? BANK_LDS_BUDGET : mfma2_bank_budget();
            const uint32_t chunk_max = (uint32_t)(budget / (ksteps * 1024)) * 16;
            for (uint32_t first = 0; first < sc.n_templates; first += chunk_max) {
                MfmaLaunch L{&sc, first, std::min(chunk_max, sc.n_templates - first), c->d_L + k * L_per_class, Lpitch, Lrows};
                rc = variant == 1 ? dispatch_mfma(c, L, n_blocks) : dispatch_mfma_v2(c, L, (unsigned)prop.multiProcessorCount);
                if (rc) return rc;
            }
        }
        FOCR_HIP(c, hipEventRecord(c->ev[2], c->stream));
        unsigned long long n_cand = 0;
        FOCR_HIP(c, hipMemcpyAsync(&n_cand, (unsigned long long *)c->d_counter + 1, 8, hipMemcpyDeviceToHost, c->stream));
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        if (n_cand > c->cand_capacity) {
            if (n_cand > ((unsigned long long)1 << 33)) return fail(c, FOCR_ERR_OVERFLOW, "scan_mfma: more than 2^33 candidates; scan fewer pages per call");
            want_cand = (size_t)n_cand + (size_t)n_cand / 8 + 1024;
            continue;
        }
        c->n_cand = (size_t)n_cand;
        // 3. exact verify
        if ((rc = ensure_hit_capacity(c, std::max<size_t>(c->hit_capacity, (size_t)n_cand)))) return rc;
        if (n_cand) {
            hipLaunchKernelGGL(verify_kernel, dim3((unsigned)((n_cand + 255) / 256)), dim3(256), 0, c->stream, c->d_cand, n_cand,
                               c->d_pages, (uint32_t)c->pitch, (uint32_t)c->rows_alloc, (uint32_t)c->n_templates, c->d_order_of,
                               c->d_tconst, c->d_needles, c->d_needle_off, thr_d, c->d_hit_keys, c->d_hit_sims,
                               (unsigned long long *)c->d_counter, (unsigned long long)c->hit_capacity);
            FOCR_HIP(c, hipGetLastError());
        }
        FOCR_HIP(c, hipEventRecord(c->ev[3], c->stream));
        unsigned long long n_hits = 0;
        FOCR_HIP(c, hipMemcpyAsync(&n_hits, c->d_counter, 8, hipMemcpyDeviceToHost, c->stream));
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        c->n_hits_raw = (size_t)n_hits;  // <= n_cand <= hit_capacity
        FOCR_HIP(c, hipEventElapsedTime(&c->ms[0], c->ev[0], c->ev[1]));
        FOCR_HIP(c, hipEventElapsedTime(&c->ms[1], c->ev[1], c->ev[2]));
        FOCR_HIP(c, hipEventElapsedTime(&c->ms[2], c->ev[2], c->ev[3]));
        c->counters[0] = n_cand;
        c->counters[1] = n_hits;
        c->launches_collect();
        return FOCR_OK;
    }
    return fail(c, FOCR_ERR_OVERFLOW, "scan_mfma: candidate buffer kept overflowing");
}

}  // namespace focr
