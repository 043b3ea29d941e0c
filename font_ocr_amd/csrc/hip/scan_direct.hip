// scan_direct.hip — exact evaluation of every (window, template) pair with v_dot4_u32_u8.
//
// This is the straightforward device formulation of the reference's hot loop
// (src/ncc.cpp:302-392): one lane per window, the window's bytes held in
// registers, every template of the size class streamed past it as scalar
// operands, integer dot product by v_dot4_u32_u8, the f64 epilogue of
// common.h, hits appended to an unordered list (ordering and the 1024 cap are
// restored by order.hip).  Window statistics (patch_sum / patch_rnorm,
// src/ncc.rs:306-312) are computed from the same registers, so this path needs
// no summed-area tables at all.  It is the cross-check for the MFMA prefilter
// path and the fallback for shapes that path does not cover.
#include "common.h"

namespace focr {

constexpr int DTX = 64;  // windows per tile row (one wave)
constexpr int DTY = 4;   // tile rows (waves per block)
constexpr int DLDW = 26; // dwords per LDS tile row: covers byte columns [0, 104) = 64 windows + 32 px + alignment

template <int NDW, int MAXH>
__global__ __launch_bounds__(256) void scan_direct_kernel(const uint8_t *__restrict__ pages, uint32_t pitch,
                                                          uint32_t rows_alloc, uint32_t r_w, uint32_t r_h, uint32_t n_w,
                                                          uint32_t n_h, const uint32_t *__restrict__ bank,
                                                          const TemplateConst *__restrict__ tc, uint32_t n_class,
                                                          KeyFmt fmt, double thr_d, uint64_t *__restrict__ hit_keys,
                                                          float *__restrict__ hit_sims, unsigned long long *__restrict__ counter,
                                                          unsigned long long capacity, uint32_t page_base, int rust) {
    constexpr int LROWS = DTY + MAXH - 1;
    __shared__ uint32_t tile[LROWS][DLDW];
    const uint32_t page = page_base + blockIdx.z;
    const uint32_t x0 = blockIdx.x * DTX;      // byte column of LDS column 0 (dword aligned)
    const uint32_t y0 = 1 + blockIdx.y * DTY;  // y = 0 is never searched (src/ncc.cpp:302)
    const uint8_t *pg = pages + (size_t)page * rows_alloc * pitch;
    for (uint32_t i = threadIdx.x; i < LROWS * DLDW; i += 256) {
        uint32_t r = i / DLDW, cdw = i % DLDW;
        uint32_t gy = y0 + r, gx = x0 + cdw * 4;
        uint32_t v = 0;
        if (gy < rows_alloc && gx < pitch) v = *reinterpret_cast<const uint32_t *>(pg + (size_t)gy * pitch + gx);
        tile[r][cdw] = v;
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const uint32_t x = x0 + 1 + lane, y = y0 + wy;  // x = 0 is never searched (src/ncc.rs:281)
    const bool valid = (x + n_w <= r_w) && (y + n_h <= r_h);
    const uint32_t col = 1 + lane, cb = col >> 2, sh = col & 3;

    uint32_t win[MAXH][NDW];
    uint32_t s_p = 0, s2_p = 0;
#pragma unroll
    for (int j = 0; j < MAXH; j++) {
#pragma unroll
        for (int k = 0; k < NDW; k++) {
            uint32_t lo = tile[wy + j][cb + k], hi = tile[wy + j][cb + k + 1];
            uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, sh);
            // keep only the window's own n_w x n_h bytes
            uint32_t keep = n_w >= (uint32_t)(4 * k + 4) ? 0xffffffffu
                            : (n_w <= (uint32_t)(4 * k) ? 0u : ((1u << (8 * (n_w - 4 * k))) - 1u));
            w = ((uint32_t)j < n_h) ? (w & keep) : 0u;
            win[j][k] = w;
            s_p = __builtin_amdgcn_udot4(w, 0x01010101u, s_p, false);
            s2_p = __builtin_amdgcn_udot4(w, w, s2_p, false);
        }
    }
    const double rnorm_p = window_rnorm(s_p, (uint64_t)s2_p, (double)(n_w * n_h));

    for (uint32_t t = 0; t < n_class; t++) {
        const uint32_t *tp = bank + (size_t)t * (MAXH * NDW);
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < MAXH; j++)
#pragma unroll
            for (int k = 0; k < NDW; k++) acc = __builtin_amdgcn_udot4(win[j][k], tp[j * NDW + k], acc, false);
        const double s_n = tc[t].s_n, n_recip = tc[t].n_recip, rnorm_n = tc[t].rnorm_n;
        double sim;
        bool emits;
        if (rust) {
            emits = rust_similarity(acc, s_p, (uint64_t)s2_p, s_n, tc[t].norm2_n, (double)(n_w * n_h), thr_d, &sim);
        } else {
            sim = ncc_similarity(acc, s_p, s_n, n_recip, rnorm_n, rnorm_p);
            emits = ncc_emits(sim, thr_d);
        }
        if (valid && emits) {
            unsigned long long idx = atomicAdd(counter, 1ull);
            if (idx < capacity) {
                hit_keys[idx] = fmt.pack(page, y, x, tc[t].index);
                hit_sims[idx] = (float)sim;
            }
        }
    }
}

// Tall (32 < n_h <= 255) or wide (16 < n_w <= 32, an extension: the reference panics, src/ncc.rs:392) templates:
// the window no longer fits in registers, so its rows are re-read from the
// LDS tile and TC templates are accumulated per pass over the rows (their rows arrive as scalar operands).
// Same arithmetic, same emission rule.  With sims == nullptr only the key is appended (MFMA mode: the hit joins
// the candidate list and verify_kernel recomputes its similarity with the other candidates).
constexpr int TALL_TC = 8;

template <int NDW>
__global__ __launch_bounds__(256) void scan_tall_kernel(const uint8_t *__restrict__ pages, uint32_t pitch, uint32_t rows_alloc,
                                                        uint32_t r_w, uint32_t r_h, uint32_t n_w, uint32_t n_h,
                                                        const uint32_t *__restrict__ bank, const TemplateConst *__restrict__ tc,
                                                        uint32_t n_class, KeyFmt fmt, double thr_d, uint64_t *__restrict__ hit_keys,
                                                        float *__restrict__ hit_sims, unsigned long long *__restrict__ counter,
                                                        unsigned long long capacity, uint32_t page_base, int rust) {
    extern __shared__ uint32_t tall_tile[];  // [DTY + n_h - 1][DLDW]
    const uint32_t lrows = DTY + n_h - 1;
    const uint32_t page = page_base + blockIdx.z;
    const uint32_t x0 = blockIdx.x * DTX, y0 = 1 + blockIdx.y * DTY;
    const uint8_t *pg = pages + (size_t)page * rows_alloc * pitch;
    for (uint32_t i = threadIdx.x; i < lrows * DLDW; i += 256) {
        uint32_t r = i / DLDW, cdw = i % DLDW;
        uint32_t gy = y0 + r, gx = x0 + cdw * 4;
        uint32_t v = 0;
        if (gy < rows_alloc && gx < pitch) v = *reinterpret_cast<const uint32_t *>(pg + (size_t)gy * pitch + gx);
        tall_tile[i] = v;
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const uint32_t x = x0 + 1 + lane, y = y0 + wy;
    const bool valid = (x + n_w <= r_w) && (y + n_h <= r_h);
    const uint32_t col = 1 + lane, cb = col >> 2, sh = col & 3;
    uint32_t keep[NDW];
#pragma unroll
    for (int k = 0; k < NDW; k++)
        keep[k] = n_w >= (uint32_t)(4 * k + 4) ? 0xffffffffu : (n_w <= (uint32_t)(4 * k) ? 0u : ((1u << (8 * (n_w - 4 * k))) - 1u));
    auto window_row = [&](uint32_t j, uint32_t (&w)[NDW]) {
        const uint32_t *row = tall_tile + (wy + j) * DLDW + cb;
#pragma unroll
        for (int k = 0; k < NDW; k++) w[k] = __builtin_amdgcn_alignbyte(row[k + 1], row[k], sh) & keep[k];
    };

    uint32_t s_p = 0, s2_p = 0;
    for (uint32_t j = 0; j < n_h; j++) {
        uint32_t w[NDW];
        window_row(j, w);
#pragma unroll
        for (int k = 0; k < NDW; k++) {
            s_p = __builtin_amdgcn_udot4(w[k], 0x01010101u, s_p, false);
            s2_p = __builtin_amdgcn_udot4(w[k], w[k], s2_p, false);
        }
    }
    const double rnorm_p = window_rnorm(s_p, (uint64_t)s2_p, (double)(n_w * n_h));

    const uint32_t stride = n_h * NDW;  // dwords per template
    for (uint32_t t0 = 0; t0 < n_class; t0 += TALL_TC) {
        uint32_t acc[TALL_TC];
        const uint32_t *tp[TALL_TC];
#pragma unroll
        for (int u = 0; u < TALL_TC; u++) {
            acc[u] = 0;
            tp[u] = bank + (size_t)min(t0 + u, n_class - 1) * stride;  // the tail repeats the last template (ignored below)
        }
        for (uint32_t j = 0; j < n_h; j++) {
            uint32_t w[NDW];
            window_row(j, w);
#pragma unroll
            for (int u = 0; u < TALL_TC; u++)
#pragma unroll
                for (int k = 0; k < NDW; k++) acc[u] = __builtin_amdgcn_udot4(w[k], tp[u][j * NDW + k], acc[u], false);
        }
#pragma unroll
        for (int u = 0; u < TALL_TC; u++) {
            const uint32_t t = t0 + u;
            if (t >= n_class) break;
            double sim;
            bool emits;
            if (rust) {
                emits = rust_similarity(acc[u], s_p, (uint64_t)s2_p, tc[t].s_n, tc[t].norm2_n, (double)(n_w * n_h), thr_d, &sim);
            } else {
                sim = ncc_similarity(acc[u], s_p, tc[t].s_n, tc[t].n_recip, tc[t].rnorm_n, rnorm_p);
                emits = ncc_emits(sim, thr_d);
            }
            if (valid && emits) {
                unsigned long long idx = atomicAdd(counter, 1ull);
                if (idx < capacity) {
                    hit_keys[idx] = fmt.pack(page, y, x, tc[t].index);
                    if (hit_sims) hit_sims[idx] = (float)sim;
                }
            }
        }
    }
}

template <int NDW>
static void launch_tall_one(focr_ctx *c, const SizeClass &sc, size_t k, double thr_d, uint64_t *keys, float *sims,
                            unsigned long long *counter, unsigned long long capacity, int rust) {
    dim3 grid((unsigned)((c->r_w - sc.n_w + DTX - 1) / DTX), (unsigned)((c->r_h - sc.n_h + DTY - 1) / DTY), (unsigned)c->sub_np);
    const uint64_t win = (uint64_t)(c->r_w - sc.n_w) * (c->r_h - sc.n_h) * c->sub_np;
    char name[64];
    snprintf(name, sizeof name, "scan_tall_kernel<%d>", NDW);
    c->launch_begin(name, sc.n_templates, win * sc.n_w * sc.n_h * sc.n_templates, win * NDW * 4 * sc.n_h * sc.n_templates);
    const size_t lds = (size_t)(DTY + sc.n_h - 1) * DLDW * 4;
    hipLaunchKernelGGL((scan_tall_kernel<NDW>), grid, dim3(256), lds, c->stream, c->d_pages, (uint32_t)c->pitch,
                       (uint32_t)c->rows_alloc, (uint32_t)c->r_w, (uint32_t)c->r_h, sc.n_w, sc.n_h,
                       c->d_direct_bank + c->direct_bank_off[k], c->d_tconst + sc.first, sc.n_templates, c->fmt, thr_d, keys, sims,
                       counter, capacity, (uint32_t)c->sub_p0, rust);
    c->launch_end();
}

// Scan one tall class (both modes call this).  sims == nullptr: keys only.
int launch_scan_tall(focr_ctx *c, size_t k, double thr_d, uint64_t *keys, float *sims, unsigned long long *counter,
                     unsigned long long capacity, int rust) {
    const SizeClass &sc = c->classes[k];
    switch (sc.ndw) {
        case 1: launch_tall_one<1>(c, sc, k, thr_d, keys, sims, counter, capacity, rust); break;
        case 2: launch_tall_one<2>(c, sc, k, thr_d, keys, sims, counter, capacity, rust); break;
        case 3: launch_tall_one<3>(c, sc, k, thr_d, keys, sims, counter, capacity, rust); break;
        case 4: launch_tall_one<4>(c, sc, k, thr_d, keys, sims, counter, capacity, rust); break;
        case 5: launch_tall_one<5>(c, sc, k, thr_d, keys, sims, counter, capacity, rust); break;
        case 6: launch_tall_one<6>(c, sc, k, thr_d, keys, sims, counter, capacity, rust); break;
        case 7: launch_tall_one<7>(c, sc, k, thr_d, keys, sims, counter, capacity, rust); break;
        case 8: launch_tall_one<8>(c, sc, k, thr_d, keys, sims, counter, capacity, rust); break;
        default: return fail(c, FOCR_ERR_INVALID, "scan_tall: unsupported size class");
    }
    FOCR_HIP(c, hipGetLastError());
    c->counters[3] += (uint64_t)(c->r_w - sc.n_w) * (c->r_h - sc.n_h) * sc.ndw * 4 * sc.n_h * sc.n_templates * c->sub_np;
    return FOCR_OK;
}

template <int NDW, int MAXH>
static void launch_one(focr_ctx *c, const SizeClass &sc, size_t k, double thr_d, int rust) {
    dim3 grid((unsigned)((c->r_w - sc.n_w + DTX - 1) / DTX), (unsigned)((c->r_h - sc.n_h + DTY - 1) / DTY),
              (unsigned)c->sub_np);
    const uint64_t win = (uint64_t)(c->r_w - sc.n_w) * (c->r_h - sc.n_h) * c->sub_np;
    char name[64];
    snprintf(name, sizeof name, "scan_direct_kernel<%d,%d>", NDW, MAXH);
    c->launch_begin(name, sc.n_templates, win * sc.n_w * sc.n_h * sc.n_templates, win * NDW * 4 * MAXH * sc.n_templates);
    hipLaunchKernelGGL((scan_direct_kernel<NDW, MAXH>), grid, dim3(256), 0, c->stream, c->d_pages, (uint32_t)c->pitch,
                       (uint32_t)c->rows_alloc, (uint32_t)c->r_w, (uint32_t)c->r_h, sc.n_w, sc.n_h,
                       c->d_direct_bank + c->direct_bank_off[k], c->d_tconst + sc.first, sc.n_templates,
                       c->fmt, thr_d, c->d_hit_keys, c->d_hit_sims, (unsigned long long *)c->d_counter,
                       (unsigned long long)c->hit_capacity, (uint32_t)c->sub_p0, rust);
    c->launch_end();
}

int ensure_hit_capacity(focr_ctx *c, size_t want) {
    if (c->hit_capacity >= want) return FOCR_OK;
    if (want > ((size_t)1 << 33)) return fail(c, FOCR_ERR_OVERFLOW, "more than 2^33 raw hits in one batch; scan fewer pages per call");
    FOCR_HIP(c, hipStreamSynchronize(c->stream));
    for (void *p : {(void *)c->d_hit_keys, (void *)c->d_hit_keys_alt, (void *)c->d_hit_sims, (void *)c->d_hit_sims_alt})
        if (p) (void)hipFree(p);
    c->d_hit_keys = c->d_hit_keys_alt = nullptr;
    c->d_hit_sims = c->d_hit_sims_alt = nullptr;
    c->hit_capacity = 0;
    if (hipMalloc(&c->d_hit_keys, want * 8) != hipSuccess || hipMalloc(&c->d_hit_keys_alt, want * 8) != hipSuccess ||
        hipMalloc(&c->d_hit_sims, want * 4) != hipSuccess || hipMalloc(&c->d_hit_sims_alt, want * 4) != hipSuccess)
        return fail(c, FOCR_ERR_NOMEM, "hit buffers: hipMalloc failed");
    c->hit_capacity = want;
    return FOCR_OK;
}

int launch_scan_direct(focr_ctx *c, float threshold, int rust) {
    const double thr_d = (double)threshold;  // src/ncc.cpp:83, 288
    int rc = ensure_hit_capacity(c, std::max<size_t>(c->hit_capacity, std::max<size_t>(1u << 20, c->sub_np * 65536)));
    if (rc) return rc;
    for (int attempt = 0; attempt < 3; attempt++) {
        c->launches_reset();
        FOCR_HIP(c, hipMemsetAsync(c->d_counter, 0, 64 * sizeof(uint32_t), c->stream));
        FOCR_HIP(c, hipEventRecord(c->ev[0], c->stream));
        FOCR_HIP(c, hipEventRecord(c->ev[1], c->stream));
        for (size_t k = 0; k < c->classes.size(); k++) {
            const SizeClass &sc = c->classes[k];
            if (sc.n_w >= c->r_w || sc.n_h >= c->r_h) continue;  // no window with x,y >= 1 fits
            if (sc.tall) {
                if ((rc = launch_scan_tall(c, k, thr_d, c->d_hit_keys, c->d_hit_sims, (unsigned long long *)c->d_counter,
                                           (unsigned long long)c->hit_capacity, rust)))
                    return rc;
                continue;
            }
            switch (sc.ndw * 100 + sc.maxh) {
                case 116: launch_one<1, 16>(c, sc, k, thr_d, rust); break;
                case 216: launch_one<2, 16>(c, sc, k, thr_d, rust); break;
                case 316: launch_one<3, 16>(c, sc, k, thr_d, rust); break;
                case 416: launch_one<4, 16>(c, sc, k, thr_d, rust); break;
                case 132: launch_one<1, 32>(c, sc, k, thr_d, rust); break;
                case 232: launch_one<2, 32>(c, sc, k, thr_d, rust); break;
                case 332: launch_one<3, 32>(c, sc, k, thr_d, rust); break;
                case 432: launch_one<4, 32>(c, sc, k, thr_d, rust); break;
                default: return fail(c, FOCR_ERR_INVALID, "scan_direct: unsupported size class");
            }
            FOCR_HIP(c, hipGetLastError());
            c->counters[3] += (uint64_t)(c->r_w - sc.n_w) * (c->r_h - sc.n_h) * sc.ndw * 4 * sc.maxh * sc.n_templates * c->sub_np;
        }
        FOCR_HIP(c, hipEventRecord(c->ev[2], c->stream));
        FOCR_HIP(c, hipEventRecord(c->ev[3], c->stream));
        unsigned long long n = 0;
        FOCR_HIP(c, hipMemcpyAsync(&n, c->d_counter, 8, hipMemcpyDeviceToHost, c->stream));
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        c->n_hits_raw = n;
        c->n_cand = n;
        if (n <= c->hit_capacity) {
            FOCR_HIP(c, hipEventElapsedTime(&c->ms[1], c->ev[1], c->ev[2]));
            c->counters[0] = n;
            c->counters[1] = n;
            c->launches_collect();
            return FOCR_OK;
        }
        c->counters[3] = 0;
        rc = ensure_hit_capacity(c, (size_t)n + (size_t)n / 8 + 1024);  // grow and rescan
        if (rc) return rc;
    }
    return fail(c, FOCR_ERR_OVERFLOW, "scan_direct: hit buffer kept overflowing");
}

}  // namespace focr
