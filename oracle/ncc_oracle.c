/*
 * ncc_oracle.c — CPU restatement of the reference NCC template-matching path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP
 * product path in font_ocr_amd/csrc.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product never does.
 *
 * Parity pinning: the reference (aconz2/font-ocr) ships no tests, fixtures or
 * golden vectors.  The kernel half of this restatement (oracle_ncc_u8) is
 * pinned against the reference kernel itself, compiled unmodified from
 * /root/reference/src/ncc.cpp into oracle/_ref/libncc_ref.so (see
 * oracle/Makefile, oracle/gen_golden.py, tests/test_oracle.py and the golden
 * vectors under tests/golden/).  The Rust half (summed-area tables,
 * prepare_for_size, process_hits) cannot be built here (no rustc); it is
 * restated from the text of /root/reference/src/ncc.rs and is pinned only by
 * that text — "parity unpinned" for those functions beyond the invariants the
 * tests check (window sums against brute force, ordering properties).
 *
 * Every function cites the reference file:line it follows.  Plain C11, scalar,
 * compiled with -ffp-contract=off so that the only fused operation is the
 * explicit fma() the reference's epilogue uses.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint16_t x, y;
    float similarity;
} OracleMatch; /* src/ncc.cpp:7-10, src/ncc.rs:66-72 */

typedef struct {
    int32_t x, y, w, h; /* rect origin + size, src/ncc.rs:74-79 */
    float similarity;
    uint32_t letter; /* code point (or any caller id) */
} OracleHit;

/* ---- summed-area tables: src/ncc.rs:937-974 ---------------------------- */

/* ncc_sum_table, src/ncc.rs:938-955 (u32, wrapping like Rust release). */
void oracle_sum_table(const uint8_t *px, size_t r_w, size_t r_h, uint32_t *out) {
    out[0] = px[0];
    for (size_t x = 1; x < r_w; x++) out[x] = (uint32_t)px[x] + out[x - 1];
    for (size_t y = 1; y < r_h; y++) out[y * r_w] = (uint32_t)px[y * r_w] + out[(y - 1) * r_w];
    for (size_t y = 1; y < r_h; y++)
        for (size_t x = 1; x < r_w; x++)
            out[y * r_w + x] = (uint32_t)px[y * r_w + x] + out[y * r_w + x - 1] +
                               out[(y - 1) * r_w + x] - out[(y - 1) * r_w + x - 1];
}

/* ncc_sumsqr_table, src/ncc.rs:957-974.  First row/column hold p*p only (not
 * cumulative) exactly as the reference writes them; queries at x,y >= 1
 * telescope to the exact window sum regardless. */
void oracle_sumsqr_table(const uint8_t *px, size_t r_w, size_t r_h, uint64_t *out) {
    for (size_t x = 0; x < r_w; x++) {
        uint64_t p = px[x];
        out[x] = p * p;
    }
    for (size_t y = 0; y < r_h; y++) {
        uint64_t p = px[y * r_w];
        out[y * r_w] = p * p;
    }
    for (size_t y = 1; y < r_h; y++)
        for (size_t x = 1; x < r_w; x++) {
            uint64_t p = px[y * r_w + x];
            out[y * r_w + x] = p * p + out[y * r_w + x - 1] + out[(y - 1) * r_w + x] -
                               out[(y - 1) * r_w + x - 1];
        }
}

/* ncc_sum_table_sum_nz, src/ncc.rs:976-983 */
static inline uint32_t sum_nz(const uint32_t *s, size_t r_w, size_t x, size_t y, size_t w, size_t h) {
    int64_t a = s[(y + h - 1) * r_w + (x + w - 1)];
    int64_t b = s[(y + h - 1) * r_w + (x - 1)];
    int64_t c = s[(y - 1) * r_w + (x + w - 1)];
    int64_t d = s[(y - 1) * r_w + (x - 1)];
    return (uint32_t)(a - b + d - c);
}

/* ncc_sumsqr_table_sum_nz, src/ncc.rs:1006-1013 */
static inline uint64_t sumsqr_nz(const uint64_t *s, size_t r_w, size_t x, size_t y, size_t w, size_t h) {
    int64_t a = (int64_t)s[(y + h - 1) * r_w + (x + w - 1)];
    int64_t b = (int64_t)s[(y + h - 1) * r_w + (x - 1)];
    int64_t c = (int64_t)s[(y - 1) * r_w + (x + w - 1)];
    int64_t d = (int64_t)s[(y - 1) * r_w + (x - 1)];
    return (uint64_t)(a - b + d - c);
}

/* Searcher::prepare_for_size, src/ncc.rs:263-318.
 * patch_sum/patch_rnorm are [r_h][r_w]; start_end is [2*r_h].  Entries outside
 * [start,end) of a row are left untouched (stale), as in the reference. */
void oracle_prepare_for_size(const uint32_t *sum_table, const uint64_t *sumsqr_table, size_t r_w,
                             size_t r_h, size_t n_w, size_t n_h, uint32_t *patch_sum,
                             double *patch_rnorm, uint16_t *start_end) {
    size_t n = n_h * n_w;
    if (r_w < n_w || r_h < n_h) return;
    size_t x_searches = r_w - n_w + 1;
    size_t y_searches = r_h - n_h + 1;
    for (size_t y = 1; y < y_searches; y++) {
        size_t start = 1;
        while (start < x_searches) { /* src/ncc.rs:280-290 */
            if (sum_nz(sum_table, r_w, start, y, n_w, n_h) != 0) break;
            start++;
        }
        size_t end = x_searches - 1; /* src/ncc.rs:291-301 */
        while (end > start) {
            if (sum_nz(sum_table, r_w, end, y, n_w, n_h) != 0) break;
            end--;
        }
        end = end + 1;
        /* NB (reference quirk kept): when x_searches == 1 the reference computes
         * start = 1, end = 0 + 1 = 1; when the row is blank start = x_searches and
         * end = x_searches - 1 + 1 ... the `while x > start` loop does not run when
         * x_searches - 1 <= start, so end = x_searches. */
        for (size_t x = start; x < end; x++) { /* src/ncc.rs:306-312 */
            uint32_t s_p = sum_nz(sum_table, r_w, x, y, n_w, n_h);
            uint64_t s2_p = sumsqr_nz(sumsqr_table, r_w, x, y, n_w, n_h);
            double norm = (double)s2_p - ((double)((uint64_t)s_p * (uint64_t)s_p)) / (double)n;
            patch_sum[y * r_w + x] = s_p;
            patch_rnorm[y * r_w + x] = 1. / sqrt(norm);
        }
        start_end[y * 2 + 0] = (uint16_t)start;
        start_end[y * 2 + 1] = (uint16_t)end;
    }
}

/* copy_needle_n_u8, src/ncc.rs:925-935: dense n_w x n_h -> N-wide zero-padded rows */
void oracle_copy_needle(const uint8_t *needle, size_t n_w, size_t n_h, size_t N, uint8_t *out) {
    for (size_t y = 0; y < n_h; y++) {
        for (size_t x = 0; x < n_w; x++) out[y * N + x] = needle[y * n_w + x];
        for (size_t x = n_w; x < N; x++) out[y * N + x] = 0;
    }
}

/* ncc_8_u8 / ncc_16_u8, src/ncc.cpp:48-251 / 253-396, restated as one scalar
 * routine over an N-wide (N = 8 or 16) zero-padded needle.  N = 32 is this
 * build's extension for templates 17..32 px wide (the reference panics there,
 * src/ncc.rs:392): the same arithmetic, no
 * reference behaviour to pin it against.
 *
 *   acc  = sum_{j<n_h} sum_{i<N} T[j][i] * R[y+j][x+i]        (ncc.cpp:316-321)
 *   num  = fma(-((double)s_n * (double)s_p), 1/n, (double)acc) (ncc.cpp:212, 358)
 *   sim  = num * (rnorm_n * rnorm_p)                           (ncc.cpp:214-215, 360-361)
 *   emit iff sim > (double)thr && sim != +inf, in (y, x) order, stop at n_out.
 *
 * The int->double conversions follow the vector path (_mm256_cvtepi32_pd:
 * signed), which is what every window but a row's last (end-start)%4 (or %16)
 * takes; the scalar tails convert unsigned — identical below 2^31, and GCC
 * contracts the tail's `a - b*c` into the same vfnmadd (checked by objdump of
 * oracle/_ref/libncc_ref.so).  The padded columns i >= n_w multiply zeros; like
 * the reference this routine still reads the N image bytes of every row
 * (fixed-width inner loop, so the compiler vectorises it), i.e. it over-reads
 * up to N-n_w bytes past the last image row: callers pad the page buffer.
 */
static inline uint32_t dot_rows_8(const uint8_t *r, size_t r_w, const uint8_t *t, size_t n_h) {
    uint32_t acc = 0;
    for (size_t j = 0; j < n_h; j++)
        for (size_t i = 0; i < 8; i++) acc += (uint32_t)t[j * 8 + i] * (uint32_t)r[j * r_w + i];
    return acc;
}
static inline uint32_t dot_rows_16(const uint8_t *r, size_t r_w, const uint8_t *t, size_t n_h) {
    uint32_t acc = 0;
    for (size_t j = 0; j < n_h; j++)
        for (size_t i = 0; i < 16; i++) acc += (uint32_t)t[j * 16 + i] * (uint32_t)r[j * r_w + i];
    return acc;
}

static inline uint32_t dot_rows_32(const uint8_t *r, size_t r_w, const uint8_t *t, size_t n_h) {
    uint32_t acc = 0;
    for (size_t j = 0; j < n_h; j++)
        for (size_t i = 0; i < 32; i++) acc += (uint32_t)t[j * 32 + i] * (uint32_t)r[j * r_w + i];
    return acc;
}

size_t oracle_ncc_u8(const uint8_t *reference, size_t r_w, size_t r_h, const uint8_t *needle_N,
                     size_t N, size_t n_w, size_t n_h, const uint32_t *patch_sum,
                     const double *patch_rnorm, const uint16_t *start_end, float threshold,
                     OracleMatch *out, size_t n_out) {
    size_t n = n_w * n_h;
    if (r_h < n_h || n_out == 0 || (N != 8 && N != 16 && N != 32)) return 0;
    size_t y_searches = r_h - n_h + 1;

    uint32_t s_n = 0, s2_n = 0; /* ncc.cpp:73-81, 278-286 */
    for (size_t i = 0; i < n_h; i++)
        for (size_t j = 0; j < N; j++) {
            s_n += needle_N[i * N + j];
            s2_n += (uint32_t)needle_N[i * N + j] * (uint32_t)needle_N[i * N + j];
        }

    double threshold_d = threshold;
    double norm2_n = (double)s2_n - (double)((uint64_t)s_n * (uint64_t)s_n) / (double)n; /* :84, :289 */
    double rnorm_n = 1. / sqrt(norm2_n);
    double n_recip = 1. / (double)n;
    double s_n_d = (double)s_n;

    size_t cnt = 0;
    for (size_t y = 1; y < y_searches; y++) {
        size_t start = start_end[y * 2 + 0], end = start_end[y * 2 + 1];
        for (size_t x = start; x < end; x++) {
            const uint8_t *r0 = reference + y * r_w + x;
            uint32_t acc = N == 8 ? dot_rows_8(r0, r_w, needle_N, n_h)
                           : N == 16 ? dot_rows_16(r0, r_w, needle_N, n_h) : dot_rows_32(r0, r_w, needle_N, n_h);
            double acc_d = (double)(int32_t)acc;
            double s_p_d = (double)(int32_t)patch_sum[y * r_w + x];
            double num = fma(-(s_n_d * s_p_d), n_recip, acc_d);
            double den = rnorm_n * patch_rnorm[y * r_w + x];
            double sim = num * den;
            if (sim > threshold_d && !(sim == INFINITY)) {
                out[cnt].x = (uint16_t)x;
                out[cnt].y = (uint16_t)y;
                out[cnt].similarity = (float)sim;
                cnt++;
                if (cnt == n_out) return n_out; /* ncc.cpp:225-227, 371-373 */
            }
        }
    }
    return cnt;
}

/* Searcher::search_n_u8, src/ncc.rs:406-483 — the scalar scan behind `ncc --rust`.
 * Same windows and the same integer dot product as the AVX2 path, but
 *   num = acc - (s_n * s_p) / n            (a division, :450; windows with s_p == 0 or num < 0 are skipped, :447-453)
 *   sim = num / sqrt(norm2_n * norm2_p)    (:455-460)
 *   emit iff sim != +inf && sim > thr      (:466), similarity narrowed to f32 (:461)
 * a needle with s_n == 0 yields nothing (:431-433) and there is no cap on the number of matches (a Vec).
 * `cap` here only bounds the output buffer; returns the number of matches found (may exceed cap: the caller sized
 * the buffer too small).  Parity-unpinned: no Rust toolchain here, restated from the text. */
size_t oracle_search_rust_u8(const uint8_t *reference, size_t r_w, size_t r_h, const uint8_t *needle,
                             size_t n_w, size_t n_h, const uint32_t *sum_table, const uint64_t *sumsqr_table,
                             const uint16_t *start_end, float threshold, OracleMatch *out, size_t cap) {
    size_t n = n_w * n_h, cnt = 0;
    if (r_h < n_h || r_w < n_w) return 0;
    size_t y_searches = r_h - n_h + 1;
    uint32_t s_n = 0, s2_n = 0; /* image_sum_sumsqr over the dense needle */
    for (size_t i = 0; i < n; i++) {
        s_n += needle[i];
        s2_n += (uint32_t)needle[i] * (uint32_t)needle[i];
    }
    if (s_n == 0) return 0;
    for (size_t y = 1; y < y_searches; y++) {
        size_t start = start_end[y * 2 + 0], end = start_end[y * 2 + 1];
        for (size_t x = start; x < end; x++) {
            uint32_t acc = 0;
            for (size_t j = 0; j < n_h; j++)
                for (size_t i = 0; i < n_w; i++)
                    acc += (uint32_t)needle[j * n_w + i] * (uint32_t)reference[(y + j) * r_w + x + i];
            uint32_t s_p = sum_nz(sum_table, r_w, x, y, n_w, n_h);
            uint64_t s2_p = sumsqr_nz(sumsqr_table, r_w, x, y, n_w, n_h);
            if (s_p == 0) continue;
            double num = (double)acc - (double)((uint64_t)s_n * (uint64_t)s_p) / (double)n;
            if (num < 0.) continue;
            double norm2_n = (double)s2_n - (double)((uint64_t)s_n * (uint64_t)s_n) / (double)n;
            double norm2_p = (double)s2_p - (double)((uint64_t)s_p * (uint64_t)s_p) / (double)n;
            double den = sqrt(norm2_n * norm2_p);
            double sim = num / den;
            if (sim != INFINITY && sim > (double)threshold) {
                if (cnt < cap) {
                    out[cnt].x = (uint16_t)x;
                    out[cnt].y = (uint16_t)y;
                    out[cnt].similarity = (float)sim;
                }
                cnt++;
            }
        }
    }
    return cnt;
}

/* ---- process_hits: src/ncc.rs:723-786 + partition_by 1036-1052 --------- */

static inline int32_t f32_total_key(float f) { /* f32::total_cmp ordering key */
    int32_t b;
    memcpy(&b, &f, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}

/* stable merge sort on a key extracted from OracleHit (Rust sort_by_key is stable) */
static void stable_sort_hits(OracleHit *a, size_t n, int by_x) {
    if (n < 2) return;
    OracleHit *tmp = (OracleHit *)malloc(n * sizeof(OracleHit));
    for (size_t width = 1; width < n; width *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * width) {
            size_t mid = lo + width < n ? lo + width : n;
            size_t hi = lo + 2 * width < n ? lo + 2 * width : n;
            size_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                int32_t ki = by_x ? a[i].x : a[i].y, kj = by_x ? a[j].x : a[j].y;
                if (kj < ki) tmp[k++] = a[j++];
                else tmp[k++] = a[i++];
            }
            while (i < mid) tmp[k++] = a[i++];
            while (j < hi) tmp[k++] = a[j++];
        }
        memcpy(a, tmp, n * sizeof(OracleHit));
    }
    free(tmp);
}

/* process_hits.  in: all_hits[n_hits] in get_hits order (offset-major, then
 * alphabet order, then (y,x)).  out: out_hits (capacity n_hits) holds the
 * de-duplicated characters line after line; line_ends[k] = exclusive end index
 * of line k in out_hits (capacity n_hits).  Returns the number of lines.
 * The reference panics on an empty hit list (partition_by's unwrap at
 * src/ncc.rs:1040); this restatement returns 0 lines instead. */
size_t oracle_process_hits(const OracleHit *all_hits, size_t n_hits, float anchor_threshold,
                           int32_t overlap, OracleHit *out_hits, size_t *line_ends) {
    /* (1) keep_y: src/ncc.rs:726-731 ; y < 65536 on this path (u16 wire format) */
    uint8_t *keep = (uint8_t *)calloc(65536, 1);
    OracleHit *hits = (OracleHit *)malloc((n_hits ? n_hits : 1) * sizeof(OracleHit));
    size_t m = 0;
    for (size_t i = 0; i < n_hits; i++)
        if (all_hits[i].similarity >= anchor_threshold) keep[all_hits[i].y & 0xffff] = 1;
    for (size_t i = 0; i < n_hits; i++) /* (2) src/ncc.rs:732-738 */
        if (keep[all_hits[i].y & 0xffff]) hits[m++] = all_hits[i];
    free(keep);
    if (m == 0) {
        free(hits);
        return 0;
    }
    stable_sort_hits(hits, m, 0); /* (3) src/ncc.rs:741 */

    size_t n_lines = 0, n_out = 0;
    size_t i = 0;
    while (i < m) { /* (4) partition by equal y: src/ncc.rs:747 */
        size_t j = i + 1;
        while (j < m && hits[j].y == hits[i].y) j++;
        stable_sort_hits(hits + i, j - i, 1); /* src/ncc.rs:749-752 */
        /* (5) partition_by anchored on the first element of the group:
         * src/ncc.rs:755-757 with partition_by 1042-1048 (`last` only moves when a
         * group closes). */
        size_t g = i;
        while (g < j) {
            size_t e = g + 1;
            while (e < j && abs(hits[g].x - hits[e].x) <= overlap) e++;
            /* (6) max_by total_cmp, last maximum wins: src/ncc.rs:761-764 */
            size_t best = g;
            for (size_t k = g + 1; k < e; k++)
                if (f32_total_key(hits[k].similarity) >= f32_total_key(hits[best].similarity)) best = k;
            out_hits[n_out++] = hits[best];
            g = e;
        }
        line_ends[n_lines++] = n_out;
        i = j;
    }
    free(hits);
    return n_lines;
}

/* ---- whole-page driver (get_hits without rasterisation): src/ncc.rs:576-702 */

typedef size_t (*ncc_kernel_fn)(const uint8_t *, size_t, size_t, const uint8_t *, size_t, size_t,
                                uint32_t *, size_t, const uint32_t *, const double *,
                                const uint16_t *, float, OracleMatch *, size_t);

typedef struct {
    uint32_t n_w, n_h;
    uint32_t offset; /* byte offset of the dense n_w*n_h needle in `needles` */
} OracleTemplate;

/* Scan one page with a whole bank, the way get_hits + Searcher::search_c_u8 do
 * (src/ncc.rs:332-404, 587-701): window statistics are rebuilt whenever the
 * template size changes (cache on last size, src/ncc.rs:264-268), the needle is
 * padded to N = 8 if n_w <= 8 else 16 (src/ncc.rs:337, 364), each template gets
 * at most `cap` matches.  `page` must be r_w*r_h bytes followed by >= 32 bytes
 * of readable padding when the reference kernels are used (over-read quirk).
 * k8/k16 = reference kernels from oracle/_ref (cpu_baseline kind "reference"),
 * or NULL to use oracle_ncc_u8.  counts[t] receives the per-template count;
 * matches is [n_templates][cap].  Returns total matches. */
size_t oracle_scan_page(const uint8_t *page, size_t r_w, size_t r_h, const uint8_t *needles,
                        const OracleTemplate *tmpl, size_t n_templates, float threshold, size_t cap,
                        ncc_kernel_fn k8, ncc_kernel_fn k16, uint32_t *counts,
                        OracleMatch *matches) {
    size_t npx = r_w * r_h;
    uint32_t *sum_table = (uint32_t *)malloc(npx * 4);
    uint64_t *sumsqr_table = (uint64_t *)malloc(npx * 8);
    uint32_t *patch_sum = (uint32_t *)calloc(npx, 4);
    double *patch_rnorm = (double *)calloc(npx, 8);
    uint16_t *start_end = (uint16_t *)calloc(r_h * 2, 2);
    size_t acc_len = r_w * 8 + 8; /* src/ncc.rs:242 */
    uint32_t *acc = (uint32_t *)calloc(acc_len + 8, 4);
    uint8_t needle_N[32 * 256];
    oracle_sum_table(page, r_w, r_h, sum_table);
    oracle_sumsqr_table(page, r_w, r_h, sumsqr_table);
    size_t last_w = 0, last_h = 0, total = 0;
    for (size_t t = 0; t < n_templates; t++) {
        size_t n_w = tmpl[t].n_w, n_h = tmpl[t].n_h;
        counts[t] = 0;
        if (n_w == 0 || n_h == 0 || n_w > 32 || n_h > 255 || n_w > r_w || n_h > r_h) continue;
        if (n_w != last_w || n_h != last_h) {
            oracle_prepare_for_size(sum_table, sumsqr_table, r_w, r_h, n_w, n_h, patch_sum,
                                    patch_rnorm, start_end);
            last_w = n_w;
            last_h = n_h;
        }
        size_t N = n_w <= 8 ? 8 : n_w <= 16 ? 16 : 32;
        oracle_copy_needle(needles + tmpl[t].offset, n_w, n_h, N, needle_N);
        ncc_kernel_fn k = N == 8 ? k8 : N == 16 ? k16 : NULL; /* no reference kernel for the N = 32 extension */
        size_t c;
        if (k)
            c = k(page, r_w, r_h, needle_N, n_w, n_h, acc, acc_len, patch_sum, patch_rnorm, start_end,
                  threshold, matches + t * cap, cap);
        else
            c = oracle_ncc_u8(page, r_w, r_h, needle_N, N, n_w, n_h, patch_sum, patch_rnorm,
                              start_end, threshold, matches + t * cap, cap);
        counts[t] = (uint32_t)c;
        total += c;
    }
    free(sum_table);
    free(sumsqr_table);
    free(patch_sum);
    free(patch_rnorm);
    free(start_end);
    free(acc);
    return total;
}

/* get_hits' loop with search_u8 (`--rust`, src/ncc.rs:320-330, 650-654): as oracle_scan_page, scalar arithmetic,
 * no cap semantics (counts[t] may exceed `cap`, only the first `cap` matches are stored). */
size_t oracle_scan_page_rust(const uint8_t *page, size_t r_w, size_t r_h, const uint8_t *needles,
                             const OracleTemplate *tmpl, size_t n_templates, float threshold, size_t cap,
                             uint32_t *counts, OracleMatch *matches) {
    size_t npx = r_w * r_h;
    uint32_t *sum_table = (uint32_t *)malloc(npx * 4);
    uint64_t *sumsqr_table = (uint64_t *)malloc(npx * 8);
    uint32_t *patch_sum = (uint32_t *)calloc(npx, 4);
    double *patch_rnorm = (double *)calloc(npx, 8);
    uint16_t *start_end = (uint16_t *)calloc(r_h * 2, 2);
    oracle_sum_table(page, r_w, r_h, sum_table);
    oracle_sumsqr_table(page, r_w, r_h, sumsqr_table);
    size_t last_w = 0, last_h = 0, total = 0;
    for (size_t t = 0; t < n_templates; t++) {
        size_t n_w = tmpl[t].n_w, n_h = tmpl[t].n_h;
        counts[t] = 0;
        if (n_w == 0 || n_h == 0 || n_w > 16 || n_w > r_w || n_h > r_h) continue; /* todo!() above 16, src/ncc.rs:328 */
        if (n_w != last_w || n_h != last_h) {
            oracle_prepare_for_size(sum_table, sumsqr_table, r_w, r_h, n_w, n_h, patch_sum, patch_rnorm, start_end);
            last_w = n_w;
            last_h = n_h;
        }
        size_t c = oracle_search_rust_u8(page, r_w, r_h, needles + tmpl[t].offset, n_w, n_h, sum_table, sumsqr_table,
                                         start_end, threshold, matches + t * cap, cap);
        counts[t] = (uint32_t)c;
        total += c;
    }
    free(sum_table);
    free(sumsqr_table);
    free(patch_sum);
    free(patch_rnorm);
    free(start_end);
    return total;
}

/* Page-parallel driver mirroring the reference's rayon par_iter over pages
 * (src/ncc.rs:839-847): one page per OpenMP task, `threads` workers.  pages are
 * packed with a stride of page_stride bytes (>= r_w*r_h + 32).  counts is
 * [n_pages][n_templates], matches [n_pages][n_templates][cap]; pass
 * matches == NULL to keep only the counts (timing runs).  Returns total hits. */
size_t oracle_scan_pages_mt(const uint8_t *pages, size_t page_stride, size_t n_pages, size_t r_w,
                            size_t r_h, const uint8_t *needles, const OracleTemplate *tmpl,
                            size_t n_templates, float threshold, size_t cap, ncc_kernel_fn k8,
                            ncc_kernel_fn k16, int threads, uint32_t *counts, OracleMatch *matches) {
    size_t total = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : total)
    for (long p = 0; p < (long)n_pages; p++) {
        OracleMatch *m = matches ? matches + (size_t)p * n_templates * cap
                                 : (OracleMatch *)malloc(n_templates * cap * sizeof(OracleMatch));
        total += oracle_scan_page(pages + (size_t)p * page_stride, r_w, r_h, needles, tmpl,
                                  n_templates, threshold, cap, k8, k16,
                                  counts + (size_t)p * n_templates, m);
        if (!matches) free(m);
    }
    return total;
}
