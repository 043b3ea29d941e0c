// post.hip — process_hits on the device (reference: src/ncc.rs:723-786 + partition_by 1036-1052).
//
//  (1) keep_y: rows y with any hit of similarity >= anchor_threshold (f32 >=), per page   :726-731
//  (2) keep hits on those rows, in get_hits order                                       :732-738
//  (3) stable sort by y, then each equal-y run stable-sorted by x                       :741-752
//      => total order (y, x, get_hits order); get_hits order of two hits with equal
//      (x, y) is their template index, so the sort key is (page, y, x, t).
//  (4) per line, groups anchored on the group's first element: a hit joins while
//      |x - x_first| <= overlap                                                         :755-757, 1042-1048
//  (5) per group the maximum similarity by f32::total_cmp, LAST maximum wins            :761-764
//  (6) lines ascending in y, characters ascending in x.
#include <cstring>

#include "common.h"

namespace focr {

int sort_pairs_u64_f32(focr_ctx *c, uint64_t *&keys, uint64_t *&keys_alt, float *&vals, float *&vals_alt, size_t n,
                       unsigned end_bit);
int exclusive_scan_u64(focr_ctx *c, const uint64_t *in, uint64_t *out, size_t n);

__device__ __forceinline__ int32_t total_key(float f) {  // f32::total_cmp as a signed-int order
    int32_t b = __float_as_int(f);
    return b ^ (int32_t)(((uint32_t)(b >> 31)) >> 1);
}

__global__ void mark_anchor_rows(const focr_match_t *__restrict__ m, const uint64_t *__restrict__ keys, size_t n,
                                 float anchor, uint32_t r_h, uint8_t *__restrict__ keep) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (m[i].similarity >= anchor) keep[(size_t)(keys[i] >> 48) * r_h + m[i].y] = 1;
}

__global__ void copy_sims(const focr_match_t *__restrict__ m, size_t n, float *__restrict__ sims) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sims[i] = m[i].similarity;
}

// One thread per line start walks its line (sorted by x, then t) and records, for the k-th group,
// the index of the winning element at choice[line_start + k].  packed[i] = (is kept line start) << 32 | n_groups.
__global__ void walk_lines(const uint64_t *__restrict__ keys, const float *__restrict__ sims, size_t n, uint32_t r_h,
                           int32_t overlap, const uint8_t *__restrict__ keep, uint32_t *__restrict__ choice,
                           uint64_t *__restrict__ packed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t line = keys[i] >> 32;  // (page, y)
    bool start = (i == 0) || ((keys[i - 1] >> 32) != line);
    if (!start || !keep[(size_t)(line >> 16) * r_h + (uint32_t)(line & 0xffff)]) {
        packed[i] = 0;
        return;
    }
    uint32_t groups = 0;
    size_t g = i;
    while (g < n && (keys[g] >> 32) == line) {
        const int32_t x_first = (int32_t)((keys[g] >> 16) & 0xffff);
        size_t best = g;
        int32_t best_key = total_key(sims[g]);
        size_t e = g + 1;
        while (e < n && (keys[e] >> 32) == line) {
            int32_t x = (int32_t)((keys[e] >> 16) & 0xffff);
            int32_t d = x_first - x;
            if ((d < 0 ? -d : d) > overlap) break;
            int32_t k = total_key(sims[e]);
            if (k >= best_key) {  // max_by keeps the last maximum
                best_key = k;
                best = e;
            }
            e++;
        }
        choice[i + groups] = (uint32_t)best;
        groups++;
        g = e;
    }
    packed[i] = ((uint64_t)1 << 32) | groups;
}

__global__ void emit_chars(const uint64_t *__restrict__ keys, const float *__restrict__ sims, size_t n,
                           const uint32_t *__restrict__ choice, const uint64_t *__restrict__ packed,
                           const uint64_t *__restrict__ scanned, const uint32_t *__restrict__ t_w,
                           const uint32_t *__restrict__ t_h, const uint32_t *__restrict__ t_letter,
                           uint64_t *__restrict__ line_char_off, focr_hit_t *__restrict__ chars) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t p = packed[i];
    if (!(p >> 32)) return;
    uint32_t groups = (uint32_t)p;
    uint64_t line_idx = scanned[i] >> 32, off = scanned[i] & 0xffffffffu;
    line_char_off[line_idx] = off;
    for (uint32_t k = 0; k < groups; k++) {
        uint32_t e = choice[i + k];
        uint64_t key = keys[e];
        uint32_t t = (uint32_t)(key & 0xffff);
        focr_hit_t h;
        h.x = (uint16_t)((key >> 16) & 0xffff);
        h.y = (uint16_t)((key >> 32) & 0xffff);
        h.w = (uint16_t)t_w[t];
        h.h = (uint16_t)t_h[t];
        h.similarity = sims[e];
        h.letter = t_letter[t];
        h.template_index = t;
        chars[off + k] = h;
    }
}

// page_line_off[p] = number of kept lines on pages < p
__global__ void page_offsets(const uint64_t *__restrict__ keys, size_t n, const uint64_t *__restrict__ scanned,
                             uint64_t total_lines, uint32_t n_pages, uint64_t *__restrict__ page_line_off) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > n_pages) return;
    uint64_t lo = 0, hi = n, v = (uint64_t)p << 48;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    page_line_off[p] = lo < n ? (scanned[lo] >> 32) : total_lines;
}

}  // namespace focr

using namespace focr;

extern "C" {

int focr_process_hits(focr_ctx_t *c, float anchor_threshold, int32_t overlap) {
    if (!c) return FOCR_ERR_INVALID;
    if (!c->scanned) return fail(c, FOCR_ERR_STATE, "focr_process_hits: no scan results");
    FOCR_HIP(c, hipSetDevice(c->device));
    c->processed = false;
    const size_t n = c->n_matches, n_pages = c->n_pages;
    c->h_page_line_off.assign(n_pages + 1, 0);
    c->h_line_char_off.assign(1, 0);
    c->h_chars.clear();
    c->n_chars = c->n_lines = 0;
    if (n == 0) {  // the reference panics on an empty hit list (src/ncc.rs:1040); we return zero lines
        c->processed = true;
        c->ms[4] = 0.f;
        return FOCR_OK;
    }
    FOCR_HIP(c, hipEventRecord(c->ev[5], c->stream));
    uint8_t *keep = nullptr;
    uint32_t *choice = nullptr;
    uint64_t *packed = nullptr, *scanned = nullptr, *d_line_off = nullptr, *d_page_off = nullptr;
    focr_hit_t *d_chars = nullptr;
    auto cleanup = [&]() {
        for (void *p : {(void *)keep, (void *)choice, (void *)packed, (void *)scanned, (void *)d_line_off, (void *)d_page_off,
                        (void *)d_chars})
            if (p) (void)hipFree(p);
    };
#define PH(expr)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            cleanup();                                                                        \
            return fail(c, FOCR_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                                     \
    } while (0)
    PH(hipMalloc(&keep, n_pages * c->r_h));
    PH(hipMemsetAsync(keep, 0, n_pages * c->r_h, c->stream));
    PH(hipMalloc(&choice, n * 4));
    PH(hipMalloc(&packed, n * 8));
    PH(hipMalloc(&scanned, n * 8));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(mark_anchor_rows, dim3(nb), dim3(256), 0, c->stream, c->d_matches, c->d_match_keys, n,
                       anchor_threshold, (uint32_t)c->r_h, keep);
    // sort (key = page|y|x|t, value = similarity); the hit buffers of the scan are reused as scratch
    hipLaunchKernelGGL(copy_sims, dim3(nb), dim3(256), 0, c->stream, c->d_matches, n, c->d_hit_sims);
    PH(hipMemcpyAsync(c->d_hit_keys, c->d_match_keys, n * 8, hipMemcpyDeviceToDevice, c->stream));
    int rc = sort_pairs_u64_f32(c, c->d_hit_keys, c->d_hit_keys_alt, c->d_hit_sims, c->d_hit_sims_alt, n, 64);
    if (rc) {
        cleanup();
        return rc;
    }
    hipLaunchKernelGGL(walk_lines, dim3(nb), dim3(256), 0, c->stream, c->d_hit_keys, c->d_hit_sims, n, (uint32_t)c->r_h,
                       overlap, keep, choice, packed);
    rc = exclusive_scan_u64(c, packed, scanned, n);
    if (rc) {
        cleanup();
        return rc;
    }
    uint64_t last_scan = 0, last_packed = 0;
    PH(hipMemcpyAsync(&last_scan, scanned + (n - 1), 8, hipMemcpyDeviceToHost, c->stream));
    PH(hipMemcpyAsync(&last_packed, packed + (n - 1), 8, hipMemcpyDeviceToHost, c->stream));
    PH(hipStreamSynchronize(c->stream));
    const uint64_t tot = last_scan + last_packed;
    c->n_lines = (size_t)(tot >> 32);
    c->n_chars = (size_t)(tot & 0xffffffffu);
    PH(hipMalloc(&d_line_off, (c->n_lines + 1) * 8));
    PH(hipMalloc(&d_page_off, (n_pages + 1) * 8));
    PH(hipMalloc(&d_chars, (c->n_chars ? c->n_chars : 1) * sizeof(focr_hit_t)));
    hipLaunchKernelGGL(emit_chars, dim3(nb), dim3(256), 0, c->stream, c->d_hit_keys, c->d_hit_sims, n, choice, packed,
                       scanned, c->d_t_w, c->d_t_h, c->d_t_letter, d_line_off, d_chars);
    hipLaunchKernelGGL(page_offsets, dim3((unsigned)((n_pages + 1 + 255) / 256)), dim3(256), 0, c->stream, c->d_hit_keys, n,
                       scanned, (uint64_t)c->n_lines, (uint32_t)n_pages, d_page_off);
    PH(hipGetLastError());
    c->h_line_char_off.assign(c->n_lines + 1, 0);
    c->h_chars.resize(c->n_chars);
    if (c->n_lines) PH(hipMemcpyAsync(c->h_line_char_off.data(), d_line_off, c->n_lines * 8, hipMemcpyDeviceToHost, c->stream));
    c->h_line_char_off[c->n_lines] = c->n_chars;
    PH(hipMemcpyAsync(c->h_page_line_off.data(), d_page_off, (n_pages + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    if (c->n_chars)
        PH(hipMemcpyAsync(c->h_chars.data(), d_chars, c->n_chars * sizeof(focr_hit_t), hipMemcpyDeviceToHost, c->stream));
    PH(hipEventRecord(c->ev[6], c->stream));
    PH(hipStreamSynchronize(c->stream));
    PH(hipEventElapsedTime(&c->ms[4], c->ev[5], c->ev[6]));
#undef PH
    cleanup();
    c->processed = true;
    return FOCR_OK;
}

size_t focr_total_chars(focr_ctx_t *c) { return (c && c->processed) ? c->n_chars : 0; }
size_t focr_total_lines(focr_ctx_t *c) { return (c && c->processed) ? c->n_lines : 0; }

int focr_get_lines(focr_ctx_t *c, uint64_t *page_line_offsets, uint64_t *line_char_offsets, focr_hit_t *chars) {
    if (!c) return FOCR_ERR_INVALID;
    if (!c->processed) return fail(c, FOCR_ERR_STATE, "focr_get_lines: call focr_process_hits first");
    if (page_line_offsets) memcpy(page_line_offsets, c->h_page_line_off.data(), c->h_page_line_off.size() * 8);
    if (line_char_offsets) memcpy(line_char_offsets, c->h_line_char_off.data(), c->h_line_char_off.size() * 8);
    if (chars && c->n_chars) memcpy(chars, c->h_chars.data(), c->n_chars * sizeof(focr_hit_t));
    return FOCR_OK;
}

}  // extern "C"
