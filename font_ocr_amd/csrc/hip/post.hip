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

int exclusive_scan_u64(focr_ctx *c, const uint64_t *in, uint64_t *out, size_t n);
int finish_results(focr_ctx *c);

__device__ __forceinline__ int32_t total_key(float f) {  // f32::total_cmp as a signed-int order
    int32_t b = __float_as_int(f);
    return b ^ (int32_t)(((uint32_t)(b >> 31)) >> 1);
}

// (1) keep_y: rows with a kept hit of similarity >= anchor_threshold; also the extent of every (page, row) line in
// the sorted hit list (one thread per hit looks at its neighbours), so that the line walk needs no search.
__global__ void mark_anchor_rows(const uint64_t *__restrict__ hkeys, const float *__restrict__ hsims, const uint8_t *__restrict__ keep,
                                 const uint64_t *__restrict__ n_p, uint64_t ub, KeyFmt fmt, float anchor, uint32_t r_h, uint8_t *__restrict__ keep_row,
                                 uint32_t *__restrict__ line_b, uint32_t *__restrict__ line_e) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)min(*n_p, ub);  // device-side count, clamped to what the grid and buffers were sized for
    if (i >= n) return;
    const uint64_t k = hkeys[i];
    const size_t row = (size_t)fmt.page(k) * r_h + fmt.y(k);
    if (i == 0 || fmt.line(hkeys[i - 1]) != fmt.line(k)) line_b[row] = (uint32_t)i;
    if (i + 1 == n || fmt.line(hkeys[i + 1]) != fmt.line(k)) line_e[row] = (uint32_t)(i + 1);
    if (keep[i] && hsims[i] >= anchor) keep_row[row] = 1;
}

// wave64 unsigned max via DPP (row_shr 1/2/4/8, row_bcast 15/31): the result is uniform (read from lane 63)
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// One wave per anchored (page, row): walks the row's extent of the (page, y, x, t)-sorted hit list 64 elements at a
// time.  Hits cut off by the per-call cap (keep[i] == 0) are not part of the reference's all_hits and are transparent
// here.  Groups are anchored on their first element (partition_by keeps `last` until a group closes,
// src/ncc.rs:1042-1048), so group boundaries are sequential, but every step is a handful of wave operations: a ballot
// for the group's extent inside the chunk, a DPP max-reduction of the similarity order (f32::total_cmp), and a ballot
// of the lanes that reach it — the highest such lane is the LAST maximum, which is what max_by keeps (:761-764).
// Outputs: choice[b + k] = winning element of the row's k-th group, row_groups[row] = 1<<32 | number of groups.
constexpr uint32_t WALK_ROWS = 16;  // (page, row) entries per wave
__global__ __launch_bounds__(256) void walk_lines(const uint64_t *__restrict__ keys, const float *__restrict__ sims,
                                                  const uint8_t *__restrict__ keep, KeyFmt fmt, uint32_t n_rows_total,
                                                  int32_t overlap, const uint8_t *__restrict__ keep_row,
                                                  const uint32_t *__restrict__ line_b, const uint32_t *__restrict__ line_e,
                                                  uint32_t *__restrict__ choice, uint64_t *__restrict__ row_groups) {
    // A wave looks at WALK_ROWS consecutive (page, row) entries at once and walks the anchored ones among them (one text line in
    // fifteen rows at BASELINE configs[1]): one wave per ROW meant 92 160 waves per batch of which 6 000 had work — 77 us alone and
    // 0.33 ms in flight for a few microseconds of arithmetic.
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t row_first = wave_id * WALK_ROWS;
    if (row_first >= n_rows_total) return;
    const uint32_t my_row = row_first + lane;
    const bool mine = lane < WALK_ROWS && my_row < n_rows_total && keep_row[my_row] != 0;  // keep_row / line_b / line_e / row_groups are [page][y] with pitch r_h
    const uint32_t my_b = mine ? line_b[my_row] : 0u, my_e = mine ? line_e[my_row] : 0u;  // written by mark_anchor_rows for every row that has hits
    for (uint64_t todo = __builtin_amdgcn_ballot_w64(mine); todo; todo &= todo - 1) {
    const int src = (int)__builtin_ctzll(todo);
    const uint32_t wave = row_first + (uint32_t)src;  // the row this pass walks
    const uint64_t b = (uint32_t)__builtin_amdgcn_readlane((int)my_b, src), e = (uint32_t)__builtin_amdgcn_readlane((int)my_e, src);

    uint32_t groups = 0;
    bool open = false;            // a group is open (carried across chunks)
    int32_t anchor = 0;           // x of the open group's first element
    uint32_t best_ord = 0, best_idx = 0;  // winner so far of the open group: total_cmp order, element index
    for (uint64_t base = b; base < e; base += 64) {
        const uint64_t i = base + lane;
        const bool valid = i < e && keep[i];
        const int32_t x = valid ? (int32_t)fmt.x(keys[i]) : 0x7fffffff;
        const uint32_t ord = valid ? ((uint32_t)total_key(sims[i]) ^ 0x80000000u) : 0u;  // unsigned order of total_cmp
        const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid);
        uint32_t pos = 0;  // wave-uniform cursor inside the chunk
        while (pos < 64) {
            if (!open) {
                const uint64_t cand = vmask & (~0ull << pos);  // the next kept element opens a group
                if (!cand) break;
                pos = (uint32_t)__builtin_ctzll(cand);
                anchor = __builtin_amdgcn_readlane(x, (int)pos);
                best_ord = 0;
                best_idx = 0;
                open = true;
            }
            // kept members of the open group at/after pos; the first kept non-member closes it
            const bool in = valid && lane >= pos && (x - anchor <= overlap) && (anchor - x <= overlap);
            const uint64_t inmask = __builtin_amdgcn_ballot_w64(in);
            const uint64_t brk = vmask & ~inmask & (~0ull << pos);
            const uint32_t stop = brk ? (uint32_t)__builtin_ctzll(brk) : 64u;
            const bool member = in && lane < stop;
            const uint32_t mx = wave_umax(member ? ord : 0u);
            const uint64_t top = __builtin_amdgcn_ballot_w64(member && ord == mx);
            if (top && mx >= best_ord) {  // later elements win ties: this chunk's members come after the carried ones
                best_ord = mx;
                best_idx = (uint32_t)(base - b) + (63u - (uint32_t)__builtin_clzll(top));
            }
            if (stop < 64) {  // the group closes inside this chunk
                if (lane == 0) choice[b + groups] = (uint32_t)b + best_idx;
                groups++;
                open = false;
            }
            pos = stop;
        }
    }
    if (open) {  // the row ends with a group still open
        if (lane == 0) choice[b + groups] = (uint32_t)b + best_idx;
        groups++;
    }
    if (lane == 0) row_groups[wave] = ((uint64_t)1 << 32) | groups;
    }  // anchored rows of this wave
}

// one thread per hit slot: slot i of a row is the row's (i - line_b)-th output character if the row has that many groups
__global__ void emit_chars(const uint64_t *__restrict__ keys, const float *__restrict__ sims, const uint64_t *__restrict__ n_p, uint64_t ub, KeyFmt fmt, uint32_t r_h,
                           const uint8_t *__restrict__ keep_row, const uint32_t *__restrict__ line_b,
                           const uint32_t *__restrict__ choice, const uint64_t *__restrict__ row_groups,
                           const uint64_t *__restrict__ scanned, const uint32_t *__restrict__ t_w,
                           const uint32_t *__restrict__ t_h, const uint32_t *__restrict__ t_letter,
                           uint64_t *__restrict__ line_char_off, focr_hit_t *__restrict__ chars) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(*n_p, ub)) return;
    const uint64_t ki = keys[i];
    const size_t row = (size_t)fmt.page(ki) * r_h + fmt.y(ki);
    if (!keep_row[row]) return;
    const uint32_t k = (uint32_t)i - line_b[row];
    if (k >= (uint32_t)row_groups[row]) return;
    const uint64_t sc = scanned[row];
    const uint64_t off = (sc & 0xffffffffu) + k;
    if (k == 0) line_char_off[sc >> 32] = sc & 0xffffffffu;
    const uint32_t e = choice[i];
    const uint64_t key = keys[e];
    const uint32_t t = fmt.t(key);
    focr_hit_t h;
    h.x = (uint16_t)fmt.x(key);
    h.y = (uint16_t)fmt.y(key);
    h.w = (uint16_t)t_w[t];
    h.h = (uint16_t)t_h[t];
    h.similarity = sims[e];
    h.letter = t_letter[t];
    h.template_index = t;
    chars[off] = h;
}

// page_line_off[p] = number of kept lines on pages < p = the row scan at the page's first row
__global__ void page_offsets(const uint64_t *__restrict__ scanned, uint32_t r_h, uint32_t n_pages, uint64_t *__restrict__ page_line_off) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > n_pages) return;
    page_line_off[p] = scanned[(size_t)min(p, n_pages) * r_h] >> 32;  // p == n_pages: the scan's grand total
}

// the batch's characters to a caller's device buffer, queued behind process_hits on the context's stream: their number is still
// on the device (the low word of the row scan's grand total), so the copy is a kernel that reads it there
__global__ __launch_bounds__(256) void copy_chars_kernel(const uint32_t *__restrict__ src, const uint64_t *__restrict__ total, uint32_t *__restrict__ dst, uint64_t dst_dwords) {
    constexpr uint64_t DW = sizeof(focr_hit_t) / 4;
    const uint64_t n = min((*total & 0xffffffffull) * DW, dst_dwords);  // a buffer that is too small is reported when the batch is completed
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// Queue the copy of the last focr_process_hits' characters into dst (device memory, dst_bytes) on the context's stream; false if
// there is nothing queued to copy from (no hits at all / results already final): the caller copies after completion instead.
bool post_queue_chars_copy(focr_ctx *c, void *dst, size_t dst_bytes) {
    static_assert(sizeof(focr_hit_t) % 4 == 0, "focr_hit_t is copied as dwords");
    if (!c->post_pending || !c->post_scanned.p || !c->post_chars.p || !dst) return false;
    const size_t n_rows_total = c->n_pages * c->r_h;
    hipLaunchKernelGGL(copy_chars_kernel, dim3(64), dim3(256), 0, c->stream, (const uint32_t *)c->post_chars.p, (const uint64_t *)c->post_scanned.p + n_rows_total,
                       (uint32_t *)dst, (uint64_t)(dst_bytes / 4));
    return hipGetLastError() == hipSuccess;
}

}  // namespace focr

using namespace focr;

extern "C" {

int focr_process_hits(focr_ctx_t *c, float anchor_threshold, int32_t overlap) {
    if (!c) return FOCR_ERR_INVALID;
    if (!c->scanned) return fail(c, FOCR_ERR_STATE, "focr_process_hits: no scan results");
    FOCR_HIP(c, hipSetDevice(c->device));
    c->processed = false;
    c->lines_on_host = false;
    c->post_anchor = anchor_threshold;
    c->post_overlap = overlap;
    c->n_chars = c->n_lines = 0;
    // all hits in (page, y, x, t) order; d_keep marks the ones that survive their call's cap.  Their number lives on the
    // device (c->d_n_hits); `ub` is what the scan sized its buffers for (the exact count unless the scan ran on estimates)
    const size_t ub = c->ub_hits, n_pages = c->n_pages;
    if (!c->sizes_pending && (c->n_hits == 0 || c->n_matches == 0)) {  // the reference panics on an empty hit list (src/ncc.rs:1040); we return zero lines
        c->processed = true;
        c->ms[4] = 0.f;
        return FOCR_OK;
    }
    if (ub >= 0xffffffffull) return fail(c, FOCR_ERR_OVERFLOW, "focr_process_hits: more than 2^32 hits in one batch");
    FOCR_HIP(c, hipEventRecord(c->ev[5], c->stream));
    // grow-only device scratch (no allocation in the steady state); outputs are sized by their bounds: characters <= hits,
    // lines <= page rows
    const size_t n_rows_total = n_pages * c->r_h;
    uint8_t *keep_row = (uint8_t *)c->post_keep.ensure(c, n_rows_total + 8);
    uint32_t *choice = (uint32_t *)c->post_choice.ensure(c, (ub + 1) * 4);
    uint64_t *packed = (uint64_t *)c->post_packed.ensure(c, (n_rows_total + 1) * 8);    // per row: 1<<32 | groups
    uint64_t *scanned = (uint64_t *)c->post_scanned.ensure(c, (n_rows_total + 1) * 8);
    uint64_t *d_page_off = (uint64_t *)c->post_page_off.ensure(c, (n_pages + 1) * 8);
    uint32_t *line_b = (uint32_t *)c->post_line_be.ensure(c, n_rows_total * 8), *line_e = line_b ? line_b + n_rows_total : nullptr;
    uint64_t *d_line_off = (uint64_t *)c->post_line_off.ensure(c, (n_rows_total + 1) * 8);
    focr_hit_t *d_chars = (focr_hit_t *)c->post_chars.ensure(c, (ub + 1) * sizeof(focr_hit_t));
    if (!line_b || !keep_row || !choice || !packed || !scanned || !d_page_off || !d_line_off || !d_chars)
        return fail(c, FOCR_ERR_NOMEM, "focr_process_hits: hipMalloc failed");
    const uint8_t *keep = (const uint8_t *)c->ord_keep.p;
    {
        ClearList clear{};  // one launch instead of two memsets (common.h)
        if (!clear.add(keep_row, n_rows_total)) return fail(c, FOCR_ERR_INVALID, "focr_process_hits: clear list full or region too large");
        if (!clear.add(packed, (n_rows_total + 1) * 8)) return fail(c, FOCR_ERR_INVALID, "focr_process_hits: clear list full or region too large");
        if (int rc = launch_clear(c, clear)) return rc;
    }
    const unsigned nb = (unsigned)((ub + 255) / 256);
    if (ub)
        hipLaunchKernelGGL(mark_anchor_rows, dim3(nb), dim3(256), 0, c->stream, c->d_hkeys, c->d_hsims, keep, c->d_n_hits, (uint64_t)ub, c->fmt,
                           anchor_threshold, (uint32_t)c->r_h, keep_row, line_b, line_e);
    hipLaunchKernelGGL(walk_lines, dim3((unsigned)(((n_rows_total + WALK_ROWS - 1) / WALK_ROWS * 64 + 255) / 256)), dim3(256), 0, c->stream, c->d_hkeys, c->d_hsims,
                       keep, c->fmt, (uint32_t)n_rows_total, overlap, keep_row, line_b, line_e, choice, packed);
    int rc;
    if ((rc = exclusive_scan_u64(c, packed, scanned, n_rows_total + 1))) return rc;
    // packed[n_rows_total] = 0, so the scan's last entry is the grand total (lines << 32 | characters): into the result block
    FOCR_HIP(c, hipMemcpyAsync(c->h_res + 3, scanned + n_rows_total, 8, hipMemcpyDeviceToHost, c->stream));
    if (ub)
        hipLaunchKernelGGL(emit_chars, dim3(nb), dim3(256), 0, c->stream, c->d_hkeys, c->d_hsims, c->d_n_hits, (uint64_t)ub, c->fmt, (uint32_t)c->r_h,
                           keep_row, line_b, choice, packed, scanned, c->d_t_w, c->d_t_h, c->d_t_letter, d_line_off, d_chars);
    hipLaunchKernelGGL(page_offsets, dim3((unsigned)((n_pages + 1 + 255) / 256)), dim3(256), 0, c->stream, scanned, (uint32_t)c->r_h,
                       (uint32_t)n_pages, d_page_off);
    FOCR_HIP(c, hipGetLastError());
    FOCR_HIP(c, hipEventRecord(c->ev[6], c->stream));
    c->post_pending = true;
    c->processed = true;
    if (!c->sizes_pending) return finish_results(c);  // exact sizes: complete now, as the call always did
    return FOCR_OK;
}

// results stay in HBM until somebody asks for them
static int fetch_lines(focr_ctx *c) {
    int rc0 = finish_results(c);
    if (rc0) return rc0;
    if (c->lines_on_host) return FOCR_OK;
    c->h_page_line_off.assign(c->n_pages + 1, 0);
    c->h_line_char_off.assign(c->n_lines + 1, 0);
    c->h_chars.resize(c->n_chars);
    if (c->n_matches && (c->n_lines || c->n_chars)) {
        FOCR_HIP(c, hipSetDevice(c->device));
        if (c->n_lines)
            FOCR_HIP(c, hipMemcpyAsync(c->h_line_char_off.data(), c->post_line_off.p, c->n_lines * 8, hipMemcpyDeviceToHost, c->io_stream));
        FOCR_HIP(c, hipMemcpyAsync(c->h_page_line_off.data(), c->post_page_off.p, (c->n_pages + 1) * 8, hipMemcpyDeviceToHost, c->io_stream));
        if (c->n_chars)
            FOCR_HIP(c, hipMemcpyAsync(c->h_chars.data(), c->post_chars.p, c->n_chars * sizeof(focr_hit_t), hipMemcpyDeviceToHost, c->io_stream));
        FOCR_HIP(c, hipStreamSynchronize(c->io_stream));
    }
    c->h_line_char_off[c->n_lines] = c->n_chars;
    c->lines_on_host = true;
    return FOCR_OK;
}

size_t focr_total_chars(focr_ctx_t *c) { return (c && c->processed && finish_results(c) == FOCR_OK) ? c->n_chars : 0; }
size_t focr_total_lines(focr_ctx_t *c) { return (c && c->processed && finish_results(c) == FOCR_OK) ? c->n_lines : 0; }

int focr_get_lines(focr_ctx_t *c, uint64_t *page_line_offsets, uint64_t *line_char_offsets, focr_hit_t *chars) {
    if (!c) return FOCR_ERR_INVALID;
    if (!c->processed) return fail(c, FOCR_ERR_STATE, "focr_get_lines: call focr_process_hits first");
    int rc = fetch_lines(c);
    if (rc) return rc;
    if (page_line_offsets) memcpy(page_line_offsets, c->h_page_line_off.data(), c->h_page_line_off.size() * 8);
    if (line_char_offsets) memcpy(line_char_offsets, c->h_line_char_off.data(), c->h_line_char_off.size() * 8);
    if (chars && c->n_chars) memcpy(chars, c->h_chars.data(), c->n_chars * sizeof(focr_hit_t));
    return FOCR_OK;
}

// Same as focr_get_lines but straight from the device into the caller's buffers (page-locked ones make it a plain DMA)
int focr_get_lines_into(focr_ctx_t *c, uint64_t *page_line_offsets, uint64_t *line_char_offsets, focr_hit_t *chars) {
    if (!c || !page_line_offsets || !line_char_offsets || !chars) return FOCR_ERR_INVALID;
    if (!c->processed) return fail(c, FOCR_ERR_STATE, "focr_get_lines_into: call focr_process_hits first");
    int rc = finish_results(c);
    if (rc) return rc;
    FOCR_HIP(c, hipSetDevice(c->device));
    if (c->n_matches && (c->n_lines || c->n_chars)) {
        if (c->n_lines) FOCR_HIP(c, hipMemcpyAsync(line_char_offsets, c->post_line_off.p, c->n_lines * 8, hipMemcpyDeviceToHost, c->io_stream));
        FOCR_HIP(c, hipMemcpyAsync(page_line_offsets, c->post_page_off.p, (c->n_pages + 1) * 8, hipMemcpyDeviceToHost, c->io_stream));
        if (c->n_chars) FOCR_HIP(c, hipMemcpyAsync(chars, c->post_chars.p, c->n_chars * sizeof(focr_hit_t), hipMemcpyDeviceToHost, c->io_stream));
        FOCR_HIP(c, hipStreamSynchronize(c->io_stream));
    } else {
        for (size_t p = 0; p <= c->n_pages; p++) page_line_offsets[p] = 0;
    }
    line_char_offsets[c->n_lines] = c->n_chars;
    return FOCR_OK;
}

const focr_hit_t *focr_lines_device_chars(focr_ctx_t *c) {
    return (c && c->processed && finish_results(c) == FOCR_OK && c->n_chars) ? (const focr_hit_t *)c->post_chars.p : nullptr;
}

}  // extern "C"
