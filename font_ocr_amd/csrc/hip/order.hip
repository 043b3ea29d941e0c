// order.hip — restore the reference's emission order and its per-call cap.
//
// The reference emits, per (page, template) call, matches in strictly (y, x)-ascending order and stops at
// n_out (src/ncc.cpp:224-228, 370-374); get_hits then concatenates the calls template after template and
// process_hits re-sorts everything by (y, x) (src/ncc.rs:741-752).  The device keeps ONE list of hits, sorted
// by the packed key (page, y, x, template) — process_hits order — and derives the per-call lists from it:
//
//   hits sorted by (page, y, x, t)            <- one radix sort over the key's ~36 significant bits
//   stable sort of (page*T + t, hit index)    <- 2-3 more passes; stable, so (y, x) order survives inside a segment
//   rank inside the (page, t) segment < cap   <- d_keep[hit]: what the reference's early stop keeps
//   CSR match lists in (page, t, y, x) order  <- exactly what N x T reference calls return
//
// The radix sorts and the prefix sums are rocPRIM device primitives (plain library ops on ~1e6 keys, not the hot
// path); everything else is hand-written.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace focr {

__device__ __forceinline__ uint64_t lower_bound_u64(const uint64_t *__restrict__ a, uint64_t n, uint64_t v) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

static int ensure_sort_tmp(focr_ctx *c, size_t tmp) {
    if (c->sort_tmp_bytes >= tmp && c->d_sort_tmp) return FOCR_OK;
    FOCR_HIP(c, hipStreamSynchronize(c->stream));
    if (c->d_sort_tmp) (void)hipFree(c->d_sort_tmp);
    c->d_sort_tmp = nullptr;
    if (hipMalloc(&c->d_sort_tmp, tmp ? tmp : 16) != hipSuccess) return fail(c, FOCR_ERR_NOMEM, "order: hipMalloc failed");
    c->sort_tmp_bytes = tmp ? tmp : 16;
    return FOCR_OK;
}

int sort_pairs_u64_f32(focr_ctx *c, uint64_t *&keys, uint64_t *&keys_alt, float *&vals, float *&vals_alt, size_t n,
                       unsigned end_bit) {
    if (n < 2) return FOCR_OK;
    size_t tmp = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp, keys, keys_alt, vals, vals_alt, n, 0u, end_bit, c->stream) != hipSuccess)
        return fail(c, FOCR_ERR_NO_DEVICE, "radix_sort_pairs (size query) failed");
    int rc = ensure_sort_tmp(c, tmp);
    if (rc) return rc;
    if (rocprim::radix_sort_pairs(c->d_sort_tmp, tmp, keys, keys_alt, vals, vals_alt, n, 0u, end_bit, c->stream) != hipSuccess)
        return fail(c, FOCR_ERR_NO_DEVICE, "radix_sort_pairs failed");
    std::swap(keys, keys_alt);
    std::swap(vals, vals_alt);
    return FOCR_OK;
}

int sort_keys_u64(focr_ctx *c, uint64_t *&keys, uint64_t *&keys_alt, size_t n, unsigned end_bit) {
    if (n < 2) return FOCR_OK;
    size_t tmp = 0;
    if (rocprim::radix_sort_keys(nullptr, tmp, keys, keys_alt, n, 0u, end_bit, c->stream) != hipSuccess)
        return fail(c, FOCR_ERR_NO_DEVICE, "radix_sort_keys (size query) failed");
    int rc = ensure_sort_tmp(c, tmp);
    if (rc) return rc;
    if (rocprim::radix_sort_keys(c->d_sort_tmp, tmp, keys, keys_alt, n, 0u, end_bit, c->stream) != hipSuccess)
        return fail(c, FOCR_ERR_NO_DEVICE, "radix_sort_keys failed");
    std::swap(keys, keys_alt);
    return FOCR_OK;
}

// exclusive prefix sum of n u64 values on the context's stream (rocPRIM), scratch grown on demand
int exclusive_scan_u64(focr_ctx *c, const uint64_t *in, uint64_t *out, size_t n) {
    if (n == 0) return FOCR_OK;
    size_t tmp = 0;
    if (rocprim::exclusive_scan(nullptr, tmp, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream) != hipSuccess)
        return fail(c, FOCR_ERR_NO_DEVICE, "exclusive_scan (size query) failed");
    int rc = ensure_sort_tmp(c, tmp);
    if (rc) return rc;
    if (rocprim::exclusive_scan(c->d_sort_tmp, tmp, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream) != hipSuccess)
        return fail(c, FOCR_ERR_NO_DEVICE, "exclusive_scan failed");
    return FOCR_OK;
}

// Every kernel below takes its element count from device memory (`n_p`), clamped to the host-side upper bound `ub`
// its grid and buffers were sized for: the host never has to wait for a count between the phases of a scan
// (scan_mfma.hip: "sizes").  With exact sizes ub == *n_p.

// MFMA path: candidates (sorted) + flags -> dense hit arrays, order preserved
__global__ void compact_hits(const uint64_t *__restrict__ keys, const float *__restrict__ sims, const uint64_t *__restrict__ flags,
                             const uint64_t *__restrict__ pos, const unsigned long long *__restrict__ n_p, uint64_t ub,
                             uint64_t *__restrict__ hkeys, float *__restrict__ hsims) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = min((uint64_t)*n_p, ub);
    if (i >= n || !flags[i]) return;
    hkeys[pos[i]] = keys[i];
    hsims[pos[i]] = sims[i];
}

// Entries past the count get the largest segment key, so that the (stable) sort leaves them behind the real ones.
__global__ void build_segment_keys(const uint64_t *__restrict__ hkeys, const uint64_t *__restrict__ n_p, uint64_t ub, KeyFmt fmt,
                                   uint32_t n_templates, uint32_t page_base, uint64_t *__restrict__ k2, float *__restrict__ v) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ub) return;
    if (i >= *n_p) {
        k2[i] = ~0ull;
        v[i] = 0.f;
        return;
    }
    const uint64_t k = hkeys[i];
    k2[i] = (uint64_t)(fmt.page(k) - page_base) * n_templates + fmt.t(k);
    v[i] = __uint_as_float((uint32_t)i);  // the hit's index rides along as the sort value
}

// one thread per (page, template) segment: extent in the segment-sorted list + capped count
__global__ void segment_bounds(const uint64_t *__restrict__ k2, const uint64_t *__restrict__ n_p, uint64_t ub, uint32_t n_seg, uint32_t cap,
                               uint64_t *__restrict__ seg_start, uint32_t *__restrict__ seg_count,
                               uint64_t *__restrict__ seg_count64) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_seg) return;
    if (s == n_seg) {
        seg_count64[s] = 0;
        return;
    }
    const uint64_t n = min(*n_p, ub);
    const uint64_t b = lower_bound_u64(k2, n, (uint64_t)s);
    const uint64_t e = lower_bound_u64(k2, n, (uint64_t)s + 1);
    uint64_t cnt = e - b;
    if (cnt > cap) cnt = cap;  // the reference stops the call at n_out matches (src/ncc.cpp:225-227)
    seg_start[s] = b;
    seg_count[s] = (uint32_t)cnt;
    seg_count64[s] = cnt;
}

// one thread per hit in segment order: rank inside the segment decides whether the reference would have emitted it
__global__ void emit_matches(const uint64_t *__restrict__ k2, const float *__restrict__ v, const uint64_t *__restrict__ n_p, uint64_t ub,
                             KeyFmt fmt, uint32_t cap, const uint64_t *__restrict__ hkeys, const float *__restrict__ hsims,
                             const uint64_t *__restrict__ seg_start, const uint64_t *__restrict__ seg_offset,
                             focr_match_t *__restrict__ out, uint8_t *__restrict__ keep) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= min(*n_p, ub)) return;
    const uint32_t seg = (uint32_t)k2[j], i = __float_as_uint(v[j]);
    const uint64_t rank = j - seg_start[seg];
    const bool kept = rank < cap;
    keep[i] = kept ? 1 : 0;
    if (!kept) return;
    const uint64_t k = hkeys[i];
    focr_match_t m;
    m.x = (uint16_t)fmt.x(k);
    m.y = (uint16_t)fmt.y(k);
    m.similarity = hsims[i];
    out[seg_offset[seg] + rank] = m;
}

// result sizes of a scan -> the context's device result block (copied to pinned host memory at the end of the scan):
// res[0] candidates, res[1] hits, res[2] matches after the cap, res[4] |= 1 if a count exceeded the bound its phase ran with
__global__ void record_scan_sizes(const unsigned long long *__restrict__ n_cand_p, uint64_t ub_c, const uint64_t *__restrict__ n_hits_p,
                                  uint64_t ub_h, const uint64_t *__restrict__ total_p, uint64_t *__restrict__ res) {
    if (n_cand_p) res[0] = *n_cand_p;
    res[1] = *n_hits_p;
    res[2] = *total_p;
    if ((n_cand_p && *n_cand_p > ub_c) || *n_hits_p > ub_h) res[4] |= 1;
}

static int ensure_seg_arrays(focr_ctx *c, size_t n_seg) {
    if (c->seg_alloc >= n_seg + 1) return FOCR_OK;
    FOCR_HIP(c, hipStreamSynchronize(c->stream));
    for (void *p : {(void *)c->d_seg_count, (void *)c->d_seg_start, (void *)c->d_seg_offset})
        if (p) (void)hipFree(p);
    c->d_seg_count = nullptr;
    c->d_seg_start = c->d_seg_offset = nullptr;
    c->seg_alloc = 0;
    // d_seg_start doubles as the u64 copy of the counts during the scan: 2*(n_seg+1) entries
    if (hipMalloc(&c->d_seg_count, (n_seg + 1) * 4) != hipSuccess || hipMalloc(&c->d_seg_start, 2 * (n_seg + 1) * 8) != hipSuccess ||
        hipMalloc(&c->d_seg_offset, (n_seg + 1) * 8) != hipSuccess)
        return fail(c, FOCR_ERR_NOMEM, "order: hipMalloc failed");
    c->seg_alloc = n_seg + 1;
    return FOCR_OK;
}

static int ensure_matches(focr_ctx *c, size_t want) {
    if (c->matches_alloc >= want && c->d_matches) return FOCR_OK;
    FOCR_HIP(c, hipStreamSynchronize(c->stream));
    want = std::max<size_t>(want + want / 8, 1024);
    if (c->d_matches) (void)hipFree(c->d_matches);
    c->d_matches = nullptr;
    c->matches_alloc = 0;
    if (hipMalloc(&c->d_matches, want * sizeof(focr_match_t)) != hipSuccess) return fail(c, FOCR_ERR_NOMEM, "order: hipMalloc failed");
    c->matches_alloc = want;
    return FOCR_OK;
}

// hits already sorted by the packed (page, y, x, t) key -> per-call lists + keep flags.  `n_p`: device-side number of
// hits, `ub`: host-side upper bound the buffers and grids are sized for (exact sizes: ub == *n_p).  Leaves the result
// sizes in c->d_res; no host wait.
int order_sorted_hits(focr_ctx *c, uint64_t *hkeys, float *hsims, const uint64_t *n_p, size_t ub, const unsigned long long *n_cand_p, size_t ub_c) {
    const size_t n_seg = c->sub_np * c->n_templates;  // (page, template) calls of the pages being processed
    int rc;
    if ((rc = ensure_seg_arrays(c, c->n_pages * c->n_templates))) return rc;
    if ((rc = ensure_matches(c, ub))) return rc;  // matches <= hits
    uint64_t *k2 = (uint64_t *)c->ord_k2.ensure(c, (ub + 1) * 8), *k2_alt = (uint64_t *)c->ord_k2_alt.ensure(c, (ub + 1) * 8);
    float *v = (float *)c->ord_v.ensure(c, (ub + 1) * 4), *v_alt = (float *)c->ord_v_alt.ensure(c, (ub + 1) * 4);
    uint8_t *keep = (uint8_t *)c->ord_keep.ensure(c, ub + 1);
    if (!k2 || !k2_alt || !v || !v_alt || !keep) return fail(c, FOCR_ERR_NOMEM, "order: hipMalloc failed");
    c->d_hkeys = hkeys;
    c->d_hsims = hsims;
    uint64_t *count64 = c->d_seg_start + (n_seg + 1);
    const unsigned nb = (unsigned)((ub + 255) / 256);
    if (ub) {
        hipLaunchKernelGGL(build_segment_keys, dim3(nb), dim3(256), 0, c->stream, hkeys, n_p, (uint64_t)ub, c->fmt, (uint32_t)c->n_templates,
                           (uint32_t)c->sub_p0, k2, v);
        FOCR_HIP(c, hipGetLastError());
        if ((rc = sort_pairs_u64_f32(c, k2, k2_alt, v, v_alt, ub, c->fmt.bp + c->fmt.bt))) return rc;  // LSD radix sort: stable
        // (the sort may have swapped k2/v with their alternates; they are local pointers, the DevBufs keep ownership)
    }
    hipLaunchKernelGGL(segment_bounds, dim3((unsigned)((n_seg + 1 + 255) / 256)), dim3(256), 0, c->stream, k2, n_p, (uint64_t)ub,
                       (uint32_t)n_seg, c->cap, c->d_seg_start, c->d_seg_count, count64);
    FOCR_HIP(c, hipGetLastError());
    if ((rc = exclusive_scan_u64(c, count64, c->d_seg_offset, n_seg + 1))) return rc;
    if (ub) {
        hipLaunchKernelGGL(emit_matches, dim3(nb), dim3(256), 0, c->stream, k2, v, n_p, (uint64_t)ub, c->fmt, c->cap, hkeys, hsims,
                           c->d_seg_start, c->d_seg_offset, c->d_matches, keep);
        FOCR_HIP(c, hipGetLastError());
    }
    hipLaunchKernelGGL(record_scan_sizes, dim3(1), dim3(1), 0, c->stream, n_cand_p, (uint64_t)ub_c, n_p, (uint64_t)ub, c->d_seg_offset + n_seg, c->d_res);
    FOCR_HIP(c, hipGetLastError());
    FOCR_HIP(c, hipMemcpyAsync(c->h_res, c->d_res, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    FOCR_HIP(c, hipEventRecord(c->ev[4], c->stream));
    c->d_n_hits = n_p;
    c->ub_hits = ub;
    c->ordered = true;
    return FOCR_OK;
}

// MFMA path: `keys` = candidates sorted by key (entries past the count: anything), `flags[i]` = passed the exact test
// (0 past the count, flags[ub_c] = 0), `pos` = scratch.  Afterwards the hits are dense in d_hit_keys / d_hit_sims_alt
// and their number is the device value pos[ub_c]; the caller goes on with order_sorted_hits.
int compact_candidates(focr_ctx *c, const uint64_t *keys, const float *sims, const uint64_t *flags, uint64_t *pos,
                       const unsigned long long *n_cand_p, size_t ub_c) {
    int rc;
    if ((rc = exclusive_scan_u64(c, flags, pos, ub_c + 1))) return rc;  // pos[ub_c] = number of hits
    if (ub_c) {  // d_hit_keys / d_hit_sims_alt are free here (capacity >= ub_c + 1)
        hipLaunchKernelGGL(compact_hits, dim3((unsigned)((ub_c + 255) / 256)), dim3(256), 0, c->stream, keys, sims, flags, pos, n_cand_p, (uint64_t)ub_c,
                           c->d_hit_keys, c->d_hit_sims_alt);
        FOCR_HIP(c, hipGetLastError());
    }
    return FOCR_OK;
}

// direct path: unordered, already verified hits in d_hit_keys / d_hit_sims; their exact number is known to the host
int order_hits(focr_ctx *c) {
    const size_t n = c->n_hits_raw;
    int rc;
    if ((rc = sort_pairs_u64_f32(c, c->d_hit_keys, c->d_hit_keys_alt, c->d_hit_sims, c->d_hit_sims_alt, n, c->fmt.bits()))) return rc;
    // the count as a device-side value for the shared kernels
    uint64_t *cnt = c->d_res + 7;
    c->n_hits_raw_u64 = n;
    FOCR_HIP(c, hipMemcpyAsync(cnt, &c->n_hits_raw_u64, 8, hipMemcpyHostToDevice, c->stream));
    return order_sorted_hits(c, c->d_hit_keys, c->d_hit_sims, cnt, n, nullptr, 0);
}

}  // namespace focr
