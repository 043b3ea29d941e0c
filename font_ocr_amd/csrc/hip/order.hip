// order.hip — restore the reference's emission order and its per-call cap.
//
// The reference emits, per (page, template) call, matches in strictly
// (y, x)-ascending order and stops at n_out (src/ncc.cpp:224-228, 370-374).
// The device scan produces an unordered hit list keyed
//   key = ((page * T + t) << 32) | (y << 16) | x,
// so a single radix sort on the key followed by "keep the first `cap` of every
// (page, t) segment" reproduces exactly the lists N x T reference calls return.
// The sort is rocPRIM's device radix sort (a plain library primitive, not on
// the hot path: ~1e4-1e6 keys); everything else is hand-written.
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

// one thread per (page, template) segment: extent in the sorted list + capped count
__global__ void segment_bounds(const uint64_t *__restrict__ keys, uint64_t n, uint32_t n_seg, uint32_t cap,
                               uint64_t *__restrict__ seg_start, uint32_t *__restrict__ seg_count,
                               uint64_t *__restrict__ seg_count64) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_seg) return;
    if (s == n_seg) {
        seg_count64[s] = 0;
        return;
    }
    uint64_t b = lower_bound_u64(keys, n, (uint64_t)s << 32);
    uint64_t e = lower_bound_u64(keys, n, ((uint64_t)s + 1) << 32);
    uint64_t cnt = e - b;
    if (cnt > cap) cnt = cap;
    seg_start[s] = b;
    seg_count[s] = (uint32_t)cnt;
    seg_count64[s] = cnt;
}

// one thread per sorted hit: rank inside its segment decides whether it survives the cap
__global__ void compact_matches(const uint64_t *__restrict__ keys, const float *__restrict__ sims, uint64_t n,
                                uint32_t n_templates, uint32_t cap, const uint64_t *__restrict__ seg_start,
                                const uint64_t *__restrict__ seg_offset, focr_match_t *__restrict__ out,
                                uint64_t *__restrict__ out_keys) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t key = keys[i];
    uint32_t seg = (uint32_t)(key >> 32);
    uint64_t rank = i - seg_start[seg];
    if (rank >= cap) return;
    uint64_t o = seg_offset[seg] + rank;
    uint32_t x = (uint32_t)(key & 0xffff), y = (uint32_t)((key >> 16) & 0xffff);
    focr_match_t m;
    m.x = (uint16_t)x;
    m.y = (uint16_t)y;
    m.similarity = sims[i];
    out[o] = m;
    uint32_t page = seg / n_templates, t = seg % n_templates;
    // process_hits order: page, then y, then x, then get_hits order (= template index)
    out_keys[o] = ((uint64_t)page << 48) | ((uint64_t)y << 32) | ((uint64_t)x << 16) | (uint64_t)t;
}

int sort_pairs_u64_f32(focr_ctx *c, uint64_t *&keys, uint64_t *&keys_alt, float *&vals, float *&vals_alt, size_t n,
                       unsigned end_bit) {
    if (n < 2) return FOCR_OK;
    size_t tmp = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp, keys, keys_alt, vals, vals_alt, n, 0u, end_bit, c->stream) != hipSuccess)
        return fail(c, FOCR_ERR_NO_DEVICE, "radix_sort_pairs (size query) failed");
    if (c->sort_tmp_bytes < tmp || !c->d_sort_tmp) {
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_sort_tmp) (void)hipFree(c->d_sort_tmp);
        c->d_sort_tmp = nullptr;
        if (hipMalloc(&c->d_sort_tmp, tmp) != hipSuccess) return fail(c, FOCR_ERR_NOMEM, "order: hipMalloc failed");
        c->sort_tmp_bytes = tmp;
    }
    if (rocprim::radix_sort_pairs(c->d_sort_tmp, tmp, keys, keys_alt, vals, vals_alt, n, 0u, end_bit, c->stream) != hipSuccess)
        return fail(c, FOCR_ERR_NO_DEVICE, "radix_sort_pairs failed");
    std::swap(keys, keys_alt);
    std::swap(vals, vals_alt);
    return FOCR_OK;
}

// exclusive prefix sum of n u64 values on the context's stream (rocPRIM), scratch grown on demand
int exclusive_scan_u64(focr_ctx *c, const uint64_t *in, uint64_t *out, size_t n) {
    if (n == 0) return FOCR_OK;
    size_t tmp = 0;
    if (rocprim::exclusive_scan(nullptr, tmp, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream) != hipSuccess)
        return fail(c, FOCR_ERR_NO_DEVICE, "exclusive_scan (size query) failed");
    if (c->sort_tmp_bytes < tmp || !c->d_sort_tmp) {
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_sort_tmp) (void)hipFree(c->d_sort_tmp);
        c->d_sort_tmp = nullptr;
        if (hipMalloc(&c->d_sort_tmp, tmp ? tmp : 16) != hipSuccess) return fail(c, FOCR_ERR_NOMEM, "order: hipMalloc failed");
        c->sort_tmp_bytes = tmp ? tmp : 16;
    }
    if (rocprim::exclusive_scan(c->d_sort_tmp, tmp, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream) != hipSuccess)
        return fail(c, FOCR_ERR_NO_DEVICE, "exclusive_scan failed");
    return FOCR_OK;
}

int sort_keys_u64(focr_ctx *c, uint64_t *&keys, uint64_t *&keys_alt, size_t n, unsigned end_bit) {
    if (n < 2) return FOCR_OK;
    size_t tmp = 0;
    if (rocprim::radix_sort_keys(nullptr, tmp, keys, keys_alt, n, 0u, end_bit, c->stream) != hipSuccess)
        return fail(c, FOCR_ERR_NO_DEVICE, "radix_sort_keys (size query) failed");
    if (c->sort_tmp_bytes < tmp || !c->d_sort_tmp) {
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_sort_tmp) (void)hipFree(c->d_sort_tmp);
        c->d_sort_tmp = nullptr;
        if (hipMalloc(&c->d_sort_tmp, tmp) != hipSuccess) return fail(c, FOCR_ERR_NOMEM, "order: hipMalloc failed");
        c->sort_tmp_bytes = tmp;
    }
    if (rocprim::radix_sort_keys(c->d_sort_tmp, tmp, keys, keys_alt, n, 0u, end_bit, c->stream) != hipSuccess)
        return fail(c, FOCR_ERR_NO_DEVICE, "radix_sort_keys failed");
    std::swap(keys, keys_alt);
    return FOCR_OK;
}

// ---- ordering for the MFMA path: candidates are sorted before the exact verify, survivors are flagged ------------
// pos = exclusive scan of flags over n+1 entries (flags[n] = 0), so pos[i] = number of hits before candidate i and
// pos[n] = total hits.

__global__ void segment_bounds_flagged(const uint64_t *__restrict__ keys, uint64_t n, uint32_t n_seg, uint32_t cap,
                                       const uint64_t *__restrict__ pos, uint64_t *__restrict__ seg_hit0,
                                       uint32_t *__restrict__ seg_count, uint64_t *__restrict__ seg_count64) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_seg) return;
    if (s == n_seg) {
        seg_count64[s] = 0;
        return;
    }
    const uint64_t b = lower_bound_u64(keys, n, (uint64_t)s << 32);
    const uint64_t e = lower_bound_u64(keys, n, ((uint64_t)s + 1) << 32);
    const uint64_t hb = pos[b], he = pos[e];  // b, e <= n and pos has n+1 entries
    uint64_t cnt = he - hb;
    if (cnt > cap) cnt = cap;
    seg_hit0[s] = hb;
    seg_count[s] = (uint32_t)cnt;
    seg_count64[s] = cnt;
}

__global__ void compact_flagged(const uint64_t *__restrict__ keys, const float *__restrict__ sims,
                                const uint64_t *__restrict__ flags, const uint64_t *__restrict__ pos, uint64_t n,
                                uint32_t n_templates, uint32_t cap, const uint64_t *__restrict__ seg_hit0,
                                const uint64_t *__restrict__ seg_offset, focr_match_t *__restrict__ out,
                                uint64_t *__restrict__ out_keys) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const uint64_t key = keys[i];
    const uint32_t seg = (uint32_t)(key >> 32);
    const uint64_t rank = pos[i] - seg_hit0[seg];
    if (rank >= cap) return;  // the reference stopped scanning this (page, template) at `cap` matches
    const uint64_t o = seg_offset[seg] + rank;
    const uint32_t x = (uint32_t)(key & 0xffff), y = (uint32_t)((key >> 16) & 0xffff);
    focr_match_t m;
    m.x = (uint16_t)x;
    m.y = (uint16_t)y;
    m.similarity = sims[i];
    out[o] = m;
    const uint32_t page = seg / n_templates, t = seg % n_templates;
    out_keys[o] = ((uint64_t)page << 48) | ((uint64_t)y << 32) | ((uint64_t)x << 16) | (uint64_t)t;
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
    if (c->d_match_keys) (void)hipFree(c->d_match_keys);
    c->d_matches = nullptr;
    c->d_match_keys = nullptr;
    c->matches_alloc = 0;
    if (hipMalloc(&c->d_matches, want * sizeof(focr_match_t)) != hipSuccess || hipMalloc(&c->d_match_keys, want * 8) != hipSuccess)
        return fail(c, FOCR_ERR_NOMEM, "order: hipMalloc failed");
    c->matches_alloc = want;
    return FOCR_OK;
}

int order_sorted_candidates(focr_ctx *c, const uint64_t *keys, const float *sims, const uint64_t *flags, uint64_t *pos, size_t n) {
    const size_t n_seg = c->n_pages * c->n_templates;
    int rc;
    if ((rc = ensure_seg_arrays(c, n_seg))) return rc;
    if ((rc = ensure_matches(c, n))) return rc;  // matches <= hits <= candidates: no size read-back needed before compaction
    if ((rc = exclusive_scan_u64(c, flags, pos, n + 1))) return rc;
    uint64_t *count64 = c->d_seg_start + (n_seg + 1);
    hipLaunchKernelGGL(segment_bounds_flagged, dim3((unsigned)((n_seg + 1 + 255) / 256)), dim3(256), 0, c->stream, keys, (uint64_t)n,
                       (uint32_t)n_seg, c->cap, pos, c->d_seg_start, c->d_seg_count, count64);
    FOCR_HIP(c, hipGetLastError());
    if ((rc = exclusive_scan_u64(c, count64, c->d_seg_offset, n_seg + 1))) return rc;
    if (n) {
        hipLaunchKernelGGL(compact_flagged, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, keys, sims, flags, pos,
                           (uint64_t)n, (uint32_t)c->n_templates, c->cap, c->d_seg_start, c->d_seg_offset, c->d_matches,
                           c->d_match_keys);
        FOCR_HIP(c, hipGetLastError());
    }
    uint64_t total = 0, hits = 0;
    FOCR_HIP(c, hipMemcpyAsync(&total, c->d_seg_offset + n_seg, 8, hipMemcpyDeviceToHost, c->stream));
    FOCR_HIP(c, hipMemcpyAsync(&hits, pos + n, 8, hipMemcpyDeviceToHost, c->stream));
    FOCR_HIP(c, hipEventRecord(c->ev[4], c->stream));
    FOCR_HIP(c, hipStreamSynchronize(c->stream));
    c->n_matches = (size_t)total;
    c->n_hits_raw = (size_t)hits;
    c->ordered = true;
    FOCR_HIP(c, hipEventElapsedTime(&c->ms[3], c->ev[3], c->ev[4]));
    FOCR_HIP(c, hipEventElapsedTime(&c->ms[5], c->ev[0], c->ev[4]));
    return FOCR_OK;
}

int order_hits(focr_ctx *c) {
    const size_t n = c->n_hits_raw;
    const size_t n_seg = c->n_pages * c->n_templates;
    int rc;
    if (c->seg_alloc < n_seg + 1) {
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        for (void *p : {(void *)c->d_seg_count, (void *)c->d_seg_start, (void *)c->d_seg_offset})
            if (p) (void)hipFree(p);
        c->d_seg_count = nullptr;
        c->d_seg_start = c->d_seg_offset = nullptr;
        c->seg_alloc = 0;
        // d_seg_start doubles as the u64 copy of the counts during the scan: 2*(n_seg+1) entries
        if (hipMalloc(&c->d_seg_count, (n_seg + 1) * 4) != hipSuccess ||
            hipMalloc(&c->d_seg_start, 2 * (n_seg + 1) * 8) != hipSuccess ||
            hipMalloc(&c->d_seg_offset, (n_seg + 1) * 8) != hipSuccess)
            return fail(c, FOCR_ERR_NOMEM, "order: hipMalloc failed");
        c->seg_alloc = n_seg + 1;
    }
    unsigned seg_bits = 1;
    while (((uint64_t)1 << seg_bits) < n_seg) seg_bits++;
    if ((rc = sort_pairs_u64_f32(c, c->d_hit_keys, c->d_hit_keys_alt, c->d_hit_sims, c->d_hit_sims_alt, n, 32 + seg_bits)))
        return rc;

    uint64_t *count64 = c->d_seg_start + (n_seg + 1);
    hipLaunchKernelGGL(segment_bounds, dim3((unsigned)((n_seg + 1 + 255) / 256)), dim3(256), 0, c->stream, c->d_hit_keys,
                       (uint64_t)n, (uint32_t)n_seg, c->cap, c->d_seg_start, c->d_seg_count, count64);
    FOCR_HIP(c, hipGetLastError());
    if ((rc = exclusive_scan_u64(c, count64, c->d_seg_offset, n_seg + 1))) return rc;
    uint64_t total = 0;
    FOCR_HIP(c, hipMemcpyAsync(&total, c->d_seg_offset + n_seg, 8, hipMemcpyDeviceToHost, c->stream));
    FOCR_HIP(c, hipStreamSynchronize(c->stream));
    c->n_matches = (size_t)total;
    if (c->matches_alloc < c->n_matches || !c->d_matches) {
        size_t want = std::max<size_t>(c->n_matches + c->n_matches / 8, 1024);
        if (c->d_matches) (void)hipFree(c->d_matches);
        if (c->d_match_keys) (void)hipFree(c->d_match_keys);
        c->d_matches = nullptr;
        c->d_match_keys = nullptr;
        c->matches_alloc = 0;
        if (hipMalloc(&c->d_matches, want * sizeof(focr_match_t)) != hipSuccess ||
            hipMalloc(&c->d_match_keys, want * 8) != hipSuccess)
            return fail(c, FOCR_ERR_NOMEM, "order: hipMalloc failed");
        c->matches_alloc = want;
    }
    if (n) {
        hipLaunchKernelGGL(compact_matches, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_hit_keys,
                           c->d_hit_sims, (uint64_t)n, (uint32_t)c->n_templates, c->cap, c->d_seg_start, c->d_seg_offset,
                           c->d_matches, c->d_match_keys);
        FOCR_HIP(c, hipGetLastError());
    }
    FOCR_HIP(c, hipEventRecord(c->ev[4], c->stream));
    FOCR_HIP(c, hipStreamSynchronize(c->stream));
    FOCR_HIP(c, hipEventElapsedTime(&c->ms[3], c->ev[3], c->ev[4]));
    FOCR_HIP(c, hipEventElapsedTime(&c->ms[5], c->ev[0], c->ev[4]));
    return FOCR_OK;
}

}  // namespace focr
