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
// That is the SORTING form (order_hits: the direct scan's unordered hits; order_sorted_hits_sort: banks above 4 096 templates);
// its radix sorts and prefix sums are rocPRIM device primitives.  The MFMA scan's default is the COUNTING form further down
// (order_sorted_hits: the row tail already delivers the hits sorted, and a hit's place in its call is a count, not a sort).
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
static int order_sorted_hits_sort(focr_ctx *c, uint64_t *hkeys, float *hsims, const uint64_t *n_p, size_t ub, const unsigned long long *n_cand_p, size_t ub_c) {
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

// ---------------------------------------------------------------------------------------------
// The same ordering pass without a sort ("counting" form, the default).  Hits are sorted by (page, y, x, t); a hit's place
// in its (page, template) call is the number of earlier hits of the same page with the same t — a stable counting sort by t
// inside each page, with T (templates) buckets:
//   order_units    page boundaries by binary search; every page's hits are cut into units of ORDER_UNIT consecutive hits
//   unit_hist      one wave per unit: histogram over t (LDS) -> uhist[unit][t]
//   unit_prefix    one thread per (page, t): running sum over the page's units -> uhist becomes the unit's base rank;
//                  the (page, t) total, capped, is the call's match count (src/ncc.cpp:225-227)
//   prefix         match offsets (CSR) over the (page, t) calls
//   unit_emit      one wave per unit, its hits 64 at a time, in order: rank = base + earlier hits of the same t in the unit
//                  (lanes of one group with equal t are found with one ballot per bit of t) -> keep flag, match at its place
// Six launches of a few microseconds each instead of ~14 (two radix passes over 2.7 M pairs with their histograms and scans,
// binary-searched segment bounds, a library scan).  Banks with more than ORDER_T_MAX templates or batches with more than
// 2^31 hits take the sorting form above.
constexpr uint32_t ORDER_UNIT = 2048, ORDER_T_MAX = 4096;

__global__ __launch_bounds__(1024) void order_units_kernel(const uint64_t *__restrict__ hkeys, const uint64_t *__restrict__ n_p, uint64_t ub, KeyFmt fmt,
                                                           uint32_t page_base, uint32_t n_pages, uint32_t *__restrict__ page_start /* n_pages + 1 */,
                                                           uint32_t *__restrict__ page_unit0 /* n_pages + 1 */, uint32_t *__restrict__ unit_page,
                                                           uint32_t *__restrict__ unit_begin, uint32_t *__restrict__ unit_end) {
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t carry_s;
    const uint64_t n = min(*n_p, ub);
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t p0 = 0; p0 <= n_pages; p0 += 1024) {  // pages 1024 at a time (a batch rarely has more)
        const uint32_t p = p0 + threadIdx.x;
        uint32_t b = 0, e = 0, units = 0;
        if (p < n_pages) {
            b = (uint32_t)lower_bound_u64(hkeys, n, fmt.pack(page_base + p, 0, 0, 0));
            e = (uint32_t)lower_bound_u64(hkeys, n, fmt.pack(page_base + p + 1, 0, 0, 0));
            units = (e - b + ORDER_UNIT - 1) / ORDER_UNIT;
            page_start[p] = b;
        } else if (p == n_pages) {
            page_start[p] = (uint32_t)n;
        }
        uint32_t incl = units;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if ((int)lane >= o) incl += v;
        }
        if (lane == 63) wave_sum[wv] = incl;
        __syncthreads();
        uint32_t u0 = carry_s + incl - units;
        for (uint32_t q = 0; q < wv; q++) u0 += wave_sum[q];
        if (p <= n_pages) page_unit0[p] = u0;
        for (uint32_t k = 0; k < units; k++) {
            unit_page[u0 + k] = p;
            unit_begin[u0 + k] = b + k * ORDER_UNIT;
            unit_end[u0 + k] = min(e, b + (k + 1) * ORDER_UNIT);
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = u0 + units;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void unit_hist_kernel(const uint64_t *__restrict__ hkeys, KeyFmt fmt, uint32_t T, const uint32_t *__restrict__ page_unit0,
                                                        uint32_t n_pages, const uint32_t *__restrict__ unit_begin, const uint32_t *__restrict__ unit_end,
                                                        uint32_t *__restrict__ uhist) {
    extern __shared__ uint32_t hist_lds[];  // 4 waves x T
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t *h = hist_lds + (size_t)wv * T;
    const uint32_t n_units = page_unit0[n_pages], n_waves = gridDim.x * 4;
    for (uint32_t u = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv); u < n_units; u += n_waves) {
        for (uint32_t i = lane; i < T; i += 64) h[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t b = unit_begin[u], e = unit_end[u];
        for (uint32_t i = b + lane; i < e; i += 64) atomicAdd(&h[fmt.t(hkeys[i])], 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t i = lane; i < T; i += 64) uhist[(size_t)u * T + i] = h[i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// exclusive prefix of n u32 counts -> n + 1 u64 offsets by ONE 1 024-thread workgroup, coalesced 16-byte loads (cnt padded with zeros to a
// multiple of 4 entries behind n): a kernel of its own (the sorting form) or the last workgroup of unit_prefix_kernel
typedef unsigned int ov4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void offsets_u64_block(const uint32_t *__restrict__ cnt, uint32_t n, uint64_t *__restrict__ off, uint64_t *wave_sum) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t seg = ((n + 15) / 16 + 255) / 256 * 256, b = min(n, wv * seg), e = min(n, b + seg);
    const ov4u *cnt4 = reinterpret_cast<const ov4u *>(cnt);
    uint64_t sum = 0;
    for (uint32_t i = b + 4 * lane; i < e; i += 256) {
        const ov4u v = cnt4[i / 4];
        sum += (uint64_t)v[0] + v[1] + v[2] + v[3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) wave_sum[wv] = sum;
    __syncthreads();
    uint64_t carry = 0;
    for (uint32_t q = 0; q < wv; q++) carry += wave_sum[q];
    for (uint32_t i0 = b; i0 < e; i0 += 256) {
        const uint32_t i = i0 + 4 * lane;
        ov4u v = ov4u{0, 0, 0, 0};
        if (i < e) v = cnt4[i / 4];
        const uint64_t tot4 = (uint64_t)v[0] + v[1] + v[2] + v[3];
        uint64_t incl = tot4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t u = __shfl_up(incl, o);
            if ((int)lane >= o) incl += u;
        }
        if (i < e) {
            uint64_t p0 = carry + incl - tot4;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (i + k < n) off[i + k] = p0;
                p0 += v[k];
            }
        }
        carry += __shfl(incl, 63);
    }
    if (threadIdx.x == 0) {
        uint64_t tot = 0;
        for (int q = 0; q < 16; q++) tot += wave_sum[q];
        off[n] = tot;
    }
}
__global__ __launch_bounds__(1024) void offsets_u64_kernel(const uint32_t *__restrict__ cnt, uint32_t n, uint64_t *__restrict__ off) {
    __shared__ uint64_t wave_sum[16];
    offsets_u64_block(cnt, n, off, wave_sum);
}

// one thread per (page, t) call: running sum over the page's units -> the units' base ranks, the call's capped match count; the
// kernel's LAST workgroup then turns the counts into the CSR offsets (offsets_u64_block: a launch less in the batch's chain)
__global__ __launch_bounds__(1024) void unit_prefix_kernel(uint32_t T, uint32_t n_pages, const uint32_t *__restrict__ page_unit0, uint32_t *__restrict__ uhist,
                                                           uint32_t cap, uint32_t *__restrict__ seg_count, uint32_t n_seg_padded, uint32_t *__restrict__ done,
                                                           uint64_t *__restrict__ seg_offset) {
    __shared__ uint64_t wave_sum[16];
    __shared__ uint32_t is_last;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;  // (page, t) call
    if (s < n_seg_padded) {
        if (s >= n_pages * T) {  // the padding the offsets read 16 bytes at a time
            seg_count[s] = 0;
        } else {
            const uint32_t p = s / T, t = s % T;
            uint32_t run = 0;
            for (uint32_t u = page_unit0[p]; u < page_unit0[p + 1]; u++) {
                const uint32_t v = uhist[(size_t)u * T + t];
                uhist[(size_t)u * T + t] = run;
                run += v;
            }
            seg_count[s] = min(run, cap);  // the reference stops the call at n_out matches (src/ncc.cpp:225-227)
        }
    }
    __syncthreads();  // the workgroup's stores are in this XCD's L2; one agent-scope release per workgroup writes them back (rows.hip, last_workgroup)
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        is_last = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (is_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (threadIdx.x == 0) *done = 0;  // for the next launch
        offsets_u64_block(seg_count, n_pages * T, seg_offset, wave_sum);
    }
}

__global__ __launch_bounds__(256) void unit_emit_kernel(const uint64_t *__restrict__ hkeys, const float *__restrict__ hsims, KeyFmt fmt, uint32_t T,
                                                        uint32_t page_base, const uint32_t *__restrict__ page_unit0, uint32_t n_pages,
                                                        const uint32_t *__restrict__ unit_page, const uint32_t *__restrict__ unit_begin,
                                                        const uint32_t *__restrict__ unit_end, const uint32_t *__restrict__ uhist, uint32_t cap,
                                                        const uint64_t *__restrict__ seg_offset, focr_match_t *__restrict__ out, uint8_t *__restrict__ keep) {
    extern __shared__ uint32_t cnt_lds[];  // 4 waves x T: hits of each t seen so far in the unit
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t *cnt = cnt_lds + (size_t)wv * T;
    const uint32_t n_units = page_unit0[n_pages], n_waves = gridDim.x * 4;
    const uint64_t lt_mask = (1ull << lane) - 1;
    for (uint32_t u = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv); u < n_units; u += n_waves) {
        const uint32_t *ubase = uhist + (size_t)u * T;
        for (uint32_t i = lane; i < T; i += 64) cnt[i] = ubase[i];  // the unit's base ranks
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t b = unit_begin[u], e = unit_end[u], p = unit_page[u];
        uint64_t k_next = b + lane < e ? hkeys[b + lane] : 0;  // the next group's key and similarity are loaded a group ahead
        float s_next = b + lane < e ? hsims[b + lane] : 0.f;
        for (uint32_t i0 = b; i0 < e; i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool valid = i < e;
            const uint64_t k = k_next;
            const float sim = s_next;
            if (i + 64 < e) {
                k_next = hkeys[i + 64];
                s_next = hsims[i + 64];
            }
            const uint32_t t = valid ? fmt.t(k) : 0xffffffffu;
            // lanes of this group with the same t: one ballot per bit of t
            uint64_t peers = __builtin_amdgcn_ballot_w64(valid);
            for (uint32_t bit = 0; bit < fmt.bt; bit++) {
                const bool one = (t >> bit) & 1;
                const uint64_t m = __builtin_amdgcn_ballot_w64(one);
                peers &= one ? m : ~m;
            }
            if (valid) {
                const uint32_t rank = cnt[t] + (uint32_t)__builtin_popcountll(peers & lt_mask);
                const bool kept = rank < cap;
                keep[i] = kept ? 1 : 0;
                if (kept) {
                    focr_match_t m;
                    m.x = (uint16_t)fmt.x(k);
                    m.y = (uint16_t)fmt.y(k);
                    m.similarity = sim;
                    out[seg_offset[(size_t)p * T + t] + rank] = m;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // the highest lane of every peer group moves the counter on
            if (valid && (peers >> lane) == 1ull) cnt[t] += (uint32_t)__builtin_popcountll(peers);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}

int order_sorted_hits(focr_ctx *c, uint64_t *hkeys, float *hsims, const uint64_t *n_p, size_t ub, const unsigned long long *n_cand_p, size_t ub_c) {
    const uint32_t T = (uint32_t)c->n_templates, n_pages = (uint32_t)c->sub_np;
    const size_t n_seg = (size_t)n_pages * T;
    if (T > ORDER_T_MAX || ub >= ((size_t)1 << 31) || n_seg >= ((size_t)1 << 31)) return order_sorted_hits_sort(c, hkeys, hsims, n_p, ub, n_cand_p, ub_c);
    int rc;
    if ((rc = ensure_seg_arrays(c, c->n_pages * c->n_templates + 8))) return rc;
    if ((rc = ensure_matches(c, ub))) return rc;  // matches <= hits
    const size_t max_units = ub / ORDER_UNIT + n_pages + 1;
    uint8_t *keep = (uint8_t *)c->ord_keep.ensure(c, ub + 1);
    // unit tables: page_start, page_unit0 (n_pages + 1 each), unit_page / begin / end (max_units each)
    uint32_t *tab = (uint32_t *)c->ord_v.ensure(c, (2 * ((size_t)n_pages + 1) + 3 * max_units) * 4);
    uint32_t *uhist = (uint32_t *)c->ord_k2.ensure(c, max_units * T * 4);
    if (!keep || !tab || !uhist) return fail(c, FOCR_ERR_NOMEM, "order: hipMalloc failed");
    uint32_t *page_start = tab, *page_unit0 = tab + n_pages + 1, *unit_page = page_unit0 + n_pages + 1, *unit_begin = unit_page + max_units,
             *unit_end = unit_begin + max_units;
    c->d_hkeys = hkeys;
    c->d_hsims = hsims;
    const unsigned unit_blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((max_units + 3) / 4, (size_t)c->n_cus * 4));
    const size_t lds = (size_t)4 * T * 4;
    hipLaunchKernelGGL(order_units_kernel, dim3(1), dim3(1024), 0, c->stream, hkeys, n_p, (uint64_t)ub, c->fmt, (uint32_t)c->sub_p0, n_pages, page_start, page_unit0,
                       unit_page, unit_begin, unit_end);
    FOCR_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(unit_hist_kernel, dim3(unit_blocks), dim3(256), lds, c->stream, hkeys, c->fmt, T, (const uint32_t *)page_unit0, n_pages,
                       (const uint32_t *)unit_begin, (const uint32_t *)unit_end, uhist);
    FOCR_HIP(c, hipGetLastError());
    const uint32_t n_seg_padded = (uint32_t)((n_seg + 3) / 4 * 4 + 4);
    // (page, t) totals + CSR offsets: the kernel's last workgroup does the prefix (its counter is zero between launches: the last workgroup resets it)
    hipLaunchKernelGGL(unit_prefix_kernel, dim3((n_seg_padded + 1023) / 1024), dim3(1024), 0, c->stream, T, n_pages, (const uint32_t *)page_unit0, uhist, c->cap,
                       c->d_seg_count, n_seg_padded, c->d_counter + ORDER_DONE_WORD, c->d_seg_offset);
    FOCR_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(unit_emit_kernel, dim3(unit_blocks), dim3(256), lds, c->stream, hkeys, hsims, c->fmt, T, (uint32_t)c->sub_p0, (const uint32_t *)page_unit0,
                       n_pages, (const uint32_t *)unit_page, (const uint32_t *)unit_begin, (const uint32_t *)unit_end, (const uint32_t *)uhist, c->cap,
                       (const uint64_t *)c->d_seg_offset, c->d_matches, keep);
    FOCR_HIP(c, hipGetLastError());
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
