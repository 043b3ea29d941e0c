// rows.hip — the hits-first row tail of the MFMA scan: candidates -> sorted, verified hits, without a global sort.
//
// The prefilter leaves an unordered list of candidate keys (page, y, x, t).  The reference's order (process_hits order:
// page, y, x, template; src/ncc.rs:741-752) used to be restored by a 5-pass radix sort of all candidates, followed by the
// exact verify, a flag scan and a compaction (12 launches, ~0.5 ms alone on the chip at BASELINE configs[1]: the legacy tail,
// still in scan_mfma.hip for banks this file does not cover).  But four candidates in ten fall to the verify, the verify needs
// no sorted input — it needs neighbouring lanes on neighbouring windows, and the scan kernels' flush order has that: the 64 keys
// of a flush are 64 adjacent windows of one page row — and a hit's (page, y) is one of only sub_np * r_h page rows (92 160 at
// configs[1]).  So the candidates are verified where they lie and only the HITS are bucketed by page row — by a segment of 2^k
// pixels of a page row where rows are wide and banks large (row_segments below; "row" in the names of this file means such a
// bucket) — and sorted per bucket:
//
//   verify_list       the reference arithmetic on every candidate in FLUSH order (verify_candidate_*, mfma_common.h), template rows
//   / verify_chunks   and records from LDS (in chunk passes for banks above it); a hit takes the next free slot of its bucket (one
//                     returning atomic per distinct bucket among the wave's hits) — similarity + slot in place; the kernel's last
//                     workgroup turns the buckets' hit counts into their dense positions, the hit total and the largest bucket
//   hit_scatter       hit -> hbase[bucket] + slot: the dense (key, similarity) arrays in bucket order, arbitrary inside a bucket
//   row_sort          one WAVE per bucket, keys with their similarities, in place: <= 64 keys ranked in registers by lane-to-lane
//                     comparison; else the sub-keys (x, t) in registers, counting sort by x in wave-private LDS, rank inside the
//                     x-bin by t.  Buckets above 1024 go onto a list and through a second launch (capacity 4096) -> (page, y, x, t)
//                     order: what order.hip takes over
//
// Three launches (+ the second capacity class of the sort where such a bucket is expected), no counting in the scan kernels' flush
// path, every pass behind the verify touches hits only.  A bucket above 4096 hits (very low thresholds): the placed hits go
// through the library radix sort; banks with templates taller than 32 px or more than 4096 templates: the legacy tail.  Exact
// mode knows the largest bucket before it chooses; estimated mode goes by the previous scan's, and a bucket that turns out too
// large sets the overflow bit: the batch is redone with exact sizes.  (Round 3's row tail — candidates bucketed, sorted, verified,
// compacted: seven launches — was kept for A/B through round 4; its last commit is 5ed8d3e.)
#include "mfma_common.h"

namespace focr {

int order_sorted_hits(focr_ctx *c, uint64_t *hkeys, float *hsims, const uint64_t *n_p, size_t ub, const unsigned long long *n_cand_p, size_t ub_c);
int ensure_hit_capacity(focr_ctx *c, size_t want);

// exclusive prefix of n u32 counts by ONE workgroup: base[0..n] (base[n] = total); *total_out = total, *max_out = max
// (u64 each; either may be null); zero[0..n) is cleared on the way if given (the scatter's per-row cursors).  Each of the 16
// waves owns a contiguous segment (a multiple of 256 entries) and walks it 256 entries at a time — 16-byte loads, coalesced
// — twice: sums first, then the prefix.  The arrays are padded to a multiple of 4 entries by their owner (rows_begin).
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
// the prefix as the work of ONE 1 024-thread workgroup: a kernel of its own (below), or the last workgroup of the kernel that
// produced the counts (the hits-first tail's verify: a launch less in the batch's chain of kernels)
__device__ __forceinline__ void row_prefix_block(const uint32_t *__restrict__ cnt, uint32_t n, uint32_t *__restrict__ base, uint32_t *__restrict__ zero,
                                                 uint64_t *__restrict__ total_out, uint64_t *__restrict__ max_out, uint32_t *wave_sum, uint32_t *wave_max) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t seg = ((n + 15) / 16 + 255) / 256 * 256, b = min(n, wv * seg), e = min(n, b + seg);
    const v4u *cnt4 = reinterpret_cast<const v4u *>(cnt);
    uint32_t sum = 0, mx = 0;
    for (uint32_t i0 = b + 4 * lane; i0 < e; i0 += 4 * 256) {
        v4u vv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            vv[q] = v4u{0, 0, 0, 0};
            if (i0 + q * 256 < e) vv[q] = cnt4[(i0 + q * 256) / 4];  // entries past n (inside the padding) are zero
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            sum += vv[q][0] + vv[q][1] + vv[q][2] + vv[q][3];
            mx = max(max(mx, max(vv[q][0], vv[q][1])), max(vv[q][2], vv[q][3]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum += (uint32_t)__shfl_xor((int)sum, o);
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    }
    if (lane == 0) {
        wave_sum[wv] = sum;
        wave_max[wv] = mx;
    }
    __syncthreads();
    uint32_t carry = 0;
    for (uint32_t q = 0; q < wv; q++) carry += wave_sum[q];
    for (uint32_t i00 = b; i00 < e; i00 += 4 * 256) {  // four tiles' loads in flight, then their scans
        v4u vv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t i = i00 + q * 256 + 4 * lane;
            vv[q] = v4u{0, 0, 0, 0};
            if (i < e) vv[q] = cnt4[i / 4];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t i = i00 + q * 256 + 4 * lane;
            const v4u v = vv[q];
            const uint32_t tot4 = v[0] + v[1] + v[2] + v[3];
            uint32_t incl = tot4;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = __shfl_up(incl, o);
                if ((int)lane >= o) incl += u;
            }
            if (i < e) {
                const uint32_t p0 = carry + incl - tot4;
                reinterpret_cast<v4u *>(base)[i / 4] = v4u{p0, p0 + v[0], p0 + v[0] + v[1], p0 + v[0] + v[1] + v[2]};
                if (zero) reinterpret_cast<v4u *>(zero)[i / 4] = v4u{0, 0, 0, 0};
            }
            carry += __shfl(incl, 63);
        }
    }
    __syncthreads();  // base[n] below may share a 16-byte group with the last wave's stores
    if (threadIdx.x == 0) {
        uint32_t tot = 0, m = 0;
        for (int q = 0; q < 16; q++) {
            tot += wave_sum[q];
            m = max(m, wave_max[q]);
        }
        base[n] = tot;
        if (total_out) *total_out = tot;
        if (max_out) *max_out = m;
    }
}
__global__ __launch_bounds__(1024) void row_prefix_kernel(const uint32_t *__restrict__ cnt, uint32_t n, uint32_t *__restrict__ base,
                                                          uint32_t *__restrict__ zero, uint64_t *__restrict__ total_out,
                                                          uint64_t *__restrict__ max_out) {
    __shared__ uint32_t wave_sum[16], wave_max[16];
    row_prefix_block(cnt, n, base, zero, total_out, max_out, wave_sum, wave_max);
}

// "Am I the kernel's last workgroup?" — called by every thread of a workgroup when its work is done.  The last one to arrive sees
// everything the others wrote (their release at the counter, its acquire here: the L1 of this CU may hold lines another CU has
// rewritten since) and may go on with work that needs all of it: the launch it saves stood 40-50 us in a batch's chain of kernels
// whenever other batches' kernels filled the chip.
struct TailWork {
    uint32_t *done;  // zeroed with the scan's counters (ClearList)
    uint32_t n_rows;
    uint32_t *hbase;
    uint64_t *total_out, *max_out;
};
__device__ __forceinline__ bool last_workgroup(uint32_t *done) {
    __shared__ uint32_t is_last;
    __syncthreads();  // the workgroup's stores have left its waves (workgroup-scope release: they are in this XCD's L2) ...
    if (threadIdx.x == 0) {
        // ... and ONE agent-scope release writes that L2's dirty lines back for the other XCDs (a workgroup lives on one XCD).  Not one
        // per thread: 8 192 L2 write-backs per launch instead of 512 doubled the verify's time alone on the chip (183 -> 383 us).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        is_last = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (is_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (threadIdx.x == 0) *done = 0;  // zero between launches (and zeroed with the scan's counters anyway)
    }
    return is_last != 0;
}

// Sort every row's candidates by (x, t) — the low bt + bx bits of the key, unique inside a row — in place.  One WAVE per row
// (fixed stride: dense text rows are spread evenly over the waves), wave-private LDS, no workgroup barrier: the row's
// sub-keys come into LDS with all loads in flight at once, then a counting sort by x-bin (bin = x >> xs, at most XBINS bins:
// one x per bin for pages up to 1024 px wide) — count, exclusive prefix, place — and every element finds its rank among the
// few elements of its own bin and goes straight back to the row's slots in global memory.  O(n) LDS operations per row.
//   LIST = false: all rows; a row above CAP goes onto `big` (BASELINE configs[1]: one or two rows per batch exceed 1024).
//   LIST = true : the rows on `big`, with a larger CAP and one wave per workgroup; a row above that CAP sets the overflow bit
//                 (batch redone through the legacy tail).
constexpr uint32_t XBINS = 1024;
constexpr unsigned VERIFY_THREADS = 1024;
//   PAY: every key carries a float (the hits-first tail sorts verified hits with their similarities): loaded with the keys,
//        stored at the key's new place; `limit` = entries the arrays hold (a bucket that would reach past it is left alone: the
//        hit count exceeded its estimate and the batch is redone, record_scan_sizes).
template <int CAP, int WAVES, bool LIST, bool PAY>
__global__ __launch_bounds__(WAVES * 64) void row_sort_kernel(uint32_t n_rows, const uint32_t *__restrict__ base, const uint32_t *__restrict__ fill,
                                                              uint64_t *__restrict__ bucket, uint32_t sub_bits, uint32_t bt, uint32_t seg_mask, uint32_t xs, uint32_t n_bins,
                                                              uint32_t *__restrict__ big, unsigned long long *__restrict__ flags_word, float *__restrict__ pay,
                                                              unsigned long long limit) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sort_lds[];  // per wave: XBINS + 1 bin starts, CAP placed sub-keys
    constexpr int K = CAP / 64;  // sub-keys per lane, in registers
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t *cur = sort_lds + (size_t)wv * (XBINS + 1 + CAP);  // cur[n_bins] = n after the prefix
    uint32_t *out = cur + XBINS + 1;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + wv), n_waves = gridDim.x * WAVES;
    const uint64_t sub_mask = (1ull << sub_bits) - 1;
    const uint32_t n_items = LIST ? min(big[0], n_rows) : n_rows;  // the list has room for every row
    // the next row's size and position are fetched (scalar loads) while the current row is sorted
    uint32_t it = wave, r = 0, n = 0, b = 0;
    if (it < n_items) {
        r = LIST ? big[1 + it] : it;
        n = fill[r];
        b = base[r];
    }
    while (it < n_items) {
        const uint32_t it2 = it + n_waves;
        uint32_t r2 = 0, n2 = 0, b2 = 0;
        if (it2 < n_items) {
            r2 = LIST ? big[1 + it2] : it2;
            n2 = fill[r2];
            b2 = base[r2];
        }
        if (PAY && (unsigned long long)b + n > limit) {
            // past the arrays' end (estimated sizes too small): nothing to sort, the batch is redone
        } else if (n > (uint32_t)CAP) {
            if (lane == 0) {
                if (LIST || !big) {  // beyond the last capacity class, or no second launch was planned for this batch (estimated sizes): redo
                    atomicOr(flags_word, 2ull);
                } else {
                    big[1 + atomicAdd(big, 1u)] = r;  // room for every row
                }
            }
        } else if (n >= 2 && n <= 64 && !LIST) {
            // a row that fits one key per lane (most rows that are not text lines): rank by comparing with every other lane's key
            // — no LDS, no bins; keys are unique, so the rank is the key's place
            uint64_t *row = bucket + b;
            const uint64_t key = row[(uint32_t)lane < n ? lane : n - 1];
            float pv = 0.f;
            if (PAY) pv = pay[b + ((uint32_t)lane < n ? lane : n - 1)];
            const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; j++) {  // j is wave-uniform: v_readlane
                const uint64_t other = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)khi, (int)j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)klo, (int)j);
                rank += other < key;
            }
            if ((uint32_t)lane < n) {
                row[rank] = key;
                if (PAY) pay[b + rank] = pv;
            }
        } else if (n >= 2) {
            uint64_t *row = bucket + b;
            uint32_t sub[K], slot[K];
            uint64_t key[K];
            float pv[PAY ? K : 1];
#pragma unroll
            for (int k = 0; k < K; k++) {  // all of the row's loads in flight at once: nothing but loads in this loop
                const uint32_t j = (uint32_t)lane + 64u * k;
                key[k] = 0;
                if (PAY) pv[k] = 0.f;
                if (64u * k < n) {  // wave-uniform branch; lanes past the end re-read the last key
                    key[k] = row[j < n ? j : n - 1];
                    if (PAY) pv[k] = pay[b + (j < n ? j : n - 1)];
                }
            }
#pragma unroll
            for (int k = 0; k < K; k++) sub[k] = (uint32_t)(key[k] & sub_mask);
            const uint32_t hi32 = (uint32_t)(key[0] >> 32), lo32 = (uint32_t)key[0];
            // (page, y) of the row from lane 0's first key (n >= 2: it has one)
            const uint64_t high = (((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(hi32) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(lo32)) & ~sub_mask;  // the builtin returns int: no sign extension
            for (uint32_t i = lane; i <= n_bins; i += 64) cur[i] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int k = 0; k < K; k++)  // count per x-bin; the returned value is the element's slot inside its bin
                if (64u * k < n && (uint32_t)lane + 64u * k < n) slot[k] = atomicAdd(&cur[((sub[k] >> bt) & seg_mask) >> xs], 1u);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            uint32_t carry = 0;  // exclusive prefix over the bins, 64 at a time; cur[n_bins] = n
            for (uint32_t i0 = 0; i0 <= n_bins; i0 += 64) {
                const uint32_t i = i0 + lane, v = i < n_bins ? cur[i] : 0;
                uint32_t incl = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t u = __shfl_up(incl, o);
                    if (lane >= o) incl += u;
                }
                if (i <= n_bins) cur[i] = carry + incl - v;
                carry += __shfl(incl, 63);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int k = 0; k < K; k++)
                if (64u * k < n && (uint32_t)lane + 64u * k < n) out[cur[((sub[k] >> bt) & seg_mask) >> xs] + slot[k]] = sub[k];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int k = 0; k < K; k++)
                if (64u * k < n && (uint32_t)lane + 64u * k < n) {  // rank among the few elements of the same bin
                    const uint32_t e = sub[k], bin = ((e >> bt) & seg_mask) >> xs, lo = cur[bin], hi = cur[bin + 1];
                    uint32_t rank = 0;
                    for (uint32_t i = lo; i < hi; i++) rank += out[i] < e;
                    row[lo + rank] = high | e;
                    if (PAY) pay[b + lo + rank] = pv[k];
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the next row reuses the LDS buffers
        }
        it = it2, r = r2, n = n2, b = b2;
    }
}

// The exact verify of the candidate list in flush order.  A candidate costs 15 page-row loads (neighbouring lanes: neighbouring
// windows, a few cache lines per wave instruction) and 15 template-row loads — a 64-way GATHER per wave instruction when they come
// from global memory, which kept the texture addresser busy for most of the kernel's time (round 2: 0.22 ms for 3.8 M candidates).
// MODE 0: template rows from global memory · 1: the bank's verify operand staged in LDS once per workgroup, 16 bytes per row (all
// of it fits 144 KiB) · 2: 12 bytes per row (every template at most 12 px wide and the whole operand within half a CU's LDS): 64
// VGPRs, two workgroups per CU — the kernel waits on memory four fifths of its time, and in flight it has only the CUs the scan
// leaves free, so waves per CU are what it runs on.
template <int MODE>
__global__ __launch_bounds__(VERIFY_THREADS, MODE == 2 ? 8 : 1) void verify_list_kernel(
    const uint64_t *__restrict__ cand, const unsigned long long *__restrict__ n_cand_p, unsigned long long cap, const VerifyArgs va, uint32_t lds_rows, const RowHist rows,
    float *__restrict__ sims, uint32_t *__restrict__ slots, uint32_t *__restrict__ row_hits, const TailWork tw) {
    // LDS: [template records: n_templates x 32 B][template rows, 16 B (MODE 1) or 12 B (MODE 2) each]
    extern __shared__ __attribute__((aligned(16))) v4i verify_lds[];
    __shared__ uint32_t wave_sum[16], wave_max[16];
    constexpr bool LDS = MODE == 1;
    VerifyMeta *meta = reinterpret_cast<VerifyMeta *>(verify_lds);
    v4i *needle_lds = verify_lds + 2 * va.n_templates;
    uint32_t *needle12 = reinterpret_cast<uint32_t *>(needle_lds);
    for (uint32_t i = threadIdx.x; i < 2 * va.n_templates; i += blockDim.x) verify_lds[i] = reinterpret_cast<const v4i *>(va.vmeta)[i];
    if (MODE == 1)
        for (uint32_t i = threadIdx.x; i < lds_rows; i += blockDim.x) needle_lds[i] = va.needles16[i];
    if (MODE == 2)
        for (uint32_t i = threadIdx.x; i < lds_rows; i += blockDim.x) {
            const v4i r = va.needles16[i];
            needle12[3 * i] = (uint32_t)r[0], needle12[3 * i + 1] = (uint32_t)r[1], needle12[3 * i + 2] = (uint32_t)r[2];
        }
    __syncthreads();
    const unsigned long long n = min(*n_cand_p, cap);
    const int lane = threadIdx.x & 63;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned long long first = (unsigned long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63u);
    uint64_t key_next = first + lane < n ? cand[first + lane] : 0;  // the next step's key is loaded a step ahead
    for (unsigned long long i0 = first; i0 < n; i0 += stride) {  // wave-uniform trip count
        const unsigned long long i = i0 + lane;
        const bool valid = i < n;
        const uint64_t key = key_next;
        if (i + stride < n) key_next = cand[i + stride];
        float sim = 0.f;
        const bool emit = valid && (MODE == 2 ? verify_candidate_narrow(key, va, needle12, meta, &sim) : verify_candidate_meta<LDS>(key, va, needle_lds, meta, &sim));
        // a hit's slot inside its bucket: the wave's 64 candidates come from a handful of page rows
        const uint32_t r = emit ? row_of_key(key, rows) : 0xffffffffu;
        uint32_t slot = 0xffffffffu;
        uint64_t todo = __builtin_amdgcn_ballot_w64(emit);
        while (todo) {
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)__builtin_ctzll(todo));
            const uint64_t peers = __builtin_amdgcn_ballot_w64(r == r0);
            uint32_t start = 0;
            if (lane == (int)__builtin_ctzll(peers)) start = atomicAdd(row_hits + r0, (uint32_t)__builtin_popcountll(peers));
            start = (uint32_t)__builtin_amdgcn_readlane((int)start, (int)__builtin_ctzll(peers));
            if (r == r0) slot = start + __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
            todo &= ~peers;
        }
        if (valid) {
            sims[i] = sim;
            slots[i] = slot;
        }
    }
    // the kernel's last workgroup turns the buckets' hit counts into their dense positions (+ the hit total and the largest bucket)
    if (last_workgroup(tw.done)) row_prefix_block(row_hits, tw.n_rows, tw.hbase, nullptr, tw.total_out, tw.max_out, wave_sum, wave_max);
}

// Banks whose verify operand does not fit the LDS whole (BASELINE configs[2]: 1 520 templates, 287 KB of 12-byte rows + 48 KB
// of records): the operand is cut into CHUNKS of consecutive templates (by global template index; d_vrows_t / d_vmeta_t are
// ordered that way) that do fit, and the kernel makes one pass over its candidates per chunk — chunk rows and records staged in
// LDS, every wave walking its own piece of the list, collecting the keys whose template lies in the chunk in a wave-private
// queue in LDS and verifying them 64 at a time, so every verify step runs on full waves whatever the mix of templates in the
// list.  Reading the keys once more per chunk costs 8 B per candidate and pass, so chunks are as large as the LDS allows (one
// workgroup per CU: BASELINE configs[2] in 3 chunks 2.19 ms alone on the chip; 5 chunks of 80 KB 3.14 ms; the template rows
// gathered from global memory, verify_list_kernel<0>, 3.35 ms — a 64-way gather per wave instruction).
constexpr uint32_t MAX_VERIFY_CHUNKS = 16, CHUNK_QUEUE = 128;
struct ChunkTable {
    uint32_t n;
    uint32_t t_lo[MAX_VERIFY_CHUNKS + 1];    // chunk k: templates [t_lo[k], t_lo[k + 1])
    uint32_t row_lo[MAX_VERIFY_CHUNKS + 1];  // ... rows [row_lo[k], row_lo[k + 1]) of d_vrows_t
    // LDS layout, per chunk: [its records: 32 B each][its rows]; behind the LARGEST chunk's data (data_bytes, a multiple of 16) the
    // waves' queues.  (Round 4 laid the rows out behind the largest chunk's RECORDS and sized the data for the largest record count
    // plus the largest row count — two maxima that can come from different chunks: a bank of many short templates followed by
    // tall ones asked for more LDS than a CU has although every chunk fitted.)
    uint32_t data_bytes;
};
template <int ROWB>
__global__ __launch_bounds__(VERIFY_THREADS, 4) void verify_chunks_kernel(
    const uint64_t *__restrict__ cand, const unsigned long long *__restrict__ n_cand_p, unsigned long long cap, const VerifyArgs va, const ChunkTable ct,
    const uint32_t *__restrict__ vrows_t, const VerifyMeta *__restrict__ vmeta_t, const RowHist rows, float *__restrict__ sims, uint32_t *__restrict__ slots,
    uint32_t *__restrict__ row_hits, const TailWork tw) {
    // LDS: [the chunk's records: 32 B each][the chunk's rows: ROWB each] ... at data_bytes: [per wave: CHUNK_QUEUE keys][per wave: CHUNK_QUEUE list positions]
    extern __shared__ __attribute__((aligned(16))) v4i verify_lds[];
    __shared__ uint32_t wave_sum[16], wave_max[16];
    constexpr uint32_t ROWDW = ROWB / 4;
    VerifyMeta *meta = reinterpret_cast<VerifyMeta *>(verify_lds);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint64_t *q_base = reinterpret_cast<uint64_t *>(verify_lds + ct.data_bytes / 16);
    uint64_t *q_key = q_base + (size_t)wv * CHUNK_QUEUE;
    uint32_t *q_pos = reinterpret_cast<uint32_t *>(q_base + (size_t)(VERIFY_THREADS / 64) * CHUNK_QUEUE) + (size_t)wv * CHUNK_QUEUE;
    const unsigned long long n = min(*n_cand_p, cap);
    // the wave's own piece of the list: whole groups of 64
    const unsigned long long n_waves = (unsigned long long)gridDim.x * (VERIFY_THREADS / 64);
    const unsigned long long per_wave = ((n + n_waves - 1) / n_waves + 63) / 64 * 64;
    const unsigned long long wb = min(n, ((unsigned long long)blockIdx.x * (VERIFY_THREADS / 64) + wv) * per_wave), we = min(n, wb + per_wave);
    // one verify step over the first `m` queued candidates (m <= 64): the reference arithmetic, the hit's slot in its bucket
    auto step = [&](uint32_t m, const VerifyMeta *meta_c, const uint32_t *rows_c) {
        const bool valid = (uint32_t)lane < m;
        const uint64_t key = valid ? q_key[lane] : 0;
        const uint32_t pos = valid ? q_pos[lane] : 0;
        float sim = 0.f;
        bool emit = false;
        if (valid) {
            if (ROWB == 12) emit = verify_candidate_narrow(key, va, rows_c, meta_c, &sim);
            else emit = verify_candidate_meta<true>(key, va, reinterpret_cast<const v4i *>(rows_c), meta_c, &sim);
        }
        const uint32_t r = emit ? row_of_key(key, rows) : 0xffffffffu;
        uint32_t slot = 0xffffffffu;
        uint64_t todo = __builtin_amdgcn_ballot_w64(emit);
        while (todo) {
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)__builtin_ctzll(todo));
            const uint64_t peers = __builtin_amdgcn_ballot_w64(r == r0);
            uint32_t start = 0;
            if (lane == (int)__builtin_ctzll(peers)) start = atomicAdd(row_hits + r0, (uint32_t)__builtin_popcountll(peers));
            start = (uint32_t)__builtin_amdgcn_readlane((int)start, (int)__builtin_ctzll(peers));
            if (r == r0) slot = start + __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
            todo &= ~peers;
        }
        if (valid && pos < n) {  // (pos comes out of the wave's own queue: always below n — the test costs nothing and a write never leaves the arrays)
            sims[pos] = sim;
            slots[pos] = slot;
        }
    };
    for (uint32_t k = 0; k < ct.n; k++) {
        const uint32_t t_lo = ct.t_lo[k], t_hi = ct.t_lo[k + 1], row_lo = ct.row_lo[k], n_rows_c = ct.row_lo[k + 1] - row_lo;
        uint32_t *rows_lds = reinterpret_cast<uint32_t *>(verify_lds + 2 * (t_hi - t_lo));  // behind THIS chunk's records
        __syncthreads();  // the previous chunk's readers are done
        for (uint32_t i = threadIdx.x; i < 2 * (t_hi - t_lo); i += VERIFY_THREADS) verify_lds[i] = reinterpret_cast<const v4i *>(vmeta_t + t_lo)[i];
        for (uint32_t i = threadIdx.x; i < n_rows_c * ROWDW; i += VERIFY_THREADS) rows_lds[i] = vrows_t[(size_t)row_lo * ROWDW + i];
        __syncthreads();
        // records by global template index and rows by their index in d_vrows_t, as the verify functions address them
        const VerifyMeta *meta_c = meta - t_lo;
        const uint32_t *rows_c = rows_lds - (size_t)row_lo * ROWDW;
        uint32_t count = 0;  // wave-uniform: queued candidates of this chunk
        for (unsigned long long i0 = wb; i0 < we; i0 += 4 * 64) {
            // four groups of keys per round, their loads in flight together (the walk is latency-bound otherwise: one round trip per group and pass)
            uint64_t key4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned long long i = i0 + 64 * q + lane;
                key4[q] = i < we ? cand[i] : ~0ull;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned long long i = i0 + 64 * q + lane;
                const bool valid = i < we;
                const uint64_t key = key4[q];
                const uint32_t t = va.fmt.t(key);
                // a template index outside the bank belongs to no chunk: such a key (cannot happen) is given to the last chunk, whose
                // verify flags it instead of leaving its outputs unwritten
                const bool in = valid && ((t >= t_lo && t < t_hi) || (k + 1 == ct.n && t >= t_hi));
                const uint64_t mask = __builtin_amdgcn_ballot_w64(in);
                if (mask) {
                    if (in) {
                        const uint32_t p = count + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                        q_key[p] = key;
                        q_pos[p] = (uint32_t)i;
                    }
                    count += (uint32_t)__builtin_popcountll(mask);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (count >= 64) {
                        step(64, meta_c, rows_c);
                        const uint32_t rest = count - 64;  // < 64: moves to the front of the queue
                        const uint64_t kk = (uint32_t)lane < rest ? q_key[64 + lane] : 0;
                        const uint32_t pp = (uint32_t)lane < rest ? q_pos[64 + lane] : 0;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        if ((uint32_t)lane < rest) {
                            q_key[lane] = kk;
                            q_pos[lane] = pp;
                        }
                        count = rest;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    }
                }
            }
        }
        if (count) step(count, meta_c, rows_c);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (last_workgroup(tw.done)) row_prefix_block(row_hits, tw.n_rows, tw.hbase, nullptr, tw.total_out, tw.max_out, wave_sum, wave_max);
}

// hit -> its dense place: hbase[bucket] + slot (no atomics: the slots were handed out by the verify)
__global__ __launch_bounds__(256) void hit_scatter_kernel(const uint64_t *__restrict__ cand, const unsigned long long *__restrict__ n_cand_p, unsigned long long cap,
                                                          const RowHist rows, const uint32_t *__restrict__ hbase, const float *__restrict__ sims,
                                                          const uint32_t *__restrict__ slots, uint64_t *__restrict__ hkeys, float *__restrict__ hsims,
                                                          unsigned long long hit_cap) {
    const unsigned long long n = min(*n_cand_p, cap);
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t slot = slots[i];
        if (slot == 0xffffffffu) continue;
        const uint64_t key = cand[i];
        const unsigned long long pos = (unsigned long long)hbase[row_of_key(key, rows)] + slot;
        if (pos < hit_cap) {  // estimated sizes: a hit count above its bound is flagged by record_scan_sizes and the batch redone
            hkeys[pos] = key;
            hsims[pos] = sims[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side

// x-segments per page row: a bucket should hold what one wave sorts in registers (<= 1024 candidates; up to 4096 go through
// the second, slower launch).  Candidates per row grow with the row's windows x templates: the first scan of a setup takes one
// segment per <= 2^19 of them (BASELINE configs[1]: 608 x 380 -> one segment, a row or two per batch just above 1024;
// configs[2]: 1200 x 1520 -> 5 segments of 256 px), later scans halve the segments while the largest bucket stays above 2048
// (ctx.hip, finish_results: configs[2] settles at 128 px).  The segmentation never changes a result.
void row_segments(const focr_ctx *c, uint32_t *seg_shift, uint32_t *n_seg) {
    uint32_t sh = c->row_seg_shift;
    if (!sh) {
        while (((size_t)1 << sh) < c->r_w) sh++;  // one segment
        while (sh > 5 && ((size_t)1 << sh) * c->n_templates > ((size_t)1 << 19)) sh--;
    }
    *seg_shift = sh;
    *n_seg = (uint32_t)((c->r_w + ((size_t)1 << sh) - 1) >> sh);
}

static size_t row_buckets(const focr_ctx *c) {
    uint32_t sh, ns;
    row_segments(c, &sh, &ns);
    return c->sub_np * c->r_h * ns;
}

bool rows_applicable(const focr_ctx *c) {
    if (c->tail_mode == 0) return false;  // focr_ctx_set_row_tail(0): the legacy tail, for A/B
    for (const SizeClass &sc : c->classes)
        if (sc.tall) return false;  // scan_tall_kernel appends its candidates without counting them per bucket
    static_assert(sizeof(VerifyMeta) == 32, "VerifyMeta is staged in LDS as two 16-byte words per template");
    return row_buckets(c) <= ((size_t)1 << 22) && c->fmt.bp + c->fmt.by <= 32 && c->n_templates <= 4096;  // 4096 x 32 B of template records in the verify's LDS
}

// hits-first tail, before the scan kernels: zeroed hit counters per bucket; the scan kernels count nothing (row_hist.cnt = null)
int rows2_begin(focr_ctx *c, ClearList &clear) {
    const size_t n_rows = row_buckets(c);
    const size_t padded = (n_rows + 1 + 3) / 4 * 4 + 4;  // row_prefix_kernel moves 16 bytes at a time
    uint32_t *hits = (uint32_t *)c->rows_hits.ensure(c, padded * 4);
    if (!hits || !c->rows_hbase.ensure(c, padded * 4)) return fail(c, FOCR_ERR_NOMEM, "rows: hipMalloc failed");
    uint32_t *big = (uint32_t *)c->rows_big.ensure(c, ((size_t)n_rows + 1) * 4 + 8);  // [0]: length of the list of large buckets
    if (!big) return fail(c, FOCR_ERR_NOMEM, "rows: hipMalloc failed");
    if (!clear.add(hits, padded * 4)) return fail(c, FOCR_ERR_INVALID, "rows: clear list full or region too large");
    if (!clear.add(big, 8)) return fail(c, FOCR_ERR_INVALID, "rows: clear list full or region too large");
    uint32_t seg_shift, n_seg;
    row_segments(c, &seg_shift, &n_seg);
    c->row_hist = RowHist{nullptr, (uint32_t)c->r_h, c->fmt.bt + c->fmt.bx, c->fmt.by, (uint32_t)c->sub_p0, c->fmt.bt, c->fmt.bx, seg_shift, n_seg};
    return FOCR_OK;
}

uint32_t rows_capacity_for(uint64_t row_max) { return row_max <= 4096 ? 4096u : 0u; }  // rows above 1024 take the second sort launch

// test hook (focr_debug_set_tail_grid): the persistent tail kernels' grids scaled by num / den — results must not depend on a grid
static unsigned tail_grid(const focr_ctx *c, unsigned blocks) {
    if (!c->dbg_grid_num || !c->dbg_grid_den) return blocks;
    return (unsigned)std::max<uint64_t>(1, (uint64_t)blocks * c->dbg_grid_num / c->dbg_grid_den);
}

// The verify operand's place for this bank: 2 = 12-byte rows in LDS, two workgroups per CU; 1 = 16-byte rows in LDS; 0 = global memory
static int verify_mode(const focr_ctx *c, size_t *lds, uint32_t *rows_out) {
    size_t all_rows = 0;
    uint32_t max_w = 0;
    for (const TemplateConst &tc : c->h_tconst) {
        all_rows += (size_t)tc.n_h * (tc.n_w > 16 ? 2u : 1u);
        max_w = std::max<uint32_t>(max_w, tc.n_w);
    }
    const size_t meta_bytes = c->n_templates * sizeof(VerifyMeta);  // <= 4096 templates here: 128 KiB at most
    const bool narrow = max_w <= 12 && meta_bytes + all_rows * 12 <= ((size_t)80 << 10) - 256;
    const bool in_lds = meta_bytes + all_rows * 16 <= ((size_t)144 << 10);
    *lds = meta_bytes + (narrow ? all_rows * 12 : in_lds ? all_rows * 16 : 0);
    *rows_out = (uint32_t)all_rows;
    return narrow ? 2 : in_lds ? 1 : 0;
}

// hits-first tail, phase 1 (right behind the scan kernels): exact verify of the candidate list in flush order, hit counts and
// slots per bucket, prefix -> d_res[6] = hits, d_res[5] = the largest bucket.  Records ev[3] behind the verify.
int rows2_verify(focr_ctx *c, double thr_d, const unsigned long long *n_cand_p, size_t ub_c) {
    const uint32_t n_rows = (uint32_t)row_buckets(c);
    float *csims = (float *)c->scan_pos.ensure(c, (ub_c + 1) * 4);
    uint32_t *cslots = (uint32_t *)c->scan_flags.ensure(c, (ub_c + 1) * 4);
    if (!csims || !cslots) return fail(c, FOCR_ERR_NOMEM, "rows: hipMalloc failed");
    const unsigned cus = c->n_cus;
    uint32_t *hits = (uint32_t *)c->rows_hits.p, *hbase = (uint32_t *)c->rows_hbase.p;
    const TailWork tw{c->d_counter + TAIL_DONE_WORD, n_rows, hbase, c->d_res + 6, c->d_res + 5};  // the verify's last workgroup does the prefix
    // The verify's workgroups are persistent (they walk the list with the grid's stride), so where the scan kernel is confined to
    // scan_cus CUs — several batches in flight — the verify asks for no more workgroups than fit the CUs the scan leaves free: a
    // verify launched in the gap between two scan launches would otherwise put a workgroup on every CU of the chip and hold it
    // until the kernel ends, and the next scan launch's workgroups wait for whole CUs (C2, three in flight, one box, alternating:
    // 512 workgroups 32.55 Gpx/s · 64: 32.82 · 32: 31.2 with the scan launch at 1.57 instead of 1.68 ms, but the lane then waits
    // for its verify).  The whole-bank-in-LDS forms only: the chunked verify of large banks keeps the chip (below).
    // ... unless the host has said that nothing follows this batch (focr_pipe_announce_last / _end_of_stream) — nobody will scan behind it while its
    // tail runs: then the tail takes the chip (the last batch of a run: a 20-step timed region ends 0.7-1.2 ms earlier)
    const bool nobody_behind = c->tail_full_chip;
    const unsigned vcus = (c->scan_cus && c->scan_cus < cus && !nobody_behind) ? std::max(cus - c->scan_cus, cus / 16) : cus;
    if (ub_c) {
        const VerifyArgs va = verify_args(c, thr_d);
        size_t lds;
        uint32_t all_rows;
        const int mode = verify_mode(c, &lds, &all_rows);
        if (mode == 0 && c->vrow_bytes && c->chunked_verify) {
            // the operand does not fit the LDS whole: chunks of consecutive templates that do (verify_chunks_kernel)
            const bool narrow = c->vrow_bytes == 12;
            const size_t queue_bytes = (size_t)(VERIFY_THREADS / 64) * CHUNK_QUEUE * 12 + 64;
            const size_t budget = ((size_t)150 << 10) - queue_bytes;  // one workgroup per CU: fewer, larger chunks beat more waves per CU (above)
            ChunkTable ct{};
            size_t bytes = 0;
            uint32_t t0 = 0;
            bool ok = true;
            for (uint32_t t = 0; t <= c->n_templates && ok; t++) {
                const size_t add = t < c->n_templates ? sizeof(VerifyMeta) + (size_t)(c->h_vrow0_t[t + 1] - c->h_vrow0_t[t]) * c->vrow_bytes : 0;
                if (t == c->n_templates || bytes + add > budget) {
                    if (t == t0 || ct.n == MAX_VERIFY_CHUNKS) {
                        ok = false;  // a single template above the budget, or too many chunks: global loads after all
                        break;
                    }
                    ct.t_lo[ct.n] = t0;
                    ct.row_lo[ct.n] = c->h_vrow0_t[t0];
                    ct.data_bytes = std::max(ct.data_bytes, (uint32_t)((bytes + 15) & ~(size_t)15));  // this chunk's records + rows (bytes <= budget)
                    ct.n++;
                    t0 = t;
                    bytes = 0;
                }
                bytes += add;
            }
            if (ok) {
                ct.t_lo[ct.n] = (uint32_t)c->n_templates;
                ct.row_lo[ct.n] = c->h_vrow0_t[c->n_templates];
                const size_t lds_c = (size_t)ct.data_bytes + queue_bytes;  // <= 150 KiB by construction: every chunk's data is within the budget
                // one workgroup per CU of the whole chip: at configs[2] the chunk passes are a fifth of a lane's chain, and the lane
                // waits for them (half / a third / a quarter of the CUs: 7.25 / 7.21 / 7.12 Gpx/s against 7.28)
                const unsigned nbc = tail_grid(c, (unsigned)std::max<size_t>(1, std::min<size_t>((ub_c + VERIFY_THREADS - 1) / VERIFY_THREADS, (size_t)cus)));
                hipError_t attr = hipSuccess;
#define FOCR_VERIFY_CHUNKS(R)                                                                                                                                  \
    attr = hipFuncSetAttribute(reinterpret_cast<const void *>(verify_chunks_kernel<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);                \
    if (attr == hipSuccess)                                                                                                                                    \
    hipLaunchKernelGGL(verify_chunks_kernel<R>, dim3(nbc), dim3(VERIFY_THREADS), lds_c, c->stream, (const uint64_t *)c->d_cand, n_cand_p, (unsigned long long)ub_c, va, \
                       ct, (const uint32_t *)c->d_vrows_t, (const VerifyMeta *)c->d_vmeta_t, c->row_hist, csims, cslots, hits, tw)
                if (narrow) {
                    FOCR_VERIFY_CHUNKS(12);
                } else {
                    FOCR_VERIFY_CHUNKS(16);
                }
#undef FOCR_VERIFY_CHUNKS
                if (attr == hipSuccess) {
                    FOCR_HIP(c, hipGetLastError());
                    goto verified;
                }
                (void)hipGetLastError();  // the device refuses that much LDS: the rows come from global memory instead (below)
            }
        }
        const unsigned per_cu = mode == 2 ? 2u : (mode == 0 && c->n_templates * sizeof(VerifyMeta) <= ((size_t)64 << 10) ? 2u : 1u);
        const unsigned nb = tail_grid(c, (unsigned)std::max<size_t>(1, std::min<size_t>((ub_c + VERIFY_THREADS - 1) / VERIFY_THREADS, (size_t)vcus * per_cu)));
#define FOCR_VERIFY_LIST(M)                                                                                                                                      \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(verify_list_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
    hipLaunchKernelGGL(verify_list_kernel<M>, dim3(nb), dim3(VERIFY_THREADS), lds, c->stream, (const uint64_t *)c->d_cand, n_cand_p, (unsigned long long)ub_c, va, \
                       M == 0 ? 0u : all_rows, c->row_hist, csims, cslots, hits, tw)
        if (mode == 2) {
            FOCR_VERIFY_LIST(2);
        } else if (mode == 1) {
            FOCR_VERIFY_LIST(1);
        } else {
            FOCR_VERIFY_LIST(0);
        }
#undef FOCR_VERIFY_LIST
        FOCR_HIP(c, hipGetLastError());
    }
verified:
    FOCR_HIP(c, hipEventRecord(c->ev[3], c->stream));
    if (!ub_c) {  // no candidate can exist (nothing was verified): the prefix of all-zero counts, as a launch of its own
        hipLaunchKernelGGL(row_prefix_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint32_t *)hits, n_rows, hbase, (uint32_t *)nullptr, c->d_res + 6, c->d_res + 5);
        FOCR_HIP(c, hipGetLastError());
    }
    return FOCR_OK;
}

// hits-first tail, phase 2: hits to their buckets, every bucket sorted with its similarities -> the dense sorted hits in
// d_hit_keys / d_hit_sims_alt (their number: d_res[6]).  ub_h: bound on the hits (exact sizes: the count itself).
int rows2_place(focr_ctx *c, const unsigned long long *n_cand_p, size_t ub_c, size_t ub_h, bool big_expected, bool sort) {
    const uint32_t n_rows = (uint32_t)row_buckets(c);
    int rc;
    if ((rc = ensure_hit_capacity(c, std::max<size_t>(c->hit_capacity, ub_h + 1)))) return rc;
    const unsigned cus = c->n_cus;
    const uint32_t *hits = (const uint32_t *)c->rows_hits.p, *hbase = (const uint32_t *)c->rows_hbase.p;
    if (ub_c) {
        const unsigned nb = tail_grid(c, (unsigned)std::max<size_t>(1, std::min<size_t>((ub_c + 255) / 256, (size_t)cus * 16)));
        hipLaunchKernelGGL(hit_scatter_kernel, dim3(nb), dim3(256), 0, c->stream, (const uint64_t *)c->d_cand, n_cand_p, (unsigned long long)ub_c, c->row_hist, hbase,
                           (const float *)c->scan_pos.p, (const uint32_t *)c->scan_flags.p, c->d_hit_keys, c->d_hit_sims_alt, (unsigned long long)c->hit_capacity);
        FOCR_HIP(c, hipGetLastError());
    }
    if (!sort) return FOCR_OK;  // a bucket beyond the row sort's capacity (exact sizes know): the caller sorts the placed hits with the library sort
    unsigned long long *flags_word = (unsigned long long *)(c->d_res + 4);
    const unsigned row_blocks = tail_grid(c, (unsigned)std::max<size_t>(1, std::min<size_t>(((size_t)n_rows + 3) / 4, (size_t)cus * 8)));
    const uint32_t seg_w = 1u << c->row_hist.seg_shift;
    uint32_t xs = 0;
    while ((seg_w >> xs) > XBINS) xs++;
    const uint32_t n_bins = seg_w >> xs;
    uint32_t *big = (uint32_t *)c->rows_big.p;  // allocated and zeroed in rows2_begin
    auto k1 = row_sort_kernel<1024, 4, false, true>;
    auto k2 = row_sort_kernel<4096, 1, true, true>;
    const size_t lds1 = (size_t)4 * (XBINS + 1 + 1024) * 4, lds2 = (size_t)(XBINS + 1 + 4096) * 4;
    hipLaunchKernelGGL(k1, dim3(row_blocks), dim3(256), lds1, c->stream, n_rows, hbase, hits, c->d_hit_keys, c->fmt.bt + c->fmt.bx, c->fmt.bt, seg_w - 1, xs, n_bins,
                       big_expected ? big : (uint32_t *)nullptr, flags_word, c->d_hit_sims_alt, (unsigned long long)c->hit_capacity);
    FOCR_HIP(c, hipGetLastError());
    // Buckets above 1 024 hits (the list `big`) get a second launch only where one is expected: exact sizes know the largest bucket,
    // estimated sizes go by the previous scan's (+ 25 %).  Without the launch a bucket that lands on the list after all is an
    // overflow like any other estimate that proved too small (flag bit 1: the batch is redone with exact sizes) — even a launch of
    // ONE wave that finds the list empty stood 110 us in a lane's chain of kernels (it needs 20 KB of LDS on a CU the other
    // batches' kernels keep full: profiles/r04_timeline_bench_c2.log), for a list that BASELINE configs[1] and [2] never fill.
    if (big_expected) {
        hipLaunchKernelGGL(k2, dim3(cus), dim3(64), lds2, c->stream, n_rows, hbase, hits, c->d_hit_keys, c->fmt.bt + c->fmt.bx, c->fmt.bt, seg_w - 1, xs, n_bins, big, flags_word,
                           c->d_hit_sims_alt, (unsigned long long)c->hit_capacity);
        FOCR_HIP(c, hipGetLastError());
    }
    return FOCR_OK;
}

}  // namespace focr
