// mfma_common.h — declarations shared by the MFMA prefilter kernels (scan_mfma.hip, scan_mfma2.hip): threshold planes,
// item queues, K layouts.
#pragma once
#include "common.h"

namespace focr {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int32_t REJECT = 0x3fffffff;  // |negL| of a window the reference never emits
constexpr uint32_t WBUF = 64;           // wave-private candidate staging entries in LDS (no atomics on the way in)

// one global atomic per flush: lane 0 reserves `count` slots, the wave copies its staged keys out coalesced
__device__ __forceinline__ void flush_wave_candidates(uint64_t *wbuf, uint32_t count, int lane, uint64_t *__restrict__ cand,
                                                      unsigned long long *__restrict__ cand_counter, unsigned long long cand_cap, const RowHist &rows) {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(cand_counter, (unsigned long long)count);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    base = ((unsigned long long)hi << 32) | lo;
    const bool valid = (uint32_t)lane < count;
    const uint64_t key = valid ? wbuf[lane] : 0;
    if (valid && base + lane < cand_cap) cand[base + lane] = key;
    if (rows.cnt) {  // row counts (rows.hip): one atomic per distinct row among the flushed keys, not one per key
        const uint32_t r = valid ? row_of_key(key, rows) : 0xffffffffu;
        uint64_t todo = __builtin_amdgcn_ballot_w64(valid);
        while (todo) {
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)__builtin_ctzll(todo));
            const uint64_t peers = __builtin_amdgcn_ballot_w64(r == r0);
            if (lane == (int)__builtin_ctzll(peers)) atomicAdd(rows.cnt + r0, (uint32_t)__builtin_popcountll(peers));
            todo &= ~peers;
        }
    }
}

// ---- threshold planes ------------------------------------------------------------------------
// The statistics kernel turns every window of a size class into ONE number, the prefilter threshold L of that window, and
// stores it as a 16-bit integer in units of a per-class power of two S ("threshold plane", 2 B per window and class):
//
//     L(w) = kappa * norm_p(w)  -  c * rho_max * dnorm(w)            candidate  <=>  G(w, t) > L(w)
//
// kappa = c*thr - e_max - margins (scan_mfma.hip header).  The second term exists for classes whose LAST COLUMN the MFMA
// does not multiply ("column drop": a 9-wide template costs 3 K-steps of 64 bytes for 135 taps; its first 8 columns fit 2):
// with m = mean of the window over the kept columns, beta = the unit mean-centred template, sigma = sum of beta over the
// dropped column, and beta'_k = beta_k + sigma / n_keep on the kept columns (so that sum beta' = 0),
//     sum_all a_k beta_k = sum_keep a_k beta'_k + sum_drop (a_j - m) beta_j          (exact)
// the MFMA evaluates the first sum with the int8 rounding of c*beta' (error <= e_max * norm_keep <= e_max * norm_p: the kept
// box's norm about its own mean never exceeds the full box's), and Cauchy-Schwarz bounds the second by
// dnorm(w) * rho_t,  dnorm = |a_drop - m|_2,  rho_t = |beta_drop|_2 <= rho_max.  Hence sim > thr  =>  G > L: no false
// negatives, for either sign of kappa.  Everything up to the square roots is exact integer arithmetic:
//     V = n*s2 - s^2 = n * norm_p^2                                  (full box; V > 0 <=> the reference's rnorm is finite)
//     W = n_k^2*q2 - 2*n_k*s_k*q1 + D*s_k^2 = n_k^2 * dnorm^2        (q1, q2: sums over the dropped column, D its taps;
//                                                                      evaluated as an f32 upper bound: dropped_column_W_upper)
// then L = kq*sqrt(V) - crk*sqrt(W) in f32 with kq rounded towards -inf and crk up (host: plane_params) — f32 errors stay
// below 1 for |L| < 4e6 and are absorbed by a "- 2" — and the stored value is the C-in itself in units of S:
//     nq = -floor((L - 2) / S)  as int16,     C-in = nq * S = nq << log2(S),     D = G + C-in > 0  <=>  G > floor((L - 2) / S) * S
// a threshold rounded TOWARDS -INF (a lower threshold only admits more candidates), whatever the signs; -32768 where the reference
// never emits (x = 0, y = 0, window outside the page, zero variance): S >= K / 2 for K bytes per window (plane_params), so that
// 32768 * S exceeds every |G| <= K * 127 * 128 — the pair never passes.  Round 2 stored the window norm (rounded towards zero)
// and multiplied by kappa in the scan kernel, which raised the threshold for kappa < 0; rounds 3-4 stored L as f16 rounded
// towards -inf (the same directed rounding, but five vector instructions per window and class in the scan kernel's item prologue
// to turn it into an integer: convert, multiply, floor, clamp, convert).  The integer plane needs ONE (a shift): S = 64 at
// BASELINE configs[1], i.e. thresholds of 1e5 .. 6e5 in steps of 64 — as fine as f16's 11 bits there.
struct PlaneParams {
    float kq;        // kappa / sqrt(n), towards -inf
    float crk;       // c * rho_max * (1 + 1e-4) / n_keep, upwards (0: nothing dropped)
    float S;         // power of two, >= K / 2: |L - 2| / S <= 16384 for every window whose |L| is below 2^28
    float inv_S;     // 1 / S (exact)
    uint32_t shift;  // log2(S), 5 .. 14
};
constexpr int16_t PLANE_NEVER = -32768;  // the reference never emits at this window: an unreachable threshold

// f32 square root: the raw 1-ulp instruction on the device, the correctly rounded one in the host model
__host__ __device__ inline float sqrt_fast(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return __builtin_sqrtf(x);
#endif
}

// L of one window in f32 from its exact integer statistics (see above).  V, W as floats (conversion error 2^-24 relative).
__host__ __device__ inline float threshold_f32(const PlaneParams &p, float Vf, float Wf) {
    return __builtin_fmaf(-p.crk, sqrt_fast(Wf), p.kq * sqrt_fast(Vf));
}
// the same for a class without a dropped column (crk = 0, W = 0): fma(-0, 0, x) = x, bit for bit — two instructions less
__host__ __device__ inline float threshold_f32_nodrop(const PlaneParams &p, float Vf) { return p.kq * sqrt_fast(Vf); }
// An UPPER bound of W = n_k^2 * dnorm^2 = n_k^2 q2 + D s_k^2 - 2 n_k s_k q1 in f32 (the exact value needs 64-bit integer
// multiplies, quarter-rate instructions on the vector unit; this is five full-rate ones).  Every factor is an integer below
// 2^24, so each of the three products carries at most two roundings and the sum three more: |error| < 7 * 2^-24 * (t1 + t2 + t3)
// — the 2^-21 * (t1 + t2 + t3) added on top makes the result >= W (>= 0) always.  An over-estimated dnorm only lowers the
// threshold: by sqrt(2^-21 * 3 n_k^2 q2) / n_k * c * rho, a few units of L on thresholds of 10^5.
__host__ __device__ inline float dropped_column_W_upper(uint32_t n_k, uint32_t D, uint32_t s_k, uint32_t q1, uint32_t q2) {
    const float sk = (float)s_k;
    const float t1 = (float)(n_k * n_k) * (float)q2, t3 = (float)D * (sk * sk), t2 = (float)(2 * n_k) * (sk * (float)q1);
    return __builtin_fmaf(t1 + t2 + t3, 0x1p-21f, (t1 + t3) - t2);
}
// legacy int32 table entry (scan_mfma2_kernel): -(floor(L) - 2)
__host__ __device__ inline int32_t threshold_negL(float Lf) {
    float f = __builtin_floorf(Lf) - 2.0f;
    f = __builtin_fminf(__builtin_fmaxf(f, -1.0e9f), 1.0e9f);
    return -(int32_t)f;
}
// Plane value of a window that can emit: nq = -floor((L - 2) / S), clamped to +-32767 (beyond that — a --threshold of +-1e30 —
// the clamp does the right thing by itself: a threshold above 32767 * S is unreachable either way, one below -32767 * S passes
// every pair either way; NaN cannot occur: kq and crk are finite, V and W finite and >= 0).  The same arithmetic on the host
// (the model) and on the device (exact scaling by a power of two, floor, an exact conversion): bit-identical by construction,
// checked by tests/test_gpu_parity.py::test_threshold_plane_values_device_equals_host.
__host__ __device__ inline int16_t plane_value(const PlaneParams &p, float Lf) {
    // (L - 2) / S in one fused step: S is a power of two, so the product and the constant are exact and the single rounding is that
    // of L - 2 scaled — bit-identical to (Lf - 2.0f) * inv_S, one instruction less
    // ... and the sign goes into the same step: -floor(t) = ceil(-t), and fma(L, -1/S, 2/S) = -fma(L, 1/S, -2/S) bit for bit (rounding to
    // nearest is symmetric), so the value is ceil of one fma; the clamp is one integer median (the conversion saturates at
    // +-2^31 by itself, on the device as in the host's std::clamp below): four instructions per window and class, not six
    const float t = __builtin_ceilf(__builtin_fmaf(Lf, -p.inv_S, 2.0f * p.inv_S));
#if defined(__HIP_DEVICE_COMPILE__)
    const int q = (int)t;  // (v_cvt_i32_f32 saturates)
    return (int16_t)(q < -32767 ? -32767 : q > 32767 ? 32767 : q);  // v_med3_i32
#else
    const float c = t < -32767.0f ? -32767.0f : t > 32767.0f ? 32767.0f : t;
    return (int16_t)(int)c;
#endif
}
// C-in of one window from its plane value: nq * S (|C-in| <= 2^29: G + C-in cannot wrap).  One shift.
__host__ __device__ inline int prefilter_cin(uint32_t shift, int16_t nq) { return (int)((uint32_t)(int)nq << shift); }

// How the waves of a persistent scan kernel get their work.  Items (MT consecutive live M-tiles) are split into one
// contiguous range per XCD: workgroups b and b + 8 share an XCD and its L2, and neighbouring M-tiles share page rows.  Inside
// a range the waves TAKE items from a queue instead of owning a fixed stride: a workgroup needs an empty CU (all of its LDS,
// all 512 VGPRs of each SIMD), so with other contexts' kernels on the GPU the workgroups of one launch start up to a few
// 100 us apart, and the time per item varies with the candidates it finds; under a fixed split the launch ends when its
// unluckiest wave does (BASELINE configs[1], alone on the GPU: 2.20 -> 1.83 ms).
// A ticket (one agent-scope atomic on the range's counter, requested while the wave still works on the last item it holds,
// so that its latency is never waited for) grants ITEMS_PER_TICKET consecutive items: one counter word serves about 88
// returning atomics per microsecond (MI355X_MICROARCH.md, "dequeue"), and BASELINE configs[1] has 87 000 items per XCD —
// with one item per ticket the queue itself set a floor of ~1 ms under the launch (round 2: 1.26 ms for a 6-N-tile bank whose
// MFMAs take 0.3 ms).  A wave whose range is exhausted goes on with the next XCD's; it stops once it has seen every range
// exhausted, which every wave reaches after at most n_xc extra requests.
// The queue (QUEUE_XCDS counters, QUEUE_STRIDE dwords apart) is zeroed by the clear launch that opens every scan (launch_scan_mfma, ClearList).
constexpr uint32_t ITEMS_PER_TICKET = 4;  // 2 / 4 / 8 measured: 0.73 / 0.70 / 0.70 ms for a 6-N-tile bank
struct ItemTaker {
    uint32_t *queue;
    uint32_t n_items, n_xc, per_xc, cur_q, hops, ticket_v;
    uint32_t b_cur, b_end;  // items of the current ticket not handed out yet (wave-uniform)
    bool want;              // the current ticket's last item has been handed out: request() asks for the next ticket
    int lane;
    __device__ __forceinline__ void ask() {
        if (lane == 0) ticket_v = __hip_atomic_fetch_add(queue + cur_q * QUEUE_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // call once per item, after the item's loads have been issued: asks for the next ticket when this was the ticket's last item
    __device__ __forceinline__ void request() {
        if (want) {
            want = false;
            ask();
        }
    }
    __device__ __forceinline__ void init(uint32_t *q, uint32_t items, int lane_) {
        queue = q;
        n_items = items;
        lane = lane_;
        n_xc = min(QUEUE_XCDS, gridDim.x);  // small launches have fewer workgroups than XCDs
        per_xc = (n_items + n_xc - 1) / n_xc;
        cur_q = blockIdx.x % n_xc;
        hops = 0;
        ticket_v = 0;
        b_cur = b_end = 0;
        want = false;
        ask();
    }
    // the wave's next item (wave-uniform), or false when there is none left
    __device__ __forceinline__ bool next(uint32_t &item) {
        if (b_cur < b_end) {
            item = b_cur++;
            want = b_cur == b_end;
            return true;
        }
        for (;;) {
            const uint32_t ticket = __builtin_amdgcn_readfirstlane(ticket_v);
            const uint32_t qb = cur_q * per_xc, qe = min(n_items, qb + per_xc);
            // ticket * ITEMS_PER_TICKET cannot wrap: tickets <= items / ITEMS_PER_TICKET + waves of the launch
            if (qb < qe && ticket < (qe - qb + ITEMS_PER_TICKET - 1) / ITEMS_PER_TICKET) {
                b_cur = qb + ticket * ITEMS_PER_TICKET;
                b_end = min(qe, b_cur + ITEMS_PER_TICKET);
                item = b_cur++;
                want = b_cur == b_end;
                return true;
            }
            if (++hops >= n_xc) return false;
            cur_q = cur_q + 1 == n_xc ? 0 : cur_q + 1;
            ask();
        }
    }
};

// K layouts.  One MFMA K-step consumes 64 bytes of a window = four 16-byte k-groups, one per lane group
// g = lane>>4.  A k-group is built from whole dwords of image rows, so it lands in the operand registers
// straight from (byte-unaligned) global loads with no byte shuffling:
//   LAYOUT_W16 (n_w 13..16): group q = row q, columns 0..15.                 K-step ks, lane group g: q = 4ks+g
//   LAYOUT_W8  (n_w <= 8)  : group q = rows 2q, 2q+1, columns 0..7 each.     q = 4ks+g
//   LAYOUT_W12 (n_w 9..12) : rows are 12 bytes = 3 dwords; a quad of rows (4m..4m+3) fills exactly three groups, one per
//     dword COLUMN c = 0, 1, 2: group (m, c) = dword c (columns 4c..4c+3) of rows 4m, 4m+1, 4m+2, 4m+3.
//     (K-step, lane group) of group (m, c): ks = 3*(m/4) + c, g = m%4 — a K-step is a 4-column strip of 16 rows, so the
//     last K-step of a 16-row block holds nothing but zeros for templates of kept width <= 8 and the kernels skip its MFMAs for
//     such a size class (PlaneArgs::seg_full).  A lane loads its 4 rows once (12 contiguous bytes each) whatever the order.
//     16 rows -> 3 K-steps (192 bytes) instead of 4.
// Template bytes outside n_w x n_h are zero, so the extra image bytes the A side picks up do not matter.
enum { LAYOUT_W16 = 1, LAYOUT_W8 = 2, LAYOUT_W12 = 3 };

// (row j, column x) of a template -> (K-step, lane group, byte in the 16-byte group)
__host__ __device__ inline void kgroup_of(uint32_t layout, uint32_t j, uint32_t x, uint32_t *ks, uint32_t *g, uint32_t *byte) {
    if (layout == LAYOUT_W16) {
        *ks = j / 4, *g = j % 4, *byte = x;
    } else if (layout == LAYOUT_W8) {
        const uint32_t q = j / 2;
        *ks = q / 4, *g = q % 4, *byte = 8 * (j % 2) + x;
    } else {
        const uint32_t m = j / 4;
        *ks = 3 * (m / 4) + x / 4, *g = m % 4, *byte = 4 * (j % 4) + x % 4;
    }
}

// One kernel pass = one bank chunk (contiguous N-tiles of a super-class) made of up to MAX_SEGS segments,
// each segment being the tiles of one size class (its own negL table).
constexpr int MAX_SEGS = 8;
struct MfmaSeg {
    const int32_t *negL;  // class's C-in table [page][Lrows][Lpitch]
    uint32_t tile_end;    // one past the segment's last N-tile, in chunk-local numbering (segments are contiguous)
    uint32_t pad;
};
struct MfmaSegs {
    MfmaSeg s[MAX_SEGS];
    uint32_t n;
};

struct MfmaLaunch {
    uint32_t layout, ksteps;
    size_t q_offset;      // byte offset of the chunk's first N-tile in d_qbank
    size_t tg_offset;     // entry offset of the chunk's template ids in d_tglobal
    uint32_t n_tiles16;   // N-tiles in the chunk
    uint32_t n_templates; // real templates in the chunk
    uint32_t mtx, n_rows;   // window enumeration of the pass: M-tiles per row, searched rows (y = 1 + row)
    const uint64_t *live_list;   // packed (page << 32 | row << 12 | col) of the M-tiles that have something to scan
    const uint32_t *live_count;  // device-side length of live_list
    uint32_t super_index;
    uint32_t *queue;      // the launch's item queue (zeroed at the start of the scan)
    MfmaSegs segs;
    uint32_t Lpitch, Lrows;
    uint64_t alg_macs;    // algorithmic MACs of the chunk (true template area x searched windows x templates x pages)
};

// ---- exact verify of one candidate: the reference arithmetic, operation for operation (common.h) ----
typedef v4i v4i_b1 __attribute__((aligned(1)));  // byte-aligned 16-byte view (gfx950 global loads take any alignment)
// Everything the verify needs to know about a template, by GLOBAL template index, 32 bytes: one LDS read instead of three
// dependent gathers from global memory (order_of -> TemplateConst -> needle16_row).
struct VerifyMeta {
    double s_n, n_recip, rnorm_n;
    uint16_t n_w, n_h;
    uint32_t row0;  // first row of the template in needles16
};
struct VerifyArgs {
    const uint8_t *pages;
    uint32_t pitch, rows_alloc;
    KeyFmt fmt;
    const uint32_t *order_of;        // global template index -> class-ordered index
    const TemplateConst *tc;         // class-ordered
    const v4i *needles16;            // every template as n_h rows of 16 bytes (zero padded; two halves per row above 16 px)
    const uint32_t *needle16_row;    // class-ordered first row
    double thr_d;
    const struct VerifyMeta *vmeta;           // by global template index (the row tail's verify stages it in LDS)
    uint32_t n_templates, n_pages, r_w, r_h;  // bounds of a well-formed key
    unsigned long long *flags_word;           // d_res[4]: bit 2 = a key outside those bounds was met (internal error, never a fault)
};
VerifyArgs verify_args(const focr_ctx *c, double thr_d);  // scan_mfma.hip
// LDS: the whole of needles16 is staged in LDS at `lds` (rows.hip, verify_list_kernel) — a compile-time choice, so that every
// template-row load is a plain ds_read or a plain global load, never a per-lane choice between address spaces
template <bool LDS>
__device__ __forceinline__ bool verify_candidate_t(uint64_t key, const VerifyArgs &va, const v4i *lds, float *sim_out) {
    const uint32_t page = va.fmt.page(key), t = va.fmt.t(key), x = va.fmt.x(key), y = va.fmt.y(key);
    if (t >= va.n_templates || page >= va.n_pages || x >= va.r_w || y >= va.r_h) {  // cannot happen; would otherwise be a wild read
        atomicOr(va.flags_word, 4ull);
        *sim_out = 0.f;
        return false;
    }
    const uint32_t ci = va.order_of[t];
    const TemplateConst c = va.tc[ci];
    // template rows: 16 bytes each, zero padded past n_w (two 16-byte halves per row for the 17..32-wide extension)
    const uint32_t halves = c.n_w > 16 ? 2 : 1;
    const uint32_t row0 = va.needle16_row[ci];
    const v4i *nd = va.needles16 + row0;
    const uint8_t *pg = va.pages + ((size_t)page * va.rows_alloc + y) * va.pitch + x;  // rows have >= 64 readable bytes past r_w
    uint32_t acc = 0, s_p = 0, s2_p = 0;
    for (uint32_t hf = 0; hf < halves; hf++) {
        // byte mask of the window's own columns (the padded template columns are zero, but s_p / s2_p need the mask)
        const uint32_t w_here = min(c.n_w - 16 * hf, 16u);
        v4i keep;
#pragma unroll
        for (int k = 0; k < 4; k++)
            keep[k] = w_here >= (uint32_t)(4 * k + 4) ? -1 : (w_here <= (uint32_t)(4 * k) ? 0 : (int)((1u << (8 * (w_here - 4 * k))) - 1u));
#pragma unroll 8
        for (uint32_t j = 0; j < c.n_h; j++) {
            const v4i a = *reinterpret_cast<const v4i_b1 *>(pg + (size_t)j * va.pitch + 16 * hf) & keep;
            v4i b;
            if (LDS) b = lds[row0 + j * halves + hf];
            else b = nd[j * halves + hf];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                acc = __builtin_amdgcn_udot4((uint32_t)a[k], (uint32_t)b[k], acc, false);     // src/ncc.cpp:316-321
                s_p = __builtin_amdgcn_udot4((uint32_t)a[k], 0x01010101u, s_p, false);        // patch_sum, src/ncc.rs:307
                s2_p = __builtin_amdgcn_udot4((uint32_t)a[k], (uint32_t)a[k], s2_p, false);   // sum of squares, src/ncc.rs:308
            }
        }
    }
    const double rnorm_p = window_rnorm(s_p, (uint64_t)s2_p, (double)(c.n_w * c.n_h));
    const double sim = ncc_similarity(acc, s_p, c.s_n, c.n_recip, c.rnorm_n, rnorm_p);
    *sim_out = (float)sim;
    return ncc_emits(sim, va.thr_d);
}
// The row tail's form: template metadata from `meta` (LDS copy of va.vmeta, by global template index), template rows from
// LDS (LDS = true) or global memory.  Same arithmetic, same order of operations.
template <bool LDS>
__device__ __forceinline__ bool verify_candidate_meta(uint64_t key, const VerifyArgs &va, const v4i *lds, const VerifyMeta *meta, float *sim_out) {
    const uint32_t page = va.fmt.page(key), t = va.fmt.t(key), x = va.fmt.x(key), y = va.fmt.y(key);
    if (t >= va.n_templates || page >= va.n_pages || x >= va.r_w || y >= va.r_h) {  // cannot happen; would otherwise be a wild read
        atomicOr(va.flags_word, 4ull);
        *sim_out = 0.f;
        return false;
    }
    const VerifyMeta c = meta[t];
    const uint32_t halves = c.n_w > 16 ? 2 : 1;
    const v4i *nd = va.needles16 + c.row0;
    const uint8_t *pg = va.pages + ((size_t)page * va.rows_alloc + y) * va.pitch + x;  // rows have >= 64 readable bytes past r_w
    uint32_t acc = 0, s_p = 0, s2_p = 0;
    for (uint32_t hf = 0; hf < halves; hf++) {
        const uint32_t w_here = min((uint32_t)c.n_w - 16 * hf, 16u);
        v4i keep;
#pragma unroll
        for (int k = 0; k < 4; k++)
            keep[k] = w_here >= (uint32_t)(4 * k + 4) ? -1 : (w_here <= (uint32_t)(4 * k) ? 0 : (int)((1u << (8 * (w_here - 4 * k))) - 1u));
        // 16 page rows in flight at once (the kernel waits on memory four fifths of its time: one round trip per 16 rows, not
        // two); rows past n_h are read — every page has 48 readable rows below it — but not accumulated
        for (uint32_t j0 = 0; j0 < c.n_h; j0 += 16) {
            v4i a[16];
#pragma unroll
            for (int jj = 0; jj < 16; jj++) a[jj] = *reinterpret_cast<const v4i_b1 *>(pg + (size_t)(j0 + jj) * va.pitch + 16 * hf);
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                const uint32_t j = j0 + jj;
                if (j < c.n_h) {
                    const v4i aw = a[jj] & keep;
                    v4i b;
                    if (LDS) b = lds[c.row0 + j * halves + hf];
                    else b = nd[j * halves + hf];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        acc = __builtin_amdgcn_udot4((uint32_t)aw[k], (uint32_t)b[k], acc, false);    // src/ncc.cpp:316-321
                        s_p = __builtin_amdgcn_udot4((uint32_t)aw[k], 0x01010101u, s_p, false);       // patch_sum, src/ncc.rs:307
                        s2_p = __builtin_amdgcn_udot4((uint32_t)aw[k], (uint32_t)aw[k], s2_p, false);  // sum of squares, src/ncc.rs:308
                    }
                }
            }
        }
    }
    const double rnorm_p = window_rnorm(s_p, (uint64_t)s2_p, (double)((uint32_t)c.n_w * c.n_h));
    const double sim = ncc_similarity(acc, s_p, c.s_n, c.n_recip, c.rnorm_n, rnorm_p);
    *sim_out = (float)sim;
    return ncc_emits(sim, va.thr_d);
}
// The same for banks whose templates are all at most 12 px wide: template rows of 12 bytes (three dwords) from LDS, page rows as
// 12-byte loads, eight in flight — half the LDS and half the registers of the 16-byte form, so that TWO 1 024-thread workgroups
// share a CU (verify_list_kernel<2>, rows.hip).  Same arithmetic, same order of operations.
typedef int v3i __attribute__((ext_vector_type(3)));
typedef v3i v3i_b1 __attribute__((aligned(1)));
__device__ __forceinline__ bool verify_candidate_narrow(uint64_t key, const VerifyArgs &va, const uint32_t *lds_rows, const VerifyMeta *meta, float *sim_out) {
    const uint32_t page = va.fmt.page(key), t = va.fmt.t(key), x = va.fmt.x(key), y = va.fmt.y(key);
    if (t >= va.n_templates || page >= va.n_pages || x >= va.r_w || y >= va.r_h) {  // cannot happen; would otherwise be a wild read
        atomicOr(va.flags_word, 4ull);
        *sim_out = 0.f;
        return false;
    }
    const VerifyMeta c = meta[t];
    const uint8_t *pg = va.pages + ((size_t)page * va.rows_alloc + y) * va.pitch + x;  // rows have >= 64 readable bytes past r_w
    const uint32_t w = c.n_w;  // <= 12
    uint32_t keep[3];
#pragma unroll
    for (int k = 0; k < 3; k++) keep[k] = w >= (uint32_t)(4 * k + 4) ? 0xffffffffu : (w <= (uint32_t)(4 * k) ? 0u : ((1u << (8 * (w - 4 * k))) - 1u));
    const uint32_t *nd = lds_rows + 3 * c.row0;
    uint32_t acc = 0, s_p = 0, s2_p = 0;
    for (uint32_t j0 = 0; j0 < c.n_h; j0 += 8) {  // rows past n_h are read (every page has 48 readable rows below it) but not accumulated
        v3i a[8];
#pragma unroll
        for (int jj = 0; jj < 8; jj++) a[jj] = *reinterpret_cast<const v3i_b1 *>(pg + (size_t)(j0 + jj) * va.pitch);
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {
            const uint32_t j = j0 + jj;
            if (j < c.n_h) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const uint32_t aw = (uint32_t)a[jj][k] & keep[k], b = nd[3 * j + k];
                    acc = __builtin_amdgcn_udot4(aw, b, acc, false);             // src/ncc.cpp:316-321
                    s_p = __builtin_amdgcn_udot4(aw, 0x01010101u, s_p, false);   // patch_sum, src/ncc.rs:307
                    s2_p = __builtin_amdgcn_udot4(aw, aw, s2_p, false);          // sum of squares, src/ncc.rs:308
                }
            }
        }
    }
    const double rnorm_p = window_rnorm(s_p, (uint64_t)s2_p, (double)((uint32_t)c.n_w * c.n_h));
    const double sim = ncc_similarity(acc, s_p, c.s_n, c.n_recip, c.rnorm_n, rnorm_p);
    *sim_out = (float)sim;
    return ncc_emits(sim, va.thr_d);
}
__device__ __forceinline__ bool verify_candidate(uint64_t key, const VerifyArgs &va, float *sim_out) {
    return verify_candidate_t<false>(key, va, nullptr, sim_out);
}

// Per-launch description of the threshold planes of the pass's size classes (scan_mfma2s_kernel).
constexpr int MAX_PLANE_VALUES = 4;  // size classes per pass on the plane path (the kernel is instantiated for 1 / 2 / 4)
struct PlaneArgs {
    const uint16_t *planes;  // int16 plane values (nq): value 0 of the sub-batch's first page; values are `stride` elements apart
    size_t stride;
    uint32_t nv;
    uint32_t shift[MAX_SEGS];      // per segment of the launch: log2 of the unit of its class's plane
    uint32_t seg_value[MAX_SEGS];  // per segment: the plane (value) of its class
    uint32_t seg_full[MAX_SEGS];   // per segment: 0 if the class's templates are all zero in the last K-step (LAYOUT_W12, kept width <= 8)
    uint32_t seg_dead_from[MAX_SEGS];  // per segment: first N-tile (chunk-local) that holds dead / padding slots — a class's live templates come first
};

// scan_mfma.hip (host)
PlaneParams plane_params(const focr_ctx *c, size_t k, double thr_d);
// scan_mfma2.hip
uint32_t mfma2_chunk_tiles(uint32_t ksteps);
int dispatch_mfma_v2s(focr_ctx *c, const MfmaLaunch &L, const PlaneArgs &A, unsigned n_cus);  // A = templates, B = windows, threshold planes
int dispatch_mfma_v2(focr_ctx *c, const MfmaLaunch &L, unsigned n_cus);                       // A = windows, int32 negL tables

}  // namespace focr
