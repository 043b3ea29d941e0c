// mfma_common.h — declarations shared by the MFMA prefilter kernels (scan_mfma.hip, scan_mfma2.hip).
#pragma once
#include "common.h"

namespace focr {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int32_t REJECT = 0x3fffffff;  // |negL| of a window the reference never emits
constexpr uint32_t WBUF = 64;           // wave-private candidate staging entries in LDS (no atomics on the way in)

// one global atomic per flush: lane 0 reserves `count` slots, the wave copies its staged keys out coalesced
__device__ __forceinline__ void flush_wave_candidates(uint64_t *wbuf, uint32_t count, int lane, uint64_t *__restrict__ cand,
                                                      unsigned long long *__restrict__ cand_counter, unsigned long long cand_cap) {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(cand_counter, (unsigned long long)count);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    base = ((unsigned long long)hi << 32) | lo;
    if ((uint32_t)lane < count && base + lane < cand_cap) cand[base + lane] = wbuf[lane];
}

// How the waves of a persistent scan kernel get their work.  Items (MT consecutive live M-tiles) are split into one
// contiguous range per XCD: workgroups b and b + 8 share an XCD and its L2, and neighbouring M-tiles share page rows.  Inside
// a range the waves TAKE items from a queue (one agent-scope atomic per item, requested one item ahead so that its latency
// is never waited for) instead of owning a fixed stride: a workgroup needs an empty CU (all of its LDS, all 512 VGPRs of
// each SIMD), so with other contexts' kernels on the GPU the workgroups of one launch start up to a few 100 us apart, and
// the time per item varies with the candidates it finds; under a fixed split the launch ends when its unluckiest wave does
// (BASELINE configs[1], alone on the GPU: 2.20 -> 1.83 ms).  A wave whose range is exhausted goes on with the next XCD's; it
// stops once it has seen every range exhausted, which every wave reaches after at most n_xc extra requests.
// The queue (QUEUE_XCDS counters, QUEUE_STRIDE dwords apart) is zeroed by the memset that opens every scan (launch_scan_mfma).
struct ItemTaker {
    uint32_t *queue;
    uint32_t n_items, n_xc, per_xc, cur_q, hops, ticket_v;
    int lane;
    __device__ __forceinline__ void request() {
        if (lane == 0) ticket_v = __hip_atomic_fetch_add(queue + cur_q * QUEUE_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        request();
    }
    // the wave's next item (wave-uniform), or false when there is none left.  Call request() once per item, after the item's
    // loads have been issued, for the ticket this reads the next time round.
    __device__ __forceinline__ bool next(uint32_t &item) {
        for (;;) {
            const uint32_t ticket = __builtin_amdgcn_readfirstlane(ticket_v);
            const uint32_t qb = cur_q * per_xc, qe = min(n_items, qb + per_xc);
            if (qb < qe && ticket < qe - qb) {
                item = qb + ticket;
                return true;
            }
            if (++hops >= n_xc) return false;
            cur_q = cur_q + 1 == n_xc ? 0 : cur_q + 1;
            request();
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
//     last K-step of a 16-row block holds nothing but zeros for templates of n_w <= 8 and the kernels skip its MFMAs for
//     such a size class (Mfma3Args::seg_full).  A lane loads its 4 rows once (12 contiguous bytes each) whatever the order.
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

// ---- two-stage prefilter (scan_mfma3.hip; host data: lowrank.hip) ----
// Window norms of one super-class, planar: norms[v][page][Lrows][Lpitch] as f16 rounded towards zero (a lower bound of the
// norm, at most 2^-10 below it), value v = sqrt(V / n) of box v
// (V = n*s2 - s^2, exact integer).  Values 0 .. n_cls-1 are the super-class's size classes and carry the class's emit
// flag in the sign (> 0: the reference can emit there: x, y >= 1, window inside the page, variance > 0; <= 0: never);
// when no class has the frame's box, one more value holds the frame norm.  |value| is always the norm.
constexpr int LR_MAX_VALUES = 4;
// Stage-2 K slots (LR_K = 32): slot = 16 b + 4 g + v lives in lane group g, element 4 b + v of the 8-element bf16
// operand.  The 2 + n_cls "extras" (R, N_F, one threshold slot per class) sit at compile-time element positions so the
// kernel places them with a select on the lane group only: extra e < 4 -> element 7 of lane group e, extra e >= 4 ->
// element 6 of lane group e - 4.  The r principal directions fill the remaining slots in increasing slot order.
__host__ __device__ inline uint32_t lr_extra_slot(uint32_t e) { return e < 4 ? 16 + 4 * e + 3 : 16 + 4 * (e - 4) + 2; }
inline uint32_t lr_comp_slot(uint32_t j, uint32_t n_extras) {  // host: slot of principal direction j
    for (uint32_t s = 0, k = 0; s < LR_K; s++) {
        bool is_extra = false;
        for (uint32_t e = 0; e < n_extras; e++) is_extra |= lr_extra_slot(e) == s;
        if (is_extra) continue;
        if (k == j) return s;
        k++;
    }
    return 0xffffffffu;
}

struct Mfma3Args {
    const uint16_t *norms;   // f16 bits: value 0 of the sub-batch's first page; values are `norm_stride` elements apart
    size_t norm_stride;      // elements between consecutive values
    uint32_t nv, n_cls, frame_value;  // frame_value: index of the value that holds the frame norm
    float inv_lambda;
    float thr_lo;            // thr_eff widened by 2^-20 away from the emitting side: theta = thr_lo * norm_c is a lower bound
    uint32_t theta_add;      // 0xffff if thr_lo < 0 (bf16 rounding of theta towards -inf), else 0
    float kq[MAX_SEGS];      // per segment of the launch: L = floor(kq * norm_c) - 2, the exact-taps stage's threshold
    uint32_t seg_value[MAX_SEGS];  // per segment: the norm value of its class
    uint32_t seg_full[MAX_SEGS];   // per segment: 0 if the class's templates are all zero in the last K-step (LAYOUT_W12, n_w <= 8)
};

// scan_mfma3.hip
uint32_t mfma3_chunk_tiles(uint32_t ksteps);
int dispatch_mfma_v3(focr_ctx *c, const MfmaLaunch &L, const Mfma3Args &A3, const int8_t *basis, const uint16_t *gbank, unsigned n_cus);
// scan_mfma2.hip
uint32_t mfma2_chunk_tiles(uint32_t ksteps);
int dispatch_mfma_v2s(focr_ctx *c, const MfmaLaunch &L, const Mfma3Args &A3, unsigned n_cus);  // roles swapped, norms instead of negL
int dispatch_mfma_v2(focr_ctx *c, const MfmaLaunch &L, unsigned n_cus);

}  // namespace focr
