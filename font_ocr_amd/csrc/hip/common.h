// common.h — shared declarations of libfocr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "focr_ncc.h"

namespace focr {

// ---- geometry of one size class -------------------------------------------
// A size class is the set of templates that share (n_w, n_h); window statistics
// depend only on the class (reference: prepare_for_size is cached per size,
// src/ncc.rs:264-268).
struct SizeClass {
    uint32_t n_w, n_h;
    uint32_t ndw;          // dwords per padded template row in the direct kernels (1..4; 5..8 for wide classes)
    uint32_t maxh;         // padded row count in the direct kernel (16 or 32; = n_h for tall classes)
    bool tall;             // n_h > 32 or n_w > 16: scanned by scan_tall_kernel in both modes (no MFMA layout)
    uint32_t n_templates;  // templates in this class
    uint32_t first;        // index of the class's first entry in the class-ordered arrays
    // MFMA prefilter layout
    uint32_t keep_w;          // columns the MFMA multiplies: n_w, or n_w - 1 for a class whose last column is bounded instead
                              // (n_w = 9, 13: one K layout narrower; mfma_common.h, "threshold planes")
    uint32_t layout;          // K layout of the MFMA prefilter (LAYOUT_W8 / W12 / W16, mfma_common.h)
    uint32_t k_groups;        // 16-byte k-groups per window (multiple of 4)
    uint32_t n_tiles16;       // ceil(n_templates / 16)
    uint32_t q_offset;        // byte offset of the class's quantised templates in d_qbank
    uint32_t tg_offset;       // entry offset of the class's template ids in d_tglobal (16 per N-tile, ~0 = padding/dead)
    uint32_t n_live;          // templates that can emit: they take the class's first n_live slots (dead ones and padding follow)
};

// Classes whose A fragments are identical (same K layout, same number of K-steps) are scanned in one kernel
// pass: their N-tiles are concatenated; only the C-in table (negL) changes from class to class.
struct SuperClass {
    uint32_t layout, ksteps;
    std::vector<uint32_t> classes;     // indices into focr_ctx::classes
    std::vector<uint32_t> tile_first;  // first N-tile of each class inside the super-class
    uint32_t n_tiles;
    size_t q_offset, tg_offset;
    // per scan: window enumeration of the pass (smallest searchable template) and its live-tile list
    uint32_t min_w, min_h, mtx, n_rows;
    size_t live_offset;
};

// Per-template constants, computed once on the host in IEEE double exactly as
// the reference's kernel prologue does (src/ncc.cpp:73-86, 278-291).
struct TemplateConst {
    double s_n;      // (double)s_n
    double n_recip;  // 1 / n
    double rnorm_n;  // 1 / sqrt(norm2_n)   (+inf for a constant needle)
    double norm2_n;  // s2_n - s_n^2 / n, for the scalar Rust scan's formula (src/ncc.rs:455)
    uint32_t index;  // global template index (get_hits order)
    uint32_t n_w, n_h;
    uint32_t pad;
};

// Device-side key of a candidate / hit: (page, y, x, t) packed with the minimal field widths of the batch, so that
// one radix sort over bits [0, bits()) puts hits in process_hits order (page, y, x, template) with few passes.
struct KeyFmt {
    uint32_t bt, bx, by, bp;  // bits of template index, x, y, page
    __host__ __device__ uint32_t bits() const { return bt + bx + by + bp; }
    __host__ __device__ uint64_t pack(uint32_t page, uint32_t y, uint32_t x, uint32_t t) const {
        return ((((uint64_t)page << by | y) << bx | x) << bt) | t;
    }
    __host__ __device__ uint32_t t(uint64_t k) const { return (uint32_t)(k & ((1ull << bt) - 1)); }
    __host__ __device__ uint32_t x(uint64_t k) const { return (uint32_t)((k >> bt) & ((1ull << bx) - 1)); }
    __host__ __device__ uint32_t y(uint64_t k) const { return (uint32_t)((k >> (bt + bx)) & ((1ull << by) - 1)); }
    __host__ __device__ uint32_t page(uint64_t k) const { return (uint32_t)(k >> (bt + bx + by)); }
    __host__ __device__ uint64_t line(uint64_t k) const { return k >> (bt + bx); }  // (page << by) | y
};

// The BUCKETS of the row path of the tail (rows.hip).  A bucket is a page row cut into n_seg segments of
// 2^seg_shift pixels (x >> seg_shift): one segment for narrow pages and small banks, more where a whole row would hold
// more candidates than one wave sorts in LDS (BASELINE configs[2]: 1200-px rows x 1520 templates).  Buckets ascend with the
// key: (page, y, x-segment).  (cnt: round 3's tail counted every flushed candidate towards its bucket in the scan kernels' flush
// path; always null since round 5 — the hits-first tail counts hits, in the verify.)
struct RowHist {
    uint32_t *cnt;       // [sub_np * r_h * n_seg], zeroed at the start of the scan
    uint32_t r_h;
    uint32_t shift;      // bt + bx: key >> shift = (page << by) | y   (at most 32 bits)
    uint32_t by;
    uint32_t page_base;  // first page of the sub-batch (keys carry absolute page numbers)
    uint32_t bt, bx;     // x = (key >> bt) & (2^bx - 1)
    uint32_t seg_shift, n_seg;
};
// Everything a scan needs zeroed, in ONE launch (clear_kernel, scan_mfma.hip).  A hipMemsetAsync costs the submitting thread several
// times a kernel launch (six of them stood between a batch's hand-over and its first kernel: ~0.3 ms of a 1.9 ms step).
struct ClearList {
    void *p[8];       // 8-byte aligned
    uint32_t n8[8];   // 8-byte words
    uint32_t n;
    bool add(void *ptr, size_t bytes) {  // the region is zeroed in whole 8-byte words: its owner allocates up to 7 bytes of slack
        if (n >= 8 || bytes / 8 >= 0xffffffffull) return false;
        p[n] = ptr;
        n8[n] = (uint32_t)((bytes + 7) / 8);
        n++;
        return true;
    }
};

__host__ __device__ inline uint32_t row_of_key(uint64_t key, const RowHist &h) {
    const uint32_t line = (uint32_t)(key >> h.shift), x = (uint32_t)(key >> h.bt) & ((1u << h.bx) - 1u);
    return (((line >> h.by) - h.page_base) * h.r_h + (line & ((1u << h.by) - 1u))) * h.n_seg + (x >> h.seg_shift);
}

}  // namespace focr

// Item queues of the persistent scan kernels (scan_mfma2.hip): one per launch, QUEUE_XCDS counters QUEUE_STRIDE dwords
// apart, all zeroed with the counters by the clear launch at the start of a scan (ClearList).
constexpr uint32_t COUNTER_WORDS = 64, QUEUE_XCDS = 8, QUEUE_STRIDE = 32, MAX_SCAN_QUEUES = 128;
constexpr uint32_t TAIL_DONE_WORD = 56, ORDER_DONE_WORD = 57;  // d_counter words: workgroups of the verify / of unit_prefix that have finished ("last workgroup" work)
constexpr size_t COUNTER_BYTES = (COUNTER_WORDS + (size_t)MAX_SCAN_QUEUES * QUEUE_XCDS * QUEUE_STRIDE) * sizeof(uint32_t);

struct focr_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    // An executor (pipe.hip) runs the contexts of one LANE on one stream: a lane's batches follow each other in stream order with no
    // host round trip in between, each in a context of its own (its own pages, scratch and results).  Such a context does not own
    // its stream, waits for ITS batch's last kernel (batch_event, recorded by the executor) instead of for the whole stream —
    // the lane's next batch is queued behind it — and reads results back on the lane's side stream (io_stream) for the same reason.
    bool owns_stream = true;
    hipStream_t io_stream = nullptr;    // device -> host / device -> device copies of finished results (== stream for a context of its own)
    hipEvent_t batch_event = nullptr;   // set by the executor behind a batch's last kernel; consumed by the first wait (wait_batch, ctx.hip)
    bool tail_full_chip = false;        // the executor says nothing scans behind this batch (focr_pipe_end_of_stream / _announce_last): its tail may take every CU (rows2_verify)
    std::string err;

    // bank
    size_t n_templates = 0;
    std::vector<focr_template_t> h_templates;
    std::vector<focr::SizeClass> classes;
    std::vector<focr::SuperClass> supers;
    std::vector<focr::TemplateConst> h_tconst;  // class-ordered
    focr::TemplateConst *d_tconst = nullptr;    // class-ordered
    uint32_t *d_direct_bank = nullptr;          // class-ordered, [maxh][ndw] dwords each
    std::vector<size_t> direct_bank_off;        // dword offset per class
    int8_t *d_qbank = nullptr;                  // quantised i8 templates for the MFMA prefilter (per-lane B layout)
    bool column_drop = true;                    // bound the last column of 9- / 13-wide classes instead of multiplying it (takes effect at the next bank upload)
    std::vector<uint32_t> mfma_slot;            // per class-ordered template: its slot inside its class's N-tiles (tile = slot / 16)
    // ---- result sizes (ctx.hip: finish_results) ----
    // Every phase after the scan kernel takes its element count from device memory; the host only supplies upper bounds for
    // grids and buffers.  Exact mode reads the counts between the phases (as round 1 did); estimated mode (same bank,
    // geometry, threshold, cap as the previous scan) bounds them by the previous scan's counts + a margin (4 .. 20 %), launches everything
    // without waiting, and reads all sizes once at the end; a count above its bound redoes the batch in exact mode.
    uint64_t *d_res = nullptr;   // [0] candidates [1] hits [2] matches [3] lines << 32 | chars [4] overflow flag [7] scratch count
    uint64_t *h_res = nullptr;   // pinned copy
    uint32_t *h_live = nullptr;  // pinned copy of the live M-tile counts (d_counter + 8 ..), 40 entries
    const uint64_t *d_n_hits = nullptr;  // device-side number of hits of the last scan
    size_t ub_hits = 0;                  // the bound its buffers were sized for
    uint64_t n_hits_raw_u64 = 0;
    bool sizes_pending = false, post_pending = false, estimated = false, estimates_enabled = true;
    size_t est_cand = 0, est_hits = 0, ub_cand = 0;
    // how much the counts of consecutive scans of one setup have differed lately (relative; decays by a quarter per scan):
    // the next scan's bounds are the last counts + 3 x this, between 4 % and 20 % (finish_results)
    double est_var = 0.0667;
    uint64_t est_last_cand = 0, est_last_hits = 0;
    uint64_t est_sig = 0, bank_gen = 0, bank_hash = 0, counters_redone = 0;
    float scan_thr = 0.f, post_anchor = 0.f;
    int scan_mode = 0;
    int32_t post_overlap = 0;
    bool force_split = false;                   // tests: take scan_split without waiting for an overflow (focr_debug_force_split)
    int dbg_stats_form = 0;  // tests / A-B: 1 = the LDS-tiled statistics kernel for every class (focr_debug_set_stats_form; 0: the register form where it applies)
    uint32_t dbg_grid_num = 0, dbg_grid_den = 0;  // tests: the tail's persistent kernels on num / den times their workgroups (focr_debug_set_tail_grid; 0: as designed)
    int prefilter = 0;                          // FOCR_PREFILTER_*: auto / plane kernel / legacy kernel (focr_ctx_set_prefilter)
    uint16_t *d_planes = nullptr;               // threshold planes, int16: [super-class][value][page][Lrows][Lpitch] (mfma_common.h)
    size_t planes_bytes = 0;
    uint32_t *d_tglobal = nullptr;              // class-ordered -> global template index, 0xffffffff = never emits
    uint32_t *d_order_of = nullptr;             // global template index -> class-ordered index
    std::vector<double> mfma_c_scale, mfma_e_max, mfma_rho_max;  // per class: quantisation scale, max rounding-error norm, max norm of a unit template's dropped column
    uint8_t *d_needles = nullptr;               // dense needles (class-ordered, for verify)
    std::vector<uint32_t> h_needle_off;         // class-ordered byte offsets into d_needles
    uint32_t *d_needle_off = nullptr;
    uint8_t *d_needles16 = nullptr;             // class-ordered, n_h rows of 16 bytes each (verify operand)
    uint32_t *d_needle16_row = nullptr;         // class-ordered first row index into d_needles16
    void *d_vmeta = nullptr;                    // VerifyMeta by GLOBAL template index (mfma_common.h): what the verify needs about a template, 32 B
    // the verify operand once more, ordered by GLOBAL template index, for banks that do not fit the LDS whole: a chunk of
    // consecutive templates is then one contiguous piece (verify_chunks_kernel, rows.hip)
    uint32_t *d_vrows_t = nullptr;              // rows of vrow_bytes each (12 when every template is at most 12 px wide, else 16), by global index
    void *d_vmeta_t = nullptr;                  // VerifyMeta by global index whose row0 counts rows of d_vrows_t
    std::vector<uint32_t> h_vrow0_t;            // [n_templates + 1] first row of every template in d_vrows_t
    uint32_t vrow_bytes = 0;                    // 0: no such copy (a template wider than 16 px)
    uint32_t *d_t_w = nullptr, *d_t_h = nullptr, *d_t_letter = nullptr;  // by global template index

    // pages: [n_pages][rows_alloc][pitch] ink-high u8, zero padded
    size_t n_pages = 0, r_w = 0, r_h = 0, pitch = 0, rows_alloc = 0;
    size_t pages_capacity = 0;  // pages d_pages was allocated for (>= n_pages)
    unsigned n_cus = 0;         // compute units of the device (read once, focr_ctx_create)
    unsigned scan_cus = 0;      // CUs the persistent scan kernel may occupy, 0 = all (focr_ctx_set_scan_cus)
    uint8_t *d_pages = nullptr;
    uint8_t *d_pages_i8 = nullptr;  // the same pages as int8 (ink - 128, i.e. byte ^ 0x80; padding = 0x80): the MFMA prefilter's window operand,
                                    // written at ingest so that the scan kernels need no v_xor per fragment dword
    uint8_t *d_stage = nullptr;  // device staging for uploads
    size_t stage_bytes = 0;
    // the executor's second page set (pipe.hip, focr_pipe_prefetch): the NEXT batch of a lane is ingested here, on the lane's copy
    // stream, while the lane still scans d_pages; the two sets change places when that batch starts (pages_alt_swap, ctx.hip)
    struct PageSet {
        uint8_t *u8 = nullptr, *i8 = nullptr;
        size_t capacity = 0, r_w = 0, r_h = 0, pitch = 0, rows_alloc = 0;
    } alt;

    // scan results
    size_t sub_p0 = 0, sub_np = 0;  // page range the scan pipeline is currently working on (normally the whole batch)
    focr::KeyFmt fmt{};
    bool scanned = false;
    uint32_t cap = FOCR_MAX_MATCHES;
    size_t hit_capacity = 0;     // entries in d_hit_keys / d_hit_sims
    uint64_t *d_hit_keys = nullptr, *d_hit_keys_alt = nullptr;
    float *d_hit_sims = nullptr, *d_hit_sims_alt = nullptr;
    uint32_t *d_counter = nullptr;  // u64 [0] hits, u64 [1] candidates, u32 [8..47] live M-tile counts, then the scan kernels' item queues
    unsigned stats_turn = 0;        // this batch's place in the device's chain of statistics launches (launch_scan_mfma)
    uint32_t scan_queues_used = 0;  // item queues handed out since the last reset (launch_scan_mfma)
    size_t cand_capacity = 0, cand_alt_capacity = 0;
    uint64_t *d_cand = nullptr, *d_cand_alt = nullptr;
    bool ordered = false;  // the scan path already produced d_matches (MFMA path); order_hits is skipped
    int32_t *d_L = nullptr;  // prefilter thresholds [class][page][r_h][pitchL]
    size_t L_bytes = 0;
    void *d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    size_t n_hits_raw = 0;    // hits before the cap
    size_t n_cand = 0;
    uint32_t *d_seg_count = nullptr;   // [n_pages*T] capped counts
    uint64_t *d_seg_start = nullptr;   // [n_pages*T] start in the sorted arrays
    uint64_t *d_seg_offset = nullptr;  // [n_pages*T + 1] CSR offsets of the capped lists
    size_t seg_alloc = 0;
    focr_match_t *d_matches = nullptr;  // capped, ordered by (page, template, y, x)
    // all hits (before the cap) in process_hits order (page, y, x, t); d_keep[i] = survives its (page, template) cap
    uint64_t *d_hkeys = nullptr;
    float *d_hsims = nullptr;
    size_t n_hits = 0;
    size_t matches_alloc = 0;
    size_t n_matches = 0;

    // process_hits results (device-resident; copied to the host on focr_get_lines)
    struct DevBuf {  // grow-only device scratch
        void *p = nullptr;
        size_t bytes = 0;
        void *ensure(focr_ctx *c, size_t want);
        void *ensure_keep(focr_ctx *c, size_t want, size_t keep_bytes);
        void release();
    };
    DevBuf scan_flags, scan_pos, scan_live, scan_live_list;
    // row path of the tail (rows.hip): candidates bucketed by page row, sorted + verified per row
    focr::RowHist row_hist{};   // what the scan kernels' flush path counts into (cnt == nullptr: legacy tail)
    DevBuf rows_hits, rows_hbase, rows_big;
    uint32_t row_cap = 0;       // per-row candidate capacity the row kernel was instantiated for in the last scan
    uint32_t est_row_max = 0;   // largest row of the previous scan of this setup (estimated mode picks the capacity from it)
    uint32_t row_seg_shift = 0; // log2 of the x-segment width of the buckets (0: not chosen yet for this setup; rows.hip, row_segments)
    bool chunked_verify = true; // hits-first tail, banks above the LDS: verify in chunk passes (false: template rows gathered from global memory; FOCR_VERIFY_GLOBAL, A/B)
    int tail_mode = 1;          // focr_ctx_set_row_tail(): 0 = the legacy tail (radix sort + verify + compaction), 1 = hits-first row tail
                                // (verify in flush order, hits bucketed + sorted: the default)
    DevBuf ord_k2, ord_k2_alt, ord_v, ord_v_alt, ord_keep;
    DevBuf acc_matches, acc_seg_count, acc_hkeys, acc_hsims;  // split-batch mode: results appended sub-batch by sub-batch
    DevBuf post_line_be;
    DevBuf post_keep, post_choice, post_owner, post_packed, post_scanned, post_page_off, post_line_off, post_chars;
    bool lines_on_host = false;
    bool processed = false;
    size_t n_chars = 0, n_lines = 0;
    std::vector<uint64_t> h_page_line_off, h_line_char_off;
    std::vector<focr_hit_t> h_chars;

    hipEvent_t ev[8] = {};
    float ms[6] = {};
    uint64_t counters[4] = {};

    // per-launch timing of the scan kernels (focr_last_launches)
    std::vector<focr_launch_info_t> launches;
    std::vector<hipEvent_t> launch_events;  // pool, two per launch
    void launch_begin(const char *name, uint32_t n_templates, uint64_t alg, uint64_t issued);
    void launch_end();
    void launches_reset() { launches.clear(); }
    void launches_collect();  // after a stream sync: fill ms
};

namespace focr {

void set_global_error(const std::string &s);
int fail(focr_ctx *ctx, int code, const std::string &msg);

#define FOCR_HIP(ctx, expr)                                                                       \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return focr::fail((ctx), FOCR_ERR_NO_DEVICE,                                          \
                              std::string(#expr) + ": " + hipGetErrorString(e_));                 \
    } while (0)

// launchers implemented in the .hip files
int launch_scan_direct(focr_ctx *ctx, float threshold, int rust_formula);
int launch_scan_mfma(focr_ctx *ctx, float threshold);
int launch_clear(focr_ctx *c, const focr::ClearList &l);  // zero every region of the list in one launch (scan_mfma.hip)
int exclusive_scan_u64(focr_ctx *c, const uint64_t *in, uint64_t *out, size_t n);
int order_hits(focr_ctx *ctx);  // direct path: unordered hits in d_hit_keys / d_hit_sims -> everything below
int finish_results(focr_ctx *c);  // wait for the stream once and read the result sizes of the last scan / process_hits
int build_mfma_bank(focr_ctx *ctx, const uint8_t *needles);
void bank_host_prepare(focr_ctx *c, const focr_template_t *templates, size_t n_templates, const uint8_t *needles,
                       std::vector<uint32_t> &direct, std::vector<uint8_t> &dense);
void layout_supers(focr_ctx *c);  // size classes -> super-classes, MFMA K layouts, bank offsets (host only)
int pages_alt_ingest(focr_ctx *c, const void *d_luma, size_t n_pages, size_t r_w, size_t r_h, int invert, hipStream_t s);  // ctx.hip
int pages_alt_swap(focr_ctx *c, size_t n_pages, size_t r_w, size_t r_h);
void ctx_share_stream(focr_ctx *c, hipStream_t lane_stream, hipStream_t io_stream);  // ctx.hip: the context joins an executor's lane
bool post_queue_chars_copy(focr_ctx *c, void *dst, size_t dst_bytes);  // post.hip: the batch's characters to a device buffer, queued on the context's stream
int wait_batch(focr_ctx *c);  // ctx.hip: until the context's queued work is done (its batch's event inside an executor, else its stream)
int quantise_bank(focr_ctx *c, const uint8_t *dense, std::vector<int8_t> &qbank, std::vector<uint32_t> &tglobal, std::vector<uint32_t> &order_of);  // host only

// ---- device helpers: the reference's f64 epilogue, operation for operation ----
// Compiled with -ffp-contract=off: the only fused operation is the explicit fma.

// patch_rnorm, src/ncc.rs:309-311: 1 / sqrt(s2 - (s*s)/n), IEEE division and sqrt.
__device__ __forceinline__ double window_rnorm(uint32_t s_p, uint64_t s2_p, double n_d) {
    double norm = (double)s2_p - ((double)((uint64_t)s_p * (uint64_t)s_p)) / n_d;
    return 1.0 / __builtin_sqrt(norm);
}

// similarity, src/ncc.cpp:352-361 (== 207-215): fnmadd(s_n * s_p, 1/n, acc) * (rnorm_n * rnorm_p)
// with the vector path's signed int32 -> f64 conversions (_mm256_cvtepi32_pd).
__device__ __forceinline__ double ncc_similarity(uint32_t acc, uint32_t s_p, double s_n_d, double n_recip,
                                                 double rnorm_n, double rnorm_p) {
    double num = __builtin_fma(-(s_n_d * (double)(int32_t)s_p), n_recip, (double)(int32_t)acc);
    double den = rnorm_n * rnorm_p;
    return num * den;
}

// The scalar Rust scan's arithmetic and skips (`ncc --rust`, src/ncc.rs:431-433, 445-470): returns "emits".
__device__ __forceinline__ bool rust_similarity(uint32_t acc, uint32_t s_p, uint64_t s2_p, double s_n_d, double norm2_n,
                                                double n_d, double thr_d, double *sim) {
    if (s_n_d == 0.0 || s_p == 0) return false;                                          // :431-433, :447-449
    const double num = (double)acc - (double)((uint64_t)s_n_d * (uint64_t)s_p) / n_d;    // :450
    if (num < 0.) return false;                                                          // :451-453
    const double norm2_p = (double)s2_p - (double)((uint64_t)s_p * (uint64_t)s_p) / n_d;  // :456
    const double den = __builtin_sqrt(norm2_n * norm2_p);                                // :459
    *sim = num / den;                                                                    // :460
    return !(*sim == __builtin_inf()) && *sim > thr_d;                                   // :466
}

// emit test, src/ncc.cpp:362-366: (sim > thr) && !(sim == +inf); NaN fails both compares.
__device__ __forceinline__ bool ncc_emits(double sim, double thr_d) {
    return (sim > thr_d) && !(sim == __builtin_inf());
}

}  // namespace focr
