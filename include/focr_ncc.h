/*
 * focr_ncc.h — C ABI of the MI355X-native NCC template-matching scan.
 *
 * This is the drop-in boundary for the hot path of aconz2/font-ocr's `ncc`
 * tool.  Everything is `extern "C"`, plain pointers and sizes; no C++ or
 * torch types.  Implemented by font_ocr_amd/csrc (libfocr_hip.so), hand-written
 * HIP for gfx950.  There is no CPU fallback: every compute entry point fails
 * (FOCR_ERR_NO_DEVICE / returns 0 matches and sets the error string) when no
 * HIP device is usable.
 *
 * Two layers:
 *
 *  (1) The reference's own FFI symbols, `ncc_8_u8` / `ncc_16_u8`
 *      (reference: src/ncc.cpp:48-63 and 253-268; Rust declarations
 *      src/ncc.rs:92-126).  Same names, same parameters, same return
 *      convention, so the reference's Rust host links against this library
 *      unchanged.  One call = one template over one page.
 *
 *  (2) A batched API (`focr_*`) that the reference's per-page driver
 *      (get_hits / Searcher, src/ncc.rs:544-721, 231-404) maps onto: upload a
 *      template bank once, keep N pages resident in HBM, scan all
 *      (page x template) pairs in one pass, read back per-template match
 *      lists that are identical to what N x T calls of (1) would return, and
 *      optionally run the anchor/line/overlap post-process (process_hits,
 *      src/ncc.rs:723-786) on the device.
 */
#ifndef FOCR_NCC_H
#define FOCR_NCC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Wire format of one match: reference `Match` (src/ncc.cpp:7-10) == `MatchC`
 * (src/ncc.rs:66-72).  8 bytes, repr(C). */
typedef struct focr_match {
    uint16_t x, y;
    float similarity;
} focr_match_t;

/* MAX_MATCHES, src/ncc.rs:31: the reference gives every kernel call a
 * 1024-entry output buffer; a scan stops once it is full. */
#define FOCR_MAX_MATCHES 1024

/* ------------------------------------------------------------------------ */
/* (1) Drop-in kernels.  Replaces src/ncc.cpp:48-251 (ncc_8_u8) and
 * src/ncc.cpp:253-396 (ncc_16_u8).
 *
 *  reference   r_w*r_h inverted grey page, row-major (caller-owned, host)
 *  needle_u8   n_h rows of N bytes (N = 8 resp. 16), zero-padded past n_w
 *  acc,acc_len caller scratch; the reference memsets it, so does this
 *              implementation (contents after the call: all zero)
 *  patch_sum   [r_h][r_w] window sums, valid on [start,end) of each row
 *  patch_rnorm [r_h][r_w] 1/sqrt(window variance*n), same validity
 *  start_end   [2*r_h] per-row [start,end) x-extent (src/ncc.rs:313-314)
 *  threshold   emit iff sim > threshold and sim != +inf
 *  out,n_out   caller buffer; returns the number of matches written, in
 *              (y,x)-ascending order; == n_out means "full, scan stopped".
 *
 * Thread-safe: each calling thread gets its own device stream and staging
 * buffers.  No error channel exists in the reference signature; on a device
 * failure these return 0 and focr_last_error_global() describes it.
 * Residency: the reference's host calls these once per template with the SAME
 * page and window tables, so a thread's inputs stay on the device and are sent
 * again only when a 64-bit content hash (non-cryptographic) of an input or the
 * geometry changes — a hash collision between two different inputs of one
 * geometry (~2^-64 per pair) would reuse the stale copy.
 * FOCR_COMPAT_ALWAYS_UPLOAD=1 in the environment disables it (every call
 * uploads every input). */
size_t ncc_8_u8(uint8_t *reference, size_t r_w, size_t r_h, uint8_t *needle_u8, size_t n_w,
                size_t n_h, uint32_t *acc, size_t acc_len, uint32_t *patch_sum,
                double *patch_rnorm, uint16_t *start_end, float threshold, focr_match_t *out,
                size_t n_out);
size_t ncc_16_u8(uint8_t *reference, size_t r_w, size_t r_h, uint8_t *needle_u8, size_t n_w,
                 size_t n_h, uint32_t *acc, size_t acc_len, uint32_t *patch_sum,
                 double *patch_rnorm, uint16_t *start_end, float threshold, focr_match_t *out,
                 size_t n_out);

/* ------------------------------------------------------------------------ */
/* (2) Batched API. */

typedef struct focr_ctx focr_ctx_t;

enum {
    FOCR_OK = 0,
    FOCR_ERR_NO_DEVICE = 1,  /* no usable HIP device / runtime error (see focr_last_error) */
    FOCR_ERR_INVALID = 2,    /* bad argument (sizes, null pointers, n_w > 16 ...) */
    FOCR_ERR_STATE = 3,      /* call order (scan before bank/pages, results before scan) */
    FOCR_ERR_OVERFLOW = 4,   /* internal candidate buffer could not be grown enough */
    FOCR_ERR_NOMEM = 5
};

/* One template of the bank: a dense n_w x n_h A8 glyph raster (canvas pixels
 * verbatim, src/ncc.rs:640-641, 894-896).  n_w <= 16 as in the reference (it
 * panics above that, src/ncc.rs:392); 17 <= n_w <= 32 is accepted as an
 * extension.  1 <= n_h <= 255.  Heights above 32 and widths above 16 take a
 * slower exact kernel.  Templates are indexed in
 * get_hits order: sub-pixel offset major, alphabet order minor
 * (src/ncc.rs:587, 630). */
typedef struct focr_template {
    uint32_t letter;    /* Unicode code point (src/ncc.rs:676-680) */
    uint16_t n_w, n_h;  /* canvas size */
    uint32_t offset;    /* byte offset of the n_w*n_h pixels in `needles` */
    uint16_t shift_x, shift_y; /* index of the sub-pixel offset on the x / y grid */
    float off_x, off_y;        /* offset[0], offset[1] (src/ncc.rs:569) */
    float corrected_off_y;     /* offset[1] + y_offset (src/ncc.rs:629) */
    float bearing_x;           /* typographic left bearing in px (src/ncc.rs:673) */
} focr_template_t;

/* Which device formulation scans the bank.  Both give identical results. */
enum {
    FOCR_SCAN_MFMA = 0,   /* i8 MFMA conservative prefilter + exact verify (default, fast) */
    FOCR_SCAN_DIRECT = 1, /* exact v_dot4 evaluation of every (window, template) */
    FOCR_SCAN_RUST = 2    /* same kernel with the arithmetic of the reference's scalar Rust scan
                           * (`ncc --rust`, src/ncc.rs:406-483): num = acc - s_n*s_p/n (a division),
                           * sim = num / sqrt(norm2_n * norm2_p), windows with s_p == 0 or num < 0
                           * are skipped, templates with s_n == 0 yield nothing; that path has no
                           * 1024 cap (pass cap = UINT32_MAX) */
};

/* One raw or post-processed hit with its template, as MatchWithLetter
 * (src/ncc.rs:74-79): rect = (x, y, w, h). */
typedef struct focr_hit {
    uint16_t x, y;
    uint16_t w, h;
    float similarity;
    uint32_t letter;
    uint32_t template_index;
} focr_hit_t;

/* Context = one device + one stream + all buffers.  One per thread/GPU (the contexts of an executor, focr_pipe_*, share
 * their lane's stream). */
int focr_ctx_create(int device, focr_ctx_t **out);
void focr_ctx_destroy(focr_ctx_t *ctx);
const char *focr_last_error(const focr_ctx_t *ctx);
const char *focr_last_error_global(void);
int focr_device_count(void); /* 0 when no device / no driver */

/* Upload the bank (replaces the per-page re-rasterisation + per-call needle
 * padding of src/ncc.rs:587-649, 340-344, 367-371).  Copies everything. */
int focr_bank_upload(focr_ctx_t *ctx, const focr_template_t *templates, size_t n_templates,
                     const uint8_t *needles, size_t needles_len);

/* Reserve n_pages resident pages of r_w x r_h (Searcher::new storage,
 * src/ncc.rs:231-261, minus the tables this design does not need). */
int focr_pages_alloc(focr_ctx_t *ctx, size_t n_pages, size_t r_w, size_t r_h);
/* Copy `count` host pages (tightly packed r_w*r_h luma8 each) into slots
 * [first, first+count).  invert != 0 applies image_to_u8's 255-px on the
 * device (src/ncc.rs:887-892); invert == 0 means the bytes are already
 * ink-high. */
int focr_pages_upload(focr_ctx_t *ctx, size_t first, size_t count, const uint8_t *luma,
                      int invert);
/* Page-locked host memory for page pixels.  From such a buffer
 * focr_pages_upload is one asynchronous DMA on the context's stream: it
 * returns as soon as the copy is queued, so the host can decode the next batch
 * (or drive a second context) meanwhile.  The buffer must stay untouched until
 * the context's next synchronising call returns (focr_sync, focr_scan). */
int focr_host_alloc(size_t bytes, void **out);
void focr_host_free(void *p);
/* Page-lock memory the caller already owns (page-aligned start; e.g. slabs that image decoders began to fill before
 * the HIP runtime was up), with the same effect on focr_pages_upload as focr_host_alloc memory. */
int focr_host_register(void *p, size_t bytes);
void focr_host_unregister(void *p);
/* Same, from a device pointer (pages already in HBM, e.g. a torch tensor). */
int focr_pages_upload_device(focr_ctx_t *ctx, size_t first, size_t count, const void *d_luma,
                             int invert);

/* Scan every resident page with every template (Searcher::search_c_u8 for all
 * templates of all pages, src/ncc.rs:332-404 + 587-701).  Asynchronous on the
 * context's stream up to the point where result sizes are needed.
 * threshold: as --threshold (src/ncc.rs:507).  cap: per-(page,template) match
 * limit, FOCR_MAX_MATCHES for reference behaviour. */
int focr_scan(focr_ctx_t *ctx, float threshold, uint32_t cap, int mode);

/* Results of the last scan.  counts is [n_pages][n_templates] (value == cap
 * means the reference would have warned "got >= 1024 matches",
 * src/ncc.rs:395-397). */
int focr_get_counts(focr_ctx_t *ctx, uint32_t *counts);
size_t focr_total_matches(focr_ctx_t *ctx);
/* All matches in (page, template, y, x) order; offsets is
 * [n_pages*n_templates + 1] (CSR), matches has focr_total_matches entries. */
int focr_get_matches(focr_ctx_t *ctx, uint64_t *offsets, focr_match_t *matches);

/* process_hits on the device (src/ncc.rs:723-786 + partition_by 1036-1052)
 * over the last scan's hits, for every page.  A page without hits yields zero
 * lines (the reference panics there, src/ncc.rs:1040). */
int focr_process_hits(focr_ctx_t *ctx, float anchor_threshold, int32_t overlap);
size_t focr_total_chars(focr_ctx_t *ctx);
size_t focr_total_lines(focr_ctx_t *ctx);
/* page_line_offsets: [n_pages+1] index into line_char_offsets;
 * line_char_offsets: [total_lines+1] index into chars; chars: [total_chars]. */
int focr_get_lines(focr_ctx_t *ctx, uint64_t *page_line_offsets, uint64_t *line_char_offsets,
                   focr_hit_t *chars);
/* focr_get_lines without the intermediate host copy: device -> the caller's buffers ([n_pages+1], [total_lines+1],
 * [total_chars] entries; all three required). */
int focr_get_lines_into(focr_ctx_t *ctx, uint64_t *page_line_offsets, uint64_t *line_char_offsets,
                        focr_hit_t *chars);
/* Device pointer to the focr_total_chars() post-processed characters (page, line, x order) of the
 * last focr_process_hits, valid until the next scan/process call; NULL if there are none.  For
 * device-side consumers (e.g. the RCCL gather of match lists across GPUs). */
const focr_hit_t *focr_lines_device_chars(focr_ctx_t *ctx);

/* Device time (ms, HIP events on the context's stream) of the phases of the
 * last focr_scan / focr_process_hits: [0] window statistics, [1] scan kernel
 * (MFMA prefilter or direct), [2] exact verify, [3] ordering + cap,
 * [4] process_hits, [5] whole scan.  For bench.py's roofline object. */
int focr_last_timings(focr_ctx_t *ctx, float ms[6]);
/* Work counters of the last scan: [0] candidates out of the prefilter,
 * [1] hits before the cap, [2] algorithmic MACs (true template area x searched
 * windows), [3] MACs issued incl. padding. */
int focr_last_counters(focr_ctx_t *ctx, uint64_t c[4]);
int focr_sync(focr_ctx_t *ctx);

/* Upper bound on the compute units the persistent MFMA scan kernel occupies
 * (one workgroup per CU); 0 = all of them (default).  With several contexts in
 * flight on one GPU (one host thread each, batches taken round-robin) leaving
 * some CUs free lets the other contexts' statistics / sort / verify / ordering
 * kernels run beside the scan instead of queueing behind it (DESIGN.md
 * section 5). */
int focr_ctx_set_scan_cus(focr_ctx_t *ctx, unsigned max_cus);

/* Which MFMA prefilter kernel FOCR_SCAN_MFMA uses.  Results are identical in every mode (conservative filters in front of
 * the same exact verify); only the speed differs.
 *   AUTO / ONE_STAGE  scan_mfma2s_kernel: A = templates, B = windows, C-in from the int16 threshold planes (scan_mfma2.hip)
 *   LEGACY            round 1's kernel (A = windows) with per-class int32 threshold tables — what size classes with more
 *                     than 4 K-steps always use; selectable as a cross-check
 * (Value 2 was round 2's two-stage low-rank prefilter: exact, not faster, removed in round 3 — DESIGN.md, dead ends.) */
enum {
    FOCR_PREFILTER_AUTO = 0,
    FOCR_PREFILTER_ONE_STAGE = 1,
    FOCR_PREFILTER_LEGACY = 3
};
int focr_ctx_set_prefilter(focr_ctx_t *ctx, int mode);

/* Column drop (default on): size classes 9 or 13 px wide give their last column to a Cauchy-Schwarz bound inside the
 * prefilter threshold instead of multiplying it, and take the next narrower MFMA K layout (9x15 templates: 2 K-steps of 64
 * bytes instead of 3).  Results are identical either way; off = every column multiplied (cross-check, A/B).  Takes effect at
 * the next focr_bank_upload. */
int focr_ctx_set_column_drop(focr_ctx_t *ctx, int on);

/* Tail of the MFMA scan (rows.hip).  1 (default) = the hits-first row path: the candidates are verified where the scan kernels
 * left them (flush order: neighbouring lanes, neighbouring windows), then only the HITS are bucketed by page row and sorted
 * per bucket with their similarities.  0 = the legacy tail (library radix sort of all candidates, verify, flag scan, compaction), which
 * also serves batches the row path does not cover (banks with templates taller than 32 px or more than 4096 templates).
 * Results are identical in either mode. */
int focr_ctx_set_row_tail(focr_ctx_t *ctx, int mode);

/* Result sizes.  Every phase behind the scan kernel takes its element count from device memory.  A scan of the same
 * setup as the context's previous one (same bank, batch geometry, threshold, cap) bounds its buffers by the previous
 * scan's counts + a margin (4 .. 20 %, following how much consecutive counts differ), queues all phases (and a following focr_process_hits) without a host wait, and the first call
 * that needs a size (any getter, focr_sync, focr_pipe_wait) waits once; a count above its bound makes that call redo the
 * batch with exact sizes, so results never depend on the estimate.  on = 0 restores round 1's behaviour (the host
 * reads the candidate and hit counts between the phases of every scan).  Default: on. */
int focr_ctx_set_size_estimates(focr_ctx_t *ctx, int on);
/* How the size estimates have fared on this context: batches redone with exact sizes because a count exceeded its bound
 * (since the context was created), the margin the next estimated scan would add to the last counts (0.04 .. 0.2), and the
 * largest bucket of the last scan that took a row tail (hits of a page row / x-segment with the hits-first tail, candidates with round
 * 3's; 0: it took the legacy tail).  Any pointer may
 * be NULL. */
int focr_size_estimate_stats(focr_ctx_t *ctx, uint64_t *redone, double *margin, uint32_t *row_max);

/* ---- batches in flight ---------------------------------------------------
 * The executor form of the page parallelism of src/ncc.rs:839-847 (rayon
 * par_iter over pages: every worker is kept fed, no central gate).  A device
 * gets LANES streams (three by default: one batch's small kernels overlap
 * another's MFMA scan, DESIGN.md section 5: 24 -> 31 Gpx/s at configs[1]) and
 * every lane a ring of DEPTH contexts (two by default), so LANES x DEPTH
 * batches can be outstanding.  A batch is queued on the device — all of its
 * kernels, in ticket order, by one thread of the executor — the moment it is
 * submitted: batch k + LANES sits behind batch k on their lane's stream while
 * k still runs and while k's results wait to be read, the scans follow each
 * other in ticket order through a chain of device events, and no host thread
 * has to wake up between two batches (a submitting thread that pauses for a few
 * milliseconds finds the device still busy).  With more than one lane the scan
 * kernel of each context is capped to 7/8 of the CUs (focr_ctx_set_scan_cus).
 *
 *   focr_pipe_create(dev, 3, &p); focr_pipe_bank_upload(p, ...); n = focr_pipe_contexts(p);   // 6
 *   for each batch b:   if (b >= n) { wait(t[b-n], &ctx); read results from ctx; release(t[b-n]); }
 *                       focr_pipe_submit(p, pages_b, ..., &t[b]);
 *
 * Ticket t (1, 2, 3 ...) runs in context (t - 1) % n, on lane (t - 1) % LANES.
 * focr_pipe_submit blocks while that context still holds an unreleased batch.
 * pages == NULL rescans the pages already resident in that context (set up
 * through focr_pipe_context).  Host page buffers must stay valid until the
 * batch's focr_pipe_wait returns.  Batches are completed by the thread that
 * asks for them (wait / host_results / release); these may be called from a
 * different thread than submit.  Memory: every context holds its own pages,
 * scratch and results (and a second page set once host pages have gone through
 * it, see focr_pipe_prefetch) — about 0.6 GB per context at configs[1]. */
typedef struct focr_pipe focr_pipe_t;
int focr_pipe_create(int device, unsigned n_lanes, focr_pipe_t **out);  /* DEPTH = 2 (FOCR_PIPE_DEPTH in the environment overrides) */
int focr_pipe_create2(int device, unsigned n_lanes, unsigned depth, focr_pipe_t **out);  /* 1..8 lanes of 1..4 contexts */
void focr_pipe_destroy(focr_pipe_t *pipe);
unsigned focr_pipe_contexts(const focr_pipe_t *pipe);  /* LANES x DEPTH: batches that can be outstanding */
unsigned focr_pipe_lanes(const focr_pipe_t *pipe);
focr_ctx_t *focr_pipe_context(focr_pipe_t *pipe, unsigned index);  /* index < focr_pipe_contexts: the context of tickets index + 1, index + 1 + n, ... */
int focr_pipe_bank_upload(focr_pipe_t *pipe, const focr_template_t *templates, size_t n_templates,
                          const uint8_t *needles, size_t needles_len);
/* One batch = pages (host luma8 or, with pages_on_device != 0, a device
 * pointer; NULL = resident) -> focr_scan(threshold, cap, mode) -> if
 * process_hits != 0, focr_process_hits(anchor_threshold, overlap).  Host pages
 * cross PCIe on a copy stream of the batch's lane (as if announced with
 * focr_pipe_prefetch at this moment).  If chars_out != NULL (a device buffer of
 * chars_out_bytes) the batch's characters (focr_hit_t[focr_total_chars]) are
 * also copied there when the batch is completed (focr_pipe_wait), so that the
 * caller can release the context at once and still hand the characters to a
 * collective; FOCR_ERR_OVERFLOW if they do not fit. */
int focr_pipe_submit(focr_pipe_t *pipe, const void *pages, int pages_on_device, size_t n_pages, size_t r_w,
                     size_t r_h, int invert, float threshold, uint32_t cap, int mode, int process_hits,
                     float anchor_threshold, int32_t overlap, void *chars_out, size_t chars_out_bytes,
                     uint64_t *ticket);
/* Announce the host pages of a batch that will be submitted AFTER every batch announced or submitted so far, and start
 * their way to the device now: one DMA into a staging buffer of the lane the batch will run on, then the ingest (inversion,
 * pitched copies) into the ALTERNATE page set of the batch's context, both on a copy stream of the lane — under the scans of
 * the batches in flight, instead of at the head of the batch's own chain of kernels.  When the batch is queued, the context's
 * two page sets change places.  The matching focr_pipe_submit must bring the same pointer (FOCR_ERR_STATE otherwise:
 * announced batches must be submitted in the order they were announced); with another geometry or `invert` the announcement
 * is void and the batch is copied again.  At most one announcement per context: with n = focr_pipe_contexts, announce batch
 * b + n at the earliest after submitting batch b (blocks until the context's previous announced batch has been queued; the
 * executor's lock is not held meanwhile).  Page-locked memory (focr_host_alloc) makes the copy asynchronous.  The pages must
 * stay valid and unchanged until that batch's focr_pipe_wait returns.  One thread announces and submits.  Cost: the second
 * page set doubles a context's page memory (u8 + int8 copies: 138 MB per 128 pages of 608x720).  Optional: a batch that was
 * not announced takes the same road when it is submitted (src/ncc.rs:575, 880-892: the reference decodes and converts inside
 * the page loop). */
int focr_pipe_prefetch(focr_pipe_t *pipe, const void *pages, size_t n_pages, size_t r_w, size_t r_h, int invert);
/* Optional hints: nothing will be submitted behind a batch for now (the end of the host's input).  The tail kernels of that last
 * batch then take the whole GPU instead of the share the persistent scan kernel of a following batch would leave them — the
 * stream's last results arrive about a millisecond earlier; results are the same either way.  Batches are queued on the device
 * the moment they are submitted, so say it BEFORE the last submit: focr_pipe_announce_last marks the next batch submitted.
 * focr_pipe_end_of_stream, called right after the last submit, marks the newest batch if the executor has not queued it yet
 * (it normally has: kept for hosts that learn of the end only afterwards). */
int focr_pipe_announce_last(focr_pipe_t *pipe);
int focr_pipe_end_of_stream(focr_pipe_t *pipe);
/* Results on the host: with fetch on, completing a batch (focr_pipe_host_results / _wait / _release) also copies its
 * per-(page, template) counts and, if process_hits ran, its lines (focr_get_lines layout) into page-locked memory of its
 * context, on the lane's side stream; focr_pipe_host_results waits for the batch like focr_pipe_wait and hands out
 * pointers that stay valid until focr_pipe_release.  Set fetch before the first submit. */
typedef struct focr_host_results {
    const uint32_t *counts;        /* [n_pages][n_templates] */
    const uint64_t *page_line_off; /* [n_pages + 1]; NULL without process_hits */
    const uint64_t *line_char_off; /* [n_lines + 1] */
    const focr_hit_t *chars;       /* [n_chars] */
    size_t n_pages, n_templates, n_matches, n_lines, n_chars;
    float device_ms;               /* device time of the batch's scan + process_hits */
} focr_host_results_t;
int focr_pipe_set_fetch(focr_pipe_t *pipe, int on);
int focr_pipe_host_results(focr_pipe_t *pipe, uint64_t ticket, focr_host_results_t *out);
/* Blocks until the batch is done; returns its status and the context that
 * holds its results (all getters of this header apply). */
int focr_pipe_wait(focr_pipe_t *pipe, uint64_t ticket, focr_ctx_t **ctx);
/* Where a ticket's time went (after its focr_pipe_wait, before its release): host stamps in microseconds since the executor
 * was created, and the device-side interval between the previous ticket's last kernel and this one's (HIP events; < 0 when the
 * previous ticket was not available: first ticket, retired out of order).  A host that sees low throughput can tell a late
 * submit (submit_us far behind the previous ticket's), a late executor (enqueue_*), a slow device (device_gap_ms) and a late
 * consumer (done_us: when the waiting thread saw the batch complete) apart. */
typedef struct focr_ticket_times {
    double submit_us, enqueue_begin_us, scan_queued_us, enqueue_end_us, done_us;
    float device_gap_ms;
} focr_ticket_times_t;
int focr_pipe_ticket_times(focr_pipe_t *pipe, uint64_t ticket, focr_ticket_times_t *out);
/* The context may take its next batch; the results of `ticket` are gone. */
int focr_pipe_release(focr_pipe_t *pipe, uint64_t ticket);

/* ---- every GPU of the node ------------------------------------------------
 * The same executor over several devices (the page parallelism of src/ncc.rs:839-847 uses every core the host has; this
 * uses every GPU): one focr_pipe per device, created and given the bank in parallel; batch k (in submission order) goes to
 * device k % n_devices and, there, to the next context.  Tickets are the fleet's own, 1, 2, 3 ... in submission order; retire
 * them in that order and the output order is the submission order whatever the device count.  No collective: results
 * converge on the host that consumes them (a device-resident consumer uses focr_rccl.h).  `devices` == NULL or
 * n_devices == 0: all visible devices.  At most focr_fleet_slots() = n_devices * lanes_per_device * DEPTH batches are
 * outstanding: with every context holding an unreleased batch focr_fleet_submit returns FOCR_ERR_STATE ("release the oldest
 * ticket first") instead of waiting for a release that a single-threaded consumer could never make; with
 * pages_on_device != 0 the pointer must belong to focr_fleet_device_of(ticket it will get) — host pages are the normal case.
 * The `ncc` binary is this loop. */
typedef struct focr_fleet focr_fleet_t;
int focr_fleet_create(const int *devices, unsigned n_devices, unsigned lanes_per_device, focr_fleet_t **out);
void focr_fleet_destroy(focr_fleet_t *fleet);
unsigned focr_fleet_devices(const focr_fleet_t *fleet);
unsigned focr_fleet_lanes(const focr_fleet_t *fleet);
unsigned focr_fleet_slots(const focr_fleet_t *fleet);  /* batches that can be outstanding over all devices */
focr_pipe_t *focr_fleet_pipe(focr_fleet_t *fleet, unsigned index);  /* the executor of the index-th device */
int focr_fleet_device_of(const focr_fleet_t *fleet, uint64_t ticket);  /* HIP device index ticket maps / will map to */
int focr_fleet_bank_upload(focr_fleet_t *fleet, const focr_template_t *templates, size_t n_templates,
                           const uint8_t *needles, size_t needles_len);
int focr_fleet_set_fetch(focr_fleet_t *fleet, int on);
int focr_fleet_announce_last(focr_fleet_t *fleet);  /* the last n_devices batches end their devices' streams: call before submitting them (focr_pipe_announce_last on every executor) */
int focr_fleet_end_of_stream(focr_fleet_t *fleet);  /* focr_pipe_end_of_stream on every device's executor */
int focr_fleet_submit(focr_fleet_t *fleet, const void *pages, int pages_on_device, size_t n_pages, size_t r_w, size_t r_h,
                      int invert, float threshold, uint32_t cap, int mode, int process_hits, float anchor_threshold,
                      int32_t overlap, uint64_t *ticket);
int focr_fleet_wait(focr_fleet_t *fleet, uint64_t ticket, focr_ctx_t **ctx);
int focr_fleet_host_results(focr_fleet_t *fleet, uint64_t ticket, focr_host_results_t *out);
int focr_fleet_release(focr_fleet_t *fleet, uint64_t ticket);

/* Per-launch record of the scan kernels of the last focr_scan (one entry per
 * (size class, bank chunk) launch), timed with HIP events on the stream the
 * kernel ran on.  alg_macs: true template area x searched windows x templates
 * of that launch; issued_macs: MACs the launch issued including padding. */
typedef struct focr_launch_info {
    char name[64];
    float ms;
    uint32_t n_templates;
    uint64_t alg_macs;
    uint64_t issued_macs;
} focr_launch_info_t;
/* Copies up to cap records, returns the number of launches of the last scan. */
size_t focr_last_launches(focr_ctx_t *ctx, focr_launch_info_t *out, size_t cap);

/* Device self-test hook used by the parity tests: evaluates
 * 1/sqrt((double)s2 - (double)(s*s)/(double)n) on the device for n_items
 * triples, so the f64 divide/sqrt rounding can be compared with the host's. */
int focr_debug_rnorm(focr_ctx_t *ctx, const uint32_t *s, const uint64_t *s2, const uint32_t *n,
                     size_t n_items, double *out);

/* Test hook: make focr_scan take its candidate-overflow fallback (the batch scanned in page sub-ranges and appended)
 * without waiting for an overflow. */
int focr_debug_force_split(focr_ctx_t *ctx, int on);

/* Diagnostic: the phases of the context's last batch on the device's clock, in milliseconds since a per-device origin (the creation of
 * the device's first context): [0] statistics start, [1] statistics end, [2] scan kernels end, [3] verify end, [4] ordering end,
 * [5] process_hits start, [6] process_hits end, [7] start and [8] end of the scan launch with the most work; -1 where not available.
 * What a kernel trace shows, without a profiler in the process. */
int focr_debug_phase_stamps(focr_ctx_t *ctx, double out[9]);

/* Test hook: launch the persistent kernels of the scan's tail (exact verify in its list and chunk forms, hit scatter, row sort) on
 * num / den times the workgroups they are designed for (0 / 0: as designed).  Results must be identical for every grid. */
int focr_debug_set_tail_grid(focr_ctx_t *ctx, uint32_t num, uint32_t den);

/* Test hooks for the window statistics: form 1 = the LDS-tiled kernel for every size class (0: the register form for classes whose
 * kept width is 8 px, scan_mfma.hip); focr_debug_planes copies the int16 threshold planes of the context's last MFMA scan to the
 * host ([value][page][Lrows][Lpitch] per pass, Lpitch = (r_w + 63) / 64 * 64 + 64, Lrows = (r_h + 7) / 8 * 8 + 8; out = NULL: only
 * their number).  Both forms must write the same planes wherever the scan kernel reads them. */
int focr_debug_set_stats_form(focr_ctx_t *ctx, int form);
int focr_debug_planes(focr_ctx_t *ctx, uint16_t *out, size_t capacity, size_t *n_values);

/* Host model of the MFMA prefilter's bound (no device needed; used by the CPU tests, tests/test_prefilter_host.py): builds
 * the quantised bank exactly as focr_bank_upload does and evaluates, for n_windows caller-supplied ink-high patches of
 * frame_w x frame_h bytes (row-major; every template must fit the frame, its window is the frame's top-left n_w x n_h
 * box), per (window, template):
 *   sim[w * n_templates + t]   the exact similarity (double; NaN where the reference cannot emit: zero variance)
 *   d[w * n_templates + t]     G + C-in as the device forms them: the int8 MFMA sum over the kept columns plus
 *                              the window's plane value (the statistics kernel's arithmetic: f32, then -floor((L - 2) / S)
 *                              as int16) times S; the pair is a candidate iff d > 0
 * column_drop: as focr_ctx_set_column_drop.  info[4 * k ..] = {c_scale, e_max, rho_max, kept width} of size class k (in
 * order of first appearance), n_info = capacity of info in doubles. */
int focr_debug_prefilter(const focr_template_t *templates, size_t n_templates, const uint8_t *needles, size_t needles_len,
                         int column_drop, const uint8_t *windows, size_t n_windows, uint32_t frame_w, uint32_t frame_h,
                         float threshold, double *sim, int64_t *d, double *info, size_t n_info);

/* The threshold planes' values (mfma_common.h: -floor((L - 2) / 2^shift) as int16, clamped to +-32767), host flavour, for the
 * CPU tests. */
void focr_debug_plane_value(const float *L, size_t n, uint32_t shift, int16_t *out);
/* The same as the device computes them (the GPU tests compare the two bit for bit). */
int focr_debug_plane_value_device(focr_ctx_t *ctx, const float *L, size_t n, uint32_t shift, int16_t *out);

#ifdef __cplusplus
}
#endif
#endif /* FOCR_NCC_H */
