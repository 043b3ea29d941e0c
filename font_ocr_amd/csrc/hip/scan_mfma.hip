// scan_mfma.hip — the fast scan: i8 MFMA conservative prefilter + exact verify.
//
// The reference evaluates, for every window w and template t (src/ncc.cpp:302-392),
//     sim = num / (norm_n * norm_p),  num = sum_k a_k b_k - s_n s_p / n = sum_k a_k (b_k - mean_t)
// and emits iff sim > thr.  Almost no (w, t) pair passes, so the device splits the work:
//
//  1. window statistics (stats_kernel): per size class and window, the exact integer sums s_p, s2_p, V = n*s2 - s^2 (V > 0 <=> the
//     reference's rnorm is finite, src/ncc.rs:309-311) and — for a class whose last column the MFMA does not multiply — that
//     column's sums.  From them the window's prefilter THRESHOLD L(w) in f32 (mfma_common.h, "threshold planes"), stored as the
//     MFMA's C-in in units of a per-class power of two, rounded TOWARDS -INF as a threshold ("threshold plane", an int16 per window
//     and class); the most negative value where the reference never emits (x = 0, y = 0, out of range, zero variance => rnorm = inf/NaN).  A lower threshold only admits more
//     candidates, so the directed rounding has no sign cases: the filter is conservative for negative --threshold too (round 2
//     stored the window norm rounded towards zero and multiplied by kappa in the scan kernel, which RAISED the threshold for
//     kappa < 0).  (Legacy form, still used for size classes with more than 4 K-steps: negL(w) = -(floor(L) - 2) as int32, or
//     -REJECT.)  The kernel also marks every 16-window M-tile that has a live window; compact_live_tiles makes the work list.
//  2. MFMA prefilter (scan_mfma2.hip): every template is mean-centred, scaled by a class-wide constant c/norm_n(t) and rounded
//     to int8 with the rounding chosen so that sum_k bq_k = 0.  G(w,t) = sum_k (a_k - 128) bq_k  (= sum_k a_k bq_k) is one
//     v_mfma_i32_16x16x64_i8 chain over the window's bytes (16 templates x 16 windows, K = 64 bytes per instruction) with
//     C-in = plane value << log2(S), so "D > 0" <=> G > L(w) rounded down to a multiple of S.  Cauchy-Schwarz bounds the rounding error:
//         | c*num/norm_n - G | = | sum_k (a_k - mean_w) e_k | <= norm_p * ||e_t||_2
//     hence sim > thr  ==>  G > (c*thr - max_t ||e_t||) * norm_p =: kappa * norm_p; kappa carries an extra relative margin for
//     the f64 roundings of the exact formula.  Classes 9 or 13 px wide leave their last column to a second Cauchy-Schwarz term,
//     L(w) = kappa * norm_p(w) - c * rho_max * dnorm(w), and take the next narrower K layout (column drop, mfma_common.h).  The
//     filter has no false negatives (host model of the device arithmetic: prefilter_model.hip, tests/test_prefilter_host.py).
//     Survivors go to a candidate list.
//  3. the hits-first row tail (rows.hip): candidates verified exactly where they lie — the reference formula, operation for
//     operation (verify_candidate, mfma_common.h / common.h) — hits bucketed by page row and sorted per bucket; order.hip derives
//     the per-call ranks and the cap.  (verify_kernel below + the library radix sort = the legacy tail, kept as a fallback.)
//
// Sizes: every phase behind the scan kernel takes its element count from device memory; exact / estimated mode: see
// launch_scan_mfma and ctx.hip (finish_results).
//
// Layout: one operand = windows (fragments straight from the page's int8 copy), the other = the quantised bank staged once per
// block in LDS in exactly the per-lane order the MFMA wants; the K layouts (how image rows map to 16-byte k-groups) are
// in mfma_common.h.  The scan kernels are in scan_mfma2.hip.
#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstring>
#include <mutex>

#include "mfma_common.h"

namespace focr {

int ensure_hit_capacity(focr_ctx *c, size_t want);
int sort_keys_u64(focr_ctx *c, uint64_t *&keys, uint64_t *&keys_alt, size_t n, unsigned end_bit);
int launch_scan_tall(focr_ctx *c, size_t k, double thr_d, uint64_t *keys, float *sims, unsigned long long *counter,
                     unsigned long long capacity, int rust);
int compact_candidates(focr_ctx *c, const uint64_t *keys, const float *sims, const uint64_t *flags, uint64_t *pos,
                       const unsigned long long *n_cand_p, size_t ub_c);
int order_sorted_hits(focr_ctx *c, uint64_t *hkeys, float *hsims, const uint64_t *n_p, size_t ub, const unsigned long long *n_cand_p, size_t ub_c);
// rows.hip: the row path of the tail
bool rows_applicable(const focr_ctx *c);
uint32_t rows_capacity_for(uint64_t row_max);
int rows2_begin(focr_ctx *c, ClearList &clear);  // the hits-first tail (rows.hip): verify in flush order, then only hits are placed and sorted
int rows2_verify(focr_ctx *c, double thr_d, const unsigned long long *n_cand_p, size_t ub_c);
int rows2_place(focr_ctx *c, const unsigned long long *n_cand_p, size_t ub_c, size_t ub_h, bool big_expected, bool sort);
int sort_pairs_u64_f32(focr_ctx *c, uint64_t *&keys, uint64_t *&keys_alt, float *&vals, float *&vals_alt, size_t n, unsigned end_bit);


// ---------------------------------------------------------------------------------------------
// 1. window statistics -> threshold planes (or the legacy int32 tables)
//
// Everything a scan needs zeroed, in one launch (ClearList, common.h)
__global__ __launch_bounds__(256) void clear_kernel(const ClearList l) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x, step = gridDim.x * 256;
    for (uint32_t r = 0; r < l.n; r++) {
        uint64_t *p = reinterpret_cast<uint64_t *>(l.p[r]);
        for (uint32_t i = tid; i < l.n8[r]; i += step) p[i] = 0;
    }
}
int launch_clear(focr_ctx *c, const ClearList &l) {
    size_t words = 0;
    for (uint32_t r = 0; r < l.n; r++) words += l.n8[r];
    if (!words) return FOCR_OK;
    hipLaunchKernelGGL(clear_kernel, dim3((unsigned)std::min<size_t>(1024, (words + 1023) / 1024)), dim3(256), 0, c->stream, l);
    FOCR_HIP(c, hipGetLastError());
    return FOCR_OK;
}

// Separable sliding sums: a block stages a (64 + n_w) x (32 + n_h - 1) byte tile, computes the horizontal
// n_w-sums H (and H2 of squares) of every tile row once (v_dot4 on masked dwords), then each thread slides a
// vertical n_h-window down its column: S(y+1) = S(y) + H(y+n_h) - H(y).  ~40 instructions per window instead
// of ~300 for the direct evaluation.  Everything up to the square roots is exact integer arithmetic (V, W of
// mfma_common.h, "threshold planes"); window is live  <=>  V > 0, which is exactly the reference's "norm > 0"
// (src/ncc.rs:309: (f64)s2 - (f64)(s*s)/(f64)n is > 0 iff V > 0, = 0 iff V = 0, because V/n >= 1/n is far above the
// rounding error of the division).
//   DROP: the class's last column is bounded, not multiplied: the horizontal sums run over the KEPT width (round 5: two dwords
//         instead of three for BASELINE configs[1]'s 9-wide class) and the thread also slides the sums of the dropped column (bytes
//         of the tile) down; the full box's sums are the two added, W comes from both.
//   PAIR: the kept box is itself a size class of the pass (BASELINE configs[1]: 8x15 beside 9x15): its plane comes out of
//         the same launch (its statistics are the kept box's), one launch instead of two.
constexpr int STX = 64, STY = 32, SLDW = 21;  // 32 window rows per block (64 = less halo, fewer blocks per CU: measured, no gain)
static inline size_t stats_lds_bytes(uint32_t n_h) { return (size_t)(STY + n_h - 1) * (SLDW * 4 + STX * 4 + STX * 2); }

struct StatsOut {  // what a statistics launch writes for one size class
    PlaneParams p;
    void *out;     // OUT = 1: int16 threshold plane, OUT = 0: int32 negL table; [page][Lrows][Lpitch]
};

// IDX: uint32_t on the plane path (a pass's planes span < 4 GiB: launch_scan_mfma), size_t for the int32 tables
template <int OUT, typename IDX>
__device__ __forceinline__ void stats_store(const StatsOut &o, IDX idx, bool emit, float Lf) {
    if (OUT) {
        // a pass's planes span < 4 GiB (launch_scan_mfma), so the entry's BYTE offset fits 32 bits: uniform base + 32-bit lane offset,
        // no 64-bit vector add per store
        const uint32_t byte_off = (uint32_t)idx * 2u;
        *reinterpret_cast<int16_t *>(reinterpret_cast<char *>(o.out) + byte_off) = emit ? plane_value(o.p, Lf) : PLANE_NEVER;
    } else {
        reinterpret_cast<int32_t *>(o.out)[idx] = emit ? threshold_negL(Lf) : -REJECT;
    }
}

template <int NDW, bool SMALLN, int OUT, bool DROP, bool PAIR>  // NDW: dwords of the KEPT width (n_w, or n_w - 1 with DROP)
__global__ __launch_bounds__(256) void stats_kernel(const uint8_t *__restrict__ pages, uint32_t pitch, uint32_t rows_alloc,
                                                    uint32_t r_w, uint32_t r_h, uint32_t n_w, uint32_t n_h, const StatsOut A,
                                                    const StatsOut B, uint32_t Lpitch, uint32_t Lrows,
                                                    uint8_t *__restrict__ live, uint32_t mtx, uint32_t n_rows) {
    // dynamic LDS, sized for this class's n_h (stats_lds_bytes): ~21 KB at n_h = 15 -> 7 blocks per CU; the kernel
    // lives on that occupancy (global-load latency, two barriers per tile)
    extern __shared__ uint32_t stats_lds[];
    const uint32_t page = blockIdx.z, x0 = blockIdx.x * STX, y0 = blockIdx.y * STY;
    const uint32_t rows = STY + n_h - 1;
    uint32_t (*tile)[SLDW] = reinterpret_cast<uint32_t (*)[SLDW]>(stats_lds);
    uint32_t (*H2)[STX] = reinterpret_cast<uint32_t (*)[STX]>(stats_lds + rows * SLDW);
    uint16_t (*H)[STX] = reinterpret_cast<uint16_t (*)[STX]>(stats_lds + rows * (SLDW + STX));  // row sums <= 16 * 255
    const uint8_t *pg = pages + (size_t)page * rows_alloc * pitch;
    uint32_t any_ink = 0;
    for (uint32_t i = threadIdx.x; i < rows * SLDW; i += 256) {
        uint32_t r = i / SLDW, cdw = i % SLDW;
        uint32_t gy = y0 + r, gx = x0 + cdw * 4;
        uint32_t v = 0;
        if (gy < rows_alloc && gx + 4 <= pitch) v = *reinterpret_cast<const uint32_t *>(pg + (size_t)gy * pitch + gx);
        tile[r][cdw] = v;
        any_ink |= v;
    }
    // Blank paper under the whole tile (page margins: ~1 block in 10): every window here has zero variance — "never emits",
    // no M-tile marked live — so the sliding sums are skipped and the entries just say so (an M-tile of this block can still be
    // live through another size class of the pass, whose launch stages a wider tile: its reads must find a defined value).
    if (!__syncthreads_or((int)(any_ink != 0))) {
        const uint32_t col = threadIdx.x & 63, x = x0 + col;
        if (x < Lpitch)
            for (uint32_t k = 0; k < STY / 4; k++) {
                const uint32_t y = y0 + (threadIdx.x >> 6) * (STY / 4) + k;
                if (y >= Lrows) break;
                const size_t idx = ((size_t)page * Lrows + y) * Lpitch + x;
                stats_store<OUT, size_t>(A, idx, false, 0.f);
                if (PAIR) stats_store<OUT, size_t>(B, idx, false, 0.f);
            }
        return;
    }
    const uint32_t kw = DROP ? n_w - 1 : n_w;  // the kept width: what the horizontal sums cover
    {  // horizontal sums
        const uint32_t lane = threadIdx.x & 63, cb = lane >> 2, sh = lane & 3;
        for (uint32_t r = threadIdx.x >> 6; r < rows; r += 4) {
            uint32_t h = 0, h2 = 0;
#pragma unroll
            for (int k = 0; k < NDW; k++) {
                uint32_t lo = tile[r][cb + k], hi = tile[r][cb + k + 1];
                uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, sh);
                uint32_t keep = kw >= (uint32_t)(4 * k + 4) ? 0xffffffffu
                                : (kw <= (uint32_t)(4 * k) ? 0u : ((1u << (8 * (kw - 4 * k))) - 1u));
                w &= keep;
                h = __builtin_amdgcn_udot4(w, 0x01010101u, h, false);
                h2 = __builtin_amdgcn_udot4(w, w, h2, false);
            }
            H[r][lane] = (uint16_t)h;
            H2[r][lane] = h2;
        }
    }
    __syncthreads();
    // the wave's strip of window rows as a scalar: row numbers, LDS row offsets and the "row exists" tests below stay out of the vector unit
    const uint32_t col = threadIdx.x & 63, strip = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t x = x0 + col;
    if (x >= Lpitch) return;
    constexpr uint32_t PER = STY / 4;  // window rows per thread
    const uint32_t r0 = strip * PER;
    // the class's last column as bytes of the tile (DROP): column x + n_w - 1 of the page = byte col + n_w - 1 of a tile row
    const uint8_t *lastc = reinterpret_cast<const uint8_t *>(&tile[0][0]) + col + n_w - 1;
    uint32_t s_k = 0, s2_k = 0, q1 = 0, q2 = 0;  // sums of the kept box and of the dropped column
    for (uint32_t j = 0; j < n_h; j++) {
        s_k += H[r0 + j][col];
        s2_k += H2[r0 + j][col];
        if (DROP) {
            const uint32_t b = lastc[(size_t)(r0 + j) * (SLDW * 4)];
            q1 += b;
            q2 += __umul24(b, b);
        }
    }
    const uint32_t n = n_w * n_h, n_k = (n_w - 1) * n_h;
    // searched windows: x in [1, r_w - n_w], y in [1, r_h - n_h]  (src/ncc.rs:279-282, src/ncc.cpp:302)
    const bool x_ok = x >= 1 && x + n_w <= r_w, xk_ok = x >= 1 && x + n_w - 1 <= r_w;
    const uint32_t ya = y0 + r0;
    // M-tile marks: one store per 16-lane group and window row, decided by ballot (blank paper is never scanned;
    // the reference prunes it too, src/ncc.rs:280-301).  Row y of the image is tile row y - 1.
    const bool mark_lane = (col & 15) == 0 && (x >> 4) < mtx;
    const uint32_t live_i = (page * n_rows + ya) * mtx + (x >> 4);  // entry of image row ya + 1 (a pass has < 2^31 M-tiles: launch_scan_mfma)
    typedef typename std::conditional<OUT == 1, uint32_t, size_t>::type idx_t;
    idx_t idx = ((idx_t)page * Lrows + ya) * Lpitch + x;  // the window's entry in the planes; one row further per step
#pragma unroll
    for (uint32_t k = 0; k < PER; k++, idx += Lpitch) {
        const uint32_t y = ya + k;
        if (y < Lrows) {
            // V = n*s2 - s*s, exact; V > 0 <=> the reference's rnorm is finite.  SMALLN (n <= 256): both products
            // fit 32 bits (n*s2 <= n^2 * 255^2 < 2^32, s <= 255 n < 2^16).
            const uint32_t s = DROP ? s_k + q1 : s_k, s2 = DROP ? s2_k + q2 : s2_k;  // the full box
            bool nz;
            float Vf;
            if (SMALLN) {  // n <= 256: n, s, s2 < 2^24 -> full-rate 24-bit multiplies (a 32-bit v_mul_lo is a quarter-rate instruction)
                const uint32_t V = __umul24(n, s2) - __umul24(s, s);
                nz = V != 0;
                Vf = (float)V;
            } else {
                const uint64_t V = (uint64_t)n * s2 - (uint64_t)s * s;
                nz = V != 0;
                Vf = (float)V;
            }
            const bool y_ok = y >= 1 && y + n_h <= r_h;
            const bool emit = x_ok && y_ok && nz;
            bool any = emit;
            stats_store<OUT, idx_t>(A, idx, emit, DROP ? threshold_f32(A.p, Vf, dropped_column_W_upper(n_k, n_h, s_k, q1, q2)) : threshold_f32_nodrop(A.p, Vf));
            if (PAIR) {  // the kept box as a size class of its own: (n_w - 1) x n_h, nothing dropped
                bool nzk;
                float Vkf;
                if (SMALLN) {
                    const uint32_t Vk = __umul24(n_k, s2_k) - __umul24(s_k, s_k);
                    nzk = Vk != 0;
                    Vkf = (float)Vk;
                } else {
                    const uint64_t Vk = (uint64_t)n_k * s2_k - (uint64_t)s_k * s_k;
                    nzk = Vk != 0;
                    Vkf = (float)Vk;
                }
                const bool emit_k = xk_ok && y_ok && nzk;
                any |= emit_k;
                stats_store<OUT, idx_t>(B, idx, emit_k, threshold_f32_nodrop(B.p, Vkf));
            }
            const uint64_t lm = __builtin_amdgcn_ballot_w64(any);
            if (mark_lane && ((lm >> col) & 0xffffu) && y >= 1 && y <= n_rows) live[live_i + k * mtx - mtx] = 1;
        }
        if (k + 1 < PER) {  // slide down one row
            s_k += H[r0 + k + n_h][col] - H[r0 + k][col];
            s2_k += H2[r0 + k + n_h][col] - H2[r0 + k][col];
            if (DROP) {
                const uint32_t bi = lastc[(size_t)(r0 + k + n_h) * (SLDW * 4)], bo = lastc[(size_t)(r0 + k) * (SLDW * 4)];
                q1 += bi - bo;
                q2 += __umul24(bi, bi) - __umul24(bo, bo);
            }
        }
    }
}

// The same statistics for kept widths of 4, 8, 12 and 16 columns (8: BASELINE configs[1] and [2], 8-wide classes and 9-wide ones with
// their last column dropped; the description below is for 8, template parameter KQ = 2), VERTICAL sums first and no LDS: a lane owns four neighbouring columns (one dword of every page row), slides the
// n_h-row sums of its four columns down S8_ROWS window rows — C1 = sum of bytes, C2 = sum of squares, from the row that enters and
// the row that leaves: d = in - out, C1 += d, C2 += d * (in + out) — and the horizontal 8-sums come out of the lanes' registers:
// window x = 4L + i covers columns 4L + i .. 4L + i + 7 = the rest of lane L's dword, all of lane L + 1's, the first i columns
// of lane L + 2's; the dropped ninth column is column i of lane L + 2.  Two row sums from lane L + 1 and eight column sums from
// lane L + 2 per row (ds_bpermute), no tile staging, no barrier, four plane values per 8-byte store.  A wave is a strip of 240
// window columns (lanes 60..63 only feed their neighbours) x S8_ROWS rows of one page; a row of the strip whose 8 + 256 columns
// are blank over the n_h rows (every C2 zero) stores "never" without the arithmetic.  Results: the same exact integers s, s2, q1,
// q2 as stats_kernel, then the same code — plane for plane identical (tests/test_gpu_parity.py: the planes of both kernels, and every parity test).
constexpr uint32_t S8_COLS = 240, S8_ROWS = 16;  // (8 / 24 / 32 rows per wave: 133 / 130 / 141 us against 130 before the rows were prefetched; 16 and 24 level after)
//   APPEND: the launch is the only statistics launch of its scan pass (BASELINE configs[1]: both classes in one PAIR launch), so a
//   marked M-tile is final: instead of a mark byte for compact_live_tiles the wave remembers its marks (16 rows x 15 M-tiles: one
//   bit per row in each quad's first lane) and appends them to the pass's work list itself, in row-major order, behind ONE atomic
//   per workgroup — no mark bytes, no compaction launch between the statistics and the scan kernel.
//   KQ = kept width / 4 (1 .. 4: kept widths 4, 8, 12, 16): window 4L + i then covers the rest of lane L's dword, lanes L + 1 .. L + KQ - 1
//   whole and the first i columns of lane L + KQ, and the dropped column is column i of lane L + KQ.
//   A workgroup is GS neighbouring strips x GB bands one below the other (up to 16 waves), and it appends in the order (band, window
//   row, strip): the work list then holds a page in blocks of whole page rows, GB x 16 rows tall, as compact_live_tiles' did (4 096
//   M-tiles per block) — the scan kernel's neighbouring items share the page rows their windows overlap in and the planes' cache lines.
template <int KQ, bool SMALLN, bool DROP, bool PAIR, bool APPEND>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void stats8_kernel(const uint8_t *__restrict__ pages, uint32_t pitch, uint32_t rows_alloc, uint32_t r_w, uint32_t r_h,
                                                      uint32_t n_w, uint32_t n_h, const StatsOut A, const StatsOut B, uint32_t Lpitch, uint32_t Lrows,
                                                      uint8_t *__restrict__ live, uint32_t mtx, uint32_t n_rows, uint32_t strips_x, uint32_t bands_y,
                                                      uint32_t GS, uint32_t GB, uint32_t sgroups, uint32_t bgroups, uint64_t *__restrict__ list,
                                                      uint32_t *__restrict__ list_count) {
    __shared__ uint32_t wg_cnt[16][S8_ROWS], wg_off[16][S8_ROWS];  // [wave][window row]: live M-tiles, and where they go inside the workgroup's block
    __shared__ uint32_t wg_wsum[4];
    __shared__ uint32_t wg_base;
    const uint32_t lane = threadIdx.x & 63, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t per_page = sgroups * bgroups, in_page = blockIdx.x % per_page;
    const uint32_t band = (in_page / sgroups) * GB + wv / GS, strip = (in_page % sgroups) * GS + wv % GS;
    uint32_t mymask = 0, page = blockIdx.x / per_page, y0 = band * S8_ROWS, xl = 0;  // APPEND: bit k = the M-tile of this quad is live in window row y0 + k
    if (strip < strips_x && band < bands_y) {  // wave-uniform (the workgroup's last strips / bands may lie outside the page)
    const uint32_t x0 = strip * S8_COLS;
    xl = x0 + 4 * lane;  // the lane's first column = its first window
    // lanes right of the row read its zero padding (>= 64 zero bytes right of every row, focr_pages_alloc)
    const uint32_t off = xl + 4 <= pitch ? xl : pitch - 4;
    const uint8_t *pg = pages + (size_t)page * rows_alloc * pitch;
    auto load_row = [&](uint32_t y) -> uint32_t {  // wave-uniform row test (pages end with >= 48 zero rows: never taken at the sizes the MFMA path covers)
        return y < rows_alloc ? *reinterpret_cast<const uint32_t *>(pg + (size_t)y * pitch + off) : 0u;
    };
    uint32_t c1[4] = {0, 0, 0, 0}, c2[4] = {0, 0, 0, 0};
    for (uint32_t j = 0; j < n_h; j += 8) {  // the first window row's sums: eight page rows per round trip to memory
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = load_row(y0 + j + i);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (j + i >= n_h) v[i] = 0;  // wave-uniform
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t b = (v[i] >> (8 * m)) & 0xffu;
                c1[m] += b;
                c2[m] += __umul24(b, b);
            }
        }
    }
    int an[KQ + 1];  // ds_bpermute addresses of lanes L + 1 .. L + KQ
#pragma unroll
    for (int q = 1; q <= KQ; q++) an[q] = (int)(lane + q < 64 ? lane + q : 63) * 4;
    const uint32_t n = n_w * n_h, n_k = (n_w - 1) * n_h;
    const bool store_lane = lane < S8_COLS / 4 && xl < Lpitch;
    const bool mark_lane = (lane & 3) == 0 && lane < S8_COLS / 4 && (xl >> 4) < mtx;  // a quad's first lane speaks for its M-tile; lanes 60..63 belong to the next strip
    char *outA = reinterpret_cast<char *>(A.out), *outB = reinterpret_cast<char *>(B.out);
    const uint32_t k_end = y0 < Lrows ? (Lrows - y0 < S8_ROWS ? Lrows - y0 : S8_ROWS) : 0u;  // wave-uniform: the band's rows inside the planes
    auto window_row = [&](uint32_t k) __attribute__((always_inline)) {
        const uint32_t y = y0 + k;
        // the rows that enter and leave when the window slides down: asked for now, used behind this row's arithmetic
        const uint32_t vi = load_row(y + n_h), vo = load_row(y);
        const uint32_t entry = ((page * Lrows + y) * Lpitch + xl) * 2u;  // byte offset of the lane's four values in a plane (a pass's planes span < 4 GiB)
        const bool y_ok = y >= 1 && y + n_h <= r_h;
        const uint32_t nzc = c2[0] | c2[1] | c2[2] | c2[3];
        if (__builtin_amdgcn_ballot_w64(nzc != 0) == 0 || !y_ok) {
            // nothing but paper under the strip's windows of this row (or a row the reference never searches): "never", no marks
            if (store_lane) {
                const uint32_t nv = (uint32_t)(uint16_t)PLANE_NEVER * 0x10001u;
                *reinterpret_cast<uint2 *>(outA + entry) = uint2{nv, nv};
                if (PAIR) *reinterpret_cast<uint2 *>(outB + entry) = uint2{nv, nv};
            }
        } else {
            const uint32_t R1 = c1[0] + c1[1] + c1[2] + c1[3], R2 = c2[0] + c2[1] + c2[2] + c2[3];
            uint32_t s_k = R1, s2_k = R2;
#pragma unroll
            for (int q = 1; q < KQ; q++) {  // the whole lanes between
                s_k += (uint32_t)__builtin_amdgcn_ds_bpermute(an[q], (int)R1);
                s2_k += (uint32_t)__builtin_amdgcn_ds_bpermute(an[q], (int)R2);
            }
            uint32_t e1[4], e2[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                e1[m] = (uint32_t)__builtin_amdgcn_ds_bpermute(an[KQ], (int)c1[m]);
                e2[m] = (uint32_t)__builtin_amdgcn_ds_bpermute(an[KQ], (int)c2[m]);
            }
            int16_t va[4], vb[4];
            bool any = false;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t x = xl + i, q1 = e1[i], q2 = e2[i];
                const bool x_ok = x >= 1 && x + n_w <= r_w, xk_ok = x >= 1 && x + n_w - 1 <= r_w;
                const uint32_t s = DROP ? s_k + q1 : s_k, s2 = DROP ? s2_k + q2 : s2_k;  // the full box
                bool nz;
                float Vf;
                if (SMALLN) {
                    const uint32_t V = __umul24(n, s2) - __umul24(s, s);
                    nz = V != 0;
                    Vf = (float)V;
                } else {
                    const uint64_t V = (uint64_t)n * s2 - (uint64_t)s * s;
                    nz = V != 0;
                    Vf = (float)V;
                }
                const bool emit = x_ok && nz;
                any |= emit;
                const float La = DROP ? threshold_f32(A.p, Vf, dropped_column_W_upper(n_k, n_h, s_k, q1, q2)) : threshold_f32_nodrop(A.p, Vf);
                va[i] = emit ? plane_value(A.p, La) : PLANE_NEVER;
                if (PAIR) {
                    bool nzk;
                    float Vkf;
                    if (SMALLN) {
                        const uint32_t Vk = __umul24(n_k, s2_k) - __umul24(s_k, s_k);
                        nzk = Vk != 0;
                        Vkf = (float)Vk;
                    } else {
                        const uint64_t Vk = (uint64_t)n_k * s2_k - (uint64_t)s_k * s_k;
                        nzk = Vk != 0;
                        Vkf = (float)Vk;
                    }
                    const bool emit_k = xk_ok && nzk;
                    any |= emit_k;
                    vb[i] = emit_k ? plane_value(B.p, threshold_f32_nodrop(B.p, Vkf)) : PLANE_NEVER;
                }
                s_k += e1[i] - c1[i];  // one column to the right
                s2_k += e2[i] - c2[i];
            }
            if (store_lane) {
                *reinterpret_cast<uint2 *>(outA + entry) = uint2{(uint32_t)(uint16_t)va[0] | ((uint32_t)(uint16_t)va[1] << 16), (uint32_t)(uint16_t)va[2] | ((uint32_t)(uint16_t)va[3] << 16)};
                if (PAIR) *reinterpret_cast<uint2 *>(outB + entry) = uint2{(uint32_t)(uint16_t)vb[0] | ((uint32_t)(uint16_t)vb[1] << 16), (uint32_t)(uint16_t)vb[2] | ((uint32_t)(uint16_t)vb[3] << 16)};
            }
            // M-tile marks: an M-tile is the 16 windows of four lanes (x0 is a multiple of 16)
            const uint64_t lm = __builtin_amdgcn_ballot_w64(any && store_lane);
            if (mark_lane && ((lm >> lane) & 0xfu) && y <= n_rows) {
                if (APPEND) mymask |= 1u << k;
                else live[(page * n_rows + y - 1) * mtx + (xl >> 4)] = 1;
            }
        }
        // APPEND behind other statistics launches of the pass (their marks are in `live`): an M-tile they marked is live too
        if (APPEND && live && mark_lane && y >= 1 && y <= n_rows && live[(page * n_rows + y - 1) * mtx + (xl >> 4)]) mymask |= 1u << k;
        {  // slide down one row
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t bi = (vi >> (8 * m)) & 0xffu, bo = (vo >> (8 * m)) & 0xffu;
                const int d = (int)bi - (int)bo;
                c1[m] += (uint32_t)d;
                c2[m] += (uint32_t)__mul24(d, (int)(bi + bo));
            }
        }
    };
    uint32_t k = 0;
    for (; k + 2 <= k_end; k += 2) {  // two window rows per trip: the second row's loads and lane exchanges overlap the first row's arithmetic
        window_row(k);
        window_row(k + 1);
    }
    if (k < k_end) window_row(k);
    }  // a strip and a band of the page
    if (APPEND) {
        uint32_t mycnt = 0;  // lane k < 16: this wave's live M-tiles in window row k
        for (uint32_t k = 0; k < S8_ROWS; k++) {
            const uint32_t cnt = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64((mymask >> k) & 1u));
            if (lane == k) mycnt = cnt;
        }
        if (lane < S8_ROWS) wg_cnt[wv][lane] = mycnt;
        if (threadIdx.x < 4) wg_wsum[threadIdx.x] = 0;  // (a workgroup may have fewer than four waves)
        __syncthreads();
        // exclusive prefix of the counts in the block's order (band, window row, strip): cell o = (band * 16 + row) * GS + strip, one
        // thread of the first four waves per cell (at most 16 waves x 16 rows = 256 cells)
        const uint32_t o = threadIdx.x, n_cells = GS * GB * S8_ROWS;
        uint32_t cw = 0, ck = 0, v = 0;
        if (o < 256 && o < n_cells) {
            const uint32_t s_ = o % GS, bk = o / GS;
            ck = bk % S8_ROWS;
            cw = (bk / S8_ROWS) * GS + s_;
            v = wg_cnt[cw][ck];
        }
        if (o < 256) {  // wave-uniform: waves 0 .. 3
            uint32_t incl = v;
#pragma unroll
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
                if (lane >= d) incl += t;
            }
            if (lane == 63) wg_wsum[wv] = incl;
            v = incl - v;  // exclusive inside the wave
        }
        __syncthreads();
        if (o < 256) {
            uint32_t before = 0;
            for (uint32_t q = 0; q < wv; q++) before += wg_wsum[q];
            if (o < n_cells) wg_off[cw][ck] = before + v;
            if (o == 0) {
                const uint32_t t = wg_wsum[0] + wg_wsum[1] + wg_wsum[2] + wg_wsum[3];
                wg_base = t ? atomicAdd(list_count, t) : 0u;
            }
        }
        __syncthreads();
        const uint32_t base = wg_base;
        const uint32_t myoff = lane < S8_ROWS ? wg_off[wv][lane] : 0u;
        for (uint32_t k = 0; k < S8_ROWS; k++) {
            const bool mine = (mymask >> k) & 1u;
            const uint64_t m = __builtin_amdgcn_ballot_w64(mine);
            if (m == 0) continue;  // wave-uniform
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const uint32_t row_at = base + (uint32_t)__builtin_amdgcn_readlane((int)myoff, (int)k);
            // the scan kernel's entry: page << 32 | tile row (image row y - 1) << 12 | M-tile column (compact_live_tiles)
            if (mine) list[row_at + rank] = ((uint64_t)page << 32) | ((uint64_t)(y0 + k - 1) << 12) | (xl >> 4);
        }
    }
}

// Live M-tiles -> packed work list (page << 32 | row << 12 | col).  A block compacts 4096 consecutive tiles
// (16 per thread) with one global atomic, so the shared counter sees ~1 atomic per 4096 tiles.  The order of the
// list does not matter for the results (every M-tile is independent; hits are sorted later).
constexpr uint32_t CLT_PER_THREAD = 16;
__global__ __launch_bounds__(256) void compact_live_tiles(const uint8_t *__restrict__ live, uint32_t n_tiles, uint32_t mtx,
                                                          uint32_t n_rows, uint32_t skip_blank, uint64_t *__restrict__ list,
                                                          uint32_t *__restrict__ count) {
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t block_base;
    const uint32_t first = (blockIdx.x * 256 + threadIdx.x) * CLT_PER_THREAD;
    uint32_t bits = 0;
    if (first + CLT_PER_THREAD <= n_tiles) {  // the thread's 16 marks in one (byte-aligned) 16-byte load
        typedef unsigned int clt_v4 __attribute__((ext_vector_type(4), aligned(1)));
        const clt_v4 m = *reinterpret_cast<const clt_v4 *>(live + first);
#pragma unroll
        for (uint32_t k = 0; k < CLT_PER_THREAD; k++)
            if (((m[k / 4] >> (8 * (k % 4))) & 0xffu) || !skip_blank) bits |= 1u << k;
    } else {
#pragma unroll
        for (uint32_t k = 0; k < CLT_PER_THREAD; k++) {
            const uint32_t i = first + k;
            if (i < n_tiles && (live[i] || !skip_blank)) bits |= 1u << k;
        }
    }
    const uint32_t cnt = (uint32_t)__builtin_popcount(bits);
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = cnt;  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if ((int)lane >= o) incl += v;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        block_base = tot ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    uint32_t pos = block_base + incl - cnt;
    for (uint32_t q = 0; q < wv; q++) pos += wave_tot[q];
    for (uint32_t k = 0; k < CLT_PER_THREAD; k++)
        if (bits & (1u << k)) {
            const uint32_t i = first + k;
            const uint32_t col = i % mtx, rowp = i / mtx, row = rowp % n_rows, page = rowp / n_rows;
            list[pos++] = ((uint64_t)page << 32) | ((uint64_t)row << 12) | col;
        }
}

// ---------------------------------------------------------------------------------------------
// 3. exact verify: the reference arithmetic on every candidate (verify_candidate, mfma_common.h)
//
// Legacy tail (fallback of the row path, rows.hip): candidates arrive radix-sorted by the packed key (page, y, x, t):
// neighbouring lanes verify the same or neighbouring windows (cache-friendly), and the survivors stay in process_hits order,
// so no atomics: flag[i] / sim[i] are written in place and order.hip compacts them and derives the per-call lists.
__global__ __launch_bounds__(256) void verify_kernel(const uint64_t *__restrict__ cand, const unsigned long long *__restrict__ n_cand_p, unsigned long long ub,
                                                     const VerifyArgs va, float *__restrict__ sims, uint64_t *__restrict__ flags) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > ub) return;  // grid and buffers are sized for `ub` candidates (+ the sentinel at ub)
    if (i >= min(*n_cand_p, ub)) {  // past the device-side count (and the sentinel: the exclusive scan of flags also yields the total)
        flags[i] = 0;
        return;
    }
    float sim;
    const bool emit = verify_candidate(cand[i], va, &sim);
    sims[i] = sim;
    flags[i] = emit ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// host side

// Size classes -> K layouts, kept widths, super-classes, bank offsets (host only).
void layout_supers(focr_ctx *c) {
    // Column drop (mfma_common.h, "threshold planes"): a class of width 4k + 1 (9, 13) gives its last column to the bound
    // and takes the next narrower K layout — BASELINE configs[1]'s 9x15 templates: 2 K-steps instead of 3.
    for (SizeClass &sc : c->classes) sc.keep_w = (c->column_drop && !sc.tall && (sc.n_w == 9 || sc.n_w == 13)) ? sc.n_w - 1 : sc.n_w;
    // K layout per class (mfma_common.h).  Narrow classes ride the 12-byte-row layout whenever a 9..12-wide
    // class exists, so that all of them share one set of A fragments (one "super-class", one kernel pass).
    bool any_mid = false;
    for (const SizeClass &sc : c->classes) any_mid |= (!sc.tall && sc.keep_w >= 9 && sc.keep_w <= 12);
    c->supers.clear();
    for (size_t k = 0; k < c->classes.size(); k++) {
        SizeClass &sc = c->classes[k];
        if (sc.tall) {  // scanned exactly by scan_tall_kernel; no quantised copy
            sc.layout = LAYOUT_W16;
            sc.k_groups = sc.n_tiles16 = 0;
            continue;
        }
        sc.layout = sc.keep_w >= 13 ? LAYOUT_W16 : (any_mid ? LAYOUT_W12 : LAYOUT_W8);
        if (sc.layout == LAYOUT_W8) sc.k_groups = ((sc.n_h + 1) / 2 + 3) / 4 * 4;   // 2 rows per group
        else if (sc.layout == LAYOUT_W12) sc.k_groups = (sc.n_h + 15) / 16 * 12;    // 16 rows -> 12 groups (3 K-steps)
        else sc.k_groups = (sc.n_h + 3) / 4 * 4;                                    // 1 row per group
        sc.n_tiles16 = (sc.n_templates + 15) / 16;
        size_t si = 0;
        for (; si < c->supers.size(); si++)
            if (c->supers[si].layout == sc.layout && c->supers[si].ksteps == sc.k_groups / 4) break;
        if (si == c->supers.size()) {
            SuperClass su{};
            su.layout = sc.layout;
            su.ksteps = sc.k_groups / 4;
            c->supers.push_back(su);
        }
        SuperClass &su = c->supers[si];
        su.classes.push_back((uint32_t)k);
        su.tile_first.push_back(su.n_tiles);
        su.n_tiles += sc.n_tiles16;
    }
    size_t q_bytes = 0, tg_entries = 0;
    for (SuperClass &su : c->supers) {
        su.q_offset = q_bytes;
        su.tg_offset = tg_entries;
        q_bytes += (size_t)su.n_tiles * su.ksteps * 1024;
        tg_entries += (size_t)su.n_tiles * 16;
        for (size_t i = 0; i < su.classes.size(); i++) {
            SizeClass &sc = c->classes[su.classes[i]];
            sc.q_offset = (uint32_t)(su.q_offset + (size_t)su.tile_first[i] * su.ksteps * 1024);
            sc.tg_offset = (uint32_t)(su.tg_offset + (size_t)su.tile_first[i] * 16);
        }
    }
}

// Which slot of its class's N-tiles each template takes: the caller's order, except that templates that never emit (constant
// needles: the space glyph) go last, next to the padding — only a class's last tiles hold dead slots then, and the scan kernel
// looks at slot ids in those tiles alone (scan_mfma2.hip, the candidate path).  The candidate KEYS carry the caller's template
// index (tglobal), so nothing outside the scan kernel sees the order.
// (Measured and dropped: grouping look-alike templates into the same tile, greedy by correlation — the four sub-pixel shifts of
// a glyph then share a tile, yet BASELINE configs[1] visits 1.31 M N-tiles per batch either way: DESIGN.md, dead ends.)
static std::vector<uint32_t> live_first_slots(const std::vector<std::vector<double>> &bp) {
    std::vector<uint32_t> slot(bp.size(), 0);
    uint32_t next = 0;
    for (size_t i = 0; i < bp.size(); i++)
        if (!bp[i].empty()) slot[i] = next++;
    for (size_t i = 0; i < bp.size(); i++)
        if (bp[i].empty()) slot[i] = next++;
    return slot;
}

// Quantise the bank (header comment; column drop: mfma_common.h).  `dense` holds the class-ordered dense needles.  Host only:
// fills the per-lane MFMA operand image of every class, the class-ordered template ids (~0 = dead / padding) and
// c->mfma_c_scale / mfma_e_max / mfma_rho_max.
int quantise_bank(focr_ctx *c, const uint8_t *dense, std::vector<int8_t> &qbank, std::vector<uint32_t> &tglobal, std::vector<uint32_t> &order_of) {
    layout_supers(c);
    size_t q_bytes = 0, tg_entries = 0;
    for (const SuperClass &su : c->supers) {
        q_bytes += (size_t)su.n_tiles * su.ksteps * 1024;
        tg_entries += (size_t)su.n_tiles * 16;
    }
    qbank.assign(q_bytes, 0);
    tglobal.assign(tg_entries, 0xffffffffu);
    order_of.assign(c->n_templates, 0);
    c->mfma_slot.assign(c->h_tconst.size(), 0);
    c->mfma_c_scale.clear();
    c->mfma_e_max.clear();
    c->mfma_rho_max.clear();
    for (size_t k = 0; k < c->classes.size(); k++) {
        SizeClass &sc = c->classes[k];
        const uint32_t n = sc.n_w * sc.n_h, ksteps = sc.k_groups / 4, kw = sc.keep_w, n_k = kw * sc.n_h;
        sc.n_live = 0;
        if (sc.tall) {
            for (uint32_t i = 0; i < sc.n_templates; i++) order_of[c->h_tconst[sc.first + i].index] = sc.first + i;
            c->mfma_c_scale.push_back(1.0);
            c->mfma_e_max.push_back(0.0);
            c->mfma_rho_max.push_back(0.0);
            continue;
        }
        // unit mean-centred templates beta; on the kept columns beta' = beta + sigma / n_k (sigma = the dropped column's sum)
        std::vector<std::vector<double>> bp(sc.n_templates);
        double max_ratio = 0.0, rho_max = 0.0;
        for (uint32_t i = 0; i < sc.n_templates; i++) {
            const TemplateConst &tc = c->h_tconst[sc.first + i];
            order_of[tc.index] = sc.first + i;
            const uint8_t *nd = dense + c->h_needle_off[sc.first + i];
            double s = 0, s2 = 0;
            for (uint32_t p = 0; p < n; p++) {
                s += nd[p];
                s2 += (double)nd[p] * nd[p];
            }
            const double mean = s / n, n2 = s2 - s * s / n;
            if (!(n2 > 0.0) || !std::isfinite(tc.rnorm_n)) continue;  // constant needle: rnorm_n = inf, never emits
            const double norm_n = std::sqrt(n2);
            double sigma = 0, rho2 = 0;
            for (uint32_t j = 0; j < sc.n_h; j++)
                for (uint32_t x = kw; x < sc.n_w; x++) {
                    const double b = (nd[j * sc.n_w + x] - mean) / norm_n;
                    sigma += b;
                    rho2 += b * b;
                }
            rho_max = std::max(rho_max, std::sqrt(rho2));
            bp[i].resize(n_k);
            for (uint32_t j = 0; j < sc.n_h; j++)
                for (uint32_t x = 0; x < kw; x++) {
                    const double b = (nd[j * sc.n_w + x] - mean) / norm_n + sigma / n_k;
                    bp[i][j * kw + x] = b;
                    max_ratio = std::max(max_ratio, std::fabs(b));
                }
        }
        const std::vector<uint32_t> slot = live_first_slots(bp);
        for (uint32_t i = 0; i < sc.n_templates; i++) {
            c->mfma_slot[sc.first + i] = slot[i];
            if (bp[i].empty()) continue;
            tglobal[sc.tg_offset + slot[i]] = c->h_tconst[sc.first + i].index;
            sc.n_live++;
        }
        const double c_scale = max_ratio > 0 ? 126.0 / max_ratio : 1.0;
        double e_max = 0.0;
        std::vector<double> rk(n_k);
        std::vector<int> bq(n_k);
        std::vector<uint32_t> idx(n_k);
        for (uint32_t i = 0; i < sc.n_templates; i++) {
            if (bp[i].empty()) continue;
            long sum = 0;
            for (uint32_t p = 0; p < n_k; p++) {
                rk[p] = c_scale * bp[i][p];
                bq[p] = (int)std::floor(rk[p]);
                sum += bq[p];
                idx[p] = p;
            }
            // largest-remainder rounding so that the int8 template sums to exactly zero
            const long deficit = -sum;  // sum(rk) = 0 in exact arithmetic, so 0 <= deficit <= n_k
            if (deficit < 0 || deficit > (long)n_k) return fail(c, FOCR_ERR_INVALID, "mfma bank: rounding deficit out of range");
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return rk[a] - bq[a] > rk[b] - bq[b]; });
            for (long d = 0; d < deficit; d++) bq[idx[d]] += 1;
            double e2 = 0;
            long check = 0;
            for (uint32_t p = 0; p < n_k; p++) {
                if (bq[p] > 127 || bq[p] < -127) return fail(c, FOCR_ERR_INVALID, "mfma bank: quantised template out of int8 range");
                double e = rk[p] - bq[p];
                e2 += e * e;
                check += bq[p];
            }
            if (check != 0) return fail(c, FOCR_ERR_INVALID, "mfma bank: quantised template does not sum to zero");
            e_max = std::max(e_max, std::sqrt(e2));
            // scatter into the per-lane MFMA B layout: [n-tile][k-step][g][n][16 bytes]
            const uint32_t nt = slot[i] / 16, nn = slot[i] % 16;
            for (uint32_t j = 0; j < sc.n_h; j++)
                for (uint32_t x = 0; x < kw; x++) {
                    uint32_t ks, g, byte;
                    kgroup_of(sc.layout, j, x, &ks, &g, &byte);
                    qbank[sc.q_offset + ((size_t)(nt * ksteps + ks) * 64 + g * 16 + nn) * 16 + byte] = (int8_t)bq[j * kw + x];
                }
        }
        c->mfma_c_scale.push_back(c_scale);
        c->mfma_e_max.push_back(e_max);
        c->mfma_rho_max.push_back(rho_max);
    }
    return FOCR_OK;
}

int build_mfma_bank(focr_ctx *c, const uint8_t *dense) {
    std::vector<int8_t> qbank;
    std::vector<uint32_t> tglobal, order_of;
    if (int rc = quantise_bank(c, dense, qbank, tglobal, order_of)) return rc;
    FOCR_HIP(c, hipMalloc((void **)&c->d_qbank, qbank.size() ? qbank.size() : 16));
    FOCR_HIP(c, hipMemcpy(c->d_qbank, qbank.data(), qbank.size(), hipMemcpyHostToDevice));
    FOCR_HIP(c, hipMalloc((void **)&c->d_tglobal, tglobal.size() ? tglobal.size() * 4 : 16));
    FOCR_HIP(c, hipMemcpy(c->d_tglobal, tglobal.data(), tglobal.size() * 4, hipMemcpyHostToDevice));
    // verify operand: every template as n_h rows of 16 bytes (zero padded), class-ordered
    std::vector<uint8_t> n16;
    std::vector<uint32_t> n16_row(c->h_tconst.size(), 0);
    for (size_t ci = 0; ci < c->h_tconst.size(); ci++) {
        const TemplateConst &tc = c->h_tconst[ci];
        n16_row[ci] = (uint32_t)(n16.size() / 16);
        const uint8_t *nd = dense + c->h_needle_off[ci];
        const uint32_t row_bytes = tc.n_w > 16 ? 32 : 16;
        for (uint32_t j = 0; j < tc.n_h; j++)
            for (uint32_t x = 0; x < row_bytes; x++) n16.push_back(x < tc.n_w ? nd[j * tc.n_w + x] : 0);
    }
    FOCR_HIP(c, hipMalloc((void **)&c->d_needles16, n16.size() ? n16.size() : 16));
    FOCR_HIP(c, hipMemcpy(c->d_needles16, n16.data(), n16.size(), hipMemcpyHostToDevice));
    {  // the verify's per-template record, by global template index
        std::vector<VerifyMeta> vm(c->n_templates);
        for (size_t ci = 0; ci < c->h_tconst.size(); ci++) {
            const TemplateConst &tc = c->h_tconst[ci];
            vm[tc.index] = VerifyMeta{tc.s_n, tc.n_recip, tc.rnorm_n, (uint16_t)tc.n_w, (uint16_t)tc.n_h, n16_row[ci]};
        }
        FOCR_HIP(c, hipMalloc(&c->d_vmeta, vm.size() * sizeof(VerifyMeta)));
        FOCR_HIP(c, hipMemcpy(c->d_vmeta, vm.data(), vm.size() * sizeof(VerifyMeta), hipMemcpyHostToDevice));
    }
    {  // the same operand by GLOBAL template index, as rows of 12 bytes (every template at most 12 px wide) or 16: chunks of consecutive
       // templates are contiguous there (verify_chunks_kernel, rows.hip: banks whose operand does not fit the LDS whole)
        uint32_t max_w = 0;
        for (const TemplateConst &tc : c->h_tconst) max_w = std::max<uint32_t>(max_w, tc.n_w);
        c->vrow_bytes = max_w <= 12 ? 12u : max_w <= 16 ? 16u : 0u;
        c->h_vrow0_t.assign(c->n_templates + 1, 0);
        if (c->vrow_bytes) {
            std::vector<size_t> ci_of(c->n_templates, 0);
            for (size_t ci = 0; ci < c->h_tconst.size(); ci++) ci_of[c->h_tconst[ci].index] = ci;
            for (size_t t = 0; t < c->n_templates; t++) c->h_vrow0_t[t + 1] = c->h_vrow0_t[t] + c->h_tconst[ci_of[t]].n_h;
            std::vector<uint8_t> rows((size_t)c->h_vrow0_t[c->n_templates] * c->vrow_bytes + 16, 0);
            std::vector<VerifyMeta> vm(c->n_templates);
            for (size_t t = 0; t < c->n_templates; t++) {
                const TemplateConst &tc = c->h_tconst[ci_of[t]];
                const uint8_t *nd = dense + c->h_needle_off[ci_of[t]];
                for (uint32_t j = 0; j < tc.n_h; j++) memcpy(&rows[((size_t)c->h_vrow0_t[t] + j) * c->vrow_bytes], nd + (size_t)j * tc.n_w, tc.n_w);
                vm[t] = VerifyMeta{tc.s_n, tc.n_recip, tc.rnorm_n, (uint16_t)tc.n_w, (uint16_t)tc.n_h, c->h_vrow0_t[t]};
            }
            FOCR_HIP(c, hipMalloc((void **)&c->d_vrows_t, rows.size()));
            FOCR_HIP(c, hipMemcpy(c->d_vrows_t, rows.data(), rows.size(), hipMemcpyHostToDevice));
            FOCR_HIP(c, hipMalloc(&c->d_vmeta_t, vm.size() * sizeof(VerifyMeta)));
            FOCR_HIP(c, hipMemcpy(c->d_vmeta_t, vm.data(), vm.size() * sizeof(VerifyMeta), hipMemcpyHostToDevice));
        }
    }
    FOCR_HIP(c, hipMalloc((void **)&c->d_needle16_row, n16_row.size() * 4));
    FOCR_HIP(c, hipMemcpy(c->d_needle16_row, n16_row.data(), n16_row.size() * 4, hipMemcpyHostToDevice));
    FOCR_HIP(c, hipMalloc((void **)&c->d_order_of, order_of.size() * 4));
    FOCR_HIP(c, hipMemcpy(c->d_order_of, order_of.data(), order_of.size() * 4, hipMemcpyHostToDevice));
    return FOCR_OK;
}

// Threshold parameters of one size class for one scan (mfma_common.h, "threshold planes"): kq towards -inf, crk upwards.
PlaneParams plane_params(const focr_ctx *c, size_t k, double thr_d) {
    const SizeClass &sc = c->classes[k];
    const double cs = c->mfma_c_scale[k], em = c->mfma_e_max[k], rho = c->mfma_rho_max[k];
    const double n = (double)sc.n_w * sc.n_h, n_k = (double)sc.keep_w * sc.n_h, D = n - n_k;
    // kappa carries a relative 1e-4 for the f64 roundings of the reference's formula (its similarity differs from the real
    // number by far less)
    const double kappa = cs * thr_d - em - 1e-4 * (cs * (1.0 + std::fabs(thr_d)) + em);
    PlaneParams p{};
    const double kq_d = kappa / std::sqrt(n);
    float kq = (float)kq_d;
    if ((double)kq > kq_d) kq = std::nextafterf(kq, -INFINITY);
    kq = std::nextafterf(kq, -INFINITY);
    p.kq = std::isfinite(kq) ? kq : -3.0e38f;  // threshold -inf: everything is a candidate
    p.crk = 0.f;
    if (D > 0 && rho > 0) {
        const double cr_d = cs * rho * (1.0 + 1e-4) / n_k;
        float cr = (float)cr_d;
        if ((double)cr < cr_d) cr = std::nextafterf(cr, INFINITY);
        p.crk = std::nextafterf(std::nextafterf(cr, INFINITY), INFINITY);  // also covers a 1-ulp-low square root of W
    }
    // |L| <= |kq| * sqrt(V) + crk * sqrt(W) with sqrt(V) <= 127.5 n, sqrt(W) <= 255 n_k sqrt(D).  The plane's unit S, a power of two
    // (mfma_common.h): every |L - 2| / S within 16384 while |L| < 2^28 (beyond that the plane value's clamp takes over: such
    // thresholds are unreachable or pass everything either way), and S >= K / 2 for the K bytes the MFMA multiplies per window, so
    // that the "never" value -32768 * S lies below every -|G| (|G| <= K * 127 * 128).  S <= 2^14: |C-in| <= 2^29, G + C-in cannot wrap.
    const double l_max = std::min(std::ldexp(1.0, 28), std::fabs((double)p.kq) * 127.5 * n + (double)p.crk * 255.0 * n_k * std::sqrt(std::max(D, 0.0)) + 4.0);
    const double s_min = std::max(l_max / 16384.0, 16.0 * (double)std::max<uint32_t>(sc.k_groups, 4) / 2.0);  // K = 16 bytes per k-group
    uint32_t e = 5;
    while (std::ldexp(1.0, (int)e) < s_min && e < 14) e++;
    p.shift = e;
    p.S = std::ldexp(1.0f, (int)e);
    p.inv_S = std::ldexp(1.0f, -(int)e);
    return p;
}

// whether a size class's statistics take the register form (stats8_kernel): threshold planes, a kept width of 8 px
static bool stats_register_form(const focr_ctx *c, const SizeClass &sc) {
    static const bool no_s8 = getenv("FOCR_NO_STATS8") != nullptr;  // A/B: the LDS-tiled kernel for every class
    // kept widths 4, 8, 12, 16 (a dropped column only exists for 9 -> 8 and 13 -> 12: layout_supers)
    return sc.keep_w % 4 == 0 && sc.keep_w >= 4 && sc.keep_w <= 16 && !no_s8 && c->dbg_stats_form == 0;
}

// one statistics launch: class k (full box), optionally together with its kept box as class `pair` (< 0: none);
// append_list / append_count: the launch is the pass's only one and appends its live M-tiles to the work list itself (stats8_kernel, APPEND)
template <int OUT>
static int launch_stats(focr_ctx *c, size_t k, int pair, double thr_d, void *out, void *out_pair, uint32_t Lpitch, uint32_t Lrows, uint8_t *live,
                        uint32_t mtx, uint32_t n_rows, uint64_t *append_list = nullptr, uint32_t *append_count = nullptr) {
    const SizeClass &sc = c->classes[k];
    // only what the scan kernels read: the windows of the pass's M-tiles (x < 16 * mtx) in the searched rows (y <= n_rows)
    dim3 grid(std::min<unsigned>(Lpitch / STX, (16 * mtx + STX - 1) / STX), std::min<unsigned>((Lrows + STY - 1) / STY, (n_rows + 1 + STY - 1) / STY),
              (unsigned)c->sub_np);
    StatsOut A{plane_params(c, k, thr_d), out}, B{};
    if (pair >= 0) B = StatsOut{plane_params(c, (size_t)pair, thr_d), out_pair};
    const bool drop = sc.keep_w != sc.n_w, small = sc.n_w * sc.n_h <= 256;
    if (append_list && !(OUT == 1 && stats_register_form(c, sc))) return fail(c, FOCR_ERR_INVALID, "scan_mfma: internal: direct append without the register form");
    if (OUT == 1 && stats_register_form(c, sc)) {  // kept width 8: the register form (stats8_kernel)
        const uint32_t cols = std::min<uint32_t>(Lpitch, 16 * mtx), rows_n = std::min<uint32_t>(Lrows, n_rows + 1);
        const uint32_t strips_x = (cols + S8_COLS - 1) / S8_COLS, bands_y = (rows_n + S8_ROWS - 1) / S8_ROWS;
        // a workgroup: GS neighbouring strips x GB bands, at most 16 waves (stats8_kernel)
        static const uint32_t gb_max = getenv("FOCR_S8_GB") ? (uint32_t)atoi(getenv("FOCR_S8_GB")) : 2u;  // 1 .. 5 measured: 34.15 / 34.60 / 33.89 / 34.23 / 33.70 Gpx/s (tools/r5_gb.sh)
        const uint32_t GS = std::min<uint32_t>(strips_x, 16), GB = std::max<uint32_t>(1, std::min<uint32_t>(std::min<uint32_t>(16 / GS, gb_max), bands_y));
        const uint32_t sgroups = (strips_x + GS - 1) / GS, bgroups = (bands_y + GB - 1) / GB;
        const uint64_t n_wgs = (uint64_t)sgroups * bgroups * c->sub_np;
        if (n_wgs >= 0x7fffffffull) return fail(c, FOCR_ERR_INVALID, "scan_mfma: batch too large for 32-bit tile ids; scan fewer pages per call");
        auto launch8 = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)n_wgs), dim3(GS * GB * 64), 0, c->stream, c->d_pages + c->sub_p0 * c->rows_alloc * c->pitch, (uint32_t)c->pitch,
                               (uint32_t)c->rows_alloc, (uint32_t)c->r_w, (uint32_t)c->r_h, sc.n_w, sc.n_h, A, B, Lpitch, Lrows, live, mtx, n_rows, strips_x, bands_y,
                               GS, GB, sgroups, bgroups, append_list, append_count);
        };
#define S8_DROPS(KQ, AP)                                                                                                                         \
    if (pair >= 0) small ? launch8(stats8_kernel<KQ, true, true, true, AP>) : launch8(stats8_kernel<KQ, false, true, true, AP>);                  \
    else if (drop) small ? launch8(stats8_kernel<KQ, true, true, false, AP>) : launch8(stats8_kernel<KQ, false, true, false, AP>);                \
    else small ? launch8(stats8_kernel<KQ, true, false, false, AP>) : launch8(stats8_kernel<KQ, false, false, false, AP>);
#define S8_PLAIN(KQ, AP) small ? launch8(stats8_kernel<KQ, true, false, false, AP>) : launch8(stats8_kernel<KQ, false, false, false, AP>);
#define S8_FORMS(AP)                                                                                                                              \
    switch (sc.keep_w) {                                                                                                                          \
        case 4: S8_PLAIN(1, AP) break;                                                                                                            \
        case 8: S8_DROPS(2, AP) break;                                                                                                            \
        case 12: S8_DROPS(3, AP) break;                                                                                                           \
        default: S8_PLAIN(4, AP) break;                                                                                                           \
    }
        if (append_list) {
            S8_FORMS(true)
        } else {
            S8_FORMS(false)
        }
#undef S8_FORMS
#undef S8_PLAIN
#undef S8_DROPS
        FOCR_HIP(c, hipGetLastError());
        return FOCR_OK;
    }
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid, dim3(256), stats_lds_bytes(sc.n_h), c->stream, c->d_pages + c->sub_p0 * c->rows_alloc * c->pitch, (uint32_t)c->pitch,
                           (uint32_t)c->rows_alloc, (uint32_t)c->r_w, (uint32_t)c->r_h, sc.n_w, sc.n_h, A, B, Lpitch, Lrows, live, mtx, n_rows);
    };
#define STATS_CASE(NDW)                                                                          \
    case NDW:                                                                                    \
        if (pair >= 0) small ? launch(stats_kernel<NDW, true, OUT, true, true>) : launch(stats_kernel<NDW, false, OUT, true, true>);       \
        else if (drop) small ? launch(stats_kernel<NDW, true, OUT, true, false>) : launch(stats_kernel<NDW, false, OUT, true, false>);     \
        else small ? launch(stats_kernel<NDW, true, OUT, false, false>) : launch(stats_kernel<NDW, false, OUT, false, false>);             \
        break;
    switch ((sc.keep_w + 3) / 4) {  // dwords of the kept width (keep_w = n_w unless the class's last column is dropped)
        STATS_CASE(1) STATS_CASE(2) STATS_CASE(3) STATS_CASE(4)
        default: return fail(c, FOCR_ERR_INVALID, "scan_mfma: unsupported box width");
    }
#undef STATS_CASE
    FOCR_HIP(c, hipGetLastError());
    return FOCR_OK;
}

VerifyArgs verify_args(const focr_ctx *c, double thr_d) {
    return VerifyArgs{c->d_pages, (uint32_t)c->pitch, (uint32_t)c->rows_alloc, c->fmt, c->d_order_of, c->d_tconst,
                      reinterpret_cast<const v4i *>(c->d_needles16), c->d_needle16_row, thr_d, reinterpret_cast<const VerifyMeta *>(c->d_vmeta),
                      (uint32_t)c->n_templates, (uint32_t)c->n_pages,
                      (uint32_t)c->r_w, (uint32_t)c->r_h, (unsigned long long *)(c->d_res + 4)};
}

// Process-wide hand-over of the scan kernel between contexts of one device (events are never destroyed).
struct ScanTurns {
    std::mutex mu;
    hipEvent_t ev[8];
    unsigned n = 0;
    bool init = false;
};
static ScanTurns scan_turns[64];
static ScanTurns stats_turns[64];  // the same for the statistics in front of the scans

int launch_scan_mfma(focr_ctx *c, float threshold) {
    const double thr_d = (double)threshold;  // src/ncc.cpp:83, 288
    const uint32_t Lpitch = (uint32_t)((c->r_w + 63) / 64 * 64 + 64), Lrows = (uint32_t)((c->r_h + 7) / 8 * 8 + 8);
    const size_t L_per_class = c->n_pages * (size_t)Lrows * Lpitch;
    // the per-class int32 threshold tables are only needed by super-classes on the legacy path
    bool need_L = false;
    for (const SuperClass &su : c->supers) need_L |= su.ksteps > 4 || su.classes.size() > (size_t)MAX_PLANE_VALUES || c->prefilter == FOCR_PREFILTER_LEGACY;
    const size_t L_bytes = need_L ? L_per_class * c->classes.size() * sizeof(int32_t) : 0;
    if (c->L_bytes < L_bytes) {
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_L) (void)hipFree(c->d_L);
        c->d_L = nullptr;
        c->L_bytes = 0;
        if (hipMalloc(&c->d_L, L_bytes) != hipSuccess) return fail(c, FOCR_ERR_NOMEM, "scan_mfma: hipMalloc(negL) failed");
        c->L_bytes = L_bytes;
    }
    int rc = ensure_hit_capacity(c, std::max<size_t>(c->hit_capacity, std::max<size_t>(1u << 20, c->sub_np * 65536)));
    if (rc) return rc;
    size_t want_cand = std::max<size_t>(c->cand_capacity, std::max<size_t>(1u << 21, c->sub_np * 131072));
    if (c->estimated) want_cand = std::max(want_cand, c->est_cand);

    for (int attempt = 0; attempt < 4; attempt++) {
        if (c->cand_capacity < want_cand) {
            FOCR_HIP(c, hipStreamSynchronize(c->stream));
            if (c->d_cand) (void)hipFree(c->d_cand);
            c->d_cand = nullptr;
            c->cand_capacity = 0;
            if (hipMalloc(&c->d_cand, want_cand * 8) != hipSuccess) return fail(c, FOCR_ERR_NOMEM, "scan_mfma: hipMalloc(cand) failed");
            c->cand_capacity = want_cand;
        }
        c->counters[3] = 0;
        c->launches_reset();
        ClearList clear{};  // everything the scan needs zeroed: one launch (launch_clear), in front of the first kernel
        if (!clear.add(c->d_counter, COUNTER_BYTES)) return fail(c, FOCR_ERR_INVALID, "scan_mfma: clear list full or region too large");  // counters + the scan kernels' item queues
        c->scan_queues_used = 0;
        if (!clear.add(c->d_res, 7 * sizeof(uint64_t))) return fail(c, FOCR_ERR_INVALID, "scan_mfma: clear list full or region too large");
        // Sizes.  Exact mode: the host reads the candidate count after the scan kernels and the hit count after the
        // verify (two waits), so every later phase runs on exact sizes.  Estimated mode (ctx.hip: same setup as the
        // previous scan): the counts stay on the device, grids and buffers take the previous counts + a margin (4 .. 20 %, ctx.hip) as bounds,
        // unused candidate slots hold the largest key so that the sort leaves them at the end; nothing waits.
        c->ub_cand = c->estimated ? std::min(c->est_cand, c->cand_capacity) : c->cand_capacity;
        // Tail: the row path (rows.hip) unless a row could exceed its capacity — exact mode finds out after the scan kernels,
        // estimated mode goes by the previous scan's largest row + 25 % (a larger one sets the overflow bit: batch redone).
        bool use_rows = rows_applicable(c);
        uint32_t row_cap = 0;
        if (use_rows && c->estimated) {
            row_cap = rows_capacity_for((uint64_t)c->est_row_max + c->est_row_max / 4 + 16);
            use_rows = c->est_row_max != 0 && row_cap != 0;
        }
        c->row_hist = RowHist{};
        if (use_rows && (rc = rows2_begin(c, clear))) return rc;  // hits-first row tail: verify in flush order, only hits are bucketed and sorted (rows.hip)
        // legacy tail, estimated sizes: unused candidate slots hold the largest key so that the radix sort leaves them at the end
        if (c->estimated && !use_rows) FOCR_HIP(c, hipMemsetAsync(c->d_cand, 0xff, c->ub_cand * 8, c->stream));
        FOCR_HIP(c, hipEventRecord(c->ev[0], c->stream));
        // `sim > +inf` is never true (NaN thresholds arrive here as +inf, focr_scan): no statistics, no scan, zero candidates
        // (kappa would be inf - inf = NaN and every window of every live tile a candidate for verify to reject)
        const bool nothing = !(thr_d < (double)INFINITY);
        if (nothing) {
            if ((rc = launch_clear(c, clear))) return rc;
            FOCR_HIP(c, hipEventRecord(c->ev[1], c->stream));
        }
        if (!nothing) {
        // 1. statistics + live-tile work lists, per super-class (classes that share one scan pass)
        size_t tiles_total = 0;
        for (SuperClass &su : c->supers) {
            su.min_w = su.min_h = 0xffffffffu;
            for (uint32_t k : su.classes) {
                const SizeClass &sc = c->classes[k];
                if (sc.n_w >= c->r_w || sc.n_h >= c->r_h) continue;
                su.min_w = std::min(su.min_w, sc.n_w);
                su.min_h = std::min(su.min_h, sc.n_h);
            }
            su.mtx = su.n_rows = 0;
            su.live_offset = tiles_total;
            if (su.min_w == 0xffffffffu) continue;  // nothing searchable
            su.mtx = (uint32_t)((c->r_w - su.min_w + 1 + 15) / 16);  // windows x in [0, r_w - min n_w]
            su.n_rows = (uint32_t)(c->r_h - su.min_h);               // y in [1, r_h - min n_h]
            const uint64_t nt = (uint64_t)su.mtx * su.n_rows * c->sub_np;
            if (nt >= 0x7fffffffull) return fail(c, FOCR_ERR_INVALID, "scan_mfma: batch too large for 32-bit tile ids; scan fewer pages per call");
            tiles_total += (size_t)nt;
        }
        uint8_t *live = (uint8_t *)c->scan_live.ensure(c, tiles_total + 24);
        uint64_t *live_list = (uint64_t *)c->scan_live_list.ensure(c, (tiles_total + 16) * 8);
        if (!live || !live_list) return fail(c, FOCR_ERR_NOMEM, "scan_mfma: hipMalloc failed");
        if (!clear.add(live, tiles_total + 16)) return fail(c, FOCR_ERR_INVALID, "scan_mfma: clear list full or region too large");
        // The statistics of the batches of one device take turns too (an event chain like the scan kernels' below): two lanes that
        // start their statistics at the same moment — a pipeline filling up from a drained state does that — share the free CUs,
        // finish together, then wait for their scan turns one behind the other, and their tails overlap again: a second steady state
        // with the same work and 7 % less throughput (two batches completing together, then 2.5 and 3.0 ms: DESIGN.md section 5,
        // "two rhythms").  In the staggered state a batch's statistics never meet another's, and the chain costs nothing.
        static const bool stats_chain = getenv("FOCR_NO_STATS_CHAIN") == nullptr;
        ScanTurns &st = stats_turns[(unsigned)c->device % 64];
        if (stats_chain) {
            std::lock_guard<std::mutex> turn(st.mu);
            if (!st.init) {
                for (hipEvent_t &e : st.ev) FOCR_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                st.init = true;
            }
            if (st.n) FOCR_HIP(c, hipStreamWaitEvent(c->stream, st.ev[(st.n - 1) % 8], 0));
            // (the event of THIS batch's statistics is recorded below, under the same lock order: reserve its place now)
            c->stats_turn = st.n++;
        }
        if ((rc = launch_clear(c, clear))) return rc;
        // which super-classes take the plane path (scan_mfma2s_kernel), and their threshold planes
        const size_t plane = c->sub_np * (size_t)Lrows * Lpitch;  // int16 values per plane
        std::vector<int> two(c->supers.size(), 0);
        std::vector<size_t> plane_off(c->supers.size(), 0);
        size_t plane_vals = 0;
        // two[si]: 0 = legacy path (per-class int32 negL tables, scan_mfma2_kernel: > 4 K-steps or > 4 size classes),
        //          1 = int16 threshold planes + scan_mfma2s_kernel (A = templates, B = windows)
        for (size_t si = 0; si < c->supers.size(); si++) {
            const SuperClass &su = c->supers[si];
            if (!su.mtx || su.ksteps > 4 || su.classes.size() > (size_t)MAX_PLANE_VALUES || c->prefilter == FOCR_PREFILTER_LEGACY) continue;
            const uint32_t nv = (uint32_t)su.classes.size();
            if ((size_t)(nv <= 1 ? 1 : nv <= 2 ? 2 : 4) * plane * 2 >= ((size_t)1 << 32)) continue;  // the plane kernel addresses a pass's planes with 32-bit offsets
            two[si] = 1;
            plane_off[si] = plane_vals;
            plane_vals += (size_t)(nv <= 1 ? 1 : nv <= 2 ? 2 : 4) * plane;  // the kernel is instantiated for 1 / 2 / 4 values
        }
        if (c->planes_bytes < plane_vals * 2) {
            FOCR_HIP(c, hipStreamSynchronize(c->stream));
            if (c->d_planes) (void)hipFree(c->d_planes);
            c->d_planes = nullptr;
            c->planes_bytes = 0;
            if (hipMalloc((void **)&c->d_planes, plane_vals * 2) != hipSuccess) return fail(c, FOCR_ERR_NOMEM, "scan_mfma: hipMalloc(planes) failed");
            c->planes_bytes = plane_vals * 2;
        }
        for (size_t si = 0; si < c->supers.size(); si++) {
            const SuperClass &su = c->supers[si];
            if (!su.mtx) continue;
            uint8_t *lv = live + su.live_offset;
            std::vector<char> done(su.classes.size(), 0);
            std::vector<size_t> order;  // classes whose last column is dropped first: they can take their kept box along
            for (int pass = 0; pass < 2; pass++)
                for (size_t v = 0; v < su.classes.size(); v++)
                    if ((c->classes[su.classes[v]].keep_w != c->classes[su.classes[v]].n_w) == (pass == 0)) order.push_back(v);
            struct StatsLaunch {
                size_t v, k, pv;
                int pair;
            };
            std::vector<StatsLaunch> todo;  // the pass's statistics launches
            for (size_t v : order) {
                if (done[v]) continue;
                const size_t k = su.classes[v];
                const SizeClass &sc = c->classes[k];
                if (!two[si] && (sc.n_w >= c->r_w || sc.n_h >= c->r_h)) continue;  // nothing searchable: its tiles are skipped below
                // a class whose last column is dropped computes its kept box's statistics anyway: if that box is a size class
                // of this pass too, both come out of one launch
                int pair = -1;
                size_t pv = 0;
                if (sc.keep_w != sc.n_w)
                    for (size_t u = 0; u < su.classes.size(); u++) {
                        const SizeClass &o = c->classes[su.classes[u]];
                        if (u != v && !done[u] && o.n_w == sc.keep_w && o.n_h == sc.n_h && o.keep_w == o.n_w) pair = (int)su.classes[u], pv = u;
                    }
                todo.push_back(StatsLaunch{v, k, pv, pair});
                done[v] = 1;
                if (pair >= 0) done[pv] = 1;
            }
            // ONE launch for the whole pass, in the register form: its marks are final and it appends the live M-tiles to the work
            // list itself (stats8_kernel, APPEND) — no mark bytes, no compaction launch in front of the scan kernel
            static const bool no_append = getenv("FOCR_NO_STATS_APPEND") != nullptr;  // A/B
            // (several launches: one in the register form goes LAST and merges the marks the others left in `live`)
            for (size_t i = 0; i + 1 < todo.size(); i++)
                if (stats_register_form(c, c->classes[todo[i].k]) && !stats_register_form(c, c->classes[todo.back().k])) std::swap(todo[i], todo.back());
            const bool direct = two[si] && !todo.empty() && stats_register_form(c, c->classes[todo.back().k]) && !no_append;
            for (const StatsLaunch &L : todo) {
                if (two[si]) {
                    uint16_t *base = c->d_planes + plane_off[si];
                    const bool last = direct && &L == &todo.back();
                    rc = launch_stats<1>(c, L.k, L.pair, thr_d, base + L.v * plane, L.pair >= 0 ? base + L.pv * plane : nullptr, Lpitch, Lrows,
                                         last && todo.size() == 1 ? nullptr : lv, su.mtx, su.n_rows, last ? live_list + su.live_offset : nullptr,
                                         last ? c->d_counter + 8 + si : nullptr);
                } else {
                    rc = launch_stats<0>(c, L.k, L.pair, thr_d, c->d_L + L.k * L_per_class, L.pair >= 0 ? c->d_L + (size_t)L.pair * L_per_class : nullptr, Lpitch, Lrows, lv,
                                         su.mtx, su.n_rows);
                }
                if (rc) return rc;
            }
            if (direct) continue;
            const uint32_t nt = (uint32_t)((uint64_t)su.mtx * su.n_rows * c->sub_np);
            hipLaunchKernelGGL(compact_live_tiles, dim3((nt + 256 * CLT_PER_THREAD - 1) / (256 * CLT_PER_THREAD)), dim3(256), 0, c->stream, live + su.live_offset, nt, su.mtx,
                               su.n_rows, 1u, live_list + su.live_offset, c->d_counter + 8 + si);
            FOCR_HIP(c, hipGetLastError());
        }
        FOCR_HIP(c, hipEventRecord(c->ev[1], c->stream));
        if (stats_chain) {
            std::lock_guard<std::mutex> turn(st.mu);
            FOCR_HIP(c, hipEventRecord(st.ev[c->stats_turn % 8], c->stream));
        }
        // 2. MFMA prefilter: one launch per (super-class, bank chunk that fits the LDS budget).
        // With several contexts in flight on one GPU the persistent scan kernels take turns: each context's launches
        // wait (on the device, hipStreamWaitEvent) for the previous context's to finish.  Two of them sharing the
        // MFMA pipes finish no sooner than one after the other; in turn each runs at its full rate while the other
        // contexts' small kernels use the CUs left free by focr_ctx_set_scan_cus.
        if (c->supers.size() > 40) return fail(c, FOCR_ERR_INVALID, "scan_mfma: too many super-classes");
        {
        // (an executor queues its batches from ONE thread in ticket order, pipe.hip: its scans enter this chain in that order
        // with no host-side gate; contexts driven by threads of their own take their turn in the order they get here)
        ScanTurns &tn = scan_turns[(unsigned)c->device % 64];
        std::lock_guard<std::mutex> turn(tn.mu);  // held only while enqueueing
        if (!tn.init) {
            for (hipEvent_t &e : tn.ev) FOCR_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            tn.init = true;
        }
        if (tn.n) FOCR_HIP(c, hipStreamWaitEvent(c->stream, tn.ev[(tn.n - 1) % 8], 0));
        for (size_t si = 0; si < c->supers.size(); si++) {
            const SuperClass &su = c->supers[si];
            if (!su.mtx) continue;
            const uint32_t chunk_tiles = mfma2_chunk_tiles(su.ksteps);
            uint32_t t0 = 0;
            while (t0 < su.n_tiles) {
                MfmaLaunch L{};
                L.layout = su.layout;
                L.ksteps = su.ksteps;
                L.Lpitch = Lpitch;
                L.Lrows = Lrows;
                L.mtx = su.mtx;
                L.n_rows = su.n_rows;
                L.live_list = live_list + su.live_offset;
                L.live_count = c->d_counter + 8 + si;
                L.super_index = (uint32_t)si;
                const uint32_t t_limit = std::min(su.n_tiles, t0 + chunk_tiles);
                uint32_t t1 = t0;
                PlaneArgs A3{};
                for (size_t i = 0; i < su.classes.size() && L.segs.n < (uint32_t)MAX_SEGS; i++) {
                    const SizeClass &sc = c->classes[su.classes[i]];
                    const uint32_t cb = su.tile_first[i], ce = cb + sc.n_tiles16;
                    const uint32_t b = std::max(cb, t1), e = std::min(ce, t_limit);
                    if (b >= e || b != t1) continue;  // segments must tile [t0, t1) contiguously
                    if (!two[si] && (sc.n_w >= c->r_w || sc.n_h >= c->r_h)) {  // no searchable window: skip the class's tiles
                        if (L.segs.n == 0) {
                            t0 = t1 = e;
                            continue;
                        }
                        break;
                    }
                    A3.shift[L.segs.n] = plane_params(c, su.classes[i], thr_d).shift;  // the unit of the class's plane
                    A3.seg_value[L.segs.n] = (uint32_t)i;
                    A3.seg_full[L.segs.n] = (su.layout == LAYOUT_W12 && sc.keep_w <= 8) ? 0u : 1u;  // mfma_common.h: K layouts
                    {  // tiles of the class from n_live / 16 on hold dead / padding slots; chunk-local numbering
                        const uint32_t dead = cb + sc.n_live / 16;
                        A3.seg_dead_from[L.segs.n] = dead > t0 ? dead - t0 : 0u;
                    }
                    MfmaSeg &sg = L.segs.s[L.segs.n++];
                    sg.negL = c->d_L + su.classes[i] * L_per_class;
                    sg.tile_end = e - t0;
                    const uint32_t real = std::min(sc.n_templates, (e - cb) * 16) - (b - cb) * 16;
                    L.n_templates += real;
                    if (sc.n_w < c->r_w && sc.n_h < c->r_h)
                        L.alg_macs += (uint64_t)(c->r_w - sc.n_w) * (c->r_h - sc.n_h) * sc.n_w * sc.n_h * real * c->sub_np;
                    t1 = e;
                }
                if (L.segs.n == 0) {
                    if (t1 == t0) break;
                    continue;
                }
                L.n_tiles16 = t1 - t0;
                L.q_offset = su.q_offset + (size_t)t0 * su.ksteps * 1024;
                L.tg_offset = su.tg_offset + (size_t)t0 * 16;
                const unsigned cus = c->scan_cus ? std::min(c->scan_cus, c->n_cus) : c->n_cus;
                if (c->scan_queues_used >= MAX_SCAN_QUEUES) return fail(c, FOCR_ERR_INVALID, "scan_mfma: too many scan passes for one call (bank too large)");
                L.queue = c->d_counter + COUNTER_WORDS + (size_t)(c->scan_queues_used++) * QUEUE_XCDS * QUEUE_STRIDE;
                if (two[si]) {
                    A3.planes = c->d_planes + plane_off[si];
                    A3.stride = plane;
                    A3.nv = (uint32_t)su.classes.size();
                    if ((rc = dispatch_mfma_v2s(c, L, A3, cus))) return rc;
                } else if ((rc = dispatch_mfma_v2(c, L, cus))) {
                    return rc;
                }
                t0 = t1;
            }
        }
        FOCR_HIP(c, hipEventRecord(tn.ev[tn.n % 8], c->stream));
        tn.n++;
        }
        // tall classes: exact scan straight into the candidate list
        for (size_t k = 0; k < c->classes.size(); k++) {
            const SizeClass &sc = c->classes[k];
            if (!sc.tall || sc.n_w >= c->r_w || sc.n_h >= c->r_h) continue;
            if ((rc = launch_scan_tall(c, k, thr_d, c->d_cand, nullptr, (unsigned long long *)c->d_counter + 1,
                                       (unsigned long long)c->ub_cand, 0)))
                return rc;
        }
        }  // !nothing
        FOCR_HIP(c, hipEventRecord(c->ev[2], c->stream));
        FOCR_HIP(c, hipMemcpyAsync(c->h_live, c->d_counter + 8, 40 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        const unsigned long long *n_cand_p = (const unsigned long long *)c->d_counter + 1;
        size_t ub_c = c->ub_cand;
        if (nothing) use_rows = false;
        // buckets above the row sort's first capacity class get a second launch (rows.hip): estimated sizes go by the previous scan's largest + 25 %
        bool big_expected = (uint64_t)c->est_row_max + c->est_row_max / 4 + 16 > 1024;
        if (!c->estimated) {
            unsigned long long n_cand = 0;
            FOCR_HIP(c, hipMemcpyAsync(&n_cand, n_cand_p, 8, hipMemcpyDeviceToHost, c->stream));
            FOCR_HIP(c, hipStreamSynchronize(c->stream));
            if (n_cand > c->cand_capacity) {
                if (n_cand > ((unsigned long long)1 << 31)) return fail(c, FOCR_ERR_OVERFLOW, "scan_mfma: more than 2^31 candidates in one pass");
                want_cand = (size_t)n_cand + (size_t)n_cand / 8 + 1024;
                continue;
            }
            ub_c = (size_t)n_cand;
        }
        if (use_rows) {
            // 3a'. hits-first row path: verify the candidates where they lie, then bucket + sort the hits only (rows.hip)
            if ((rc = rows2_verify(c, thr_d, n_cand_p, ub_c))) return rc;
            size_t ub_h = std::min(ub_c, c->est_hits);
            bool sort_rows = true;
            if (!c->estimated) {  // exact number of hits and the largest bucket
                uint64_t hits = 0, row_max = 0;
                FOCR_HIP(c, hipMemcpyAsync(&hits, c->d_res + 6, 8, hipMemcpyDeviceToHost, c->stream));
                FOCR_HIP(c, hipMemcpyAsync(&row_max, c->d_res + 5, 8, hipMemcpyDeviceToHost, c->stream));
                FOCR_HIP(c, hipStreamSynchronize(c->stream));
                ub_h = (size_t)hits;
                big_expected = row_max > 1024;
                sort_rows = rows_capacity_for(row_max) != 0;  // a bucket beyond the sort's largest capacity: library sort of the placed hits
                row_cap = sort_rows ? rows_capacity_for(row_max) : 0;
            }
            c->row_cap = row_cap;
            if ((rc = rows2_place(c, n_cand_p, ub_c, ub_h, big_expected, sort_rows))) return rc;
            if (!sort_rows && (rc = sort_pairs_u64_f32(c, c->d_hit_keys, c->d_hit_keys_alt, c->d_hit_sims_alt, c->d_hit_sims, ub_h, c->fmt.bits()))) return rc;
            return order_sorted_hits(c, c->d_hit_keys, c->d_hit_sims_alt, c->d_res + 6, ub_h, n_cand_p, ub_c);
        }
        c->row_cap = 0;
        // 3b. legacy tail: sort the candidates into emission order, verify them exactly in place, compact + cap (order.hip)
        if ((rc = ensure_hit_capacity(c, std::max<size_t>(c->hit_capacity, ub_c + 1)))) return rc;
        uint64_t *flags = (uint64_t *)c->scan_flags.ensure(c, (ub_c + 1) * 8);
        uint64_t *pos = (uint64_t *)c->scan_pos.ensure(c, (ub_c + 1) * 8);
        if (!flags || !pos) return fail(c, FOCR_ERR_NOMEM, "scan_mfma: hipMalloc failed");
        if (c->cand_alt_capacity < c->cand_capacity) {
            FOCR_HIP(c, hipStreamSynchronize(c->stream));
            if (c->d_cand_alt) (void)hipFree(c->d_cand_alt);
            c->d_cand_alt = nullptr;
            c->cand_alt_capacity = 0;
            if (hipMalloc(&c->d_cand_alt, c->cand_capacity * 8) != hipSuccess) return fail(c, FOCR_ERR_NOMEM, "scan_mfma: hipMalloc failed");
            c->cand_alt_capacity = c->cand_capacity;
        }
        if ((rc = sort_keys_u64(c, c->d_cand, c->d_cand_alt, ub_c, c->fmt.bits()))) return rc;
        hipLaunchKernelGGL(verify_kernel, dim3((unsigned)((ub_c + 1 + 255) / 256)), dim3(256), 0, c->stream, c->d_cand, n_cand_p, (unsigned long long)ub_c,
                           verify_args(c, thr_d), c->d_hit_sims, flags);
        FOCR_HIP(c, hipGetLastError());
        FOCR_HIP(c, hipEventRecord(c->ev[3], c->stream));
        if ((rc = compact_candidates(c, c->d_cand, c->d_hit_sims, flags, pos, n_cand_p, ub_c))) return rc;
        size_t ub_h = std::min(ub_c, c->est_hits);
        if (!c->estimated) {  // exact number of hits for the ordering pass
            uint64_t hits = 0;
            FOCR_HIP(c, hipMemcpyAsync(&hits, pos + ub_c, 8, hipMemcpyDeviceToHost, c->stream));
            FOCR_HIP(c, hipStreamSynchronize(c->stream));
            ub_h = (size_t)hits;
        }
        return order_sorted_hits(c, c->d_hit_keys, c->d_hit_sims_alt, pos + ub_c, ub_h, n_cand_p, ub_c);
    }
    return fail(c, FOCR_ERR_OVERFLOW, "scan_mfma: candidate buffer kept overflowing");
}

}  // namespace focr
