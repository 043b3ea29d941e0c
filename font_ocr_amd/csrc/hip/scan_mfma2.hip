// scan_mfma2.hip — the barrier-free i8 MFMA prefilter kernels.
//
// The maths is in scan_mfma.hip's header, the K layouts and the item queues in mfma_common.h.  Every wave is an independent
// worker: it takes items of MT consecutive 16-window M-tiles from its XCD's queue, reads its window fragments straight from
// the page in HBM/L2 (byte-unaligned global loads that land in the MFMA operand registers; the page is read ~16x per class,
// all but the first time from L2), and streams the whole quantised bank chunk past them from LDS.  The bank is staged once
// per workgroup, so a kernel has exactly one barrier; there is no per-tile fill/drain.  One workgroup per CU.
//
//   scan_mfma2s_kernel  the default (<= 4 K-steps, <= 4 size classes per pass): A = templates, B = windows, the C-in of a
//                       lane is the threshold of its own window, one shift away from its int16 threshold-plane value; 16
//                       waves x 4 M-tiles per CU, 128 VGPRs.
//   scan_mfma2_kernel   round 1's form (A = windows, int32 negL rows re-loaded per size class), kept for 5..8 K-steps and
//                       as a cross-check (FOCR_PREFILTER_LEGACY); 8 waves x 4..8 M-tiles for the long layouts.
#include <algorithm>

#include "mfma_common.h"

namespace focr {

typedef v4i v4i_u __attribute__((aligned(1)));  // byte-aligned views: gfx950 global loads take any alignment
typedef int v2i __attribute__((ext_vector_type(2)));
typedef v2i v2i_u __attribute__((aligned(1)));
typedef int v3i __attribute__((ext_vector_type(3)));
typedef v3i v3i_u __attribute__((aligned(1)));
typedef int int_u __attribute__((aligned(1)));

// N-tiles of one launch: the whole 160 KiB of a CU's LDS belong to its one workgroup — ksteps KiB of quantised templates
// + 64 B of template ids per tile, next to the waves' candidate staging buffers.  As few launches per super-class as
// possible: every launch re-loads the window fragments (BASELINE configs[2]'s 95 N-tiles at 3 K-steps: 49 + 46 tiles in two
// launches instead of the three a 136 KiB budget gave).
uint32_t mfma2_chunk_tiles(uint32_t ksteps) {
    const size_t fixed = (size_t)16 * WBUF * 8 + 256;
    return (uint32_t)(((160u << 10) - fixed) / ((size_t)ksteps * 1024 + 64));
}

template <int KSTEPS, int RPG, int MT, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void scan_mfma2_kernel(
    const uint8_t *__restrict__ pages, uint32_t pitch, uint32_t rows_alloc, const uint64_t *__restrict__ live_list,
    const uint32_t *__restrict__ live_count, uint32_t page_base, const v4i *__restrict__ qbank, uint32_t n_tiles16, const MfmaSegs segs, uint32_t Lpitch, uint32_t Lrows,
    const uint32_t *__restrict__ tglobal, const KeyFmt fmt, uint64_t *__restrict__ cand,
    unsigned long long *__restrict__ cand_counter, unsigned long long cand_cap, uint32_t *__restrict__ queue, const RowHist rows) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    v4i *bank = reinterpret_cast<v4i *>(smem);
    const uint32_t bank_vec = n_tiles16 * KSTEPS * 64;
    uint32_t *tg_lds = reinterpret_cast<uint32_t *>(smem + (size_t)bank_vec * 16 + (size_t)NW * WBUF * 8);
    for (uint32_t i = threadIdx.x; i < bank_vec; i += NW * 64) bank[i] = qbank[i];
    for (uint32_t i = threadIdx.x; i < n_tiles16 * 16; i += NW * 64) tg_lds[i] = tglobal[i];
    __syncthreads();  // the only barrier

    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index as a scalar: all tile coordinates stay in SGPRs
    uint64_t *wbuf = reinterpret_cast<uint64_t *>(smem + (size_t)bank_vec * 16) + w * WBUF;
    uint32_t wcount = 0;  // wave-uniform number of staged candidates

    const uint32_t total_mt = *live_count;  // live 16-window M-tiles of this pass (blank paper is skipped)
    const uint32_t n_items = (total_mt + MT - 1) / MT;
    ItemTaker take;  // XCD-aware split of the work list + item queue (mfma_common.h)
    take.init(queue, n_items, lane);

    v4i afrag[MT][KSTEPS];
    for (uint32_t item; take.next(item);) {
        // coordinates of the item's M-tiles (wave-uniform, scalar loads)
        const uint32_t m0 = item * MT;
        uint32_t px[MT], py[MT], pp[MT];
        bool pv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            pv[mt] = m0 + mt < total_mt;
            const uint64_t e = live_list[pv[mt] ? m0 + mt : total_mt - 1];
            px[mt] = 16 * (uint32_t)(e & 0xfff);
            py[mt] = 1 + (uint32_t)((e >> 12) & 0xfffff);  // y = 0 is never searched (src/ncc.cpp:302)
            pp[mt] = (uint32_t)(e >> 32);
        }

        size_t loff[MT];
        v4i nl[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            loff[mt] = ((size_t)pp[mt] * Lrows + py[mt]) * Lpitch + px[mt] + 4 * g;
            nl[mt] = *reinterpret_cast<const v4i *>(segs.s[0].negL + loff[mt]);
        }
        // A fragments: lane (r, g) of K-step ks holds the 16 bytes of k-group 4*ks+g of window px+r.
        // Byte-unaligned 16-byte (8-byte) global loads land directly in the MFMA operand registers.
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const uint8_t *base = pages + ((size_t)pp[mt] * rows_alloc + py[mt]) * pitch + px[mt] + r;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) {
                v4i a;
                if (RPG == LAYOUT_W16) {
                    a = *reinterpret_cast<const v4i_u *>(base + (size_t)(4 * ks + g) * pitch);
                } else if (RPG == LAYOUT_W8) {
                    const uint8_t *p0 = base + (size_t)(2 * (4 * ks + g)) * pitch;
                    const v2i lo = *reinterpret_cast<const v2i_u *>(p0), hi = *reinterpret_cast<const v2i_u *>(p0 + pitch);
                    a = v4i{lo[0], lo[1], hi[0], hi[1]};
                } else {
                    // LAYOUT_W12: K-step 3*(m/4) + c = dword column c of the rows of quad m = 4*(ks/3)+g (mfma_common.h)
                    const uint8_t *q0 = base + (size_t)(4 * (4 * (ks / 3) + g)) * pitch + 4 * (ks % 3);
                    a = v4i{*reinterpret_cast<const int_u *>(q0), *reinterpret_cast<const int_u *>(q0 + pitch),
                            *reinterpret_cast<const int_u *>(q0 + 2 * (size_t)pitch), *reinterpret_cast<const int_u *>(q0 + 3 * (size_t)pitch)};
                }
                afrag[mt][ks] = a;  // the page copy is already int8 (ink - 128, written at ingest; the templates sum to zero: the bias cancels exactly)
            }
        }
        take.request();  // the next item's ticket
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {  // M-tiles past the end of the enumeration never flag
            const int keep = pv[mt] ? -1 : 0;
            nl[mt] = (nl[mt] & keep) | (v4i{-REJECT, -REJECT, -REJECT, -REJECT} & ~keep);
        }
        // B fragments of N-tile nt are in registers before the tile starts; each one is re-loaded for
        // N-tile nt+1 right after its last MFMA has issued, so the LDS latency hides behind the other K-steps.
        v4i bf[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) bf[ks] = bank[ks * 64 + lane];
        uint32_t nt = 0;
        for (uint32_t sgi = 0; sgi < segs.n; sgi++) {  // one segment = the N-tiles of one size class
            const uint32_t seg_end = segs.s[sgi].tile_end;
            if (sgi) {  // next size class of the pass -> its C-in table (scalar loads stay outside the N-tile loop)
                const int32_t *tab = segs.s[sgi].negL;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) nl[mt] = *reinterpret_cast<const v4i *>(tab + loff[mt]);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const int keep = pv[mt] ? -1 : 0;
                    nl[mt] = (nl[mt] & keep) | (v4i{-REJECT, -REJECT, -REJECT, -REJECT} & ~keep);
                }
            }
        for (; nt < seg_end; nt++) {
            v4i acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = nl[mt];
            const uint32_t nxt = nt + 1 < n_tiles16 ? nt + 1 : nt;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[mt][ks], bf[ks], acc[mt], 0, 0, 0);
                bf[ks] = bank[(nxt * KSTEPS + ks) * 64 + lane];
                // pin the re-load here: left alone, hipcc sinks all of them behind the last K-step and the
                // next N-tile then opens with a full LDS round trip
                __builtin_amdgcn_sched_barrier(0);
            }
            // any output > 0 ?  two v_max3_i32 per M-tile; the per-M-tile split is redone on the rare path only
            int m = max(max(acc[0][0], acc[0][1]), max(acc[0][2], acc[0][3]));
#pragma unroll
            for (int mt = 1; mt < MT; mt++) {
                m = max(max(m, acc[mt][0]), acc[mt][1]);
                m = max(max(m, acc[mt][2]), acc[mt][3]);
            }
            if (__builtin_amdgcn_ballot_w64(m > 0) != 0) {  // wave-uniform; taken for roughly one N-tile in ten
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const int mmt = max(max(acc[mt][0], acc[mt][1]), max(acc[mt][2], acc[mt][3]));
                    if (__builtin_amdgcn_ballot_w64(mmt > 0) == 0) continue;  // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const bool f = acc[mt][i] > 0;
                        const uint64_t mask = __builtin_amdgcn_ballot_w64(f);
                        if (mask) {  // wave-uniform
                            // dead / padding templates (tglobal = ~0) never emit; they only get here for thr <= 0
                            const uint32_t tg = f ? tg_lds[nt * 16 + r] : 0xffffffffu;
                            const bool ok = tg != 0xffffffffu;
                            const uint64_t okmask = __builtin_amdgcn_ballot_w64(ok);
                            const uint32_t cnt = (uint32_t)__builtin_popcountll(okmask);
                            if (cnt) {
                                if (wcount + cnt > WBUF) {
                                    flush_wave_candidates(wbuf, wcount, lane, cand, cand_counter, cand_cap, rows);
                                    wcount = 0;
                                }
                                const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(okmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)okmask, 0u));
                                // keep the key arithmetic inside this rare block (opaque inputs stop the compiler
                                // from hoisting 8 M-tiles' worth of 64-bit keys out of the N-tile loop and spilling them)
                                uint32_t pg = pp[mt], yy = py[mt], xx = px[mt];
                                asm volatile("" : "+s"(pg), "+s"(yy), "+s"(xx));
                                if (ok) wbuf[wcount + pos] = fmt.pack(page_base + pg, yy, xx + 4 * g + i, tg);
                                wcount += cnt;
                            }
                        }
                    }
                }
            }
        }
        }
    }
    if (wcount) flush_wave_candidates(wbuf, wcount, lane, cand, cand_counter, cand_cap, rows);
}

// ---------------------------------------------------------------------------------------------
// The same one-stage prefilter with the operand roles swapped: (A = templates, B = windows), so D[template][window] puts
// window px + r on lane (r, g) for all four accumulator registers.  The C-in of a lane is then the threshold of its OWN
// window: one int16 threshold-plane value per size class and M-tile (mfma_common.h) is loaded once per item and becomes the
// int32 C-in with one shift — no per-class int32 table, no reload of C-in rows in the middle of the N-tile loop when the size
// class changes.
// 16 waves per workgroup, one workgroup per CU, 128 VGPRs (4 waves per SIMD).  Measured and not adopted (dead ends): 12- and
// 8-wave workgroups, 96 VGPRs (a fifth wave slot per SIMD left to other kernels).
constexpr int V2S_NW = 16, V2S_OCC = 4;
template <int KSTEPS, int RPG, int MT, int NW, int NV>
__global__ __launch_bounds__(NW * 64, V2S_OCC) void scan_mfma2s_kernel(
    const uint8_t *__restrict__ pages, uint32_t pitch, uint32_t rows_alloc, const uint64_t *__restrict__ live_list,
    const uint32_t *__restrict__ live_count, uint32_t page_base, const v4i *__restrict__ qbank, uint32_t n_tiles16, const MfmaSegs segs, uint32_t Lpitch, uint32_t Lrows,
    const PlaneArgs P, const uint32_t *__restrict__ tglobal, const KeyFmt fmt, uint64_t *__restrict__ cand,
    unsigned long long *__restrict__ cand_counter, unsigned long long cand_cap, uint32_t *__restrict__ queue, const RowHist rows) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    v4i *bank = reinterpret_cast<v4i *>(smem);
    const uint32_t bank_vec = n_tiles16 * KSTEPS * 64;
    uint32_t *tg_lds = reinterpret_cast<uint32_t *>(smem + (size_t)bank_vec * 16 + (size_t)NW * WBUF * 8);
    for (uint32_t i = threadIdx.x; i < bank_vec; i += NW * 64) bank[i] = qbank[i];
    for (uint32_t i = threadIdx.x; i < n_tiles16 * 16; i += NW * 64) tg_lds[i] = tglobal[i];
    __syncthreads();  // the only barrier

    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint64_t *wbuf = reinterpret_cast<uint64_t *>(smem + (size_t)bank_vec * 16) + w * WBUF;
    uint32_t wcount = 0;

    const uint32_t total_mt = *live_count;
    const uint32_t n_items = (total_mt + MT - 1) / MT;
    uint32_t shift_of_value[NV];  // log2 of the unit of the threshold plane per value (= per size class of the super-class)
#pragma unroll
    for (int v = 0; v < NV; v++) {
        shift_of_value[v] = 0;
#pragma unroll
        for (int sg = 0; sg < MAX_PLANE_VALUES; sg++)
            if ((uint32_t)sg < segs.n && P.seg_value[sg] == (uint32_t)v) shift_of_value[v] = P.shift[sg];
    }

    // byte offsets of the lane's rows inside an M-tile's fragment, relative to the wave-uniform base (layouts: mfma_common.h)
    uint32_t lane_off[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        lane_off[j] = RPG == LAYOUT_W16 ? (uint32_t)r + (uint32_t)g * pitch
                      : RPG == LAYOUT_W8 ? (uint32_t)r + (uint32_t)(2 * g + (j & 1)) * pitch
                                         : (uint32_t)r + (uint32_t)(4 * g + j) * pitch;
    const uint32_t r_key = (uint32_t)r << fmt.bt;  // the lane's window inside its M-tile, in key position
    uint32_t plane_off[NV];  // the lane's own window inside an M-tile's entries of plane v, in bytes (all planes of a pass span < 4 GiB: launch_scan_mfma)
#pragma unroll
    for (int v = 0; v < NV; v++) plane_off[v] = (uint32_t)r * 2 + (uint32_t)v * (uint32_t)(P.stride * 2);

    ItemTaker take;
    take.init(queue, n_items, lane);
    v4i afrag[MT][KSTEPS];
    for (uint32_t item; take.next(item);) {
        const uint32_t m0 = item * MT;
        uint32_t px[MT], py[MT], pp[MT];
        bool pv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            pv[mt] = m0 + mt < total_mt;
            const uint64_t e = live_list[pv[mt] ? m0 + mt : total_mt - 1];
            px[mt] = 16 * (uint32_t)(e & 0xfff);
            py[mt] = 1 + (uint32_t)((e >> 12) & 0xfffff);  // y = 0 is never searched (src/ncc.cpp:302)
            pp[mt] = (uint32_t)(e >> 32);
        }
        // The lane offsets are made opaque here, once per item: left visible, "pages + lane offset" is loop-invariant and gets
        // hoisted as a 64-bit vector pointer, and every load then pays a 64-bit vector add of its uniform part; opaque, the
        // uniform part stays in scalar registers and the load takes "scalar base + 32-bit vector offset" directly.
        uint32_t lo[4] = {lane_off[0], lane_off[1], lane_off[2], lane_off[3]}, po[NV];
        asm volatile("" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]));
#pragma unroll
        for (int v = 0; v < NV; v++) {
            po[v] = plane_off[v];
            asm volatile("" : "+v"(po[v]));
        }
        int16_t nrm[MT][NV];  // threshold-plane values of the lane's own window px + r, one per size class
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const uint8_t *np = reinterpret_cast<const uint8_t *>(P.planes + ((size_t)pp[mt] * Lrows + py[mt]) * Lpitch + px[mt]);  // wave-uniform
#pragma unroll
            for (int v = 0; v < NV; v++)  // -floor((L - 2) / S) as int16 (a sign-extending load); -32768 = never
                nrm[mt][v] = *reinterpret_cast<const int16_t *>(np + po[v]);
        }
        // K-step-major issue order: the N-tile loop's first MFMAs need K-step 0 of all M-tiles.  An address is a wave-uniform
        // 64-bit base (page, row, M-tile, K-step: scalar arithmetic) plus a 32-bit lane offset that never changes (lane_off[],
        // computed once per kernel) — no vector instruction per load; the pages are the int8 copy, so no v_xor either.
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const uint8_t *ub = pages + ((size_t)pp[mt] * rows_alloc + py[mt]) * pitch + px[mt];  // wave-uniform
                v4i a;
                if (RPG == LAYOUT_W16) {
                    a = *reinterpret_cast<const v4i_u *>(ub + (size_t)(4 * ks) * pitch + lo[0]);
                } else if (RPG == LAYOUT_W8) {
                    const uint8_t *u0 = ub + (size_t)(8 * ks) * pitch;
                    const v2i w0 = *reinterpret_cast<const v2i_u *>(u0 + lo[0]), w1 = *reinterpret_cast<const v2i_u *>(u0 + lo[1]);
                    a = v4i{w0[0], w0[1], w1[0], w1[1]};
                } else {
                    // LAYOUT_W12: K-step 3*(m/4) + c = dword column c of the rows of quad m = 4*(ks/3)+g (mfma_common.h)
                    const uint8_t *u0 = ub + (size_t)(16 * (ks / 3)) * pitch + 4 * (ks % 3);
                    a = v4i{*reinterpret_cast<const int_u *>(u0 + lo[0]), *reinterpret_cast<const int_u *>(u0 + lo[1]),
                            *reinterpret_cast<const int_u *>(u0 + lo[2]), *reinterpret_cast<const int_u *>(u0 + lo[3])};
                }
                afrag[mt][ks] = a;
            }
        }
        take.request();  // the next item's ticket
        // C-in of the lane's own window per size class (prefilter_cin, mfma_common.h: one shift; a window the class never emits at
        // holds -32768, an unreachable threshold).  M-tiles past the end of the enumeration repeat the last live one:
        // they are dropped where candidates are emitted (rare path), not here
        int cin[MT][NV];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int v = 0; v < NV; v++) {
                cin[mt][v] = prefilter_cin(shift_of_value[v], nrm[mt][v]);
            }
        v4i bf[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) bf[ks] = bank[ks * 64 + lane];
        uint32_t nt = 0;
        for (uint32_t sgi = 0; sgi < segs.n; sgi++) {  // one segment = the N-tiles of one size class
            const uint32_t seg_end = segs.s[sgi].tile_end, sv = P.seg_value[sgi];
            const bool full = P.seg_full[sgi] != 0;  // false: the class's templates are all zero in the last K-step (12-byte rows, n_w <= 8)
            const uint32_t dead_from = P.seg_dead_from[sgi];
            int ci[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                ci[mt] = cin[mt][0];
#pragma unroll
                for (int v = 1; v < NV; v++) ci[mt] = sv == (uint32_t)v ? cin[mt][v] : ci[mt];
            }
            for (; nt < seg_end; nt++) {
                v4i acc[MT];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) acc[mt] = v4i{ci[mt], ci[mt], ci[mt], ci[mt]};
                const uint32_t nxt = nt + 1 < n_tiles16 ? nt + 1 : nt;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ks++) {
                    if (RPG != LAYOUT_W12 || ks + 1 < KSTEPS || full) {  // wave-uniform
#pragma unroll
                        for (int mt = 0; mt < MT; mt++)
                            acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bf[ks], afrag[mt][ks], acc[mt], 0, 0, 0);
                    }
                    bf[ks] = bank[(nxt * KSTEPS + ks) * 64 + lane];
                    __builtin_amdgcn_sched_barrier(0);  // pin the re-load here (see scan_mfma2_kernel)
                }
                int m = max(max(acc[0][0], acc[0][1]), max(acc[0][2], acc[0][3]));
#pragma unroll
                for (int mt = 1; mt < MT; mt++) {
                    m = max(max(m, acc[mt][0]), acc[mt][1]);
                    m = max(max(m, acc[mt][2]), acc[mt][3]);
                }
                if (__builtin_amdgcn_ballot_w64(m > 0) != 0) {  // wave-uniform; taken for roughly one N-tile in ten
                    // The candidate path is kept short (a sixth of the kernel at BASELINE configs[1], where six windows in ten that pass
                    // are real matches): ONE LDS read per visit for the lane's four template ids (lane (r, g), register i: template
                    // 4g + i of the tile, window px + r), one compare per register — its mask is the ballot — and a key whose high
                    // word and low-word base are formed once per M-tile (x = px + r and the template id fill disjoint bit fields of
                    // the low word: launch_v2s checks bt + bx <= 32), so a staged key costs one v_or3 and one LDS write.
                    v4i tg4 = reinterpret_cast<const v4i *>(tg_lds)[nt * 4 + g];
                    if (nt >= dead_from) {  // the class's last tiles: dead / padding slots (id ~0) never emit.  Wave-uniform, rare
#pragma unroll
                        for (int mt = 0; mt < MT; mt++)
#pragma unroll
                            for (int i = 0; i < 4; i++) acc[mt][i] = tg4[i] == -1 ? -1 : acc[mt][i];
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        if (!pv[mt]) continue;  // past the end of the enumeration (a repeat of the last live M-tile): wave-uniform
                        const int mmt = max(max(acc[mt][0], acc[mt][1]), max(acc[mt][2], acc[mt][3]));
                        if (__builtin_amdgcn_ballot_w64(mmt > 0) == 0) continue;  // wave-uniform
                        uint32_t pg = pp[mt], yy = py[mt], xx = px[mt];
                        asm volatile("" : "+s"(pg), "+s"(yy), "+s"(xx));  // keep the key arithmetic inside this rare block
                        const uint64_t kb = fmt.pack(page_base + pg, yy, xx, 0);  // scalar
                        const uint32_t k_hi = (uint32_t)(kb >> 32), k_lo = (uint32_t)kb | r_key;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const bool ok = acc[mt][i] > 0;
                            const uint64_t okmask = __builtin_amdgcn_ballot_w64(ok);
                            if (!okmask) continue;  // wave-uniform
                            const uint32_t cnt = (uint32_t)__builtin_popcountll(okmask);
                            if (wcount + cnt > WBUF) {
                                flush_wave_candidates(wbuf, wcount, lane, cand, cand_counter, cand_cap, rows);
                                wcount = 0;
                            }
                            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(okmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)okmask, 0u));
                            if (ok) reinterpret_cast<uint2 *>(wbuf)[wcount + pos] = uint2{k_lo | (uint32_t)tg4[i], k_hi};
                            wcount += cnt;
                        }
                    }
                }
            }
        }
    }
    if (wcount) flush_wave_candidates(wbuf, wcount, lane, cand, cand_counter, cand_cap, rows);
}

template <int KSTEPS, int RPG, int NV>
static void launch_v2s(focr_ctx *c, const MfmaLaunch &L, const PlaneArgs &A3, unsigned n_cus) {
    constexpr int MT = 4, NW = V2S_NW;
    const uint32_t n_tiles16 = L.n_tiles16;
    const size_t lds = (size_t)n_tiles16 * KSTEPS * 1024 + (size_t)NW * WBUF * 8 + (size_t)n_tiles16 * 16 * 4;
    const uint64_t total_mt = (uint64_t)L.mtx * L.n_rows * c->sub_np;
    const uint64_t n_items = (total_mt + MT - 1) / MT;
    unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)n_cus * (16 / NW), (n_items + NW - 1) / NW);  // 16 waves per CU: one workgroup, or two of 8 waves (experiment builds)
    auto kern = scan_mfma2s_kernel<KSTEPS, RPG, MT, NW, NV>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    uint64_t ksteps_issued = 0;  // K-steps per live M-tile: the last one is skipped for classes that are all zero there
    for (uint32_t i = 0, t = 0; i < L.segs.n; t = L.segs.s[i].tile_end, i++) ksteps_issued += (uint64_t)(L.segs.s[i].tile_end - t) * (KSTEPS - (A3.seg_full[i] ? 0 : 1));
    const uint64_t issued = 16 * ksteps_issued * 16 * 64;  // per live M-tile; scaled by the live count after the scan
    char name[64];
    snprintf(name, sizeof name, "scan_mfma2s_kernel<%d,%d,%d,%d>", KSTEPS, RPG, MT, NW);
    c->launch_begin(name, L.n_templates | (L.super_index << 24), L.alg_macs, issued);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, c->stream, c->d_pages_i8 + c->sub_p0 * c->rows_alloc * c->pitch, (uint32_t)c->pitch, (uint32_t)c->rows_alloc,
                       L.live_list, L.live_count, (uint32_t)c->sub_p0, reinterpret_cast<const v4i *>(c->d_qbank + L.q_offset), n_tiles16, L.segs, L.Lpitch, L.Lrows, A3,
                       c->d_tglobal + L.tg_offset, c->fmt, c->d_cand, (unsigned long long *)c->d_counter + 1, (unsigned long long)c->ub_cand, L.queue, c->row_hist);
    c->launch_end();
}


int dispatch_mfma_v2s(focr_ctx *c, const MfmaLaunch &L, const PlaneArgs &A3, unsigned n_cus) {
    // the kernel forms a key's low word from disjoint bit fields (x and template id); banks hold <= 65535 templates, pages <= 65535 px
    if (c->fmt.bt + c->fmt.bx > 32) return fail(c, FOCR_ERR_INVALID, "scan_mfma2s: key format too wide");
    const uint32_t nvp = A3.nv <= 1 ? 1 : (A3.nv <= 2 ? 2 : 4);
#define CASE2S(K, R)                                            \
    case (K) * 10 + (R):                                        \
        if (nvp == 1) launch_v2s<K, R, 1>(c, L, A3, n_cus);     \
        else if (nvp == 2) launch_v2s<K, R, 2>(c, L, A3, n_cus); \
        else launch_v2s<K, R, 4>(c, L, A3, n_cus);              \
        break;
    switch (L.ksteps * 10 + L.layout) {
        CASE2S(1, 1) CASE2S(2, 1) CASE2S(3, 1) CASE2S(4, 1)
        CASE2S(1, 2) CASE2S(2, 2) CASE2S(3, 2) CASE2S(4, 2)
        CASE2S(3, 3)
        default: return fail(c, FOCR_ERR_INVALID, "scan_mfma2s: unsupported size class");
    }
#undef CASE2S
    FOCR_HIP(c, hipGetLastError());
    return FOCR_OK;
}

template <int KSTEPS, int RPG, int MT, int NW>
static void launch_v2(focr_ctx *c, const MfmaLaunch &L, unsigned n_cus) {
    const uint32_t n_tiles16 = L.n_tiles16;
    const size_t lds = (size_t)n_tiles16 * KSTEPS * 1024 + (size_t)NW * WBUF * 8 + (size_t)n_tiles16 * 16 * 4;
    const uint64_t total_mt = (uint64_t)L.mtx * L.n_rows * c->sub_np;  // upper bound; the live count is on the device
    const uint64_t n_items = (total_mt + MT - 1) / MT;
    unsigned grid = (unsigned)std::min<uint64_t>(n_cus, (n_items + NW - 1) / NW);
    auto kern = scan_mfma2_kernel<KSTEPS, RPG, MT, NW>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const v4i *qb = reinterpret_cast<const v4i *>(c->d_qbank + L.q_offset);
    const uint64_t issued = 16 * (uint64_t)n_tiles16 * 16 * KSTEPS * 64;  // per live M-tile; scaled by the live count after the scan
    char name[64];
    snprintf(name, sizeof name, "scan_mfma2_kernel<%d,%d,%d,%d>", KSTEPS, RPG, MT, NW);
    c->launch_begin(name, L.n_templates | (L.super_index << 24), L.alg_macs, issued);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, c->stream, c->d_pages_i8 + c->sub_p0 * c->rows_alloc * c->pitch, (uint32_t)c->pitch, (uint32_t)c->rows_alloc,
                       L.live_list, L.live_count, (uint32_t)c->sub_p0, qb, n_tiles16, L.segs, L.Lpitch, L.Lrows, c->d_tglobal + L.tg_offset,
                       c->fmt, c->d_cand, (unsigned long long *)c->d_counter + 1,
                       (unsigned long long)c->ub_cand, L.queue, c->row_hist);
    c->launch_end();
}

int dispatch_mfma_v2(focr_ctx *c, const MfmaLaunch &L, unsigned n_cus) {
    const uint32_t ks = L.ksteps, rpg = L.layout;
    // 16 waves x 4 M-tiles per CU measured best for <= 4 K-steps (8 x 8 and 12 x 6 were 6-9 % slower at C2); the
    // 5..8 K-step layouts need the registers of 8 waves per CU.
#define CASE(K, R, M)                                           \
    case (K) * 10 + (R):                                        \
        if ((K) <= 4) launch_v2<K, R, 4, 16>(c, L, n_cus);      \
        else launch_v2<K, R, M, 8>(c, L, n_cus);                \
        break;
    switch (ks * 10 + rpg) {
        CASE(1, 1, 8) CASE(2, 1, 8) CASE(3, 1, 8) CASE(4, 1, 8) CASE(5, 1, 4) CASE(6, 1, 4) CASE(7, 1, 4) CASE(8, 1, 4)
        CASE(1, 2, 8) CASE(2, 2, 8) CASE(3, 2, 8) CASE(4, 2, 8)
        CASE(3, 3, 8) CASE(6, 3, 4)
        default: return fail(c, FOCR_ERR_INVALID, "scan_mfma: unsupported size class");
    }
#undef CASE
    FOCR_HIP(c, hipGetLastError());
    return FOCR_OK;
}

}  // namespace focr
