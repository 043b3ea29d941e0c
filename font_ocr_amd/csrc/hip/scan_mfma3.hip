// scan_mfma3.hip — two-stage MFMA prefilter (the maths and the host data: lowrank.hip).
//
// Same worker structure as scan_mfma2.hip (every wave an independent worker over MT 16-window M-tiles of the live-tile
// list, A fragments straight from HBM/L2, one barrier after staging the bank), but per item:
//
//   stage 1   y = U^ (a - 128): the int8 basis (2 N-tiles = 32 rows, the unused ones zero) against the window fragments.
//             Operand roles are (A = basis, B = windows), so lane (r, g) ends up with rows 4g..4g+3 of both basis tiles
//             for ITS OWN window px + r — exactly the 8 K-slots {16 b + 4 g + v} that lane group g feeds to stage 2:
//             no lane movement between the stages.
//   mid       y -> bf16; sum of squares over the lane groups (2 shuffles) -> R(w); R, N_F and the per-class threshold
//             slots theta_c = thr_lo * norm_c(w) go into their compile-time element positions (mfma_common.h).
//   stage 2   one v_mfma_f32_16x16x32_bf16 per (M-tile, N-tile): D2[template][window]; any D2 > 0 sets bit nt of the
//             M-tile's mask.  No branch, no candidate handling in this loop.
//   stage 3   for the set bits only: the exact-taps int8 stage of scan_mfma2.hip on that (M-tile, N-tile) block, again as
//             (A = templates, B = windows), so the C-in of lane (r, g) is the threshold of its own window for all four
//             registers: -(floor(kq * norm_c) - 2), from the norm the lane already holds.  Survivors are candidates.
//
// Window norms come from scan_mfma.hip's stats_kernel<.., NORMS = true> (one plane per size class, one for the frame).
#include <algorithm>

#include "mfma_common.h"

namespace focr {

typedef v4i v4i_u __attribute__((aligned(1)));  // byte-aligned views: gfx950 global loads take any alignment
typedef int v2i __attribute__((ext_vector_type(2)));
typedef v2i v2i_u __attribute__((aligned(1)));
typedef int v3i __attribute__((ext_vector_type(3)));
typedef v3i v3i_u __attribute__((aligned(1)));
typedef int int_u __attribute__((aligned(1)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
constexpr size_t V3_LDS_TOTAL = 160 << 10;

// N-tiles of one launch: i8 templates (ksteps KiB) + bf16 stage-2 operand (1 KiB) + template ids (64 B) per tile, next to
// the basis, the candidate staging buffers and a little slack; the per-M-tile masks have 64 bits
uint32_t mfma3_chunk_tiles(uint32_t ksteps) {
    const size_t fixed = (size_t)LR_BASIS_TILES * ksteps * 1024 + (size_t)16 * WBUF * 8 + 1024;
    return (uint32_t)std::min<size_t>(64, (V3_LDS_TOTAL - fixed) / ((size_t)ksteps * 1024 + 1024 + 64));
}

__device__ __forceinline__ uint32_t pack_bf16_bits(uint32_t hi_f32_bits, uint32_t lo_f32_bits) {
    return __builtin_amdgcn_perm(hi_f32_bits, lo_f32_bits, 0x07060302u);  // {hi[31:16], lo[31:16]}
}

#ifdef FOCR_MFMA3_PROF
// experiment builds only (make hip EXTRA=-DFOCR_MFMA3_PROF): per-phase wave cycles, summed over all waves
__device__ unsigned long long focr_prof[8];
#define PROF_STAMP(i)                                        \
    {                                                        \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        prof_acc[i] += now_ - prof_t;                        \
        prof_t = now_;                                       \
    }
#else
#define PROF_STAMP(i)
#endif

// wave64 AND-reduction via DPP (row_shr 1/2/4/8 inside the rows of 16, row_bcast 15/31 across them); lanes without a
// source keep the identity ~0.  The result is uniform (read from lane 63).
__device__ __forceinline__ uint32_t wave_and(uint32_t v) {
    v &= (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false);
    v &= (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false);
    v &= (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false);
    v &= (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false);
    v &= (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xa, 0xf, false);
    v &= (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <int KSTEPS, int RPG, int MT, int NW, int NV>
__global__ __launch_bounds__(NW * 64, NW / 4) void scan_mfma3_kernel(
    const uint8_t *__restrict__ pages, uint32_t pitch, uint32_t rows_alloc, const uint64_t *__restrict__ live_list,
    const uint32_t *__restrict__ live_count, uint32_t page_base, const v4i *__restrict__ qbank, const v4i *__restrict__ basis_g,
    const v4i *__restrict__ gbank_g, uint32_t n_tiles16, const MfmaSegs segs, uint32_t Lpitch, uint32_t Lrows, const Mfma3Args P,
    const uint32_t *__restrict__ tglobal, const KeyFmt fmt, uint64_t *__restrict__ cand,
    unsigned long long *__restrict__ cand_counter, unsigned long long cand_cap, uint32_t *__restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem3[];
    v4i *bank = reinterpret_cast<v4i *>(smem3);
    const uint32_t bank_vec = n_tiles16 * KSTEPS * 64;
    v4i *basis = bank + bank_vec;
    constexpr uint32_t basis_vec = LR_BASIS_TILES * KSTEPS * 64;
    v4i *gb = basis + basis_vec;
    const uint32_t gb_vec = n_tiles16 * 64;
    uint64_t *wbuf_all = reinterpret_cast<uint64_t *>(gb + gb_vec);
    uint32_t *tg_lds = reinterpret_cast<uint32_t *>(wbuf_all + (size_t)NW * WBUF);
    for (uint32_t i = threadIdx.x; i < bank_vec; i += NW * 64) bank[i] = qbank[i];
    for (uint32_t i = threadIdx.x; i < basis_vec; i += NW * 64) basis[i] = basis_g[i];
    for (uint32_t i = threadIdx.x; i < gb_vec; i += NW * 64) gb[i] = gbank_g[i];
    for (uint32_t i = threadIdx.x; i < n_tiles16 * 16; i += NW * 64) tg_lds[i] = tglobal[i];
    __syncthreads();  // the only barrier

    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint64_t *wbuf = wbuf_all + w * WBUF;
    uint32_t wcount = 0;

    const uint32_t total_mt = *live_count;
    const uint32_t n_items = (total_mt + MT - 1) / MT;
    ItemTaker take;  // XCD-aware split of the work list + item queue (mfma_common.h)
    take.init(queue, n_items, lane);
    const uint32_t n_extras = 2 + P.n_cls;
    // segments (size classes) of the launch: at most LR_MAX_CLASSES, boundaries and norm-value indices as scalars
    const uint32_t seg_end0 = segs.n > 0 ? segs.s[0].tile_end : 0xffffffffu, seg_end1 = segs.n > 1 ? segs.s[1].tile_end : 0xffffffffu;
    const uint32_t seg_end2 = segs.n > 2 ? segs.s[2].tile_end : 0xffffffffu;
    const uint32_t seg_val0 = P.seg_value[0], seg_val1 = P.seg_value[1], seg_val2 = P.seg_value[2], seg_val3 = P.seg_value[3];
    float kq_of_value[NV];  // threshold scale of the exact-taps stage per norm value (= per class of the super-class)
#pragma unroll
    for (int v = 0; v < NV; v++) {
        kq_of_value[v] = 0.f;
#pragma unroll
        for (int sg = 0; sg < (int)LR_MAX_CLASSES; sg++)
            if ((uint32_t)sg < segs.n && P.seg_value[sg] == (uint32_t)v) kq_of_value[v] = P.kq[sg];
    }

#ifdef FOCR_MFMA3_PROF
    unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
#endif
#ifdef FOCR_MFMA3_PROF
    unsigned long long prof_items = 0;
#endif
    for (uint32_t item; take.next(item);) {
#ifdef FOCR_MFMA3_PROF
        prof_items++;
#endif
        const uint32_t m0 = item * MT;
        uint32_t px[MT], py[MT], pp[MT];
        bool pv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            pv[mt] = m0 + mt < total_mt;
            const uint64_t e = live_list[pv[mt] ? m0 + mt : total_mt - 1];
            px[mt] = 16 * (uint32_t)(e & 0xfff);
            py[mt] = 1 + (uint32_t)((e >> 12) & 0xfffff);
            pp[mt] = (uint32_t)(e >> 32);
        }
        // norms of the lane's own window px + r (all four lane groups read the same values)
        float nrm[MT][NV];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const uint16_t *np = P.norms + ((size_t)pp[mt] * Lrows + py[mt]) * Lpitch + px[mt] + r;
#pragma unroll
            for (int v = 0; v < NV; v++) nrm[mt][v] = (float)__builtin_bit_cast(_Float16, np[(size_t)v * P.norm_stride]);  // f16, a lower bound
        }
        // A fragments (as scan_mfma2.hip): lane (r, g) of K-step ks holds the 16 bytes of k-group 4*ks+g of window px+r
        v4i afrag[MT][KSTEPS];
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
                afrag[mt][ks] = a ^ (int)0x80808080;  // u8 -> i8 (a - 128): every int8 operand row sums to zero
            }
        }
        take.request();  // the next item's ticket
        PROF_STAMP(0)
        // ---- stage 1: y = basis x windows^T ----
        v4i y[MT][LR_BASIS_TILES];
#pragma unroll
        for (int b = 0; b < (int)LR_BASIS_TILES; b++) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) y[mt][b] = v4i{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) {
                const v4i uf = basis[(b * KSTEPS + ks) * 64 + lane];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) y[mt][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(uf, afrag[mt][ks], y[mt][b], 0, 0, 0);
            }
        }
        PROF_STAMP(1)
        // ---- mid: bf16 operand of stage 2 ----
        v4i zf[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            float f[8];
            float ss = 0.f;
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    f[4 * b + v] = (float)y[mt][b][v];  // |y| < 2^23: exact
                    ss = __builtin_fmaf(f[4 * b + v], f[4 * b + v], ss);  // the extras' basis rows are zero: no masking needed
                }
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            float nF = 0.f;
#pragma unroll
            for (int v = 0; v < NV; v++) nF = ((uint32_t)v == P.frame_value) ? __builtin_fabsf(nrm[mt][v]) : nF;
            // R >= sqrt(N_F^2 - |P(a - mean)|^2): N_F (widened above to an upper bound) squared and widened by 2^-20, the
            // subtrahend already narrowed by inv_lambda's margin; sqrt 1 ulp + its own product: another 2^-20
            nF *= 1.f + 0x1p-10f + 0x1p-19f;  // the stored f16 norm is a lower bound, at most 2^-10 (+ f32 roundings) below the norm
            const float nF2 = nF * nF * (1.f + 0x1p-20f);
            float R = __builtin_amdgcn_sqrtf(__builtin_fmaxf(__builtin_fmaf(-ss, P.inv_lambda, nF2), 0.f)) * (1.f + 0x1p-20f);
            // extras as f32 bit patterns already rounded in the safe direction at bit 16 (bf16 = the high half):
            //   R, N_F up; theta_c = thr_lo * norm_c towards -inf; classes that cannot emit here get an unreachable threshold
            uint32_t X[4 + 4];
            X[0] = __float_as_uint(R) + 0xffffu;
            X[1] = __float_as_uint(nF * (1.f + 0x1p-20f)) + 0xffffu;
#pragma unroll
            for (int ci = 0; ci < (int)LR_MAX_CLASSES; ci++) {
                float th = 3.0e38f;
                uint32_t add = 0;
                if (ci < NV && (uint32_t)ci < P.n_cls) {
                    const float nc = nrm[mt][ci < NV ? ci : 0];
                    if (nc > 0.f) {
                        // lower bound of thr * norm_c: the stored norm is a lower bound (fine for thr >= 0); a negative
                        // threshold needs the upper bound of the norm
                        th = P.thr_lo * (P.theta_add ? nc * (1.f + 0x1p-10f + 0x1p-19f) : nc);
                        add = P.theta_add;
                    }
                }
                X[2 + ci] = __float_as_uint(th) + add;
            }
            X[6] = X[7] = 0;
            // element 7 <- extra g (g < n_extras), element 6 <- extra 4 + g (4 + g < n_extras)
            const uint32_t x7 = g == 0 ? X[0] : g == 1 ? X[1] : g == 2 ? X[2] : X[3];
            const uint32_t x6 = g == 0 ? X[4] : g == 1 ? X[5] : 0u;
            const bool e7 = (uint32_t)g < n_extras, e6 = (uint32_t)(4 + g) < n_extras;
            const v2bf p01 = v2bf{(__bf16)f[0], (__bf16)f[1]}, p23 = v2bf{(__bf16)f[2], (__bf16)f[3]}, p45 = v2bf{(__bf16)f[4], (__bf16)f[5]};
            const v2bf p67 = v2bf{(__bf16)f[6], (__bf16)f[7]};
            const uint32_t u67 = __builtin_bit_cast(uint32_t, p67);
            const uint32_t lo6 = e6 ? (x6 >> 16) : (u67 & 0xffffu), hi7 = e7 ? (x7 & 0xffff0000u) : (u67 & 0xffff0000u);
            zf[mt] = v4i{(int)__builtin_bit_cast(uint32_t, p01), (int)__builtin_bit_cast(uint32_t, p23), (int)__builtin_bit_cast(uint32_t, p45),
                         (int)(hi7 | lo6)};
        }
        PROF_STAMP(2)
        // ---- stage 2: D2[template][window] per (M-tile, N-tile); bit nt of the M-tile's mask <=> some D2 > 0 ----
        // No compare, no scalar traffic in the loop: the sign bit of max3(d0, d1, d2) & d3 (integer view: set iff all four
        // are negative) is shifted into a per-M-tile register, 32 N-tiles per register; one wave-wide AND per M-tile and
        // 32 N-tiles then tells which blocks have a non-negative value in some lane.  (+0.0 counts as a flag: harmless.)
        uint64_t mask[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) mask[mt] = 0;
        v4i gq = gb[lane];
        auto tile_signs = [&](const v4i gop, uint32_t (&sbr)[MT]) {  // one N-tile: MT MFMAs, their sign bits shifted in
            v4f d[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
                d[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, gop), __builtin_bit_cast(v8bf, zf[mt]), v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const v4i di = __builtin_bit_cast(v4i, d[mt]);
                const int m3 = max(max(di[0], di[1]), di[2]) & di[3];  // only its sign is used: set iff all four are negative
                sbr[mt] = __builtin_amdgcn_alignbit(sbr[mt], (uint32_t)m3, 31);  // (sb << 1) | sign
            }
        };
        for (uint32_t nt0 = 0; nt0 < n_tiles16; nt0 += 32) {
            const uint32_t nblk = min(32u, n_tiles16 - nt0), last = n_tiles16 - 1;
            uint32_t sb[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) sb[mt] = 0xffffffffu;
            for (uint32_t k = 0; k < nblk; k += 2) {  // two N-tiles per trip: the operands ping-pong between two register sets
                const v4i g1 = gb[min(nt0 + k + 1, last) * 64 + lane];
                tile_signs(gq, sb);
                gq = gb[min(nt0 + k + 2, last) * 64 + lane];
                if (k + 1 < nblk) tile_signs(g1, sb);
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const uint32_t all_neg = wave_and(sb[mt]);  // bit j: tile nt0 + nblk - 1 - j is negative in every lane
                const uint32_t flagged = __builtin_bitreverse32(~all_neg) >> (32 - nblk);  // bit k: tile nt0 + k
                mask[mt] |= (uint64_t)flagged << nt0;
            }
        }
        PROF_STAMP(3)
        // ---- stage 3: exact-taps int8 check of the flagged (M-tile, N-tile) blocks ----
        // Two blocks of an M-tile per step (independent accumulator chains hide each other's LDS and MFMA latency; an odd
        // last block is paired with itself and its copy ignored).
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            uint64_t m = pv[mt] ? mask[mt] : 0;  // M-tiles past the end of the enumeration never flag
            if (m == 0) continue;
            // C-in of the lane's own window per size class of the launch: -(floor(kq * norm_c) - 2) (scan_mfma.hip:
            // conservative for |L| < 4e6), or -REJECT where the class never emits
            int cin[NV];
#pragma unroll
            for (int v = 0; v < NV; v++) {
                const float nc = nrm[mt][v];
                float Lf = __builtin_floorf(kq_of_value[v] * nc) - 2.0f;
                Lf = __builtin_fminf(__builtin_fmaxf(Lf, -1.0e9f), 1.0e9f);
                cin[v] = nc > 0.f ? -(int)Lf : -REJECT;
            }
            while (m) {
                const uint32_t nta = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                const bool two = m != 0;
                const uint32_t ntb = two ? (uint32_t)__builtin_ctzll(m) : nta;
                m &= m - 1;  // (0 & anything) stays 0
#ifdef FOCR_MFMA3_PROF
                prof_acc[5] += two ? 2 : 1;
#endif
                // the blocks' size classes: segment boundaries and their norm values are launch constants in SGPRs
                const uint32_t va = nta < seg_end0 ? seg_val0 : nta < seg_end1 ? seg_val1 : nta < seg_end2 ? seg_val2 : seg_val3;
                const uint32_t vb = ntb < seg_end0 ? seg_val0 : ntb < seg_end1 ? seg_val1 : ntb < seg_end2 ? seg_val2 : seg_val3;
                int ca = cin[0], cb = cin[0];
#pragma unroll
                for (int v = 1; v < NV; v++) {
                    ca = va == (uint32_t)v ? cin[v] : ca;
                    cb = vb == (uint32_t)v ? cin[v] : cb;
                }
                v4i acca = v4i{ca, ca, ca, ca}, accb = v4i{cb, cb, cb, cb};
                v4i ba[KSTEPS], bb[KSTEPS];
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ks++) {
                    ba[ks] = bank[(nta * KSTEPS + ks) * 64 + lane];
                    bb[ks] = bank[(ntb * KSTEPS + ks) * 64 + lane];
                }
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ks++) {
                    acca = __builtin_amdgcn_mfma_i32_16x16x64_i8(ba[ks], afrag[mt][ks], acca, 0, 0, 0);
                    accb = __builtin_amdgcn_mfma_i32_16x16x64_i8(bb[ks], afrag[mt][ks], accb, 0, 0, 0);
                }
                if (!two) accb = v4i{-1, -1, -1, -1};
                const int mm = max(max(max(acca[0], acca[1]), max(acca[2], acca[3])), max(max(accb[0], accb[1]), max(accb[2], accb[3])));
                if (__builtin_amdgcn_ballot_w64(mm > 0) == 0) continue;  // wave-uniform
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const v4i acc = h ? accb : acca;
                    const uint32_t nt = h ? ntb : nta;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const bool f = acc[i] > 0;
                        const uint64_t fm = __builtin_amdgcn_ballot_w64(f);
                        if (!fm) continue;  // wave-uniform
                        // lane (r, g), register i: template 4g + i of the tile, window px + r
                        const uint32_t tg = f ? tg_lds[nt * 16 + 4 * g + i] : 0xffffffffu;
                        const bool ok = tg != 0xffffffffu;  // dead / padding templates never emit
                        const uint64_t okmask = __builtin_amdgcn_ballot_w64(ok);
                        const uint32_t cnt = (uint32_t)__builtin_popcountll(okmask);
                        if (!cnt) continue;
                        if (wcount + cnt > WBUF) {
                            flush_wave_candidates(wbuf, wcount, lane, cand, cand_counter, cand_cap);
                            wcount = 0;
                        }
                        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(okmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)okmask, 0u));
                        if (ok) wbuf[wcount + pos] = fmt.pack(page_base + pp[mt], py[mt], px[mt] + r, tg);
                        wcount += cnt;
                    }
                }
            }
        }
        PROF_STAMP(4)
    }
    if (wcount) flush_wave_candidates(wbuf, wcount, lane, cand, cand_counter, cand_cap);
#ifdef FOCR_MFMA3_PROF
    if (lane == 0)
    {
        for (int i = 0; i < 6; i++) atomicAdd(&focr_prof[i], prof_acc[i]);
        atomicAdd(&focr_prof[6], prof_items);
    }
#endif
}

template <int KSTEPS, int RPG, int NV>
static void launch_v3(focr_ctx *c, const MfmaLaunch &L, const Mfma3Args &A3, const int8_t *basis, const uint16_t *gbank, unsigned n_cus) {
    constexpr int MT = 4, NW = 16;
    const uint32_t n_tiles16 = L.n_tiles16;
    const size_t lds = (size_t)n_tiles16 * KSTEPS * 1024 + (size_t)LR_BASIS_TILES * KSTEPS * 1024 + (size_t)n_tiles16 * 1024 + (size_t)NW * WBUF * 8 +
                       (size_t)n_tiles16 * 16 * 4;
    const uint64_t total_mt = (uint64_t)L.mtx * L.n_rows * c->sub_np;
    const uint64_t n_items = (total_mt + MT - 1) / MT;
    unsigned grid = (unsigned)std::min<uint64_t>(n_cus, (n_items + NW - 1) / NW);
    auto kern = scan_mfma3_kernel<KSTEPS, RPG, MT, NW, NV>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // MACs issued per live M-tile: stage 1 (2 tiles of int8 K = 64 K-steps) + stage 2 (bf16 K = 32 per N-tile); the exact-taps
    // stage's share depends on the data and is not counted here
    const uint64_t issued = 16 * ((uint64_t)LR_BASIS_TILES * 16 * KSTEPS * 64 + (uint64_t)n_tiles16 * 16 * 32);
    char name[64];
    snprintf(name, sizeof name, "scan_mfma3_kernel<%d,%d,%d,%d>", KSTEPS, RPG, MT, NW);
    c->launch_begin(name, L.n_templates | (L.super_index << 24), L.alg_macs, issued);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, c->stream, c->d_pages + c->sub_p0 * c->rows_alloc * c->pitch, (uint32_t)c->pitch,
                       (uint32_t)c->rows_alloc, L.live_list, L.live_count, (uint32_t)c->sub_p0, reinterpret_cast<const v4i *>(c->d_qbank + L.q_offset),
                       reinterpret_cast<const v4i *>(basis), reinterpret_cast<const v4i *>(gbank), n_tiles16, L.segs, L.Lpitch, L.Lrows, A3,
                       c->d_tglobal + L.tg_offset, c->fmt, c->d_cand, (unsigned long long *)c->d_counter + 1, (unsigned long long)c->ub_cand, L.queue);
    c->launch_end();
}

#ifdef FOCR_MFMA3_PROF
extern "C" int focr_debug_prof(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(focr_prof), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(focr_prof), z, sizeof z);
    }
    return 0;
}
#endif

int dispatch_mfma_v3(focr_ctx *c, const MfmaLaunch &L, const Mfma3Args &A3, const int8_t *basis, const uint16_t *gbank, unsigned n_cus) {
    const uint32_t nvp = A3.nv <= 1 ? 1 : (A3.nv <= 2 ? 2 : 4);
#define CASE3(K, R)                                                              \
    case (K) * 10 + (R):                                                         \
        if (nvp == 1) launch_v3<K, R, 1>(c, L, A3, basis, gbank, n_cus);         \
        else if (nvp == 2) launch_v3<K, R, 2>(c, L, A3, basis, gbank, n_cus);    \
        else launch_v3<K, R, 4>(c, L, A3, basis, gbank, n_cus);                  \
        break;
    switch (L.ksteps * 10 + L.layout) {
        CASE3(1, 1) CASE3(2, 1) CASE3(3, 1) CASE3(4, 1)
        CASE3(1, 2) CASE3(2, 2) CASE3(3, 2) CASE3(4, 2)
        CASE3(3, 3)
        default: return fail(c, FOCR_ERR_INVALID, "scan_mfma3: unsupported size class");
    }
#undef CASE3
    FOCR_HIP(c, hipGetLastError());
    return FOCR_OK;
}

}  // namespace focr
