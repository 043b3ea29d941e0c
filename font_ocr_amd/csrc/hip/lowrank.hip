// lowrank.hip — host side of the two-stage MFMA prefilter (kernel: scan_mfma3.hip).
//
// The single-stage prefilter (scan_mfma.hip) pays K = n_w * n_h taps for every (window, template) pair.  A glyph bank
// is far from full rank: the 380 templates of BASELINE configs[1] (95 glyphs x 4 sub-pixel shifts, 135 taps) keep
// 92 % of their energy in 28 principal directions.  So a first stage bounds ALL templates of a window at once:
//
//   basis  U^ : r integer (int8, zero-sum) rows spanning (about) the bank's principal subspace, frame = union box
//   stage 1:  y = U^ (a - 128)            one i8 MFMA chain per window against 2 N-tiles (32 rows) instead of T/16
//   stage 2:  for every template t        one bf16 MFMA (K = 32) per (16 windows x 16 templates):
//       D2 = sum_j y_j g_tj  +  R(w) rho_t  +  N_F(w) kappa_t  -  thr_eff norm_c(w)
//     g_t      : least-squares coefficients of the unit template t^ on the rows of U^ (stored in bf16; the residual
//                below is computed for the STORED values, so their rounding costs nothing)
//     e_t      = t^ - U^T g_t,  rho_t = |e_t orthogonal to rowspace(U^)|,  rho_par_t = |e_t inside it| (tiny)
//     R(w)     >= |(I - P)(a - mean)| = sqrt(N_F^2 - y^T (U^ U^T)^-1 y),  bounded with  y^T G^-1 y >= |y|^2 / lambda_max(G)
//     N_F(w)   : norm of the frame window;  norm_c(w): norm of the window of t's size class
//   Then  <a, t^> = sum_j y_j g_tj + <a - mean, e_t>  <=  sum_j y_j g_tj + R rho_t + N_F rho_par_t   (Cauchy-Schwarz),
//   and the reference emits only if <a, t^> / norm_c > thr.  Hence  "reference emits  =>  D2 > 0":
//   stage 2 has no false negatives; kappa_t carries rho_par_t plus explicit margins for the bf16 rounding of y
//   (2^-8 sqrt(lambda_max) |g_t|) and the f32 accumulation inside the MFMA (see build()).
//   Only the (16 windows x 16 templates) blocks with some D2 > 0 (about 7 % on text pages) go through the exact-taps int8
//   stage of scan_mfma.hip, whose candidates are verified with the reference arithmetic as before.
//
// Everything here is plain host C++ (double precision); it runs once per bank upload.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "mfma_common.h"

namespace focr {

// ---- bf16 helpers (host) ----
static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// smallest bf16 >= x (x finite)
static inline uint16_t bf16_round_up(double x) {
    uint16_t h = f32_to_bf16_rne((float)x);
    while ((double)bf16_to_f32(h) < x) {
        if (h == 0x8000u) h = 0x0000u;              // -0 -> +0
        else if (h & 0x8000u) h = (uint16_t)(h - 1);  // negative: towards zero
        else h = (uint16_t)(h + 1);
    }
    return h;
}

// f32 -> f16 rounded towards zero -> f32, for values in f16's normal range (window norms: [0.04, 2886]) and 0
static inline float f16_rtz(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u &= 0xffffe000u;  // drop the 13 mantissa bits f16 does not have
    memcpy(&f, &u, 4);
    return f;
}

// dense symmetric positive definite solve (Cholesky), n <= 32
static bool cholesky(std::vector<double> &A, int n) {
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) return false;
        d = std::sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return true;
}
static void chol_solve(const std::vector<double> &L, int n, const double *b, double *x) {
    std::vector<double> y(n);
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k];
        y[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = y[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
}

// Builds the two-stage data of one super-class.  `dense` = class-ordered dense needles (as build_mfma_bank gets them).
// Appends to `basis_bytes` (int8, per-lane MFMA operand layout, LR_BASIS_TILES x ksteps KiB) and `g_bytes`
// (bf16, per-lane layout, 1 KiB per N-tile of the super-class); fills su.lr.  Leaves su.lr.available = false when the
// super-class does not qualify (too few templates, too many classes, degenerate basis).
void build_lowrank(focr_ctx *c, SuperClass &su, const uint8_t *dense, std::vector<int8_t> &basis_bytes, std::vector<uint16_t> &g_bytes) {
    LowRank &lr = su.lr;
    lr = LowRank{};
    const uint32_t n_cls = (uint32_t)su.classes.size();
    if (n_cls == 0 || n_cls > LR_MAX_CLASSES || su.n_tiles < 4) return;
    uint32_t fw = 0, fh = 0;
    for (uint32_t k : su.classes) {
        fw = std::max(fw, c->classes[k].n_w);
        fh = std::max(fh, c->classes[k].n_h);
    }
    const uint32_t D = fw * fh;
    const int r = (int)(LR_K - 2 - n_cls);  // principal directions kept
    // unit, mean-centred templates in the frame (zero outside their own box)
    struct Tv {
        uint32_t ci, cls_i;  // class-ordered index, position of its class inside the super-class
        std::vector<double> v;
    };
    std::vector<Tv> tv;
    for (uint32_t i = 0; i < n_cls; i++) {
        const SizeClass &sc = c->classes[su.classes[i]];
        const uint32_t n = sc.n_w * sc.n_h;
        for (uint32_t q = 0; q < sc.n_templates; q++) {
            const uint32_t ci = sc.first + q;
            if (!std::isfinite(c->h_tconst[ci].rnorm_n)) continue;  // constant needle: never emits
            const uint8_t *nd = dense + c->h_needle_off[ci];
            double s = 0, s2 = 0;
            for (uint32_t p = 0; p < n; p++) s += nd[p], s2 += (double)nd[p] * nd[p];
            const double mean = s / n, n2 = s2 - s * s / n;
            if (!(n2 > 0)) continue;
            Tv t;
            t.ci = ci;
            t.cls_i = i;
            t.v.assign(D, 0.0);
            const double inv = 1.0 / std::sqrt(n2);
            for (uint32_t j = 0; j < sc.n_h; j++)
                for (uint32_t x = 0; x < sc.n_w; x++) t.v[j * fw + x] = (nd[j * sc.n_w + x] - mean) * inv;
            tv.push_back(std::move(t));
        }
    }
    if ((int)tv.size() < 2 * r || (int)D <= r) return;  // nothing to gain from a subspace this size
    // covariance and its dominant r-dimensional subspace by orthogonal iteration (any basis of the subspace will do)
    std::vector<double> C((size_t)D * D, 0.0);
    for (const Tv &t : tv)
        for (uint32_t a = 0; a < D; a++) {
            const double va = t.v[a];
            if (va == 0.0) continue;
            double *row = &C[(size_t)a * D];
            for (uint32_t b = 0; b < D; b++) row[b] += va * t.v[b];
        }
    std::vector<double> Q((size_t)D * r), Z((size_t)D * r);
    uint64_t seed = 0x9e3779b97f4a7c15ull;
    for (double &q : Q) {
        seed = seed * 6364136223846793005ull + 1442695040888963407ull;
        q = (double)(int64_t)(seed >> 11) / (double)(1ull << 52) - 1.0;
    }
    auto orth = [&](std::vector<double> &M) -> bool {  // modified Gram-Schmidt on the r columns of the D x r matrix
        for (int j = 0; j < r; j++) {
            for (int k = 0; k < j; k++) {
                double d = 0;
                for (uint32_t a = 0; a < D; a++) d += M[(size_t)a * r + j] * M[(size_t)a * r + k];
                for (uint32_t a = 0; a < D; a++) M[(size_t)a * r + j] -= d * M[(size_t)a * r + k];
            }
            double nn = 0;
            for (uint32_t a = 0; a < D; a++) nn += M[(size_t)a * r + j] * M[(size_t)a * r + j];
            if (!(nn > 1e-24)) return false;
            nn = 1.0 / std::sqrt(nn);
            for (uint32_t a = 0; a < D; a++) M[(size_t)a * r + j] *= nn;
        }
        return true;
    };
    if (!orth(Q)) return;
    for (int it = 0; it < 48; it++) {
        for (uint32_t a = 0; a < D; a++) {
            const double *row = &C[(size_t)a * D];
            for (int j = 0; j < r; j++) Z[(size_t)a * r + j] = 0;
            for (uint32_t b = 0; b < D; b++) {
                const double cab = row[b];
                if (cab == 0.0) continue;
                for (int j = 0; j < r; j++) Z[(size_t)a * r + j] += cab * Q[(size_t)b * r + j];
            }
        }
        Q.swap(Z);
        if (!orth(Q)) return;
    }
    // The columns of Q are combinations of zero-sum vectors, hence zero-sum.  Quantise with one common scale and
    // largest-remainder rounding so that every integer row sums to exactly zero (then (a - 128) and a give the same y).
    double qmax = 0;
    for (double q : Q) qmax = std::max(qmax, std::fabs(q));
    const double s = 126.0 / qmax;
    std::vector<int> U((size_t)r * D);  // integer rows
    {
        std::vector<double> rk(D);
        std::vector<uint32_t> idx(D);
        for (int j = 0; j < r; j++) {
            long sum = 0;
            for (uint32_t a = 0; a < D; a++) {
                rk[a] = s * Q[(size_t)a * r + j];
                U[(size_t)j * D + a] = (int)std::floor(rk[a]);
                sum += U[(size_t)j * D + a];
                idx[a] = a;
            }
            const long deficit = -sum;
            if (deficit < 0 || deficit > (long)D) return;
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return rk[a] - U[(size_t)j * D + a] > rk[b] - U[(size_t)j * D + b]; });
            for (long d = 0; d < deficit; d++) U[(size_t)j * D + idx[d]] += 1;
            long check = 0;
            for (uint32_t a = 0; a < D; a++) {
                const int u = U[(size_t)j * D + a];
                if (u > 127 || u < -127) return;
                check += u;
            }
            if (check != 0) return;
        }
    }
    // Gram matrix G = U U^T, its Cholesky factor and an upper bound of lambda_max (power iteration + 0.5 %, capped by
    // the infinity norm, which is always an upper bound)
    std::vector<double> G((size_t)r * r);
    for (int i = 0; i < r; i++)
        for (int j = 0; j < r; j++) {
            double d = 0;
            for (uint32_t a = 0; a < D; a++) d += (double)U[(size_t)i * D + a] * U[(size_t)j * D + a];
            G[(size_t)i * r + j] = d;
        }
    double linf = 0;
    for (int i = 0; i < r; i++) {
        double rs = 0;
        for (int j = 0; j < r; j++) rs += std::fabs(G[(size_t)i * r + j]);
        linf = std::max(linf, rs);
    }
    std::vector<double> pv(r, 1.0), pw(r);
    double lam = 0;
    for (int it = 0; it < 300; it++) {
        double nn = 0;
        for (int i = 0; i < r; i++) {
            double d = 0;
            for (int j = 0; j < r; j++) d += G[(size_t)i * r + j] * pv[j];
            pw[i] = d;
            nn += d * d;
        }
        nn = std::sqrt(nn);
        lam = nn;
        for (int i = 0; i < r; i++) pv[i] = pw[i] / nn;
    }
    // power iteration converges from below; the residual norm bounds the distance to an eigenvalue
    {
        double res = 0;
        for (int i = 0; i < r; i++) {
            double d = 0;
            for (int j = 0; j < r; j++) d += G[(size_t)i * r + j] * pv[j];
            res += (d - lam * pv[i]) * (d - lam * pv[i]);
        }
        lam = std::min(linf, (lam + std::sqrt(res)) * 1.005);
    }
    std::vector<double> Lc = G;
    if (!cholesky(Lc, r)) return;

    // per template: coefficients, residual split, margins; scatter into the per-lane stage-2 operand layout
    uint32_t slot_of[LR_K];
    for (int j = 0; j < r; j++) slot_of[j] = lr_comp_slot((uint32_t)j, 2 + n_cls);
    const size_t g_base = g_bytes.size();
    g_bytes.resize(g_base + (size_t)su.n_tiles * 512, 0);  // 64 lanes x 8 bf16 per N-tile
    // Padding slots and dead templates (constant needles) must never flag a block: an all-zero row would give D2 = +0.0,
    // which the kernel's sign test counts as a flag.  -1 in the N_F slot makes their D2 = -N_F(w) < 0 on every window
    // with any variance; live templates overwrite the whole row below.
    {
        const uint32_t slot = lr_extra_slot(1), b = slot / 16, gq = (slot % 16) / 4, v = slot % 4;
        for (uint32_t nt = 0; nt < su.n_tiles; nt++)
            for (uint32_t nn = 0; nn < 16; nn++) g_bytes[g_base + ((size_t)nt * 64 + gq * 16 + nn) * 8 + 4 * b + v] = 0xbf80u;
    }
    std::vector<double> rhs(r), g(r), e(D), proj(r), coef(r);
    double rho_sum = 0, rho_max = 0;
    for (const Tv &t : tv) {
        for (int j = 0; j < r; j++) {
            double d = 0;
            for (uint32_t a = 0; a < D; a++) d += (double)U[(size_t)j * D + a] * t.v[a];
            rhs[j] = d;
        }
        chol_solve(Lc, r, rhs.data(), g.data());
        uint16_t gb[LR_K] = {0};
        double gnorm2 = 0;
        for (int j = 0; j < r; j++) {
            gb[j] = f32_to_bf16_rne((float)g[j]);
            g[j] = (double)bf16_to_f32(gb[j]);  // the stored value is what stage 2 multiplies with
        }
        // e = t^ - U^T g (for the stored g), split into its part inside / orthogonal to rowspace(U)
        for (uint32_t a = 0; a < D; a++) {
            double d = t.v[a];
            for (int j = 0; j < r; j++) d -= (double)U[(size_t)j * D + a] * g[j];
            e[a] = d;
        }
        for (int j = 0; j < r; j++) {
            double d = 0;
            for (uint32_t a = 0; a < D; a++) d += (double)U[(size_t)j * D + a] * e[a];
            proj[j] = d;
        }
        chol_solve(Lc, r, proj.data(), coef.data());
        double e2 = 0, par2 = 0;
        for (uint32_t a = 0; a < D; a++) {
            double p = 0;
            for (int j = 0; j < r; j++) p += (double)U[(size_t)j * D + a] * coef[j];
            par2 += p * p;
            e2 += (e[a] - p) * (e[a] - p);
        }
        // bf16 keeps 8 significant bits: round-to-nearest is off by at most 2^-8 relative.  |y| <= sqrt(lambda_max) N_F,
        // so |sum_j (bf16(y_j) - y_j) g_j| <= 2^-8 sqrt(lambda_max) |g| N_F
        for (int j = 0; j < r; j++) gnorm2 += g[j] * g[j];
        const double rho = std::sqrt(e2) * (1.0 + 1e-9) + 1e-12;
        const double eps_y = std::ldexp(1.0, -8) * std::sqrt(lam) * std::sqrt(gnorm2);
        // + 3e-4: f32 accumulation inside the MFMA (<= 34 terms, sum of magnitudes <= ~4.5 N_F: < 4e-5 N_F even at one
        // truncation per add) and the f32 roundings of R, N_F, norm_c on the device (each < 1e-6 relative)
        const double kappa = (std::sqrt(par2) + eps_y) * (1.0 + 1e-6) + 3e-4;
        // K slots (mfma_common.h): extras R, N_F, one per class; the principal directions fill the other slots
        uint16_t slot_val[LR_K] = {0};
        for (int j = 0; j < r; j++) slot_val[slot_of[j]] = gb[j];
        slot_val[lr_extra_slot(0)] = bf16_round_up(rho);
        slot_val[lr_extra_slot(1)] = bf16_round_up(kappa);
        for (uint32_t i = 0; i < n_cls; i++) slot_val[lr_extra_slot(2 + i)] = (i == t.cls_i) ? 0xbf80u /* -1.0 */ : 0u;
        rho_sum += rho;
        rho_max = std::max(rho_max, rho);
        // position of the template inside the super-class's N-tiles
        const SizeClass &sc = c->classes[su.classes[t.cls_i]];
        const uint32_t q = t.ci - sc.first, nt = su.tile_first[t.cls_i] + q / 16, nn = q % 16;
        for (uint32_t slot = 0; slot < LR_K; slot++) {
            const uint32_t b = slot / 16, gq = (slot % 16) / 4, v = slot % 4;  // slot = 16 b + 4 g + v  ->  lane group g, element 4 b + v
            g_bytes[g_base + ((size_t)nt * 64 + gq * 16 + nn) * 8 + 4 * b + v] = slot_val[slot];
        }
    }
    // basis in the per-lane MFMA operand layout of the super-class's K layout: [tile b][k-step][lane = g*16 + n][16 B]
    const size_t b_base = basis_bytes.size();
    basis_bytes.resize(b_base + (size_t)LR_BASIS_TILES * su.ksteps * 1024, 0);
    for (int j = 0; j < r; j++) {
        const uint32_t b = slot_of[j] / 16, n = slot_of[j] % 16;  // stage 1 leaves y_j in the lane and register of its slot
        for (uint32_t row = 0; row < fh; row++)
            for (uint32_t x = 0; x < fw; x++) {
                uint32_t ks, gq, byte;
                kgroup_of(su.layout, row, x, &ks, &gq, &byte);
                basis_bytes[b_base + ((size_t)(b * su.ksteps + ks) * 64 + gq * 16 + n) * 16 + byte] = (int8_t)U[(size_t)j * D + row * fw + x];
            }
    }
    lr.available = true;
    lr.r = (uint32_t)r;
    lr.n_cls = n_cls;
    lr.frame_w = fw;
    lr.frame_h = fh;
    lr.frame_class = -1;
    for (uint32_t i = 0; i < n_cls; i++)
        if (c->classes[su.classes[i]].n_w == fw && c->classes[su.classes[i]].n_h == fh) lr.frame_class = (int)i;
    lr.basis_offset = b_base;
    lr.g_offset = g_base * 2;
    // sum_j y_j^2 * inv_lambda <= y^T G^-1 y = |P (a - mean)|^2   (1e-5: f32 rounding of the sum of squares on the device)
    lr.inv_lambda = (float)((1.0 - 1e-5) / lam);
    if ((double)lr.inv_lambda > (1.0 - 1e-5) / lam) lr.inv_lambda = std::nextafterf(lr.inv_lambda, 0.f);
    lr.mean_rho = rho_sum / (double)tv.size();
    lr.max_rho = rho_max;
    lr.n_live = (uint32_t)tv.size();
}

}  // namespace focr

// ---------------------------------------------------------------------------------------------
// Host model of the two-stage bound, for CPU tests of the mathematics (tests/test_lowrank_host.py): the same data
// build_lowrank() hands to the device and the same arithmetic scan_mfma3.hip's mid stage performs (f32 operations, bf16
// operands, f32 accumulation), evaluated for caller-supplied frame-sized windows of super-class 0.
using namespace focr;

extern "C" int focr_debug_lowrank(const focr_template_t *templates, size_t n_templates, const uint8_t *needles, size_t needles_len,
                                  const uint8_t *windows, size_t n_windows, float threshold, double *info, double *sim, float *d2) {
    if (!templates || !n_templates || !needles || !info) return FOCR_ERR_INVALID;
    for (size_t t = 0; t < n_templates; t++)
        if (templates[t].n_w == 0 || templates[t].n_h == 0 || templates[t].n_w > 16 || templates[t].n_h > 32 ||
            (size_t)templates[t].offset + (size_t)templates[t].n_w * templates[t].n_h > needles_len)
            return FOCR_ERR_INVALID;
    focr_ctx ctx;  // host state only: no device call below
    focr_ctx *c = &ctx;
    std::vector<uint32_t> direct;
    std::vector<uint8_t> dense;
    bank_host_prepare(c, templates, n_templates, needles, direct, dense);
    layout_supers(c);
    for (int i = 0; i < 8; i++) info[i] = 0;
    if (c->supers.empty()) return FOCR_OK;
    SuperClass &su = c->supers[0];
    std::vector<int8_t> basis;
    std::vector<uint16_t> gb;
    build_lowrank(c, su, dense.data(), basis, gb);
    const LowRank &lr = su.lr;
    info[0] = lr.available;
    info[1] = lr.r;
    info[2] = lr.n_cls;
    info[3] = lr.frame_w;
    info[4] = lr.frame_h;
    info[5] = lr.mean_rho;
    info[6] = lr.max_rho;
    info[7] = lr.inv_lambda;
    if (!lr.available || !windows || !n_windows || !sim || !d2) return FOCR_OK;
    const uint32_t fw = lr.frame_w, fh = lr.frame_h, D = fw * fh, n_extras = 2 + lr.n_cls;
    // integer basis rows back out of the per-lane layout: row of slot s = tile s/16, n = s%16
    std::vector<int> U((size_t)LR_K * D, 0);
    for (uint32_t s = 0; s < LR_K; s++)
        for (uint32_t row = 0; row < fh; row++)
            for (uint32_t x = 0; x < fw; x++) {
                uint32_t ks, gq, byte;
                kgroup_of(su.layout, row, x, &ks, &gq, &byte);
                U[(size_t)s * D + row * fw + x] = basis[lr.basis_offset + ((size_t)((s / 16) * su.ksteps + ks) * 64 + gq * 16 + s % 16) * 16 + byte];
            }
    const double thr_d = (double)threshold;
    double thr_eff = thr_d - 1e-4 * (1.0 + std::fabs(thr_d));
    thr_eff = std::max(thr_eff, -2.0);
    const double tl = thr_eff >= 0 ? thr_eff * (1.0 - std::ldexp(1.0, -20)) : thr_eff * (1.0 + std::ldexp(1.0, -20));
    float thr_lo = (float)tl;
    if ((double)thr_lo > tl) thr_lo = std::nextafterf(thr_lo, -INFINITY);
    const uint32_t theta_add = thr_lo < 0.f ? 0xffffu : 0u;
    auto up16 = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0xffffu) >> 16); };
    for (size_t wi = 0; wi < n_windows; wi++) {
        const uint8_t *a = windows + wi * D;
        // window norms of every class box and of the frame, as the statistics kernel computes them (f32)
        float nrm[LR_MAX_CLASSES + 1];
        double norm_exact[LR_MAX_CLASSES + 1];
        for (uint32_t v = 0; v <= lr.n_cls; v++) {
            const uint32_t bw = v < lr.n_cls ? c->classes[su.classes[v]].n_w : fw, bh = v < lr.n_cls ? c->classes[su.classes[v]].n_h : fh;
            uint64_t s1 = 0, s2 = 0;
            for (uint32_t j = 0; j < bh; j++)
                for (uint32_t x = 0; x < bw; x++) s1 += a[j * fw + x], s2 += (uint64_t)a[j * fw + x] * a[j * fw + x];
            const uint64_t V = (uint64_t)bw * bh * s2 - s1 * s1;
            nrm[v] = f16_rtz(sqrtf((float)V * (1.0f / (float)(bw * bh))));  // the statistics kernel stores f16, rounded towards zero
            norm_exact[v] = std::sqrt((double)V / (double)(bw * bh));
        }
        const float nF = nrm[lr.n_cls] * (1.f + 0x1p-10f + 0x1p-19f);  // upper bound from the stored lower bound
        // stage 1 (exact integers), bf16 of y, sum of squares in f32
        uint16_t zslot[LR_K];
        float ss = 0.f;
        for (uint32_t s = 0; s < LR_K; s++) {
            long y = 0;
            for (uint32_t k = 0; k < D; k++) y += (long)U[(size_t)s * D + k] * ((int)a[k] - 128);
            const float yf = (float)y;
            ss = fmaf(yf, yf, ss);
            zslot[s] = f32_to_bf16_rne(yf);
        }
        const float nF2 = nF * nF * (1.f + 0x1p-20f);
        const float R = sqrtf(fmaxf(fmaf(-ss, lr.inv_lambda, nF2), 0.f)) * (1.f + 0x1p-20f);
        zslot[lr_extra_slot(0)] = up16(R);
        zslot[lr_extra_slot(1)] = up16(nF * (1.f + 0x1p-20f));
        for (uint32_t ci = 0; ci < lr.n_cls; ci++) {
            float th = 3.0e38f;
            uint32_t add = 0;
            if (nrm[ci] > 0.f) th = thr_lo * (theta_add ? nrm[ci] * (1.f + 0x1p-10f + 0x1p-19f) : nrm[ci]), add = theta_add;
            uint32_t u;
            memcpy(&u, &th, 4);
            zslot[lr_extra_slot(2 + ci)] = (uint16_t)((u + add) >> 16);
        }
        (void)n_extras;
        for (size_t t = 0; t < n_templates; t++) {
            sim[wi * n_templates + t] = NAN;
            d2[wi * n_templates + t] = 0.f;
        }
        for (uint32_t i = 0; i < lr.n_cls; i++) {
            const SizeClass &sc = c->classes[su.classes[i]];
            for (uint32_t q = 0; q < sc.n_templates; q++) {
                const TemplateConst &tc = c->h_tconst[sc.first + q];
                const uint32_t nt = su.tile_first[i] + q / 16, nn = q % 16;
                float acc = 0.f;
                for (uint32_t s = 0; s < LR_K; s++) {
                    const uint32_t b = s / 16, gq = (s % 16) / 4, v = s % 4;
                    const float gv = bf16_to_f32(gb[lr.g_offset / 2 + ((size_t)nt * 64 + gq * 16 + nn) * 8 + 4 * b + v]);
                    acc = fmaf(gv, bf16_to_f32(zslot[s]), acc);
                }
                d2[wi * n_templates + tc.index] = acc;
                if (std::isfinite(tc.rnorm_n) && norm_exact[i] > 0) {
                    const uint8_t *nd = dense.data() + c->h_needle_off[sc.first + q];
                    double num = 0;
                    for (uint32_t j = 0; j < sc.n_h; j++)
                        for (uint32_t x = 0; x < sc.n_w; x++) num += (double)a[j * fw + x] * nd[j * sc.n_w + x];
                    double s1 = 0;
                    for (uint32_t j = 0; j < sc.n_h; j++)
                        for (uint32_t x = 0; x < sc.n_w; x++) s1 += a[j * fw + x];
                    num -= tc.s_n * s1 * tc.n_recip;
                    sim[wi * n_templates + tc.index] = num * tc.rnorm_n / norm_exact[i];
                }
            }
        }
    }
    return FOCR_OK;
}
