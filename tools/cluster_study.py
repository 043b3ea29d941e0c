"""numpy study (CPU, no device): would a glyph-cluster ENVELOPE prefilter pay?

The four (C2) / sixteen (C3) sub-pixel shifts of a glyph are near-duplicates.  Stage one would scan one centroid per glyph
(95 -> 6 N-tiles instead of 24 / 95) with the same kernel, and the verify would expand a surviving (window, cluster) pair to
the cluster's members.  The bound (same Cauchy-Schwarz form as the quantisation error, mfma_common.h):

    g(w, t) = g(w, c) + sum_keep (a - m) (beta'_t - gamma_c)  <=  g(w, c) + norm_keep(w) * |beta'_t - gamma_c|

and the sharper angle form with beta'_t = alpha_t gamma_c + rho_t, rho_t orthogonal to gamma_c:

    g(w, t) <= alpha_t g(w, c) + |rho_t| sqrt(norm_keep^2 - g(w, c)^2 / |gamma_c|^2)

A cluster is a candidate when any member could still exceed its class's threshold l_k(w) = L_k(w) / c_k.  This script counts,
on whole synthetic pages, today's candidates (per template) and the cluster candidates under both bounds, with per-member
and with bank-wide worst-case parameters (what a single threshold plane could carry), and prints the verify load
(cluster candidates x members) relative to today's.  Gate: <= 2 x today's candidates -> build.

    python tools/cluster_study.py [--config c2|c3] [--pages 2] [--thr 0.8]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from font_ocr_amd import Bank, synth_page  # noqa: E402
from font_ocr_amd.bank import ASCII95, SYNTH_SEED_BASE  # noqa: E402


def unit_templates(bank, keep_w, keep_h):
    """beta' of every template on the common kept box (keep_w x keep_h, column drop + row drop folded in the same way the
    product folds the dropped column: beta' = beta + sigma / n_k on the kept taps), rho = norm of the dropped part."""
    T = len(bank)
    bp = np.zeros((T, keep_h * keep_w))
    rho = np.zeros(T)
    live = np.zeros(T, bool)
    for t in range(T):
        nd = bank.needle(t).astype(np.float64)
        n = nd.size
        n2 = (nd * nd).sum() - nd.sum() ** 2 / n
        if not n2 > 0:
            continue
        live[t] = True
        b = (nd - nd.mean()) / np.sqrt(n2)
        kept = b[:keep_h, :keep_w]
        drop_mask = np.ones_like(b, bool)
        drop_mask[:keep_h, :keep_w] = False
        sigma = b[drop_mask].sum()
        rho[t] = np.sqrt((b[drop_mask] ** 2).sum())
        bp[t] = (kept + sigma / kept.size).reshape(-1)
    return bp, rho, live


def window_sums(img, w, h):
    """exact sliding sums of a (H, W) int64 image over w x h boxes -> (H - h + 1, W - w + 1)"""
    c = np.pad(img, ((1, 0), (1, 0))).cumsum(0).cumsum(1)
    return c[h:, w:] - c[:-h, w:] - c[h:, :-w] + c[:-h, :-w]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--pages", type=int, default=2)
    ap.add_argument("--thr", type=float, default=0.8)
    ap.add_argument("--split", type=int, default=1, help="clusters per glyph (consecutive shifts share a cluster)")
    a = ap.parse_args()
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    if a.config == "c2":
        bank = Bank.load(os.path.join(root, "tests", "golden", "bank_dejavu13_ascii95_x2.bin"))
        r_w, r_h = 608, 720
    else:
        bank = Bank.rasterize("/usr/share/fonts/truetype/dejavu/DejaVuSansMono.ttf", 13.0, 2, 2, alphabet=ASCII95)
        r_w, r_h = 1200, 1600 if a.pages > 0 else 0
        r_w, r_h = 608, 720  # the study needs the bank, not the page size: keep the matrices small
    T = len(bank)
    nA = bank.n_alphabet
    sizes = sorted({(int(t["n_w"]), int(t["n_h"])) for t in bank.templates})
    kw, kh = min(s[0] for s in sizes), min(s[1] for s in sizes)
    print(f"bank: {T} templates, {nA} glyphs x {T // nA} shifts ({a.split} cluster(s) per glyph), size classes {sizes}; common kept box {kw}x{kh}")
    bp, rho, live = unit_templates(bank, kw, kh)
    n_k = kw * kh
    cls = np.array([sizes.index((int(t["n_w"]), int(t["n_h"]))) for t in bank.templates])
    c_scale = 126.0 / np.abs(bp).max()
    e_max = np.sqrt(n_k / 12.0) * 1.3  # what largest-remainder rounding of n_k taps leaves, roughly (product: measured per class)
    kappa = (c_scale * a.thr - e_max) / c_scale  # in unit-template units
    rho_max = np.array([rho[(cls == k) & live].max() if ((cls == k) & live).any() else 0 for k in range(len(sizes))])
    print(f"c = {c_scale:.1f}, kappa/c = {kappa:.4f}, rho_max per class = {np.round(rho_max, 3)}")

    # clusters: one per glyph (all its shifts)
    members = T // nA // a.split
    glyph = np.arange(T) % nA + nA * (np.arange(T) // nA // members)  # cluster id
    nA = nA * a.split
    gam = np.zeros((nA, n_k))
    delta = np.zeros(T)
    alpha = np.zeros(T)
    rnorm = np.zeros(T)
    for g in range(nA):
        m = np.where((glyph == g) & live)[0]
        if not len(m):
            continue
        gam[g] = bp[m].mean(0)
        gg = (gam[g] ** 2).sum()
        for t in m:
            delta[t] = np.linalg.norm(bp[t] - gam[g])
            alpha[t] = bp[t] @ gam[g] / gg
            rnorm[t] = np.linalg.norm(bp[t] - alpha[t] * gam[g])
    lv = live
    print(f"member distance to its centroid |beta' - gamma|: mean {delta[lv].mean():.3f} max {delta[lv].max():.3f}; "
          f"|rho| mean {rnorm[lv].mean():.3f} max {rnorm[lv].max():.3f}; alpha {alpha[lv].min():.3f}..{alpha[lv].max():.3f}; "
          f"|beta'| {np.linalg.norm(bp[lv], axis=1).min():.3f}..{np.linalg.norm(bp[lv], axis=1).max():.3f}")
    gnorm = np.linalg.norm(gam, axis=1)

    tot = dict(now=0, simple_member=0, simple_max=0, angle_member=0, angle_max=0, hits=0, windows=0)
    for p in range(a.pages):
        page = 255 - synth_page(bank, SYNTH_SEED_BASE + p, r_w, r_h).astype(np.int64)  # ink-high
        H, W = r_h - kh + 1, r_w - kw + 1
        # kept-box windows as rows of a matrix (float32: exact for u8 x small sums; the study needs counts, not bits)
        win = np.lib.stride_tricks.sliding_window_view(page.astype(np.float32), (kh, kw)).reshape(H * W, n_k)
        s_k = window_sums(page, kw, kh).reshape(-1).astype(np.float64)
        s2_k = window_sums(page * page, kw, kh).reshape(-1).astype(np.float64)
        Nk = np.sqrt(np.maximum(s2_k - s_k * s_k / n_k, 0))  # norm of the kept box about its own mean
        # per class: full-box norm and dnorm on the kept box's grid (windows whose full box leaves the page: never emitted)
        ell = np.full((len(sizes), H * W), np.inf)
        for k, (w_, h_) in enumerate(sizes):
            n = w_ * h_
            Hf, Wf = r_h - h_ + 1, r_w - w_ + 1
            s = window_sums(page, w_, h_).astype(np.float64)
            s2 = window_sums(page * page, w_, h_).astype(np.float64)
            V = s2 - s * s / n
            norm_p = np.sqrt(np.maximum(V, 0))
            # dropped part: all taps outside the kept box; dnorm^2 = sum_drop (a - m)^2, m = kept mean
            sk = window_sums(page, kw, kh)[:Hf, :Wf].astype(np.float64)
            s2k = window_sums(page * page, kw, kh)[:Hf, :Wf].astype(np.float64)
            q1, q2, D = s - sk, s2 - s2k, n - n_k
            m = sk / n_k
            dn2 = np.maximum(q2 - 2 * m * q1 + D * m * m, 0)
            L = kappa * norm_p - rho_max[k] * np.sqrt(dn2)
            L[V <= 0] = np.inf
            L[0, :] = np.inf
            L[:, 0] = np.inf  # x = 0, y = 0 are never searched
            full = np.full((H, W), np.inf)
            full[:Hf, :Wf] = L
            ell[k] = full.reshape(-1)
        G = win @ bp.T.astype(np.float32)  # (windows, T)
        Gc = win @ gam.T.astype(np.float32)  # (windows, glyphs)
        ell_t = ell[cls].T  # (windows, T) view-ish
        cand = (G > ell_t) & lv[None, :]
        tot["now"] += int(cand.sum())
        tot["windows"] += int(np.isfinite(ell).any(0).sum())
        # exact hits would need the full-box sims; the candidate count is what the study compares
        ell_min = ell.min(0)
        for name, member in (("simple_member", True), ("simple_max", False)):
            clus = np.zeros((H * W, nA), bool)
            if member:
                for t in np.where(lv)[0]:
                    clus[:, glyph[t]] |= Gc[:, glyph[t]] > ell[cls[t]] - Nk * delta[t]
            else:
                thr_w = ell_min - Nk * delta[lv].max()
                clus = Gc > thr_w[:, None]
                clus[:, gnorm == 0] = False
            tot[name] += int(clus.sum())
        for name, member in (("angle_member", True), ("angle_max", False)):
            clus = np.zeros((H * W, nA), bool)
            if member:
                for t in np.where(lv)[0]:
                    g = glyph[t]
                    gc = Gc[:, g].astype(np.float64)
                    resid = np.sqrt(np.maximum(Nk * Nk - gc * gc / gnorm[g] ** 2, 0))
                    clus[:, g] |= alpha[t] * gc + rnorm[t] * resid > ell[cls[t]]
            else:
                # one plane: worst-case member parameters over the whole bank (alpha_min where l > 0 ... keep it simple: scan both ends)
                a_lo, a_hi, r_mx = alpha[lv].min(), alpha[lv].max(), rnorm[lv].max()
                for g in range(nA):
                    if gnorm[g] == 0:
                        continue
                    gc = Gc[:, g].astype(np.float64)
                    resid = np.sqrt(np.maximum(Nk * Nk - gc * gc / gnorm[g] ** 2, 0))
                    clus[:, g] = np.maximum(a_lo * gc, a_hi * gc) + r_mx * resid > ell_min
            tot[name] += int(clus.sum())
        print(f"page {p}: candidates now {int(cand.sum())}", flush=True)
    print(f"\nthreshold {a.thr}, {a.pages} page(s) {r_w}x{r_h}: candidates today {tot['now']} ({tot['now'] / a.pages / 1e3:.1f} k per page)")
    for k in ("simple_member", "simple_max", "angle_member", "angle_max"):
        print(f"  {k:14s}: cluster candidates {tot[k]:10d} = {tot[k] / tot['now']:.2f} x today's; verify load x {members} members = "
              f"{tot[k] * members / tot['now']:.2f} x today's")


if __name__ == "__main__":
    main()
