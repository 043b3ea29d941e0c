"""Host model of the two-stage MFMA prefilter's bound (font_ocr_amd/csrc/hip/lowrank.hip, focr_debug_lowrank): the
low-rank data the device uses and the arithmetic its mid stage performs, evaluated on the CPU.  The property that
matters: "the reference emits (sim > thr)  =>  stage 2 flags the pair (D2 > 0)" — no false negatives — on text
windows, noise windows and degenerate windows, for glyph banks and for banks that do not compress at all.  No GPU."""
import ctypes as C

import numpy as np
import pytest

from font_ocr_amd import _native as N
from font_ocr_amd import synth_page
from font_ocr_amd.bank import SYNTH_SEED_BASE


def _lowrank(bank, windows, thr):
    lib = N.hip()
    T = len(bank)
    info = np.zeros(8, np.float64)
    nw = 0 if windows is None else len(windows)
    sim = np.zeros((max(nw, 1), T), np.float64)
    d2 = np.zeros((max(nw, 1), T), np.float32)
    w = None if windows is None else np.ascontiguousarray(windows, np.uint8)
    rc = lib.focr_debug_lowrank(bank.templates.ctypes.data, T, bank.needles.ctypes.data, bank.needles.size,
                                None if w is None else w.ctypes.data, nw, float(thr), info.ctypes.data, sim.ctypes.data, d2.ctypes.data)
    assert rc == 0
    return info, sim[:nw], d2[:nw]


def _patches(page_luma, fw, fh, rng, n):
    inv = 255 - page_luma
    ys = rng.integers(1, inv.shape[0] - fh, n)
    xs = rng.integers(1, inv.shape[1] - fw, n)
    return np.stack([inv[y:y + fh, x:x + fw].reshape(-1) for y, x in zip(ys, xs)])


def test_glyph_bank_compresses_and_bound_holds(bank_x2):
    info, _, _ = _lowrank(bank_x2, None, 0.8)
    available, r, n_cls, fw, fh, mean_rho, max_rho, inv_lambda = info
    assert available == 1 and (r, n_cls, fw, fh) == (28, 2, 9, 15)
    assert mean_rho < 0.35 and max_rho < 0.65, (mean_rho, max_rho)
    rng = np.random.default_rng(3)
    page = synth_page(bank_x2, SYNTH_SEED_BASE + 11, 608, 720)
    text = _patches(page, 9, 15, rng, 6000)
    # windows sitting exactly on glyph origins (similarity ~1 with the stamped template) are what must never be lost
    noise = rng.integers(0, 256, (500, 135), dtype=np.uint8)
    flat = np.full((3, 135), 77, np.uint8)                   # zero variance: never emits
    spike = np.zeros((4, 135), np.uint8)
    spike[np.arange(4), [0, 8, 126, 134]] = 255              # a single pixel: tiny norms, extreme ratios
    col8 = np.zeros((2, 9, 15), np.uint8).reshape(2, 15, 9)
    col8[:, :, 8] = 200                                      # ink only in the frame's last column: the 8-wide box is flat
    wins = np.concatenate([text, noise, flat, spike, col8.reshape(2, -1)])
    for thr in (0.8, 0.3, 0.99, -0.5, 0.0):
        _, sim, d2 = _lowrank(bank_x2, wins, thr)
        emits = sim > thr  # NaN (never emits) compares false
        assert emits.sum() > (50 if thr >= 0.8 else 100) or thr > 0.9
        missed = emits & ~(d2 > 0)
        assert not missed.any(), (thr, int(missed.sum()), sim[missed][:5], d2[missed][:5])
        if thr == 0.8:
            pairs = np.isfinite(sim).sum()
            assert (d2 > 0).sum() < 0.004 * pairs, ((d2 > 0).sum(), pairs)      # the filter filters: < 0.4 % of pairs pass
            assert (d2 > 0).sum() < 25 * max(int(emits.sum()), 1)
    # the margin is real but small: pairs just below the threshold are mostly ruled out
    _, sim, d2 = _lowrank(bank_x2, text, 0.8)
    far = sim < 0.3
    assert (d2[far] > 0).mean() < 0.002


def test_aligned_glyph_windows_are_flagged(bank_x2):
    """Every stamped glyph origin of a synthetic page: the window there matches its template with sim ~ 1."""
    page, truth = synth_page(bank_x2, SYNTH_SEED_BASE + 12, 608, 720, with_truth=True)
    inv = 255 - page
    wins, tidx = [], []
    for h in truth[:800]:
        x, y = int(h["x"]), int(h["y"])
        if x + 9 <= 608 and y + 15 <= 720 and x >= 1 and y >= 1:
            wins.append(inv[y:y + 15, x:x + 9].reshape(-1))
            tidx.append(int(h["template_index"]))
    wins = np.stack(wins)
    _, sim, d2 = _lowrank(bank_x2, wins, 0.8)
    own = sim[np.arange(len(wins)), tidx]
    assert np.nanmin(own) > 0.8 and np.nanmedian(own) > 0.99  # neighbouring glyphs overlap the 9-px frame a little
    assert (d2[np.arange(len(wins)), tidx] > 0).all()
    assert not ((sim > 0.8) & ~(d2 > 0)).any()


def test_incompressible_bank_is_still_conservative():
    """Random-noise templates have no low-rank structure (rho ~ 0.85): the bound must hold anyway (AUTO would not
    pick the two-stage path for such a bank; TWO_STAGE may be forced)."""
    from font_ocr_amd.bank import TEMPLATE_DTYPE, Bank

    rng = np.random.default_rng(9)
    tm, needles, off = [], [], 0
    for k in range(96):
        w, h = (9, 15) if k % 3 else (8, 15)
        nd = rng.integers(0, 256, (h, w), dtype=np.uint8)
        t = np.zeros(1, TEMPLATE_DTYPE)
        t["letter"], t["n_w"], t["n_h"], t["offset"] = 65 + k % 26, w, h, off
        tm.append(t)
        needles.append(nd.reshape(-1))
        off += nd.size
    bank = Bank(np.concatenate(tm), np.concatenate(needles), len(tm), 0, 0, 13.0, 8.0)
    info, _, _ = _lowrank(bank, None, 0.5)
    assert info[0] == 1 and info[5] > 0.6  # available, but it does not compress
    # windows = noisy copies of templates, so that real matches exist
    wins = []
    for k in range(0, 96, 2):
        nd = bank.needle(k).astype(np.int32)
        fr = rng.integers(0, 256, (15, 9)).astype(np.int32)
        fr[:, : nd.shape[1]] = np.clip(nd + rng.integers(-30, 30, nd.shape), 0, 255)
        wins.append(fr.reshape(-1).astype(np.uint8))
    wins = np.stack(wins + list(rng.integers(0, 256, (200, 135), dtype=np.uint8)))
    for thr in (0.5, 0.9, 0.1):
        _, sim, d2 = _lowrank(bank, wins, thr)
        emits = sim > thr
        assert emits.sum() >= 20 or thr > 0.8
        assert not (emits & ~(d2 > 0)).any(), thr
