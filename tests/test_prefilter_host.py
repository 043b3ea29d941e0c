"""Host model of the MFMA prefilter's bound (font_ocr_amd/csrc/hip/prefilter_model.hip, focr_debug_prefilter): the quantised
bank focr_bank_upload builds and the threshold arithmetic the statistics / scan kernels perform (shared inline functions of
mfma_common.h), evaluated on the CPU.  The property the fast path rests on: "the reference emits (sim > thr)  =>  the
prefilter flags the pair (G + C-in > 0)" — no false negatives — on text, noise, degenerate and adversarial windows, for
positive AND negative thresholds (round 2's f16 window norms raised the threshold for negative ones: VERDICT r02 weak #1),
with and without the column drop.  No GPU."""
import ctypes as C

import numpy as np
import pytest

from font_ocr_amd import _native as N
from font_ocr_amd import synth_page
from font_ocr_amd.bank import SYNTH_SEED_BASE, TEMPLATE_DTYPE, Bank


def _model(bank, windows, fw, fh, thr, drop=True):
    lib = N.hip()
    T = len(bank)
    w = np.ascontiguousarray(windows, np.uint8).reshape(-1, fw * fh)
    nw = len(w)
    sim = np.zeros((max(nw, 1), T), np.float64)
    d = np.zeros((max(nw, 1), T), np.int64)
    info = np.zeros(64, np.float64)
    rc = lib.focr_debug_prefilter(bank.templates.ctypes.data, T, bank.needles.ctypes.data, bank.needles.size, int(drop),
                                  w.ctypes.data if nw else None, nw, fw, fh, float(thr), sim.ctypes.data, d.ctypes.data, info.ctypes.data, info.size)
    assert rc == 0
    return sim[:nw], d[:nw], info.reshape(-1, 4)


def _patches(page_luma, fw, fh, rng, n):
    inv = 255 - page_luma
    ys = rng.integers(1, inv.shape[0] - fh, n)
    xs = rng.integers(1, inv.shape[1] - fw, n)
    return np.stack([inv[y:y + fh, x:x + fw].reshape(-1) for y, x in zip(ys, xs)])


def _bank_of(needles):
    tm, flat, off = [], [], 0
    for nd in needles:
        t = np.zeros(1, TEMPLATE_DTYPE)
        t["letter"], t["n_w"], t["n_h"], t["offset"] = 65 + len(tm) % 26, nd.shape[1], nd.shape[0], off
        tm.append(t)
        flat.append(nd.reshape(-1).astype(np.uint8))
        off += nd.size
    return Bank(np.concatenate(tm), np.concatenate(flat), len(tm), 0, 0, 13.0, 8.0)


THRESHOLDS = (0.8, 0.3, 0.99, 0.0, -0.05, -0.25, -0.5, -0.9)


@pytest.mark.parametrize("drop", [True, False], ids=["column-drop", "full-width"])
def test_glyph_bank_bound_holds(bank_x2, drop):
    _, _, info = _model(bank_x2, np.zeros((0, 135)), 9, 15, 0.8, drop)
    # classes in order of first appearance: 8x15 (shift 0), 9x15 (shifts 1/4, 1/2, 3/4)
    assert info[0][3] == 8 and info[1][3] == (8 if drop else 9)
    assert info[0][2] == 0 and (info[1][2] > 0.1) == drop  # rho_max: only a dropped column has one
    rng = np.random.default_rng(3)
    page = synth_page(bank_x2, SYNTH_SEED_BASE + 11, 608, 720)
    text = _patches(page, 9, 15, rng, 2500)
    noise = rng.integers(0, 256, (300, 135), dtype=np.uint8)
    flat = np.full((3, 135), 77, np.uint8)                   # zero variance: never emits
    spike = np.zeros((4, 135), np.uint8)
    spike[np.arange(4), [0, 8, 126, 134]] = 255              # a single pixel: tiny norms, extreme ratios
    col8 = np.zeros((3, 15, 9), np.uint8)
    col8[0, :, 8] = 200                                      # ink only in the dropped column: the kept box is flat
    col8[1, ::2, 8] = 255                                    # ... and as rough as it gets
    col8[2, :, :8] = 255                                     # kept box saturated, dropped column blank
    # every 9-wide template itself, and with its last column replaced by noise / inverted: similarity ~1 with the bound loaded
    own = []
    for t in range(95, 380, 7):
        nd = bank_x2.needle(t).copy()
        own.append(nd.reshape(-1))
        nd2 = nd.copy()
        nd2[:, 8] = rng.integers(0, 256, 15)
        own.append(nd2.reshape(-1))
        nd3 = nd.copy()
        nd3[:, 8] = 255 - nd3[:, 8]
        own.append(nd3.reshape(-1))
    wins = np.concatenate([text, noise, flat, spike, col8.reshape(3, -1), np.stack(own)])
    for thr in THRESHOLDS:
        sim, d, _ = _model(bank_x2, wins, 9, 15, thr, drop)
        emits = sim > thr  # NaN (never emits) compares false
        assert emits.sum() > (50 if thr >= 0.8 else 100) or thr > 0.9
        missed = emits & ~(d > 0)
        assert not missed.any(), (thr, drop, int(missed.sum()), sim[missed][:5], d[missed][:5])
        if thr == 0.8:  # the filter filters: of the pairs the reference rejects, well under 1 % pass
            pairs = np.isfinite(sim).sum()
            assert ((d > 0) & ~emits).sum() < 0.004 * pairs, (((d > 0) & ~emits).sum(), pairs)


def test_column_drop_costs_few_candidates(bank_x2):
    """On page windows the bound for the dropped ninth column admits ~1.2-1.3 x the candidates of the full-width filter
    (measured on whole pages: DESIGN.md section 4) — the price of a third fewer MFMAs."""
    rng = np.random.default_rng(5)
    page = synth_page(bank_x2, SYNTH_SEED_BASE + 12, 608, 720)
    text = _patches(page, 9, 15, rng, 5000)
    _, d1, _ = _model(bank_x2, text, 9, 15, 0.8, True)
    _, d0, _ = _model(bank_x2, text, 9, 15, 0.8, False)
    c1, c0 = int((d1 > 0).sum()), int((d0 > 0).sum())
    assert c0 > 200 and c0 <= c1 < 1.6 * c0, (c0, c1)


def _two_level(w, h, lo=100, hi=101, seed=0):
    """Half the pixels `lo`, half `hi`: mean-centred values +-1/2 quantise EXACTLY (e_max = 0), so nothing but the margins of
    the threshold arithmetic stands between a near-threshold similarity and the filter."""
    rng = np.random.default_rng(seed)
    nd = np.full(w * h, lo, np.uint8)
    nd[rng.permutation(w * h)[: w * h // 2]] = hi
    return nd.reshape(h, w)


@pytest.mark.parametrize("thr", [-0.05, -0.25, -0.3, -0.9, 0.25])
def test_adversarial_exact_bank_near_threshold_windows(thr):
    """a bank whose quantisation is exact (e_max = 0) and uniform-noise windows whose similarity lies
    within 1e-4 of the threshold — picked out of millions in numpy — are where a threshold formed from a LOWER bound of the
    window norm misses true hits when kappa < 0.  The model (= the device arithmetic) must flag every emitting one."""
    nd = _two_level(8, 15)
    bank = _bank_of([nd])
    _, _, info = _model(bank, np.zeros((0, 120)), 8, 15, thr)
    assert info[0][1] < 1e-9 and info[0][3] == 8  # e_max = 0
    b = nd.reshape(-1).astype(np.float64)
    beta = (b - b.mean()) / np.sqrt(((b - b.mean()) ** 2).sum())
    rng = np.random.default_rng(20261004)
    near, old_missed = [], 0
    sgn = np.sign(beta)
    for it in range(3):
        if it < 1 and abs(thr) <= 0.3:  # plain uniform noise (the recipe of the round-2 review)
            a = rng.integers(0, 256, (500_000, 120), dtype=np.uint8)
        else:  # noise of amplitude +-30 around mid-grey plus lam x the template's sign pattern, lam around the value that gives thr
            lam0 = thr * 17.6 / np.sqrt(1.0 - thr * thr)
            lam = rng.uniform(lam0 - 0.15 * abs(lam0) - 0.5, lam0 + 0.15 * abs(lam0) + 0.5, (500_000, 1))
            a = np.clip(np.rint(128.0 + lam * sgn[None, :] + rng.uniform(-30.5, 30.5, (500_000, 120))), 0, 255).astype(np.uint8)
        af = a.astype(np.float32)
        s = af.sum(1, dtype=np.float64)
        s2 = (af * af).sum(1, dtype=np.float64)
        norm = np.sqrt(s2 - s * s / 120.0)
        sim = (af @ beta.astype(np.float32)).astype(np.float64) / norm
        pick = np.abs(sim - thr) < 1e-4
        near.append(a[pick])
    near = np.concatenate(near)
    assert len(near) > 60
    sim, d, _ = _model(bank, near, 8, 15, thr)
    emits = sim[:, 0] > thr
    assert emits.sum() > 20
    missed = emits & ~(d[:, 0] > 0)
    assert not missed.any(), (thr, int(missed.sum()), sim[missed, 0][:5] - thr, d[missed, 0][:5])
    # teeth: round 2's arithmetic (f16 norm rounded TOWARDS ZERO, multiplied by kappa whatever its sign) on the same windows
    if thr < 0:
        c = 126.0 / np.abs(beta).max()
        kappa = c * thr - 1e-4 * (c * (1.0 + abs(thr)))
        a = near.astype(np.float64)
        V = 120.0 * (a * a).sum(1) - a.sum(1) ** 2
        nrm32 = np.sqrt((V.astype(np.float32) * np.float32(1.0 / 120.0)).astype(np.float32)).astype(np.float32)
        h = nrm32.astype(np.float16)
        h = np.where(h.astype(np.float32) > nrm32, np.nextafter(h, np.float16(0)), h).astype(np.float32)  # towards zero
        L_old = np.floor(np.float32(kappa) * h) - 2.0
        G = np.rint(c * (near.astype(np.float64) - 128.0) @ beta)  # exact: c * beta = +-126
        old_missed = int((emits & ~(G > L_old)).sum())
        print(f"thr {thr}: {len(near)} near-threshold windows, {int(emits.sum())} emit; round 2's threshold would drop {old_missed}")
        assert old_missed > 0 or thr > -0.2  # the windows are adversarial: the old arithmetic does lose hits here


def test_dropped_column_adversarial():
    """The column-drop bound at its limit: templates whose ninth column carries a lot of their energy, windows built from the
    templates with that column perturbed.  Emitting pairs must be flagged; Cauchy-Schwarz may not be overtaken."""
    rng = np.random.default_rng(7)
    needles = []
    for k in range(48):
        nd = rng.integers(0, 256, (15, 9), dtype=np.uint8)
        if k % 3 == 0:
            nd[:, :8] = rng.integers(100, 104, (15, 8))   # nearly flat kept box, loud last column
        if k % 3 == 1:
            nd[:, 8] = nd[:, 7]                            # last column correlated with its neighbour
        needles.append(nd)
    needles.append(_two_level(9, 15, seed=3))
    needles.append(np.full((15, 9), 9, np.uint8))          # constant: never emits
    bank = _bank_of(needles)
    _, _, info = _model(bank, np.zeros((0, 135)), 9, 15, 0.5)
    assert info[0][3] == 8 and info[0][2] > 0.3            # rho_max is large here
    wins = []
    for nd in needles[:-1]:
        for _ in range(40):
            w = nd.astype(np.int64) + rng.integers(-40, 41, nd.shape) * (rng.random() < 0.7)
            if rng.random() < 0.5:
                w[:, 8] = rng.integers(0, 256, 15)
            wins.append(np.clip(w, 0, 255).astype(np.uint8).reshape(-1))
    wins = np.stack(wins)
    for thr in (0.9, 0.5, 0.1, -0.2, -0.7):
        sim, d, _ = _model(bank, wins, 9, 15, thr)
        emits = sim > thr
        assert emits.sum() > 500
        missed = emits & ~(d > 0)
        assert not missed.any(), (thr, int(missed.sum()), sim[missed][:5], d[missed][:5])


def test_plane_value_helper_matches_numpy():
    """The plane's value, -floor((L - 2) / S) as int16 clamped to +-32767, against numpy on a sweep of thresholds."""
    lib = N.hip()
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.normal(0, 1, 20000) * 10.0 ** rng.integers(0, 8, 20000), [0.0, 2.0, 65.9, 66.0, -61.9, -62.0, 1e9, -1e9, np.inf, -np.inf]]).astype(np.float32)
    for shift in (5, 6, 10):
        out = np.zeros(len(x), np.int16)
        lib.focr_debug_plane_value(x.ctypes.data_as(C.c_void_p), len(x), shift, out.ctypes.data_as(C.c_void_p))
        with np.errstate(invalid="ignore", over="ignore"):
            t = np.floor((x - np.float32(2.0)).astype(np.float32) * np.float32(2.0 ** -shift))
        want = (-np.clip(t, -32767.0, 32767.0)).astype(np.int16)
        assert np.array_equal(out, want), (shift, x[out != want][:5], out[out != want][:5], want[out != want][:5])
