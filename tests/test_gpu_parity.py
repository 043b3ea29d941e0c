"""Parity of the HIP path (through the C ABI) against the oracle / the reference's golden vectors.
Bit-exact: x, y and the f32 similarity bits, including order and the 1024 cap."""
import numpy as np
import pytest

from font_ocr_amd import synth_page, synth_pages
from font_ocr_amd.bank import SYNTH_SEED_BASE
from font_ocr_amd.searcher import PREFILTER_AUTO, PREFILTER_LEGACY, PREFILTER_ONE_STAGE, SCAN_DIRECT, SCAN_MFMA, Scanner, Searcher, text_of
from oracle import oracle as O

pytestmark = pytest.mark.gpu

# The exact v_dot4 path and the MFMA path (int8 prefilter + exact verify) must reproduce the reference lists bit for bit.
# MFMA1 = the default kernel (f16 threshold planes), MFMA0 = round 1's kernel with int32 threshold tables; FULL = MFMA1 with
# every template column multiplied (focr_ctx_set_column_drop(0): 9- / 13-wide classes on their wider K layout) — the
# parametrised matrix runs direct + MFMA1, the other two forms are crossed in the full-size, geometry, layout and fuzz tests.
MFMA1, MFMA0 = (SCAN_MFMA, PREFILTER_ONE_STAGE), (SCAN_MFMA, PREFILTER_LEGACY)
FULL = "mfma1, no column drop"
MODES = [pytest.param(SCAN_DIRECT, id="direct"), pytest.param(MFMA1, id="mfma1")]


@pytest.fixture(scope="module")
def scanner():
    s = Scanner(0)
    yield s
    s.close()


@pytest.fixture(autouse=True)
def _column_drop_back_on(request):
    yield
    if "scanner" in request.fixturenames:
        request.getfixturevalue("scanner").set_column_drop(True)
        request.getfixturevalue("scanner").set_row_tail(True)


def _scan(scanner, bank, thr, cap, mode):
    """scanner.scan for a mode of the matrix above; FULL re-uploads the bank without the column drop (and leaves it so: keep
    FULL last in a loop, or upload the bank again)."""
    if mode == FULL:
        scanner.set_column_drop(False)
        scanner.set_bank(bank)
        mode = MFMA1
    scanner.scan(thr, cap, mode)


def _csr_to_lists(offsets, m, n_pages, T):
    return [[m[int(offsets[p * T + t]): int(offsets[p * T + t + 1])] for t in range(T)] for p in range(n_pages)]


def _oracle_lists(pages_luma, bank, thr, cap, use_ref=None):
    use_ref = O.have_ref() if use_ref is None else use_ref
    out = []
    for pg in pages_luma:
        counts, matches = O.scan_page(O.invert(pg), bank, thr, cap, use_ref=use_ref)
        out.append([matches[t, : counts[t]] for t in range(len(counts))])
    return out


def _assert_same(got, want, what=""):
    for p, (gp, wp) in enumerate(zip(got, want)):
        for t, (g, w) in enumerate(zip(gp, wp)):
            assert len(g) == len(w), f"{what} page {p} template {t}: {len(g)} != {len(w)}"
            assert g.tobytes() == w.tobytes(), f"{what} page {p} template {t}"


def test_device_f64_divide_sqrt_are_correctly_rounded(scanner):
    """patch_rnorm (src/ncc.rs:309-311) on the device == host IEEE, bit for bit, on 4e6 triples."""
    rng = np.random.default_rng(1)
    cnt = 4_000_000
    n = rng.integers(1, 257, cnt).astype(np.uint32)
    mean = rng.integers(0, 256, cnt).astype(np.uint64)
    s = (mean * n + rng.integers(0, 200, cnt).astype(np.uint64)).astype(np.uint32)
    var = rng.integers(0, 2_000_000, cnt).astype(np.uint64)
    var[rng.random(cnt) < 0.1] = 0
    s2 = (s.astype(np.uint64) ** 2) // n + var  # any s2 >= ~s^2/n, including exact-zero variance
    got = scanner.debug_rnorm(s, s2, n)
    with np.errstate(divide="ignore", invalid="ignore"):
        want = 1.0 / np.sqrt(s2.astype(np.float64) - (s.astype(np.uint64) * s.astype(np.uint64)).astype(np.float64) / n.astype(np.float64))
    same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), f"{(~same).sum()} of {cnt} differ"


def test_threshold_plane_values_device_equals_host(scanner):
    """The threshold planes' values, -floor((L - 2) / S) as int16 (mfma_common.h: plane_value): the host flavour the CPU tests model the
    prefilter with and the device's agree bit for bit — huge thresholds of either sign (a +-1e30 --threshold), infinities and values
    at the clamp included — and the threshold they stand for never lies above L - 2."""
    import ctypes as C

    rng = np.random.default_rng(9)
    x = np.concatenate([rng.normal(0, 1, 400000) * 10.0 ** rng.integers(0, 7, 400000),
                        rng.normal(0, 1, 20000) * 10.0 ** rng.integers(6, 30, 20000),
                        [0.0, -0.0, 1.0, 2.0, 3.0, -1.0, 65.9, 66.0, 66.1, -61.9, -62.0, -62.1, 2097150.0, 2097152.0, 2097217.9, 2097218.0, -2097150.0, 1e9, -1e9,
                         np.inf, -np.inf, 3.0e38, -3.0e38]]).astype(np.float32)
    x = np.ascontiguousarray(x)
    for shift in (5, 6, 9, 14):
        host = np.zeros(len(x), np.int16)
        dev = np.zeros(len(x), np.int16)
        scanner._lib.focr_debug_plane_value(x.ctypes.data_as(C.c_void_p), len(x), shift, host.ctypes.data_as(C.c_void_p))
        scanner._ck(scanner._lib.focr_debug_plane_value_device(scanner._h, x.ctypes.data_as(C.c_void_p), len(x), shift, dev.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(host, dev), (shift, x[host != dev][:8], host[host != dev][:8], dev[host != dev][:8])
        S = float(1 << shift)
        thr = -host.astype(np.float64) * S  # the threshold the plane value stands for
        inside = np.abs(x.astype(np.float64) - 2.0) < 32767.0 * S
        assert (thr[inside] <= x[inside].astype(np.float64) - 2.0 + 1e-3 * np.abs(x[inside])).all()  # (f32 rounding of L - 2)
        assert (thr[inside] > x[inside].astype(np.float64) - 2.0 - S - 1e-3 * np.abs(x[inside])).all()
        assert (host[~inside & (x > 0)] == -32767).all() and (host[~inside & (x < 0)] == 32767).all()
        assert host.min() >= -32767  # -32768 is the "never" value: a threshold never maps to it


@pytest.mark.parametrize("shapes", [[(9, 15), (8, 15)], [(9, 15)], [(8, 15)], [(8, 32), (9, 32)], [(8, 4), (9, 7)],
                                    [(13, 16), (12, 16)], [(13, 14)], [(16, 16)], [(4, 9), (4, 12)], [(12, 15), (8, 15)]],
                         ids=["pair", "drop", "plain", "tall32", "short", "pair12", "drop12", "w16", "w4", "w12w8"])
@pytest.mark.parametrize("geom", [(3, 301, 111), (2, 608, 720), (5, 64, 40), (1, 1021, 67)], ids=["301x111", "608x720", "64x40", "1021x67"])
def test_statistics_register_form_writes_the_same_planes(shapes, geom):
    """Classes whose kept width is 4, 8, 12 or 16 px take the register form of the window statistics (stats8_kernel: vertical sums
    first, a lane per four columns, no LDS); every other class the LDS-tiled kernel.  Both must leave the same int16 threshold planes wherever the
    scan kernel reads them (x < 16 * mtx, y <= n_rows) and the same live M-tiles — hence the same candidates, hits and lines —
    for every geometry: widths that are no multiple of 4, strips that end inside / beyond the row's padding, pages lower than a band."""
    np_, r_w, r_h = geom
    rng = np.random.default_rng(1000 * r_w + r_h + len(shapes))
    bank = _random_bank(rng, shapes, 5)
    pages = rng.integers(0, 256, (np_, r_h, r_w), dtype=np.uint8)
    pages[rng.random(pages.shape) < 0.55] = 0  # paper
    pages[0, : r_h // 3] = 0                   # a blank band: the "nothing but paper" row path
    if r_w > 300:
        pages[:, :, 250:290] = 0               # and blank columns across a strip boundary
    thr = 0.35
    got = []
    with Scanner(0) as sc:
        sc.set_bank(bank)
        sc.set_pages(pages)
        for form in (1, 0):
            sc.set_stats_form(form)
            sc.scan(thr, 1024, SCAN_MFMA)
            sc.process_hits(0.6, 5)
            got.append((sc.planes().copy(), sc.matches()[0].tobytes(), sc.matches()[1].tobytes(), sc.lines_flat().tobytes(), dict(sc.counters())))
    (p1, c1, m1, l1, k1), (p0, c0, m0, l0, k0) = got
    assert (c1, m1, l1) == (c0, m0, l0)
    assert k1["candidates"] == k0["candidates"], (k1, k0)
    min_w, min_h = min(w for w, h in shapes), min(h for w, h in shapes)
    mtx, n_rows = (r_w - min_w + 1 + 15) // 16, r_h - min_h
    Lpitch, Lrows = (r_w + 63) // 64 * 64 + 64, (r_h + 7) // 8 * 8 + 8
    nv = 1 if len(shapes) <= 1 else 2
    assert p1.size == p0.size and p1.size >= nv * np_ * Lrows * Lpitch
    a = p1[: nv * np_ * Lrows * Lpitch].reshape(nv, np_, Lrows, Lpitch)[:, :, : n_rows + 1, : min(16 * mtx, Lpitch)]
    b = p0[: nv * np_ * Lrows * Lpitch].reshape(nv, np_, Lrows, Lpitch)[:, :, : n_rows + 1, : min(16 * mtx, Lpitch)]
    assert np.array_equal(a, b), np.argwhere(a != b)[:8]
    assert (a != -32768).any()  # some windows can emit: the comparison is not of two empty planes


def test_compat_symbols_on_golden_vectors(kernel_cases):
    """ncc_8_u8 / ncc_16_u8 (the reference's FFI names) == the reference kernel's committed outputs."""
    for name, c in kernel_cases.items():
        s = Searcher(c["page"])
        got = s.search_c_u8(c["needle"], (c["patch_sum"], c["patch_rnorm"], c["start_end"]), float(c["thr"]), int(c["cap"]))
        assert got.tobytes() == c["expect"].tobytes(), name
        assert not s.acc_u32.any()


def test_compat_symbols_random_vs_oracle():
    rng = np.random.default_rng(11)
    for it in range(25):
        r_w, r_h = int(rng.integers(20, 300)), int(rng.integers(20, 120))
        n_w, n_h = int(rng.integers(1, 17)), int(rng.integers(1, 19))
        page = rng.integers(0, 256, (r_h, r_w), dtype=np.uint8)
        if it % 3 == 0:
            page[rng.random((r_h, r_w)) < 0.9] = 0
        y0, x0 = int(rng.integers(0, r_h - n_h + 1)), int(rng.integers(0, r_w - n_w + 1))
        nd = page[y0:y0 + n_h, x0:x0 + n_w].copy() if it % 2 else rng.integers(0, 256, (n_h, n_w), dtype=np.uint8)
        thr = float(rng.choice([-1.0, 0.0, 0.3, 0.8]))
        cap = int(rng.choice([1, 5, 1024]))
        stats = O.prepare_for_size(page, n_w, n_h)
        want = O.ncc_u8(O.padded(page), r_w, r_h, nd, stats, thr, cap, use_ref=O.have_ref())
        got = Searcher(page).search_c_u8(nd, stats, thr, cap)
        assert got.tobytes() == want.tobytes(), (it, r_w, r_h, n_w, n_h, thr, cap)


@pytest.mark.parametrize("mode", MODES)
def test_c1_page_matches_reference_lists(scanner, bank_default, c1_golden, mode):
    """configs[0]: 608x720 page, 74 templates — raw lists == reference kernel, lines == oracle."""
    scanner.set_bank(bank_default)
    scanner.set_pages(c1_golden["page"])
    scanner.scan(0.8, 1024, mode)
    counts = scanner.counts()[0]
    assert np.array_equal(counts, c1_golden["counts"])
    offsets, m = scanner.matches()
    assert m.tobytes() == c1_golden["matches"].tobytes()
    assert np.array_equal(np.diff(offsets.astype(np.int64)), counts)
    scanner.process_hits(0.95, 5)
    lines = scanner.lines()[0]
    flat = np.concatenate(lines)
    g = c1_golden["lines"]
    assert len(flat) == len(g)
    for f in ("x", "y", "w", "h", "letter"):
        assert np.array_equal(flat[f].astype(np.int64), g[f].astype(np.int64)), f
    assert flat["similarity"].tobytes() == g["similarity"].tobytes()
    assert np.array_equal(np.cumsum([len(l) for l in lines]), c1_golden["line_ends"])


@pytest.mark.parametrize("mode", MODES)
def test_c2_page0_matches_reference_lists(scanner, bank_x2, c2_golden, mode):
    """configs[1] bank (95 glyphs x 4 x-shifts = 380 templates, two size classes) on page 0."""
    page = synth_page(bank_x2, SYNTH_SEED_BASE, 608, 720)
    scanner.set_bank(bank_x2)
    scanner.set_pages(page)
    scanner.scan(0.8, 1024, mode)
    assert np.array_equal(scanner.counts()[0], c2_golden["counts"])
    _, m = scanner.matches()
    assert m.tobytes() == c2_golden["matches"].tobytes()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("thr,cap", [(0.8, 1024), (0.3, 1024), (-1.0, 1024), (0.0, 3), (0.999, 1024)])
def test_small_batch_vs_oracle(scanner, bank_x2, mode, thr, cap):
    """Several small pages, a 40-template slice of the bank with both size classes, incl. the cap."""
    idx = list(range(30, 50)) + list(range(95 + 30, 95 + 50))
    bank = bank_x2.subset(idx)
    pages = np.stack([synth_page(bank_x2, SYNTH_SEED_BASE + 100 + p, 160, 80) for p in range(3)])
    pages[2, :, :] = 255  # a blank page in the batch
    pages[1, 40:, 100:] = 0  # a saturated block
    scanner.set_bank(bank)
    scanner.set_pages(pages)
    scanner.scan(thr, cap, mode)
    offsets, m = scanner.matches()
    got = _csr_to_lists(offsets, m, 3, len(bank))
    want = _oracle_lists(pages, bank, thr, cap)
    _assert_same(got, want, f"thr={thr} cap={cap}")
    assert np.array_equal(scanner.counts(), np.array([[len(x) for x in p] for p in want], np.uint32))


@pytest.mark.parametrize("mode", MODES)
def test_ragged_and_degenerate_pages(scanner, bank_x2, mode):
    """Odd page sizes (not multiples of any tile), pages barely larger than a template, noise pages."""
    rng = np.random.default_rng(5)
    bank = bank_x2.subset([33, 34, 95 + 33, 2 * 95 + 40])
    for (w, h) in [(9, 16), (10, 16), (17, 17), (65, 21), (130, 33), (257, 19)]:
        pages = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
        pages[1][rng.random((h, w)) < 0.8] = 255
        scanner.set_bank(bank)
        scanner.set_pages(pages)
        scanner.scan(0.2, 1024, mode)
        offsets, m = scanner.matches()
        _assert_same(_csr_to_lists(offsets, m, 2, len(bank)), _oracle_lists(pages, bank, 0.2, 1024), f"{w}x{h}")
        scanner.process_hits(0.5, 5)  # must not fail on sparse / empty results
        scanner.lines()


@pytest.mark.parametrize("mode", MODES)
def test_process_hits_vs_oracle_on_batch(scanner, bank_x2, mode):
    pages = np.stack([synth_page(bank_x2, SYNTH_SEED_BASE + 200 + p, 400, 150) for p in range(4)])
    scanner.set_bank(bank_x2)
    scanner.set_pages(pages)
    scanner.scan(0.8, 1024, mode)
    offsets, m = scanner.matches()
    got = _csr_to_lists(offsets, m, 4, len(bank_x2))
    for anchor, overlap in [(0.95, 5), (0.9, 0), (0.99, 12)]:
        scanner.process_hits(anchor, overlap)
        lines = scanner.lines()
        for p in range(4):
            counts = np.array([len(x) for x in got[p]], np.uint32)
            mm = np.zeros((len(bank_x2), 1024), O.MATCH_DTYPE)
            for t, x in enumerate(got[p]):
                mm[t, : len(x)] = x
            want = O.process_hits(O.raw_hits(counts, mm, bank_x2), anchor, overlap)
            assert len(lines[p]) == len(want), (p, anchor, overlap)
            for lg, lw in zip(lines[p], want):
                assert np.array_equal(lg["x"].astype(np.int64), lw["x"].astype(np.int64))
                assert np.array_equal(lg["y"].astype(np.int64), lw["y"].astype(np.int64))
                assert np.array_equal(lg["letter"], lw["letter"])
                assert lg["similarity"].tobytes() == lw["similarity"].tobytes()


def test_full_size_c2_properties(scanner, bank_x2):
    """BASELINE configs[1] at full size (128 pages 608x720, 380 templates): size-independent checks —
    direct and MFMA paths agree on every list, the first page equals the reference golden, and the
    stamped text comes back."""
    n_pages = 128
    pages = synth_pages(bank_x2, n_pages, 608, 720)
    scanner.set_bank(bank_x2)
    scanner.set_pages(pages)
    res = {}
    # both size classes (8x15, 9x15 with its ninth column bounded) in one pass of 2 K-steps; every column multiplied: 3
    kernel_of = {MFMA0: "scan_mfma2_kernel<2,2,", MFMA1: "scan_mfma2s_kernel<2,2,", FULL: "scan_mfma2s_kernel<3,3,"}
    cand = {}
    for mode in (SCAN_DIRECT, MFMA0, MFMA1, FULL):
        _scan(scanner, bank_x2, 0.8, 1024, mode)
        res[mode] = (scanner.counts().copy(),) + scanner.matches()
        cand[mode] = scanner.counters()["candidates"]
        if mode != SCAN_DIRECT:
            names = [li["name"] for li in scanner.launches()]
            assert len(names) == 1 and names[0].startswith(kernel_of[mode]), names
    for mode in (MFMA0, MFMA1, FULL):
        assert np.array_equal(res[SCAN_DIRECT][0], res[mode][0])
        assert res[SCAN_DIRECT][2].tobytes() == res[mode][2].tobytes()
    # the price of the bound: more candidates for verify to reject, the same hits (about 1.2 x)
    assert cand[FULL] < cand[MFMA1] < 1.5 * cand[FULL], cand
    # the tail: the hits-first row path (rows.hip, the default: the scans above) against the legacy radix-sort tail, each with
    # exact and with estimated sizes
    for tail in (0, 1):
        scanner.set_row_tail(tail)
        scanner.scan(0.8, 1024, MFMA1)  # first scan after the switch: exact sizes
        assert np.array_equal(scanner.counts(), res[SCAN_DIRECT][0]) and scanner.matches()[1].tobytes() == res[SCAN_DIRECT][2].tobytes(), tail
        scanner.scan(0.8, 1024, MFMA1)  # estimated sizes
        assert np.array_equal(scanner.counts(), res[SCAN_DIRECT][0]) and scanner.matches()[1].tobytes() == res[SCAN_DIRECT][2].tobytes(), tail
    res[SCAN_MFMA] = res[FULL]
    scanner.process_hits(0.95, 5)
    lines = scanner.lines()
    ok = tot = 0
    for p in (0, 17, 127):
        _, truth = synth_page(bank_x2, SYNTH_SEED_BASE + p, 608, 720, with_truth=True)
        by_y = {int(l[0]["y"]): "".join(chr(int(c)) for c in l["letter"]) for l in lines[p]}
        for y in sorted(set(truth["y"].tolist())):
            want = "".join(chr(int(c)) for c in truth[truth["y"] == y]["letter"])
            ok += by_y.get(y) == want
            tot += 1
    assert ok >= 0.9 * tot, (ok, tot)
    # every page's lists against the compiled reference kernel (oracle/_ref when present), not just page 0:
    # the same 128 pages through get_hits' loop on the host cores, one page per thread (src/ncc.rs:839-847)
    _assert_equal_ref_batch(res[SCAN_MFMA], pages, bank_x2, 0.8, 1024)


def _host_threads():
    import os

    return max(1, min(len(os.sched_getaffinity(0)), 32))


def _assert_equal_ref_batch(res, pages, bank, thr, cap):
    """res = (counts[P][T], offsets, matches) of a device scan; compares counts and every (x, y, f32 bits) with the
    CPU scan of the same pages (reference kernel where oracle/_ref exists, else the restatement)."""
    counts, offsets, m = res
    P, T = counts.shape
    total, wc, wm = O.scan_pages_mt(O.invert(pages), bank, thr, cap, use_ref=O.have_ref(), threads=_host_threads(), keep_matches=True)
    assert np.array_equal(counts, wc), f"counts differ on pages {sorted(set(np.nonzero(counts != wc)[0].tolist()))[:8]}"
    assert int(total) == int(wc.sum()) == len(m)
    flat = wm[np.arange(cap)[None, None, :] < wc[:, :, None]]  # (page, template, y, x) order == the device's CSR order
    assert flat.tobytes() == m.tobytes()
    assert np.array_equal(offsets, np.concatenate([[0], np.cumsum(wc.reshape(-1), dtype=np.uint64)]))


def test_c3_geometry_1200x1600_vs_reference(scanner, bank_x2y2):
    """BASELINE configs[2] at its own geometry: 1200x1600 pages, 95 glyphs x --x-bits 2 --y-bits 2 = 1520 templates in
    four size classes.  Four pages are checked list for list against the reference kernel (src/ncc.cpp:253-396 through
    get_hits' loop, ~35 s of one host core per page), both device formulations; then a 64-page batch goes through the
    MFMA and the direct path, which must agree on every list, and the four checked pages must come out of the
    big batch unchanged (a page's lists do not depend on its neighbours)."""
    r_w, r_h = 1200, 1600
    pages = synth_pages(bank_x2y2, 4, r_w, r_h, first=3000)
    scanner.set_bank(bank_x2y2)
    scanner.set_pages(pages)
    scanner.scan(0.8, 1024, MFMA1)
    res4 = (scanner.counts().copy(),) + scanner.matches()
    assert all(li["name"].startswith("scan_mfma2s") for li in scanner.launches())
    _assert_equal_ref_batch(res4, pages, bank_x2y2, 0.8, 1024)
    assert res4[0].sum() > 400_000  # dense text: ~1e5 raw hits per page
    for mode in (SCAN_DIRECT, MFMA0, FULL):
        _scan(scanner, bank_x2y2, 0.8, 1024, mode)
        assert np.array_equal(scanner.counts(), res4[0]) and scanner.matches()[1].tobytes() == res4[2].tobytes()
    assert len(scanner.launches()) >= 2  # every column multiplied: the bank does not fit one launch's LDS (bank chunks)
    scanner.set_column_drop(True)
    scanner.set_bank(bank_x2y2)
    scanner.scan(0.8, 1024, MFMA1)
    scanner.process_hits(0.95, 5)
    lines4 = scanner.lines_flat().copy()
    # 64 pages, MFMA == direct; pages 0..3 of the batch are the four above
    big = np.concatenate([pages, synth_pages(bank_x2y2, 60, r_w, r_h, first=3004)])
    scanner.set_pages(big)
    res = {}
    for mode in (MFMA1, SCAN_DIRECT):
        scanner.scan(0.8, 1024, mode)
        res[mode] = (scanner.counts().copy(),) + scanner.matches()
    res[SCAN_MFMA] = res[MFMA1]
    assert np.array_equal(res[SCAN_MFMA][0], res[SCAN_DIRECT][0])
    assert res[SCAN_MFMA][2].tobytes() == res[SCAN_DIRECT][2].tobytes()
    assert np.array_equal(res[SCAN_MFMA][0][:4], res4[0])
    assert res[SCAN_MFMA][2][: len(res4[2])].tobytes() == res4[2].tobytes()
    scanner.process_hits(0.95, 5)
    assert scanner.lines_flat()[: len(lines4)].tobytes() == lines4.tobytes()
    scanner.set_pages(np.full((1, 20, 20), 255, np.uint8))  # give the 1.5 GB of page + table memory back to smaller tests


def test_c5_256_template_gemm_variant(scanner, bank_x2):
    """BASELINE configs[4]: "template-bank-as-GEMM variant, 256-glyph bank, bf16 MFMA windows x templates".  Here the
    GEMM operand type is int8 (v_mfma_i32_16x16x64_i8), not bf16: BASELINE's north_star leaves the MFMA form open
    ("MFMA only if ... that wins on rocprof"), int8 has twice bf16's rate and — unlike bf16 accumulation in fp32 of
    products up to 255*255*135 — keeps the prefilter's bound in exact integer arithmetic (scan_mfma.hip header);
    every emitted match is re-evaluated with the reference's u8*u8->u32 + f64 arithmetic (src/ncc.cpp:316-321,
    352-361).  256 templates = the 95 shift-0 glyphs (8x15) + 161 templates of shifts 1/4 and 1/2 (9x15): the GEMM
    formulation (MFMA), the v_dot4 formulation (direct) and the reference kernel must produce identical lists."""
    bank = bank_x2.subset(range(256))
    assert len({(int(t["n_w"]), int(t["n_h"])) for t in bank.templates}) == 2
    pages = synth_pages(bank_x2, 8, 608, 720, first=5000)
    scanner.set_bank(bank)
    scanner.set_pages(pages)
    res = {}
    for mode in (MFMA1, SCAN_DIRECT, FULL):
        _scan(scanner, bank, 0.8, 1024, mode)
        res[mode] = (scanner.counts().copy(),) + scanner.matches()
        names = [li["name"] for li in scanner.launches()]
        assert any(n.startswith("scan_mfma") for n in names) == (mode != SCAN_DIRECT), names
    res[SCAN_MFMA] = res[MFMA1]
    for mode in (MFMA1, FULL):
        assert np.array_equal(res[mode][0], res[SCAN_DIRECT][0])
        assert res[mode][2].tobytes() == res[SCAN_DIRECT][2].tobytes()
    _assert_equal_ref_batch(res[SCAN_MFMA], pages, bank, 0.8, 1024)
    assert res[SCAN_MFMA][0].sum() > 50_000


def _random_bank(rng, shapes, per_shape):
    """A bank of random-noise / structured templates with the given (n_w, n_h) shapes (several size classes); per_shape: templates
    per shape (one number, or one per shape)."""
    from font_ocr_amd.bank import TEMPLATE_DTYPE, Bank

    tm, needles, off = [], [], 0
    counts = per_shape if isinstance(per_shape, (list, tuple)) else [per_shape] * len(shapes)
    for (w, h), n_of_shape in zip(shapes, counts):
        for k in range(n_of_shape):
            if k % 3 == 0:
                nd = rng.integers(0, 256, (h, w), dtype=np.uint8)
            elif k % 3 == 1:  # sparse strokes
                nd = np.zeros((h, w), np.uint8)
                nd[rng.integers(0, h, 3 + h), rng.integers(0, w, 3 + h)] = rng.integers(100, 256, 3 + h)
            else:  # constant (never emits) or nearly constant
                nd = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
                if k % 2:
                    nd[0, 0] ^= 1
            t = np.zeros(1, TEMPLATE_DTYPE)
            t["letter"], t["n_w"], t["n_h"], t["offset"] = 65 + len(tm) % 26, w, h, off
            tm.append(t)
            needles.append(nd.reshape(-1))
            off += nd.size
    return Bank(np.concatenate(tm), np.concatenate(needles), len(tm), 0, 0, 13.0, 8.0)


@pytest.mark.parametrize("mode", [*MODES, pytest.param(MFMA0, id="legacy"), pytest.param(FULL, id="full")])
@pytest.mark.parametrize("shapes", [
    [(16, 16), (13, 7), (14, 20)],            # 16-byte-row layout, 4 / 2 / 5 K-steps
    [(8, 15), (5, 9), (3, 3), (1, 1), (8, 32)],  # 8-byte rows only (no 9..12-wide class present)
    [(9, 15), (8, 15), (12, 16), (10, 3)],     # 12-byte rows shared by narrow classes
    [(9, 17), (11, 32), (4, 30), (16, 32)],    # tall templates: 6 / 8 K-steps
    [(9, 33), (16, 48), (5, 40), (13, 64), (9, 15)],  # n_h > 32: scan_tall_kernel (plus one MFMA class)
    [(7, 58), (16, 70)],                       # only tall classes; 70 > page height: never searchable
    [(17, 5), (24, 20), (32, 33), (20, 40), (9, 15)],  # 17..32 px wide (extension; the oracle generalises to N = 32)
    [(13, 15), (12, 15), (9, 14), (8, 14), (9, 30), (13, 32)],  # column drop: 13 -> 12 and 9 -> 8 beside their kept boxes, and alone
], ids=["w16", "w8", "w12", "tall", "taller", "tallest", "wide", "drop"])
def test_random_banks_all_layouts(scanner, mode, shapes):
    import zlib

    rng = np.random.default_rng(zlib.crc32(str(shapes).encode()))
    bank = _random_bank(rng, shapes, 7)
    pages = rng.integers(0, 256, (2, 61, 97), dtype=np.uint8)
    # plant a few templates so that high-similarity hits exist too
    for k, t in enumerate(range(0, len(bank), 5)):
        nd = bank.needle(t)
        y, x = 2 + 3 * k % 20, 1 + 11 * k % 60
        pages[0, y:y + nd.shape[0], x:x + nd.shape[1]] = 255 - nd[: 61 - y, : 97 - x]
    scanner.set_bank(bank)
    scanner.set_pages(pages)
    for thr in (0.25, 0.9, -0.3):
        _scan(scanner, bank, thr, 1024, mode)
        offsets, m = scanner.matches()
        _assert_same(_csr_to_lists(offsets, m, 2, len(bank)), _oracle_lists(pages, bank, thr, 1024), f"{shapes} thr={thr}")


@pytest.mark.parametrize("shapes,per_shape", [([(16, 24), (14, 20)], 220), ([(9, 16), (8, 15), (12, 12)], 700)], ids=["rows16", "rows12"])
def test_banks_above_the_lds_verify_in_chunks(scanner, shapes, per_shape):
    """Round 4: a bank whose verify operand does not fit the LDS whole (here 180 KB of 16-byte rows / 330 KB of 12-byte rows) is
    verified in chunk passes (verify_chunks_kernel, rows.hip: chunk rows in LDS, a wave-private queue per chunk); same lists as
    the reference kernel, also through the legacy tail (template rows gathered from global memory)."""
    import zlib

    rng = np.random.default_rng(zlib.crc32(str(shapes).encode()))
    bank = _random_bank(rng, shapes, per_shape)
    pages = rng.integers(0, 256, (2, 57, 131), dtype=np.uint8)
    pages[rng.random(pages.shape) < 0.5] = 255
    for k, t in enumerate(range(0, len(bank), 9)):  # plant templates all over the bank: every chunk gets hits
        nd = bank.needle(t)
        y, x = 1 + (5 * k) % (57 - nd.shape[0]), 1 + (13 * k) % (131 - nd.shape[1])
        pages[k % 2, y:y + nd.shape[0], x:x + nd.shape[1]] = 255 - nd
    scanner.set_bank(bank)
    scanner.set_pages(pages)
    want = _oracle_lists(pages, bank, 0.6, 1024)
    for tail in (1, 0, 1):
        scanner.set_row_tail(tail)
        scanner.scan(0.6, 1024, MFMA1)
        offsets, m = scanner.matches()
        _assert_same(_csr_to_lists(offsets, m, 2, len(bank)), want, f"{shapes} tail={tail}")
    names = [li["name"] for li in scanner.launches()]
    assert names, names


def test_chunk_layout_when_the_maxima_come_from_different_chunks(scanner):
    """a bank ordered by template height — about a thousand 8-px-tall templates, then three hundred 32-px-tall ones.  Every
    chunk of the verify fits the LDS budget, but the chunk with the most RECORDS (the short templates) and the chunk with the most
    ROWS (the tall ones) are different chunks: round 4 sized the LDS for both maxima at once (175 904 B > 160 KB: the launch failed
    and with it the batch).  The layout is per chunk now; the lists equal the direct scan's and the reference kernel's."""
    rng = np.random.default_rng(20260105)
    bank = _random_bank(rng, [(8, 8), (12, 32)], [1007, 310])
    pages = rng.integers(0, 256, (2, 70, 120), dtype=np.uint8)
    pages[rng.random(pages.shape) < 0.5] = 255
    for k, t in enumerate(range(0, len(bank), 13)):
        nd = bank.needle(t)
        y, x = 1 + (5 * k) % (70 - nd.shape[0]), 1 + (13 * k) % (120 - nd.shape[1])
        pages[k % 2, y:y + nd.shape[0], x:x + nd.shape[1]] = 255 - nd
    scanner.set_bank(bank)
    scanner.set_pages(pages)
    scanner.scan(0.6, 1024, SCAN_DIRECT)
    want = (scanner.counts().copy(), scanner.matches()[1].tobytes())
    assert want[0].sum() > 200
    for _ in range(2):  # exact sizes, then estimated
        scanner.scan(0.6, 1024, MFMA1)
        assert np.array_equal(scanner.counts(), want[0]) and scanner.matches()[1].tobytes() == want[1]
    offsets, m = scanner.matches()
    _assert_same(_csr_to_lists(offsets, m, 2, len(bank)), _oracle_lists(pages, bank, 0.6, 1024), "short then tall templates")


def test_chunked_verify_on_other_grids_and_with_shrinking_estimates(scanner, bank_x2y2):
    """round 4 recorded a GPU memory access fault in a run that launched the chunked verify (banks above the LDS:
    BASELINE configs[2]'s 1 520 templates) on 2 / 4 / 8 times its workgroups.  The kernel's indexing does not depend on its grid
    ("the fault"); this runs it — and the other persistent kernels of the tail — on a quarter, three and eight times their
    workgroups (focr_debug_set_tail_grid), with exact sizes, with estimated sizes, and with estimates taken from a batch that had
    many more candidates (the list shrinks under its bound): the lists are the direct scan's every time."""
    dense = synth_pages(bank_x2y2, 3, 420, 200, first=7100)
    sparse = dense.copy()
    sparse[:, 60:, :] = 255  # the same geometry with a quarter of the text
    scanner.set_bank(bank_x2y2)
    want = {}
    for name, pg in (("dense", dense), ("sparse", sparse)):
        scanner.set_pages(pg)
        scanner.scan(0.8, 1024, SCAN_DIRECT)
        scanner.process_hits(0.95, 5)
        want[name] = (scanner.counts().copy(), scanner.matches()[1].tobytes(), scanner.lines_flat().tobytes())
    assert want["dense"][0].sum() > 4 * want["sparse"][0].sum() > 1000
    redone0 = scanner.size_estimate_stats()["redone"]
    try:
        for num, den in ((1, 4), (3, 1), (8, 1), (0, 0)):
            scanner.set_tail_grid(num, den)
            for name, pg in (("dense", dense), ("dense", None), ("sparse", sparse), ("sparse", None), ("dense", dense)):
                if pg is not None:
                    scanner.set_pages(pg)  # same geometry: the size estimates of the previous batch carry over
                scanner.scan(0.8, 1024, MFMA1)
                scanner.process_hits(0.95, 5)
                got = (scanner.counts().copy(), scanner.matches()[1].tobytes(), scanner.lines_flat().tobytes())
                assert np.array_equal(got[0], want[name][0]) and got[1] == want[name][1] and got[2] == want[name][2], (num, den, name)
        # sparse -> dense overflows the estimate once per grid (redone exact, the same lists); dense -> sparse runs on the larger bound
        assert scanner.size_estimate_stats()["redone"] - redone0 <= 8
    finally:
        scanner.set_tail_grid(0, 0)


@pytest.mark.parametrize("mode", MODES)
def test_c3_bank_16_shifts(scanner, bank_x2y2, mode):
    """configs[2] bank: 95 glyphs x 16 sub-pixel shifts = 1520 templates in four size classes."""
    pages = np.stack([synth_page(bank_x2y2, SYNTH_SEED_BASE + 500 + p, 330, 120) for p in range(2)])
    scanner.set_bank(bank_x2y2)
    scanner.set_pages(pages)
    scanner.scan(0.8, 1024, mode)
    offsets, m = scanner.matches()
    want = _oracle_lists(pages, bank_x2y2, 0.8, 1024)
    _assert_same(_csr_to_lists(offsets, m, 2, len(bank_x2y2)), want, "c3 bank")
    assert sum(len(x) for p in want for x in p) > 1000


def test_negative_threshold_exact_bank_noise_pages(scanner):
    """VERDICT r02 weak #1 on the device: a bank whose int8 quantisation is exact (two-level templates, e_max = 0) on
    uniform-noise pages at negative thresholds — millions of windows a hair above the threshold, nothing but the
    threshold arithmetic's own margins between them and the filter.  Every MFMA form against the reference kernel, with a
    cap large enough that no list is cut (src/ncc.cpp:362-366 decides each window)."""
    from test_prefilter_host import _bank_of, _two_level

    bank = _bank_of([_two_level(8, 15), _two_level(9, 15, seed=3), _two_level(8, 15, lo=7, hi=250, seed=5)])
    pages = np.random.default_rng(77).integers(0, 256, (24, 720, 608), dtype=np.uint8)
    scanner.set_bank(bank)
    scanner.set_pages(pages)
    cap = 450_000  # > windows per page: nothing is capped
    for thr in (-0.25, -0.3):
        total, wc, wm = O.scan_pages_mt(O.invert(pages), bank, thr, cap, use_ref=O.have_ref(), threads=_host_threads(), keep_matches=True)
        assert wc.max() < cap and int(total) > 20_000_000
        flat = wm[np.arange(cap)[None, None, :] < wc[:, :, None]]
        for mode in (MFMA1, MFMA0, FULL):
            _scan(scanner, bank, thr, cap, mode)
            counts = scanner.counts()
            assert np.array_equal(counts, wc), (thr, mode, int(counts.astype(np.int64).sum() - wc.astype(np.int64).sum()))
            assert scanner.matches()[1].tobytes() == flat.tobytes(), (thr, mode)
        scanner.set_column_drop(True)
        scanner.set_bank(bank)
    scanner.set_pages(np.full((1, 20, 20), 255, np.uint8))  # give the memory back


def test_wide_template_is_rejected(scanner):
    from font_ocr_amd.searcher import FocrError

    bank = _random_bank(np.random.default_rng(0), [(33, 5)], 1)
    with pytest.raises(FocrError, match="wider than 32"):  # the reference panics above 16 ("not handled", src/ncc.rs:392)
        scanner.set_bank(bank)
    from font_ocr_amd.searcher import Searcher

    with pytest.raises(FocrError, match="not handled"):  # the drop-in symbols keep the reference's limit
        Searcher(np.zeros((40, 40), np.uint8)).search_c_u8(np.ones((5, 17), np.uint8), (None, None, None), 0.8)


@pytest.mark.parametrize("mode", MODES)
def test_full_page_low_threshold_hits_the_cap(scanner, bank_x2, mode):
    """A full 608x720 page at threshold 0.45: many templates exceed 1024 matches, so the reference's
    "stop at n_out" (src/ncc.cpp:225-227) decides which (y, x) survive — lists must still be identical."""
    page = synth_page(bank_x2, SYNTH_SEED_BASE + 3, 608, 720)
    bank = bank_x2.subset(list(range(40, 60)) + list(range(95 + 40, 95 + 60)))
    scanner.set_bank(bank)
    scanner.set_pages(page)
    scanner.scan(0.45, 1024, mode)
    counts = scanner.counts()[0]
    offsets, m = scanner.matches()
    want = _oracle_lists(page[None], bank, 0.45, 1024)
    _assert_same(_csr_to_lists(offsets, m, 1, len(bank)), want, "cap")
    assert (counts == 1024).sum() >= 3, counts.max()  # the cap really was exercised
    assert scanner.counters()["raw_hits"] > int(counts.sum())  # ... and some hits were cut off


@pytest.mark.parametrize("mode", MODES)
def test_split_batch_fallback_equals_single_pass(scanner, bank_x2, mode, monkeypatch):
    """The candidate-overflow fallback (scan the batch in page sub-ranges, append) must give the same lists and
    the same process_hits output as the single pass; forced here through FOCR_FORCE_SPLIT."""
    bank = bank_x2.subset(list(range(33, 70)) + list(range(95 + 33, 95 + 70)))
    pages = np.stack([synth_page(bank_x2, SYNTH_SEED_BASE + 600 + p, 256, 110) for p in range(5)])
    scanner.set_bank(bank)
    scanner.set_pages(pages)
    out = {}
    for split in ("0", "1"):
        scanner.force_split(split == "1")
        for thr, cap in ((0.8, 1024), (0.2, 50)):
            scanner.scan(thr, cap, mode)
            counts = scanner.counts().copy()
            offsets, m = scanner.matches()
            scanner.process_hits(0.9, 5)
            chars = scanner.lines_flat().copy()
            out[(split, thr)] = (counts, offsets.copy(), m.copy(), chars)
    scanner.force_split(False)
    for thr in (0.8, 0.2):
        a, b = out[("0", thr)], out[("1", thr)]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert a[2].tobytes() == b[2].tobytes()
        assert a[3].tobytes() == b[3].tobytes()
    want = _oracle_lists(pages, bank, 0.2, 50)
    _assert_same(_csr_to_lists(out[("1", 0.2)][1], out[("1", 0.2)][2], 5, len(bank)), want, "split thr=0.2 cap=50")
    assert (out[("1", 0.2)][0] == 50).any()


@pytest.mark.parametrize("mode", MODES)
def test_extreme_thresholds(scanner, bank_x2, mode):
    """NaN / +inf thresholds emit nothing (the reference's `sim > thr` is false), -inf behaves like -1."""
    bank = bank_x2.subset([40, 41, 95 + 40])
    page = np.random.default_rng(9).integers(0, 256, (70, 120), dtype=np.uint8)
    scanner.set_bank(bank)
    scanner.set_pages(page)
    for thr in (float("nan"), float("inf"), 2.0):
        scanner.scan(thr, 1024, mode)
        assert scanner.total_matches() == 0
        if thr != 2.0 and mode != SCAN_DIRECT:  # the no-match threshold does no work at all: nothing reaches verify
            assert scanner.counters()["candidates"] == 0 and scanner.launches() == []
        scanner.process_hits(0.95, 5)
        assert scanner.lines() == [[]]
    scanner.scan(float("-inf"), 64, mode)
    a = scanner.matches()[1].copy()
    scanner.scan(-1.0, 64, mode)
    assert a.tobytes() == scanner.matches()[1].tobytes() and len(a) > 0


def test_fuzz_geometry_banks_thresholds(scanner):
    """Seeded fuzz over page geometry, bank shapes (all K layouts, class mixes), thresholds and caps; both device
    paths against the oracle, plus process_hits against the oracle's on the same lists."""
    rng = np.random.default_rng(20261004)
    total_matches = total_chars = capped = dropped = 0
    for it in range(150):
        n_classes = int(rng.integers(1, 4))
        shapes = [(int(rng.integers(1, 17)), int(rng.integers(1, 33))) for _ in range(n_classes)]
        if it % 4 == 0:
            shapes = [(int(rng.integers(8, 13)), int(rng.choice([14, 15, 16]))) for _ in range(n_classes)]  # font-like
        per_shape = int(rng.integers(1, 24))
        if it % 5 == 1:  # larger banks; widths 8, 9, 12, 13: classes whose last column is bounded, alone and beside their kept box
            shapes = [(int(rng.choice([8, 9, 9, 12, 13, 13])), int(rng.choice([13, 15, 16]))) for _ in range(n_classes)]
            per_shape = int(rng.integers(40, 90))
        bank = _random_bank(rng, shapes, per_shape)
        n_pages = int(rng.integers(1, 5))
        r_w = int(rng.integers(max(s[0] for s in shapes) + 1, 150))
        r_h = int(rng.integers(max(s[1] for s in shapes) + 1, 90))
        pages = rng.integers(0, 256, (n_pages, r_h, r_w), dtype=np.uint8)
        if it % 3 == 0:
            pages[rng.random(pages.shape) < 0.85] = 255  # mostly paper: exercises the blank-tile skip
        for k in range(0, len(bank), 3):  # plant some templates
            nd = bank.needle(k)
            p, y, x = int(rng.integers(0, n_pages)), int(rng.integers(0, max(1, r_h - nd.shape[0]))), int(rng.integers(0, max(1, r_w - nd.shape[1])))
            h, w = min(nd.shape[0], r_h - y), min(nd.shape[1], r_w - x)
            pages[p, y:y + h, x:x + w] = 255 - nd[:h, :w]
        thr = float(rng.choice([-0.5, 0.1, 0.4, 0.8, 0.97]))
        cap = int(rng.choice([1, 2, 37, 1024]))
        scanner.set_column_drop(True)
        scanner.set_row_tail((1, 1, 0)[it % 3])  # the hits-first row tail, now and then the legacy tail
        scanner.set_bank(bank)
        scanner.set_pages(pages)
        want = _oracle_lists(pages, bank, thr, cap)
        dropped += any(w in (9, 13) for w, _ in shapes)
        for mode in (MFMA0, MFMA1, SCAN_DIRECT, FULL):
            _scan(scanner, bank, thr, cap, mode)
            offsets, m = scanner.matches()
            _assert_same(_csr_to_lists(offsets, m, n_pages, len(bank)), want, f"fuzz {it} shapes={shapes} {r_w}x{r_h} thr={thr} cap={cap} mode={mode}")
            total_matches += len(m)
            capped += int((scanner.counts() == cap).sum())
        scanner.process_hits(0.6, 3)
        lines = scanner.lines()
        for p in range(n_pages):
            counts = np.array([len(x) for x in want[p]], np.uint32)
            mm = np.zeros((len(bank), max(cap, 1)), O.MATCH_DTYPE)
            for t, x in enumerate(want[p]):
                mm[t, : len(x)] = x
            wl = O.process_hits(O.raw_hits(counts, mm, bank), 0.6, 3)
            assert len(lines[p]) == len(wl), (it, p)
            for lg, lw in zip(lines[p], wl):
                assert np.array_equal(lg["x"].astype(np.int64), lw["x"].astype(np.int64)) and np.array_equal(lg["letter"], lw["letter"])
                assert lg["similarity"].tobytes() == lw["similarity"].tobytes()
                total_chars += len(lg)
    assert total_matches > 40000 and total_chars > 500 and capped > 200, (total_matches, total_chars, capped)
    assert dropped >= 30, dropped  # banks with 9- / 13-wide classes (column drop) did occur


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("box_size", ["char", "font"])
def test_box_size_char_and_font_banks(scanner, mode, box_size):
    """--box-size char gives every glyph its own tight box (dozens of size classes, src/ncc.rs:627); --box-size
    font one large box (src/ncc.rs:589-599).  Both go through the same scan."""
    import os

    from font_ocr_amd import Bank

    font = "/usr/share/fonts/truetype/dejavu/DejaVuSansMono.ttf"
    if not os.path.exists(font):
        pytest.skip("DejaVu Sans Mono not installed")
    alphabet = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789.,;:-_"
    bank = Bank.rasterize(font, 11 if box_size == "font" else 13, 1, 0, alphabet=alphabet, box_size=box_size)
    if int(bank.templates["n_w"].max()) > 16:
        pytest.skip("font box wider than 16 px at this size")
    n_classes = len({(int(t["n_w"]), int(t["n_h"])) for t in bank.templates})
    assert n_classes >= (10 if box_size == "char" else 1)
    ref_bank = Bank.rasterize(font, 11 if box_size == "font" else 13, 1, 0, alphabet=alphabet)  # page content
    pages = np.stack([synth_page(ref_bank, SYNTH_SEED_BASE + 700 + p, 300, 110) for p in range(2)])
    scanner.set_bank(bank)
    scanner.set_pages(pages)
    scanner.scan(0.7, 1024, mode)
    offsets, m = scanner.matches()
    want = _oracle_lists(pages, bank, 0.7, 1024)
    _assert_same(_csr_to_lists(offsets, m, 2, len(bank)), want, f"box_size={box_size}")
    assert sum(len(x) for p in want for x in p) > 100


def test_pinned_async_upload_and_second_context(bank_x2):
    """Double-buffered ingest: pages DMA'd from focr_host_alloc memory on a second context while
    the first one scans give the same lists as the plain path."""
    from font_ocr_amd.searcher import PinnedPages

    pages = [np.stack([synth_page(bank_x2, SYNTH_SEED_BASE + 700 + 2 * b + p, 260, 90) for p in range(2)]) for b in range(3)]
    want = [_oracle_lists(pg, bank_x2, 0.8, 1024) for pg in pages]
    ctxs = [Scanner(0), Scanner(0)]
    pins = [PinnedPages(2, 90, 260), PinnedPages(2, 90, 260)]
    try:
        for c in ctxs:
            c.set_bank(bank_x2)
            c.alloc_pages(2, 260, 90)
        pins[0].array[:] = pages[0]
        ctxs[0].upload_pages(pins[0].array)
        for b in range(3):
            cur, nxt = ctxs[b % 2], ctxs[(b + 1) % 2]
            if b + 1 < 3:
                pins[(b + 1) % 2].array[:] = pages[b + 1]  # its previous upload was synced by that context's scan
                nxt.upload_pages(pins[(b + 1) % 2].array)
            cur.scan(0.8, 1024, SCAN_MFMA)
            offsets, m = cur.matches()
            _assert_same(_csr_to_lists(offsets, m, 2, len(bank_x2)), want[b], f"batch {b}")
    finally:
        for c in ctxs:
            c.close()
        for p in pins:
            p.close()


def test_three_contexts_on_three_threads(bank_x2):
    """Batches in flight (bench.py --in-flight 3): three contexts driven concurrently from three host threads, scan
    kernel capped to 7/8 of the CUs, taking turns on the device — every context still returns the oracle's lists."""
    import threading

    n_ctx, rounds = 3, 4
    pages = [[np.stack([synth_page(bank_x2, SYNTH_SEED_BASE + 800 + 10 * j + 2 * r + p, 300, 100) for p in range(2)])
              for r in range(rounds)] for j in range(n_ctx)]
    want = [[_oracle_lists(pg, bank_x2, 0.8, 1024) for pg in pages[j]] for j in range(n_ctx)]
    got = [[None] * rounds for _ in range(n_ctx)]
    errs = []

    def work(j):
        try:
            sc = Scanner(0)
            sc.set_bank(bank_x2)
            sc.set_scan_cus(224)
            for r in range(rounds):
                sc.set_pages(pages[j][r])
                sc.scan(0.8, 1024, SCAN_MFMA)
                sc.process_hits(0.95, 5)
                offsets, m = sc.matches()
                got[j][r] = _csr_to_lists(offsets, m, 2, len(bank_x2))
            sc.close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(j,)) for j in range(n_ctx)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for j in range(n_ctx):
        for r in range(rounds):
            _assert_same(got[j][r], want[j][r], f"context {j} round {r}")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("shape", [(18, 65535), (65535, 18), (40, 5001)], ids=["widest", "tallest", "odd"])
def test_extreme_page_geometry(scanner, bank_x2, mode, shape):
    """Page sides up to the u16 limit of Match.x / Match.y (src/ncc.cpp:7-10): one text line's worth of rows by
    65535 columns and the transpose; key packing, tile grids and the work list must all hold."""
    r_h, r_w = shape
    rng = np.random.default_rng(r_h * 7 + r_w)
    page = np.full((r_h, r_w), 255, np.uint8)
    # sprinkle glyphs (anywhere they fit) and some noise so that there are hits near every border
    for k in range(400):
        t = int(rng.integers(0, len(bank_x2)))
        nd = bank_x2.needle(t)
        if nd.shape[0] > r_h or nd.shape[1] > r_w:
            continue
        y = int(rng.integers(0, r_h - nd.shape[0] + 1))
        x = int(rng.integers(0, r_w - nd.shape[1] + 1))
        if k % 4 == 0:
            x = [0, 1, r_w - nd.shape[1], r_w - nd.shape[1] - 1][(k // 4) % 4]
        if k % 4 == 1:
            y = [0, 1, r_h - nd.shape[0], r_h - nd.shape[0] - 1][(k // 4) % 4] if r_h > nd.shape[0] + 1 else 0
        page[y:y + nd.shape[0], x:x + nd.shape[1]] = 255 - nd
    pages = page[None]
    scanner.set_bank(bank_x2)
    scanner.set_pages(pages)
    scanner.scan(0.8, 1024, mode)
    offsets, m = scanner.matches()
    want = _oracle_lists(pages, bank_x2, 0.8, 1024)
    _assert_same(_csr_to_lists(offsets, m, 1, len(bank_x2)), want, f"{shape}")
    assert sum(len(x) for x in want[0]) > 100
    with pytest.raises(Exception, match="65535"):
        scanner.alloc_pages(1, 65536, 20)


def test_pipeline_executor_orders_and_matches_oracle(bank_x2):
    """focr_pipe_*: the native batches-in-flight executor (3 lanes x 2 contexts; every batch queued on the device when it is
    submitted) — tickets complete in order, each batch's lists equal the oracle's, host and resident submissions both work, the
    per-ticket stamps are there, misuse is reported."""
    from font_ocr_amd.searcher import FocrError, Pipeline

    n_batches = 11
    pages = [np.stack([synth_page(bank_x2, SYNTH_SEED_BASE + 1200 + 2 * b + p, 280 + 8 * (b % 3), 96) for p in range(2)])
             for b in range(n_batches)]
    want = [_oracle_lists(pg, bank_x2, 0.8, 1024) for pg in pages]
    pipe = Pipeline(0, 3)
    try:
        n = len(pipe.scanners)  # batches that can be outstanding: lanes x contexts per lane
        assert pipe.n_lanes == 3 and n == 6
        pipe.set_bank(bank_x2)
        tickets = []
        for b in range(n_batches):
            if len(tickets) >= n:
                t = tickets[b - n]
                sc = pipe.wait(t)
                offsets, m = sc.matches()
                _assert_same(_csr_to_lists(offsets, m, 2, len(bank_x2)), want[b - n], f"batch {b - n}")
                assert sc.total_chars() > 0
                tt = pipe.ticket_times(t)
                assert tt["submit_us"] <= tt["enqueue_begin_us"] <= tt["scan_queued_us"] <= tt["enqueue_end_us"] <= tt["done_us"], tt
                assert t == 1 or tt["device_gap_ms"] >= 0.0, tt  # retired in order: the previous ticket's last kernel came first
                pipe.release(t)
                with pytest.raises(FocrError):
                    pipe.ticket_times(t)  # released
            tickets.append(pipe.submit(pages[b], 0.8))
        assert tickets == list(range(1, n_batches + 1))
        for b in range(n_batches - n, n_batches):
            sc = pipe.wait(tickets[b])
            offsets, m = sc.matches()
            _assert_same(_csr_to_lists(offsets, m, 2, len(bank_x2)), want[b], f"batch {b}")
            pipe.release(tickets[b])
        # resident rescan: ticket 12 runs in context (12 - 1) % 6 = 5, which last held ticket 6 = batch 5
        t = pipe.submit(None, 0.8)
        assert t == n_batches + 1
        last_b = max(b for b in range(n_batches) if b % n == (t - 1) % n)
        sc = pipe.wait(t)
        offsets, m = sc.matches()
        _assert_same(_csr_to_lists(offsets, m, 2, len(bank_x2)), want[last_b], "resident rescan")
        pipe.release(t)
        with pytest.raises(FocrError):
            pipe.release(t)  # already released
    finally:
        pipe.close()


def test_pipeline_two_threads_and_close_with_batches_outstanding(bank_x2):
    """The executor's API allows one thread to submit and another to retire (wait / read / release); and destroying an executor with
    batches still queued or running must finish them and return (no hang, no crash).  Child-free: a deadlock here would hang the
    suite, so the retiring thread is joined with a time limit."""
    import threading

    from font_ocr_amd.searcher import Pipeline

    n_batches = 40
    pages = [np.stack([synth_page(bank_x2, SYNTH_SEED_BASE + 3000 + 2 * (b % 5) + p, 300, 110) for p in range(2)]) for b in range(5)]
    want = []
    with Scanner(0) as sc:
        sc.set_bank(bank_x2)
        for pg in pages:
            sc.set_pages(pg)
            sc.scan(0.8, 1024, SCAN_MFMA)
            sc.process_hits(0.95, 5)
            want.append((sc.matches()[1].tobytes(), sc.lines_flat().tobytes()))
    pipe = Pipeline(0, 3)
    errs, tickets, cv = [], [], threading.Condition()

    def retire():
        try:
            for b in range(n_batches):
                with cv:
                    cv.wait_for(lambda: len(tickets) > b, timeout=60)
                    t = tickets[b]
                s2 = pipe.wait(t)
                got = (s2.matches()[1].tobytes(), s2.lines_flat().tobytes())
                if got != want[b % 5]:
                    errs.append(f"batch {b} differs")
                pipe.release(t)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    try:
        pipe.set_bank(bank_x2)
        th = threading.Thread(target=retire)
        th.start()
        for b in range(n_batches):  # submit blocks while the batch's context is unreleased: the other thread releases
            t = pipe.submit(pages[b % 5], 0.8)
            with cv:
                tickets.append(t)
                cv.notify_all()
        th.join(timeout=120)
        assert not th.is_alive(), "the retiring thread is stuck"
        assert not errs, errs
        for b in range(len(pipe.scanners)):  # six batches queued / running, nobody waits for them
            pipe.submit(pages[b % 5], 0.8)
    finally:
        pipe.close()  # finishes them first


def test_direct_scans_on_an_executor_context_back_to_back(bank_x2):
    """A context of an executor may also be driven directly (bench.py's isolated-kernel leg does: scan + process_hits several times
    with no wait in between, then reads the last result): every scan's front — the clear, the statistics — must stay behind the
    previous scan's tail in the context's buffers.  (Round 5's front-stream experiment ran them on another stream and faulted
    exactly here.)  Estimated sizes: the first scan is exact, the following ones run on its sizes without a host wait."""
    from font_ocr_amd.searcher import Pipeline

    pages = np.stack([synth_page(bank_x2, SYNTH_SEED_BASE + 3100 + p, 300, 110) for p in range(4)])
    with Scanner(0) as sc:
        sc.set_bank(bank_x2)
        sc.set_pages(pages)
        sc.scan(0.8, 1024, SCAN_MFMA)
        sc.process_hits(0.95, 5)
        want = (sc.matches()[1].tobytes(), sc.lines_flat().tobytes())
    pipe = Pipeline(0, 3)
    try:
        pipe.set_bank(bank_x2)
        t = pipe.submit(pages, 0.8)  # through the executor once: the context's pages are resident
        c_ = pipe.wait(t)
        assert (c_.matches()[1].tobytes(), c_.lines_flat().tobytes()) == want
        pipe.release(t)
        for _ in range(6):  # directly, no wait between the calls
            c_.scan(0.8, 1024, SCAN_MFMA)
            c_.process_hits(0.95, 5)
        assert (c_.matches()[1].tobytes(), c_.lines_flat().tobytes()) == want
        t = pipe.submit(pages, 0.8)  # and through the executor again
        c2 = pipe.wait(t)
        assert (c2.matches()[1].tobytes(), c2.lines_flat().tobytes()) == want
        pipe.release(t)
    finally:
        pipe.close()


def test_pipeline_chars_out_with_estimated_and_redone_batches(bank_x2):
    """chars_out of focr_pipe_submit: the characters' copy into the caller's device buffer is queued behind process_hits on the lane's
    stream (a kernel that reads their number on the device) — also when the batch runs on size estimates, and a batch whose estimates
    prove too small is redone and copied again after the redo.  One lane, one context: every batch meets its predecessor's estimates."""
    from font_ocr_amd.bank import HIT_DTYPE
    from font_ocr_amd.searcher import PinnedPages, Pipeline

    dense = synth_pages(bank_x2, 3, 400, 200, first=8800)
    sparse = dense.copy()
    sparse[:, 50:, :] = 255
    want = {}
    with Scanner(0) as sc:
        sc.set_bank(bank_x2)
        for name, pg in (("dense", dense), ("sparse", sparse)):
            sc.set_pages(pg)
            sc.scan(0.8, 1024, SCAN_DIRECT)
            sc.process_hits(0.95, 5)
            want[name] = sc.lines_flat().tobytes()
    assert len(want["dense"]) > 3 * len(want["sparse"]) > 0
    pipe = Pipeline(0, 1, 1)
    try:
        pipe.set_bank(bank_x2)
        pin = PinnedPages(1, 1, 1 << 20)  # page-locked host memory is device-accessible: a device buffer as far as the executor is concerned
        buf = pin.array.reshape(-1)
        redone0 = pipe.scanners[0].size_estimate_stats()["redone"]
        for name, pg in (("sparse", sparse), ("sparse", sparse), ("dense", dense), ("dense", dense), ("sparse", sparse)):
            buf[:] = 0
            t = pipe.submit(pg, 0.8, chars_out=(buf.ctypes.data, buf.size))
            sc = pipe.wait(t)
            n = sc.total_chars() * HIT_DTYPE.itemsize
            assert n == len(want[name])
            got = buf[:n].tobytes()
            pipe.release(t)
            assert got == want[name], name
        assert pipe.scanners[0].size_estimate_stats()["redone"] >= redone0 + 1  # sparse -> dense: the bound was too small
    finally:
        pipe.close()
        pin.close()


def test_pipeline_prefetch_matches_oracle(bank_x2):
    """Round 4: batches announced ahead with focr_pipe_prefetch (their DMA and ingest run on a copy stream, into the lane's alternate
    page set, while the lane still works on its previous batch) — same lists as the oracle, announcements must be submitted in order,
    a batch that was not announced (or announced with another inversion) is still uploaded by its lane."""
    from font_ocr_amd.searcher import FocrError, PinnedPages, Pipeline

    n_batches, n_lanes = 13, 3
    pins = []
    for b in range(n_batches):
        pin = PinnedPages(2, 96, 288)
        for p in range(2):
            pin.array[p] = synth_page(bank_x2, SYNTH_SEED_BASE + 1300 + 2 * b + p, 288, 96)
        pins.append(pin)
    want = [_oracle_lists(pin.array, bank_x2, 0.8, 1024) for pin in pins]
    pipe = Pipeline(0, n_lanes)
    try:
        pipe.set_bank(bank_x2)
        n = len(pipe.scanners)  # 6 contexts: one announcement each
        for b in range(n):
            pipe.prefetch(pins[b].array)
        with pytest.raises(FocrError):
            pipe.prefetch(pins[n].array)  # every context already holds an announcement
        with pytest.raises(FocrError):
            pipe.submit(pins[1].array, 0.8)  # batch 0 was announced for this ticket
        tickets = []
        for b in range(n_batches):
            if len(tickets) >= n:
                t = tickets[b - n]
                sc = pipe.wait(t)
                offsets, m = sc.matches()
                _assert_same(_csr_to_lists(offsets, m, 2, len(bank_x2)), want[b - n], f"batch {b - n}")
                pipe.release(t)
            if b + 1 == n_batches:
                pipe.announce_last()  # nothing follows the next batch for now: its tail may take the whole chip; same lists
            tickets.append(pipe.submit(pins[b].array, 0.8))
            if b + n < n_batches:
                pipe.prefetch(pins[b + n].array)
        pipe.end_of_stream()  # the late form of the hint (the batch is normally queued already: no effect); same lists
        for b in range(n_batches - n, n_batches):
            sc = pipe.wait(tickets[b])
            offsets, m = sc.matches()
            _assert_same(_csr_to_lists(offsets, m, 2, len(bank_x2)), want[b], f"batch {b}")
            pipe.release(tickets[b])
        t = pipe.submit(pins[0].array, 0.8)  # not announced: the lane uploads it itself
        sc = pipe.wait(t)
        offsets, m = sc.matches()
        _assert_same(_csr_to_lists(offsets, m, 2, len(bank_x2)), want[0], "batch that was not announced")
        pipe.release(t)
        # a larger batch announced to a lane whose staging buffer is too small for it: the buffer grows behind the lane's last ingest
        big = PinnedPages(3, 128, 336)
        pins.append(big)
        for p in range(3):
            big.array[p] = synth_page(bank_x2, SYNTH_SEED_BASE + 1400 + p, 336, 128)
        pipe.prefetch(big.array)
        t = pipe.submit(big.array, 0.8)
        sc = pipe.wait(t)
        offsets, m = sc.matches()
        _assert_same(_csr_to_lists(offsets, m, 3, len(bank_x2)), _oracle_lists(big.array, bank_x2, 0.8, 1024), "larger announced batch")
        pipe.release(t)
        pipe.prefetch(pins[2].array, invert=False)  # announced with the other inversion: the announcement is void, the lane uploads the batch itself
        t = pipe.submit(pins[2].array, 0.8)
        sc = pipe.wait(t)
        offsets, m = sc.matches()
        _assert_same(_csr_to_lists(offsets, m, 2, len(bank_x2)), want[2], "batch announced with another inversion")
        pipe.release(t)
        for b in (3, 4, 5, 6, 7, 8, 9, 10):  # the page sets change places batch after batch in every context: announced batches only
            pipe.prefetch(pins[b].array)
            t = pipe.submit(pins[b].array, 0.8)
            sc = pipe.wait(t)
            offsets, m = sc.matches()
            _assert_same(_csr_to_lists(offsets, m, 2, len(bank_x2)), want[b], f"announced batch {b} after the void one")
            pipe.release(t)
        t = pipe.submit(None, 0.8)  # rescan of the lane's resident pages: they are the set that was swapped in (batch 4's lane: ticket order)
        sc = pipe.wait(t)
        pipe.release(t)
        pipe.prefetch(pins[1].array)  # an announcement that is never submitted: destroying the pipe must cope
    finally:
        pipe.close()
        for pin in pins:
            pin.close()


@pytest.mark.parametrize("mode", MODES)
def test_process_hits_ties_and_group_anchoring(scanner, bank_x2, mode):
    """process_hits corner semantics (src/ncc.rs:755-764, 1042-1048): exact similarity ties (duplicated templates score
    the same on the same window -> `max_by` keeps the LAST maximum, i.e. the later template), groups anchored on their
    first element (a chain of hits each within `overlap` of its neighbour but not of the group's first splits), and
    an anchor threshold that only some lines reach."""
    base = [t for t in range(len(bank_x2)) if int(bank_x2.templates[t]["shift_x"]) == 0][1:40]  # shift-0 glyphs, no space
    idx = base + base[:10] + base[:5]  # duplicates: templates len(base)+k == k, and a third copy of the first five
    bank = bank_x2.subset(idx)
    letters = bank.templates["letter"].copy()
    letters[len(base):] += 1000  # make the copies distinguishable in the output
    bank.templates["letter"] = letters
    n_w, n_h = int(bank.templates[0]["n_w"]), int(bank.templates[0]["n_h"])
    rng = np.random.default_rng(11)
    ink = np.zeros((3, 130, 420), np.uint32)
    for p in range(3):
        for li in range(4):
            x = 10
            while x + n_w < 400:
                t = int(rng.integers(0, len(base)))
                y = 8 + 30 * li
                nd = bank.needle(t).astype(np.uint32)
                if li == 3:
                    nd = nd // 2 + (rng.integers(0, 40, nd.shape)).astype(np.uint32)  # faint, noisy line: below the anchor
                ink[p, y:y + n_h, x:x + n_w] += nd
                x += int(rng.choice([3, 4, 6, 8, 9]))  # dense chains of overlapping stamps -> long partition_by chains
    pages = (255 - np.minimum(ink, 255)).astype(np.uint8)
    scanner.set_bank(bank)
    scanner.set_pages(pages)
    scanner.scan(0.5, 1024, mode)
    offsets, m = scanner.matches()
    got = _csr_to_lists(offsets, m, 3, len(bank))
    n_ties = 0
    for anchor, overlap in [(0.95, 5), (0.8, 3), (0.999, 9), (0.6, 0)]:
        scanner.process_hits(anchor, overlap)
        lines = scanner.lines()
        for p in range(3):
            counts = np.array([len(x) for x in got[p]], np.uint32)
            mm = np.zeros((len(bank), 1024), O.MATCH_DTYPE)
            for t, x in enumerate(got[p]):
                mm[t, : len(x)] = x
            want = O.process_hits(O.raw_hits(counts, mm, bank), anchor, overlap)
            assert len(lines[p]) == len(want), (p, anchor, overlap)
            for lg, lw in zip(lines[p], want):
                assert np.array_equal(lg["x"], lw["x"]) and np.array_equal(lg["y"], lw["y"])
                assert np.array_equal(lg["letter"], lw["letter"])
                assert lg["similarity"].tobytes() == lw["similarity"].tobytes()
                n_ties += int((lg["letter"] >= 1000).sum())
    assert n_ties > 20  # the later copy won its ties


@pytest.mark.parametrize("thr", [0.8, 0.3, -0.5])
def test_rust_scan_mode_vs_oracle(scanner, bank_x2, thr):
    """FOCR_SCAN_RUST (`ncc --rust`): the scalar scan's arithmetic (division instead of fma,
    sqrt of the norm product), its skips (s_p == 0, num < 0, s_n == 0) and its missing cap, against the oracle's
    restatement of src/ncc.rs:406-483 — positions exact, similarities bit-identical."""
    from font_ocr_amd.searcher import SCAN_RUST

    pages = np.stack([synth_page(bank_x2, SYNTH_SEED_BASE + 1300 + p, 250, 90) for p in range(2)])
    pages[1, 40:, 120:] = np.random.default_rng(3).integers(0, 256, (50, 130), dtype=np.uint8)  # noise: many low-sim hits
    scanner.set_bank(bank_x2)
    scanner.set_pages(pages)
    cap = 0xFFFFFFFF
    scanner.scan(thr, cap, SCAN_RUST)
    counts = scanner.counts()
    offsets, m = scanner.matches()
    T = len(bank_x2)
    big = 1 << 15
    for p in range(2):
        wc, wm = O.scan_page_rust(O.invert(pages[p]), bank_x2, thr, cap=big)
        assert wc.max() < big
        assert np.array_equal(counts[p], wc), (p, thr)
        for t in range(T):
            got = m[offsets[p * T + t]: offsets[p * T + t + 1]]
            want = wm[t, : wc[t]]
            assert got.tobytes() == want.tobytes(), (p, t, thr)
    if thr < 0:
        assert counts.max() > 1024  # more than the AVX2 path's cap: the Rust path has none
    # and it differs from the C path where it should: negative numerators are skipped
    if thr < 0:
        scanner.scan(thr, cap, SCAN_DIRECT)
        assert scanner.counts().sum() > counts.sum()


@pytest.mark.parametrize("mode", MODES[1:])
def test_size_estimates_and_redo_on_overflow(bank_x2, mode):
    """Repeat scans of one setup run on the previous scan's counts + a margin (4 .. 20 %) without host waits (focr_ctx_set_size_estimates);
    a batch with far more hits than the previous one must overflow those bounds, be redone with exact sizes, and still give
    the reference lists and process_hits output — also when process_hits was already queued behind the scan."""
    bank = bank_x2.subset(list(range(33, 80)) + list(range(95 + 33, 95 + 80)))
    dense = np.stack([synth_page(bank_x2, SYNTH_SEED_BASE + 900 + p, 300, 130) for p in range(3)])
    sparse = np.full_like(dense, 255)
    sparse[:, 20:40, 30:120] = dense[:, 20:40, 30:120]  # a few glyphs only
    want = {"sparse": _oracle_lists(sparse, bank, 0.7, 1024), "dense": _oracle_lists(dense, bank, 0.7, 1024)}
    n_sparse, n_dense = (sum(len(x) for p in want[k] for x in p) for k in ("sparse", "dense"))
    assert n_dense > 5 * n_sparse + 100_000 or n_dense > 20 * n_sparse
    with Scanner(0) as sc:
        sc.set_bank(bank)
        sc.set_pages(sparse)

        def check(which):
            offsets, m = sc.matches()
            _assert_same(_csr_to_lists(offsets, m, 3, len(bank)), want[which], which)

        sc.scan(0.7, 1024, mode)       # first scan of the setup: exact sizes
        check("sparse")
        sc.scan(0.7, 1024, mode)       # second: estimated sizes, nothing waits until the getter
        sc.process_hits(0.9, 5)
        lines_a = sc.lines_flat().copy()
        check("sparse")
        sc.upload_pages(dense, 0)      # same geometry, ~10x the hits: the estimates are far too small
        sc.scan(0.7, 1024, mode)
        sc.process_hits(0.9, 5)        # queued behind the scan that will have to be redone
        lines_b = sc.lines_flat().copy()
        check("dense")
        sc.scan(0.7, 1024, mode)       # estimates now come from the dense batch
        check("dense")
        sc.process_hits(0.9, 5)
        assert sc.lines_flat().tobytes() == lines_b.tobytes() and len(lines_b) > len(lines_a)
        sc.set_size_estimates(False)   # round-1 behaviour: counts read between the phases
        sc.scan(0.7, 1024, mode)
        check("dense")
        sc.process_hits(0.9, 5)
        assert sc.lines_flat().tobytes() == lines_b.tobytes()
        sc.upload_pages(sparse, 0)
        sc.set_size_estimates(True)
        sc.scan(0.7, 1024, mode)       # bounds from the dense batch: plenty
        sc.process_hits(0.9, 5)
        check("sparse")
        assert sc.lines_flat().tobytes() == lines_a.tobytes()


def test_unexpected_large_bucket_under_estimated_sizes_redoes_the_batch():
    """Round 4: the second sort launch (buckets above 1 024 hits) exists only where such a bucket is expected.  Two batches of one
    setup with the SAME number of hits within the estimates' margin — the first spreads them over every page row (largest bucket
    ~ 500), the second packs them into a few rows (> 1 024 each): no count exceeds its bound, the bucket does, the batch is redone
    with exact sizes and the lists equal the reference's."""
    rng = np.random.default_rng(77)
    bank = _random_bank(rng, [(8, 15)], 40)
    r_w, r_h = 180, 96
    noise = rng.integers(0, 256, (2, r_h, r_w), dtype=np.uint8)
    spread = np.full((2, r_h, r_w), 255, np.uint8)
    spread[:, :, 20:64] = noise[:, :, 20:64]          # a noise strip through every row: few windows per row
    packed = np.full((2, r_h, r_w), 255, np.uint8)
    packed[:, 30:38, :] = noise[:, 30:38, :]           # full-width noise in a band of rows: many windows in few rows
    thr = 0.02                                         # noise against noise: about four windows in ten pass
    cap = 4096  # never reached: ~1 400 hits per (page, template)
    want = {"spread": _oracle_lists(spread, bank, thr, cap), "packed": _oracle_lists(packed, bank, thr, cap)}
    n = {k: sum(len(x) for p in v for x in p) for k, v in want.items()}
    rows = {k: max(int(np.bincount(np.concatenate([x["y"] for x in p if len(x)]).astype(np.int64)).max()) for p in v) for k, v in want.items()}
    assert rows["spread"] < 800 and rows["packed"] > 1100, rows          # largest bucket = hits of one page row
    assert n["packed"] < 1.15 * n["spread"] + 8000, n                    # inside the estimates' margin: only the bucket overflows
    with Scanner(0) as sc:
        sc.set_bank(bank)
        sc.set_pages(spread)
        sc.scan(thr, cap, MFMA1)   # exact sizes
        sc.scan(thr, cap, MFMA1)   # estimated
        offsets, m = sc.matches()
        _assert_same(_csr_to_lists(offsets, m, 2, len(bank)), want["spread"], "spread")
        redone0 = sc.size_estimate_stats()["redone"]
        sc.upload_pages(packed, 0)
        sc.scan(thr, cap, MFMA1)   # estimated from the spread batch: no second sort launch planned
        offsets, m = sc.matches()
        _assert_same(_csr_to_lists(offsets, m, 2, len(bank)), want["packed"], "packed")
        assert sc.size_estimate_stats()["redone"] == redone0 + 1
        sc.scan(thr, cap, MFMA1)   # now the large bucket is expected: estimated sizes, second launch, nothing redone
        offsets, m = sc.matches()
        _assert_same(_csr_to_lists(offsets, m, 2, len(bank)), want["packed"], "packed again")
        assert sc.size_estimate_stats()["redone"] == redone0 + 1


def test_compat_symbols_in_the_reference_call_pattern(bank_x2):
    """The unmodified reference host calls ncc_8_u8 / ncc_16_u8 once per template with the same page and, per size class,
    the same window tables (src/ncc.rs:332-404, 587-701).  All 380 calls of one 608x720 page through the drop-in symbols:
    identical lists, and — page and tables stay resident on the device between calls — faster than the reference's own
    kernel on one host core."""
    import time

    page = synth_page(bank_x2, SYNTH_SEED_BASE + 77, 608, 720)
    inv = O.invert(page)
    flat = O.padded(inv)
    tabs = O.tables(inv)
    stats = {}
    for t in range(len(bank_x2)):
        nd = bank_x2.needle(t)
        if nd.shape not in stats:
            stats[nd.shape] = O.prepare_for_size(inv, nd.shape[1], nd.shape[0], tabs)
    searcher = Searcher(inv)
    searcher.search_c_u8(bank_x2.needle(1), stats[bank_x2.needle(1).shape], 0.8)  # context creation, first upload
    t0 = time.perf_counter()
    got = [searcher.search_c_u8(bank_x2.needle(t), stats[bank_x2.needle(t).shape], 0.8) for t in range(len(bank_x2))]
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    want = [O.ncc_u8(flat, 608, 720, bank_x2.needle(t), stats[bank_x2.needle(t).shape], 0.8, 1024, use_ref=O.have_ref()) for t in range(len(bank_x2))]
    t_ref = time.perf_counter() - t0
    assert sum(len(w) for w in want) > 15000
    for t, (g, w) in enumerate(zip(got, want)):
        assert g.tobytes() == w.tobytes(), t
    if O.have_ref():
        assert t_gpu < t_ref, (t_gpu, t_ref)
    print(f"380 drop-in calls: {t_gpu * 1e3:.0f} ms on the device path, {t_ref * 1e3:.0f} ms with the reference kernel on one core")
    # ... and as the reference's host calls them: from a pool of threads, one page each (rayon par_iter, src/ncc.rs:839-847); every
    # calling thread has its own stream and resident inputs
    import threading

    n_thr = 8
    searchers = [Searcher(inv) for _ in range(n_thr)]
    outs = [None] * n_thr

    def work(j):
        outs[j] = [searchers[j].search_c_u8(bank_x2.needle(t), stats[bank_x2.needle(t).shape], 0.8) for t in range(len(bank_x2))]

    for rep_ in range(2):  # the first round creates the threads' contexts
        ths = [threading.Thread(target=work, args=(j,)) for j in range(n_thr)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        t_pool = time.perf_counter() - t0
    for j in range(n_thr):
        for t, (g, w) in enumerate(zip(outs[j], want)):
            assert g.tobytes() == w.tobytes(), (j, t)
    print(f"{n_thr} calling threads, one page each: {t_pool * 1e3:.0f} ms = {n_thr * 608 * 720 / t_pool / 1e6:.1f} Mpx/s through the drop-in symbols "
          f"(one thread: {608 * 720 / t_gpu / 1e6:.1f} Mpx/s)")


def test_pipeline_ticket_gate_never_blocks_on_failed_or_direct_batches():
    """tools/gate_check.py: the executor queues its batches in ticket order; a batch that fails before its scan and a
    direct-mode batch (no turn in the scan chain) between MFMA batches must not hold the later tickets up.  Child process with a
    time limit: a deadlock must fail, not hang."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gate_check.py")], capture_output=True, text=True, timeout=90)
    assert r.returncode == 0 and "gate ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_pipeline_soak_mfma_equals_direct():
    """tools/stress_pipeline.py: 60 different batches (sizes, geometries, thresholds, blank pages) through the three-lane
    pipeline — MFMA prefilter with item queues, estimated result sizes, scans taking turns — each compared with the direct
    scan of the same pages on another context: matches, counts and lines must be identical."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_pipeline.py"), "--batches", "60", "--pages", "12"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_fleet_orders_batches_over_devices(bank_x2):
    """focr_fleet_* (the multi-device executor of the C ABI): batch k goes to device k % n_devices; retired in submission order
    the results are those of a single context, whatever the device count.  The one GPU of the test box is listed twice
    (two executors, eight contexts on it), which exercises the ticket arithmetic of a two-device fleet."""
    from font_ocr_amd.searcher import Fleet

    bank = bank_x2.subset(list(range(33, 70)) + list(range(95 + 33, 95 + 70)))
    batches = [synth_pages(bank_x2, 1 + (k % 3), 300, 130, first=9000 + 10 * k) for k in range(11)]
    want = []
    with Scanner(0) as sc:
        sc.set_bank(bank)
        for pg in batches:
            sc.set_pages(pg)
            sc.scan(0.8, 1024, SCAN_MFMA)
            sc.process_hits(0.95, 5)
            want.append((sc.counts().copy(), sc.matches()[1].tobytes(), sc.lines_flat().tobytes()))
    assert sum(len(w[2]) for w in want) > 1000
    for devices in ([0], [0, 0]):
        fl = Fleet(devices, lanes=2)
        try:
            assert fl.n_devices == len(devices) and fl.lanes == 2 and fl.slots == 4 * len(devices)
            fl.set_bank(bank)
            inflight, got = [], []

            def retire():
                t = inflight.pop(0)
                assert fl.device_of(t) == 0
                v = fl.wait(t)
                got.append((v.counts().copy(), v.matches()[1].tobytes(), v.lines_flat().tobytes()))
                fl.release(t)

            for k, pg in enumerate(batches):
                if len(inflight) == fl.slots:
                    if k == fl.slots:  # every context holds an unreleased batch: refused, not a silent self-deadlock
                        from font_ocr_amd.searcher import FocrError

                        with pytest.raises(FocrError, match="release the oldest"):
                            fl.submit(pg, 0.8, 1024, SCAN_MFMA, True, 0.95, 5)
                    retire()
                t = fl.submit(pg, 0.8, 1024, SCAN_MFMA, True, 0.95, 5)
                assert t == k + 1
                inflight.append(t)
            while inflight:
                retire()
            for k in range(len(batches)):
                assert np.array_equal(got[k][0], want[k][0]) and got[k][1] == want[k][1] and got[k][2] == want[k][2], (devices, k)
        finally:
            fl.close()
