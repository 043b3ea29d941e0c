"""End-to-end `ncc` CLI on the GPU: default text output, --csv, --raw against the oracle pipeline."""
import os
import subprocess

import numpy as np
import pytest

from font_ocr_amd import ASCII95, Bank, save_pgm, synth_page
from font_ocr_amd.bank import SYNTH_SEED_BASE, format_f32
from oracle import oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCC = os.path.join(ROOT, "font_ocr_amd", "bin", "ncc")
FONT = "/usr/share/fonts/truetype/dejavu/DejaVuSansMono.ttf"


@pytest.mark.skipif(not os.path.exists(FONT), reason="DejaVu Sans Mono not installed")
def test_cli_text_csv_raw(tmp_path):
    if not os.path.exists(NCC):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "font_ocr_amd", "csrc"), "cli"], check=True)
    alphabet = ASCII95[1:60]  # no space
    bank = Bank.rasterize(FONT, 13, 1, 0, alphabet=alphabet)
    pages = [synth_page(bank, SYNTH_SEED_BASE + 400 + p, 300 + 20 * (p % 2), 130) for p in range(3)]  # two page sizes
    paths = []
    for p, pg in enumerate(pages):
        paths.append(str(tmp_path / f"p{p}.pgm"))
        save_pgm(paths[-1], pg)
    common = [NCC, "-f", FONT, "-t", "13", "--x-bits", "1", "-a", alphabet]
    want_lines = []
    want_raw0 = None
    for p, pg in enumerate(pages):
        counts, matches = O.scan_page(O.invert(pg), bank, 0.8, use_ref=O.have_ref())
        hits = O.raw_hits(counts, matches, bank)
        if p == 0:
            want_raw0 = (counts, matches)
        want_lines.append(O.process_hits(hits, 0.95, 5))

    r = subprocess.run(common + ["-i"] + paths, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want_text = "".join("".join(chr(int(c)) for c in l["letter"]) + "\n" for lines in want_lines for l in lines)
    assert r.stdout == want_text
    assert len(want_text) > 50

    r = subprocess.run(common + ["--csv", "--rust", "-i"] + paths, capture_output=True, text=True)  # --rust = direct device path
    assert r.returncode == 0, r.stderr
    want_csv = []
    for p, lines in enumerate(want_lines):
        for l in lines:
            for c in l:
                cx, cy = np.float32(c["x"]) + np.float32(c["w"]) * np.float32(0.5), np.float32(c["y"]) + np.float32(c["h"]) * np.float32(0.5)
                want_csv.append(f"{p},{int(c['letter'])},{format_f32(cx)},{format_f32(cy)},{int(c['x'])},{int(c['y'])},{int(c['w'])},{int(c['h'])}")
    assert r.stdout.splitlines() == want_csv

    r = subprocess.run(common + ["--raw", "-i", paths[0]], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    counts, matches = want_raw0
    rows = r.stdout.splitlines()
    assert len(rows) == int(counts.sum())
    k = 0
    for t in range(len(counts)):
        d = bank.templates[t]
        for m in matches[t, : counts[t]]:
            f = rows[k].split(",")
            assert [int(f[0]), int(f[3]), int(f[4]), int(f[5]), int(f[6])] == [int(d["letter"]), int(m["x"]), int(m["y"]), int(d["n_w"]), int(d["n_h"])]
            assert f[1] == format_f32(np.float32(m["x"]) + np.float32(d["n_w"]) * np.float32(0.5))
            assert f[9] == format_f32(d["off_x"]) and f[10] == format_f32(d["off_y"])
            k += 1
    r = subprocess.run(common + ["--raw", "-i"] + paths, capture_output=True, text=True)
    assert r.returncode == 101  # assert!(args.img.len() == 1), src/ncc.rs:834


@pytest.mark.skipif(not os.path.exists(FONT), reason="DejaVu Sans Mono not installed")
def test_cli_batched_pipeline_is_order_preserving(tmp_path):
    """The decode/scan pipeline (batches, mixed page sizes, decode-ahead back-pressure) prints the same bytes
    whatever the batch size; a missing image panics (exit 101) like image::open(..).unwrap(), src/ncc.rs:575."""
    alphabet = ASCII95[1:40]
    bank = Bank.rasterize(FONT, 13, 0, 0, alphabet=alphabet)
    paths = []
    for p in range(9):
        pg = synth_page(bank, SYNTH_SEED_BASE + 900 + p, 280 + 24 * (p % 3), 120 + 15 * (p % 2))
        paths.append(str(tmp_path / f"q{p}.pgm"))
        save_pgm(paths[-1], pg)
    cmd = [NCC, "-f", FONT, "-t", "13", "-a", alphabet, "--csv", "-i"] + paths
    outs = []
    for batch, contexts, devices in (("256", "1", "1"), ("1", "1", "1"), ("2", "1", "99"), ("4", "1", "1"), ("2", "2", "99"), ("1", "3", "1"), ("3", "3", "99")):
        # FOCR_CLI_DEVICES caps the GPUs the batches are dealt over (99 = every visible one): same bytes on stdout
        r = subprocess.run(cmd, capture_output=True, text=True,
                           env=dict(os.environ, FOCR_CLI_BATCH=batch, FOCR_CLI_CONTEXTS=contexts, FOCR_CLI_DEVICES=devices))
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout)
    assert len(outs[0].splitlines()) > 200
    assert all(o == outs[0] for o in outs[1:])
    pages_seen = [int(l.split(",")[0]) for l in outs[0].splitlines()]
    assert pages_seen == sorted(pages_seen) and set(pages_seen) == set(range(9))
    r = subprocess.run(cmd + [str(tmp_path / "missing.pgm")], capture_output=True, text=True, env=dict(os.environ, FOCR_CLI_BATCH="2"))
    assert r.returncode == 101 and "cannot open image" in r.stderr
    # -v: the reference's per-template diagnostics on stderr (src/ncc.rs:657-666, 703-718); stdout is unchanged
    r = subprocess.run(cmd[:-len(paths)] + paths[:2] + ["-v"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == "".join(l + "\n" for l in outs[0].splitlines() if int(l.split(",")[0]) < 2)
    assert r.stderr.count("needle size") == 2 * len(alphabet) and "hits: " in r.stderr and "overall " in r.stderr
    # ... preceded by the reference's font-metrics preamble (src/ncc.rs:791-802)
    for line in ("metrics Metrics { units_per_em: 2048, ascent: 1901.0, descent: -483.0,", "ascent  ", "descent -", "font_bbox size <", "line_space 2384 "):
        assert line in r.stderr, (line, r.stderr[:600])


@pytest.mark.skipif(not os.path.exists(FONT), reason="DejaVu Sans Mono not installed")
def test_cli_spaces_extension(tmp_path):
    """--spaces (extension): blanks for gaps of whole advances; off by default as in the reference."""
    alphabet = ASCII95[33:59]  # A..Z
    bank = Bank.rasterize(FONT, 13, 0, 0, alphabet=alphabet)
    n_w, n_h = int(bank.templates[0]["n_w"]), int(bank.templates[0]["n_h"])
    rng = np.random.default_rng(5)
    ink = np.zeros((120, 330), np.uint32)
    want = []
    for li in range(3):
        cells = sorted(rng.choice(30, size=18, replace=False).tolist())
        text, prev = "", None
        for c in cells:
            t = int(rng.integers(0, len(alphabet)))
            x, y = 40 + 8 * c, 20 + 30 * li
            ink[y:y + n_h, x:x + n_w] += bank.needle(t)
            text += ("" if prev is None else " " * (c - prev - 1)) + alphabet[t]
            prev = c
        want.append(text)
    page = (255 - np.minimum(ink, 255)).astype(np.uint8)
    path = str(tmp_path / "gaps.pgm")
    save_pgm(path, page)
    cmd = [NCC, "-f", FONT, "-t", "13", "-a", alphabet, "-i", path]
    r = subprocess.run(cmd + ["--spaces"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == want
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == [w.replace(" ", "") for w in want]


@pytest.mark.skipif(not os.path.exists(FONT), reason="DejaVu Sans Mono not installed")
def test_cli_wide_templates_extension(tmp_path):
    """-t 30 gives 19 px wide templates: the reference panics ("not handled", src/ncc.rs:392) and so does the default
    CLI; --allow-wide (extension) scans them with the exact kernel and reads the stamped text back."""
    alphabet = ASCII95[33:59]  # A..Z
    bank = Bank.rasterize(FONT, 30, 1, 0, alphabet=alphabet)
    assert int(bank.templates["n_w"].max()) > 16
    page, truth = synth_page(bank, SYNTH_SEED_BASE + 950, 700, 220, with_truth=True)
    path = str(tmp_path / "big.pgm")
    save_pgm(path, page)
    cmd = [NCC, "-f", FONT, "-t", "30", "--x-bits", "1", "-a", alphabet, "-i", path]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 101 and "not handled" in r.stderr
    r = subprocess.run(cmd + ["--allow-wide"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    counts, matches = O.scan_page(O.invert(page), bank, 0.8)
    want = O.process_hits(O.raw_hits(counts, matches, bank), 0.95, 5)
    assert r.stdout == "".join("".join(chr(int(c)) for c in l["letter"]) + "\n" for l in want)
    lines = {}
    for t in truth:
        lines.setdefault(int(t["y"]), []).append((int(t["x"]), chr(int(t["letter"]))))
    stamped = ["".join(ch for _, ch in sorted(v)) for _, v in sorted(lines.items())]
    assert r.stdout.splitlines() == stamped and len(stamped) >= 3
