#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the build container only).

The reference ships no fixtures, so every expected output here is produced by the reference's
own kernels, compiled unmodified from /root/reference/src/ncc.cpp into oracle/_ref/libncc_ref.so
(oracle/Makefile) and driven through ctypes.  Inputs and expected outputs are DATA; no reference
source text is stored.  Fonts: DejaVu Sans Mono (Courier New is not available offline).

  bank_dejavu13_ascii95_x2.bin     95 glyphs x 4 x-shifts (BASELINE configs[1])
  bank_dejavu13_ascii95_x2y2.bin   95 glyphs x 16 shifts   (configs[2])
  bank_dejavu13_default_x0.bin     74-char default alphabet (configs[0])
  kernel_cases.npz                 per-call vectors for ncc_8_u8 / ncc_16_u8
  c1_page.npz                      one 608x720 page, 74 templates: raw lists + post-processed lines
  c2_page0.npz                     page 0 of configs[1]: counts + lists for the 380-template bank

Usage: python oracle/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from font_ocr_amd import ASCII95, DEFAULT_ALPHABET, Bank, synth_page  # noqa: E402
from font_ocr_amd.bank import SYNTH_SEED_BASE  # noqa: E402
from oracle import oracle as O  # noqa: E402

FONT = "/usr/share/fonts/truetype/dejavu/DejaVuSansMono.ttf"
GOLD = os.path.join(ROOT, "tests", "golden")


def kernel_case(name, page_inv, needle, thr, cap=1024):
    """Run the REFERENCE kernel on (page, needle) with stats from the restated prepare_for_size."""
    r_h, r_w = page_inv.shape
    n_h, n_w = needle.shape
    stats = O.prepare_for_size(page_inv, n_w, n_h)
    out = O.ncc_u8(O.padded(page_inv), r_w, r_h, needle, stats, thr, cap, use_ref=True)
    return {
        f"{name}/page": page_inv,
        f"{name}/needle": np.ascontiguousarray(needle, np.uint8),
        f"{name}/thr": np.float32(thr),
        f"{name}/cap": np.int64(cap),
        f"{name}/patch_sum": stats[0],
        f"{name}/patch_rnorm": stats[1],
        f"{name}/start_end": stats[2],
        f"{name}/expect": out,
    }


def main():
    assert O.have_ref(), "oracle/_ref/libncc_ref.so missing: run make -C oracle in the build container"
    os.makedirs(GOLD, exist_ok=True)
    b_x2 = Bank.rasterize(FONT, 13, 2, 0, alphabet=ASCII95)
    b_x2y2 = Bank.rasterize(FONT, 13, 2, 2, alphabet=ASCII95)
    b_def = Bank.rasterize(FONT, 13, 0, 0, alphabet=DEFAULT_ALPHABET)
    b_x2.save(os.path.join(GOLD, "bank_dejavu13_ascii95_x2.bin"))
    b_x2y2.save(os.path.join(GOLD, "bank_dejavu13_ascii95_x2y2.bin"))
    b_def.save(os.path.join(GOLD, "bank_dejavu13_default_x0.bin"))

    rng = np.random.default_rng(20261003)
    cases = {}
    A = ASCII95.index

    # (a) text crop, 8-wide and 9-wide templates, default threshold
    crop_luma, crop_truth = synth_page(b_x2, SYNTH_SEED_BASE + 7, 200, 96, with_truth=True)
    crop = O.invert(crop_luma)
    # templates that were actually stamped on the crop (one 8-wide = shift 0, two 9-wide)
    t8 = [int(t) for t in crop_truth["template_index"] if t < 95]
    t9 = [int(t) for t in crop_truth["template_index"] if t >= 95]
    cases.update(kernel_case("text_w8", crop, b_x2.needle(t8[0]), 0.8))
    cases.update(kernel_case("text_w9", crop, b_x2.needle(t9[0]), 0.8))
    cases.update(kernel_case("text_w9_low_thr", crop, b_x2.needle(t9[-1]), 0.5))
    # (b) threshold -1: every finite window passes -> the 1024 cap and (y,x) order
    cases.update(kernel_case("cap_w8", crop, b_x2.needle(A("o")), -1.0))
    cases.update(kernel_case("cap_w9", crop, b_x2.needle(95 + A("o")), -1.0))
    cases.update(kernel_case("cap_small_17", crop, b_x2.needle(A("x")), 0.3, cap=17))
    # (c) degenerate pages
    blank = np.zeros((48, 64), np.uint8)
    cases.update(kernel_case("blank", blank, b_x2.needle(A("A")), 0.8))
    solid = np.full((48, 64), 255, np.uint8)
    cases.update(kernel_case("solid", solid, b_x2.needle(A("A")), -1.0))
    dot = np.full((48, 64), 200, np.uint8)
    dot[20, 30] = 201
    cases.update(kernel_case("solid_one_pixel", dot, b_x2.needle(95 + A("A")), -1.0))
    # space glyph (all-zero needle): rnorm_n = inf -> never emits
    cases.update(kernel_case("space_needle", crop, b_x2.needle(A(" ")), -1.0))
    # (d) noise pages with odd template shapes (n_h < 4 takes the single-row loops)
    noise = rng.integers(0, 256, (56, 72), dtype=np.uint8)
    for (w, h) in [(16, 16), (12, 5), (3, 3), (8, 8), (1, 2), (16, 20), (5, 1), (9, 15)]:
        nd = rng.integers(0, 256, (h, w), dtype=np.uint8)
        cases.update(kernel_case(f"noise_{w}x{h}", noise, nd, 0.05))
    # self-match: needle cut out of the noise page must score 1.0 at its origin
    cases.update(kernel_case("noise_self_9x15", noise, noise[11:26, 23:32].copy(), 0.9))
    cases.update(kernel_case("noise_self_8x13", noise, noise[1:14, 1:9].copy(), 0.9))
    # (e) glyphs touching the edges: origin (0,0) is never searched, (1,1) and the far corner are
    gl = b_x2.needle(A("W"))
    edge = np.zeros((40, 50), np.uint8)
    edge[0:15, 0:8] = gl
    edge[40 - 15:, 50 - 8:] = gl
    cases.update(kernel_case("edge_origin_and_far", edge, gl, 0.8))
    edge2 = np.zeros((40, 50), np.uint8)
    edge2[1:16, 1:9] = gl
    edge2[1:16, 50 - 8:] = gl
    edge2[40 - 15:, 1:9] = gl
    cases.update(kernel_case("edge_one_one", edge2, gl, 0.8))
    # saturated block + text
    sat = crop.copy()
    sat[30:60, 60:120] = 255
    cases.update(kernel_case("saturated_block", sat, b_x2.needle(95 + A("H")), 0.6))
    np.savez_compressed(os.path.join(GOLD, "kernel_cases.npz"), **cases)
    names = sorted({k.split("/")[0] for k in cases})
    print(f"kernel_cases.npz: {len(names)} cases")
    for n in names:
        print(f"   {n}: {len(cases[n + '/expect'])} matches")

    # C1: reference's own CPU-runnable case (one 608x720 page, default alphabet, x-bits 0)
    page, truth = synth_page(b_def, SYNTH_SEED_BASE, 608, 720, with_truth=True)
    inv = O.invert(page)
    counts, matches = O.scan_page(inv, b_def, 0.8, use_ref=True)
    hits = O.raw_hits(counts, matches, b_def)
    lines = O.process_hits(hits, 0.95, 5)
    flat = np.concatenate(lines) if lines else np.zeros(0, O.HIT_DTYPE)
    line_ends = np.cumsum([len(l) for l in lines]).astype(np.int64)
    np.savez_compressed(
        os.path.join(GOLD, "c1_page.npz"), page=page, counts=counts,
        matches=np.concatenate([matches[t, : counts[t]] for t in range(len(counts))]),
        lines=flat, line_ends=line_ends, truth=truth,
    )
    print(f"c1_page.npz: {int(counts.sum())} raw hits, {len(lines)} lines, {len(flat)} chars, {len(truth)} stamped")

    # C2 page 0 (380 templates)
    page = synth_page(b_x2, SYNTH_SEED_BASE, 608, 720)
    counts, matches = O.scan_page(O.invert(page), b_x2, 0.8, use_ref=True)
    np.savez_compressed(
        os.path.join(GOLD, "c2_page0.npz"), counts=counts,
        matches=np.concatenate([matches[t, : counts[t]] for t in range(len(counts))]),
        page_crc=np.uint32(__import__("zlib").crc32(page.tobytes())),
    )
    print(f"c2_page0.npz: {int(counts.sum())} raw hits")


if __name__ == "__main__":
    main()
