"""Host-side pieces around the scan: the `ncc` CLI surface (src/ncc.rs:486-542), Rust-compatible float
printing, image decode.  The GPU end-to-end CLI test is in test_gpu_cli.py."""
import os
import subprocess
import zlib

import numpy as np
import pytest

from font_ocr_amd import load_image, save_pgm
from font_ocr_amd.bank import format_f32

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCC = os.path.join(ROOT, "font_ocr_amd", "bin", "ncc")


@pytest.fixture(scope="module")
def ncc_bin():
    if not os.path.exists(NCC):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "font_ocr_amd", "csrc"), "cli"], check=True)
    return NCC


def test_format_f32_matches_rust_display():
    # values as Rust's `{}` prints f32 (shortest round-trip, no exponent, no trailing ".0")
    for v, want in [(12.5, "12.5"), (12.0, "12"), (0.1, "0.1"), (304.0, "304"), (7.5, "7.5"), (0.25, "0.25"),
                    (1e-7, "0.0000001"), (16777216.0, "16777216"), (-3.5, "-3.5"), (0.0, "0"), (1.0 / 3.0, "0.33333334"),
                    (0.8, "0.8"), (7.8266602, "7.82666")]:
        assert format_f32(np.float32(v)) == want, (v, format_f32(np.float32(v)))


def test_cli_flags_and_errors(ncc_bin):
    r = subprocess.run([ncc_bin, "--help"], capture_output=True, text=True)
    assert r.returncode == 0
    for flag in ("--img", "--font", "--text-size", "--x-bits", "--y-bits", "--hinting", "--threshold", "--anchor-threshold",
                 "--overlap", "--alphabet", "--box-size", "--x-padding", "--y-padding", "--save-letters", "--rust", "--verbose",
                 "--csv", "--raw"):
        assert flag in r.stdout, flag
    assert "[default: 0.8]" in r.stdout and "[default: 0.95]" in r.stdout and "[default: 5]" in r.stdout
    r = subprocess.run([ncc_bin, "-t", "13"], capture_output=True, text=True)
    assert r.returncode == 2 and "--font <FONT>" in r.stderr  # clap: missing required argument
    r = subprocess.run([ncc_bin, "-f", "x.ttf"], capture_output=True, text=True)
    assert r.returncode == 2 and "--text-size" in r.stderr
    r = subprocess.run([ncc_bin, "-f", "x.ttf", "-t", "13", "--bogus"], capture_output=True, text=True)
    assert r.returncode == 2
    r = subprocess.run([ncc_bin, "-f", "/nonexistent.ttf", "-t", "13"], capture_output=True, text=True)
    assert r.returncode == 101  # Font::from_path(..).unwrap() panics, src/ncc.rs:561


def test_pnm_and_png_decode(tmp_path):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (17, 23), dtype=np.uint8)
    save_pgm(tmp_path / "a.pgm", img)
    assert np.array_equal(load_image(tmp_path / "a.pgm"), img)
    # P6: rgb -> luma with the image crate's weights
    rgb = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    with open(tmp_path / "b.ppm", "wb") as f:
        f.write(b"P6\n# comment\n7 5\n255\n" + rgb.tobytes())
    r32 = rgb.astype(np.uint32)
    want = ((2126 * r32[..., 0] + 7152 * r32[..., 1] + 722 * r32[..., 2]) // 10000).astype(np.uint8)
    assert np.array_equal(load_image(tmp_path / "b.ppm"), want)
    # P2 ascii
    with open(tmp_path / "c.pgm", "w") as f:
        f.write("P2\n3 2\n255\n0 128 255\n1 2 3\n")
    assert load_image(tmp_path / "c.pgm").tolist() == [[0, 128, 255], [1, 2, 3]]

    def png(data, w, h, depth, ctype, filt_rows):
        def chunk(t, d):
            return len(d).to_bytes(4, "big") + t + d + zlib.crc32(t + d).to_bytes(4, "big")
        return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([depth, ctype, 0, 0, 0]))
                + chunk(b"IDAT", zlib.compress(filt_rows)) + chunk(b"IEND", b""))

    # 8-bit grey with Sub and Up filters
    raw = bytearray()
    for y in range(img.shape[0]):
        row = img[y].astype(np.int32)
        if y % 2 == 0:
            raw += bytes([1]) + bytes(((row - np.concatenate([[0], row[:-1]])) % 256).astype(np.uint8))
        else:
            raw += bytes([2]) + bytes(((row - img[y - 1].astype(np.int32)) % 256).astype(np.uint8))
    (tmp_path / "d.png").write_bytes(png(None, img.shape[1], img.shape[0], 8, 0, bytes(raw)))
    assert np.array_equal(load_image(tmp_path / "d.png"), img)
    with pytest.raises(OSError):
        load_image(tmp_path / "missing.png")


def test_line_text_and_space_extension():
    """focr_line_text: the reference's concatenation (src/ncc.rs:869-876) and the opt-in gap -> blanks extension."""
    import ctypes as C

    from font_ocr_amd import _native as N
    from font_ocr_amd.bank import HIT_DTYPE
    from font_ocr_amd.searcher import text_of

    adv = 7.8267
    cells = [0, 1, 2, 4, 5, 9, 10]  # one blank after the third letter, three after the fifth
    line = np.zeros(len(cells), HIT_DTYPE)
    line["x"] = [int(45 + c * adv) for c in cells]  # origins are floor(pen), as synth pages / real scans give them
    line["letter"] = [ord(ch) for ch in "abcd"] + [0xE9, 0x20AC, 0x1F600]  # 2-, 3- and 4-byte UTF-8 too
    assert text_of([line]) == "abcdé€\U0001F600"
    assert text_of([line], advance_px=adv) == "abc dé   €\U0001F600"
    assert text_of([line[:0]], advance_px=adv) == ""
    # size query + truncation contract
    host = N.host()
    need = host.focr_line_text(line.ctypes.data_as(C.c_void_p), len(line), adv, 1, None, 0)
    assert need == len("abc dé   €\U0001F600".encode())
    buf = C.create_string_buffer(6)
    assert host.focr_line_text(line.ctypes.data_as(C.c_void_p), len(line), adv, 1, buf, 6) == need
    assert buf.value == b"abc d"


def test_image_probe_and_decode_into_caller_memory(tmp_path):
    """focr_image_probe (header only) and focr_image_load_luma8_into (the `ncc` binary's decoders write straight into
    page-locked batch slabs): PGM in place, other formats through the general decoder, too-small slots refused."""
    import ctypes as C

    from font_ocr_amd import _native as N

    host = N.host()
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (31, 45), dtype=np.uint8)
    save_pgm(tmp_path / "a.pgm", img)
    with open(tmp_path / "c.pgm", "wb") as f:  # comment lines in the header, payload right after one whitespace byte
        f.write(b"P5\n# made by hand\n45 31\n# another\n255\n" + img.tobytes())
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(31))
    import struct

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(tmp_path / "b.png", "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 45, 31, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    err = C.create_string_buffer(256)
    for name in ("a.pgm", "c.pgm", "b.png"):
        w, h = C.c_size_t(), C.c_size_t()
        assert host.focr_image_probe(str(tmp_path / name).encode(), C.byref(w), C.byref(h), err, 256) == 0, err.value
        assert (w.value, h.value) == (45, 31)
        dst = np.full(45 * 31 + 7, 0xAB, np.uint8)
        w2, h2 = C.c_size_t(), C.c_size_t()
        assert host.focr_image_load_luma8_into(str(tmp_path / name).encode(), dst.ctypes.data, 45 * 31, C.byref(w2), C.byref(h2), err, 256) == 0, err.value
        assert (w2.value, h2.value) == (45, 31)
        assert np.array_equal(dst[: 45 * 31].reshape(31, 45), img) and (dst[45 * 31:] == 0xAB).all(), name
        assert host.focr_image_load_luma8_into(str(tmp_path / name).encode(), dst.ctypes.data, 45 * 31 - 1, C.byref(w2), C.byref(h2), err, 256) != 0
    with open(tmp_path / "t.pgm", "wb") as f:
        f.write(b"P5\n45 31\n255\n" + img.tobytes()[:-5])
    assert host.focr_image_load_luma8_into(str(tmp_path / "t.pgm").encode(), dst.ctypes.data, 45 * 31, C.byref(w2), C.byref(h2), err, 256) != 0
    assert b"truncated" in err.value
    assert host.focr_image_probe(str(tmp_path / "missing.pgm").encode(), C.byref(w), C.byref(h), err, 256) != 0
