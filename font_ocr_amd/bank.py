"""Template bank, synthetic pages and image I/O — thin Python views over libfocr_host /
libfocr_raster (include/focr_host.h).  Host-side plumbing of the reference's get_hits
(src/ncc.rs:544-649); no scan arithmetic happens here.
"""
import ctypes as C

import numpy as np

from . import _native as N

# src/ncc.rs:28-29
DEFAULT_ALPHABET = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789=+<>(){};:/-"
# "95-glyph bank" of BASELINE.json: printable ASCII 0x20..0x7E
ASCII95 = "".join(chr(c) for c in range(0x20, 0x7F))

BOX_SIZES = {"font": 0, "alphabet": 1, "char": 2}  # src/ncc.rs:40-50

TEMPLATE_DTYPE = np.dtype(
    [("letter", "<u4"), ("n_w", "<u2"), ("n_h", "<u2"), ("offset", "<u4"), ("shift_x", "<u2"), ("shift_y", "<u2"),
     ("off_x", "<f4"), ("off_y", "<f4"), ("corrected_off_y", "<f4"), ("bearing_x", "<f4")]
)
assert TEMPLATE_DTYPE.itemsize == C.sizeof(N.Template)

HIT_DTYPE = np.dtype(
    [("x", "<u2"), ("y", "<u2"), ("w", "<u2"), ("h", "<u2"), ("similarity", "<f4"), ("letter", "<u4"),
     ("template_index", "<u4")]
)
assert HIT_DTYPE.itemsize == C.sizeof(N.Hit)

MATCH_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("similarity", "<f4")])
assert MATCH_DTYPE.itemsize == C.sizeof(N.Match)


class Bank:
    """The (offset x alphabet) template bank of one `ncc` run, in get_hits order
    (offset-major, alphabet-minor; src/ncc.rs:587, 630)."""

    def __init__(self, templates, needles, n_alphabet, x_bits, y_bits, text_size, advance_px):
        self.templates = np.ascontiguousarray(templates, dtype=TEMPLATE_DTYPE)
        self.needles = np.ascontiguousarray(needles, dtype=np.uint8)
        self.n_alphabet = int(n_alphabet)
        self.x_bits = int(x_bits)
        self.y_bits = int(y_bits)
        self.text_size = float(text_size)
        self.advance_px = float(advance_px)

    def __len__(self):
        return len(self.templates)

    def needle(self, t):
        """Dense n_h x n_w A8 raster of template t (canvas pixels verbatim, src/ncc.rs:640)."""
        d = self.templates[t]
        n = int(d["n_w"]) * int(d["n_h"])
        return self.needles[int(d["offset"]): int(d["offset"]) + n].reshape(int(d["n_h"]), int(d["n_w"]))

    def subset(self, idx):
        """Bank restricted to the given template indices (needles repacked)."""
        tm = self.templates[list(idx)].copy()
        parts, off = [], 0
        for i, t in enumerate(idx):
            nd = self.needle(t).reshape(-1)
            parts.append(nd)
            tm[i]["offset"] = off
            off += nd.size
        needles = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        return Bank(tm, needles, len(tm), 0, 0, self.text_size, self.advance_px)

    def _as_struct(self):
        s = N.BankStruct()
        s.templates = self.templates.ctypes.data_as(C.POINTER(N.Template))
        s.n_templates = len(self.templates)
        s.needles = self.needles.ctypes.data_as(C.POINTER(C.c_uint8))
        s.needles_len = self.needles.size
        s.n_alphabet = self.n_alphabet
        s.x_bits = self.x_bits
        s.y_bits = self.y_bits
        s.text_size = self.text_size
        s.advance_px = self.advance_px
        return s

    @staticmethod
    def _from_struct(s):
        n = s.n_templates
        tm = np.frombuffer(C.string_at(s.templates, n * TEMPLATE_DTYPE.itemsize), dtype=TEMPLATE_DTYPE).copy()
        nd = np.frombuffer(C.string_at(s.needles, s.needles_len), dtype=np.uint8).copy()
        return Bank(tm, nd, s.n_alphabet, s.x_bits, s.y_bits, s.text_size, s.advance_px)

    def save(self, path):
        s = self._as_struct()
        if N.host().focr_bank_save(str(path).encode(), C.byref(s)) != 0:
            raise OSError(f"cannot write bank {path}")

    @staticmethod
    def load(path):
        s = N.BankStruct()
        if N.host().focr_bank_load(str(path).encode(), C.byref(s)) != 0:
            raise OSError(f"cannot read bank {path}")
        try:
            return Bank._from_struct(s)
        finally:
            N.host().focr_bank_free(C.byref(s))

    @staticmethod
    def rasterize(font, text_size, x_bits=0, y_bits=0, hinting=False, alphabet=DEFAULT_ALPHABET, box_size="alphabet",
                  x_padding=0, y_padding=0):
        """get_hits' bank loop (src/ncc.rs:563-573, 587-649) via FreeType on the CPU."""
        cps = (C.c_uint32 * len(alphabet))(*[ord(c) for c in alphabet])
        s = N.BankStruct()
        err = C.create_string_buffer(256)
        rc = N.raster().focr_raster_bank(str(font).encode(), float(text_size), int(x_bits), int(y_bits), int(bool(hinting)),
                                         cps, len(alphabet), BOX_SIZES[box_size], int(x_padding), int(y_padding),
                                         C.byref(s), err, 256)
        if rc != 0:
            raise RuntimeError(f"rasterisation failed: {err.value.decode()}")
        try:
            return Bank._from_struct(s)
        finally:
            N.host().focr_bank_free(C.byref(s))


def synth_page(bank, seed, r_w, r_h, with_truth=False):
    """Synthetic luma8 page (255 = paper) composited from the bank; SURVEY.md section 8(d)."""
    out = np.empty((r_h, r_w), np.uint8)
    s = bank._as_struct()
    cap = 1 << 16
    truth = np.zeros(cap if with_truth else 0, HIT_DTYPE)
    n = N.host().focr_synth_page(C.byref(s), int(seed), r_w, r_h, out.ctypes.data,
                                 truth.ctypes.data if with_truth else None, cap if with_truth else 0)
    if with_truth:
        return out, truth[: min(n, cap)].copy()
    return out


SYNTH_SEED_BASE = 0xF0C50000  # page p of a synthetic set uses seed SYNTH_SEED_BASE + p


def synth_pages(bank, n_pages, r_w, r_h, first=0):
    """n_pages synthetic pages as one (n_pages, r_h, r_w) uint8 array."""
    out = np.empty((n_pages, r_h, r_w), np.uint8)
    for p in range(n_pages):
        out[p] = synth_page(bank, SYNTH_SEED_BASE + first + p, r_w, r_h)
    return out


def load_image(path):
    """image::open(path).into_luma8() (src/ncc.rs:575): PNM or PNG -> (h, w) uint8."""
    px = C.POINTER(C.c_uint8)()
    w, h = C.c_size_t(), C.c_size_t()
    err = C.create_string_buffer(256)
    if N.host().focr_image_load_luma8(str(path).encode(), C.byref(px), C.byref(w), C.byref(h), err, 256) != 0:
        raise OSError(err.value.decode())
    try:
        return np.frombuffer(C.string_at(px, w.value * h.value), np.uint8).reshape(h.value, w.value).copy()
    finally:
        C.CDLL(None).free(px)


def save_pgm(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    if N.host().focr_image_save_pgm(str(path).encode(), img.ctypes.data, img.shape[1], img.shape[0]) != 0:
        raise OSError(f"cannot write {path}")


def format_f32(v):
    """Rust `Display` of an f32 (src/ncc.rs:685-697, 855-864)."""
    buf = C.create_string_buffer(64)
    N.host().focr_format_f32(float(v), buf, 64)
    return buf.value.decode()
