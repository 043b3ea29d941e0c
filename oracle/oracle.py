"""Python handle on the parity checkers (TEST INFRASTRUCTURE — see ncc_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

  liboracle.so        CPU restatement of the reference path (ncc_oracle.c)
  _ref/libncc_ref.so  the reference's AVX2 kernels, compiled unmodified in the build
                      container from /root/reference/src/ncc.cpp (oracle/Makefile)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

MATCH_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("similarity", "<f4")])
HIT_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("similarity", "<f4"), ("letter", "<u4")])
TEMPLATE_DTYPE = np.dtype([("n_w", "<u4"), ("n_h", "<u4"), ("offset", "<u4")])

_KERNEL_ARGS = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t]

_lib = None
_ref = None


def build():
    """Compile liboracle.so (and _ref/libncc_ref.so when /root/reference is present)."""
    import subprocess

    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("FOCR_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")  # FOCR_ORACLE_LIB: the `make asan` build
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_sum_table.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.oracle_sumsqr_table.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.oracle_prepare_for_size.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_copy_needle.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
        L.oracle_ncc_u8.restype = C.c_size_t
        L.oracle_ncc_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t]
        L.oracle_process_hits.restype = C.c_size_t
        L.oracle_process_hits.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]
        L.oracle_scan_page.restype = C.c_size_t
        L.oracle_scan_page.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float,
                                       C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_scan_page_rust.restype = C.c_size_t
        L.oracle_scan_page_rust.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float,
                                            C.c_size_t, C.c_void_p, C.c_void_p]
        L.oracle_scan_pages_mt.restype = C.c_size_t
        L.oracle_scan_pages_mt.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_float, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def ref():
    """The compiled reference kernels, or None when oracle/_ref/libncc_ref.so is absent."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libncc_ref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        for name in ("ncc_8_u8", "ncc_16_u8"):
            fn = getattr(R, name)
            fn.restype = C.c_size_t
            fn.argtypes = _KERNEL_ARGS
        _ref = R
    return _ref


def have_ref():
    return ref() is not None


def invert(luma):
    """image_to_u8, src/ncc.rs:887-892."""
    return (255 - np.asarray(luma, np.uint8)).astype(np.uint8)


def padded(page_inv):
    """Flat copy of an inverted page followed by 64 readable zero bytes (reference over-read)."""
    flat = np.zeros(page_inv.size + 64, np.uint8)
    flat[: page_inv.size] = page_inv.reshape(-1)
    return flat


def tables(page_inv):
    r_h, r_w = page_inv.shape
    px = np.ascontiguousarray(page_inv, np.uint8)
    s = np.zeros((r_h, r_w), np.uint32)
    s2 = np.zeros((r_h, r_w), np.uint64)
    lib().oracle_sum_table(px.ctypes.data, r_w, r_h, s.ctypes.data)
    lib().oracle_sumsqr_table(px.ctypes.data, r_w, r_h, s2.ctypes.data)
    return s, s2


def prepare_for_size(page_inv, n_w, n_h, tabs=None):
    """Searcher::prepare_for_size (src/ncc.rs:263-318) -> patch_sum, patch_rnorm, start_end."""
    r_h, r_w = page_inv.shape
    s, s2 = tabs if tabs is not None else tables(page_inv)
    patch_sum = np.zeros((r_h, r_w), np.uint32)
    patch_rnorm = np.zeros((r_h, r_w), np.float64)
    start_end = np.zeros(2 * r_h, np.uint16)
    lib().oracle_prepare_for_size(s.ctypes.data, s2.ctypes.data, r_w, r_h, n_w, n_h, patch_sum.ctypes.data,
                                  patch_rnorm.ctypes.data, start_end.ctypes.data)
    return patch_sum, patch_rnorm, start_end


def pad_needle(needle):
    """copy_needle_n_u8 (src/ncc.rs:925-935): (n_h, n_w) -> (n_h, N) with N = 8 | 16 (| 32: this build's extension)."""
    n_h, n_w = needle.shape
    N = 8 if n_w <= 8 else 16 if n_w <= 16 else 32
    out = np.zeros((n_h, N), np.uint8)
    out[:, :n_w] = needle
    return out


def ncc_u8(page_flat, r_w, r_h, needle, stats, threshold, cap=1024, use_ref=False):
    """One kernel call (ncc_8_u8 / ncc_16_u8 by width).  page_flat must come from padded()."""
    patch_sum, patch_rnorm, start_end = stats
    n_h, n_w = needle.shape
    nd = pad_needle(np.ascontiguousarray(needle, np.uint8))
    N = nd.shape[1]
    out = np.zeros(cap, MATCH_DTYPE)
    if use_ref:
        R = ref()
        if R is None:
            raise RuntimeError("oracle/_ref/libncc_ref.so not built")
        acc = np.zeros(r_w * 8 + 8 + 16, np.uint32)
        fn = R.ncc_8_u8 if N == 8 else R.ncc_16_u8
        c = fn(page_flat.ctypes.data, r_w, r_h, nd.ctypes.data, n_w, n_h, acc.ctypes.data, r_w * 8 + 8,
               patch_sum.ctypes.data, patch_rnorm.ctypes.data, start_end.ctypes.data, threshold, out.ctypes.data, cap)
    else:
        c = lib().oracle_ncc_u8(page_flat.ctypes.data, r_w, r_h, nd.ctypes.data, N, n_w, n_h, patch_sum.ctypes.data,
                                patch_rnorm.ctypes.data, start_end.ctypes.data, threshold, out.ctypes.data, cap)
    return out[:c].copy()


def _bank_arrays(bank):
    tm = np.zeros(len(bank.templates), TEMPLATE_DTYPE)
    tm["n_w"] = bank.templates["n_w"]
    tm["n_h"] = bank.templates["n_h"]
    tm["offset"] = bank.templates["offset"]
    return tm, np.ascontiguousarray(bank.needles, np.uint8)


def _kernel_ptrs(use_ref):
    if not use_ref:
        return None, None
    R = ref()
    if R is None:
        raise RuntimeError("oracle/_ref/libncc_ref.so not built")
    return C.cast(R.ncc_8_u8, C.c_void_p), C.cast(R.ncc_16_u8, C.c_void_p)


def scan_page(page_inv, bank, threshold, cap=1024, use_ref=False):
    """get_hits' search loop for one page (src/ncc.rs:587-701) -> (counts[T], matches[T][cap])."""
    r_h, r_w = page_inv.shape
    flat = padded(page_inv)
    tm, needles = _bank_arrays(bank)
    counts = np.zeros(len(tm), np.uint32)
    matches = np.zeros((len(tm), cap), MATCH_DTYPE)
    k8, k16 = _kernel_ptrs(use_ref)
    lib().oracle_scan_page(flat.ctypes.data, r_w, r_h, needles.ctypes.data, tm.ctypes.data, len(tm), threshold, cap,
                           k8, k16, counts.ctypes.data, matches.ctypes.data)
    return counts, matches


def scan_page_rust(page_inv, bank, threshold, cap=1 << 16):
    """get_hits' loop through the scalar Rust scan (`ncc --rust`, src/ncc.rs:320-330, 406-483): no cap semantics —
    counts[t] is the full number of matches, matches holds the first `cap` of each template."""
    r_h, r_w = page_inv.shape
    flat = padded(page_inv)
    tm, needles = _bank_arrays(bank)
    counts = np.zeros(len(tm), np.uint32)
    matches = np.zeros((len(tm), cap), MATCH_DTYPE)
    lib().oracle_scan_page_rust(flat.ctypes.data, r_w, r_h, needles.ctypes.data, tm.ctypes.data, len(tm), threshold, cap,
                                counts.ctypes.data, matches.ctypes.data)
    return counts, matches


def scan_pages_mt(pages_inv, bank, threshold, cap=1024, use_ref=False, threads=1, keep_matches=False):
    """Page-parallel CPU scan (mirrors rayon par_iter over pages, src/ncc.rs:839-847)."""
    n_pages, r_h, r_w = pages_inv.shape
    stride = r_w * r_h + 64
    buf = np.zeros(n_pages * stride, np.uint8)
    for p in range(n_pages):
        buf[p * stride: p * stride + r_w * r_h] = pages_inv[p].reshape(-1)
    tm, needles = _bank_arrays(bank)
    counts = np.zeros((n_pages, len(tm)), np.uint32)
    matches = np.zeros((n_pages, len(tm), cap), MATCH_DTYPE) if keep_matches else None
    k8, k16 = _kernel_ptrs(use_ref)
    total = lib().oracle_scan_pages_mt(buf.ctypes.data, stride, n_pages, r_w, r_h, needles.ctypes.data, tm.ctypes.data,
                                       len(tm), threshold, cap, k8, k16, threads, counts.ctypes.data,
                                       matches.ctypes.data if keep_matches else None)
    return total, counts, matches


def raw_hits(counts, matches, bank):
    """Flatten per-template lists into get_hits' all_hits order (template-major, then (y,x))."""
    parts = []
    for t in range(len(counts)):
        c = int(counts[t])
        if c == 0:
            continue
        h = np.zeros(c, HIT_DTYPE)
        h["x"] = matches[t, :c]["x"]
        h["y"] = matches[t, :c]["y"]
        h["w"] = bank.templates[t]["n_w"]
        h["h"] = bank.templates[t]["n_h"]
        h["similarity"] = matches[t, :c]["similarity"]
        h["letter"] = bank.templates[t]["letter"]
        parts.append(h)
    return np.concatenate(parts) if parts else np.zeros(0, HIT_DTYPE)


def process_hits(hits, anchor_threshold=0.95, overlap=5):
    """process_hits (src/ncc.rs:723-786) -> list of lines, each an array of HIT_DTYPE."""
    hits = np.ascontiguousarray(hits, HIT_DTYPE)
    n = len(hits)
    out = np.zeros(max(n, 1), HIT_DTYPE)
    ends = np.zeros(max(n, 1), np.uint64)
    nl = lib().oracle_process_hits(hits.ctypes.data, n, anchor_threshold, overlap, out.ctypes.data, ends.ctypes.data)
    lines, start = [], 0
    for k in range(nl):
        e = int(ends[k])
        lines.append(out[start:e].copy())
        start = e
    return lines
