"""Second witness for the Rust half of the path (TEST INFRASTRUCTURE, like everything under oracle/).

A literal, statement-by-statement Python transliteration of the reference's Rust functions that cannot be
compiled here (no rustc): written from the text of /root/reference/src/ncc.rs, NOT from oracle/ncc_oracle.c, so
that the two readings of the Rust are independent.  tests/test_oracle.py runs both on the same inputs; where they
agree, the device path (post.hip, stats_kernel) has two restatements behind its parity tests instead of one.
Parity stays "unpinned" for these rows in the strict sense (nothing executes the Rust), and says so.

Pure-Python loops: use on small inputs only.

  ncc_sum_table / ncc_sumsqr_table          src/ncc.rs:938-974
  ncc_sum_table_sum_nz / ..sumsqr..sum_nz   src/ncc.rs:976-983, 1006-1013
  Searcher::prepare_for_size                src/ncc.rs:263-318
  partition_by                              src/ncc.rs:1036-1052
  process_hits                              src/ncc.rs:723-786
"""
import math
import struct

U32 = 0xFFFFFFFF
U64 = 0xFFFFFFFFFFFFFFFF


class Array2:
    """src/ncc.rs:198-229: row-major box indexed (x, y) -> data[y * cols + x]."""

    def __init__(self, rows, cols, fill=0):
        self.rows, self.cols = rows, cols
        self.data = [fill] * (rows * cols)

    def __getitem__(self, xy):
        x, y = xy
        return self.data[y * self.cols + x]

    def __setitem__(self, xy, v):
        x, y = xy
        self.data[y * self.cols + x] = v


def array2_from(rows_of_rows):
    a = Array2(len(rows_of_rows), len(rows_of_rows[0]))
    for y, row in enumerate(rows_of_rows):
        for x, v in enumerate(row):
            a[(x, y)] = int(v)
    return a


def ncc_sum_table(pixels):  # src/ncc.rs:938-955; u32 arithmetic (wraps in release builds)
    ret = Array2(pixels.rows, pixels.cols)
    ret[(0, 0)] = pixels[(0, 0)]
    for x in range(1, pixels.cols):
        ret[(x, 0)] = (pixels[(x, 0)] + ret[(x - 1, 0)]) & U32
    for y in range(1, pixels.rows):
        ret[(0, y)] = (pixels[(0, y)] + ret[(0, y - 1)]) & U32
    for y in range(1, pixels.rows):
        for x in range(1, pixels.cols):
            ret[(x, y)] = (pixels[(x, y)] + ret[(x - 1, y)] + ret[(x, y - 1)] - ret[(x - 1, y - 1)]) & U32
    return ret


def ncc_sumsqr_table(pixels):  # src/ncc.rs:957-974: first row / column hold p*p only (NOT cumulative)
    ret = Array2(pixels.rows, pixels.cols)
    for x in range(pixels.cols):
        p = pixels[(x, 0)]
        ret[(x, 0)] = p * p
    for y in range(pixels.rows):
        p = pixels[(0, y)]
        ret[(0, y)] = p * p
    for y in range(1, pixels.rows):
        for x in range(1, pixels.cols):
            p = pixels[(x, y)]
            ret[(x, y)] = (p * p + ret[(x - 1, y)] + ret[(x, y - 1)] - ret[(x - 1, y - 1)]) & U64
    return ret


def _as_i64(v):
    v &= U64
    return v - (1 << 64) if v >> 63 else v


def ncc_sum_table_sum_nz(s, xy, wh):  # src/ncc.rs:976-983
    (x, y), (w, h) = xy, wh
    a = s[(x + w - 1, y + h - 1)]
    b = s[(x - 1, y + h - 1)]
    c = s[(x + w - 1, y - 1)]
    d = s[(x - 1, y - 1)]
    return (a - b + d - c) & U32  # `as u32`


def ncc_sumsqr_table_sum_nz(s, xy, wh):  # src/ncc.rs:1006-1013
    (x, y), (w, h) = xy, wh
    a = _as_i64(s[(x + w - 1, y + h - 1)])
    b = _as_i64(s[(x - 1, y + h - 1)])
    c = _as_i64(s[(x + w - 1, y - 1)])
    d = _as_i64(s[(x - 1, y - 1)])
    return (a - b + d - c) & U64  # `as u64`


def prepare_for_size(reference_u8, n_w, n_h):
    """src/ncc.rs:263-318.  reference_u8: Array2 of the inverted page.  Returns (patch_sum, patch_rnorm, start_end)
    with entries outside [start, end) left at their initial 0 (the reference leaves them stale)."""
    sum_table = ncc_sum_table(reference_u8)
    sumsqr_table = ncc_sumsqr_table(reference_u8)
    r_w, r_h = reference_u8.cols, reference_u8.rows
    patch_sum = Array2(r_h, r_w)
    patch_rnorm = Array2(r_h, r_w, 0.0)
    start_end = [0] * (r_h * 2)
    n = n_h * n_w
    x_searches = r_w - n_w + 1
    y_searches = r_h - n_h + 1
    for y in range(1, y_searches):
        x = 1
        while x < x_searches:
            if ncc_sum_table_sum_nz(sum_table, (x, y), (n_w, n_h)) != 0:
                break
            x += 1
        start = x
        x = x_searches - 1
        while x > start:
            if ncc_sum_table_sum_nz(sum_table, (x, y), (n_w, n_h)) != 0:
                break
            x -= 1
        end = x + 1
        for x in range(start, end):
            s_p = ncc_sum_table_sum_nz(sum_table, (x, y), (n_w, n_h))
            s2_p = ncc_sumsqr_table_sum_nz(sumsqr_table, (x, y), (n_w, n_h))
            norm = float(s2_p) - float((s_p * s_p) & U64) / float(n)  # `as f64` of exact ints < 2^53 is exact
            patch_sum[(x, y)] = s_p
            if norm > 0.0:
                patch_rnorm[(x, y)] = 1.0 / math.sqrt(norm)
            elif norm == 0.0:
                patch_rnorm[(x, y)] = math.inf  # 1. / 0f64.sqrt()
            else:
                patch_rnorm[(x, y)] = math.nan  # sqrt of a negative
        start_end[y * 2 + 0] = start  # try_into::<u16>().unwrap(): pages are <= 65535 wide
        start_end[y * 2 + 1] = end
    return patch_sum, patch_rnorm, start_end


def partition_by(xs, pred):
    """src/ncc.rs:1036-1052.  `last` moves only when a slice closes: slices are anchored on their first element.
    Panics (here: IndexError) on an empty input, as `it.next().unwrap()` does."""
    it = iter(xs)
    i = 0
    j = 0
    try:
        last = next(it)
    except StopIteration:
        raise IndexError("partition_by: called `Option::unwrap()` on a `None` value") from None
    slices = []
    for nxt in it:
        j += 1
        if not pred(last, nxt):
            slices.append((i, j))
            i = j
            last = nxt
    slices.append((i, j + 1))
    return slices


def _f32(v):
    return struct.unpack("<f", struct.pack("<f", v))[0]


def total_cmp_key(f):
    """f32::total_cmp as a sortable signed integer (core::f32: bits ^= (((bits >> 31) as u32) >> 1) as i32)."""
    b = struct.unpack("<i", struct.pack("<f", f))[0]
    return b ^ (((b >> 31) & U32) >> 1)


def process_hits(all_hits, anchor_threshold, overlap):
    """src/ncc.rs:723-786.  all_hits: sequence of dicts with keys x, y, similarity (+ anything else, carried along),
    in get_hits order.  Returns the list of lines (lists of the same dicts)."""
    anchor_threshold = _f32(anchor_threshold)
    keep_y = set()
    for h in all_hits:
        if _f32(h["similarity"]) >= anchor_threshold:
            keep_y.add(h["y"])
    hits = []
    for h in all_hits:
        if h["y"] in keep_y:
            hits.append(h)
    hits.sort(key=lambda m: m["y"])  # sort_by_key: stable
    line_slices = partition_by(hits, lambda a, b: a["y"] == b["y"])
    lines = []
    for (i, j) in line_slices:
        hits[i:j] = sorted(hits[i:j], key=lambda m: m["x"])  # stable
    for (i, j) in line_slices:
        sl = hits[i:j]
        duplicate_slices = partition_by(sl, lambda a, b: abs(a["x"] - b["x"]) <= overlap)
        dedup = []
        for (i2, j2) in duplicate_slices:
            best = None
            for e in sl[i2:j2]:  # Iterator::max_by: the LAST maximum is returned
                if best is None or total_cmp_key(e["similarity"]) >= total_cmp_key(best["similarity"]):
                    best = e
            dedup.append(best)
        lines.append(dedup)
    return lines
