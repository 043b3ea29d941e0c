"""Host-side mirror of the reference's Searcher / get_hits / process_hits for the device path.

Everything here is a thin ctypes call into libfocr_hip.so (include/focr_ncc.h).  Names follow the
reference (src/ncc.rs): `Searcher.search_c_u8` is the per-template drop-in (FFI symbols
ncc_8_u8 / ncc_16_u8), `Scanner` is the batched form of get_hits' search loop over resident pages,
`process_hits` the anchor/line/overlap pass.  There is no CPU fallback: a missing library or
device raises.
"""
import ctypes as C

import numpy as np

from . import _native as N
from .bank import HIT_DTYPE, MATCH_DTYPE

MAX_MATCHES = 1024  # src/ncc.rs:31
SCAN_MFMA, SCAN_DIRECT, SCAN_RUST = 0, 1, 2
PREFILTER_AUTO, PREFILTER_ONE_STAGE, PREFILTER_LEGACY = 0, 1, 3


class FocrError(RuntimeError):
    pass


def device_count():
    return int(N.hip().focr_device_count())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class PinnedPages:
    """(n, r_h, r_w) uint8 array in page-locked host memory (focr_host_alloc): `.array` is a numpy view.
    Scanner.upload_pages from it is an asynchronous DMA (see include/focr_ncc.h)."""

    def __init__(self, n, r_h, r_w):
        self._lib = N.hip()
        p = C.c_void_p()
        rc = self._lib.focr_host_alloc(n * r_h * r_w, C.byref(p))
        if rc != 0:
            raise FocrError(f"[{rc}] {self._lib.focr_last_error_global().decode()}")
        self._p = p
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n, r_h, r_w))

    def close(self):
        if self._p is not None:
            self.array = None
            self._lib.focr_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Searcher:
    """Per-page state of the reference (src/ncc.rs:128-141, 231-261) for the drop-in FFI symbols.

    The caller supplies the window tables exactly as the reference's prepare_for_size produces
    them (src/ncc.rs:263-318); search_c_u8 then does what src/ncc.rs:332-404 does, with the kernel
    call landing on the GPU through the reference's own symbol names.
    """

    def __init__(self, page_inv):
        self.reference_u8 = np.ascontiguousarray(page_inv, np.uint8)
        r_h, r_w = self.reference_u8.shape
        self.acc_u32 = np.zeros(r_w * 8 + 8, np.uint32)  # src/ncc.rs:242
        self.matches_c = np.zeros(MAX_MATCHES, MATCH_DTYPE)  # src/ncc.rs:240

    def search_c_u8(self, needle, stats, threshold, n_out=MAX_MATCHES):
        """needle: (n_h, n_w) uint8; stats = (patch_sum, patch_rnorm, start_end).  Returns MATCH_DTYPE[]."""
        lib = N.hip()
        if lib.focr_device_count() <= 0:  # the FFI signature has no error channel: fail loudly here
            raise FocrError("no HIP device available; the ncc_*_u8 symbols have no CPU fallback")
        needle = np.ascontiguousarray(needle, np.uint8)
        n_h, n_w = needle.shape
        if n_w <= 8:  # src/ncc.rs:337
            width, fn = 8, lib.ncc_8_u8
        elif n_w <= 16:  # src/ncc.rs:364
            width, fn = 16, lib.ncc_16_u8
        else:
            raise FocrError("not handled")  # panic!("not handled"), src/ncc.rs:392
        padded = np.zeros((n_h, width), np.uint8)  # copy_needle_n_u8, src/ncc.rs:925-935
        padded[:, :n_w] = needle
        patch_sum, patch_rnorm, start_end = stats
        patch_sum = np.ascontiguousarray(patch_sum, np.uint32)
        patch_rnorm = np.ascontiguousarray(patch_rnorm, np.float64)
        start_end = np.ascontiguousarray(start_end, np.uint16)
        r_h, r_w = self.reference_u8.shape
        out = self.matches_c if n_out == MAX_MATCHES else np.zeros(n_out, MATCH_DTYPE)
        n = fn(_ptr(self.reference_u8), r_w, r_h, _ptr(padded), n_w, n_h, _ptr(self.acc_u32), self.acc_u32.size,
               _ptr(patch_sum), _ptr(patch_rnorm), _ptr(start_end), float(threshold), _ptr(out), n_out)
        return out[:n].copy()


class Scanner:
    """Batched device scan: one bank, N resident pages (get_hits' loop, src/ncc.rs:576-701)."""

    def __init__(self, device=0, _borrowed=None):
        self._lib = N.hip()
        self._owned = _borrowed is None
        if _borrowed is None:
            h = C.c_void_p()
            rc = self._lib.focr_ctx_create(int(device), C.byref(h))
            if rc != 0:
                raise FocrError(self._lib.focr_last_error_global().decode())
        else:  # a context that belongs to a Pipeline
            h = C.c_void_p(_borrowed)
        self._h = h
        self.bank = None
        self.n_pages = self.r_w = self.r_h = 0

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._lib.focr_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != 0:
            raise FocrError(f"[{rc}] " + self._lib.focr_last_error(self._h).decode())

    def set_bank(self, bank):
        self._ck(self._lib.focr_bank_upload(self._h, _ptr(bank.templates), len(bank.templates), _ptr(bank.needles),
                                            bank.needles.size))
        self.bank = bank

    def alloc_pages(self, n_pages, r_w, r_h):
        self._ck(self._lib.focr_pages_alloc(self._h, n_pages, r_w, r_h))
        self.n_pages, self.r_w, self.r_h = n_pages, r_w, r_h

    def upload_pages(self, luma, first=0, invert=True):
        """luma: (n, r_h, r_w) uint8 as the image decoder gives it (255 = paper)."""
        luma = np.ascontiguousarray(luma, np.uint8)
        if luma.ndim == 2:
            luma = luma[None]
        assert luma.shape[1:] == (self.r_h, self.r_w)
        self._ck(self._lib.focr_pages_upload(self._h, first, luma.shape[0], _ptr(luma), int(bool(invert))))

    def upload_pages_device(self, dev_ptr, count, first=0, invert=True):
        """Pages already in HBM (e.g. tensor.data_ptr() of a uint8 CUDA tensor)."""
        self._ck(self._lib.focr_pages_upload_device(self._h, first, count, C.c_void_p(int(dev_ptr)), int(bool(invert))))

    def set_pages(self, luma, invert=True):
        luma = np.ascontiguousarray(luma, np.uint8)
        if luma.ndim == 2:
            luma = luma[None]
        self.alloc_pages(luma.shape[0], luma.shape[2], luma.shape[1])
        self.upload_pages(luma, 0, invert)

    def scan(self, threshold=0.8, cap=MAX_MATCHES, mode=SCAN_MFMA):
        """mode: SCAN_MFMA / SCAN_DIRECT / SCAN_RUST, or (SCAN_MFMA, PREFILTER_*) to pick the MFMA prefilter too."""
        if isinstance(mode, tuple):
            mode, prefilter = mode
            self.set_prefilter(prefilter)
        self._ck(self._lib.focr_scan(self._h, float(threshold), int(cap), int(mode)))

    def set_prefilter(self, prefilter):
        """PREFILTER_AUTO / PREFILTER_ONE_STAGE / PREFILTER_LEGACY (focr_ctx_set_prefilter); results never change."""
        self._ck(self._lib.focr_ctx_set_prefilter(self._h, int(prefilter)))

    def set_row_tail(self, on):
        """focr_ctx_set_row_tail: True / 1 = hits-first row tail (default), False / 0 = the legacy radix-sort tail; results never change."""
        self._ck(self._lib.focr_ctx_set_row_tail(self._h, int(on)))

    def set_column_drop(self, on):
        """focr_ctx_set_column_drop: bound the last column of 9- / 13-wide classes instead of multiplying it (default on);
        takes effect at the next set_bank; results never change."""
        self._ck(self._lib.focr_ctx_set_column_drop(self._h, int(bool(on))))

    def set_size_estimates(self, on):
        """focr_ctx_set_size_estimates: repeat scans of one setup queue every phase without host waits (default on)."""
        self._ck(self._lib.focr_ctx_set_size_estimates(self._h, int(bool(on))))

    def size_estimate_stats(self):
        """focr_size_estimate_stats: {'redone': batches redone with exact sizes, 'margin': next estimate's margin, 'row_max': ...}."""
        r, m, x = C.c_uint64(), C.c_double(), C.c_uint32()
        self._ck(self._lib.focr_size_estimate_stats(self._h, C.byref(r), C.byref(m), C.byref(x)))
        return {"redone": int(r.value), "margin": float(m.value), "row_max": int(x.value)}

    def force_split(self, on):
        """Test hook (focr_debug_force_split): scan the batch in page sub-ranges as after a candidate overflow."""
        self._ck(self._lib.focr_debug_force_split(self._h, int(bool(on))))

    def phase_stamps(self):
        """focr_debug_phase_stamps: the last batch's phases on the device's clock (ms since the device's first context was created)."""
        out = (C.c_double * 9)()
        self._ck(self._lib.focr_debug_phase_stamps(self._h, out))
        return dict(zip(("stats_start", "stats_end", "scan_end", "verify_end", "order_end", "post_start", "post_end", "scan_launch_start", "scan_launch_end"),
                        [float(v) for v in out]))

    def set_tail_grid(self, num, den):
        """Test hook (focr_debug_set_tail_grid): the tail's persistent kernels on num / den times their workgroups (0, 0: as designed)."""
        self._ck(self._lib.focr_debug_set_tail_grid(self._h, int(num), int(den)))

    def set_stats_form(self, form):
        """Test hook (focr_debug_set_stats_form): 1 = the LDS-tiled statistics kernel for every class, 0 = the register form where it applies."""
        self._ck(self._lib.focr_debug_set_stats_form(self._h, int(form)))

    def planes(self):
        """Test hook (focr_debug_planes): the int16 threshold planes of the last MFMA scan, flat."""
        n = C.c_size_t(0)
        self._ck(self._lib.focr_debug_planes(self._h, None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.int16)
        if n.value:
            self._ck(self._lib.focr_debug_planes(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def set_scan_cus(self, max_cus):
        """Upper bound on the CUs the persistent scan kernel occupies (0 = all)."""
        self._ck(self._lib.focr_ctx_set_scan_cus(self._h, int(max_cus)))

    def sync(self):
        self._ck(self._lib.focr_sync(self._h))

    def counts(self):
        out = np.zeros((self.n_pages, len(self.bank)), np.uint32)
        self._ck(self._lib.focr_get_counts(self._h, _ptr(out)))
        return out

    def total_matches(self):
        return int(self._lib.focr_total_matches(self._h))

    def matches(self):
        """(offsets[n_pages*T+1], matches[total]) in (page, template, y, x) order."""
        n_seg = self.n_pages * len(self.bank)
        offsets = np.zeros(n_seg + 1, np.uint64)
        m = np.zeros(self.total_matches(), MATCH_DTYPE)
        self._ck(self._lib.focr_get_matches(self._h, _ptr(offsets), _ptr(m)))
        return offsets, m

    def process_hits(self, anchor_threshold=0.95, overlap=5):
        self._ck(self._lib.focr_process_hits(self._h, float(anchor_threshold), int(overlap)))

    def lines(self):
        """-> list over pages of list over lines of HIT_DTYPE arrays."""
        n_lines = int(self._lib.focr_total_lines(self._h))
        n_chars = int(self._lib.focr_total_chars(self._h))
        page_off = np.zeros(self.n_pages + 1, np.uint64)
        line_off = np.zeros(n_lines + 1, np.uint64)
        chars = np.zeros(n_chars, HIT_DTYPE)
        self._ck(self._lib.focr_get_lines(self._h, _ptr(page_off), _ptr(line_off), _ptr(chars)))
        out = []
        for p in range(self.n_pages):
            out.append([chars[int(line_off[k]): int(line_off[k + 1])] for k in range(int(page_off[p]), int(page_off[p + 1]))])
        return out

    def lines_flat(self):
        """All post-processed characters of the batch as one HIT_DTYPE array (page, line, x order)."""
        n_chars = int(self._lib.focr_total_chars(self._h))
        chars = np.zeros(n_chars, HIT_DTYPE)
        self._ck(self._lib.focr_get_lines(self._h, None, None, _ptr(chars)))
        return chars

    def device_chars(self):
        """(device pointer, count) of the post-processed characters still resident in HBM (HIT_DTYPE records)."""
        ptr = self._lib.focr_lines_device_chars(self._h)
        return (int(ptr) if ptr else 0), int(self._lib.focr_total_chars(self._h))

    def total_chars(self):
        return int(self._lib.focr_total_chars(self._h))

    def timings(self):
        ms = (C.c_float * 6)()
        self._lib.focr_last_timings(self._h, ms)
        return dict(zip(("stats", "scan", "verify", "order", "process_hits", "total"), [float(v) for v in ms]))

    def counters(self):
        c = (C.c_uint64 * 4)()
        self._lib.focr_last_counters(self._h, c)
        return dict(zip(("candidates", "raw_hits", "algorithmic_macs", "issued_macs"), [int(v) for v in c]))

    def launches(self):
        """Per-launch records of the last scan's scan kernels: list of dicts (name, ms, n_templates, macs)."""
        n = int(self._lib.focr_last_launches(self._h, None, 0))
        arr = (N.LaunchInfo * max(n, 1))()
        self._lib.focr_last_launches(self._h, arr, n)
        return [dict(name=arr[i].name.decode(), ms=float(arr[i].ms), n_templates=int(arr[i].n_templates),
                     alg_macs=int(arr[i].alg_macs), issued_macs=int(arr[i].issued_macs)) for i in range(n)]

    def debug_rnorm(self, s, s2, n):
        s = np.ascontiguousarray(s, np.uint32)
        s2 = np.ascontiguousarray(s2, np.uint64)
        n = np.ascontiguousarray(n, np.uint32)
        out = np.zeros(len(s), np.float64)
        self._ck(self._lib.focr_debug_rnorm(self._h, _ptr(s), _ptr(s2), _ptr(n), len(s), _ptr(out)))
        return out


class Pipeline:
    """Batches in flight (focr_pipe_*, include/focr_ncc.h): `n_lanes` streams on one device, `depth` contexts per lane; a batch is
    queued on the device the moment it is submitted.  submit() hands batches out round-robin over the n_lanes * depth contexts
    (`scanners`), wait() returns the Scanner view of the context holding a batch's results, release() frees that context for its
    next batch.  Calls block in native code with the GIL released."""

    def __init__(self, device=0, n_lanes=3, depth=None):
        self._lib = N.hip()
        h = C.c_void_p()
        if depth is None:
            rc = self._lib.focr_pipe_create(int(device), int(n_lanes), C.byref(h))
        else:
            rc = self._lib.focr_pipe_create2(int(device), int(n_lanes), int(depth), C.byref(h))
        if rc != 0:
            raise FocrError(self._lib.focr_last_error_global().decode())
        self._h = h
        self.n_lanes = int(self._lib.focr_pipe_lanes(h))
        self.scanners = [Scanner(device, _borrowed=self._lib.focr_pipe_context(h, i)) for i in range(int(self._lib.focr_pipe_contexts(h)))]
        self._keep = {}  # ticket -> host array kept alive while its batch is in flight

    def close(self):
        if getattr(self, "_h", None):
            for sc in self.scanners:
                sc.close()
            self._lib.focr_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_bank(self, bank):
        rc = self._lib.focr_pipe_bank_upload(self._h, _ptr(bank.templates), len(bank.templates), _ptr(bank.needles), bank.needles.size)
        if rc != 0:
            raise FocrError(f"[{rc}] {self._lib.focr_last_error_global().decode()}")
        for sc in self.scanners:
            sc.bank = bank

    def submit(self, luma=None, threshold=0.8, cap=MAX_MATCHES, mode=SCAN_MFMA, process_hits=True, anchor_threshold=0.95,
               overlap=5, invert=True, device_ptr=None, shape=None, chars_out=None):
        """luma: (n, r_h, r_w) uint8 host pages; or device_ptr + shape=(n, r_h, r_w); or neither = rescan the lane's
        resident pages.  chars_out=(device pointer, bytes): also copy the batch's characters there.  Returns the ticket."""
        t = C.c_uint64()
        if luma is not None:
            luma = np.ascontiguousarray(luma, np.uint8)
            if luma.ndim == 2:
                luma = luma[None]
            n, r_h, r_w = luma.shape
            ptr, on_dev = _ptr(luma), 0
        elif device_ptr is not None:
            n, r_h, r_w = shape
            ptr, on_dev = C.c_void_p(int(device_ptr)), 1
        else:
            n = r_h = r_w = 0
            ptr, on_dev = None, 0
        rc = self._lib.focr_pipe_submit(self._h, ptr, on_dev, n, r_w, r_h, int(bool(invert)), float(threshold), int(cap), int(mode),
                                        int(bool(process_hits)), float(anchor_threshold), int(overlap),
                                        C.c_void_p(int(chars_out[0])) if chars_out else None, int(chars_out[1]) if chars_out else 0,
                                        C.byref(t))
        if rc != 0:
            raise FocrError(f"[{rc}] {self._lib.focr_last_error_global().decode()}")
        if luma is not None:
            self._keep[t.value] = luma
            if getattr(self, "_announced", None):
                self._announced.pop(0)
        lane = self.scanners[(t.value - 1) % len(self.scanners)]
        if n:
            lane.n_pages, lane.r_w, lane.r_h = n, r_w, r_h
        return t.value

    def prefetch(self, luma, invert=True):
        """focr_pipe_prefetch: announce the host pages (n, r_h, r_w) uint8 of the batch that will be submitted after everything
        announced or submitted so far, and start their copy to the device and their ingest now (page-locked memory: PinnedPages).
        The matching submit must bring the same array (and the same `invert`, or the lane uploads the batch itself)."""
        if luma.dtype != np.uint8 or not luma.flags["C_CONTIGUOUS"] or luma.ndim != 3:
            raise ValueError("prefetch: a C-contiguous (n, r_h, r_w) uint8 array")
        n, r_h, r_w = luma.shape
        rc = self._lib.focr_pipe_prefetch(self._h, _ptr(luma), n, r_w, r_h, int(bool(invert)))
        if rc != 0:
            raise FocrError(f"[{rc}] {self._lib.focr_last_error_global().decode()}")
        self._announced = getattr(self, "_announced", [])
        self._announced.append(luma)  # kept alive until its submit takes over

    def announce_last(self):
        """focr_pipe_announce_last: the NEXT batch submitted is the stream's last for now: its tail may take the whole GPU."""
        self._lib.focr_pipe_announce_last(self._h)

    def end_of_stream(self):
        """focr_pipe_end_of_stream: nothing follows the newest batch for now — effective only if the executor has not queued that
        batch yet (say it before the last submit with announce_last)."""
        self._lib.focr_pipe_end_of_stream(self._h)

    def ticket_times(self, ticket):
        """focr_pipe_ticket_times (after wait, before release): dict of host stamps (us since the executor was created) and the
        device-side interval to the previous ticket's completion (ms; < 0: not available)."""
        t = N.TicketTimes()
        rc = self._lib.focr_pipe_ticket_times(self._h, int(ticket), C.byref(t))
        if rc != 0:
            raise FocrError(f"[{rc}] focr_pipe_ticket_times: ticket is not complete and unreleased")
        return {k: float(getattr(t, k)) for k, _ in N.TicketTimes._fields_}

    def wait(self, ticket):
        """Blocks until the batch is done; returns the Scanner whose getters (matches, lines, counts...) see it."""
        h = C.c_void_p()
        rc = self._lib.focr_pipe_wait(self._h, int(ticket), C.byref(h))
        self._keep.pop(ticket, None)
        sc = self.scanners[(ticket - 1) % len(self.scanners)]
        if rc != 0:
            raise FocrError(f"[{rc}] " + self._lib.focr_last_error(sc._h).decode())
        return sc

    def release(self, ticket):
        rc = self._lib.focr_pipe_release(self._h, int(ticket))
        if rc != 0:
            raise FocrError(f"[{rc}] focr_pipe_release: ticket is not outstanding")


def text_of(lines, advance_px=None):
    """Default `ncc` output of one page: letters of each line concatenated (src/ncc.rs:869-876).  With advance_px
    (extension): blanks are inserted where consecutive origins are more than one advance apart."""
    host = N.host()
    out = []
    for line in lines:
        line = np.ascontiguousarray(line, HIT_DTYPE)
        buf = C.create_string_buffer(8 * len(line) * 4 + 16)
        host.focr_line_text(_ptr(line), len(line), float(advance_px or 0.0), int(advance_px is not None), buf, len(buf))
        out.append(buf.value.decode("utf-8"))
    return "\n".join(out)


class Fleet:
    """Every GPU of the node (focr_fleet_*, include/focr_ncc.h): one Pipeline-like executor per device, batch k -> device
    k % n_devices; retire the tickets in submission order.  devices=None: all visible devices (the same device may be
    listed more than once: several executors share it — what the one-GPU tests do)."""

    def __init__(self, devices=None, lanes=3):
        self._lib = N.hip()
        h = C.c_void_p()
        if devices:
            arr = (C.c_int * len(devices))(*devices)
            rc = self._lib.focr_fleet_create(arr, len(devices), int(lanes), C.byref(h))
        else:
            rc = self._lib.focr_fleet_create(None, 0, int(lanes), C.byref(h))
        if rc != 0:
            raise FocrError(self._lib.focr_last_error_global().decode())
        self._h = h
        self.n_devices = int(self._lib.focr_fleet_devices(h))
        self.lanes = int(self._lib.focr_fleet_lanes(h))
        self.slots = int(self._lib.focr_fleet_slots(h))  # batches that can be outstanding
        self._views = {}  # context handle -> Scanner view
        self._keep = {}
        self.bank = None

    def close(self):
        if getattr(self, "_h", None):
            for sc in self._views.values():
                sc.close()
            self._lib.focr_fleet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_bank(self, bank):
        rc = self._lib.focr_fleet_bank_upload(self._h, _ptr(bank.templates), len(bank.templates), _ptr(bank.needles), bank.needles.size)
        if rc != 0:
            raise FocrError(f"[{rc}] {self._lib.focr_last_error_global().decode()}")
        self.bank = bank

    def device_of(self, ticket):
        return int(self._lib.focr_fleet_device_of(self._h, int(ticket)))

    def submit(self, luma, threshold=0.8, cap=MAX_MATCHES, mode=SCAN_MFMA, process_hits=True, anchor_threshold=0.95, overlap=5, invert=True):
        luma = np.ascontiguousarray(luma, np.uint8)
        if luma.ndim == 2:
            luma = luma[None]
        n, r_h, r_w = luma.shape
        t = C.c_uint64()
        rc = self._lib.focr_fleet_submit(self._h, _ptr(luma), 0, n, r_w, r_h, int(bool(invert)), float(threshold), int(cap), int(mode),
                                         int(bool(process_hits)), float(anchor_threshold), int(overlap), C.byref(t))
        if rc != 0:
            raise FocrError(f"[{rc}] {self._lib.focr_last_error_global().decode()}")
        self._keep[t.value] = luma
        return t.value

    def wait(self, ticket):
        """Blocks until the batch is done; returns a Scanner view of the context holding its results."""
        h = C.c_void_p()
        rc = self._lib.focr_fleet_wait(self._h, int(ticket), C.byref(h))
        luma = self._keep.pop(ticket, None)
        if rc != 0:
            raise FocrError(f"[{rc}] {self._lib.focr_last_error_global().decode()}")
        sc = self._views.get(h.value)
        if sc is None:
            sc = self._views[h.value] = Scanner(self.device_of(ticket), _borrowed=h.value)
        sc.bank = self.bank
        if luma is not None:
            sc.n_pages, sc.r_h, sc.r_w = luma.shape
        return sc

    def release(self, ticket):
        rc = self._lib.focr_fleet_release(self._h, int(ticket))
        if rc != 0:
            raise FocrError(f"[{rc}] focr_fleet_release: ticket is not outstanding")
