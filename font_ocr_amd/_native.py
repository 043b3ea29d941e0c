"""ctypes bindings of the in-tree native libraries (include/focr_ncc.h, include/focr_host.h).

Nothing here computes: it only declares the C ABI.  The HIP library is loaded
lazily and loudly — there is no CPU fallback for the scan.
"""
import ctypes as C
import os

# Several contexts (focr_pipe_*) mean several streams; ROCm maps a process's streams onto 4 hardware queues by
# default and a small copy sharing a queue with a context waits behind its scan kernel.
# Only effective if the HIP runtime has not initialised yet; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")


class Match(C.Structure):
    """focr_match_t == reference Match (src/ncc.cpp:7-10) / MatchC (src/ncc.rs:66-72)."""

    _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("similarity", C.c_float)]


class Template(C.Structure):
    """focr_template_t (include/focr_ncc.h)."""

    _fields_ = [
        ("letter", C.c_uint32),
        ("n_w", C.c_uint16),
        ("n_h", C.c_uint16),
        ("offset", C.c_uint32),
        ("shift_x", C.c_uint16),
        ("shift_y", C.c_uint16),
        ("off_x", C.c_float),
        ("off_y", C.c_float),
        ("corrected_off_y", C.c_float),
        ("bearing_x", C.c_float),
    ]


class Hit(C.Structure):
    """focr_hit_t (include/focr_ncc.h) == MatchWithLetter (src/ncc.rs:74-79)."""

    _fields_ = [
        ("x", C.c_uint16),
        ("y", C.c_uint16),
        ("w", C.c_uint16),
        ("h", C.c_uint16),
        ("similarity", C.c_float),
        ("letter", C.c_uint32),
        ("template_index", C.c_uint32),
    ]


class BankStruct(C.Structure):
    """focr_bank_t (include/focr_host.h)."""

    _fields_ = [
        ("templates", C.POINTER(Template)),
        ("n_templates", C.c_size_t),
        ("needles", C.POINTER(C.c_uint8)),
        ("needles_len", C.c_size_t),
        ("n_alphabet", C.c_uint32),
        ("x_bits", C.c_uint32),
        ("y_bits", C.c_uint32),
        ("text_size", C.c_float),
        ("advance_px", C.c_float),
    ]


class LaunchInfo(C.Structure):
    """focr_launch_info_t (include/focr_ncc.h)."""

    _fields_ = [("name", C.c_char * 64), ("ms", C.c_float), ("n_templates", C.c_uint32), ("alg_macs", C.c_uint64),
                ("issued_macs", C.c_uint64)]


class TicketTimes(C.Structure):
    """focr_ticket_times_t (include/focr_ncc.h)."""

    _fields_ = [("submit_us", C.c_double), ("enqueue_begin_us", C.c_double), ("scan_queued_us", C.c_double), ("enqueue_end_us", C.c_double),
                ("done_us", C.c_double), ("device_gap_ms", C.c_float)]


_NCC_ARGS = [
    C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
    C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t,
]

# every symbol include/focr_ncc.h declares: name -> (restype, argtypes)
HIP_SYMBOLS = {
    "ncc_8_u8": (C.c_size_t, _NCC_ARGS),
    "ncc_16_u8": (C.c_size_t, _NCC_ARGS),
    "focr_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "focr_ctx_destroy": (None, [C.c_void_p]),
    "focr_last_error": (C.c_char_p, [C.c_void_p]),
    "focr_last_error_global": (C.c_char_p, []),
    "focr_device_count": (C.c_int, []),
    "focr_bank_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "focr_pages_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]),
    "focr_pages_upload": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int]),
    "focr_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "focr_host_free": (None, [C.c_void_p]),
    "focr_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "focr_host_unregister": (None, [C.c_void_p]),
    "focr_pages_upload_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int]),
    "focr_scan": (C.c_int, [C.c_void_p, C.c_float, C.c_uint32, C.c_int]),
    "focr_get_counts": (C.c_int, [C.c_void_p, C.c_void_p]),
    "focr_total_matches": (C.c_size_t, [C.c_void_p]),
    "focr_get_matches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "focr_process_hits": (C.c_int, [C.c_void_p, C.c_float, C.c_int32]),
    "focr_total_chars": (C.c_size_t, [C.c_void_p]),
    "focr_total_lines": (C.c_size_t, [C.c_void_p]),
    "focr_get_lines": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "focr_lines_device_chars": (C.c_void_p, [C.c_void_p]),
    "focr_last_timings": (C.c_int, [C.c_void_p, C.c_void_p]),
    "focr_last_counters": (C.c_int, [C.c_void_p, C.c_void_p]),
    "focr_sync": (C.c_int, [C.c_void_p]),
    "focr_ctx_set_scan_cus": (C.c_int, [C.c_void_p, C.c_uint]),
    "focr_ctx_set_prefilter": (C.c_int, [C.c_void_p, C.c_int]),
    "focr_debug_force_split": (C.c_int, [C.c_void_p, C.c_int]),
    "focr_debug_set_tail_grid": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "focr_debug_phase_stamps": (C.c_int, [C.c_void_p, C.c_void_p]),
    "focr_debug_set_stats_form": (C.c_int, [C.c_void_p, C.c_int]),
    "focr_debug_planes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "focr_ctx_set_size_estimates": (C.c_int, [C.c_void_p, C.c_int]),
    "focr_size_estimate_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "focr_ctx_set_column_drop": (C.c_int, [C.c_void_p, C.c_int]),
    "focr_ctx_set_row_tail": (C.c_int, [C.c_void_p, C.c_int]),
    "focr_debug_plane_value": (None, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
    "focr_debug_plane_value_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
    "focr_debug_prefilter": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32,
                                       C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "focr_pipe_create": (C.c_int, [C.c_int, C.c_uint, C.POINTER(C.c_void_p)]),
    "focr_pipe_create2": (C.c_int, [C.c_int, C.c_uint, C.c_uint, C.POINTER(C.c_void_p)]),
    "focr_pipe_destroy": (None, [C.c_void_p]),
    "focr_pipe_lanes": (C.c_uint, [C.c_void_p]),
    "focr_pipe_announce_last": (C.c_int, [C.c_void_p]),
    "focr_pipe_ticket_times": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "focr_pipe_contexts": (C.c_uint, [C.c_void_p]),
    "focr_pipe_context": (C.c_void_p, [C.c_void_p, C.c_uint]),
    "focr_pipe_bank_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "focr_pipe_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_float,
                                   C.c_uint32, C.c_int, C.c_int, C.c_float, C.c_int32, C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_uint64)]),
    "focr_pipe_prefetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]),
    "focr_pipe_end_of_stream": (C.c_int, [C.c_void_p]),
    "focr_pipe_set_fetch": (C.c_int, [C.c_void_p, C.c_int]),
    "focr_pipe_host_results": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "focr_get_lines_into": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "focr_pipe_wait": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "focr_pipe_release": (C.c_int, [C.c_void_p, C.c_uint64]),
    "focr_fleet_create": (C.c_int, [C.POINTER(C.c_int), C.c_uint, C.c_uint, C.POINTER(C.c_void_p)]),
    "focr_fleet_destroy": (None, [C.c_void_p]),
    "focr_fleet_devices": (C.c_uint, [C.c_void_p]),
    "focr_fleet_lanes": (C.c_uint, [C.c_void_p]),
    "focr_fleet_slots": (C.c_uint, [C.c_void_p]),
    "focr_fleet_announce_last": (C.c_int, [C.c_void_p]),
    "focr_fleet_pipe": (C.c_void_p, [C.c_void_p, C.c_uint]),
    "focr_fleet_device_of": (C.c_int, [C.c_void_p, C.c_uint64]),
    "focr_fleet_bank_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "focr_fleet_set_fetch": (C.c_int, [C.c_void_p, C.c_int]),
    "focr_fleet_end_of_stream": (C.c_int, [C.c_void_p]),
    "focr_fleet_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_float, C.c_uint32,
                                    C.c_int, C.c_int, C.c_float, C.c_int32, C.POINTER(C.c_uint64)]),
    "focr_fleet_wait": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "focr_fleet_host_results": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "focr_fleet_release": (C.c_int, [C.c_void_p, C.c_uint64]),
    "focr_last_launches": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "focr_debug_rnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
}

HOST_SYMBOLS = {
    "focr_bank_free": (None, [C.POINTER(BankStruct)]),
    "focr_bank_save": (C.c_int, [C.c_char_p, C.POINTER(BankStruct)]),
    "focr_bank_load": (C.c_int, [C.c_char_p, C.POINTER(BankStruct)]),
    "focr_image_load_luma8": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]),
    "focr_image_probe": (C.c_int, [C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]),
    "focr_image_load_luma8_into": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                             C.c_char_p, C.c_size_t]),
    "focr_image_save_pgm": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "focr_synth_page": (C.c_size_t, [C.POINTER(BankStruct), C.c_uint64, C.c_size_t, C.c_size_t, C.c_void_p,
                                     C.c_void_p, C.c_size_t]),
    "focr_format_f32": (C.c_size_t, [C.c_float, C.c_char_p, C.c_size_t]),
    "focr_line_text": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_char_p, C.c_size_t]),
}

RASTER_SYMBOLS = {
    "focr_raster_bank": (C.c_int, [C.c_char_p, C.c_float, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t,
                                   C.c_int, C.c_uint32, C.c_uint32, C.POINTER(BankStruct), C.c_char_p, C.c_size_t]),
    "focr_font_metrics": (C.c_int, [C.c_char_p, C.c_void_p, C.c_char_p, C.c_size_t]),
}

# include/focr_rccl.h (libfocr_rccl.so)
RCCL_SYMBOLS = {
    "focr_gather_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "focr_gather_destroy": (None, [C.c_void_p]),
    "focr_gather_last_error": (C.c_char_p, []),
    "focr_gather_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t]),
}

_cache = {}


def _load(name, symbols, mode=C.DEFAULT_MODE):
    if name in _cache:
        return _cache[name]
    path = os.path.join(LIB_DIR, name)
    # FOCR_HOST_LIB_DIR: an alternative build of the CPU-side libraries (make asan), for the sanitizer test run
    if name == "libfocr_hip.so" and os.environ.get("FOCR_HIP_LIB"):  # an experiment build of the HIP library (tools/)
        path = os.environ["FOCR_HIP_LIB"]
    alt = os.environ.get("FOCR_HOST_LIB_DIR")
    if alt and name != "libfocr_hip.so" and os.path.exists(os.path.join(alt, name)):
        path = os.path.join(alt, name)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build the native libraries first "
            f"(python -c 'import __graft_entry__ as g; g.build()' or make -C font_ocr_amd/csrc). "
            f"There is no CPU fallback."
        )
    lib = C.CDLL(path, mode=mode)
    experiment = name == "libfocr_hip.so" and bool(os.environ.get("FOCR_HIP_LIB"))
    for sym, (restype, argtypes) in symbols.items():
        if experiment and not hasattr(lib, sym):
            continue  # an A/B build of another commit (tools/): it may predate a debug symbol; the product library must export every one
        fn = getattr(lib, sym)  # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    _cache[name] = lib
    return lib


def hip():
    """libfocr_hip.so — the HIP product path.  Raises if it is not built."""
    return _load("libfocr_hip.so", HIP_SYMBOLS)


def host():
    return _load("libfocr_host.so", HOST_SYMBOLS)


def raster():
    return _load("libfocr_raster.so", RASTER_SYMBOLS)


def rccl():
    """libfocr_rccl.so — the match-list gather over RCCL for one process driving several GPUs (loads librccl)."""
    return _load("libfocr_rccl.so", RCCL_SYMBOLS)
