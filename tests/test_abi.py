"""The C-ABI library loads and exports every symbol include/focr_ncc.h / focr_host.h declare (no GPU needed),
and the product path fails loudly without a device (no CPU fallback)."""
import os
import re
import subprocess

import pytest

from font_ocr_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b((?:focr|ncc)_[a-z0-9_]+)\s*\(", src))


@pytest.fixture(scope="module")
def hip_lib():
    path = os.path.join(N.LIB_DIR, "libfocr_hip.so")
    if not os.path.exists(path):
        import __graft_entry__ as g

        g.build()
    return N.hip()


def test_hip_library_exports_every_declared_symbol(hip_lib):
    declared = _declared("focr_ncc.h")
    assert {"ncc_8_u8", "ncc_16_u8", "focr_scan", "focr_process_hits"} <= declared
    assert declared == set(N.HIP_SYMBOLS), declared ^ set(N.HIP_SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(N.LIB_DIR, "libfocr_hip.so")], capture_output=True,
                         text=True, check=True).stdout
    exported = set(re.findall(r" T ((?:focr|ncc)_\w+)", out))
    assert declared <= exported, declared - exported


def test_host_libraries_export_every_declared_symbol():
    declared = _declared("focr_host.h")
    bound = set(N.HOST_SYMBOLS) | set(N.RASTER_SYMBOLS)
    assert declared == bound, declared ^ bound
    N.host()
    N.raster()


def test_rccl_library_exports_every_declared_symbol():
    """include/focr_rccl.h: checked with nm only — loading the library pulls in librccl, which the CPU tests avoid."""
    declared = _declared("focr_rccl.h")
    assert declared == set(N.RCCL_SYMBOLS), declared ^ set(N.RCCL_SYMBOLS)
    path = os.path.join(N.LIB_DIR, "libfocr_rccl.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "font_ocr_amd", "csrc"), "rccl"], check=True)
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    assert declared <= set(re.findall(r" T (focr_\w+)", out))


def test_no_cpu_fallback_without_device(hip_lib):
    from font_ocr_amd.searcher import FocrError, Scanner

    if hip_lib.focr_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(FocrError, match="no CPU fallback"):
        Scanner()


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under font_ocr_amd/ may reference it."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "font_ocr_amd")):
        if os.sep + "lib" in dp or os.sep + "bin" in dp:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"\boracle\b", txt, flags=re.I):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
