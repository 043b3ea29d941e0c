import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kernel_cases():
    z = np.load(os.path.join(GOLD, "kernel_cases.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    return {n: {k.split("/")[1]: z[k] for k in z.files if k.startswith(n + "/")} for n in names}


@pytest.fixture(scope="session")
def bank_x2():
    from font_ocr_amd import Bank

    return Bank.load(os.path.join(GOLD, "bank_dejavu13_ascii95_x2.bin"))


@pytest.fixture(scope="session")
def bank_x2y2():
    from font_ocr_amd import Bank

    return Bank.load(os.path.join(GOLD, "bank_dejavu13_ascii95_x2y2.bin"))


@pytest.fixture(scope="session")
def bank_default():
    from font_ocr_amd import Bank

    return Bank.load(os.path.join(GOLD, "bank_dejavu13_default_x0.bin"))


@pytest.fixture(scope="session")
def c1_golden():
    return dict(np.load(os.path.join(GOLD, "c1_page.npz")))


@pytest.fixture(scope="session")
def c2_golden():
    return dict(np.load(os.path.join(GOLD, "c2_page0.npz")))


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the CPU-side native pieces once (host libs + oracle); the HIP library is built by
    __graft_entry__.build() and is required only by the gpu tests and the symbol test."""
    import subprocess

    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "font_ocr_amd", "csrc"), "host", "raster"], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
