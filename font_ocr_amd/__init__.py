"""font_ocr_amd — MI355X-native NCC template-matching scan (the `ncc` hot path of aconz2/font-ocr).

HIP kernels + C ABI live in csrc/ (built in-tree into lib/); this package is the thin host-side
mirror used by tests and bench.py.  See DESIGN.md.
"""
from .bank import ASCII95, DEFAULT_ALPHABET, Bank, load_image, save_pgm, synth_page, synth_pages  # noqa: F401
