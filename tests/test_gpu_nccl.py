"""The device-tensor branch of the match-list gather (font_ocr_amd/shard.py over the "nccl" backend = RCCL) on a real GPU:
a world_size-1 process group in a fresh child process (the backend is initialised before any other GPU call there),
Pipeline.submit(chars_out=...) handing the lanes' characters to gather_chars — asynchronous and synchronous — and the result
compared with what Scanner.lines_flat() reads from the same batches.  tests/test_shard_gloo.py covers world_size 2 on CPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)  # first GPU work of this process
sys.path.insert(0, sys.argv[1])
import numpy as np
from font_ocr_amd import Bank, synth_pages
from font_ocr_amd.bank import HIT_DTYPE
from font_ocr_amd.searcher import SCAN_MFMA, Pipeline
from font_ocr_amd.shard import chars_from_bytes, gather_chars

bank = Bank.load(os.path.join(sys.argv[1], "tests", "golden", "bank_dejavu13_ascii95_x2.bin"))
sub = bank.subset(list(range(33, 80)) + list(range(95 + 33, 95 + 80)))
pipe = Pipeline(0, 2)
pipe.set_bank(sub)
bufs = [torch.empty(4 << 20, dtype=torch.uint8, device=dev) for _ in range(3)]
tickets, want = [], []
batches = [synth_pages(bank, 3, 300, 130, first=40 + 3 * b) for b in range(3)]
pending = []
total = 0
for b, pages in enumerate(batches):
    if len(tickets) == 2:  # a lane is needed: retire the oldest batch
        t0, b0 = tickets.pop(0)
        sc = pipe.wait(t0)
        n = sc.total_chars() * HIT_DTYPE.itemsize
        want.append(sc.lines_flat().copy())
        pipe.release(t0)  # the lane is free; its characters live on in bufs[b0]
        pending.append((gather_chars(bufs[b0][:n], 0, 1, dev, async_op=True), len(want) - 1))
    tickets.append((pipe.submit(pages, 0.8, 1024, SCAN_MFMA, True, 0.95, 5, chars_out=(bufs[b].data_ptr(), bufs[b].numel())), b))
while tickets:
    t0, b0 = tickets.pop(0)
    sc = pipe.wait(t0)
    n = sc.total_chars() * HIT_DTYPE.itemsize
    want.append(sc.lines_flat().copy())
    pipe.release(t0)
    got = gather_chars(bufs[b0][:n], 0, 1, dev)  # synchronous form
    assert chars_from_bytes(got).tobytes() == want[-1].tobytes(), "sync gather differs"
    total += len(want[-1])
for fin, k in pending:
    got = fin()
    torch.cuda.synchronize()
    assert chars_from_bytes(got).tobytes() == want[k].tobytes(), "async gather differs"
    total += len(want[k])
assert total > 300, total
pipe.close()
dist.destroy_process_group()
print("NCCL_GATHER_OK", total)
'''


def test_rccl_gather_of_device_resident_characters():
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL_GATHER_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


NATIVE_CHILD = r'''
import ctypes as C, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
from font_ocr_amd import Bank, synth_pages, _native as N
from font_ocr_amd.bank import HIT_DTYPE
from font_ocr_amd.searcher import SCAN_MFMA, Scanner

R = N.rccl()
g = C.c_void_p()
devs = (C.c_int * 1)(0)
assert R.focr_gather_create(devs, 1, C.byref(g)) == 0, R.focr_gather_last_error()
bank = Bank.load(os.path.join(sys.argv[1], "tests", "golden", "bank_dejavu13_ascii95_x2.bin"))
sub = bank.subset(list(range(33, 80)) + list(range(95 + 33, 95 + 80)))
with Scanner(0) as sc:
    sc.set_bank(sub)
    sc.set_pages(synth_pages(bank, 3, 300, 130, first=70))
    sc.scan(0.8, 1024, SCAN_MFMA)
    sc.process_hits(0.95, 5)
    want = sc.lines_flat().copy()
    ptr, n = sc.device_chars()
    nbytes = n * HIT_DTYPE.itemsize
    dst = torch.zeros(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    src = (C.c_void_p * 1)(ptr)
    sizes = (C.c_size_t * 1)(nbytes)
    assert R.focr_gather_bytes(g, src, sizes, C.c_void_p(dst.data_ptr()), dst.numel()) == 0, R.focr_gather_last_error()
    got = np.frombuffer(dst[:nbytes].cpu().numpy().tobytes(), dtype=HIT_DTYPE)
    assert got.tobytes() == want.tobytes() and len(want) > 100
    assert R.focr_gather_bytes(g, src, sizes, C.c_void_p(dst.data_ptr()), nbytes - 1) != 0  # destination too small
R.focr_gather_destroy(g)
print("NATIVE_GATHER_OK", len(want))
'''


def test_native_rccl_gather_single_process():
    """libfocr_rccl.so (include/focr_rccl.h): ncclCommInitAll + grouped ncclSend / ncclRecv of a context's device-resident
    characters to rank 0 — the single-process multi-GPU form of the path's one collective, here with the one GPU the
    test box has (rank 0 sends to itself)."""
    r = subprocess.run([sys.executable, "-c", NATIVE_CHILD, ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NATIVE_GATHER_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
