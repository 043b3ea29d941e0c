"""The N>1 path on CPU: page sharding + the gather of match lists, world_size 2 over gloo.

Each rank produces the post-processed characters of its page shard (here with the CPU oracle — the GPU scan
itself is covered by the gpu tests), the gather brings them to rank 0, and the result must equal the
single-process result for the whole page set."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _chars_of_pages(bank, first, last):
    from font_ocr_amd import synth_page
    from font_ocr_amd.bank import HIT_DTYPE, SYNTH_SEED_BASE
    from oracle import oracle as O

    out = []
    for p in range(first, last):
        page = synth_page(bank, SYNTH_SEED_BASE + 300 + p, 220, 100)
        counts, matches = O.scan_page(O.invert(page), bank, 0.8)
        for line in O.process_hits(O.raw_hits(counts, matches, bank), 0.95, 5):
            h = np.zeros(len(line), HIT_DTYPE)
            for f in ("x", "y", "w", "h", "similarity", "letter"):
                h[f] = line[f]
            h["template_index"] = p  # carry the page index through the gather
            out.append(h)
    return np.concatenate(out) if out else np.zeros(0, HIT_DTYPE)


def _worker(rank, world, port, n_pages, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from font_ocr_amd import Bank
    from font_ocr_amd.shard import chars_from_bytes, gather_chars, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bank = Bank.load(os.path.join(ROOT, "tests", "golden", "bank_dejavu13_ascii95_x2.bin")).subset(range(33, 60))
    first, last = shard_range(n_pages, rank, world)
    mine = _chars_of_pages(bank, first, last)
    t = torch.from_numpy(mine.view(np.uint8).reshape(-1).copy())
    allc = gather_chars(t, rank, world)
    if rank == 0:
        q.put(chars_from_bytes(allc).tobytes())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from font_ocr_amd.shard import shard_range

    for n in (0, 1, 5, 8, 128, 8192):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(a <= b for a, b in spans)


@pytest.mark.timeout(300)
def test_two_rank_gather_equals_single_process():
    import torch.multiprocessing as mp

    from font_ocr_amd import Bank

    n_pages, world = 5, 2  # ragged: 3 + 2 pages
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pages, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    bank = Bank.load(os.path.join(ROOT, "tests", "golden", "bank_dejavu13_ascii95_x2.bin")).subset(range(33, 60))
    want = _chars_of_pages(bank, 0, n_pages)
    assert len(want) > 0
    assert got == want.tobytes()
