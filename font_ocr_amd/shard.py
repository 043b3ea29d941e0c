"""Page sharding across ranks + gather of the match lists (the only collective of the path).

Pages are independent units (the reference parallelises over them with rayon, src/ncc.rs:839-847), so ranks take
contiguous blocks of the page set, scan them with no data-path communication, and one variable-length gather
brings the post-processed characters to rank 0 in page order.  The same code runs over RCCL ("nccl" backend,
device tensors) and over gloo (CPU tensors; used by the world_size-2 tests).
"""
import numpy as np

from .bank import HIT_DTYPE


def shard_range(n_pages, rank, world):
    """Contiguous block [first, last) of rank `rank`: page p lives on rank p // ceil(n_pages / world)."""
    per = -(-n_pages // world)
    first = min(n_pages, rank * per)
    return first, min(n_pages, first + per)


class CharGather:
    """The gather of one rank's flat uint8 tensors (HIT_DTYPE records) to rank 0, step after step, on buffers allocated once:
    per gather one exchange of the sizes (all_gather of one int64 + ONE device->host read for all ranks), one staging copy and
    the collective — no allocation, no fill, no concatenation on the way (a batch's scan is in flight beside it; every extra
    small kernel on the collective's stream queues for a compute unit behind the other batches' kernels).

    start(mine) issues the exchange and returns finish(); finish() waits for it and returns, on rank 0, (parts, sizes): parts[r]
    is a view of rank r's bytes (valid until the second start() after this one: two buffer sets alternate, so one gather may be
    in flight while the next is issued), elsewhere None.  Works over RCCL (device tensors) and gloo (CPU tensors)."""

    def __init__(self, rank, world, device, capacity=1 << 20):
        import torch

        self.rank, self.world, self.dev = rank, world, torch.device(device)
        self.cuda = self.dev.type == "cuda"
        self.n = torch.zeros(1, device=self.dev, dtype=torch.int64)
        self.all_n = torch.zeros(world, device=self.dev, dtype=torch.int64)
        self.host_n = torch.zeros(world, dtype=torch.int64, pin_memory=True) if self.cuda else None
        self.host_mine = torch.zeros(1, dtype=torch.int64, pin_memory=True) if self.cuda else None
        self.cap = 0
        self.k = 0
        self._grow(capacity)

    def _grow(self, capacity):
        import torch

        self.cap = int(capacity)
        self.buf = [torch.empty(self.cap, dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self.out = [[torch.empty(self.cap, dtype=torch.uint8, device=self.dev) for _ in range(self.world)] if self.rank == 0 else None
                    for _ in range(2)]

    def start(self, mine, async_op=True):
        import torch
        import torch.distributed as dist

        m = int(mine.numel())
        if self.cuda:  # the size from page-locked host memory: no allocation, no blocking copy
            self.host_mine[0] = m
            self.n.copy_(self.host_mine, non_blocking=True)
        else:
            self.n[0] = m
        dist.all_gather_into_tensor(self.all_n, self.n)
        if self.cuda:  # one device->host read for all ranks' sizes, waited for with the GIL released
            self.host_n.copy_(self.all_n, non_blocking=True)
            torch.cuda.current_stream(self.dev).synchronize()
            sizes = self.host_n.tolist()
        else:
            sizes = self.all_n.tolist()
        mx = max(max(sizes), 1)
        if mx > self.cap:  # every rank sees the same sizes, so every rank grows at the same gather
            if self.cuda:
                torch.cuda.synchronize(self.dev)  # the other buffer set may still be in flight
            self._grow(mx + mx // 4)
        s = self.k & 1
        self.k += 1
        buf = self.buf[s][:mx]
        if m:
            buf[:m].copy_(mine, non_blocking=self.cuda)  # (gloo: the exchange reads host memory, so a device source is copied synchronously)
        out = [o[:mx] for o in self.out[s]] if self.rank == 0 else None
        work = dist.gather(buf, out, dst=0, async_op=async_op)

        def finish():
            if async_op:
                work.wait()
            if self.rank != 0:
                return None
            return [out[r][: sizes[r]] for r in range(self.world)], sizes

        return finish


def gather_chars(mine, rank, world, device=None, async_op=False):
    """Gather every rank's flat uint8 tensor (HIT_DTYPE records) to rank 0 -> one concatenated tensor in rank
    order on rank 0, None elsewhere.  `mine` may live on the GPU (nccl) or the CPU (gloo).  One-shot form of CharGather.

    async_op=True returns a zero-argument `finish()` instead: the gather is in flight on the collective's own
    stream (it reads a private copy of `mine`), so the next batch's scan can overlap it; call finish() to wait and
    get the result."""
    import torch

    g = CharGather(rank, world, mine.device if device is None else device, capacity=max(int(mine.numel()), 1))
    fin = g.start(mine, async_op=async_op)

    def finish():
        got = fin()
        if got is None:
            return None
        parts, _ = got
        return torch.cat(parts)

    return finish if async_op else finish()


def chars_from_bytes(t):
    """uint8 tensor (CPU) -> HIT_DTYPE array."""
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=HIT_DTYPE)
