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


def gather_chars(mine, rank, world, device=None, async_op=False):
    """Gather every rank's flat uint8 tensor (HIT_DTYPE records) to rank 0 -> one concatenated tensor in rank
    order on rank 0, None elsewhere.  `mine` may live on the GPU (nccl) or the CPU (gloo).

    async_op=True returns a zero-argument `finish()` instead: the gather is in flight on the collective's own
    stream (it reads a private copy of `mine`), so the next batch's scan can overlap it; call finish() to wait and
    get the result."""
    import torch
    import torch.distributed as dist

    dev = mine.device if device is None else device
    n = torch.tensor([mine.numel()], device=dev, dtype=torch.int64)
    all_n = torch.empty(world, device=dev, dtype=torch.int64)
    dist.all_gather_into_tensor(all_n, n)
    if all_n.is_cuda:  # one device->host read for all ranks' sizes, waited for with the GIL released
        host_n = torch.empty(world, dtype=torch.int64, pin_memory=True)
        host_n.copy_(all_n, non_blocking=True)
        torch.cuda.current_stream(all_n.device).synchronize()
        sizes = host_n.tolist()
    else:
        sizes = all_n.tolist()
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=dev)
    buf[: mine.numel()] = mine
    out = [torch.empty(mx, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
    work = dist.gather(buf, out, dst=0, async_op=async_op)

    def finish():
        if async_op:
            work.wait()
        if rank != 0:
            return None
        return torch.cat([out[r][: sizes[r]] for r in range(world)])

    return finish if async_op else finish()


def chars_from_bytes(t):
    """uint8 tensor (CPU) -> HIT_DTYPE array."""
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=HIT_DTYPE)
