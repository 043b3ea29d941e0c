#!/usr/bin/env python3
"""bench.py — NCC template-matching scan throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic pages that are already resident
in HBM: window statistics + i8-MFMA prefilter + exact verify + ordering/cap (focr_scan) followed by
the on-device anchor/line/overlap pass (focr_process_hits) and, for N > 1, the RCCL gather of the
post-processed match lists to rank 0.  Workload = BASELINE configs[1]: 128 pages of 608x720 per GPU,
95-glyph bank, --x-bits 2 --y-bits 0 (380 templates), threshold 0.8; pages shard across ranks with
no data-path collective other than that final gather (weak scaling: per-GPU work fixed).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python bench.py --gpus N ...                      # launches N ranks itself (one process per GPU) when RANK / WORLD_SIZE are unset
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W        # or under an external launcher
  python bench.py --gpus 8 --config c4              # BASELINE configs[3]: 8192 pages sharded over the ranks + RCCL gather

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the MFMA prefilter launch with the
most work), timed live with HIP events on the stream it runs on; `cpu_baseline` times the reference's
own AVX2 kernel (oracle/_ref, kind "reference") or, if that artefact is absent, the CPU restatement
(kind "port") on a bounded sample of the same pages on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_W, R_H = 608, 720
PEAK_I8_MFMA_TOPS = 5000.0  # dense i8 MFMA = 2x the ~2.5 PFLOP/s bf16 dense peak (MI355X_MICROARCH.md, Matrix cores)


def traffic_of(kernel_name, workload):
    """HBM-side bytes per launch of the dominant kernel, taken from committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in
    separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md) — only when a profile exists for exactly this kernel (name
    AND source hash) and this workload (pages per launch, geometry, templates); otherwise None: a live run cannot read PMC
    counters, and a number measured for another kernel or workload would be stale.  Returns (bytes, source file) or (None, None)."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            t = json.load(open(path))
        except (OSError, ValueError):
            continue
        if t.get("kernel") == kernel_name and all(t.get("workload_key", {}).get(k) == v for k, v in workload.items()):
            # ... and for exactly this kernel BODY: the profile records a hash of the kernel's sources (a changed kernel with the same
            # name and workload would otherwise inherit a stale number)
            import hashlib

            h = hashlib.sha256()
            try:
                for src in t.get("kernel_sources", []):
                    h.update(open(os.path.join(ROOT, src), "rb").read())
            except OSError:
                continue
            if t.get("kernel_source_sha16") != h.hexdigest()[:16]:
                continue
            return t.get("traffic_bytes"), os.path.relpath(path, ROOT)
    return None, None


def effective_cpus():
    """CPUs this process can really use: the affinity mask, capped by the container's CFS quota (cgroup v2 cpu.max)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def self_launch(args):
    """`--gpus N` (N > 1) without a launcher's RANK / WORLD_SIZE: start the N ranks here, one child process per GPU, before
    anything in this process has touched the GPU (a process that has initialised HIP must not exec or fork GPU users).
    Rank 0's JSON line is this process's stdout.  The run FAILS FAST: all children are polled, the first one to exit non-zero
    ends the run within seconds — the others are terminated (a rank waiting in a rendezvous or a collective for a dead peer
    would otherwise sit there until the backend's own timeout), its exit code is returned and its last lines are repeated on
    stderr.  Every line a rank writes to stderr (or, for ranks > 0, stdout) reaches this process's stderr as `[rank k] ...`.
    The rendezvous is a file store in a private temporary directory (FOCR_BENCH_INIT): no port to reserve and lose."""
    import shutil
    import signal
    import subprocess
    import tempfile
    import threading

    if not args.dry_launch:
        import torch  # counting devices does not initialise the GPU

        have = torch.cuda.device_count()
        if have < args.gpus and not (os.environ.get("FOCR_BENCH_SHARE_GPU") and have >= 1):
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but this machine shows {have} GPU(s); refusing to measure fewer GPUs than asked for\n")
            return 2
    store_dir = tempfile.mkdtemp(prefix="focr_bench_")
    procs, tails, pumps, out0 = [], [], [], []

    def pump(stream, rank, keep, sink):
        for raw in iter(stream.readline, b""):
            line = raw.decode(errors="replace").rstrip("\n")
            keep.append(line)
            del keep[:-40]
            if sink is not None:
                sink.append(line)
            else:
                sys.stderr.write(f"[rank {rank}] {line}\n")
                sys.stderr.flush()
        stream.close()

    def stop_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)
                except OSError:
                    pass

    def on_signal(signum, _frame):  # the driver's own time limit: never leave ranks behind
        stop_all(signal.SIGTERM)
        time.sleep(0.5)
        stop_all(signal.SIGKILL)
        shutil.rmtree(store_dir, ignore_errors=True)
        os._exit(128 + signum)

    old_handlers = {s_: signal.signal(s_, on_signal) for s_ in (signal.SIGTERM, signal.SIGINT)}
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), FOCR_BENCH_INIT="file://" + os.path.join(store_dir, "store"),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), PYTHONUNBUFFERED="1")
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE,
                                 stderr=subprocess.PIPE if r == 0 else subprocess.STDOUT)
            procs.append(p)
            tails.append([])
            if r == 0:  # rank 0: stdout is the JSON line (kept), stderr is forwarded
                pumps.append(threading.Thread(target=pump, args=(p.stdout, r, [], out0), daemon=True))
                pumps.append(threading.Thread(target=pump, args=(p.stderr, r, tails[r], None), daemon=True))
            else:
                pumps.append(threading.Thread(target=pump, args=(p.stdout, r, tails[r], None), daemon=True))
        for t_ in pumps:
            t_.start()
        failed = None
        while failed is None and any(p.poll() is None for p in procs):
            for r, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = r
                    break
            else:
                time.sleep(0.05)
        if failed is None:
            failed = next((r for r, p in enumerate(procs) if p.returncode != 0), None)
        if failed is not None:
            rc = procs[failed].returncode
            stop_all(signal.SIGTERM)
            t_end = time.time() + 5.0
            while time.time() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.05)
            stop_all(signal.SIGKILL)
            for p in procs:
                p.wait()
            for t_ in pumps:
                t_.join(timeout=2.0)
            sys.stderr.write(f"bench.py: rank {failed} exited with code {rc}; the other ranks were terminated. Its last lines:\n")
            for line in tails[failed][-15:]:
                sys.stderr.write(f"    [rank {failed}] {line}\n")
            sys.stderr.flush()
            return rc if 0 < rc < 256 else 1
        for t_ in pumps:
            t_.join(timeout=5.0)
        sys.stdout.write("".join(line + "\n" for line in out0))
        sys.stdout.flush()
        return 0
    finally:
        for s_, h in old_handlers.items():
            signal.signal(s_, h)
        shutil.rmtree(store_dir, ignore_errors=True)


def rendezvous_kwargs():
    """init_process_group arguments common to the real and the dry run: the self-launcher's private file store when it set one
    (FOCR_BENCH_INIT), else the launcher's MASTER_ADDR / MASTER_PORT; a SHORT timeout, so that a rank whose peer never arrives
    fails (and with it the run, see self_launch) instead of waiting for the backend's default of many minutes."""
    import datetime

    kw = {"timeout": datetime.timedelta(seconds=int(os.environ.get("FOCR_BENCH_INIT_TIMEOUT", "300")))}
    if os.environ.get("FOCR_BENCH_INIT"):
        kw["init_method"] = os.environ["FOCR_BENCH_INIT"]
    else:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
    return kw


def injected_failure(rank, stage):
    """FOCR_BENCH_FAIL_RANK=k[:init|step] (tests/test_bench_launch.py): rank k dies at that stage, so that the launcher's
    fail-fast path can be exercised without breaking a GPU."""
    spec = os.environ.get("FOCR_BENCH_FAIL_RANK", "")
    if not spec:
        return
    k, _, st = spec.partition(":")
    if int(k) == rank and (st or "init") == stage:
        sys.stderr.write(f"bench.py: injected failure in rank {rank} at stage '{stage}' (FOCR_BENCH_FAIL_RANK)\n")
        sys.stderr.flush()
        os._exit(17)


def dry_run(args, rank, world, real_stdout):
    """`--dry-launch`: the launcher, the rendezvous, the barrier-bracketed timing, the MAX over ranks and the rank census on
    the gloo backend with a stub step — no GPU, no scan.  What tests/test_bench_launch.py runs on the CPU; never a measurement."""
    import torch
    import torch.distributed as dist

    injected_failure(rank, "init")
    if os.environ.get("FOCR_BENCH_DRY_CHATTER"):
        print(f"dry-launch rank {rank} of {world}", file=sys.stderr, flush=True)
    dist.init_process_group("gloo", rank=rank, world_size=world, **rendezvous_kwargs())
    P = args.pages_per_gpu
    for _ in range(args.warmup):
        time.sleep(0.001)
    dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if k == args.steps // 2:
            injected_failure(rank, "step")
        time.sleep(0.001 * (1 + rank))  # ranks differ: the job's time is the slowest rank's
    dt_local = time.perf_counter() - t0
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    seen = torch.tensor([1], dtype=torch.int64)
    dist.all_reduce(seen, op=dist.ReduceOp.SUM)
    mine = torch.tensor([P * R_W * R_H * args.steps / dt_local / 1e6], dtype=torch.float64)
    every = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(every, mine)
    if rank == 0:
        out = {"metric": "Mpixels/s scanned (95-glyph x --x-bits=2 bank)", "value": round(world * P * R_W * R_H * args.steps / dt / 1e6, 2), "unit": "Mpx/s",
               "n_gpus": world, "ranks_seen": int(seen.item()), "per_rank_value": [round(float(v.item()), 2) for v in every], "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "none", "data": "none", "dry_launch": True,
               "config": {"workload": "DRY LAUNCH: stub step on the gloo backend, no GPU — exercises the rank launcher and the timing protocol only"}}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    dist.destroy_process_group()


def main():
    # ROCm maps a process's streams onto 4 hardware queues by default; three contexts + the gather's streams + RCCL's
    # are more than that, and a small copy sharing a queue with a context waits behind its 2.5 ms scan kernel
    # (measured: the gather path at 17.6 instead of 21.0 Gpx/s).  Must be set before the HIP runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300,
                    help="timed steps (default 300 = 0.5 s: the timed region starts and ends with a drained pipeline, barrier + synchronise on both "
                         "sides, and filling + draining it costs one batch latency, 5 ms — 3 %% of 100 steps, 1 %% of 300)")
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--pages-per-gpu", type=int, default=128)
    ap.add_argument("--config", choices=["c2", "c3", "c4"], default="c2",
                    help="c2 = BASELINE configs[1] (608x720, 380 templates; the headline workload, weak scaling); c3 = configs[2] geometry "
                         "(1200x1600 pages, --x-bits 2 --y-bits 2 = 1520 templates); c4 = configs[3]: 8192 pages of 608x720 in all, "
                         "sharded over the ranks in contiguous blocks (strong scaling): one step = every rank scans its 8192 / N pages "
                         "once, as batches of --pages-per-gpu pages taken from its HBM-resident shard")
    ap.add_argument("--c4-pages", type=int, default=8192, help="total pages of --config c4 (BASELINE configs[3]: 8192)")
    ap.add_argument("--c3-pages", type=int, default=0,
                    help="--config c3 as a STREAM (BASELINE configs[2] names 1024 pages): that many different 1200x1600 pages, sharded over the ranks, "
                         "each rank's block HBM-resident and scanned batch by batch — every batch is new to the size estimates; "
                         "0 (default) = rescan the resident batches every step")
    ap.add_argument("--mode", choices=["mfma", "direct"], default="mfma")
    ap.add_argument("--threshold", type=float, default=0.8)
    ap.add_argument("--prefilter", choices=["auto", "one", "legacy"], default="auto",
                    help="MFMA prefilter kernel (focr_ctx_set_prefilter): auto = one = threshold planes + scan_mfma2s_kernel; "
                         "legacy = round 1's kernel and int32 threshold tables")
    ap.add_argument("--tail", choices=["hits", "legacy"], default="hits",
                    help="tail of the MFMA scan (focr_ctx_set_row_tail): hits = verify in flush order, then bucket + sort the hits (default); "
                         "legacy = round 2's radix-sort tail")
    ap.add_argument("--legacy-tail", action="store_true",
                    help="round 2's tail (library radix sort of all candidates + verify + compaction) instead of the per-row sort + verify (focr_ctx_set_row_tail(0)), for A/B")
    ap.add_argument("--no-column-drop", action="store_true",
                    help="multiply every template column in the MFMA (focr_ctx_set_column_drop(0)): round 2's 3-K-step form of the 9-wide classes, for A/B")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-pages", type=int, default=0, help="0 = 4 pages per host thread")
    ap.add_argument("--noise", action="store_true", help="uniform-random pages instead of synthetic text (worst case: nothing to prune, no hits)")
    ap.add_argument("--with-upload", action="store_true", help="also time steps that start from PAGEABLE host pages, upload and scan back to back (e2e_value_incl_h2d)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the short untimed leg that measures the pipelined PCIe-inclusive rate (e2e_value_incl_h2d_pipelined)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the two optional legs behind the default measurement: c3_value (BASELINE configs[2]'s geometry and 1520-template bank, 64 resident "
                         "1200x1600 pages per batch) and c4_stream_value (2048 DIFFERENT 608x720 pages streamed through the executor in batches of 128: "
                         "every batch is new to its context's size estimates).  Never part of `value`")
    ap.add_argument("--dry-launch", action="store_true", help="launcher / timing-protocol self-test on the gloo backend with a stub step: no GPU, not a measurement")
    ap.add_argument("--settle-s", type=float, default=0.8, help="untimed extra warm-up (seconds of steps) before the timed region")
    ap.add_argument("--force-gather", action="store_true", help="run the RCCL gather path even with one rank (self-test)")
    ap.add_argument("--scan-cus", type=int, default=-1,
                    help="CUs the persistent scan kernel occupies; 0 = all, -1 = auto: all with one batch in flight, else 7/8 "
                         "of them (the rest is left to the other contexts' small kernels; flat optimum 192..224 of 256: DESIGN.md section 5)")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="lanes of the executor (focr_pipe_*): that many HIP streams run batches side by side, so one batch's statistics / "
                         "verify / sort / ordering kernels overlap another's MFMA scan; 1 = strictly one batch at a time on the device")
    ap.add_argument("--depth", type=int, default=2,
                    help="contexts per lane: a lane's next batch is queued on the device behind the one that runs (each context holds its own "
                         "resident batch); lanes x depth batches are outstanding at the host")
    ap.add_argument("--inject-stall-ms", type=float, default=0.0,
                    help="self-test of the executor's tolerance: the submitting thread sleeps this long every --inject-stall-every timed steps")
    ap.add_argument("--inject-stall-every", type=int, default=10)
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    # Exactly ONE line may reach stdout (the JSON).  RCCL prints a version banner to stdout at init, so park
    # the real stdout and point fd 1 at stderr for everything else (Python and native code alike).
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    # several host threads drive the GPU (one per context in flight, one for the gather); their Python sections are
    # tiny, but a thread that needs the GIL back after a blocking call must not wait the default 5 ms for it
    sys.setswitchinterval(5e-5)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FOCR_BENCH_SHARE_GPU=1: a REHEARSAL of the N-rank protocol on a one-GPU box — every rank uses device 0 (the line says so and
    # is never a scaling measurement: the ranks share one chip)
    share_gpu = bool(os.environ.get("FOCR_BENCH_SHARE_GPU")) and world > 1
    if share_gpu:
        local_rank = 0
    if world != args.gpus:  # never measure another number of GPUs than the one asked for
        os.write(2, f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}\n".encode())
        raise SystemExit(2)
    if args.dry_launch:
        return dry_run(args, rank, world, real_stdout)

    import torch
    import torch.distributed as dist

    from font_ocr_amd import Bank, synth_pages
    from font_ocr_amd.bank import HIT_DTYPE
    from font_ocr_amd.searcher import SCAN_DIRECT, SCAN_MFMA, Scanner
    from font_ocr_amd.shard import CharGather

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_gather
    injected_failure(rank, "init")
    if use_dist:
        # RCCL's kernels on a high-priority stream: the collective's few workgroups must not queue behind the thousands of short
        # workgroups of the batches' small kernels for a CU (the scan leaves an eighth of the chip to all of them)
        pg_opts = None
        try:
            pg_opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except Exception:  # noqa: BLE001 - an older torch: default priority
            pg_opts = None
        if share_gpu:  # RCCL refuses two ranks on one device ("Duplicate GPU detected"): the rehearsal's collectives run over gloo on host tensors
            dist.init_process_group("gloo", rank=rank, world_size=world, **rendezvous_kwargs())
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, pg_options=pg_opts, **rendezvous_kwargs())
    coll_dev = torch.device("cpu") if share_gpu else dev  # where the collectives' tensors live

    mode = SCAN_MFMA if args.mode == "mfma" else SCAN_DIRECT
    global R_W, R_H
    bank_file = "bank_dejavu13_ascii95_x2.bin"
    if args.config == "c3":
        R_W, R_H = 1200, 1600
        bank_file = "bank_dejavu13_ascii95_x2y2.bin"
    bank = Bank.load(os.path.join(ROOT, "tests", "golden", bank_file))
    P = args.pages_per_gpu
    n_lanes = max(1, args.in_flight)
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    scan_cus = args.scan_cus if args.scan_cus >= 0 else (0 if n_lanes == 1 else n_cus - n_cus // 8)
    # The executor (focr_pipe_*, include/focr_ncc.h): n_lanes streams, --depth contexts each; every batch is queued on the device
    # the moment it is submitted, so n_ctx = lanes x depth batches are outstanding at the host
    from font_ocr_amd.searcher import Pipeline

    pipe = Pipeline(local_rank, n_lanes, max(1, args.depth))
    n_ctx = len(pipe.scanners)
    for c_ in pipe.scanners:
        if args.no_column_drop:
            c_.set_column_drop(False)
        if args.legacy_tail:
            args.tail = "legacy"
        c_.set_row_tail({"hits": 1, "legacy": 0}[args.tail])
        if os.environ.get("FOCR_BENCH_TAIL_GRID"):  # experiments: the tail's persistent kernels on num/den times their workgroups (focr_debug_set_tail_grid)
            num_, den_ = os.environ["FOCR_BENCH_TAIL_GRID"].split("/")
            c_.set_tail_grid(int(num_), int(den_))
    pipe.set_bank(bank)
    scs, pages = [], None
    shard = None  # --config c4: the rank's contiguous block of the page set, resident in HBM
    stream_pages = args.c4_pages if args.config == "c4" else (args.c3_pages if args.config == "c3" else 0)
    if stream_pages:
        from font_ocr_amd.shard import shard_range

        first, last = shard_range(stream_pages, rank, world)
        n_mine = last - first
        shard_batches = [(b0, min(P, n_mine - b0)) for b0 in range(0, n_mine, P)]
        host = synth_pages(bank, n_mine, R_W, R_H, first=first)
        pages = host[:P]
        shard = torch.from_numpy(host).to(dev)
        del host
        for j in range(n_ctx):
            c_ = pipe.scanners[j]
            c_.set_scan_cus(scan_cus)
            c_.set_prefilter({"auto": 0, "one": 1, "legacy": 3}[args.prefilter])
            scs.append(c_)
    for j in range(n_ctx if shard is None else 0):  # every rank (and every context of it) scans its own shard of the page set
        if args.noise:
            pg = np.random.default_rng(1234 + rank * n_ctx + j).integers(0, 256, (P, R_H, R_W), dtype=np.uint8)
        else:
            pg = synth_pages(bank, P, R_W, R_H, first=(rank * n_ctx + j) * P)
        if j == 0:
            pages = pg
        c_ = pipe.scanners[j]
        c_.set_scan_cus(scan_cus)
        c_.set_prefilter({"auto": 0, "one": 1, "legacy": 3}[args.prefilter])
        # inputs resident in HBM before the timed region: pages go up as a torch tensor, then device->device ingest
        d_pages = torch.from_numpy(pg).to(dev)
        c_.alloc_pages(P, R_W, R_H)
        c_.upload_pages_device(d_pages.data_ptr(), P, 0, invert=True)
        c_.sync()
        del d_pages
        scs.append(c_)
    sc = scs[0]

    class _DevBytes:  # zero-copy view of the library's device buffer for torch (no host round trip)
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    from collections import deque
    import queue
    import threading

    jobs = deque()  # tickets of the steps in flight, oldest first
    # FOCR_BENCH_TRACE=1: host-side time stamps of the submitting thread, the gather thread and the garbage collector (us since
    # t_trace0), printed to stderr at the end — what does not perturb a timing-dependent stall the way a profiler does
    trace_on = bool(os.environ.get("FOCR_BENCH_TRACE"))
    t_trace0 = time.perf_counter()
    trace_ev = []

    def stamp(who, what, t_a, t_b=None):
        if trace_on:
            trace_ev.append((who, what, (t_a - t_trace0) * 1e6, ((t_b if t_b is not None else t_a) - t_trace0) * 1e6))

    # The Python garbage collector stops the SUBMITTING thread: a full (generation 2) collection of this process's heap (torch,
    # numpy, ctypes: ~10^6 tracked objects) takes tens of milliseconds, during which nothing is submitted and every lane runs
    # dry.  FOCR_BENCH_GC: "freeze" (default) = collect once, then gc.freeze() before the timed region, so that later collections
    # only walk the few objects the steps create; "observe" = leave the collector alone; either way the longest pause inside
    # the timed region is measured and reported (host_gc).
    import gc

    gc_mode = os.environ.get("FOCR_BENCH_GC", "freeze")
    gc_t = {}
    gc_pauses = []  # (generation, start, end), perf_counter seconds

    def on_gc(phase, info):
        if phase == "start":
            gc_t[info["generation"]] = time.perf_counter()
        else:
            t_e = time.perf_counter()
            t_s = gc_t.get(info["generation"], t_e)
            gc_pauses.append((info["generation"], t_s, t_e))
            stamp("gc", f"generation {info['generation']} collected {info.get('collected', 0)}", t_s, t_e)

    gc.callbacks.append(on_gc)

    def run_step(c_):  # one pass of the hot path over one resident batch, synchronously on one context
        c_.scan(args.threshold, 1024, mode)
        c_.process_hits(0.95, 5)

    # The only collective of the path: the RCCL gather of the post-processed characters (variable length, device
    # resident) to rank 0.  retire() copies a finished step's characters out of its context (which is then released
    # for its next batch); one dedicated thread issues the collectives, strictly in step order, so every rank issues
    # them in the same order; each gather is asynchronous on RCCL's stream and is waited for one step later.
    gather_q = queue.Queue()
    gathered = {"chars": 0, "err": None}
    gather_dbg = [0.0, 0.0, 0]  # seconds issuing gathers, seconds finishing the previous ones, gathers
    main_dbg = [0.0, 0.0, 0.0, 0]  # the submitting thread: seconds waiting for the oldest batch, for a free output slot, inside submit; submits

    def gather_worker():
        torch.cuda.set_device(local_rank)  # the current device is per thread
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))  # off the legacy null stream; high priority, as RCCL's
        gather = CharGather(rank, world, coll_dev, capacity=slot_bytes)  # every buffer of the exchange allocated once
        pending = []
        while True:
            item = gather_q.get()
            try:
                def finish_oldest():
                    fin, slot = pending.pop(0)
                    got = fin()
                    torch.cuda.current_stream(dev).synchronize()  # the gather has read the slot
                    slot_free[slot].set()
                    gathered["chars"] = sum(got[1]) // HIT_DTYPE.itemsize if rank == 0 else 0

                if item is None:  # drain request
                    while pending:
                        finish_oldest()
                else:
                    slot, nbytes = item
                    t_a = time.perf_counter()
                    pending.append((gather.start(out_bufs[slot][:nbytes]), slot))
                    t_b = time.perf_counter()
                    while len(pending) > 1:
                        finish_oldest()
                    t_c = time.perf_counter()
                    gather_dbg[0] += t_b - t_a
                    gather_dbg[1] += t_c - t_b
                    gather_dbg[2] += 1
                    stamp("gather", f"issue slot {slot}", t_a, t_b)
                    stamp("gather", "finish previous", t_b, t_c)
            except Exception as e:  # noqa: BLE001 - reported by fence()
                gathered["err"] = e
                for ev_ in slot_free:  # never leave the submitter waiting
                    ev_.set()
            finally:
                gather_q.task_done()

    if use_dist:
        # device slots the lanes copy their characters into (two per lane, so a slot is rewritten 2 * n_ctx steps later)
        slot_bytes = max(1 << 20, 2 * P * (R_W // 6) * (R_H // 12) * HIT_DTYPE.itemsize)  # twice a page full of text
        out_bufs = [torch.empty(slot_bytes, dtype=torch.uint8, device=dev) for _ in range(2 * n_ctx)]
        slot_free = [threading.Event() for _ in out_bufs]
        for ev_ in slot_free:
            ev_.set()
        slot_of, next_slot = {}, [0]
        threading.Thread(target=gather_worker, daemon=True).start()

    first_page_is_seed0 = rank == 0  # rank 0's first page is synthetic page 0 = tests/golden/c2_page0.npz
    kern = {}
    phase = {}
    n_chars = 0
    timed = False
    ticket_log = []  # focr_pipe_ticket_times of every batch retired inside the timed region

    def retire():
        """Consume the oldest step in flight: its results stay on the device; with several ranks they are gathered."""
        nonlocal n_chars
        t = jobs.popleft()
        t_w = time.perf_counter()
        c_ = pipe.wait(t)
        main_dbg[0] += time.perf_counter() - t_w
        stamp("main", f"wait ticket {t}", t_w, time.perf_counter())
        t_r = time.perf_counter()
        # the same calls in the untimed steps as in the timed ones (their results kept only from the latter): on a fresh box the
        # first call of a library path pages its code in from the image — milliseconds, and a 20-step region is 35 of them
        kern_, phase_ = (kern, phase) if timed else ({}, {})
        tt = pipe.ticket_times(t)
        if timed:
            if os.environ.get("FOCR_BENCH_DUMP_TICKETS"):  # + where the batch's phases lie on the device's clock
                tt.update(c_.phase_stamps())
                tt["ticket"] = t
            ticket_log.append(tt)
        for li in c_.launches():  # launches of one kernel over different bank chunks are different launches: key by their work too
            k = kern_.setdefault((li["name"], li["alg_macs"]), dict(ms=0.0, n=0, alg=li["alg_macs"], issued=li["issued_macs"]))
            k["ms"] += li["ms"]
            k["n"] += 1
        for k_, v in c_.timings().items():
            phase_[k_] = phase_.get(k_, 0.0) + v
        if use_dist:  # the lane already copied the characters into its output slot: free it for its next batch
            slot = slot_of.pop(t)
            nbytes = c_.total_chars() * HIT_DTYPE.itemsize
            pipe.release(t)
            gather_q.put((slot, nbytes))
            stamp("main", f"retire ticket {t} (launches, timings, total_chars, release, queue the gather)", t_r, time.perf_counter())
        else:
            n_chars = c_.total_chars() or n_chars
            pipe.release(t)

    def step(k, last=False):
        """last: nothing is submitted behind this step until the pipeline has drained (the end of a timed / settling sequence): said
        BEFORE the submit (focr_pipe_announce_last) — batches are queued on the device the moment they are submitted."""
        if shard is not None:  # one step = the rank's whole shard, batch by batch, ingested device -> device from the resident tensor
            for i, (b0, nb) in enumerate(shard_batches):
                if last and i + 1 == len(shard_batches) and not os.environ.get("FOCR_BENCH_NO_EOS"):
                    pipe.announce_last()
                submit_one(device_ptr=shard[b0].data_ptr(), shape=(nb, R_H, R_W))
        else:
            if last and not os.environ.get("FOCR_BENCH_NO_EOS"):
                pipe.announce_last()
            submit_one()

    host_gap = {"last": None, "max": 0.0, "waited": 0.0}  # the submitting thread's own time between two submits (waiting for a batch excluded)

    def submit_one(device_ptr=None, shape=None):
        if len(jobs) == n_ctx:  # the context this batch maps to still holds the batch submitted n_ctx submissions ago
            w0 = main_dbg[0]
            retire()
            host_gap["waited"] += main_dbg[0] - w0
        now_ = time.perf_counter()
        if timed and host_gap["last"] is not None:
            host_gap["max"] = max(host_gap["max"], now_ - host_gap["last"] - host_gap["waited"])
        host_gap["last"], host_gap["waited"] = now_, 0.0
        if use_dist:
            slot = next_slot[0] % len(out_bufs)
            next_slot[0] += 1
            t_s = time.perf_counter()
            slot_free[slot].wait()  # its previous gather (2 * n_ctx steps ago) has read it
            slot_free[slot].clear()
            t_u = time.perf_counter()
            t = pipe.submit(None, args.threshold, 1024, mode, True, 0.95, 5, chars_out=(out_bufs[slot].data_ptr(), out_bufs[slot].numel()),
                            device_ptr=device_ptr, shape=shape)
            main_dbg[1] += t_u - t_s
            main_dbg[2] += time.perf_counter() - t_u
            main_dbg[3] += 1
            stamp("main", f"slot {slot} free", t_s, t_u)
            stamp("main", f"submit ticket {t}", t_u, time.perf_counter())
            slot_of[t] = slot
        else:
            t = pipe.submit(None, args.threshold, 1024, mode, True, 0.95, 5, device_ptr=device_ptr, shape=shape)
        jobs.append(t)

    def fence(barrier=True):
        nonlocal n_chars
        while jobs:
            retire()
        if use_dist:
            gather_q.put(None)
            gather_q.join()
            if gathered["err"] is not None:
                raise gathered["err"]
            n_chars = gathered["chars"] or n_chars
        for c_ in scs:
            c_.sync()
        torch.cuda.synchronize()
        if use_dist and barrier:
            dist.barrier()
            torch.cuda.synchronize()

    # Untimed: the W warm-up steps, then — still untimed — more of the same steps until ~0.8 s of device work has gone by:
    # the first scan of every context reads its result sizes synchronously (later ones run on those sizes), buffers grow to
    # their steady size, and the GPU's clocks take a few hundred ms under load to settle; a 20-step timed region (~50 ms)
    # measured right after 5 steps was 8 % below a 100-step one.
    for k in range(args.warmup):
        step(k, last=k + 1 == args.warmup)
    fence()
    if gc_mode == "freeze":
        # HERE, not between the settling steps and the timed region: a full collection of this heap idles the GPU for 35-50 ms,
        # its clocks fall back, and a short timed region right behind it runs its first scans 8 % slower (20 steps: 29.9 against
        # 32.2 Gpx/s at 300 steps, the same box; tools/r4_k20.sh) — the steps below bring the clocks back before the clock starts
        gc.collect()
        gc.freeze()
    t_settle = time.perf_counter()
    go = args.settle_s > 0
    while go:
        for k in range(2 * n_ctx):
            step(k, last=k + 1 == 2 * n_ctx)
        fence()
        go = time.perf_counter() - t_settle < args.settle_s
        if use_dist and world > 1:  # every rank must run the same number of rounds: each one ends in collectives (fence)
            flag = torch.tensor([1 if go else 0], device=coll_dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            go = bool(flag.item())
    timed = True
    host_gap["last"] = None
    t0 = time.perf_counter()
    n_stalls = 0
    for k in range(args.steps):
        if k == args.steps // 2:
            injected_failure(rank, "step")
        if args.inject_stall_ms > 0 and k and k % max(1, args.inject_stall_every) == 0:
            time.sleep(args.inject_stall_ms / 1e3)  # a host that is late with its next batch: the device works on what is queued
            n_stalls += 1
        step(k, last=k + 1 == args.steps)
    fence(barrier=False)
    dt_own = time.perf_counter() - t0  # this rank's own work done (per_rank_value); the job's time includes the barrier
    fence()
    dt = time.perf_counter() - t0
    timed = False
    gc_in = [(g_, b_ - a_) for g_, a_, b_ in gc_pauses if a_ < t0 + dt and b_ > t0]
    host_gc = {"mode": gc_mode, "collections_in_timed_region": len(gc_in), "longest_pause_ms": round(max([d_ for _, d_ in gc_in] or [0.0]) * 1e3, 3),
               "total_pause_ms": round(sum(d_ for _, d_ in gc_in) * 1e3, 3)}
    dt_local = dt_own
    ranks_seen, per_rank = 1, None
    if use_dist:
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        seen = torch.tensor([1], device=coll_dev, dtype=torch.int64)  # rank census over the nccl group: the line is for exactly this many GPUs
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        ranks_seen = int(seen.item())

    total_px = (stream_pages if shard is not None else world * P) * R_W * R_H * args.steps
    value = total_px / dt / 1e6
    if use_dist:  # every rank's own rate (its pages over its own time)
        my_pages = (n_mine if shard is not None else P) * args.steps
        mine = torch.tensor([my_pages * R_W * R_H / dt_local / 1e6], device=coll_dev, dtype=torch.float64)
        every = [torch.zeros(1, device=coll_dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [round(float(v.item()), 2) for v in every]
    # what the size estimates did during the timed region: the bench rescans the same resident batches every step, so the
    # estimates are at their best case (smallest margin, nothing redone); --config c4 streams different batches
    est = [c_.size_estimate_stats() for c_ in scs]
    e2e = e2e_pipe = None
    e2e_steps = 0
    if args.with_upload and shard is None:  # PCIe-inclusive, pageable host memory, upload and scan back to back on one context
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            sc.upload_pages(pages, 0, invert=True)
            run_step(sc)
        fence()
        e2e = total_px / (time.perf_counter() - t1) / 1e6
    leg_errors = {}  # an optional leg (PCIe-inclusive rate, isolated kernel, CPU baseline) that fails must not take the headline line with it

    def measure_e2e():
        """SURVEY.md section 8(d) asks for both timings: device-resident (`value`) and end to end.  A short extra leg outside the
        timed region: every step starts from PAGE-LOCKED HOST memory; the same contexts / host threads as the headline run; batch
        k + n_ctx is announced (focr_pipe_prefetch: its DMA starts on a copy stream of its lane) right after batch k is submitted, so
        every copy crosses PCIe under the scans of the batches in flight.  Never the headline value."""
        from font_ocr_amd.searcher import PinnedPages

        pins = []
        for j in range(n_ctx):
            pin = PinnedPages(P, R_H, R_W)
            pin.array[:] = pages
            pins.append(pin)

        def pipe_steps(n):
            tickets = deque()
            ahead = min(n_ctx, n)
            for k in range(ahead):
                pipe.prefetch(pins[k % n_ctx].array)
            for k in range(n):
                if len(tickets) == n_ctx:
                    t = tickets.popleft()
                    pipe.wait(t)
                    pipe.release(t)
                if k + 1 == n:
                    pipe.announce_last()
                tickets.append(pipe.submit(pins[k % n_ctx].array, args.threshold, 1024, mode, True, 0.95, 5))
                if k + ahead < n:
                    pipe.prefetch(pins[(k + ahead) % n_ctx].array)
            while tickets:
                t = tickets.popleft()
                pipe.wait(t)
                pipe.release(t)

        pipe_steps(120)  # 0.2 s: page-locking the source buffers left the GPU idle, and its clocks take that long to come back
        fence()
        n_steps = 240  # 0.4 s whatever --steps says: fill + drain of the pipeline are 1-2 % of it
        t1 = time.perf_counter()
        pipe_steps(n_steps)
        fence()
        rate = P * R_W * R_H * n_steps / (time.perf_counter() - t1) / 1e6
        for pin in pins:
            pin.close()
        return rate, n_steps

    if not args.no_e2e and shard is None and not use_dist:
        try:
            e2e_pipe, e2e_steps = measure_e2e()
        except Exception as e:  # noqa: BLE001 - reported in the line; the device-resident measurement stands
            leg_errors["e2e_value_incl_h2d_pipelined"] = repr(e)
            e2e_pipe = None
    counters = sc.counters()

    # the dominant kernel alone on the chip (no other batch in flight, all CUs): a short extra leg outside the timed
    # region, reported beside the in-flight figure so that both ways of reading "kernel duration" are on the table
    iso = {}
    if rank == 0 and n_lanes > 1 and not leg_errors:  # (a failed PCIe leg leaves the lanes in an unknown state)
        try:
            sc.set_scan_cus(0)
            for i in range(14):  # 4 to settle (the leg starts from an idle GPU), 10 measured
                run_step(sc)
                if i >= 4:
                    for li in sc.launches():
                        k = iso.setdefault((li["name"], li["alg_macs"]), dict(ms=0.0, n=0))
                        k["ms"] += li["ms"]
                        k["n"] += 1
            sc.set_scan_cus(scan_cus)
        except Exception as e:  # noqa: BLE001
            leg_errors["frac_isolated"] = repr(e)
            iso = {}

    if trace_on and rank == 0:
        t_lo, t_hi = (t0 - t_trace0) * 1e6, (t0 + dt - t_trace0) * 1e6
        print(f"[trace] timed region {t_lo:.0f} .. {t_hi:.0f} us; events longer than 3 ms inside it:", file=sys.stderr)
        for who, what, a_, b_ in trace_ev:
            if b_ - a_ > 3000 and a_ < t_hi and b_ > t_lo:
                print(f"[trace]   {who:7s} {a_:10.0f} .. {b_:10.0f} ({(b_ - a_) / 1e3:7.2f} ms)  {what}", file=sys.stderr)
        if os.environ.get("FOCR_BENCH_TRACE") == "2":
            for who, what, a_, b_ in trace_ev:
                if a_ < t_hi and b_ > t_lo:
                    print(f"[trace-all] {who:7s} {a_:10.0f} .. {b_:10.0f} ({b_ - a_:8.0f} us)  {what}", file=sys.stderr)
    if use_dist and gather_dbg[2] and rank == 0:
        print(f"[bench] gather thread per gather: issue {gather_dbg[0] / gather_dbg[2] * 1e3:.3f} ms (sizes exchange incl. its host read, staging copies, "
              f"the collective's launch), waiting for the previous one {gather_dbg[1] / gather_dbg[2] * 1e3:.3f} ms; {gather_dbg[2]} gathers; "
              f"torch allocator: {torch.cuda.memory_stats(dev).get('num_device_alloc', -1)} device allocations, "
              f"{torch.cuda.memory_stats(dev).get('num_device_free', -1)} frees; submitting thread per step: waiting for the oldest batch "
              f"{main_dbg[0] / max(main_dbg[3], 1) * 1e3:.3f} ms, for a free output slot {main_dbg[1] / max(main_dbg[3], 1) * 1e3:.3f} ms, in submit "
              f"{main_dbg[2] / max(main_dbg[3], 1) * 1e3:.3f} ms", file=sys.stderr)
    out = {
        "metric": "Mpixels/s scanned (95-glyph x --x-bits=2 bank)",
        "value": round(value, 2),
        "unit": "Mpx/s",
        "n_gpus": world,
        "ranks_seen": ranks_seen,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong" if shard is not None else "weak",
        "vs_baseline": None,
        "dtype": "i8",
        "data": "synthetic",
        "config": {
            "workload": (f"BASELINE configs[1]: batches of {P} synthetic 608x720 pages, 95-glyph DejaVu Sans Mono 13px bank, "
                         "--x-bits 2 --y-bits 0 (380 templates), threshold 0.8, cap 1024, + process_hits(0.95, 5)") if args.config == "c2"
                        else (f"BASELINE configs[3]: {args.c4_pages} synthetic 608x720 pages sharded over {world} rank(s) in contiguous blocks, each "
                              f"rank's block HBM-resident and scanned in batches of {P}; 380 templates, threshold 0.8, cap 1024, "
                              "+ process_hits(0.95, 5), RCCL gather of the match lists") if args.config == "c4"
                        else (f"BASELINE configs[2]: {stream_pages} synthetic 1200x1600 pages sharded over {world} rank(s), each rank's block HBM-resident and "
                              f"scanned in batches of {P} (different pages every batch); 95-glyph bank, --x-bits 2 --y-bits 2 (1520 templates, 16 sub-pixel "
                              "shifts), threshold 0.8, cap 1024, + process_hits(0.95, 5)") if stream_pages
                        else (f"BASELINE configs[2] geometry: batches of {P} synthetic 1200x1600 pages, 95-glyph bank, --x-bits 2 --y-bits 2 "
                              "(1520 templates, 16 sub-pixel shifts), threshold 0.8, cap 1024, + process_hits(0.95, 5)"),
            "pages_per_batch": P,
            "resident_pages_per_gpu": P * n_ctx,
            "batches_in_flight": n_lanes,
            "contexts_per_lane": n_ctx // n_lanes,
            "batches_outstanding_at_host": n_ctx,
            "scan_cus": scan_cus or n_cus,
            "templates": len(bank),
            "scan_mode": args.mode,
            "prefilter": args.prefilter,
            "column_drop": not args.no_column_drop,
            "tail": {"hits": "hits-first rows", "legacy": "legacy radix sort"}[args.tail],
            "parallelism": (f"REHEARSAL: {world} ranks on one GPU, gloo gather of match lists" if share_gpu else
                            f"pages sharded over {world} rank(s), RCCL gather of match lists" if world > 1 else "single GPU"),
        },
    }
    if e2e is not None:
        out["e2e_value_incl_h2d"] = round(e2e, 2)
    if e2e_pipe is not None:
        out["e2e_value_incl_h2d_pipelined"] = round(e2e_pipe, 2)
        out["e2e_note"] = (f"extra untimed leg of {e2e_steps} steps: every batch starts in page-locked host memory and crosses PCIe (one DMA per batch, "
                           "announced with focr_pipe_prefetch as many batches ahead as there are lanes) under the scans of the batches in flight; "
                           "`value` is the device-resident rate")
    out["size_estimates"] = {"batches_redone_exact": sum(e["redone"] for e in est), "margin": est[0]["margin"], "largest_page_row": est[0]["row_max"],
                             "note": "every step rescans the same resident batches, so the result-size estimates run at their smallest margin "
                                     "with nothing redone; different batches per step: --config c4"}
    out["host_gc"] = host_gc
    if ticket_log:
        # Where the timed region's time went, ticket by ticket (focr_pipe_ticket_times: host stamps of the executor + the device-side
        # interval between consecutive batches' last kernels) — so that ONE late submit or one slow batch can be told from uniform slowness
        def pct(v, q):
            v = sorted(v)
            return v[min(len(v) - 1, int(q * len(v)))] if v else None

        done = [t_["done_us"] for t_ in ticket_log]
        sub = [t_["submit_us"] for t_ in ticket_log]
        host_iv = [(b_ - a_) / 1e3 for a_, b_ in zip(done, done[1:])]
        dev_iv = [t_["device_gap_ms"] for t_ in ticket_log[1:] if t_["device_gap_ms"] >= 0]
        sub_iv = [(b_ - a_) / 1e3 for a_, b_ in zip(sub, sub[1:])]
        lead = [(t_["done_us"] - t_["enqueue_end_us"]) / 1e3 for t_ in ticket_log]  # how long a batch sat fully queued before its results were seen
        if os.environ.get("FOCR_BENCH_DUMP_TICKETS"):  # every retired batch's stamps, for a look at one run's rhythm
            with open(os.environ["FOCR_BENCH_DUMP_TICKETS"], "w") as f_:
                json.dump(ticket_log, f_)
        out["step_stats"] = {
            "batches": len(ticket_log),
            "first_completion_ms": round((done[0] - sub[0]) / 1e3, 3),
            "completion_interval_ms_p50": round(pct(host_iv, 0.5), 4) if host_iv else None,
            "completion_interval_ms_max": round(max(host_iv), 4) if host_iv else None,
            "device_interval_ms_p50": round(pct(dev_iv, 0.5), 4) if dev_iv else None,
            "device_interval_ms_max": round(max(dev_iv), 4) if dev_iv else None,
            "longest_submit_gap_ms": round(max(sub_iv), 4) if sub_iv else None,
            "longest_host_gap_ms": round(host_gap["max"] * 1e3, 4),
            "executor_queueing_ms_p50": round(pct([(t_["enqueue_end_us"] - t_["enqueue_begin_us"]) / 1e3 for t_ in ticket_log], 0.5), 4),
            "executor_pickup_ms_max": round(max((t_["enqueue_begin_us"] - t_["submit_us"]) / 1e3 for t_ in ticket_log), 4),
            "queued_ahead_ms_min": round(min(lead), 3),
            "injected_host_stalls": n_stalls,
            "injected_stall_ms": args.inject_stall_ms,
            "note": "per batch retired in the timed region: host-side completion intervals (when the retiring thread saw each batch done), device-side "
                    "intervals between consecutive batches' last kernels (HIP events), the longest pause between two submits (with / without the time the "
                    "submitting thread waited for the oldest batch: 'longest_host_gap_ms' is the host's own lateness), the executor thread's time "
                    "to queue one batch and its longest delay in picking one up, and the least time a batch had been fully queued before it completed",
        }
    if per_rank is not None:
        out["per_rank_value"] = per_rank
    if args.noise:
        out["data"] = "uniform random noise pages (worst case, no hits)"
    if rank == 0:
        # dominant kernel = the scan launch with the most algorithmic work
        key, k = max(kern.items(), key=lambda kv: kv[1]["alg"])
        name = key[0]
        avg_s = k["ms"] / k["n"] / 1e3
        achieved = 2.0 * k["alg"] / avg_s / 1e12
        # SURVEY.md section 8(d) algorithmic bytes per launch: every page pixel once + 8 B per emitted match
        alg_bytes = P * R_W * R_H + 8 * counters["raw_hits"]
        traffic, traffic_src = (None, None) if args.noise else traffic_of(name, {"pages": P, "r_w": R_W, "r_h": R_H, "templates": len(bank)})
        out["roofline"] = {
            "bound": "mfma",
            "kernel": name,
            "achieved": round(achieved, 2),
            "peak": PEAK_I8_MFMA_TOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_I8_MFMA_TOPS, 4),
            "traffic": traffic,
            "traffic_source": traffic_src,
            "traffic_over_algorithmic_bytes": round(traffic / alg_bytes, 2) if traffic else None,
            "avg_kernel_ms": round(k["ms"] / k["n"], 4),
            "algorithmic_macs_per_launch": k["alg"],
            "issued_macs_per_launch": k["issued"],
            "frac_issued": round(2.0 * k["issued"] / avg_s / 1e12 / PEAK_I8_MFMA_TOPS, 4),
            "note": "int8 ops (2 per MAC) over true template area x searched windows; peak = dense i8 MFMA; 'frac_issued' = the MACs the "
                    "MFMA pipes were actually given (dropped ninth column and blank tiles not counted, padding counted) over the same duration; "
                    "compulsory HBM traffic is 1 B/px (hbm_frac below), the path is MFMA-bound (SURVEY.md 8d)",
            "hbm_frac_compulsory": round(value * 1e6 * 1.0 / 8.0e12, 8),
        }
        if n_lanes > 1:
            # per-launch durations stretch when launches of several contexts share the chip; two more readings:
            step_alg = sum(v["alg"] * v["n"] for v in kern.values()) / args.steps  # algorithmic MACs per step, all scan launches
            out["roofline"]["frac_whole_step"] = round(2.0 * step_alg / (dt / args.steps) / 1e12 / PEAK_I8_MFMA_TOPS, 4)
            if key in iso:
                iso_ms = iso[key]["ms"] / iso[key]["n"]
                out["roofline"]["isolated_avg_kernel_ms"] = round(iso_ms, 4)
                out["roofline"]["frac_isolated"] = round(2.0 * k["alg"] / (iso_ms / 1e3) / 1e12 / PEAK_I8_MFMA_TOPS, 4)
            out["roofline"]["note"] += (f"; {n_lanes} batches in flight: 'achieved'/'frac' use the per-launch duration inside the timed region "
                                        "(launches of different contexts overlap, so it is longer than the kernel alone), "
                                        "'frac_isolated' = same kernel alone on all CUs (extra untimed leg), "
                                        "'frac_whole_step' = algorithmic ops of one step / wall time of one step")
        out["phases_ms_per_step"] = {k_: round(v / args.steps, 4) for k_, v in phase.items()}
        per_name = {}
        for (n_, _alg), v in kern.items():
            per_name[n_] = per_name.get(n_, 0.0) + v["ms"] / args.steps
        out["kernels_ms_per_step"] = {n_: round(v, 4) for n_, v in per_name.items()}
        out["work"] = {"candidates": counters["candidates"], "raw_hits": counters["raw_hits"], "chars_out": int(n_chars)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import oracle as O  # test infrastructure: the CPU baseline leg is allowed to use it

            threads = min(effective_cpus(), 32)
            S = args.cpu_sample_pages or min(P, 4 * threads)
            inv = (255 - pages[:S]).astype(np.uint8)
            use_ref = O.have_ref()
            passes = 2  # ~15 s of CPU work on the GPU box's 32 threads
            t0 = time.perf_counter()
            for _ in range(passes):
                total, cpu_counts, _ = O.scan_pages_mt(inv, bank, args.threshold, 1024, use_ref=use_ref, threads=threads)
            cdt = (time.perf_counter() - t0) / passes
            t0 = time.perf_counter()
            O.scan_pages_mt(inv[:2], bank, args.threshold, 1024, use_ref=use_ref, threads=1)  # single-core figure, 2 pages
            cdt1 = time.perf_counter() - t0
            out["cpu_baseline"] = {
                "value": round(S * R_W * R_H / cdt / 1e6, 4),
                "unit": "Mpx/s",
                "cores": threads,
                "kind": "reference" if use_ref else "port",
                "sample": f"{S} of the same synthetic pages x {len(bank)} templates, {passes} passes, one page per thread "
                          f"(window tables + kernel calls as src/ncc.rs:231-404), {cdt * passes:.1f} s wall, {int(total)} raw hits per pass",
                "value_1_core": round(2 * R_W * R_H / cdt1 / 1e6, 4),
            }
        except Exception as e:  # noqa: BLE001 - the reference's CPU rate is a reported baseline, not the measurement
            leg_errors["cpu_baseline"] = repr(e)

    if rank == 0 and args.config in ("c2", "c4") and not args.noise and first_page_is_seed0:
        # parity in the same run: page 0 of this very batch against the committed reference golden (tests/golden/c2_page0.npz)
        try:
            g = np.load(os.path.join(ROOT, "tests", "golden", "c2_page0.npz"))
            if shard is not None:  # c4: the context holds whichever batch it scanned last; put the shard's first batch back
                sc.set_pages(pages)
            sc.scan(0.8, 1024, mode)
            offs, m = sc.matches()
            mine = m[: int(offs[len(bank)])]
            dev_counts = sc.counts()
            same = np.array_equal(dev_counts[0], g["counts"]) and mine.tobytes() == g["matches"].tobytes()
            out["parity"] = (f"page 0: {len(mine)} raw matches bit-identical to the reference kernel's golden lists" if same
                             else "MISMATCH against tests/golden/c2_page0.npz")
            if same and args.config == "c2" and "cpu_baseline" in out:  # the CPU leg scanned the first S pages of this very batch
                S_ = cpu_counts.shape[0]
                out["parity"] += (f"; per-(page, template) match counts of all {S_} pages of the CPU leg equal the device's ({int(cpu_counts.sum())} matches)"
                                  if np.array_equal(dev_counts[:S_], cpu_counts) else "; MISMATCH of per-page counts against the CPU leg")
        except OSError:
            out["parity"] = "golden fixture not found"

    # Two more workloads for the record, after everything the default line needs: BASELINE configs[2]'s geometry and
    # the streaming form of configs[1].  Optional legs: a failure goes to optional_leg_errors, `value` is never touched.
    def extra_leg(pipe_, n_warm, n_timed, submit_step, px_per_step):
        """n_warm untimed + n_timed timed steps through executor pipe_ (same protocol as the headline: drained on both sides);
        returns (Mpx/s, ms per step, {scan launch -> (avg ms, alg MACs, launches)}, batches redone exact)."""
        slots = len(pipe_.scanners)
        jobs_, kern_ = deque(), {}
        redone0 = sum(c_.size_estimate_stats()["redone"] for c_ in pipe_.scanners)

        def retire_(keep):
            t_ = jobs_.popleft()
            c_ = pipe_.wait(t_)
            if keep:
                for li in c_.launches():
                    k_ = kern_.setdefault((li["name"], li["alg_macs"]), [0.0, 0])
                    k_[0] += li["ms"]
                    k_[1] += 1
            pipe_.release(t_)

        def run_(n, keep):
            for k_ in range(n):
                if len(jobs_) == slots:
                    retire_(keep)
                if k_ + 1 == n:
                    pipe_.announce_last()
                jobs_.append(submit_step(k_))
            while jobs_:
                retire_(keep)
            for c_ in pipe_.scanners:
                c_.sync()

        run_(n_warm, False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_(n_timed, True)
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t1
        redone = sum(c_.size_estimate_stats()["redone"] for c_ in pipe_.scanners) - redone0
        return px_per_step * n_timed / dt_ / 1e6, dt_ / n_timed * 1e3, kern_, redone

    def leg_roofline(kern_, ms_step):
        (name_, alg_), (ms_, n_) = max(kern_.items(), key=lambda kv: kv[0][1])
        step_alg = sum(a_ * v_[1] for (_, a_), v_ in kern_.items()) / max(1, max(v_[1] for v_ in kern_.values()))
        return {"kernel": name_, "avg_kernel_ms": round(ms_ / n_, 4), "frac": round(2.0 * alg_ / (ms_ / n_ / 1e3) / 1e12 / PEAK_I8_MFMA_TOPS, 4),
                "frac_whole_step": round(2.0 * step_alg / (ms_step / 1e3) / 1e12 / PEAK_I8_MFMA_TOPS, 4)}

    if rank == 0 and world == 1 and args.config == "c2" and not args.noise and not args.no_extra_legs and not use_dist and mode == SCAN_MFMA:
        try:  # c4_stream: 2048 different pages, HBM-resident as luma8, ingested device -> device batch by batch (one DMA-free pass per 16 steps)
            n_stream = 2048
            stream_dev = torch.from_numpy(synth_pages(bank, n_stream, R_W, R_H, first=100000)).to(dev)
            nb_ = n_stream // P
            val_, ms_, kern_, redone_ = extra_leg(pipe, nb_, 2 * nb_,
                                                  lambda k_: pipe.submit(None, args.threshold, 1024, mode, True, 0.95, 5, device_ptr=stream_dev[(k_ % nb_) * P].data_ptr(), shape=(P, R_H, R_W)),
                                                  P * R_W * R_H)
            out["c4_stream_value"] = round(val_, 2)
            out["c4_stream"] = dict(leg_roofline(kern_, ms_), ms_per_step=round(ms_, 4), steps=2 * nb_, pages=n_stream, batches_redone_exact=redone_,
                                    note=f"BASELINE configs[3]'s stream on one GPU: {n_stream} different synthetic pages resident in HBM as luma8, scanned in batches of {P} "
                                         "(ingest + scan + process_hits per step), two passes timed behind one untimed pass; every batch differs from the one its context "
                                         "scanned before, so the size estimates run at their working margin (redone = batches scanned a second time with exact sizes)")
            del stream_dev
        except Exception as e:  # noqa: BLE001
            leg_errors["c4_stream_value"] = repr(e)
        try:  # c3: BASELINE configs[2]'s geometry and bank on an executor of its own
            bank3 = Bank.load(os.path.join(ROOT, "tests", "golden", "bank_dejavu13_ascii95_x2y2.bin"))
            P3, W3, H3 = 64, 1200, 1600
            pipe3 = Pipeline(local_rank, n_lanes, max(1, args.depth))
            try:
                pipe3.set_bank(bank3)
                pg3 = torch.from_numpy(synth_pages(bank3, P3, W3, H3, first=200000)).to(dev)
                for c_ in pipe3.scanners:  # the same 64 pages resident in every context (rescanned every step, as the headline does)
                    c_.set_scan_cus(scan_cus)
                    c_.alloc_pages(P3, W3, H3)
                    c_.upload_pages_device(pg3.data_ptr(), P3, 0, invert=True)
                    c_.sync()
                del pg3
                val_, ms_, kern_, redone_ = extra_leg(pipe3, 2 * len(pipe3.scanners), 12,
                                                      lambda k_: pipe3.submit(None, args.threshold, 1024, mode, True, 0.95, 5), P3 * W3 * H3)
                out["c3_value"] = round(val_, 2)
                out["c3"] = dict(leg_roofline(kern_, ms_), ms_per_step=round(ms_, 4), steps=12, pages_per_batch=P3, templates=len(bank3), batches_redone_exact=redone_,
                                 note="BASELINE configs[2]'s geometry: batches of 64 synthetic 1200x1600 pages, 95-glyph bank, --x-bits 2 --y-bits 2 (1520 templates in four "
                                      "size classes: two scan launches per batch, the exact verify in chunk passes), threshold 0.8, + process_hits; 12 timed steps")
            finally:
                pipe3.close()
        except Exception as e:  # noqa: BLE001
            leg_errors["c3_value"] = repr(e)

    if leg_errors:
        out["optional_leg_errors"] = leg_errors
    if share_gpu:
        out["rehearsal"] = f"FOCR_BENCH_SHARE_GPU: all {world} ranks on ONE GPU — the launcher, the barriers, the sharding and the gather of the N-rank run on real hardware; not a scaling measurement"
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    pipe.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
