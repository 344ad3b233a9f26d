#!/usr/bin/env python3
"""Headline benchmark: 2048 x 2048 uint16 slices/s destriped + achieved HBM GB/s (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (filter_stripes semantics, production parameters,
uint16 in -> uint16 out) over a batch of 256 synthetic striped 2048 x 2048 slices that is already
resident in HBM.  N > 1: launched by torch.distributed.run, one rank per GPU; slices shard
embarrassingly (weak scaling, 256 slices per rank), the only collective is an RCCL broadcast of the
filter-constant blob before the timed region.  torch is used here for the process group only
(barrier, max-reduce of the time, that broadcast); the engine itself is the C-ABI HIP library.
"""

import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from aind_smartspim_destripe_amd import engine as eng_mod  # noqa: E402
from aind_smartspim_destripe_amd import synth  # noqa: E402

H = W = 2048
ALGO_BYTES_PER_SLICE = H * W * 2 * 2  # compulsory traffic: uint16 read + uint16 write
HBM_PEAK_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def init_dist(n_gpus):
    """Process group for N > 1 (RCCL for device tensors, gloo for host tensors)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and "RANK" not in os.environ:
        return None, 0, 1, 0  # plain `python bench.py`: no process group needed
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local)
    global _HOST_COLLECTIVES
    try:
        dist.init_process_group(backend="cpu:gloo,cuda:nccl", rank=rank, world_size=world)
    except Exception as e:  # pragma: no cover - environment dependent
        log("[bench] mixed backend init failed ({}); falling back to nccl".format(e))
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
        _HOST_COLLECTIVES = False
    return dist, rank, world, local


_HOST_COLLECTIVES = True  # the group has a gloo side: barriers / time reduction stay off the device


def host_all_reduce(dist, values, op=None):
    """All-reduce of a few scalars over the ranks (gloo when the group has it, else through the device)."""
    import torch

    t = torch.tensor(values, dtype=torch.float64)
    if not _HOST_COLLECTIVES:
        t = t.cuda()
    dist.all_reduce(t, op=op or dist.ReduceOp.SUM)
    return [float(x) for x in t.cpu()]


def broadcast_constants(dist, engine, rank):
    """RCCL broadcast (root 0) of the filter-constant blob: twiddles + per-level gain tables.

    Every rank has already planned the same constants, so a failing collective (a node without working
    xGMI / RCCL) is logged and does not stop the data path, which never communicates."""
    import ctypes

    import torch

    ptr, nbytes = engine.constants_device()
    try:
        buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        lib, ctx = engine._lib, engine._ctx
        if rank == 0:
            lib.dsx_memcpy_d2d(ctx, ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(ptr), nbytes)
            engine.sync()
        dist.broadcast(buf, src=0)
        torch.cuda.synchronize()
        if rank != 0:
            lib.dsx_memcpy_d2d(ctx, ctypes.c_void_p(ptr), ctypes.c_void_p(buf.data_ptr()), nbytes)
            engine.sync()
    except Exception as e:  # pragma: no cover - needs a multi-GPU node
        log("[bench] rank {}: constants broadcast failed ({}); using the locally planned constants".format(rank, e))
        return 0
    return nbytes


def cpu_worker(first, count, start_at):
    """Child process of the CPU baseline: the NumPy oracle over `count` planes on one thread.  Prints
    "start end" (epoch seconds of its timed region).  Never touches the GPU."""
    from oracle import destripe_oracle as orc

    p = synth.synthetic_plane(0, 256, 256)
    orc.filter_stripes(p, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, synth.ZARR_PATH_HIGH_INT)  # warm numpy
    planes = [synth.synthetic_plane(k % 32, H, W) for k in range(first, first + count)]
    while time.time() < start_at:  # common start so that the workers really run side by side
        time.sleep(0.01)
    t0 = time.time()
    for p in planes:
        out = orc.filter_stripes(p, "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, synth.ZARR_PATH_HIGH_INT)
        np.clip(out, 0, 65535).astype(np.uint16)
    print("{:.6f} {:.6f}".format(t0, time.time()), flush=True)


def cpu_baseline(n_planes, procs):
    """The NumPy oracle (a port of the reference algorithm) on the host cores, bounded sample: `procs`
    independent single-thread processes (the reference's execution model, zarr_destriper.py:1151-1165),
    `n_planes` planes in total.  Children are started BEFORE this process initialises the GPU."""
    import subprocess

    procs = max(1, min(procs, n_planes))
    per = n_planes // procs
    start_at = time.time() + 6.0 + 0.6 * per  # imports + warm-up + making the planes
    kids = [
        subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i * per), str(per), repr(start_at)],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        for i in range(procs)
    ]  # fmt: skip
    spans = []
    for k in kids:
        out, _ = k.communicate(timeout=900)
        if k.returncode != 0:
            raise RuntimeError("cpu baseline worker failed")
        t0, t1 = map(float, out.split()[-2:])
        spans.append((t0, t1))
    wall = max(t1 for _, t1 in spans) - min(t0 for t0, _ in spans)
    busy = sum(t1 - t0 for t0, t1 in spans)
    done = per * procs
    return {
        "value": round(done / wall, 4),
        "unit": "slices/s",
        "cores": procs,
        "kind": "port",
        "per_core": round(done / busy, 4),
        "sample": "{} synthetic 2048x2048 uint16 planes ({} per process, {} single-thread processes side by side), "
                  "NumPy oracle filter_stripes + uint16 cast, {:.1f} s wall".format(done, per, procs, wall),
    }  # fmt: skip


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]))
    # Only the JSON line may reach stdout: library banners (Gloo, RCCL) are written to fd 1 directly,
    # so fd 1 is pointed at stderr for the duration of the run and the line goes to the saved fd.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="slices per step per GPU")
    ap.add_argument("--cohort", type=int, default=int(os.environ.get("DSX_COHORT", "256")),
                    help="planes per launch chain (workspace size)")  # fmt: skip
    ap.add_argument("--cpu-planes", type=int, default=None,
                    help="planes of the CPU baseline sample (default: 4 per process; 0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=min(16, os.cpu_count() or 1),
                    help="single-thread oracle processes of the CPU baseline")
    ap.add_argument("--kernel-breakdown", action="store_true", help="one extra untimed step with per-kernel events")
    args = ap.parse_args()

    # CPU baseline first: its workers are spawned, which must not happen once this process holds the GPU
    cpu = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.cpu_planes != 0:
        cpu = cpu_baseline(args.cpu_planes or 4 * args.cpu_procs, args.cpu_procs)
        log("[bench] cpu baseline: {}".format(cpu))

    dist, rank, world, local = init_dist(args.gpus)
    if world != args.gpus:
        log("[bench] WORLD_SIZE {} != --gpus {}; using WORLD_SIZE".format(world, args.gpus))

    engine = eng_mod.DestripeEngine(local)
    info = engine.plan(H, W, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT,
                       max_batch=min(args.cohort, args.batch))  # fmt: skip
    blob_bytes = 0
    if dist is not None:
        blob_bytes = broadcast_constants(dist, engine, rank)

    # synthetic stack: 32 unique planes, slice z = bank[z % 32] rolled by z // 32 rows
    t0 = time.perf_counter()
    bank = synth.synthetic_bank(min(32, args.batch), H, W)
    stack = synth.synthetic_stack(args.batch, H, W, bank=bank)
    log("[bench] rank {} synthetic stack {} in {:.1f} s".format(rank, stack.shape, time.perf_counter() - t0))
    d_in = engine.alloc(stack.nbytes)
    d_out = engine.alloc(stack.nbytes)
    d_cfg = engine.alloc(4 * args.batch)
    d_in.upload(stack)

    def step():
        engine.run_device(d_in, np.uint16, args.batch, d_out, np.uint16, d_cfg)

    def barrier():
        engine.sync()
        if dist is not None:
            import torch

            torch.cuda.synchronize()
            host_all_reduce(dist, [0.0])  # barrier on the gloo side of the group

    for _ in range(args.warmup):
        step()
    barrier()
    t_start = time.perf_counter()
    engine.timer_start()
    for _ in range(args.steps):
        step()
    dev_ms = engine.timer_stop()  # HIP events on the engine stream; also synchronises it
    barrier()
    wall = time.perf_counter() - t_start

    if dist is not None:
        wall, dev_ms = host_all_reduce(dist, [wall, dev_ms], op=dist.ReduceOp.MAX)

    # sanity on the result of the last step: config branch per plane and a checksum
    cfg = d_cfg.download((args.batch,), np.int32)
    out_head = d_out.download((1, H, W), np.uint16)
    n_cells = int(cfg.sum())

    breakdown = None
    if args.kernel_breakdown and rank == 0:
        engine.profile(True)
        step()
        engine.sync()
        breakdown = {k: {"ms": round(v[0], 4), "launches": v[1]} for k, v in engine.profile_read().items()}
        engine.profile(False)

    if rank == 0:
        slices = args.batch * world * args.steps
        value = slices / wall
        ms_per_step = 1e3 * wall / args.steps
        dev_ms_per_step = dev_ms / args.steps
        achieved = args.batch * ALGO_BYTES_PER_SLICE / (dev_ms_per_step * 1e-3) / 1e9
        # HBM bytes per launch chain from the PMC counters (collected with rocprofv3 in separate
        # passes and corrected as MI355X_MICROARCH.md prescribes; see profiles/r1_traffic.json)
        traffic = None
        tpath = os.path.join(REPO, "profiles", "r1_traffic.json")
        if os.path.exists(tpath) and args.batch == 256:
            with open(tpath) as f:
                traffic = int(json.load(f)["hbm_bytes_per_step"])
        result = {
            "metric": "2048x2048 uint16 slices/s destriped",
            "value": round(value, 2),
            "unit": "slices/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "batch of {} x 2048x2048 uint16 slices per GPU, log-space wavelet-FFT destripe "
                            "(filter_stripes semantics, production cells/no-cells configs, high_int 2500), "
                            "uint16 out, inputs resident in HBM".format(args.batch),
                "slices_per_gpu": args.batch,
                "cohort": min(args.cohort, args.batch),
                "sub_cohort_streams": int(os.environ.get("DSX_STREAMS", "4")),
                "levels": info.levels,
                "fft_len": [info.fft_len[i] for i in range(info.levels)],
                "planes_with_cells_config": n_cells,
                "parallelism": "z-sharded x{}".format(world),
                "constants_broadcast_bytes": blob_bytes,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "destripe launch chain (k_dwt_fwd, k_hist, k_otsu, k_rowfilter, k_idwt) over one batch",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": args.batch * ALGO_BYTES_PER_SLICE,
                "device_ms_per_launch": round(dev_ms_per_step, 4),
                "read_only_frac": round(achieved / 2 / HBM_PEAK_GBS, 5),
            },
            "out_checksum": int(out_head.astype(np.uint64).sum()),
        }
        if breakdown is not None:
            result["kernel_ms"] = breakdown
        if cpu is not None:
            result["cpu_baseline"] = cpu
        os.write(json_fd, (json.dumps(result) + "\n").encode())

    d_in.free()
    d_out.free()
    d_cfg.free()
    engine.close()
    if dist is not None:
        host_all_reduce(dist, [0.0])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
