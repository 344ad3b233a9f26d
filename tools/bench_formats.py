#!/usr/bin/env python3
"""Device bandwidth of the format kernels (rows f1 / f3) beside their NumPy restatements.

    python tools/bench_formats.py [--planes 256] [--reps 20]

Prints one JSON line per kernel: algorithmic bytes (read + write of the uint16 data) / HIP-event time,
against the 8 TB/s HBM peak, and the NumPy time for a bounded sample of the same work on one host core.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from aind_smartspim_destripe_amd import engine as eng_mod  # noqa: E402

PEAK = 8000.0


def timed(eng, fn, reps):
    for _ in range(3):
        fn()
    eng.timer_start()
    for _ in range(reps):
        fn()
    return eng.timer_stop() / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--planes", type=int, default=256)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cpu-planes", type=int, default=16)
    args = ap.parse_args()
    from oracle import format_oracle as fo

    Z, H, W = args.planes, 2048, 2048
    brick = (64, 128, 128)
    eng = eng_mod.DestripeEngine(0)
    vol = np.random.RandomState(0).randint(0, 65536, (Z, H, W)).astype(np.uint16)
    d_a, d_b = eng.alloc(vol.nbytes), eng.alloc(vol.nbytes)
    d_c = eng.alloc(vol.nbytes // 8)
    d_a.upload(vol)
    rows = []

    def cpu(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    sample = vol[: args.cpu_planes]
    cases = [
        ("k_planes_to_bricks<8>", lambda: eng.planes_to_bricks(d_a, d_b, (Z, H, W), brick), 2 * vol.nbytes,
         lambda: fo.planes_to_bricks(sample, (args.cpu_planes, 128, 128)), 2 * sample.nbytes),
        ("k_bricks_to_planes<8>", lambda: eng.bricks_to_planes(d_b, d_a, (Z, H, W), brick), 2 * vol.nbytes,
         lambda: fo.bricks_to_planes(fo.planes_to_bricks(sample, (args.cpu_planes, 128, 128)), sample.shape),
         4 * sample.nbytes),
        ("k_downsample2<true>", lambda: eng.downsample2(d_a, d_c, (Z, H, W)), vol.nbytes + vol.nbytes // 8,
         lambda: fo.windowed_mean_u16(sample), sample.nbytes + sample.nbytes // 8),
    ]  # fmt: skip
    for name, fn, nbytes, cpu_fn, cpu_bytes in cases:
        ms = timed(eng, fn, args.reps)
        gbs = nbytes / (ms * 1e-3) / 1e9
        ct = cpu(cpu_fn)
        rows.append({
            "kernel": name, "workload": "{} x 2048 x 2048 uint16, bricks {}".format(Z, brick),
            "ms": round(ms, 4), "algorithmic_bytes": nbytes,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK, "unit": "GB/s",
                         "frac": round(gbs / PEAK, 4)},
            "planes_per_s": round(Z / (ms * 1e-3), 1),
            "cpu_baseline": {"value": round(cpu_bytes / ct / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
                             "sample": "{} planes, NumPy restatement, {:.2f} s".format(args.cpu_planes, ct)},
        })  # fmt: skip
        print(json.dumps(rows[-1]), flush=True)
    # round trip must be the identity
    back = d_a.download(vol.shape, np.uint16)
    assert np.array_equal(back, vol)
    eng.close()


if __name__ == "__main__":
    main()
