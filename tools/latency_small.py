"""Per-call latency and back-to-back rate of small cohorts (1 ... 16 planes of 2048^2 uint16, resident in HBM):
the per-slice calls of the reference's API.  Run twice: default (eager launches) and DSX_GRAPH=1 (HIP graph replay).
    python tools/latency_small.py [H W]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aind_smartspim_destripe_amd import engine as eng_mod, synth  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 2048)
e = eng_mod.DestripeEngine(0)
bank = synth.synthetic_bank(16, H, W)
for n in (1, 2, 4, 8, 16):
    e.plan(H, W, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=n)
    d_in, d_out = e.alloc(bank[:n].nbytes), e.alloc(bank[:n].nbytes)
    d_in.upload(bank[:n])
    for _ in range(5):
        e.run_device(d_in, np.uint16, n, d_out, np.uint16, None)
    e.sync()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        e.run_device(d_in, np.uint16, n, d_out, np.uint16, None)
        e.sync()
    lat = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        e.run_device(d_in, np.uint16, n, d_out, np.uint16, None)
    t_enq = (time.perf_counter() - t0) / reps
    e.sync()
    thr = (time.perf_counter() - t0) / reps
    print(json.dumps({"graph": os.environ.get("DSX_GRAPH", "0"), "planes": n, "latency_us": round(lat * 1e6, 1),
                      "back_to_back_us": round(thr * 1e6, 1), "enqueue_us": round(t_enq * 1e6, 1),
                      "planes_per_s": round(n / thr, 1), "graph_stats": e.graph_stats()}))
    d_in.free()
    d_out.free()
e.close()
