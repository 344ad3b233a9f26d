#!/usr/bin/env python3
"""Diagnosis: where do a 4-stream and a 1-stream engine (and two runs of each) differ?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aind_smartspim_destripe_amd import engine as eng_mod, synth

n, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 2048, 2048
stack = synth.synthetic_stack(n, h, w, n_unique=8)
def mk(s):
    os.environ["DSX_STREAMS"] = str(s)
    e = eng_mod.DestripeEngine(0)
    e.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=n)
    return e
e4, e1 = mk(4), mk(1)
a = e4.run(stack, out_dtype=np.float32)
b = e1.run(stack, out_dtype=np.float32)
a2 = e4.run(stack, out_dtype=np.float32)
b2 = e1.run(stack, out_dtype=np.float32)
def rep(x, y, name):
    d = x != y
    print(name, "planes differing:", [int(k) for k in np.nonzero(d.reshape(n, -1).any(1))[0]])
    for k in np.nonzero(d.reshape(n, -1).any(1))[0][:3]:
        ys, xs = np.nonzero(d[k])
        print("  plane", k, "count", len(ys), "rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(),
              "max abs", float(np.abs(x[k] - y[k]).max()), "rows hist", np.unique(ys // 64, return_counts=True))
rep(a, b, "4 vs 1")
rep(a, a2, "4 vs 4")
rep(b, b2, "1 vs 1")
for stage in (1, 2):
    for e, nm in ((e4, "e4"), (e1, "e1")):
        e.set_stop_after(stage)
    e4.run(stack[:16], out_dtype=np.float32); e1.run(stack[:16], out_dtype=np.float32)
    for lv in range(e4.levels):
        for st in (eng_mod.STAGE_DETAIL,):
            x = e4.level_array(3, lv, st); y = e1.level_array(3, lv, st)
            if not np.array_equal(x, y):
                ys, xs = np.nonzero(x != y)
                print("stage", stage, "level", lv, "plane 3 differs:", len(ys), "rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
    t4 = [e4.thresholds(3, lv) for lv in range(e4.levels)]; t1 = [e1.thresholds(3, lv) for lv in range(e1.levels)]
    if t4 != t1: print("stage", stage, "thresholds differ", t4, t1)
print("done")
