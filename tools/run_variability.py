"""Is the run-to-run spread of the headline bench a property of the process (memory placement) or of the moment (phase
of the four sub-cohort streams)?  Ten timed segments of 100 steps inside ONE process, each after a full sync; compare
with the spread between processes (run this script several times).  usage: python tools/run_variability.py [segments]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aind_smartspim_destripe_amd import engine as eng_mod, synth

segs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
batch, H, W = 256, 2048, 2048
e = eng_mod.DestripeEngine(0)
e.plan(H, W, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=batch)
bank = synth.synthetic_bank(32, H, W)
stack = synth.synthetic_stack(batch, H, W, bank=bank)
d_in = e.alloc(stack.nbytes); d_in.upload(stack)
d_out = e.alloc(stack.nbytes)
for _ in range(60):
    e.run_device(d_in, np.uint16, batch, d_out, np.uint16)
e.sync()
vals = []
for s in range(segs):
    t0 = time.perf_counter()
    for _ in range(100):
        e.run_device(d_in, np.uint16, batch, d_out, np.uint16)
    e.sync()
    dt = time.perf_counter() - t0
    vals.append(100 * batch / dt)
print("segments (planes/s):", " ".join("%.0f" % v for v in vals), "| spread %.1f %%" % (100 * (max(vals) - min(vals)) / np.mean(vals)))
