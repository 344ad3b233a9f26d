"""Host time to enqueue one launch chain (no device sync) against the device time per step: is the engine
launch-bound?  usage: python tools/host_enqueue.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aind_smartspim_destripe_amd import engine as eng_mod, synth

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = W = 2048
e = eng_mod.DestripeEngine(0)
e.plan(H, W, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=batch)
bank = synth.synthetic_bank(32, H, W)
stack = synth.synthetic_stack(batch, H, W, bank=bank)
d_in = e.alloc(stack.nbytes); d_in.upload(stack)
d_out = e.alloc(stack.nbytes)
for _ in range(5):
    e.run_device(d_in, np.uint16, batch, d_out, np.uint16)
e.sync()
for n in (1, 4, 16, 64):
    t0 = time.perf_counter()
    for _ in range(n):
        e.run_device(d_in, np.uint16, batch, d_out, np.uint16)
    t1 = time.perf_counter()
    e.sync()
    t2 = time.perf_counter()
    print("steps %3d: enqueue %.3f ms/step, total %.3f ms/step" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
