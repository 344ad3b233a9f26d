import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from aind_smartspim_destripe_amd import engine as eng_mod, synth
h = w = 2048
bank = synth.synthetic_bank(6, h, w)
stack = synth.synthetic_stack(72, h, w, bank=bank)
for mode in ("1", "3"):
    os.environ["DSX_FUSE_HIST"] = mode
    fails = 0
    for rep in range(12):
        e = eng_mod.DestripeEngine(0)
        try:
            e.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, synth.ZARR_PATH_HIGH_INT, max_batch=72)
            e.run(stack[:8], out_dtype=np.float32)
            e.run(stack, out_dtype=np.uint16)
        except Exception as ex:
            fails += 1
            print("mode", mode, "rep", rep, str(ex)[-160:], flush=True)
        finally:
            e.close()
    print("mode", mode, "failures", fails, "of 12", flush=True)
