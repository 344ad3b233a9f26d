#!/usr/bin/env python3
"""Does the level-1 forward kernel's time follow the number of 126-column strips (waves) or the pixels?
2048 x 2048 (9 strips, the last 14 % full) against 2048 x 2012 (8 full strips)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DSX_STREAMS"] = "1"
from aind_smartspim_destripe_amd import engine as E, synth  # noqa: E402

n = 256
for h, w in ((2048, 2048), (2048, 2012), (2048, 1760)):
    e = E.DestripeEngine(0)
    e.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, 2500, max_batch=n)
    bank = synth.synthetic_bank(4, h, w)
    stack = synth.synthetic_stack(n, h, w, bank=bank)
    d_in, d_out = e.alloc(stack.nbytes), e.alloc(stack.nbytes)
    d_in.upload(stack)
    e.run_device(d_in, np.uint16, n, d_out, np.uint16)
    e.sync()
    e.profile(True)
    e.run_device(d_in, np.uint16, n, d_out, np.uint16)
    e.sync()
    p = e.profile_read()
    print("%dx%d" % (h, w), "strips fwd %d" % -(-((w + 5) // 2) // 126), "inv strips %d" % -(-w // 256),
          " ".join("%s=%.3f" % (k.replace("k_", "").replace("_march", ""), v[0]) for k, v in p.items()))
    e.profile(False)
    d_in.free(); d_out.free(); e.close()
