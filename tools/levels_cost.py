#!/usr/bin/env python3
"""4-stream throughput at 2048^2 when only the first L levels run (cost of the coarse levels; diagnosis)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aind_smartspim_destripe_amd import engine as E, synth  # noqa: E402

n, h, w = 256, 2048, 2048
bank = synth.synthetic_bank(8, h, w)
stack = synth.synthetic_stack(n, h, w, bank=bank)
for L in (None, 4, 3, 2, 1):
    e = E.DestripeEngine(0)
    cells = dict(synth.CELLS_CONFIG, level=L)
    nocells = dict(synth.NO_CELLS_CONFIG, level=L)
    e.plan(h, w, cells, nocells, 2500, max_batch=n)
    d_in, d_out = e.alloc(stack.nbytes), e.alloc(stack.nbytes)
    d_in.upload(stack)
    e.run_device(d_in, np.uint16, n, d_out, np.uint16)
    e.sync()
    e.timer_start()
    for _ in range(5):
        e.run_device(d_in, np.uint16, n, d_out, np.uint16)
    ms = e.timer_stop() / 5
    print("levels", L or 8, "ms per 256 planes %.3f" % ms, "planes/s %.0f" % (n / ms * 1e3))
    d_in.free(); d_out.free(); e.close()
