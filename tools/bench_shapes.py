#!/usr/bin/env python3
"""Throughput of the device-resident path on other plane shapes (diagnosis; bench.py is the headline)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from aind_smartspim_destripe_amd import engine as E, synth

def run(h, w, n=128, steps=4):
    e = E.DestripeEngine(0)
    info = e.plan(h, w, synth.CELLS_CONFIG, synth.NO_CELLS_CONFIG, 2500, max_batch=n)
    bank = synth.synthetic_bank(8, h, w)
    stack = synth.synthetic_stack(n, h, w, bank=bank)
    d_in, d_out = e.alloc(stack.nbytes), e.alloc(n * info.out_height * info.out_width * 2)
    d_in.upload(stack)
    e.run_device(d_in, np.uint16, n, d_out, np.uint16); e.sync()
    e.timer_start()
    for _ in range(steps):
        e.run_device(d_in, np.uint16, n, d_out, np.uint16)
    ms = e.timer_stop() / steps
    print("%dx%d: %.0f planes/s (%.2f ms per %d planes), %.1f Mpx/s; fft %s halo %s" % (
        h, w, n / ms * 1e3, ms, n, n * h * w / ms / 1e3,
        [info.fft_len[i] for i in range(info.levels)], [info.fft_halo[i] for i in range(info.levels)]))
    e.close()

for hw in ((2048, 2048), (1800, 1800), (1600, 2000), (512, 512), (1024, 1024)):
    run(*hw)
