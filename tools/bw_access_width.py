#!/usr/bin/env python3
"""Streaming bandwidth of the brick re-tiling kernel with 16-byte and with 8-byte accesses per lane on the SAME
geometry (the 8-byte variant is selected by offsetting both buffers by 8 bytes: retile_vec() then drops to VEC 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aind_smartspim_destripe_amd import engine as E

class View:
    def __init__(self, buf, off): self.ptr, self.nbytes, self.engine = buf.ptr + off, buf.nbytes - off, buf.engine

e = E.DestripeEngine(0)
Z, H, W, cz, cy, cx = 256, 2048, 2048, 64, 128, 128
n = Z * H * W * 2
d_b, d_p = e.alloc(n + 64), e.alloc(n + 64)
for off in (0, 8):
    b, p = View(d_b, off), View(d_p, off)
    for rep in range(3):
        e.timer_start()
        for _ in range(10): e.bricks_to_planes(b, p, (Z, H, W), (cz, cy, cx))
        ms = e.timer_stop() / 10
    print("offset %d bytes -> %d-byte accesses: %.3f ms, %.2f TB/s" % (off, 16 if off == 0 else 8, ms, 2 * n / ms / 1e9))
