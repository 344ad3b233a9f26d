"""Deterministic synthetic SmartSPIM planes.

Pure NumPy and importable on its own (the golden-vector generator runs it under another
interpreter).  ``numpy.random.RandomState`` is the frozen legacy stream, so the same seed gives
the same plane under every NumPy version.

Plane ``k`` of a bank::

    rs    = RandomState(1000 + k)
    base  = rs.poisson(150, (H, W))
    gain  = exp(0.15 * rs.randn(H))          # multiplicative horizontal stripes, constant along x
    img   = base * gain[:, None]
    if k % 4 == 0:                           # planes "with cells": fg mean > 2500 -> cells_config
        n  = round(60 * H * W / 2048**2) (at least 1) squares of 10x10 pixels of +4000
    img   = clip(img, 0, 65535).astype(uint16)
"""

import numpy as np

# production parameters, reference code/run_capsule.py:374-390
NO_CELLS_CONFIG = {"wavelet": "db3", "level": None, "sigma": 128, "max_threshold": 12}
CELLS_CONFIG = {"wavelet": "db3", "level": None, "sigma": 64, "max_threshold": 3}
# reference zarr_destriper.py:326
ZARR_PATH_HIGH_INT = 2500


def synthetic_plane(k, height, width):
    """uint16 plane ``k`` of the synthetic bank (see module docstring)."""
    rs = np.random.RandomState(1000 + int(k))
    base = rs.poisson(150, (height, width)).astype(np.float64)
    gain = np.exp(0.15 * rs.randn(height))
    img = base * gain[:, None]
    if k % 4 == 0:
        n_cells = max(1, int(round(60.0 * height * width / (2048.0 * 2048.0))))
        side = min(10, height, width)
        ys = rs.randint(0, max(1, height - side), n_cells)
        xs = rs.randint(0, max(1, width - side), n_cells)
        for y, x in zip(ys, xs):
            img[y : y + side, x : x + side] += 4000.0
    return np.clip(img, 0, 65535).astype(np.uint16)


def synthetic_bank(n_unique, height, width, first=0):
    """``uint16[n_unique, H, W]`` of bank planes ``first .. first + n_unique - 1``."""
    out = np.empty((n_unique, height, width), dtype=np.uint16)
    for i in range(n_unique):
        out[i] = synthetic_plane(first + i, height, width)
    return out


def synthetic_stack(n_slices, height, width, bank=None, n_unique=32):
    """``uint16[n_slices, H, W]``: slice z is bank[z % n_unique] rolled by ``z // n_unique`` rows."""
    if bank is None:
        bank = synthetic_bank(min(n_unique, n_slices), height, width)
    n_unique = bank.shape[0]
    out = np.empty((n_slices, height, width), dtype=np.uint16)
    for z in range(n_slices):
        out[z] = np.roll(bank[z % n_unique], z // n_unique, axis=0)
    return out
