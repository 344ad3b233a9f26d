"""Multiscale pyramid of the destriped volume on the GPU.

Mirrors ``compute_pyramid`` (``/root/reference/code/aind_smartspim_destripe/zarr_destriper.py:365-407``) and the
level loop of ``compute_multiscale`` (``:677-794``): ``xarray_multiscale.multiscale`` with the
``windowed_mean`` reducer, ``scale_factors`` 2 per spatial axis and ``preserve_dtype=True`` -- every level
is the 2 x 2 x 2 windowed mean of the previous one, truncated back to uint16 -- without the dask
``LocalCluster`` (``:689-697``): one HIP kernel per level (``dsx_downsample2_u16``), the volume stays in HBM
between levels.  No CPU fallback.
"""

import numpy as np

from . import engine as _engine
from .mini_zarr import MiniZarrArray


def _check_scale(scale_axis):
    scale = [int(s) for s in scale_axis]
    if len(scale) < 3 or any(s != 1 for s in scale[:-3]) or scale[-3:] != [2, 2, 2]:
        raise ValueError("only scale factors (.., 2, 2, 2) are implemented (production setting, zarr_destriper.py:1176)")


def compute_pyramid(data, n_lvls, scale_axis, chunks="auto", device=0, engine=None):
    """``zarr_destriper.py:365-407``: ``[level 0 (the input), level 1, ...]``, ``n_lvls`` entries.

    ``data``: uint16 with the spatial axes last (``[Z, Y, X]`` up to ``[1, 1, Z, Y, X]``; leading axes must be
    singletons); ``chunks`` is accepted for signature parity and ignored (dense arrays come back).
    """
    _check_scale(scale_axis)
    vol = np.asarray(data)
    if vol.dtype != np.uint16:
        raise ValueError("the pyramid kernel takes uint16 volumes (what the destriped Zarr stores)")
    lead = vol.shape[:-3]
    if any(n != 1 for n in lead):
        raise ValueError("leading (t, c) axes must be singletons")
    zyx = tuple(vol.shape[-3:])
    eng = engine or _engine.DestripeEngine(device)
    levels = [vol]
    bufs = []
    try:
        d_prev = eng.alloc(max(vol.nbytes, 16))
        bufs.append(d_prev)
        d_prev.upload(np.ascontiguousarray(vol))
        for _ in range(1, int(n_lvls)):
            if min(zyx) < 2:
                break
            nxt = tuple(n // 2 for n in zyx)
            d_next = eng.alloc(max(int(np.prod(nxt)) * 2, 16))
            bufs.append(d_next)
            eng.downsample2(d_prev, d_next, zyx)
            levels.append(d_next.download(nxt, np.uint16).reshape(lead + nxt))
            d_prev, zyx = d_next, nxt
    finally:
        for b in bufs:
            b.free()
        if engine is None:
            eng.close()
    return levels


def compute_multiscale(level0_path, group_path, scale_factor=(2, 2, 2), n_levels=3, chunks=(1, 1, 64, 128, 128),
                       compressor=None, device=0):  # fmt: skip
    """Level loop of ``compute_multiscale`` (``zarr_destriper.py:746-782``) over Zarr-v2 directory stores:
    reads level 0, writes ``<group_path>/<i>`` for ``i = 1 .. n_levels - 1`` (uint16, ``"/"`` separator).
    OME-NGFF metadata (``:728-742``) is out of scope.  Returns the written arrays' shapes.
    """
    import os

    src = MiniZarrArray.open(level0_path)
    vol = src[(slice(None),) * src.ndim]
    pyr = compute_pyramid(vol, n_levels, (1,) * (vol.ndim - 3) + tuple(scale_factor), device=device)
    shapes = []
    for i, lvl in enumerate(pyr[1:], start=1):
        ck = tuple(min(c, n) for c, n in zip(chunks[-lvl.ndim :], lvl.shape))
        dst = MiniZarrArray.create(os.path.join(group_path, str(i)), lvl.shape, ck, np.uint16, compressor=compressor,
                                   dimension_separator="/")  # fmt: skip
        dst[(slice(None),) * lvl.ndim] = lvl
        shapes.append(lvl.shape)
    return shapes
