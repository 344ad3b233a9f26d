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


def write_pyramid_levels(level0_path, group_path, scale_factor=(2, 2, 2), n_levels=3, chunks=(1, 1, 64, 128, 128),
                       compressor="blosc", device=0, slab_planes=None):  # fmt: skip
    """Level loop of ``compute_multiscale`` (``zarr_destriper.py:746-782``) over Zarr-v2 directory stores
    (``zarr_destriper.compute_multiscale`` is the entry point with the reference's signature and calls this):
    writes ``<group_path>/<i>`` for ``i = 1 .. n_levels - 1`` (uint16, ``"/"`` separator), every level from the
    previous one.  OME-NGFF metadata (``:728-742``) is out of scope.  Returns the written arrays' shapes.

    The volume is streamed in z-slabs (a 34 GB channel does not fit one allocation, nor host RAM twice):
    a slab of ``2 * output z-chunk`` source planes is read, reduced by one ``dsx_downsample2_u16`` launch
    and written as whole output chunks; the 2 x 2 x 2 windows never straddle a slab because slabs start
    at even planes.
    """
    import os

    _check_scale((1,) * 2 + tuple(scale_factor))
    eng = _engine.DestripeEngine(device)
    shapes = []
    try:
        src_path = level0_path
        for i in range(1, int(n_levels)):
            src = MiniZarrArray.open(src_path)
            if src.dtype != np.uint16 or any(n != 1 for n in src.shape[:-3]):
                raise ValueError("the pyramid kernel takes uint16 volumes with singleton leading axes")
            Z, Y, X = src.shape[-3:]
            if min(Z, Y, X) < 2:
                break
            out_zyx = (Z // 2, Y // 2, X // 2)
            lead = src.shape[:-3]
            out_shape = lead + out_zyx
            ck = tuple(min(c, n) for c, n in zip(tuple(chunks)[-len(out_shape):], out_shape))
            dst = MiniZarrArray.create(os.path.join(group_path, str(i)), out_shape, ck, np.uint16,
                                       compressor=compressor, dimension_separator="/")  # fmt: skip
            slab = int(slab_planes) if slab_planes else 2 * ck[-3]
            slab += slab & 1
            d_src = eng.alloc(slab * Y * X * 2)
            d_dst = eng.alloc(max((slab // 2) * out_zyx[1] * out_zyx[2] * 2, 16))
            try:
                lead_idx = (0,) * len(lead)
                for z in range(0, 2 * out_zyx[0], slab):
                    n = min(slab, 2 * out_zyx[0] - z)  # even: an odd trailing plane is cropped
                    d_src.upload(src[lead_idx + (slice(z, z + n),)])
                    eng.downsample2(d_src, d_dst, (n, Y, X))
                    out = d_dst.download((n // 2,) + out_zyx[1:], np.uint16)
                    dst[lead_idx + (slice(z // 2, z // 2 + n // 2),)] = out
            finally:
                d_src.free()
                d_dst.free()
            shapes.append(out_shape)
            src_path = os.path.join(group_path, str(i))
    finally:
        eng.close()
    return shapes
