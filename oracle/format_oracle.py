"""NumPy restatement of the data-format steps either side of the stripe filter (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of the benches may import this
module; the product path never does.

* :func:`planes_to_bricks` / :func:`bricks_to_planes` -- what zarr's NumPy indexing does when the
  reference reads a block from, and assigns a block to, a ``(1, 1, 64, 128, 128)``-chunked array
  (``/root/reference/code/aind_smartspim_destripe/zarr_destriper.py:1066-1074`` create, ``:336`` assign):
  every chunk is a full-sized C-order brick, positions outside the array hold the fill value 0.
* :func:`windowed_mean_u16` / :func:`pyramid` -- ``compute_pyramid`` (``zarr_destriper.py:365-407``):
  ``xarray_multiscale.multiscale(reduction=windowed_mean, scale_factors=(..,2,2,2), preserve_dtype=True)``
  and its driver ``compute_multiscale`` (``:677-794``), which derives every level from the previous one.

PARITY UNPINNED for the pyramid: ``xarray_multiscale==2.1.0`` (``environment/Dockerfile:29``) is a third-party
dependency that is neither vendored in the reference nor installed here, and the reference holds no
fixture for it.  Restated from its published behaviour: windows of ``scale`` voxels per axis, trailing
voxels that do not fill a window are cropped, ``numpy.mean`` in float64, then ``astype(uint16)`` (truncation)
when ``preserve_dtype`` is set.  The brick layout is Zarr v2's published chunk layout and is pinned by the
store round trip in ``tests/test_zarr_chunk_map.py``.
"""

import numpy as np


def brick_grid(zyx, brick, z0=0):
    Z, H, W = zyx
    cz, cy, cx = brick
    return (-(-(z0 + Z) // cz), -(-H // cy), -(-W // cx))


def planes_to_bricks(planes, brick, z0=0):
    """Dense ``[Z, H, W]`` -> ``[nbz, nby, nbx, cz, cy, cx]`` (zero outside the stack)."""
    planes = np.asarray(planes)
    Z, H, W = planes.shape
    cz, cy, cx = brick
    nbz, nby, nbx = brick_grid(planes.shape, brick, z0)
    padded = np.zeros((nbz * cz, nby * cy, nbx * cx), dtype=planes.dtype)
    padded[z0 : z0 + Z, :H, :W] = planes
    return np.ascontiguousarray(padded.reshape(nbz, cz, nby, cy, nbx, cx).transpose(0, 2, 4, 1, 3, 5))


def bricks_to_planes(bricks, zyx, z0=0):
    """Inverse of :func:`planes_to_bricks`."""
    bricks = np.asarray(bricks)
    nbz, nby, nbx, cz, cy, cx = bricks.shape
    Z, H, W = zyx
    padded = bricks.transpose(0, 3, 1, 4, 2, 5).reshape(nbz * cz, nby * cy, nbx * cx)
    return np.ascontiguousarray(padded[z0 : z0 + Z, :H, :W])


def windowed_mean_u16(vol, scale=(2, 2, 2)):
    """One pyramid level: mean over ``scale`` windows in float64, truncated back to the input dtype."""
    vol = np.asarray(vol)
    sz, sy, sx = scale
    Z, Y, X = (vol.shape[0] // sz, vol.shape[1] // sy, vol.shape[2] // sx)
    v = vol[: Z * sz, : Y * sy, : X * sx].reshape(Z, sz, Y, sy, X, sx)
    return v.mean(axis=(1, 3, 5), dtype=np.float64).astype(vol.dtype)


def pyramid(vol, n_lvls, scale=(2, 2, 2)):
    """``[level 0 (the input), level 1, ...]``, ``n_lvls`` entries, each from the previous one."""
    out = [np.asarray(vol)]
    for _ in range(1, n_lvls):
        out.append(windowed_mean_u16(out[-1], scale))
    return out
