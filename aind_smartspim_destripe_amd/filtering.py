"""Drop-in mirror of ``aind_smartspim_destripe.filtering`` whose hot path runs on an MI355X.

Same names, argument meaning and error behaviour as the reference
(``/root/reference/code/aind_smartspim_destripe/filtering.py``):

* :func:`filter_stripes` (reference ``:417-491``) and :func:`log_space_fft_filtering`
  (``:139-224``) ALWAYS run through the HIP engine (``libdsx_hip.so``); there is no CPU
  fallback -- without the library or a GPU they raise :class:`~.engine.DsxError`.
* :func:`filter_streaks` is the alias BASELINE.json's north star names (the reference has no such
  function; it forwards to :func:`log_space_fft_filtering`).
* :func:`destripe_planes` is the batched form the Zarr chunk map uses instead of the per-plane
  z-loop of ``execute_worker`` (``zarr_destriper.py:319-327``).
* :func:`get_foreground_background_mean` (``:54-88``) and :func:`flatfield_correction` (``:338-414``) called
  on their own also run on the device (``dsx_foreground_background``, ``dsx_flatfield_correction``).
* The remaining helpers (``sigmoid``, ``notch``, ``gaussian_filter``, ``normalize_image``,
  ``get_hemisphere_flatfield``) are parameter / lookup helpers, plain NumPy here as they are there; the
  filter never calls them (the engine builds its own gain tables).
"""

import collections
import math
import warnings
from typing import List, Optional, Tuple

import numpy as np

from . import engine as _engine
from . import wavelets as _wavelets



# ---------------------------------------------------------------------------------------------
# host-side helpers with the reference's names
# ---------------------------------------------------------------------------------------------
def sigmoid(data: np.array):
    """``filtering.py:13-22``."""
    return 1 / (1 + np.exp(-data))


def foreground_fraction(img: np.array, center: float, crossover: float) -> float:
    """``filtering.py:25-51``."""
    z = (img - center) / crossover
    return sigmoid(z)


def _foreground_cutoff(threshold_mask):
    """Smallest float16 pixel value v with ``sigmoid((v - 400) / 20) > threshold_mask`` in the reference's
    float16 arithmetic (``filtering.py:75-78``), found by evaluating every finite float16 value; ``inf`` when
    no pixel value passes.  The decision is monotone in v, so ``float16(pixel) >= cutoff`` is the mask."""
    v = np.arange(0x0000, 0x7C00, dtype=np.uint16).view(np.float16)  # non-negative finite values, ascending
    v = np.concatenate([-v[:0:-1], v])  # negative ones (descending bit pattern = ascending value) first
    with np.errstate(over="ignore"):
        ok = foreground_fraction(v, 400, 20) > threshold_mask
    if not ok.any():
        return float("inf")
    if ok.all():
        return -float("inf")
    first = int(np.argmax(ok))
    if not ok[first:].all():  # not monotone for this threshold (cannot happen for a sigmoid)
        raise ValueError("threshold_mask does not define a cutoff")
    return float(v[first])


def get_foreground_background_mean(img: np.array, threshold_mask: Optional[float] = 0.3, device: int = 0) -> Tuple:
    """``filtering.py:54-88`` on the GPU (``dsx_foreground_background``): (foreground mean, background mean,
    float16 mask image).  ``filter_stripes`` evaluates the same statistic fused into its first kernel.
    """
    img = np.asarray(img)
    if img.size == 0:
        return 0.0, 0.0, np.zeros(img.shape, np.float16)
    cutoff = _foreground_cutoff(threshold_mask)
    eng = next((e[0] for k, e in _ENGINES.items() if k[0] == device), None)
    own = eng is None
    if own:
        eng = _engine.DestripeEngine(device)
    try:
        fore, back, mask = eng.foreground_background(_as_plane_dtype(img), cutoff)
    finally:
        if own:
            eng.close()
    return fore, back, mask.astype(np.float16)


def notch(n, sigma):
    """``filtering.py:91-115``: 1-D gaussian notch ``1 - exp(-x^2 / (2 sigma^2))``."""
    if n <= 0:
        raise ValueError("n must be positive")
    n = int(n)
    if sigma <= 0:
        raise ValueError("sigma must be positive")
    x = np.arange(n)
    return 1 - np.exp(-(x**2) / (2 * sigma**2))


def gaussian_filter(shape, sigma):
    """``filtering.py:118-136``."""
    g = notch(n=shape[-1], sigma=sigma)
    return np.broadcast_to(g, shape).copy()


def normalize_image(images: List[np.array]) -> np.ndarray:
    """``filtering.py:227-250``: scale into [1, 2] (float16 fraction)."""
    images = np.array(images)
    min_val, max_val = np.min(images), np.max(images)
    return 1 + np.divide(images - min_val, max_val - min_val).astype(np.float16)


def invert_image(image: np.array) -> np.ndarray:
    """``filtering.py:253-270``."""
    image = np.array(image)
    return image.max() - image


def get_hemisphere_flatfield(input_tile_path, tile_config, flatfields, zarr=True):
    """``filtering.py:273-335``: flat of the laser side a tile belongs to (``KeyError`` if unknown)."""
    if zarr:
        parts = str(input_tile_path).split("_")
    else:
        parts = str(input_tile_path).split("/")[-2].split("_")
    x_folder, y_folder = parts[0], parts[1]
    if tile_config.get(x_folder) is None:
        raise KeyError(f"Please, check the tile config while trying to reach: {x_folder}")
    brain_side = tile_config[x_folder].get(y_folder)
    if brain_side is None:
        raise KeyError(f"Please, check the tile config while trying to reach: {y_folder}")
    return flatfields[brain_side]


def flatfield_correction(image_tiles, flatfield, darkfield, baseline=None, device=0):
    """``filtering.py:338-414`` as a stand-alone call on the GPU (``dsx_flatfield_correction``; uint16 result).

    Shape handling and errors follow the reference line by line (including its quirk that only one plane
    passes the shape checks, ``:370-391``); the arithmetic runs in float32 on the device.  Inside
    :func:`filter_stripes` the same arithmetic is fused into the last synthesis kernel.
    """
    image_tiles = np.array(image_tiles)
    flatfield, darkfield = np.asarray(flatfield), np.asarray(darkfield)
    if image_tiles.ndim != flatfield.ndim:
        flatfield = np.expand_dims(flatfield, axis=0)
    if image_tiles.ndim != darkfield.ndim:
        darkfield = np.expand_dims(darkfield, axis=0)
    darkfield = darkfield[: image_tiles.shape[-2], : image_tiles.shape[-1]]
    if darkfield.shape != image_tiles.shape:
        raise ValueError(
            "Please, check the shape of the darkfield. "
            f"Image: {image_tiles.shape} - Darkfield: {darkfield.shape}"
        )
    if flatfield.shape != image_tiles.shape:
        raise ValueError(
            "Please, check the shape of the flatfield."
            f"Image: {image_tiles.shape} - Flatfield: {flatfield.shape}"
        )
    if baseline is None:
        baseline = np.zeros((image_tiles.shape[0],))
    baseline = np.asarray(baseline, dtype=np.float64).ravel()
    plane_shape = image_tiles.shape[-2:]
    if image_tiles.ndim < 2 or int(np.prod(image_tiles.shape[:-2])) != 1:
        raise ValueError("flatfield_correction takes one plane ([H, W] or [1, H, W])")
    # baseline[baseline_indxs] (:393-398, 409): index 0 of the array gets one value each -- the rows of a 2-D plane
    # (k_shade takes one value per row), the single plane of a [1, H, W] stack (a scalar)
    if baseline.size not in (1, image_tiles.shape[0]):
        raise ValueError(
            "operands could not be broadcast together with shapes {} {}".format(
                image_tiles.shape, (baseline.size,) + (1,) * (image_tiles.ndim - 1)))
    per_row = image_tiles.ndim == 2 and baseline.size == plane_shape[0] and baseline.size > 1
    eng = next((e[0] for k, e in _ENGINES.items() if k[0] == device), None)
    own = eng is None
    if own:
        eng = _engine.DestripeEngine(device)
    try:
        out = eng.flatfield_correction(_as_plane_dtype(image_tiles.reshape(plane_shape)), flatfield.reshape(plane_shape),
                                       darkfield.reshape(plane_shape),
                                       baseline if per_row else (float(baseline[0]) if baseline.size else 0.0))
    finally:
        if own:
            eng.close()
    return out.reshape(image_tiles.shape)


# ---------------------------------------------------------------------------------------------
# GPU hot path
# ---------------------------------------------------------------------------------------------
_ENGINES = collections.OrderedDict()
_MAX_CACHED_PLANS = 8


def _cfg_key(cfg):
    return (cfg.get("wavelet", "db3"), cfg.get("level", 0), float(cfg.get("sigma", 64)),
            float(cfg.get("max_threshold", 4)))  # fmt: skip


def _max_level(shape, filter_len=6):
    """``pywt.dwt_max_level`` over both axes (6 taps: db3)."""

    def one(n):
        if n < filter_len - 1:
            return 0
        return max(0, int(math.floor(math.log2(n // (filter_len - 1)))))

    return min(one(shape[0]), one(shape[1]))


def _warn_levels(shape, *cfgs):
    """pywt.wavedec2 warns (does not fail) when ``level`` exceeds the maximum useful level."""
    for cfg in cfgs:
        mx = _max_level(shape, _wavelets.filter_length(cfg.get("wavelet", "db3")))
        lvl = cfg.get("level", 0)
        if lvl is not None and lvl > mx:
            warnings.warn(
                f"Level value of {lvl} is too high: all coefficients will experience boundary effects.",
                UserWarning,
            )
        if lvl is not None and lvl < 0:
            raise ValueError("Level value of %d is too low . Minimum level is 0." % lvl)


def get_engine(shape, cells_config, no_cells_config, microscope_high_int=2700, flatfield=None,
               darkfield=None, max_batch=32, device=0):  # fmt: skip
    """Planned engine for a plane geometry + config pair (cached per process and device)."""
    shade_key = None
    if flatfield is not None:
        # keyed on the memory the planes occupy, not on the Python object: ``flatfields[brain_side]`` makes a new
        # view object per call (prospective flats, filtering.py:478), which would miss the cache every time
        shade_key = (_array_key(flatfield), _array_key(darkfield))
    key = (device, tuple(shape), _cfg_key(cells_config), _cfg_key(no_cells_config),
           float(microscope_high_int), shade_key, int(max_batch))  # fmt: skip
    eng = _ENGINES.get(key)
    if eng is not None:
        _ENGINES.move_to_end(key)  # least recently used entries go first
        return eng[0]
    while len(_ENGINES) >= _MAX_CACHED_PLANS:
        _, old = _ENGINES.popitem(last=False)
        old[0].close()
    e = _engine.DestripeEngine(device)
    e.plan(shape[0], shape[1], cells_config, no_cells_config, microscope_high_int, max_batch,
           flatfield, darkfield)  # fmt: skip
    # the shading arrays stay referenced, so their addresses stay unique while the plan is cached
    _ENGINES[key] = (e, flatfield, darkfield)
    return e


def release_engines():
    """Close every cached engine of this process (their workspaces go back to the device; a later call plans anew and
    its context reads the ``DSX_*`` switches of the environment again)."""
    while _ENGINES:
        _, old = _ENGINES.popitem(last=False)
        old[0].close()


def _array_key(a):
    a = np.asarray(a)
    return (a.__array_interface__["data"][0], a.shape, a.strides, a.dtype.str)


def _as_plane_dtype(image):
    image = np.asarray(image)
    if image.dtype == np.uint16 or image.dtype == np.float32:
        return np.ascontiguousarray(image)
    if np.issubdtype(image.dtype, np.integer) and image.size and image.min() >= 0 and image.max() <= 65535:
        return np.ascontiguousarray(image, dtype=np.uint16)
    return np.ascontiguousarray(image, dtype=np.float32)


def log_space_fft_filtering(
    input_image: np.array,
    wavelet: Optional[str] = "db3",
    level: Optional[int] = 0,
    sigma: Optional[int] = 64,
    max_threshold: Optional[int] = 4,
):
    """``filtering.py:139-224`` on the GPU: log -> DWT (any PyWavelets discrete wavelet but ``dmey``; ``db3``,
    the production setting, has specialised kernels) -> per-level Otsu mask, row-median
    in-paint and packed-index gaussian notch on cH -> inverse DWT -> ``exp(.) + 1.0``.

    Returns float64 ``[H + H % 2, W + W % 2]`` like the reference (computed in float32 on the
    device; within 1e-4 relative of the NumPy/SciPy path).

    A 3-D ``[n, H, W]`` input is the reference's stack mode (``:182-183, 188, 210-211``): every plane is decomposed
    on its own, but each level takes ONE Otsu threshold from the coefficients of all planes (``dsx_set_stack_mode``).
    Planes that are to be filtered independently go through :func:`destripe_planes`.
    """
    image = np.asarray(input_image)
    if image.ndim not in (2, 3):
        raise ValueError("log_space_fft_filtering takes a 2-D plane or a 3-D stack [n, H, W]")
    cfg = {"wavelet": wavelet, "level": level, "sigma": sigma, "max_threshold": max_threshold}
    if sigma <= 0:
        raise ValueError("sigma must be positive")
    _warn_levels(image.shape[-2:], cfg)
    if image.ndim == 3:
        if image.shape[0] == 0:
            raise ValueError("empty stack")
        planes = image if image.dtype in (np.uint16, np.float32) else np.stack([_as_plane_dtype(p) for p in image])
        if planes.dtype not in (np.uint16, np.float32):
            planes = planes.astype(np.float32)
        eng = get_engine(image.shape[-2:], cfg, cfg, microscope_high_int=2700, max_batch=int(image.shape[0]))
        eng.set_stack_mode(True)
        try:
            out = eng.run(np.ascontiguousarray(planes), out_dtype=np.float32)
        finally:
            eng.set_stack_mode(False)
        return out.astype(np.float64)
    eng = get_engine(image.shape, cfg, cfg, microscope_high_int=2700, max_batch=1)
    out = eng.run(_as_plane_dtype(image)[None], out_dtype=np.float32)[0]
    return out.astype(np.float64)


def filter_streaks(image, **params):
    """Alias named by BASELINE.json (upstream pystripe name): ``log_space_fft_filtering(image, **params)``."""
    return log_space_fft_filtering(input_image=image, **params)


def _resolve_shading(shadow_correction, input_tile_path):
    if shadow_correction is None:
        return None, None
    flatfield = shadow_correction.get("flatfield")
    darkfield = shadow_correction.get("darkfield")
    if not shadow_correction.get("retrospective"):
        flatfield = get_hemisphere_flatfield(
            input_tile_path=input_tile_path,
            tile_config=shadow_correction.get("tile_config"),
            flatfields=flatfield,
        )
    return flatfield, darkfield


def filter_stripes(
    image: np.array,
    input_tile_path: str,
    no_cells_config: dict,
    cells_config: dict,
    shadow_correction: Optional[dict] = None,
    microscope_high_int: Optional[int] = 2700,
) -> np.array:
    """``filtering.py:417-491`` on the GPU (one plane).

    The fg/bg statistic, the config choice, the filter and (if given) the dark/flat correction
    all run on the device.  Returns float64 without shading and uint16 with shading, like the
    reference.
    """
    image = np.asarray(image)
    if image.ndim != 2:
        raise ValueError("filter_stripes takes one 2-D plane; use destripe_planes for a stack")
    flatfield, darkfield = _resolve_shading(shadow_correction, input_tile_path)
    _warn_levels(image.shape, cells_config, no_cells_config)
    plane = _as_plane_dtype(image)[None]
    if _engine._wavelet_key(cells_config) != _engine._wavelet_key(no_cells_config):
        out = _destripe_planes_two_wavelets(plane, no_cells_config, cells_config, microscope_high_int, flatfield, darkfield,
                                            np.uint16 if flatfield is not None else np.float32, 1, False, 0)[0]  # fmt: skip
        return out if flatfield is not None else out.astype(np.float64)
    eng = get_engine(image.shape, cells_config, no_cells_config, microscope_high_int, flatfield, darkfield,
                     max_batch=1)  # fmt: skip
    if flatfield is not None:
        return eng.run(plane, out_dtype=np.uint16)[0]
    return eng.run(plane, out_dtype=np.float32)[0].astype(np.float64)


def destripe_planes(
    planes: np.ndarray,
    input_tile_path: str,
    no_cells_config: dict,
    cells_config: dict,
    shadow_correction: Optional[dict] = None,
    microscope_high_int: Optional[int] = 2700,
    out_dtype=np.uint16,
    max_batch: int = 32,
    return_config: bool = False,
    device: int = 0,
):
    """Batched ``filter_stripes`` over ``planes[n, H, W]`` (uint16 or float32): every plane is
    filtered independently, exactly as the z-loop of ``execute_worker`` does
    (``zarr_destriper.py:319-327``), in cohorts of ``max_batch`` planes per launch chain.

    ``out_dtype`` uint16 = what the Zarr path stores (clip + truncate); float32 = ``exp(y) + 1``.
    """
    planes = np.asarray(planes)
    if planes.ndim != 3:
        raise ValueError("planes must be [n, H, W]")
    flatfield, darkfield = _resolve_shading(shadow_correction, input_tile_path)
    _warn_levels(planes.shape[1:], cells_config, no_cells_config)
    if planes.dtype != np.uint16 and planes.dtype != np.float32:
        planes = np.stack([_as_plane_dtype(p) for p in planes]) if len(planes) else planes.astype(np.float32)
    if _engine._wavelet_key(cells_config) != _engine._wavelet_key(no_cells_config):
        return _destripe_planes_two_wavelets(planes, no_cells_config, cells_config, microscope_high_int, flatfield,
                                             darkfield, out_dtype, max_batch, return_config, device)  # fmt: skip
    eng = get_engine(planes.shape[1:], cells_config, no_cells_config, microscope_high_int, flatfield,
                     darkfield, max_batch=max_batch, device=device)  # fmt: skip
    return eng.run(planes, out_dtype=out_dtype, return_cfg=return_config)


def _destripe_planes_two_wavelets(planes, no_cells_config, cells_config, microscope_high_int, flatfield, darkfield,
                                  out_dtype, max_batch, return_config, device):
    """The two configs name DIFFERENT wavelets.  One launch chain decomposes a plane before its config is known (the
    fg/bg statistic is fused into the first kernel), so it cannot switch filter banks per plane; the reference decides
    first and decomposes afterwards (``filtering.py:459-467``).  Same order here: the statistic of every plane
    (``dsx_foreground_background``), the reference's decision rule, then one engine per config (both of its slots hold
    that config) over the planes that chose it."""
    n = planes.shape[0]
    cutoff = _foreground_cutoff(0.3)
    which = np.zeros(n, dtype=np.int32)
    probe = get_engine(planes.shape[1:], no_cells_config, no_cells_config, microscope_high_int, flatfield, darkfield,
                       max_batch=max_batch, device=device)  # fmt: skip
    for k in range(n):
        fore, back, _ = probe.foreground_background(planes[k], cutoff, want_mask=False)
        which[k] = 1 if (fore > back and fore > microscope_high_int) else 0
    out = None
    for w, cfg in ((0, no_cells_config), (1, cells_config)):
        idx = np.nonzero(which == w)[0]
        if idx.size == 0:
            continue
        eng = get_engine(planes.shape[1:], cfg, cfg, microscope_high_int, flatfield, darkfield,
                         max_batch=max_batch, device=device)  # fmt: skip
        res = eng.run(np.ascontiguousarray(planes[idx]), out_dtype=out_dtype)
        if out is None:
            out = np.empty((n,) + res.shape[1:], dtype=res.dtype)
        out[idx] = res
    if out is None:
        out = np.empty((0,) + tuple(s + (s & 1) for s in planes.shape[1:]), dtype=out_dtype)
    return (out, which) if return_config else out
