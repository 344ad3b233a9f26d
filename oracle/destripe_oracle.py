"""CPU oracle for the per-slice destripe hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a plain NumPy restatement of the reference algorithm
(``filter_stripes`` -> ``log_space_fft_filtering``).  It is the *checker* for the
HIP engine: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  The product package
(``aind_smartspim_destripe_amd``) never imports it and has no CPU fallback.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the real reference
(``/root/reference/code/aind_smartspim_destripe/filtering.py``) under
``/opt/conda/bin/python3.9`` (numpy 1.26.4, scipy 1.7.1, PyWavelets 1.1.1,
scikit-image 0.18.3) and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function here against those vectors
(<= 1e-11 relative in the float64 regime).

The arithmetic of the reference lives in un-vendored third-party libraries; each
function below names the reference call site and the library routine it restates:

* PyWavelets 1.1.1 (pinned 1.6.0 upstream, ``environment/Dockerfile:14-29``):
  ``wavedec2``/``waverec2`` with ``db3``, ``mode='symmetric'``
  (call sites ``filtering.py:176`` and ``filtering.py:221``).
* scikit-image 0.18.3 ``filters.threshold_otsu`` (``filtering.py:191``) on top of
  ``numpy.histogram`` (numpy 1.26.4 ``lib/histograms.py:800-850``).
* ``scipy.fftpack.rfft/irfft`` (``filtering.py:206,215``): packed real FFT.
* ``numpy.median`` (``filtering.py:201``).

Only NumPy is needed, so it runs on the default interpreter here and on the GPU box.
"""

import math
import warnings

import numpy as np

# ----------------------------------------------------------------------------------------------
# db3 filter bank (PyWavelets ``Wavelet('db3')``; values printed by pywt 1.1.1)
# ----------------------------------------------------------------------------------------------
DB3_DEC_LO = np.array(
    [
        0.03522629188570953,
        -0.08544127388202666,
        -0.13501102001025458,
        0.45987750211849154,
        0.8068915093110925,
        0.33267055295008263,
    ],
    dtype=np.float64,
)
DB3_DEC_HI = np.array(
    [
        -0.33267055295008263,
        0.8068915093110925,
        -0.45987750211849154,
        -0.13501102001025458,
        0.08544127388202666,
        0.03522629188570953,
    ],
    dtype=np.float64,
)
DB3_REC_LO = DB3_DEC_LO[::-1].copy()
DB3_REC_HI = DB3_DEC_HI[::-1].copy()
FILTER_LEN = 6
DB3_BANK = (DB3_DEC_LO, DB3_DEC_HI, DB3_REC_LO, DB3_REC_HI)


def as_bank(wavelet):
    """``"db3"`` or a filter bank ``(dec_lo, dec_hi, rec_lo, rec_hi)`` (the attributes of a ``pywt.Wavelet``; the
    tests take them from the table written by ``oracle/make_wavelet_table.py``) -> four float64 arrays."""
    if wavelet is None or (isinstance(wavelet, str) and wavelet == "db3"):
        return DB3_BANK
    if isinstance(wavelet, str):
        raise ValueError("the oracle knows 'db3' by name; pass the filter bank of any other wavelet")
    bank = tuple(np.asarray(f, dtype=np.float64) for f in wavelet)
    if len(bank) != 4 or len({len(f) for f in bank}) != 1 or len(bank[0]) % 2:
        raise ValueError("a filter bank is (dec_lo, dec_hi, rec_lo, rec_hi) of one even length")
    return bank


def dwt_max_level(data_len, filter_len=FILTER_LEN):
    """``pywt.dwt_max_level``: floor(log2(data_len / (filter_len - 1))), never below 0."""
    if data_len < filter_len - 1:
        return 0
    return max(0, int(math.floor(math.log2(data_len // (filter_len - 1)))))


def dwt_coeff_len(n, filter_len=FILTER_LEN):
    """``pywt.dwt_coeff_len`` for every mode except periodization."""
    return (n + filter_len - 1) // 2


def _symmetric_index(idx, n):
    """Half-sample symmetric extension (np.pad mode 'symmetric'), any distance."""
    p = 2 * n
    idx = np.mod(idx, p)
    return np.where(idx < n, idx, p - 1 - idx)


def dwt_axis(x, filt, axis):
    """1-D analysis along ``axis`` (PyWavelets C ``dwt_axis``, mode symmetric).

    out[i] = sum_k filt[k] * x~[2 i + 1 - k],  i = 0 .. (N + F - 1)//2 - 1, where x~ is the half-sample
    symmetric extension of x.  Computed in the dtype of ``x`` (float32 stays float32, as in pywt).
    """
    x = np.moveaxis(x, axis, -1)
    n = x.shape[-1]
    flen = len(filt)
    m = dwt_coeff_len(n, flen)
    f = filt.astype(x.dtype)
    pos = 2 * np.arange(m)[:, None] + 1 - np.arange(flen)[None, :]  # [m, F]
    idx = _symmetric_index(pos, n)
    out = np.zeros(x.shape[:-1] + (m,), dtype=x.dtype)
    for k in range(flen):
        out += f[k] * x[..., idx[:, k]]
    return np.moveaxis(out, -1, axis)


def idwt_axis(a, d, rec_lo, rec_hi, axis):
    """1-D synthesis along ``axis`` (PyWavelets C ``idwt_axis``, any non-periodization mode).

    For M coefficients and F taps the output has 2 M - F + 2 samples:
    out[2 p + b] = sum_{j < F/2} a[p + j] * rec_lo[F - 2 - 2 j + b] + d[p + j] * rec_hi[F - 2 - 2 j + b].
    """
    a = np.moveaxis(a, axis, -1)
    d = np.moveaxis(d, axis, -1)
    m = a.shape[-1]
    flen = len(rec_lo)
    n_out = 2 * m - flen + 2
    dt = np.result_type(a.dtype, d.dtype)
    out = np.zeros(a.shape[:-1] + (n_out,), dtype=dt)
    p = n_out // 2  # number of (even, odd) output pairs
    lo = rec_lo.astype(dt)
    hi = rec_hi.astype(dt)
    for b in (0, 1):
        acc = np.zeros(a.shape[:-1] + (p,), dtype=dt)
        for j in range(flen // 2):
            t = flen - 2 - 2 * j + b
            acc += a[..., j : j + p] * lo[t] + d[..., j : j + p] * hi[t]
        out[..., b::2] = acc
    return np.moveaxis(out, -1, axis)


def dwt2(x, bank=DB3_BANK):
    """One 2-D analysis level: axis 0 first, then axis 1 (pywt ``dwtn`` order).

    Returns ``aa, (da, ad, dd)`` = cA, (cH, cV, cD).
    """
    dec_lo, dec_hi = bank[0], bank[1]
    a0 = dwt_axis(x, dec_lo, 0)
    d0 = dwt_axis(x, dec_hi, 0)
    aa = dwt_axis(a0, dec_lo, 1)
    ad = dwt_axis(a0, dec_hi, 1)
    da = dwt_axis(d0, dec_lo, 1)
    dd = dwt_axis(d0, dec_hi, 1)
    return aa, (da, ad, dd)


def idwt2(aa, details, bank=DB3_BANK):
    """One 2-D synthesis level (pywt ``idwtn``): axis 1 first, then axis 0; mixed dtypes upcast."""
    da, ad, dd = details
    rec_lo, rec_hi = bank[2], bank[3]
    dt = np.result_type(aa.dtype, da.dtype, ad.dtype, dd.dtype)
    aa, da, ad, dd = (c.astype(dt, copy=False) for c in (aa, da, ad, dd))
    a0 = idwt_axis(aa, ad, rec_lo, rec_hi, 1)
    d0 = idwt_axis(da, dd, rec_lo, rec_hi, 1)
    return idwt_axis(a0, d0, rec_lo, rec_hi, 0)


def resolve_level(shape, level, filter_len=FILTER_LEN):
    """Level selection of ``pywt.wavedec2`` (``_multilevel.py:_check_level``)."""
    max_level = min(dwt_max_level(s, filter_len) for s in shape[-2:])
    if level is None:
        return max_level
    if level < 0:
        raise ValueError("Level value of %d is too low . Minimum level is 0." % level)
    if level > max_level:
        warnings.warn(
            "Level value of {} is too high: all coefficients will experience boundary "
            "effects.".format(level),
            UserWarning,
        )
    return level


def wavedec2(x, level=None, bank=DB3_BANK):
    """``pywt.wavedec2(x, wavelet, level=level)``: ``[cA_L, (cH_L, cV_L, cD_L), ..., (cH_1, ...)]``."""
    level = resolve_level(x.shape, level, len(bank[0]))
    coeffs = []
    a = x
    for _ in range(level):
        a, det = dwt2(a, bank)
        coeffs.append(det)
    coeffs.append(a)
    coeffs.reverse()
    return coeffs


def waverec2(coeffs, bank=DB3_BANK):
    """``pywt.waverec2(coeffs, wavelet)`` incl. the one-too-long trim (``_multilevel.py:333-335``)."""
    a = coeffs[0]
    for det in coeffs[1:]:
        d_shape = det[0].shape
        if a.shape[-2] == d_shape[-2] + 1:
            a = a[..., :-1, :]
        if a.shape[-1] == d_shape[-1] + 1:
            a = a[..., :-1]
        a = idwt2(a, det, bank)
    return a


# ----------------------------------------------------------------------------------------------
# Otsu threshold (skimage 0.18.3 thresholding.py:282-350 on numpy.histogram)
# ----------------------------------------------------------------------------------------------
def histogram256(q, nbins=256, return_index=False):
    """``numpy.histogram(q.ravel(), bins=256)`` restated (numpy 1.26.4 histograms.py:800-850).

    Uniform edges ``linspace(min, max, 257, dtype=q.dtype)``; index estimate
    ``((q - min) / (max - min)) * 256`` corrected by +-1 against the edges; last bin closed.
    Returns (counts int64[256], edges[257]); with ``return_index`` also the bin index of every element (raveled).

    The edges follow the reference's pinned NumPy 1.26.4 (``environment/Dockerfile:14-29``;
    ``core/function_base.py:128-177``): there ``linspace`` promotes its float32 end points to
    float64 (``asanyarray(start) * 1.0`` under value-based casting), builds
    ``arange(257) * ((last - first) / 256) + first`` in float64 and only then rounds to the
    dtype of ``q``.  NumPy >= 2 (NEP 50) would build them in float32 -- about one edge in five
    then differs by an ulp -- so the rule is written out instead of calling ``np.linspace``.
    """
    a = q.ravel()
    first, last = a.min(), a.max()
    if first == last:
        first = first - 0.5
        last = last + 0.5
    f64_first, f64_last = np.float64(first), np.float64(last)
    edges64 = np.arange(nbins + 1, dtype=np.float64) * ((f64_last - f64_first) / nbins) + f64_first
    edges64[-1] = f64_last
    edges = edges64.astype(a.dtype)
    denom = last - first
    f_idx = ((a - first) / denom) * nbins
    idx = f_idx.astype(np.intp)
    idx[idx == nbins] -= 1
    idx[a < edges[idx]] -= 1
    inc = (a >= edges[idx + 1]) & (idx != nbins - 1)
    idx[inc] += 1
    counts = np.bincount(idx, minlength=nbins).astype(np.int64)
    if return_index:
        return counts, edges, idx
    return counts, edges


def otsu_variance_curve(counts, edges):
    """(bin centres, between-class variance per split) of skimage ``threshold_otsu`` given the 256-bin histogram."""
    counts = counts.astype(float)
    bin_centers = (edges[:-1] + edges[1:]) / 2.0
    with np.errstate(divide="ignore", invalid="ignore"):
        weight1 = np.cumsum(counts)
        weight2 = np.cumsum(counts[::-1])[::-1]
        mean1 = np.cumsum(counts * bin_centers) / weight1
        mean2 = (np.cumsum((counts * bin_centers)[::-1]) / weight2[::-1])[::-1]
        variance12 = weight1[:-1] * weight2[1:] * (mean1[:-1] - mean2[1:]) ** 2
    return bin_centers, variance12


def otsu_from_histogram(counts, edges):
    """Class-variance arg-max of skimage ``threshold_otsu`` given the 256-bin histogram."""
    bin_centers, variance12 = otsu_variance_curve(counts, edges)
    idx = int(np.argmax(variance12))
    return bin_centers[idx]


def threshold_otsu(q):
    """``skimage.filters.threshold_otsu(q)`` for a float image (returns a bin centre)."""
    first_pixel = q.ravel()[0]
    if np.all(q == first_pixel):
        return first_pixel
    counts, edges = histogram256(q)
    return otsu_from_histogram(counts, edges)


# ----------------------------------------------------------------------------------------------
# fftpack packed real FFT (scipy.fftpack.rfft / irfft, call sites filtering.py:206,215)
# ----------------------------------------------------------------------------------------------
def rfft_packed(x):
    """``scipy.fftpack.rfft(x, axis=-1)``: [Re0, Re1, Im1, Re2, Im2, ...] (even n ends with Re(n/2))."""
    n = x.shape[-1]
    c = np.fft.rfft(x, axis=-1)
    out = np.empty(x.shape, dtype=np.result_type(x.dtype, np.float32))
    out[..., 0] = c[..., 0].real
    nre = n // 2  # number of k>=1 real parts
    nim = (n - 1) // 2
    out[..., 1 : 2 * nre : 2] = c[..., 1 : nre + 1].real
    out[..., 2 : 2 * nim + 1 : 2] = c[..., 1 : nim + 1].imag
    return out


def irfft_packed(y):
    """``scipy.fftpack.irfft(y, axis=-1)`` (inverse of :func:`rfft_packed`, includes 1/n)."""
    n = y.shape[-1]
    nre = n // 2
    nim = (n - 1) // 2
    c = np.zeros(y.shape[:-1] + (n // 2 + 1,), dtype=np.complex128)
    c[..., 0] = y[..., 0]
    c[..., 1 : nre + 1] = y[..., 1 : 2 * nre : 2]
    c[..., 1 : nim + 1] += 1j * y[..., 2 : 2 * nim + 1 : 2]
    return np.fft.irfft(c, n=n, axis=-1)


# ----------------------------------------------------------------------------------------------
# reference-level functions
# ----------------------------------------------------------------------------------------------
def sigmoid(data):
    """``filtering.py:13-22``."""
    return 1 / (1 + np.exp(-data))


def foreground_fraction(img, center, crossover):
    """``filtering.py:25-51``."""
    z = (img - center) / crossover
    return sigmoid(z)


def get_foreground_background_mean(img, threshold_mask=0.3):
    """``filtering.py:54-88`` (float16 sigmoid mask, class means, 0.0 for an empty class)."""
    with np.errstate(over="ignore"):
        cell_for = foreground_fraction(img.astype(np.float16), 400, 20)
    cell_for[cell_for > threshold_mask] = 1
    cell_for[cell_for <= threshold_mask] = 0
    foreground = img[cell_for == 1]
    background = img[cell_for == 0]
    foreground_mean = foreground.mean() if foreground.size else 0.0
    background_mean = background.mean() if background.size else 0.0
    return foreground_mean, background_mean, cell_for


def notch(n, sigma):
    """``filtering.py:91-115``."""
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


def filter_level(ch, sigma_rows, max_threshold, stages=None, mask_override=None, otsu_override=None):
    """Body of the per-level loop, ``filtering.py:187-217``; returns ``ch_filtered`` (float64).

    ``sigma_rows`` is ``s = ch.shape[0] * width_fraction`` (``filtering.py:213``).
    ``mask_override`` (tests only): take the hard decisions ``|ch| > threshold`` from the caller instead
    -- the parity tests use it to show that, GIVEN the same decisions, the engine agrees everywhere.
    ``otsu_override`` (tests only): take the Otsu value from the caller -- used when the class-variance curve has
    two (near-)equal maxima and the engine sits on the other one (tests/parity_util.py, "Otsu ties").
    """
    ch_sq = ch**2
    ch_power = np.sqrt(ch_sq)
    otsu = threshold_otsu(ch_sq) if otsu_override is None else ch_sq.dtype.type(otsu_override)
    otsu_threshold_sqrt = np.sqrt(otsu)
    threshold = min(max_threshold, otsu_threshold_sqrt)
    mask = ch_power > threshold
    if mask_override is not None:
        mask = np.asarray(mask_override, dtype=bool)
    foreground = ch * mask
    background = ch * (1 - mask)  # int64 (1 - mask) promotes float32 -> float64 here
    med = np.median(background, axis=-1)
    background_inpainted = background + med[..., np.newaxis] * mask
    fft = rfft_packed(background_inpainted)
    g = gaussian_filter(shape=fft.shape, sigma=sigma_rows)
    background_filtered = irfft_packed(fft * g)
    ch_filtered = foreground + background_filtered * (1 - mask)
    if stages is not None:
        stages.append(
            {
                "ch": ch,
                "otsu": float(otsu),
                "threshold": float(threshold),
                "mask_count": int(mask.sum()),
                "median": med,
                "ch_filtered": ch_filtered,
            }
        )
    return ch_filtered


def log_space_fft_filtering(
    input_image, wavelet="db3", level=0, sigma=64, max_threshold=4, return_stages=False, mask_overrides=None,
    otsu_overrides=None
):
    """``filtering.py:139-224`` for a 2-D plane.

    Stage list (``return_stages=True``) is ordered coarse -> fine like the reference loop;
    ``mask_overrides`` / ``otsu_overrides`` (tests only, same order, entries may be None): see :func:`filter_level`.
    """
    bank = as_bank(wavelet)
    input_image = np.asarray(input_image)
    if input_image.ndim == 3:
        if return_stages or mask_overrides is not None or otsu_overrides is not None:
            raise ValueError("stages / overrides are for 2-D planes")
        return _log_space_fft_filtering_stack(input_image, bank, level, sigma, max_threshold)
    if input_image.ndim != 2:
        raise ValueError("the oracle restates the 2-D plane path and the 3-D stack mode only")
    input_image_log = np.log(1.0 + input_image)
    if input_image_log.dtype == np.float16:
        input_image_log = input_image_log.astype(np.float32)
    coeffs = wavedec2(input_image_log, level=level, bank=bank)
    approx, detail = coeffs[0], coeffs[1:]
    width_fraction = sigma / min(input_image.shape)
    stages = [] if return_stages else None
    coeff_filtered = [approx]
    for i, (ch, cv, cd) in enumerate(detail):
        s = ch.shape[0] * width_fraction
        ch_filtered = filter_level(ch, s, max_threshold, stages,
                                   None if mask_overrides is None else mask_overrides[i],
                                   None if otsu_overrides is None else otsu_overrides[i])
        coeff_filtered.append((ch_filtered, cv, cd))
    img_log_filtered = waverec2(coeff_filtered, bank)
    img_filtered = np.exp(img_log_filtered) + 1.0
    if return_stages:
        return img_filtered, stages
    return img_filtered


def _log_space_fft_filtering_stack(stack, bank, level, sigma, max_threshold):
    """The 3-D input mode of ``filtering.py:139-224``: ``pywt.wavedec2`` transforms the last two axes of every plane,
    ``width_fraction = sigma / min(shape[1:])`` (``:182-183``), and the per-level loop runs on the STACKED band -- one
    Otsu threshold per level over all planes (``threshold_otsu(ch_sq)`` sees the 3-D array, ``:188``), row medians
    and FFT along the last axis, ``s = fft.shape[1] * width_fraction`` (``:210-213``)."""
    log = np.log(1.0 + stack)
    if log.dtype == np.float16:
        log = log.astype(np.float32)
    per_plane = [wavedec2(p, level=level, bank=bank) for p in log]
    nlev = len(per_plane[0]) - 1
    width_fraction = sigma / min(stack.shape[1:])
    filtered = [[c[0]] for c in per_plane]
    for i in range(1, nlev + 1):
        ch = np.stack([c[i][0] for c in per_plane])
        s = ch.shape[1] * width_fraction
        ch_f = filter_level(ch, s, max_threshold)
        for k, c in enumerate(per_plane):
            filtered[k].append((ch_f[k], c[i][1], c[i][2]))
    out = np.stack([waverec2(f, bank) for f in filtered])
    return np.exp(out) + 1.0


def flatfield_correction(image_tiles, flatfield, darkfield, baseline=None):
    """``filtering.py:338-414``."""
    image_tiles = np.array(image_tiles)
    if image_tiles.ndim != flatfield.ndim:
        flatfield = np.expand_dims(flatfield, axis=0)
    if image_tiles.ndim != darkfield.ndim:
        darkfield = np.expand_dims(darkfield, axis=0)
    # NB: the reference crops the *leading* two axes (filtering.py:377), also for 3-D stacks
    darkfield = darkfield[: image_tiles.shape[-2], : image_tiles.shape[-1]]
    if darkfield.shape != image_tiles.shape:
        raise ValueError(
            "Please, check the shape of the darkfield. "
            "Image: {} - Darkfield: {}".format(image_tiles.shape, darkfield.shape)
        )
    if flatfield.shape != image_tiles.shape:
        raise ValueError(
            "Please, check the shape of the flatfield."
            "Image: {} - Flatfield: {}".format(image_tiles.shape, flatfield.shape)
        )
    if baseline is None:
        baseline = np.zeros((image_tiles.shape[0],))
    baseline_indxs = tuple([slice(None)] + ([np.newaxis] * (image_tiles.ndim - 1)))
    # in-place masked assignment of the reference (filtering.py:399-406): the difference is cast
    # back to the dtype of ``image_tiles`` (truncation for integer tiles)
    positive = image_tiles > darkfield
    out = np.where(positive, image_tiles - darkfield, 0).astype(image_tiles.dtype)
    out = out / flatfield - baseline[baseline_indxs]
    return np.clip(out, 0, 65535).astype("uint16")


def select_config(image, no_cells_config, cells_config, microscope_high_int=2700):
    """Decision of ``filtering.py:459-467``: returns (0 = no-cells, 1 = cells, fore_mean, back_mean)."""
    fore_mean, back_mean, _ = get_foreground_background_mean(image)
    use_cells = bool(fore_mean > back_mean and fore_mean > microscope_high_int)
    return (1 if use_cells else 0), float(fore_mean), float(back_mean)


def filter_stripes(
    image,
    input_tile_path,
    no_cells_config,
    cells_config,
    shadow_correction=None,
    microscope_high_int=2700,
):
    """``filtering.py:417-491`` (retrospective flat or explicit per-hemisphere flat list)."""
    which, _, _ = select_config(image, no_cells_config, cells_config, microscope_high_int)
    cfg = cells_config if which == 1 else no_cells_config
    filtered = log_space_fft_filtering(input_image=image, **cfg)
    if shadow_correction is not None:
        flatfield = shadow_correction.get("flatfield")
        darkfield = shadow_correction.get("darkfield")
        if not shadow_correction.get("retrospective"):
            flatfield = get_hemisphere_flatfield(
                input_tile_path, shadow_correction.get("tile_config"), flatfield
            )
        filtered = flatfield_correction(filtered, flatfield, darkfield)
    return filtered


def get_hemisphere_flatfield(input_tile_path, tile_config, flatfields, zarr=True):
    """``filtering.py:273-335``."""
    if zarr:
        parts = str(input_tile_path).split("_")
    else:
        parts = str(input_tile_path).split("/")[-2].split("_")
    x_folder, y_folder = parts[0], parts[1]
    if tile_config.get(x_folder) is None:
        raise KeyError("Please, check the tile config while trying to reach: {}".format(x_folder))
    brain_side = tile_config[x_folder].get(y_folder)
    if brain_side is None:
        raise KeyError("Please, check the tile config while trying to reach: {}".format(y_folder))
    return flatfields[brain_side]
