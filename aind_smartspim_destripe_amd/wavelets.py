"""Filter banks behind the config key ``"wavelet"`` (reference: ``pywt.wavedec2(x, wavelet=...)``,
``filtering.py:176``, and ``pywt.waverec2(..., wavelet)``, ``:221``).

PyWavelets is not a dependency of this package: the coefficients of its discrete wavelets live in
``wavelet_table.npz`` (data written by ``oracle/make_wavelet_table.py`` from PyWavelets 1.1.1).  ``db3`` -- the
production setting, ``run_capsule.py:374-390`` -- runs through the specialised kernels; every other name goes
through the tap-count-generic kernels of ``csrc/dsx_wavelet.h``.
"""

import os

import numpy as np

_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wavelet_table.npz")
_cache = {}
MAX_TAPS = 104  # kernel-argument capacity (csrc/dsx_wavelet.h: kMaxTaps)


def _load():
    if not _cache:
        with np.load(_TABLE) as z:
            names = [n.decode() for n in z["names"]]
            _cache["index"] = {n: (int(o), int(k)) for n, o, k in zip(names, z["offset"], z["length"])}
            for key in ("dec_lo", "dec_hi", "rec_lo", "rec_hi"):
                _cache[key] = np.array(z[key], dtype=np.float64)
    return _cache


def wavelist():
    """Names the engine knows (``pywt.wavelist(kind="discrete")`` of PyWavelets 1.1.1)."""
    return sorted(_load()["index"])


def filter_bank(name):
    """``(dec_lo, dec_hi, rec_lo, rec_hi)`` as float64 arrays of one length.

    Accepts a name or anything with those four attributes (a ``pywt.Wavelet``).  Unknown names raise
    ``ValueError`` with PyWavelets' wording; ``dmey`` is refused: it is not a perfect-reconstruction bank
    (round trip of ``wavedec2`` / ``waverec2`` off by 2e-2), and the engine reconstructs the correction only
    (SURVEY appendix B.1), which is exact for perfect-reconstruction banks alone.
    """
    if all(hasattr(name, k) for k in ("dec_lo", "dec_hi", "rec_lo", "rec_hi")):
        bank = tuple(np.asarray(getattr(name, k), dtype=np.float64) for k in ("dec_lo", "dec_hi", "rec_lo", "rec_hi"))
        name = getattr(name, "name", "custom")
    else:
        t = _load()
        key = str(name).lower()
        if key not in t["index"]:
            raise ValueError(
                "Unknown wavelet name '{}', check wavelist() for the list of available builtin wavelets.".format(name)
            )
        o, k = t["index"][key]
        bank = tuple(t[f][o : o + k].copy() for f in ("dec_lo", "dec_hi", "rec_lo", "rec_hi"))
        name = key
    if name == "dmey":
        raise ValueError("wavelet 'dmey' is not a perfect-reconstruction filter bank and is not supported")
    n = len(bank[0])
    if n < 2 or any(len(b) != n for b in bank):
        raise ValueError("the four filters of a wavelet must have one length >= 2")
    if n > MAX_TAPS:
        raise ValueError("wavelet '{}' has {} taps; the kernels take at most {}".format(name, n, MAX_TAPS))
    return bank


def filter_length(name):
    return len(filter_bank(name)[0])
