"""Write aind_smartspim_destripe_amd/wavelet_table.npz: the filter banks of PyWavelets' discrete wavelets.

Run (only here; PyWavelets is not installed in the product's interpreter)::

    /opt/conda/bin/python3.9 -W ignore oracle/make_wavelet_table.py

The reference hands its config's ``wavelet`` name to ``pywt.wavedec2`` / ``pywt.waverec2``
(``filtering.py:176, 221``); the engine needs the four filters behind a name.  The table holds data only
(names, lengths, coefficients as printed by PyWavelets 1.1.1): ``dec_lo``, ``dec_hi``, ``rec_lo``, ``rec_hi``
of every ``pywt.wavelist(kind="discrete")`` entry, concatenated, with ``offset`` / ``length`` per name.
"""

import os

import numpy as np
import pywt

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "aind_smartspim_destripe_amd", "wavelet_table.npz")


def main():
    names, offs, lens = [], [], []
    bank = {"dec_lo": [], "dec_hi": [], "rec_lo": [], "rec_hi": []}
    pos = 0
    for name in pywt.wavelist(kind="discrete"):
        w = pywt.Wavelet(name)
        assert w.dec_len == w.rec_len
        names.append(name)
        offs.append(pos)
        lens.append(w.dec_len)
        pos += w.dec_len
        for k in bank:
            bank[k].extend(getattr(w, k))
    np.savez_compressed(
        OUT,
        names=np.array(names, dtype="S16"),
        offset=np.array(offs, dtype=np.int32),
        length=np.array(lens, dtype=np.int32),
        pywt_version=np.array(pywt.__version__, dtype="S16"),
        **{k: np.array(v, dtype=np.float64) for k, v in bank.items()},
    )
    print(OUT, len(names), "wavelets,", pos, "taps per filter,", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
