"""Writes the TIFF fixtures under tests/golden/tiff/ with the real ``tifffile`` (TEST INFRASTRUCTURE ONLY).

Run in this container with the interpreter that has tifffile (2021.7.2):

    /opt/conda/bin/python3.9 oracle/make_golden_tiff.py

The arrays are regenerated from seeds by ``tests/test_tiff_modes.py`` (``np.random.RandomState`` streams are
stable across NumPy versions), so only the encoded files are committed.  tifffile is what the reference reads
and writes planes with (``readers.py:85-86``, ``destriper.py:71-103``); the files pin ``mini_tiff.imread``.
"""

import os

import numpy as np
import tifffile

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tiff")


def plane(seed, shape=(37, 53), dtype=np.uint16):
    rs = np.random.RandomState(seed)
    if np.dtype(dtype).kind == "f":
        return rs.rand(*shape).astype(dtype)
    return rs.randint(0, np.iinfo(dtype).max + 1, shape).astype(dtype)


CASES = {
    "u16_le.tif": dict(seed=1),
    "u16_be.tif": dict(seed=2, kw=dict(byteorder=">")),
    "u16_deflate.tif": dict(seed=3, kw=dict(compression="zlib")),
    "u16_deflate_pred.tif": dict(seed=4, kw=dict(compression="zlib", predictor=True)),
    "u16_big.tif": dict(seed=5, kw=dict(bigtiff=True)),
    "u16_pages.tif": dict(seed=6, shape=(3, 20, 24), kw=dict(photometric="minisblack")),
    "f32.tif": dict(seed=7, dtype=np.float32),
    "u16_tiled.tif": dict(seed=8, kw=dict(tile=(16, 16))),
    "u8_strips.tif": dict(seed=9, dtype=np.uint8, kw=dict(rowsperstrip=5)),
}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name, c in CASES.items():
        a = plane(c["seed"], c.get("shape", (37, 53)), c.get("dtype", np.uint16))
        tifffile.imwrite(os.path.join(OUT, name), a, **c.get("kw", {}))
        back = tifffile.imread(os.path.join(OUT, name))
        assert np.array_equal(back, a)
        print(name, a.shape, a.dtype, os.path.getsize(os.path.join(OUT, name)))
    print("tifffile", tifffile.__version__, "numpy", np.__version__)
