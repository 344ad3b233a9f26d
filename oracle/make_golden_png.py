"""Writes the PNG fixtures under tests/golden/png/ with the real ``imageio`` (TEST INFRASTRUCTURE ONLY).

Run in this container with the interpreter that has imageio (2.9.0, Pillow 8.4.0 plugin):

    /opt/conda/bin/python3.9 oracle/make_golden_png.py

imageio is what the reference reads and writes PNG planes with (``readers.py:86-87``, ``destriper.py:107-110``);
the files pin ``mini_png.imread``.  The arrays are regenerated from seeds by ``tests/test_tiff_modes.py``.
"""

import os

import imageio
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "png")


def plane(seed, shape, dtype):
    rs = np.random.RandomState(seed)
    # smooth + noise: the adaptive filter of the writer then uses every filter type
    base = np.add.outer(np.arange(shape[0]) * 7, np.arange(shape[1]) * 3)
    if len(shape) == 3:
        base = base[..., None] + np.arange(shape[2]) * 11
    hi = np.iinfo(dtype).max
    return ((base * (hi // 512) + rs.randint(0, hi // 64 + 2, shape)) % (hi + 1)).astype(dtype)


CASES = {
    "u16_gray.png": dict(seed=1, shape=(37, 53), dtype=np.uint16, kw=dict(compress_level=1)),
    "u16_gray_c9.png": dict(seed=2, shape=(64, 40), dtype=np.uint16, kw=dict(compress_level=9)),
    "u8_gray.png": dict(seed=3, shape=(29, 31), dtype=np.uint8, kw={}),
    "u8_rgb.png": dict(seed=4, shape=(20, 24, 3), dtype=np.uint8, kw={}),
    "u8_rgba.png": dict(seed=5, shape=(18, 22, 4), dtype=np.uint8, kw={}),
    "u16_gray_c0.png": dict(seed=6, shape=(33, 17), dtype=np.uint16, kw=dict(compress_level=0)),
}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name, c in CASES.items():
        img = plane(c["seed"], c["shape"], c["dtype"])
        imageio.imwrite(os.path.join(OUT, name), img, **c["kw"])
        back = imageio.imread(os.path.join(OUT, name))
        assert back.dtype == img.dtype and np.array_equal(back, img), name
        print(name, img.shape, img.dtype, os.path.getsize(os.path.join(OUT, name)))
