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

# ---- files imageio / Pillow cannot WRITE but the reference READS through them: Adam7-interlaced and palette images ----
# (round 4).  They are encoded by hand below (PNG specification: passes, scanline filters, PLTE / tRNS chunks), read back
# with the REAL imageio, and what it returns is stored in tests/golden/png/expected_r4.npz: the files + those arrays pin
# ``mini_png.imread`` -- including imageio's own palette conventions (grey palette -> 2-D, anything else -> RGBA).
import struct  # noqa: E402
import zlib  # noqa: E402

ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))


def _chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _scanlines(rows, bpp, first_filter):
    """rows: uint8 [h, stride] -> filtered bytes; filter types cycle 0..4 from ``first_filter`` so that every pass uses all."""
    out = bytearray()
    prev = np.zeros(rows.shape[1], np.int64)
    for y, row in enumerate(rows.astype(np.int64)):
        kind = (first_filter + y) % 5
        left = np.concatenate([np.zeros(bpp, np.int64), row[:-bpp]]) if row.size > bpp else np.zeros_like(row)
        upleft = np.concatenate([np.zeros(bpp, np.int64), prev[:-bpp]]) if row.size > bpp else np.zeros_like(row)
        if kind == 0:
            f = row
        elif kind == 1:
            f = row - left
        elif kind == 2:
            f = row - prev
        elif kind == 3:
            f = row - ((left + prev) >> 1)
        else:
            f = row - np.array([_paeth(int(a), int(b), int(c)) for a, b, c in zip(left, prev, upleft)], np.int64)
        out.append(kind)
        out += (f & 0xFF).astype(np.uint8).tobytes()
        prev = row
    return bytes(out)


def _pack(samples, depth):
    """samples [h, w * ch] -> uint8 rows [h, stride] (big-endian 16-bit samples; 1 / 2 / 4-bit samples packed MSB first)."""
    h = samples.shape[0]
    if depth == 16:
        return samples.astype(">u2").view(np.uint8).reshape(h, -1)
    if depth == 8:
        return samples.astype(np.uint8).reshape(h, -1)
    bits = ((samples[..., None].astype(np.uint8) >> np.arange(depth - 1, -1, -1, dtype=np.uint8)) & 1).reshape(h, -1)
    return np.packbits(bits, axis=1)


def encode_png(samples, depth, ctype, interlace, plte=None, trns=None):
    """samples: [H, W] or [H, W, ch] integer array -> PNG bytes (hand-written encoder for the fixtures only)."""
    a = samples if samples.ndim == 3 else samples[..., None]
    H, W, ch = a.shape
    bpp = max(1, ch * depth // 8)
    if interlace:
        body, k = b"", 0
        for x0, y0, dx, dy in ADAM7:
            sub = a[y0::dy, x0::dx]
            if sub.shape[0] == 0 or sub.shape[1] == 0:
                continue
            body += _scanlines(_pack(sub.reshape(sub.shape[0], -1), depth), bpp, k)
            k += 1
    else:
        body = _scanlines(_pack(a.reshape(H, -1), depth), bpp, 1)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        out += _chunk(b"PLTE", bytes(plte))
    if trns is not None:
        out += _chunk(b"tRNS", bytes(trns))
    return out + _chunk(b"IDAT", zlib.compress(body, 6)) + _chunk(b"IEND", b"")


def r4_cases():
    rs = np.random.RandomState(44)
    grey_pal = np.repeat(np.arange(0, 256, 4, dtype=np.uint8)[:, None], 3, axis=1)[:50]          # 50 grey entries
    col_pal = rs.randint(0, 256, (200, 3)).astype(np.uint8)
    cases = {}
    # Adam7, every layout the plain reader knows; odd sizes leave some passes empty or one pixel wide
    cases["a7_u16_gray.png"] = encode_png(plane(11, (37, 53), np.uint16), 16, 0, True)
    cases["a7_u8_gray_tiny.png"] = encode_png(plane(12, (3, 2), np.uint8), 8, 0, True)
    cases["a7_u8_rgb.png"] = encode_png(plane(13, (21, 19, 3), np.uint8), 8, 2, True)
    cases["a7_u16_rgba.png"] = encode_png(plane(14, (9, 13, 4), np.uint16), 16, 6, True)
    cases["a7_u8_graya.png"] = encode_png(plane(15, (16, 16, 2), np.uint8), 8, 4, True)
    cases["a7_g4.png"] = encode_png(rs.randint(0, 16, (13, 11)), 4, 0, True)
    cases["a7_g1.png"] = encode_png(rs.randint(0, 2, (10, 17)), 1, 0, True)
    # layouts imageio / Pillow CONVERT on the way in (none of them is what a microscope writes; pinned all the same):
    # 1 / 2 / 4-bit grey is scaled to 8 bits, grey + alpha comes back as RGBA, 16-bit colour as its high bytes
    cases["g1.png"] = encode_png(rs.randint(0, 2, (9, 19)), 1, 0, False)
    cases["g2.png"] = encode_png(rs.randint(0, 4, (7, 13)), 2, 0, False)
    cases["g4.png"] = encode_png(rs.randint(0, 16, (11, 9)), 4, 0, False)
    cases["u8_graya.png"] = encode_png(plane(16, (12, 10, 2), np.uint8), 8, 4, False)
    cases["u16_graya.png"] = encode_png(plane(17, (8, 11, 2), np.uint16), 16, 4, False)
    cases["u16_rgb.png"] = encode_png(plane(18, (10, 7, 3), np.uint16), 16, 2, False)
    cases["u16_rgba.png"] = encode_png(plane(19, (6, 9, 4), np.uint16), 16, 6, False)
    # palette images, plain and interlaced
    cases["p8_grey.png"] = encode_png(rs.randint(3, 47, (20, 24)), 8, 3, False, plte=grey_pal.tobytes())
    cases["p8_colour.png"] = encode_png(rs.randint(0, 200, (18, 22)), 8, 3, False, plte=col_pal.tobytes())
    cases["p8_colour_trns.png"] = encode_png(rs.randint(0, 200, (18, 22)), 8, 3, False, plte=col_pal.tobytes(),
                                            trns=rs.randint(0, 256, 120).astype(np.uint8).tobytes())
    cases["p8_one_transparent.png"] = encode_png(rs.randint(0, 60, (12, 15)), 8, 3, False, plte=col_pal[:60].tobytes(),
                                                 trns=bytes([255] * 7 + [0] + [255] * 3))
    # Pillow reports the single transparent entry 0 as the integer 0, which imageio's grey test reads as "no transparency"
    cases["p8_grey_index0_transparent.png"] = encode_png(rs.randint(0, 50, (12, 15)), 8, 3, False, plte=grey_pal.tobytes(),
                                                         trns=bytes([0]))
    cases["p8_grey_trns.png"] = encode_png(rs.randint(0, 50, (12, 15)), 8, 3, False, plte=grey_pal.tobytes(),
                                           trns=bytes([255, 128, 0, 7]))
    cases["p4_colour.png"] = encode_png(rs.randint(0, 16, (11, 14)), 4, 3, False, plte=col_pal[:16].tobytes())
    cases["p2_grey.png"] = encode_png(rs.randint(0, 4, (9, 10)), 2, 3, False, plte=grey_pal[:4].tobytes())
    cases["p1_colour.png"] = encode_png(rs.randint(0, 2, (8, 21)), 1, 3, False, plte=col_pal[:2].tobytes())
    cases["a7_p8_colour_trns.png"] = encode_png(rs.randint(0, 200, (17, 23)), 8, 3, True, plte=col_pal.tobytes(),
                                               trns=rs.randint(0, 256, 200).astype(np.uint8).tobytes())
    cases["a7_p4_grey.png"] = encode_png(rs.randint(0, 16, (10, 9)), 4, 3, True, plte=grey_pal[:16].tobytes())
    return cases


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name, c in CASES.items():
        img = plane(c["seed"], c["shape"], c["dtype"])
        imageio.imwrite(os.path.join(OUT, name), img, **c["kw"])
        back = imageio.imread(os.path.join(OUT, name))
        assert back.dtype == img.dtype and np.array_equal(back, img), name
        print(name, img.shape, img.dtype, os.path.getsize(os.path.join(OUT, name)))
    expected = {}
    for name, data in r4_cases().items():
        with open(os.path.join(OUT, name), "wb") as f:
            f.write(data)
        back = np.asarray(imageio.imread(os.path.join(OUT, name)))  # the REAL imageio decides what the file holds
        expected[name] = back
        print(name, back.shape, back.dtype, len(data))
    np.savez_compressed(os.path.join(OUT, "expected_r4.npz"), **expected)
