"""PNG planes for the directory mode (reference: ``imageio`` -- ``iio.imread``, ``readers.py:86-87``;
``iio.v3.imwrite(filename, img, compress_level=compression)``, ``destriper.py:107-110``).

imageio is not installed here, so the two calls the reference makes are restated on the PNG specification (ISO/IEC
15948): zlib streams from the standard library, scanline un-filtering (Sub / Up / Average / Paeth are sequential
along a row) in the native library (``dsx_png_unfilter``).  Reader: non-interlaced greyscale, greyscale + alpha,
RGB, RGBA at 8 / 16 bits and greyscale at 1 / 2 / 4 bits, returned as ``imageio`` returns them (``[H, W]`` or
``[H, W, C]``, ``uint8`` / ``uint16``).  Writer: ``uint8`` / ``uint16`` arrays of those layouts, per-row adaptive
filter (minimum sum of absolute differences).  Pinned by files written with the real imageio 2.9.0 / Pillow 8.4.0
(``oracle/make_golden_png.py``) and cross-read with the Pillow of this image in the tests.
"""

import struct
import zlib

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


def _native():
    from . import engine

    return engine.load_library()


def imread(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _SIG:
        raise ValueError("{} is not a PNG file".format(path))
    pos, idat, head = 8, [], None
    while pos + 8 <= len(data):
        n, kind = struct.unpack(">I4s", data[pos : pos + 8])
        body = data[pos + 8 : pos + 8 + n]
        if kind == b"IHDR":
            head = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
        pos += 12 + n
    if head is None or not idat:
        raise ValueError("{}: no image data".format(path))
    width, height, depth, ctype, _, _, interlace = head
    if interlace:
        raise NotImplementedError("interlaced (Adam7) PNG files are not supported")
    if ctype == 3:
        raise NotImplementedError("palette PNG files are not supported")
    if ctype not in _CHANNELS or depth not in (1, 2, 4, 8, 16) or (depth < 8 and ctype != 0):
        raise ValueError("{}: unsupported colour type {} / bit depth {}".format(path, ctype, depth))
    ch = _CHANNELS[ctype]
    stride = (width * ch * depth + 7) // 8
    bpp = max(1, ch * depth // 8)
    raw = bytearray(zlib.decompress(b"".join(idat)))
    if len(raw) != height * (stride + 1):
        raise ValueError("{}: image data has {} bytes, expected {}".format(path, len(raw), height * (stride + 1)))
    import ctypes

    buf = (ctypes.c_ubyte * len(raw)).from_buffer(raw)
    lib = _native()
    if lib.dsx_png_unfilter(buf, height, stride, bpp) != 0:
        raise ValueError("{}: {}".format(path, (lib.dsx_last_error(None) or b"bad filter type").decode()))
    rows = np.frombuffer(raw, np.uint8).reshape(height, stride + 1)[:, 1:]
    if depth == 16:
        img = rows.reshape(height, width * ch, 2).astype(np.uint16)
        img = (img[..., 0] << 8) | img[..., 1]  # samples are big-endian
    elif depth == 8:
        img = rows.copy()
    else:
        img = np.unpackbits(rows, axis=1)[:, : width * depth].reshape(height, width, depth)
        img = (img * (1 << np.arange(depth - 1, -1, -1, dtype=np.uint8))).sum(axis=2).astype(np.uint8)
        if depth == 1:
            img = img.astype(bool)  # imageio / Pillow mode "1"
        return img
    return img.reshape(height, width) if ch == 1 else img.reshape(height, width, ch)


def _chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def _filtered(rows, bpp):
    """Per-row adaptive filtering: rows uint8 [H, stride] -> bytes with the filter-type byte in front of every row."""
    h, stride = rows.shape
    cur = rows.astype(np.int16)
    left = np.zeros_like(cur)
    left[:, bpp:] = cur[:, :-bpp]
    up = np.zeros_like(cur)
    up[1:] = cur[:-1]
    upleft = np.zeros_like(cur)
    upleft[1:, bpp:] = cur[:-1, :-bpp]
    p = left + up - upleft
    pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - upleft)
    paeth = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
    cands = [cur, cur - left, cur - up, cur - ((left + up) >> 1), cur - paeth]
    cands = [(c & 0xFF).astype(np.uint8) for c in cands]
    cost = np.stack([np.abs(c.view(np.int8).astype(np.int16)).sum(axis=1) for c in cands])  # [5, H]
    best = cost.argmin(axis=0)
    out = np.empty((h, stride + 1), np.uint8)
    out[:, 0] = best
    stack = np.stack(cands)  # [5, H, stride]
    out[:, 1:] = stack[best, np.arange(h)]
    return out.tobytes()


def imwrite(path, img, compress_level=1):
    img = np.asarray(img)
    if img.dtype == bool:
        img = img.astype(np.uint8) * 255
    if img.dtype not in (np.uint8, np.uint16):
        raise TypeError("Cannot handle this data type: {} (PNG planes are uint8 or uint16)".format(img.dtype))
    if img.ndim == 2:
        ch = 1
    elif img.ndim == 3 and img.shape[2] in (2, 3, 4):
        ch = img.shape[2]
    else:
        raise ValueError("a PNG image is [H, W] or [H, W, 2 | 3 | 4]")
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    h, w = img.shape[:2]
    depth = 8 * img.dtype.itemsize
    rows = (img.astype(">u2") if depth == 16 else img).reshape(h, -1).view(np.uint8)
    body = zlib.compress(_filtered(np.ascontiguousarray(rows), ch * depth // 8), int(compress_level))
    with open(path, "wb") as f:
        f.write(_SIG)
        f.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)))
        f.write(_chunk(b"IDAT", body))
        f.write(_chunk(b"IEND", b""))
