"""PNG planes for the directory mode (reference: ``imageio`` -- ``iio.imread``, ``readers.py:86-87``;
``iio.v3.imwrite(filename, img, compress_level=compression)``, ``destriper.py:107-110``).

imageio is not installed here, so the two calls the reference makes are restated on the PNG specification (ISO/IEC
15948): zlib streams from the standard library, scanline un-filtering (Sub / Up / Average / Paeth are sequential
along a row) in the native library (``dsx_png_unfilter``).  Reader: greyscale, greyscale + alpha, RGB, RGBA at 8 / 16
bits, greyscale at 1 / 2 / 4 bits and palette images (1 … 8 bits, ``PLTE`` + ``tRNS``), plain or Adam7-interlaced,
returned as ``imageio`` returns them (8 / 16-bit grey ``[H, W]`` as stored; 1 / 2 / 4-bit grey scaled to ``uint8``; RGB /
RGBA ``[H, W, C]`` ``uint8``, 16-bit colour as its high bytes; grey + alpha as RGBA; a palette image as Pillow's
``convert("L")`` when its palette is grey, else as RGBA).  Writer: ``uint8`` / ``uint16`` arrays of those layouts, per-row adaptive
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


_ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))  # x0, y0, dx, dy


def _samples(raw, offset, height, width, ch, depth, path):
    """Un-filter ``height`` scanlines of ``width`` pixels starting at ``raw[offset]`` (filter-type byte in front of every
    row) and return ``(samples [height, width * ch] as uint8 / uint16, bytes consumed)``."""
    import ctypes

    stride = (width * ch * depth + 7) // 8
    bpp = max(1, ch * depth // 8)
    nbytes = height * (stride + 1)
    if offset + nbytes > len(raw):
        raise ValueError("{}: image data ends early".format(path))
    buf = (ctypes.c_ubyte * nbytes).from_buffer(raw, offset)
    lib = _native()
    if lib.dsx_png_unfilter(buf, height, stride, bpp) != 0:
        raise ValueError("{}: {}".format(path, (lib.dsx_last_error(None) or b"bad filter type").decode()))
    rows = np.frombuffer(raw, np.uint8, nbytes, offset).reshape(height, stride + 1)[:, 1:]
    if depth == 16:
        img = rows.reshape(height, width * ch, 2).astype(np.uint16)
        img = (img[..., 0] << 8) | img[..., 1]  # samples are big-endian
    elif depth == 8:
        img = rows.copy()
    else:
        img = np.unpackbits(rows, axis=1)[:, : width * depth].reshape(height, width, depth)
        img = (img * (1 << np.arange(depth - 1, -1, -1, dtype=np.uint8))).sum(axis=2).astype(np.uint8)
    return img, nbytes


def _apply_palette(idx, plte, trns):
    """A palette image as ``imageio`` 2.9 (Pillow plugin, ``pil_get_frame``) returns it: a palette that is grey over the
    index range in use and carries no (truthy) transparency -> ``[H, W]`` uint8 (Pillow's ``convert("L")``);
    otherwise ``[H, W, 4]`` uint8 -- alpha from ``tRNS`` where the file has one, 255 elsewhere."""
    n = len(plte) // 3
    pal = np.zeros((256, 3), np.uint8)
    pal[:n] = np.frombuffer(plte, np.uint8, 3 * n).reshape(n, 3)
    # Pillow's info["transparency"]: the index of the one fully transparent entry of an otherwise opaque table, else the bytes
    transparency = None
    if trns is not None:
        simple = trns.count(b"\x00") == 1 and trns.replace(b"\x00", b"").strip(b"\xff") == b""
        transparency = trns.find(b"\x00") if simple else trns
    lo, hi = int(idx.min()), int(idx.max())
    valid = pal[lo : hi + 1].astype(np.int64)
    if not transparency and np.all(np.diff(valid, axis=1) == 0):
        # L = (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16 (Pillow's ITU-R 601 weights): the grey value itself
        p = pal.astype(np.uint32)
        lum = ((p[:, 0] * 19595 + p[:, 1] * 38470 + p[:, 2] * 7471 + 0x8000) >> 16).astype(np.uint8)
        return lum[idx]
    alpha = np.full(256, 255, np.uint8)
    if isinstance(transparency, int):
        alpha[transparency] = 0
    elif transparency is not None:
        alpha[: len(transparency)] = np.frombuffer(transparency, np.uint8)
    return np.concatenate([pal, alpha[:, None]], axis=1)[idx]


def imread(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _SIG:
        raise ValueError("{} is not a PNG file".format(path))
    pos, idat, head, plte, trns = 8, [], None, None, None
    while pos + 8 <= len(data):
        n, kind = struct.unpack(">I4s", data[pos : pos + 8])
        body = data[pos + 8 : pos + 8 + n]
        if kind == b"IHDR":
            head = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            plte = body
        elif kind == b"tRNS":
            trns = body
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
        pos += 12 + n
    if head is None or not idat:
        raise ValueError("{}: no image data".format(path))
    width, height, depth, ctype, _, _, interlace = head
    if interlace not in (0, 1):
        raise ValueError("{}: unknown interlace method {}".format(path, interlace))
    if ctype == 3:
        if plte is None or len(plte) < 3 or depth not in (1, 2, 4, 8):
            raise ValueError("{}: palette image without a usable PLTE chunk / bit depth {}".format(path, depth))
        ch = 1
    elif ctype not in _CHANNELS or depth not in (1, 2, 4, 8, 16) or (depth < 8 and ctype != 0):
        raise ValueError("{}: unsupported colour type {} / bit depth {}".format(path, ctype, depth))
    else:
        ch = _CHANNELS[ctype]
    raw = bytearray(zlib.decompress(b"".join(idat)))
    if not interlace:
        if len(raw) != height * ((width * ch * depth + 7) // 8 + 1):
            raise ValueError("{}: image data has {} bytes, expected {}".format(
                path, len(raw), height * ((width * ch * depth + 7) // 8 + 1)))
        img, _ = _samples(raw, 0, height, width, ch, depth, path)
    else:
        # Adam7: seven reduced images, each with its own scanlines and filters; empty passes take no bytes
        img = np.zeros((height, width * ch), np.uint16 if depth == 16 else np.uint8)
        view = img.reshape(height, width, ch)
        off = 0
        for x0, y0, dx, dy in _ADAM7:
            pw, ph = (width - x0 + dx - 1) // dx, (height - y0 + dy - 1) // dy
            if pw <= 0 or ph <= 0:
                continue
            sub, used = _samples(raw, off, ph, pw, ch, depth, path)
            off += used
            view[y0::dy, x0::dx] = sub.reshape(ph, pw, ch)
        if off != len(raw):
            raise ValueError("{}: image data has {} bytes, the seven passes take {}".format(path, len(raw), off))
    if ctype == 3:
        return _apply_palette(img.reshape(height, width), plte, trns)
    # What imageio 2.9 / Pillow 8.4 convert on the way in (pinned by files read with the real imageio, expected_r4.npz):
    if depth < 8:  # 1 / 2 / 4-bit grey is scaled to 8 bits (Pillow raw modes "1" -> convert("L"), "L;2", "L;4")
        return (img.reshape(height, width) * {1: 255, 2: 85, 4: 17}[depth]).astype(np.uint8)
    if ch == 1:
        return img.reshape(height, width)  # 8 / 16-bit grey: as stored (what the microscope planes are)
    img = img.reshape(height, width, ch)
    if depth == 16:  # Pillow has no 16-bit multi-channel mode: the high byte of every sample
        img = (img >> 8).astype(np.uint8)
    if ch == 2:  # grey + alpha comes back as RGBA
        img = np.concatenate([np.repeat(img[..., :1], 3, axis=2), img[..., 1:]], axis=2)
    return img


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
