"""Minimal TIFF reader / writer for the directory mode and the microscope flats (rows f2 / f4).

The reference reads and writes planes with ``tifffile`` (``readers.py:85-86``, ``destriper.py:71-103``,
``zarr_destriper.py:139-145, 1107``), which is not installed here.  This module covers what SmartSPIM
acquisitions and the reference's own writer produce: grayscale planes (one sample per pixel) of 8 / 16 / 32 /
64-bit unsigned, signed or float samples, little- or big-endian, classic TIFF or BigTIFF, strips or tiles,
uncompressed or Deflate (with or without the horizontal predictor), one or many pages.  LZW / JPEG / colour
images raise ``NotImplementedError``.  Files written here are plain little-endian baseline TIFF (one strip
per page) -- ``tifffile.imsave(path, img, compressionargs={"level": n})`` as called by the reference
(``destriper.py:73-77``) passes no ``compression=`` and therefore also writes uncompressed strips.
"""

import struct
import zlib

import numpy as np

_TYPES = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 7: "B", 8: "h", 9: "i", 10: "ii", 11: "f", 12: "d",
          13: "I", 16: "Q", 17: "q", 18: "Q"}  # fmt: skip
_DEFLATE = (8, 32946)


class TiffError(ValueError):
    pass


def _read_ifd(buf, off, bo, big):
    if big:
        (n,) = struct.unpack_from(bo + "Q", buf, off)
        pos, esz, cnt_fmt, inline = off + 8, 20, "Q", 8
    else:
        (n,) = struct.unpack_from(bo + "H", buf, off)
        pos, esz, cnt_fmt, inline = off + 2, 12, "I", 4
    tags = {}
    for i in range(n):
        e = pos + i * esz
        tag, typ = struct.unpack_from(bo + "HH", buf, e)
        (count,) = struct.unpack_from(bo + cnt_fmt, buf, e + 4)
        fmt = _TYPES.get(typ)
        if fmt is None:
            continue
        unit = struct.calcsize("=" + fmt)
        voff = e + 4 + inline
        if unit * count > inline:
            (voff,) = struct.unpack_from(bo + cnt_fmt, buf, voff)
        if voff + unit * count > len(buf):
            raise TiffError("tag {} points outside the file".format(tag))
        if typ == 2:
            tags[tag] = bytes(buf[voff : voff + count]).split(b"\0")[0].decode("latin-1")
        else:
            vals = struct.unpack_from(bo + fmt * count, buf, voff)
            tags[tag] = vals
    (nxt,) = struct.unpack_from(bo + ("Q" if big else "I"), buf, pos + n * esz)
    return tags, nxt


def _dtype(tags, bo):
    bits = tags.get(258, (1,))
    fmt = tags.get(339, (1,))[0]
    if len(set(bits)) != 1:
        raise NotImplementedError("mixed bits per sample")
    kind = {1: "u", 2: "i", 3: "f"}.get(fmt)
    if kind is None or bits[0] not in (8, 16, 32, 64) or (kind == "f" and bits[0] < 32):
        raise NotImplementedError("sample format {} with {} bits".format(fmt, bits[0]))
    return np.dtype(("<" if bo == "<" else ">") + kind + str(bits[0] // 8))


def _decode(raw, comp):
    if comp == 1:
        return raw
    if comp in _DEFLATE:
        return zlib.decompress(raw)
    raise NotImplementedError("TIFF compression {} (only none / deflate)".format(comp))


def _read_page(buf, tags, bo):
    W, H = int(tags[256][0]), int(tags[257][0])
    if tags.get(277, (1,))[0] != 1:
        raise NotImplementedError("only one sample per pixel (grayscale planes)")
    dt = _dtype(tags, bo)
    comp = tags.get(259, (1,))[0]
    pred = tags.get(317, (1,))[0]
    if pred not in (1, 2):
        raise NotImplementedError("TIFF predictor {}".format(pred))
    out = np.empty((H, W), dtype=dt)

    def seg(off, cnt, rows, cols):
        a = np.frombuffer(_decode(bytes(buf[off : off + cnt]), comp), dtype=dt, count=rows * cols).reshape(rows, cols)
        if pred == 2:
            a = np.cumsum(a, axis=1, dtype=dt)
        return a

    if 322 in tags:  # tiles
        tw, tl = int(tags[322][0]), int(tags[323][0])
        offs, cnts = tags[324], tags[325]
        nx = -(-W // tw)
        for i, (o, c) in enumerate(zip(offs, cnts)):
            ty, tx = divmod(i, nx)
            t = seg(o, c, tl, tw)
            y0, x0 = ty * tl, tx * tw
            out[y0 : y0 + tl, x0 : x0 + tw] = t[: H - y0, : W - x0]
    else:
        rps = min(int(tags.get(278, (H,))[0]), H)
        offs, cnts = tags[273], tags.get(279)
        if cnts is None:  # allowed for a single uncompressed strip
            cnts = (H * W * dt.itemsize,)
        for i, (o, c) in enumerate(zip(offs, cnts)):
            y0 = i * rps
            rows = min(rps, H - y0)
            if o + c > len(buf):
                raise TiffError("strip {} points outside the file".format(i))
            out[y0 : y0 + rows] = seg(o, c, rows, W)
    return out.astype(dt.newbyteorder("="), copy=False)


def imread(path):
    """All pages of a TIFF file: ``[H, W]`` for one page, ``[pages, H, W]`` for several (like ``tifffile.imread``)."""
    with open(str(path), "rb") as f:
        buf = memoryview(f.read())
    if len(buf) < 8 or bytes(buf[:2]) not in (b"II", b"MM"):
        raise TiffError("{}: not a TIFF file".format(path))
    bo = "<" if bytes(buf[:2]) == b"II" else ">"
    (magic,) = struct.unpack_from(bo + "H", buf, 2)
    if magic == 42:
        big = False
        (off,) = struct.unpack_from(bo + "I", buf, 4)
    elif magic == 43:
        big = True
        (off,) = struct.unpack_from(bo + "Q", buf, 8)
    else:
        raise TiffError("{}: bad TIFF magic {}".format(path, magic))
    pages, seen = [], set()
    while off and off not in seen:
        seen.add(off)
        tags, off = _read_ifd(buf, off, bo, big)
        if 256 not in tags or 257 not in tags:
            continue
        pages.append(_read_page(buf, tags, bo))
    if not pages:
        raise TiffError("{}: no image pages".format(path))
    if len(pages) == 1:
        return pages[0]
    if len({(p.shape, p.dtype) for p in pages}) != 1:
        raise NotImplementedError("pages of different shape / dtype")
    return np.stack(pages)


def imwrite(path, img, compression=None, software="aind_smartspim_destripe_amd"):
    """Write ``[H, W]`` (one page) or ``[pages, H, W]`` as little-endian baseline TIFF, one strip per page.

    ``compression``: ``None`` = uncompressed (what the reference's ``imsave`` ends up writing), an int = Deflate
    level.  Files beyond 4 GiB would need BigTIFF and are rejected.
    """
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[None]
    if a.ndim != 3:
        raise ValueError("imwrite takes [H, W] or [pages, H, W] arrays")
    if a.dtype.kind not in "uif" or a.dtype.itemsize not in (1, 2, 4, 8) or (a.dtype.kind == "f" and a.dtype.itemsize < 4):
        raise ValueError("unsupported dtype {}".format(a.dtype))
    a = np.ascontiguousarray(a, dtype=a.dtype.newbyteorder("<"))
    fmt = {"u": 1, "i": 2, "f": 3}[a.dtype.kind]
    sw = software.encode("ascii") + b"\0"
    chunks, pos = [b"II*\0" + b"\0\0\0\0"], 8
    ifd_offsets = []
    P, H, W = a.shape
    for p in range(P):
        data = a[p].tobytes()
        if compression is not None:
            data = zlib.compress(data, int(compression))
        data_off = pos
        chunks.append(data)
        pos += len(data)
        if pos % 2:
            chunks.append(b"\0")
            pos += 1
        sw_off = pos
        chunks.append(sw)
        pos += len(sw)
        if pos % 2:
            chunks.append(b"\0")
            pos += 1
        entries = [
            (256, 4, 1, W), (257, 4, 1, H), (258, 3, 1, a.dtype.itemsize * 8),
            (259, 3, 1, 1 if compression is None else 8), (262, 3, 1, 1), (273, 4, 1, data_off), (277, 3, 1, 1),
            (278, 4, 1, H), (279, 4, 1, len(data)), (305, 2, len(sw), sw_off), (339, 3, 1, fmt),
        ]  # fmt: skip
        ifd_offsets.append(pos)
        ifd = struct.pack("<H", len(entries))
        for tag, typ, cnt, val in entries:
            ifd += struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
        pos += len(ifd) + 4
        chunks.append(ifd)
        chunks.append(None)  # next-IFD pointer, patched below
        if pos >= 2**32:
            raise ValueError("image too large for classic TIFF")
    k = 0
    for i, c in enumerate(chunks):
        if c is None:
            k += 1
            chunks[i] = struct.pack("<I", ifd_offsets[k] if k < P else 0)
    chunks[0] = b"II*\0" + struct.pack("<I", ifd_offsets[0])
    with open(str(path), "wb") as f:
        f.write(b"".join(chunks))
