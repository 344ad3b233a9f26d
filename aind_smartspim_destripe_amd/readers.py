"""Plane readers of the directory mode: same entry points as the reference's ``readers.py``.

``SUPPORTED_READING_EXTENSIONS``, ``_get_extension`` (reference ``readers.py:16-31``), ``raw_imread``
(``:34-63``) and ``imread`` (``:66-92``) keep their names, arguments and results.  ``tifffile`` and ``imageio`` are
not installed here: TIFF goes through :mod:`mini_tiff`, PNG (``imageio``, ``readers.py:86-87``) through :mod:`mini_png`.
"""

import os
import struct
from typing import Union

import numpy as np

from . import mini_png, mini_tiff

PathLike = Union[os.PathLike, str]

SUPPORTED_READING_EXTENSIONS = [".tif", ".tiff", ".raw", ".png"]

_RAW_HEADER_BYTES = 8  # two uint32: width, height


def _get_extension(path):
    """Extension of ``path`` with its dot, ``""`` when there is none (both ``/`` and ``\\`` separate folders)."""
    name = str(path).replace("\\", "/").rsplit("/", 1)[-1]
    dot = name.rfind(".")
    return name[dot:] if dot > 0 else ""


def _raw_geometry(header: bytes):
    """``(width, height, pixel dtype)`` of a ``.raw`` header.

    The format does not record its byte order; like the reference (``readers.py:40-56``) both readings are
    tried and the one with the smaller width wins (right for every width below 65536).
    """
    if len(header) < _RAW_HEADER_BYTES:
        raise ValueError("raw plane shorter than its header")
    big = struct.unpack(">II", header[:_RAW_HEADER_BYTES])
    little = struct.unpack("<II", header[:_RAW_HEADER_BYTES])
    if little[0] < big[0]:
        return little[0], little[1], np.dtype("<u2")
    return big[0], big[1], np.dtype(">u2")


def raw_imread(path):
    """Read-only memory map of a ``.raw`` plane, shaped ``(width, height)`` as in the reference."""
    try:
        with open(path, "rb") as f:
            width, height, dtype = _raw_geometry(f.read(_RAW_HEADER_BYTES))
        return np.memmap(path, dtype=dtype, mode="r", offset=_RAW_HEADER_BYTES, shape=(width, height))
    except Exception:
        print("Bad path: %s" % path)
        raise


def imread(path: PathLike) -> np.array:
    """Plane of a ``.tif`` / ``.tiff`` / ``.raw`` / ``.png`` file; ``None`` for any other extension, like the reference."""
    path = os.fspath(path)
    loaders = {".raw": raw_imread, ".tif": mini_tiff.imread, ".tiff": mini_tiff.imread, ".png": mini_png.imread}
    loader = loaders.get(_get_extension(path))
    return None if loader is None else loader(path)
