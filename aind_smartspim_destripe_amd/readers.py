"""Image reading interface of the directory mode (``/root/reference/code/aind_smartspim_destripe/readers.py``).

Same names and behaviour as the reference module: ``SUPPORTED_READING_EXTENSIONS``, ``_get_extension``
(``:16-31``), ``raw_imread`` (``:34-63``) and ``imread`` (``:66-92``).  ``tifffile`` and ``imageio`` are not
installed here: TIFF goes through :mod:`mini_tiff`; PNG (``imageio``, ``readers.py:87-88``) is not available
offline and raises ``NotImplementedError`` (SmartSPIM acquisitions are TIFF).
"""

from pathlib import Path
from typing import Union

import numpy as np

from . import mini_tiff

PathLike = Union[Path, str]

SUPPORTED_READING_EXTENSIONS = [".tif", ".tiff", ".raw", ".png"]


def _get_extension(path):
    """File extension of ``path`` including the dot (``""`` if none)."""
    return Path(path).suffix


def raw_imread(path):
    """Memory-map a ``.raw`` plane: two uint32 (width, height) then uint16 pixels.

    The byte order is detected as the reference does (``readers.py:40-56``): read the header both ways and
    take the order that gives the smaller width.
    """
    as_uint32 = np.memmap(path, dtype=">u4", mode="r", shape=(2,))
    width_be, height_be = as_uint32[:2]
    del as_uint32
    as_uint32 = np.memmap(path, dtype="<u4", mode="r", shape=(2,))
    width_le, height_le = as_uint32[:2]
    del as_uint32
    if width_le < width_be:
        width, height, dtype = width_le, height_le, "<u2"
    else:
        width, height, dtype = width_be, height_be, ">u2"
    try:
        return np.memmap(path, dtype=dtype, mode="r", offset=8, shape=(int(width), int(height)))
    except Exception as e:
        print("Bad path: %s" % path)
        raise e


def imread(path: PathLike) -> np.array:
    """Load a TIFF or RAW plane; ``None`` for an unknown extension, like the reference."""
    path = str(path)
    img = None
    extension = _get_extension(path)
    if extension == ".raw":
        img = raw_imread(path)
    elif extension == ".tif" or extension == ".tiff":
        img = mini_tiff.imread(path)
    elif extension == ".png":
        raise NotImplementedError("PNG needs imageio, which is not available in this environment")
    return img
