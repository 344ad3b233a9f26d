"""Directory mode (TIFF / RAW planes) behind the batched GPU call.

Mirrors ``/root/reference/code/aind_smartspim_destripe/destriper.py``: ``imsave`` (``:49-110``),
``read_filter_save`` (``:113-215``), ``_find_all_images`` (``:230-264``) and ``batch_filter`` (``:267-378``) keep
their names, arguments and error behaviour.  What changes is the execution model: the reference maps
``read_filter_save`` over a ``multiprocessing.Pool`` (one plane per task, ``:366-373``); here ``batch_filter``
reads planes with a thread pool, groups them by shape / dtype and pushes every group through
``filtering.destripe_planes`` (one launch chain per cohort), then writes the results with the same pool.
Every plane still sees exactly ``filter_stripes`` semantics with the default ``microscope_high_int=2700``
(``destriper.py:194-200``).  No CPU fallback: without the HIP library the filter call raises.
"""

import logging
import os
import shutil
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Optional

import numpy as np

from . import filtering as fl
from . import mini_png, mini_tiff
from .readers import SUPPORTED_READING_EXTENSIONS, PathLike, imread

logger = logging.getLogger(__name__)

SUPPORTED_OUTPUT_EXTENSIONS = [".tif", ".tiff", ".png"]


def _get_extension(path):
    return Path(path).suffix


_TIFF_SOURCES = {".raw", ".png", ".tif", ".tiff"}  # inputs whose result is written as <stem>.tiff


def _write_plane(filename, img, compression):
    """One writer per output extension.  TIFF: the reference hands ``compressionargs={"level": ...}`` to
    ``tifffile.imsave`` without naming a codec, which writes uncompressed strips (``destriper.py:82-101``);
    ``compression`` is accepted and has no effect here either."""
    ext = Path(filename).suffix
    if ext in (".tif", ".tiff"):
        mini_tiff.imwrite(filename, img)
    elif ext == ".png":  # iio.v3.imwrite(filename, img, compress_level=compression), destriper.py:107-110
        mini_png.imwrite(filename, img, compress_level=compression)
    else:  # pragma: no cover - callers only pass supported extensions
        raise NotImplementedError(ext)


def imsave(path, img, compression=1, output_format: Optional[str] = None):
    """Save a plane next to ``path`` (reference ``destriper.py:49-110``).

    Without ``output_format`` every readable source format is saved as ``<stem>.tiff`` (anything else:
    ``NotImplementedError``); with it, as ``<stem><output_format>`` for the supported output extensions
    (anything else: ``ValueError``).
    """
    stem = os.path.splitext(path)[0]
    if output_format is not None:
        if output_format not in SUPPORTED_OUTPUT_EXTENSIONS:
            raise ValueError(
                f"Output format {output_format} is not valid! Supported extensions are: {SUPPORTED_OUTPUT_EXTENSIONS}"
            )
        return _write_plane(stem + output_format, img, compression)
    extension = _get_extension(path)
    if extension not in _TIFF_SOURCES:
        raise NotImplementedError(f"We can't save in {extension} format, available: {SUPPORTED_OUTPUT_EXTENSIONS}")
    return _write_plane(stem + ".tiff", img, compression)


def _log_unreadable(output_dir, input_path):
    """``destriper.py:178-189``: unreadable planes are listed in ``destripe_log.txt`` and skipped."""
    file_name = os.path.join(output_dir, "destripe_log.txt")
    if not os.path.exists(file_name):
        with open(file_name, "w") as f:
            f.write("Error reading the following images.  We will interpolate their content.")
    with open(file_name, "a+") as f:
        f.write("\n{}".format(str(input_path)))


def _read_with_retries(output_dir, input_path, n=3):
    for i in range(n):
        try:
            raw_image = imread(input_path)
            if raw_image is None:
                raise ValueError("unsupported extension")
            return np.asarray(raw_image)
        except Exception:
            if i == n - 1:
                _log_unreadable(output_dir, input_path)
                return None
            time.sleep(0.05)
    return None


def _save_with_retries(output_path, image, compression, output_format, nb_retry=10):
    for _ in range(nb_retry):
        try:
            imsave(output_path, image, compression=compression, output_format=output_format)
        except OSError:
            logger.error(f"Retrying writing image in {output_path}...")
            continue
        break


def _target_dtype(raw_dtype, output_dtype):
    if output_dtype is not None and isinstance(output_dtype, type):
        return output_dtype
    return raw_dtype


def read_filter_save(
    output_dir: PathLike,
    input_path: PathLike,
    output_path: PathLike,
    high_int_filter_params: dict,
    low_int_filter_params: dict,
    shadow_correction: dict = None,
    compression: Optional[int] = 1,
    output_format: Optional[str] = None,
    output_dtype: Optional[type] = None,
):
    """One plane: read (3 tries), ``filter_stripes`` on the GPU, ``astype`` to the source dtype (or
    ``output_dtype``), save (``destriper.py:113-215``)."""
    raw_image = _read_with_retries(output_dir, input_path)
    if raw_image is None:
        return
    dtype = _target_dtype(raw_image.dtype, output_dtype)
    filtered_image = fl.filter_stripes(
        image=raw_image,
        input_tile_path=input_path,
        no_cells_config=low_int_filter_params,
        cells_config=high_int_filter_params,
        shadow_correction=shadow_correction,
    )
    _save_with_retries(output_path, filtered_image.astype(dtype), compression, output_format)


def _read_filter_save(input_dict: dict):
    read_filter_save(**input_dict)


def _find_all_images(search_path: PathLike, input_path: PathLike, output_path: PathLike):
    """Readable images below ``search_path``, depth first in directory order; every sub-folder met on the way is
    re-created under ``output_path`` at the same position relative to ``input_path`` (reference
    ``destriper.py:230-264``)."""
    root_in, root_out, here = Path(input_path), Path(output_path), Path(search_path)
    assert here.is_dir()
    found = []
    for entry in here.iterdir():
        if entry.is_dir():
            root_out.joinpath(entry.relative_to(root_in)).mkdir(parents=True, exist_ok=True)
            found += _find_all_images(entry, root_in, root_out)
        elif entry.is_file() and entry.suffix in SUPPORTED_READING_EXTENSIONS:
            found.append(entry)
    return found


def _prepare_tree(input_path, output_path):
    """What ``batch_filter`` does before it filters (reference ``destriper.py:322-364``): forget the error log of
    an earlier run, collect the images (mirroring the folder tree), copy the acquisition's ``.txt`` / ``.ini``
    side files, and pair every image with its output path."""
    src_root, dst_root = Path(input_path), Path(output_path)
    stale_log = dst_root / "destripe_log.txt"
    if stale_log.exists():
        stale_log.unlink()
    logger.info(f"Looking for images in {src_root}")
    images = _find_all_images(src_root, src_root, dst_root)
    logger.info(f"Found {len(images)} compatible images")
    for side_file in src_root.iterdir():
        if side_file.suffix in (".txt", ".ini"):
            shutil.copyfile(side_file, dst_root / side_file.name)
    targets = [dst_root / img.relative_to(src_root) for img in images]
    for t in targets:
        t.parent.mkdir(parents=True, exist_ok=True)
    return images, targets


def batch_filter(
    input_path: PathLike,
    output_path: PathLike,
    workers: int,
    chunks: int,
    high_int_filt_params: dict,
    low_int_filt_params: dict,
    shadow_correction: dict,
    compression: Optional[int] = 1,
    output_format: Optional[str] = None,
    output_dtype: Optional[type] = None,
    device: int = 0,
):
    """Filter every image below ``input_path`` into the same tree under ``output_path``
    (``destriper.py:267-378``).  ``workers`` = I/O threads, ``chunks`` = planes per GPU batch
    (the reference's pool size and ``imap`` chunk size).  Returns the number of planes written.
    """
    input_path, output_path = Path(input_path), Path(output_path)
    error_path = os.path.join(output_path, "destripe_log.txt")
    img_paths, outs = _prepare_tree(input_path, output_path)
    batch = max(1, int(chunks))
    written = 0
    with ThreadPoolExecutor(max_workers=max(1, int(workers))) as pool:
        for start in range(0, len(img_paths), batch):
            paths = img_paths[start : start + batch]
            planes = list(pool.map(lambda p: _read_with_retries(output_path, p), paths))
            groups = {}
            for k, a in enumerate(planes):
                if a is None:
                    continue
                if a.ndim != 2:
                    _log_unreadable(output_path, paths[k])
                    continue
                # planes of one folder share the tile (hence the hemisphere flat, filtering.py:273-335)
                groups.setdefault((str(paths[k].parent), a.shape, a.dtype.str), []).append(k)
            jobs = []
            for ks in groups.values():
                stack = np.stack([np.ascontiguousarray(planes[k]).astype(planes[k].dtype.newbyteorder("="), copy=False)
                                  for k in ks])  # fmt: skip
                shaded = shadow_correction is not None
                # filter_stripes returns float64 (uint16 with shading) and read_filter_save casts it to the
                # source dtype: for integer sources that is the clip-free truncation the uint16 epilogue does
                want = _target_dtype(stack.dtype, output_dtype)
                as_u16 = shaded or np.dtype(want) == np.uint16
                res = fl.destripe_planes(
                    stack,
                    input_tile_path=str(paths[ks[0]]),
                    no_cells_config=low_int_filt_params,
                    cells_config=high_int_filt_params,
                    shadow_correction=shadow_correction,
                    out_dtype=np.uint16 if as_u16 else np.float32,
                    max_batch=min(batch, 64),
                    device=device,
                )
                for j, k in enumerate(ks):
                    jobs.append((outs[start + k], res[j].astype(want, copy=False)))
            list(pool.map(lambda jo: _save_with_retries(jo[0], jo[1], compression, output_format), jobs))
            written += len(jobs)
    logger.info("Done with batch filtering!")
    if os.path.exists(error_path):
        logger.error("An error happened, see destripe log for more details")
    return written
