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
from . import mini_tiff
from .readers import SUPPORTED_READING_EXTENSIONS, PathLike, imread

logger = logging.getLogger(__name__)

SUPPORTED_OUTPUT_EXTENSIONS = [".tif", ".tiff", ".png"]


def _get_extension(path):
    return Path(path).suffix


def imsave(path, img, compression=1, output_format: Optional[str] = None):
    """Save a plane (``destriper.py:49-110``): any readable input format is written as ``<stem>.tiff`` unless
    ``output_format`` names another supported extension.

    The reference passes ``compressionargs={"level": compression}`` to ``tifffile.imsave`` without a
    ``compression=`` codec, which writes uncompressed strips; so does this function.
    """
    extension = _get_extension(path)
    if output_format is None:
        if extension in (".raw", ".png", ".tif", ".tiff"):
            mini_tiff.imwrite(os.path.splitext(path)[0] + ".tiff", img)
        else:
            raise NotImplementedError(
                f"We can't save in {extension} format, available: {SUPPORTED_OUTPUT_EXTENSIONS}"
            )
    else:
        if output_format not in SUPPORTED_OUTPUT_EXTENSIONS:
            raise ValueError(
                f"Output format {output_format} is not valid! Supported extensions are: {SUPPORTED_OUTPUT_EXTENSIONS}"
            )
        filename = os.path.splitext(path)[0] + output_format
        if output_format == ".tif" or output_format == ".tiff":
            mini_tiff.imwrite(filename, img)
        elif output_format == ".png":
            raise NotImplementedError("PNG needs imageio, which is not available in this environment")


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
    """All readable images below ``search_path``; mirrors the folder tree under ``output_path``
    (``destriper.py:230-264``)."""
    input_path, output_path, search_path = Path(input_path), Path(output_path), Path(search_path)
    assert search_path.is_dir()
    img_paths = []
    for p in search_path.iterdir():
        if p.is_file():
            if p.suffix in SUPPORTED_READING_EXTENSIONS:
                img_paths.append(p)
        elif p.is_dir():
            o = output_path.joinpath(p.relative_to(input_path))
            if not o.exists():
                o.mkdir(parents=True)
            img_paths.extend(_find_all_images(p, input_path, output_path))
    return img_paths


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
    if os.path.exists(error_path):
        os.remove(error_path)
    logger.info(f"Looking for images in {input_path}")
    img_paths = _find_all_images(input_path, input_path, output_path)
    logger.info(f"Found {len(img_paths)} compatible images")
    for file in input_path.iterdir():  # copy text and ini files (:340-343)
        if Path(file).suffix in [".txt", ".ini"]:
            shutil.copyfile(file, os.path.join(output_path, os.path.split(file)[1]))
    outs = []
    for p in img_paths:
        o = output_path.joinpath(p.relative_to(input_path))
        if not o.parent.exists():
            o.parent.mkdir(parents=True)
        outs.append(o)
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
