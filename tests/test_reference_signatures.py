"""The drop-in boundary by NAME: every public function the package shares with the reference takes the reference's
parameters, in the reference's order, and the two chunk-map entry points accept exactly the keyword sets of the
reference's own call sites (``run_capsule.py:394-403`` -> ``destripe_channel``, ``zarr_destriper.py:1252-1267`` ->
``destripe_zarr``).  ``tests/golden/reference_signatures.json`` is written by ``oracle/make_reference_signatures.py``
(``ast`` over the reference's source text; names only).  CPU tests: nothing here touches a GPU."""

import ast
import inspect
import json
import os

import numpy as np
import pytest

from aind_smartspim_destripe_amd import mini_tiff, synth
from aind_smartspim_destripe_amd import zarr_destriper as zd
from aind_smartspim_destripe_amd.mini_zarr import MiniZarrArray

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "aind_smartspim_destripe_amd")
with open(os.path.join(HERE, "golden", "reference_signatures.json")) as _f:
    REF = json.load(_f)


def _module_functions(path):
    tree = ast.parse(open(path).read())
    return {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}


@pytest.mark.parametrize("module", sorted(REF["modules"]))
def test_same_named_functions_are_prefix_compatible(module):
    """A caller written against the reference must be able to call the same-named function here unchanged:
    the reference's positional parameters come first, under the same names and in the same order; parameters the
    reference gives a default keep one; whatever the engine adds behind them is optional (a default, or keyword-only
    with a default)."""
    ours = _module_functions(os.path.join(PKG, module + ".py"))
    shared = sorted(set(ours) & set(REF["modules"][module]))
    assert shared, module
    for name in shared:
        ref = REF["modules"][module][name]
        a = ours[name].args
        pos = [x.arg for x in a.posonlyargs + a.args]
        n_ref = len(ref["positional"])
        assert pos[:n_ref] == ref["positional"], (module, name, pos, ref["positional"])
        first_default = len(pos) - len(a.defaults)  # index of the first positional parameter with a default
        ref_first_default = n_ref - ref["n_defaults"]
        assert first_default <= ref_first_default, (module, name, "a parameter the reference defaults is required here")
        assert first_default <= n_ref, (module, name, "an added positional parameter has no default")
        for kw, default in zip(a.kwonlyargs, a.kw_defaults):
            assert default is not None, (module, name, kw.arg, "keyword-only extras must be optional")


def test_the_reference_exports_we_mirror_are_all_there():
    expected = {
        "zarr_destriper": ["read_json_as_dict", "get_microscope_flats", "pad_array_n_d", "execute_worker", "compute_pyramid",
                           "compute_multiscale", "destripe_zarr", "destripe_channel"],  # fmt: skip
        "filtering": ["sigmoid", "foreground_fraction", "get_foreground_background_mean", "notch", "gaussian_filter",
                      "log_space_fft_filtering", "normalize_image", "invert_image", "get_hemisphere_flatfield",
                      "flatfield_correction", "filter_stripes"],  # fmt: skip
        "destriper": ["imsave", "read_filter_save", "_find_all_images", "batch_filter"],
        "readers": ["raw_imread", "imread"],
    }
    for module, names in expected.items():
        ours = _module_functions(os.path.join(PKG, module + ".py"))
        for n in names:
            assert n in ours and n in REF["modules"][module], (module, n)


# ---- the reference's own call sites, keyword for keyword -------------------------------------------------
def _channel_fixture(tmp_path, H=32, W=48, Z=8):
    chan = tmp_path / "data" / "Ex_488_Em_525"
    names = ["431040_368180", "431040_394100"]
    for t, name in enumerate(names):
        a = MiniZarrArray.create(str(chan / (name + ".zarr") / "0"), (1, 1, Z, H, W), (1, 1, 4, 16, 16), np.uint16,
                                 compressor="zlib")  # fmt: skip
        a[0, 0] = synth.synthetic_stack(Z, H, W, n_unique=4) + np.uint16(t)
    d = tmp_path / "derivatives"
    d.mkdir()
    mini_tiff.imwrite(str(d / "DarkMaster_cropped.tif"), np.full((H + 8, W + 8), 90, np.uint16))
    flats = []
    for side in (0, 1):
        f = np.full((H, W), 1.0 + 0.25 * side, np.float32)
        mini_tiff.imwrite(str(d / "flat_{}.tif".format(side)), f)
        flats.append(f)
    return chan, names, d, flats


def test_destripe_channel_takes_the_keyword_set_of_run_capsule(tmp_path, monkeypatch):
    """``run_capsule.py:394-403`` calls ``destripe_channel`` with eight keywords (``xyz_resolution`` among them); the
    call must go through, and every tile must reach ``destripe_zarr`` with exactly the keyword set of the reference's
    own call (``zarr_destriper.py:1252-1267``) and the reference's values."""
    site = REF["call_sites"]["destripe_channel"][0]
    assert site["n_positional"] == 0 and "xyz_resolution" in site["keywords"]
    chan, names, d, flats = _channel_fixture(tmp_path)
    params = {"cells_config": synth.CELLS_CONFIG, "no_cells_config": synth.NO_CELLS_CONFIG, "retrospective": True}
    values = {
        "zarr_dataset_path": tmp_path / "data",
        "channel_name": "Ex_488_Em_525",
        "results_folder": tmp_path / "results",
        "derivatives_path": d,
        "xyz_resolution": [1.8, 1.8, 2.0],
        "estimated_channel_flats": [d / "flat_0.tif", d / "flat_1.tif"],
        "laser_tiles": {"0": [names[0]], "1": [names[1]]},
        "parameters": params,
    }
    assert sorted(values) == sorted(site["keywords"])
    calls = []

    def fake_destripe_zarr(**kw):
        calls.append(kw)
        return 8, 0.0

    monkeypatch.setattr(zd, "destripe_zarr", fake_destripe_zarr)
    done = zd.destripe_channel(**{k: values[k] for k in site["keywords"]})
    assert done == {names[0] + ".zarr": 8, names[1] + ".zarr": 8}
    ref_kw = REF["call_sites"]["destripe_zarr"][0]["keywords"]
    for kw, name, side in zip(calls, names, (0, 1)):
        assert [k for k in ref_kw if k in kw] == ref_kw  # every keyword of the reference's call is passed
        assert kw["dataset_path"] == chan / (name + ".zarr") and kw["multiscale"] == "0"
        assert kw["output_destriped_zarr"] == tmp_path / "results" / "destriped_data" / "Ex_488_Em_525" / (name + ".zarr")
        assert tuple(kw["prediction_chunksize"]) == (64, 1600, 2000) and kw["target_size_mb"] == 3072
        assert kw["n_workers"] == 0 and kw["batch_size"] == 1 and tuple(kw["super_chunksize"]) == (384, 1600, 2000)
        assert kw["results_folder"] == tmp_path / "results" and kw["derivatives_path"] == d
        assert kw["xyz_resolution"] == [1.8, 1.8, 2.0] and kw["parameters"] is params and kw["lazy_callback_fn"] is None
        np.testing.assert_array_equal(kw["flatfield"], flats[side])
    # a tile of neither laser side: the reference's ValueError (zarr_destriper.py:1246-1247)
    values["laser_tiles"] = {"0": [names[0]]}
    with pytest.raises(ValueError, match="not found in"):
        zd.destripe_channel(**values)


def test_destripe_zarr_takes_the_keyword_set_of_the_reference_call(tmp_path, monkeypatch):
    """``zarr_destriper.py:1252-1267``: fourteen keywords.  The shading dictionary is built inside as ``:1095-1130``
    does (dark from ``DarkMaster_cropped.tif``, the given flat as the retrospective one), the level ``multiscale`` of the
    tile is read, array ``0`` of a group named like the tile is written, and the pyramid follows."""
    site = REF["call_sites"]["destripe_zarr"][0]
    sig = REF["modules"]["zarr_destriper"]["destripe_zarr"]
    assert site["keywords"] == sig["positional"]
    chan, names, d, flats = _channel_fixture(tmp_path)
    out = tmp_path / "results" / "destriped_data" / "Ex_488_Em_525" / (names[0] + ".zarr")
    seen = {}

    def fake_store(src, dst, cells, no_cells, **kw):
        seen.update(src=src, dst=dst, cells=cells, no_cells=no_cells, **kw)
        return 8, 0.5

    def fake_multiscale(**kw):
        seen["multiscale"] = kw
        return []

    monkeypatch.setattr(zd, "destripe_zarr_store", fake_store)
    monkeypatch.setattr(zd, "compute_multiscale", fake_multiscale)
    values = {
        "dataset_path": chan / (names[0] + ".zarr"),
        "multiscale": "0",
        "output_destriped_zarr": out,
        "prediction_chunksize": (64, 1600, 2000),
        "target_size_mb": 3072,
        "n_workers": 0,
        "batch_size": 1,
        "super_chunksize": (384, 1600, 2000),
        "results_folder": tmp_path / "results",
        "derivatives_path": d,
        "xyz_resolution": [1.8, 1.8, 2.0],
        "parameters": {"cells_config": synth.CELLS_CONFIG, "no_cells_config": synth.NO_CELLS_CONFIG},
        "flatfield": flats[0],
        "lazy_callback_fn": None,
    }
    n, _ = zd.destripe_zarr(**{k: values[k] for k in site["keywords"]})
    assert n == 8
    assert seen["src"] == str(chan / (names[0] + ".zarr") / "0") and seen["dst"] == str(out / "0")
    assert seen["cells"] == synth.CELLS_CONFIG and seen["no_cells"] == synth.NO_CELLS_CONFIG
    sc = seen["shadow_correction"]
    assert sc["retrospective"] is True and sc["tile_config"] is None
    np.testing.assert_array_equal(sc["flatfield"], flats[0])
    np.testing.assert_array_equal(sc["darkfield"], np.full((40, 56), 90, np.uint16))
    assert seen["tile_name"] == names[0] + ".zarr" and tuple(seen["prediction_chunksize"]) == (64, 1600, 2000)
    ms = seen["multiscale"]
    assert sorted(k for k in ms if k in REF["modules"]["zarr_destriper"]["compute_multiscale"]["positional"]) == sorted(
        REF["call_sites"]["compute_multiscale"][0]["keywords"])
    assert ms["voxel_size"] == [2.0, 1.8, 1.8] and ms["n_levels"] == 3 and ms["scale_factor"] == [2, 2, 2]
    assert ms["image_name"] == names[0] + ".zarr"
    # error behaviour of the reference: too many workers (:977-978), no dark in an existing derivatives folder (:1104-1108),
    # parameters without the configs (:972-973)
    monkeypatch.setenv("CO_CPUS", "4")
    with pytest.raises(ValueError, match="Provided workers 5 > current workers 4"):
        zd.destripe_zarr(**dict(values, n_workers=5))
    os.remove(d / "DarkMaster_cropped.tif")
    with pytest.raises(FileNotFoundError, match="provide the current dark"):
        zd.destripe_zarr(**values)
    with pytest.raises(KeyError):
        zd.destripe_zarr(**dict(values, parameters={}))
    with pytest.raises(NotImplementedError):
        zd.destripe_zarr(**dict(values, lazy_callback_fn=lambda a: a))
    # no derivatives folder and no flat: nothing to correct with -> the filter runs without shading
    seen.clear()
    zd.destripe_zarr(**dict(values, derivatives_path=tmp_path / "nowhere", flatfield=None))
    assert seen["shadow_correction"] is None


def test_compute_pyramid_and_multiscale_live_under_the_reference_module_name():
    assert inspect.signature(zd.compute_pyramid).parameters.keys() >= {"data", "n_lvls", "scale_axis", "chunks"}
    p = list(inspect.signature(zd.compute_multiscale).parameters)
    assert p[:8] == REF["modules"]["zarr_destriper"]["compute_multiscale"]["positional"]
    with pytest.raises(ValueError):  # argument checks run before any GPU is touched
        zd.compute_pyramid(np.zeros((4, 4, 4), np.uint16), 2, (2, 2, 1))


def test_io_thread_budget_is_shared_among_the_ranks_of_a_node(monkeypatch):
    """Eight ranks on one node must not start eight times every core's worth of codec threads."""
    cores = len(os.sched_getaffinity(0))
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert zd.default_io_threads(1) == max(2, min(cores, 64))
    assert zd.default_io_threads(8) == max(2, min(cores // 8, 64))
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")  # torchrun: ranks of THIS node (world_size may span nodes)
    assert zd.default_io_threads(16) == max(2, min(cores // 4, 64))
    total = 4 * zd.default_io_threads(16)
    assert total <= max(cores, 8)
