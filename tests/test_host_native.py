"""CPU-side checks of the native code: the C ABI library loads and exports every symbol the header
declares, fails loudly without a GPU, and the host/device-shared FFT + planning code is correct
(built with g++ from tests/host/dsx_host_check.cpp; no GPU needed)."""

import json
import os
import re
import subprocess

import numpy as np
import pytest

from aind_smartspim_destripe_amd import engine

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("host") / "dsx_host_check")
    src = os.path.join(REPO, "tests", "host", "dsx_host_check.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, src], check=True)

    def run(*args):
        out = subprocess.run([exe] + [str(a) for a in args], check=True, capture_output=True, text=True).stdout
        return json.loads(out)

    return run


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    g.build()
    return engine.load_library()


def test_header_and_library_symbols_agree(lib):
    header = open(os.path.join(REPO, "include", "dsx.h")).read()
    declared = set(re.findall(r"\b(dsx_[a-z0-9_]+)\s*\(", header))
    declared -= {"dsx_plan_info_t"}
    assert declared == set(engine.EXPORTED_SYMBOLS), declared ^ set(engine.EXPORTED_SYMBOLS)
    for name in engine.EXPORTED_SYMBOLS:
        assert getattr(lib, name) is not None


def test_engine_fails_loudly_without_gpu(lib):
    """No CPU fallback: without a visible GPU the engine constructor raises."""
    if lib.dsx_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(engine.DsxError):
        engine.DestripeEngine(0)
    from aind_smartspim_destripe_amd import filtering

    with pytest.raises(engine.DsxError):
        filtering.log_space_fft_filtering(np.ones((16, 16), np.float32), level=1)


def test_bank_swizzled_passes_of_the_power_of_two_plans(host_check):
    """``dsx_idx_swz`` (csrc/dsx_fft_core.h): the 2 048- and 1 024-point plans address the row buffer and the twiddle table
    with the low four index bits XORed with bits 4 ... 7 (their passes scatter at power-of-two strides: all lanes of a store
    into two LDS banks otherwise).  Data placed at sw(i), twiddles at sw(t), passes run through the policy, result read
    from sw(k): the DFT to float32 round-off, as without the swizzle."""
    r = host_check("swz")
    assert set(r) == {"m2048", "m1024"} and all(v < 4e-7 for v in r.values()), r


def test_pruned_forward_plan_of_level_1(host_check):
    """StaticFft<1>::run_forward (csrc/dsx_kernels.h): passes 6, 9 and a radix-19 last pass that only computes the output
    pairs inside the low-pass band (``dsx_bfly_store<19, 9, KO>``).  Every bin the spectral step reads -- |k| <= kcut, for
    the production kcut (103 / 206) and for the last bin 2 / 4 pairs reach (107 / 215) -- must be the DFT to float32
    round-off, exactly as good as the unpruned pass (KO = 9)."""
    r = host_check("pruned")
    assert set(r) == {"ko2_kcut103", "ko2_kcut107", "ko4_kcut206", "ko4_kcut215", "ko9_full"}
    assert all(v < 4e-7 for v in r.values()), r


@pytest.mark.parametrize("m", [1, 2, 4, 12, 20, 36, 42, 63, 68, 117, 126, 132, 144, 260, 515, 567, 960, 1026, 1071, 229, 1080, 1280, 1815, 2048, 2304,
                               3003, 4693, 9252])  # (the last three: lengths of k_rowfilter_wide, 9252 = 6 * 6 * 257 with a generic pass)
def test_fft_core_against_naive_dft(host_check, m):
    r = host_check("fft", m)
    assert r["rel_err"] < 2e-6, r


def test_plan_geometry_matches_pywt(host_check):
    """Level structure of SURVEY appendix B (verified there against pywt)."""
    expect = {
        (512, 512): [258, 131, 68, 36, 20, 12],
        (2048, 2048): [1026, 515, 260, 132, 68, 36, 20, 12],
        (1800, 1800): [902, 453, 229, 117, 61, 33, 19, 12],
    }
    for (h, w), widths in expect.items():
        p = host_check("plan", h, w, 128, -1, 64, -1)
        assert p["L"] == len(widths)
        assert [lv["w"] for lv in p["levels"]] == widths
        assert [lv["h"] for lv in p["levels"]] == widths
        for lv in p["levels"]:
            assert lv["ld"] % 4 == 0 and lv["ld"] >= lv["w"]
            assert int(np.prod(lv["radix"])) == lv["M"]
            assert lv["M"] == lv["w"] or lv["M"] >= 2 * lv["w"]
    p = host_check("plan", 1600, 2000, 128, -1, 64, -1)
    assert [(lv["h"], lv["w"]) for lv in p["levels"]] == [
        (802, 1002), (403, 503), (204, 254), (104, 129), (54, 67), (29, 36), (17, 20), (11, 12)]  # fmt: skip
    p = host_check("plan", 101, 103, 128, -1, 64, -1)
    assert (p["Hout"], p["Wout"]) == (102, 104)
    p = host_check("plan", 100, 100, 64, 1, 64, 1)
    assert p["L"] == 1 and p["levels"][0]["w"] == 52
    p = host_check("plan", 64, 64, 64, 0, 64, 0)
    assert p["L"] == 0 and (p["Hout"], p["Wout"]) == (64, 64)


def test_plan_of_planes_wider_than_one_wave_holds(host_check):
    """Rows of more than 64 * 36 coefficients go to the block-per-row-pair kernel: transform lengths up to 512 * 36, direct
    or embedded (periodic halo of half a row); beyond that the plan refuses the plane (DSX_ELIMIT)."""
    p = host_check("plan", 40, 4604, 128, -1, 64, -1)
    assert (p["levels"][0]["w"], p["levels"][0]["M"]) == (2304, 2304)
    for (h, w), n in (((40, 4608), 2306), ((33, 6001), 3003), ((24, 9216), 4610), ((16, 12288), 6146), ((16, 18500), 9252),
                      ((16, 36856), 18430)):
        lv = host_check("plan", h, w, 128, -1, 64, -1)["levels"][0]
        assert lv["w"] == n and 2304 < lv["M"] <= 512 * 36, (w, lv["M"])
        assert int(np.prod(lv["radix"])) == lv["M"]
        assert (lv["M"] == n and lv["K"] == 0) or (lv["K"] == n // 2 and lv["M"] >= 2 * n + 1), lv
    with pytest.raises(subprocess.CalledProcessError) as ei:
        host_check("plan", 16, 36900, 128, -1, 64, -1)
    assert "too wide" in ei.value.stdout


@pytest.mark.parametrize("args", [
    (2048, 2048, 64, -1, 0), (2048, 2048, 128, -1, 1), (2048, 2048, 64, -1, 4), (2048, 2048, 64, -1, 7),
    (1800, 1800, 64, -1, 1), (1800, 1800, 128, -1, 2), (1600, 2000, 128, -1, 1), (512, 512, 64, -1, 0),
    (100, 100, 64, 1, 0), (16, 4608, 128, -1, 0), (16, 6001, 64, -1, 0),
])  # fmt: skip
def test_row_filter_spectral_pipeline(host_check, args):
    """Two rows per complex FFT + G1/G2 tables (direct and exact-halo modes) reproduce the
    length-w circular operator with the fftpack packed-index gains."""
    r = host_check("rows", *args)
    assert r["rel_err"] < 3e-6, r


def test_bench_gpus_n_starts_n_ranks_by_itself_and_fails_loudly_without_gpus(lib):
    """`python bench.py --gpus 2` as the driver runs it (no torchrun, no WORLD_SIZE): the parent must start two ranks
    itself.  On this GPU-less machine every rank dies in dsx_init; the launcher must report it and exit non-zero
    without printing a result line."""
    if lib.dsx_device_count() > 0:
        pytest.skip("a GPU is visible: covered by the -m gpu rehearsal")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(
        [os.sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2",
         "--cpu-planes", "0", "--settle", "0", "--shape", "64x64"],
        capture_output=True, text=True, timeout=300, env=env)  # fmt: skip
    assert r.returncode != 0
    assert "started 2 ranks" in r.stderr and "no HIP device" in r.stderr
    assert r.stdout.strip() == ""
