#!/usr/bin/env python3
"""End-to-end chunk map throughput: an uncompressed uint16 Zarr-v2 store of N planes
2048 x 2048, chunks (1,1,64,128,128), through destripe_zarr (device re-tiling, overlapped upload / filter /
download) into another store.  Prints one JSON line; the roofline of this path is the host link
(PCIe Gen5 x16, 63 GB/s spec: 16.8 MB per plane both ways -> <= 3.7 k planes/s), not HBM."""
import json, logging, os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aind_smartspim_destripe_amd import synth, zarr_destriper as zd
from aind_smartspim_destripe_amd.mini_zarr import MiniZarrArray

logging.basicConfig(level=logging.INFO, stream=sys.stderr)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
codec = sys.argv[2] if len(sys.argv) > 2 else None  # None (raw chunks), "zlib" or "blosc" (Blosc-zstd, the production codec)
H = W = 2048
root = tempfile.mkdtemp(prefix="dsx_zarr_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    t0 = time.perf_counter()
    src = MiniZarrArray.create(os.path.join(root, "in.zarr"), (1, 1, n, H, W), (1, 1, 64, 128, 128), np.uint16, compressor=codec)
    bank = synth.synthetic_bank(8, H, W)
    for z in range(0, n, 64):
        src[0, 0, z : z + 64] = synth.synthetic_stack(min(64, n - z), H, W, bank=bank)
    t_make = time.perf_counter() - t0
    res = {}
    for name, kw in (("overlapped", {}),):
        for rep in range(2):  # second pass: plan + pinned buffers exist, page cache warm
            t0 = time.perf_counter()
            planes, dt = zd.destripe_zarr_store(os.path.join(root, "in.zarr"), os.path.join(root, "out.zarr"), synth.CELLS_CONFIG,
                                          synth.NO_CELLS_CONFIG, None, prediction_chunksize=(64, H, W),
                                          output_chunks=(1, 1, 64, 128, 128), device=0, device_retile=True, io_threads=16, compressor=codec, **kw)
            res[name] = {"planes": planes, "seconds": round(time.perf_counter() - t0, 3)}
    out = MiniZarrArray.open(os.path.join(root, "out.zarr"))
    chk = int(out[0, 0, 0].astype(np.uint64).sum())
    v = res["overlapped"]["planes"] / res["overlapped"]["seconds"]
    # ---- verification (tests/test_zarr_chunk_map.py::test_chunk_map_at_production_geometry_against_the_oracle holds the
    # same statement on a 192-plane store): one plane of every stream part of the first, a middle and the last block --
    # bit-identical to the same plane filtered alone (one launch chain, one stream); the first block's picks and planes
    # 0 / 1 also against the CPU oracle (uint16 truncation: one count; see parity_util.u16_plane_against_oracle)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from parity_util import stream_part_picks, u16_plane_against_oracle
    from aind_smartspim_destripe_amd import filtering as fl
    nblk = (n + 63) // 64
    blocks = sorted({0, nblk // 2, nblk - 1})
    verified, checked, oracle_checked = True, [], []
    for b in blocks:
        z0, z1 = 64 * b, min(64 * b + 64, n)
        for z in stream_part_picks(64, z0, z1):
            plane_in = src[0, 0, z]
            got = out[0, 0, z]
            alone = fl.destripe_planes(plane_in[None], "t", synth.NO_CELLS_CONFIG, synth.CELLS_CONFIG, None, 2500,
                                       out_dtype=np.uint16, max_batch=1)[0]
            if not np.array_equal(alone, got):
                verified = False
            checked.append(int(z))
            if b == 0:
                try:
                    u16_plane_against_oracle(got, plane_in, "t", None, ("bench_zarr", z))
                    oracle_checked.append(int(z))
                except AssertionError as e:
                    verified = False
                    print("oracle mismatch", e, file=sys.stderr)
    print(json.dumps({"metric": "2048x2048 uint16 slices/s, Zarr store to Zarr store ({} chunks, tmpfs)".format(codec or "raw"), "value": round(v, 1),
                      "planes": n, "seconds": res["overlapped"]["seconds"], "store_make_s": round(t_make, 1),
                      "roofline": {"bound": "host link", "peak_planes_per_s": 3750, "frac": round(v / 3750.0, 3)},
                      "plane0_checksum": chk, "verified": verified,
                      "verification": {"planes_bit_identical_to_single_plane_runs": checked,
                                       "planes_against_the_cpu_oracle": oracle_checked, "blocks": blocks}}))
    if not verified:
        sys.exit(3)
finally:
    shutil.rmtree(root, ignore_errors=True)
