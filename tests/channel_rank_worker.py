"""One rank of a multi-rank ``destripe_channel`` run (started by tests/test_tiff_modes.py, one process per rank;
RANK / WORLD_SIZE / DSX_RDZV_DIR in the environment, all ranks on GPU 0 of the one-GPU box).  Prints one JSON line."""

import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from aind_smartspim_destripe_amd import distributed, engine, mini_tiff, synth  # noqa: E402
from aind_smartspim_destripe_amd import zarr_destriper as zd  # noqa: E402


def main():
    root, results = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    reads = []
    real = mini_tiff.imread

    def counting(path, *a, **k):
        reads.append(os.path.basename(str(path)))
        return real(path, *a, **k)

    mini_tiff.imread = counting
    eng = engine.DestripeEngine(0)
    group = distributed.RankGroup.from_env(eng)
    d = os.path.join(root, "derivatives")
    done = zd.destripe_channel(
        zarr_dataset_path=os.path.join(root, "data"), channel_name="Ex_488_Em_525", results_folder=results,
        derivatives_path=d, xyz_resolution=[1.8, 1.8, 2.0],
        estimated_channel_flats=[os.path.join(d, "flat_0.tif"), os.path.join(d, "flat_1.tif")],
        laser_tiles={"0": ["431040_368180"], "1": ["431040_394100"]},
        parameters={"cells_config": synth.CELLS_CONFIG, "no_cells_config": synth.NO_CELLS_CONFIG},
        prediction_chunksize=(4, 64, 96), output_chunks=(1, 1, 4, 32, 32), compressor="zlib",
        rank=rank, world_size=world, device=0, group=group if world > 1 or group.active else None)  # fmt: skip
    out = {"rank": rank, "done": done, "transport": group.transport, "reads": sorted(set(reads)),
           "bytes_broadcast": group.bytes_broadcast, "io_threads": zd.LAST_RUN.get("io_threads")}  # fmt: skip
    group.close()
    eng.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
