#!/bin/bash
# round-3 evidence, one session: GPU tests, driver-style bench, kernel-trace stats, HBM traffic (PMC, stamped with the
# build hash), PMC per kernel, single-stream chain trace, bench lines of the other BASELINE shapes, the 2-rank
# self-launch rehearsal, chunk-map store-to-store runs (verified).   usage: bash tools/r3_profile.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r3}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/${T}_gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/${T}_gpu_tests.log
python tools/fuzz_parity.py 600 424242 > gpurun_out/${T}_fuzz_parity.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/${T}_fuzz_parity.log
python tools/fuzz_parity.py 300 98765 wavelets > gpurun_out/${T}_fuzz_wavelets.log 2>&1; echo "wavelet fuzz rc=$?"; tail -1 gpurun_out/${T}_fuzz_wavelets.log
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_style.json 2> gpurun_out/${T}_bench_driver_style.err; echo "driver-style bench rc=$?"
python bench.py --steps 200 --warmup 20 --kernel-breakdown > gpurun_out/${T}_bench_2048.json 2> gpurun_out/${T}_bench_2048.err
rm -rf gpurun_out/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -- python3 bench.py --steps 20 --warmup 5 --settle 0 --no-verify --cpu-planes 0 > gpurun_out/${T}_bench_traced.json 2> /dev/null
cp $(ls gpurun_out/kstats/*/*kernel_stats.csv | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/kstats
bash tools/traffic.sh $T > /dev/null 2>&1
bash tools/pmc.sh $T > /dev/null 2>&1; mv gpurun_out/pmc_$T.txt gpurun_out/${T}_pmc_per_kernel.txt
bash tools/chain_trace.sh > gpurun_out/${T}_chain_trace.txt 2>&1
python bench.py --steps 200 --warmup 20 --shading --cpu-planes 64 > gpurun_out/${T}_bench_2048_shading.json 2> /dev/null
python bench.py --steps 200 --warmup 20 --shape 1800x1800 --cpu-planes 128 > gpurun_out/${T}_bench_1800.json 2> /dev/null
python bench.py --steps 200 --warmup 20 --shape 1600x2000 --cpu-planes 128 > gpurun_out/${T}_bench_1600x2000.json 2> /dev/null
python bench.py --steps 200 --warmup 20 --shape 1600x2000 --shading --cpu-planes 64 > gpurun_out/${T}_bench_1600x2000_shading.json 2> /dev/null
DSX_SHARE_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 5 --batch 64 > gpurun_out/${T}_rehearsal_self_launch_2ranks_shared_gpu.json 2> gpurun_out/${T}_rehearsal.err; echo "self-launch rehearsal rc=$?"
python tools/bench_zarr.py 4096 > gpurun_out/${T}_bench_zarr_4096.json 2> gpurun_out/${T}_bz.err; echo "zarr raw rc=$?"
python tools/bench_zarr.py 4096 blosc > gpurun_out/${T}_bench_zarr_4096_blosc.json 2> gpurun_out/${T}_bzb.err; echo "zarr blosc rc=$?"
for f in gpurun_out/${T}_bench_*.json gpurun_out/${T}_rehearsal_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d.get('roofline',{}).get('frac'), d.get('verified'), d.get('cpu_baseline',{}).get('value'), d.get('n_gpus'))"; done
head -14 gpurun_out/${T}_kernel_stats.csv
grep "mb_per_plane" gpurun_out/${T}_traffic.json
