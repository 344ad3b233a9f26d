#!/bin/bash
# round-4 evidence on the final build, two sessions (gpurun's limit is 20 minutes per call):
#   bash tools/r4_profile.sh a   GPU tests, fuzz runs (plain, wavelets, launch geometry), driver-style bench, 200-step bench with
#                                per-kernel events, kernel-trace stats, HBM traffic (PMC, stamped with the build hash), PMC per
#                                kernel, single-stream chain trace
#   bash tools/r4_profile.sh b   bench lines of the other BASELINE shapes / shading, the 2-rank self-launch rehearsal, chunk-map
#                                store-to-store runs (verified), small-cohort latency
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=r4
mkdir -p gpurun_out
if [ "${1:-a}" = "a" ]; then
timeout -k 10 600 python -m pytest tests -m gpu -q -s > gpurun_out/${T}_gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/${T}_gpu_tests.log
timeout -k 10 300 python tools/fuzz_parity.py 400 424243 > gpurun_out/${T}_fuzz_parity.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/${T}_fuzz_parity.log | cut -c1-300
timeout -k 10 300 python tools/fuzz_parity.py 200 98766 wavelets > gpurun_out/${T}_fuzz_wavelets.log 2>&1; echo "wavelet fuzz rc=$?"; tail -1 gpurun_out/${T}_fuzz_wavelets.log | cut -c1-300
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_style.json 2> gpurun_out/${T}_bench_driver_style.err; echo "driver-style bench rc=$?"
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --kernel-breakdown > gpurun_out/${T}_bench_2048.json 2> gpurun_out/${T}_bench_2048.err
rm -rf gpurun_out/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -- python3 bench.py --steps 20 --warmup 5 --settle 0 --no-verify --cpu-planes 0 > gpurun_out/${T}_bench_traced.json 2> /dev/null
cp $(ls gpurun_out/kstats/*/*kernel_stats.csv | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/kstats
bash tools/traffic.sh $T > /dev/null 2>&1
bash tools/pmc.sh $T > /dev/null 2>&1; mv gpurun_out/pmc_$T.txt gpurun_out/${T}_pmc_per_kernel.txt; rm -rf gpurun_out/pmcd_${T}_*
bash tools/chain_trace.sh > gpurun_out/${T}_chain_trace.txt 2>&1
head -14 gpurun_out/${T}_kernel_stats.csv
grep "mb_per_plane" gpurun_out/${T}_traffic.json
else
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --shading --cpu-planes 64 > gpurun_out/${T}_bench_2048_shading.json 2> /dev/null
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --shape 1800x1800 --cpu-planes 128 > gpurun_out/${T}_bench_1800.json 2> /dev/null
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --shape 1600x2000 --cpu-planes 128 > gpurun_out/${T}_bench_1600x2000.json 2> /dev/null
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --shape 1600x2000 --shading --cpu-planes 64 > gpurun_out/${T}_bench_1600x2000_shading.json 2> /dev/null
DSX_SHARE_GPU=1 timeout -k 10 200 python bench.py --gpus 2 --steps 20 --warmup 5 --batch 64 > gpurun_out/${T}_rehearsal_self_launch_2ranks_shared_gpu.json 2> gpurun_out/${T}_rehearsal.err; echo "self-launch rehearsal rc=$?"
timeout -k 10 300 python tools/bench_zarr.py 4096 > gpurun_out/${T}_bench_zarr_4096.json 2> gpurun_out/${T}_bz.err; echo "zarr raw rc=$?"
timeout -k 10 300 python tools/bench_zarr.py 4096 blosc > gpurun_out/${T}_bench_zarr_4096_blosc.json 2> gpurun_out/${T}_bzb.err; echo "zarr blosc rc=$?"
timeout -k 10 200 python tools/latency_small.py > gpurun_out/${T}_latency_small.txt 2>/dev/null; echo "latency rc=$?"
fi
for f in gpurun_out/${T}_bench_*.json gpurun_out/${T}_rehearsal_*.json; do [ -s $f ] && python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('frac_of_attainable'), d.get('verified'), d.get('cpu_baseline',{}).get('value'), d.get('n_gpus'))"; done
