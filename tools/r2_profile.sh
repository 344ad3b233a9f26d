#!/bin/bash
# round-2 evidence: kernel-trace stats of the headline bench, HBM traffic (PMC), bench lines of the other BASELINE shapes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r2}
rm -rf gpurun_out/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -- python3 bench.py --steps 20 --warmup 5 --settle 0 --no-verify --cpu-planes 0 > gpurun_out/${T}_bench_traced.json 2> /dev/null
cp $(ls gpurun_out/kstats/*/*kernel_stats.csv | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/kstats
bash tools/traffic.sh $T > /dev/null 2>&1
python bench.py --steps 200 --warmup 20 --kernel-breakdown > gpurun_out/${T}_bench_2048.json 2> gpurun_out/${T}_bench_2048.err
python bench.py --steps 200 --warmup 20 --shading --cpu-planes 64 > gpurun_out/${T}_bench_2048_shading.json 2> /dev/null
python bench.py --steps 200 --warmup 20 --shape 1800x1800 --cpu-planes 128 > gpurun_out/${T}_bench_1800.json 2> /dev/null
python bench.py --steps 200 --warmup 20 --shape 1600x2000 --cpu-planes 128 > gpurun_out/${T}_bench_1600x2000.json 2> /dev/null
for f in gpurun_out/${T}_bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['roofline']['frac'], d['verified'], d.get('cpu_baseline',{}).get('value'))"; done
head -12 gpurun_out/${T}_kernel_stats.csv
