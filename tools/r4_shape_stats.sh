#!/bin/bash
# per-kernel durations inside the 4-stream run and single-stream chain trace for another shape: bash tools/r4_shape_stats.sh 1600x2000 [--shading]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=${1:-1600x2000}; shift
mkdir -p gpurun_out; rm -rf gpurun_out/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -- python3 bench.py --shape $S "$@" --steps 20 --warmup 5 --settle 0 --no-verify --cpu-planes 0 > gpurun_out/r4_bench_traced_$S.json 2> /dev/null
cp $(ls gpurun_out/kstats/*/*kernel_stats.csv | head -1) gpurun_out/r4_kernel_stats_$S.csv; rm -rf gpurun_out/kstats
cut -c1-150 gpurun_out/r4_kernel_stats_$S.csv | head -14
rm -rf gpurun_out/ct
DSX_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ct -- python3 bench.py --shape $S "$@" --steps 1 --warmup 1 --cpu-planes 0 --settle 0 --no-verify > /dev/null 2>&1
python3 - <<'PY' | tee gpurun_out/r4_chain_trace_shape.txt
import csv, glob
rows = []
for f in glob.glob("gpurun_out/ct/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsx::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void dsx::", "").replace("dsx::", "").split("(")[0], r.get("Grid_Size", "")))
rows.sort()
half = rows[len(rows) // 2:]
t0 = half[0][0]
for s, e, n, g in half:
    print("%8.1f us  +%7.1f us  %-34s grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, n, g))
PY
rm -rf gpurun_out/ct
