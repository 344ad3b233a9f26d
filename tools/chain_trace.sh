#!/bin/bash
# every kernel of one launch chain (single stream, 256 planes or CHAIN_BATCH) in launch order with its duration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ct
DSX_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ct -- python3 bench.py --steps 1 --warmup 1 --cpu-planes 0 --settle 0 --no-verify ${CHAIN_BATCH:+--batch $CHAIN_BATCH} > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/ct/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsx::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void dsx::", "").replace("dsx::", "").split("(")[0], r.get("Grid_Size", r.get("Grid_Size_X", ""))))
rows.sort()
half = rows[len(rows) // 2:]  # the timed step
t0 = half[0][0]
for s, e, n, g in half:
    print("%8.1f us  +%7.1f us  %-34s grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, n, g))
PY
rm -rf gpurun_out/ct
