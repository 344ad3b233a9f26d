#!/bin/bash
# level-2 row filter of a 2048-wide plane: the planner's embedding (515 values in M = 1071 = 17*9*7) against the
# direct length-515 = 5*103 transform (generic O(R) pass for the prime 103), same session, single stream
cd $GRAFT_REPO_ROOT
for D in "" 1; do
  DSX_PLAN_DIRECT_LEVEL=$D DSX_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pd -- python3 bench.py --steps 2 --warmup 1 --settle 0 --no-verify --cpu-planes 0 > /dev/null 2>&1
  python3 - "$D" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pd/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_rowfilter" in r["Kernel_Name"]:
            rows[(r["Kernel_Name"].split("(")[0].replace("void dsx::", ""), r.get("Grid_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("direct level:", sys.argv[1] or "none")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print("   %-34s grid %-8s  %8.1f us" % (k[0], k[1], sum(v) / len(v)))
PY
  rm -rf gpurun_out/pd
done
