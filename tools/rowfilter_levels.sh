#!/bin/bash
# per-launch durations of k_rowfilter<18> (level 1 and level 2 at 2048^2) for the given builds, single stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "$@"; do
  rm -rf gpurun_out/rl
  DSX_LIB=$L DSX_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rl -- python3 bench.py --steps 2 --warmup 1 --cpu-planes 0 > /dev/null 2>&1
  python3 - "$L" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("gpurun_out/rl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_rowfilter<18" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
d = [x[1] / 1e6 for x in rows]
print(sys.argv[1], "k_rowfilter<18,..> launches (ms):", " ".join("%.3f" % x for x in d))
PY
done
rm -rf gpurun_out/rl
