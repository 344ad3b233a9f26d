#!/bin/bash
# instruction-cache requests / misses per kernel (SQC counters) in the 4-stream bench and in a single-stream run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for S in 4 1; do
  rm -rf gpurun_out/ic_$S
  DSX_STREAMS=$S rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES -d gpurun_out/ic_$S --output-format csv -- python3 bench.py --steps 2 --warmup 1 --settle 0 --no-verify --cpu-planes 0 > /dev/null 2> gpurun_out/ic_$S.err
  python3 - $S <<'PY'
import csv, glob, sys, collections
S = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/ic_%s/**/*counter_collection.csv" % S, recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsx::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dsx::", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
print("streams =", S)
for k, m in sorted(agg.items(), key=lambda kv: -kv[1].get("SQC_ICACHE_REQ", 0)):
    req, hit, miss = m.get("SQC_ICACHE_REQ", 0), m.get("SQC_ICACHE_HITS", 0), m.get("SQC_ICACHE_MISSES", 0)
    if req > 0:
        print("  %-34s req %.3e  hit %5.1f %%  miss %5.1f %%" % (k, req, 100 * hit / req, 100 * miss / req))
PY
  rm -rf gpurun_out/ic_$S
done
