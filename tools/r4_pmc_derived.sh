#!/bin/bash
# derived rocprofv3 metrics per kernel of one single-stream chain over 64 planes: bash tools/r4_pmc_derived.sh [HxW]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=${1:-2048x2048}
export DSX_STREAMS=1
ARGS="--shape $S --batch 64 --steps 1 --warmup 1 --settle 0 --no-verify --cpu-planes 0"
P=0
for SET in "VALUBusy SALUBusy VALUUtilization" "MemUnitBusy MemUnitStalled WriteUnitStalled" "LDSBankConflict L2CacheHit" "VALUInsts SALUInsts VFetchInsts VWriteInsts LDSInsts"; do
  P=$((P+1))
  rm -rf gpurun_out/pmcx_$P
  rocprofv3 --pmc $SET -d gpurun_out/pmcx_$P --output-format csv -- python3 bench.py $ARGS > /dev/null 2> gpurun_out/pmcx_$P.err
  tail -2 gpurun_out/pmcx_$P.err | cut -c1-200
done
python3 - <<'PY' | tee gpurun_out/r4_pmc_derived.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/pmcx_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsx::" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dsx::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
names = sorted({c for k in agg for c in agg[k]})
print("kernel".ljust(38), " ".join(n[:12].rjust(12) for n in names))
for k in sorted(agg, key=lambda k: -agg[k].get("VALUInsts", 0) * cnt[k].get("VALUInsts", 1)):
    print(k[:38].ljust(38), " ".join(("%12.2f" % (agg[k][n] / cnt[k][n])) if n in agg[k] else " " * 12 for n in names))
PY
rm -rf gpurun_out/pmcx_*
