#!/bin/bash
# SQ counters per kernel of one single-stream chain over 64 planes of another shape: bash tools/r4_pmc_shape.sh 1600x2000
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=${1:-1600x2000}
export DSX_STREAMS=1
ARGS="--shape $S --batch 64 --steps 1 --warmup 1 --settle 0 --no-verify --cpu-planes 0"
P=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE"; do
  P=$((P+1))
  rm -rf gpurun_out/pmcs_$P
  rocprofv3 --pmc $SET -d gpurun_out/pmcs_$P --output-format csv -- python3 bench.py $ARGS > /dev/null 2> gpurun_out/pmcs_$P.err
done
python3 - <<'PY' | tee gpurun_out/r4_pmc_shape.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/pmcs_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsx::" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dsx::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    m = {c: agg[k][c] / cnt[k][c] for c in agg[k]}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-40s waves %7d VALU %9.3e LDS %9.3e | of wave-cycles: valu %4.1f%% lds %4.1f%% wait_any %4.1f%% wait_inst %4.1f%% wait_lds %4.1f%% | lds conflict cycles / idx-active %4.1f%% (conflict %9.3e idx %9.3e) | busy %9.3e gui %9.3e" % (
        k, m.get("SQ_WAVES",0), m.get("SQ_INSTS_VALU",0), m.get("SQ_INSTS_LDS",0), 100*m.get("SQ_ACTIVE_INST_VALU",0)/wc, 100*m.get("SQ_ACTIVE_INST_LDS",0)/wc,
        100*m.get("SQ_WAIT_ANY",0)/wc, 100*m.get("SQ_WAIT_INST_ANY",0)/wc, 100*m.get("SQ_WAIT_INST_LDS",0)/wc,
        100*m.get("SQ_LDS_BANK_CONFLICT",0)/max(1,m.get("SQ_LDS_IDX_ACTIVE",1)), m.get("SQ_LDS_BANK_CONFLICT",0), m.get("SQ_LDS_IDX_ACTIVE",0), m.get("SQ_BUSY_CYCLES",0), m.get("GRBM_GUI_ACTIVE",0)))
PY
rm -rf gpurun_out/pmcs_*
