#!/bin/bash
# SQ / TCC counters per kernel of one single-stream launch chain over 64 planes (separate passes, no tracing).
# usage: bash tools/pmc.sh <tag> [lib]     -> gpurun_out/pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-cur}; LIB=${2:-}
[ -n "$LIB" ] && export DSX_LIB=$LIB
export DSX_STREAMS=1
ARGS="--batch 64 --steps 1 --warmup 1 --settle 0 --no-verify --cpu-planes 0"
P=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  P=$((P+1))
  rm -rf gpurun_out/pmcd_$TAG_$P
  rocprofv3 --pmc $SET -d gpurun_out/pmcd_${TAG}_$P --output-format csv -- python3 bench.py $ARGS > /dev/null 2> gpurun_out/pmcd_${TAG}_$P.err
done
python3 - $TAG <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/pmcd_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsx::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dsx::", "") + " grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open("gpurun_out/pmc_%s.txt" % tag, "w") as out:
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0) / max(1, cnt[k].get("SQ_WAVE_CYCLES", 1))):
        m = {c: agg[k][c] / cnt[k][c] for c in agg[k]}
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        line = "%-44s waves %7d  VALU %9.3e SALU %9.3e LDS %9.3e VMEMr %8.2e VMEMw %8.2e | of wave-cycles: active %4.1f%% (valu %4.1f%% lds %4.1f%%) wait_any %4.1f%% wait_inst %4.1f%% | ldsconf/idx %4.1f%% | fetch*2 %8.1f MB write %8.1f MB  L2hit %4.1f%% | busy %9.3e gui %9.3e" % (
            k, m.get("SQ_WAVES", 0), m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_LDS", 0),
            m.get("SQ_INSTS_VMEM_RD", 0), m.get("SQ_INSTS_VMEM_WR", 0),
            100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_LDS", 0) / wc,
            100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc,
            100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, m.get("SQ_LDS_IDX_ACTIVE", 1)),
            2 * m.get("FETCH_SIZE", 0) / 1e3, m.get("WRITE_SIZE", 0) / 1e3,
            100 * m.get("TCC_HIT_sum", 0) / max(1, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)),
            m.get("SQ_BUSY_CYCLES", 0), m.get("GRBM_GUI_ACTIVE", 0))
        out.write(line + "\n")
print(open("gpurun_out/pmc_%s.txt" % tag).read())
PY
rm -rf gpurun_out/pmcd_${TAG}_*
