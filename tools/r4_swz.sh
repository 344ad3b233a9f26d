#!/bin/bash
# bank-swizzled LDS addressing of the power-of-two row-filter plans: product (addressing written out) against the generic
# index policy (tools/build_variant.sh swzgen -DDSX_SWZ=2) and plain addressing (tools/build_variant.sh noswz -DDSX_SWZ=0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_zarr_chunk_map.py -m gpu -x -q > gpurun_out/r4_swz_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 gpurun_out/r4_swz_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
L=$PWD/aind_smartspim_destripe_amd/_lib
python - <<'PY'
# the three builds must return the same bits
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from aind_smartspim_destripe_amd import engine as E, synth
stack = synth.synthetic_stack(12, 402, 2000, n_unique=4)
outs = []
for lib in ("hip", "swzgen", "noswz"):
    os.environ["DSX_LIB"] = os.path.join(os.getcwd(), "aind_smartspim_destripe_amd", "_lib", "libdsx_%s.so" % lib)
    import subprocess, json
    code = "import os,sys,numpy as np,hashlib;sys.path.insert(0,os.getcwd());from aind_smartspim_destripe_amd import engine as E, synth;s=synth.synthetic_stack(12,402,2000,n_unique=4);e=E.DestripeEngine(0);e.plan(402,2000,synth.CELLS_CONFIG,synth.NO_CELLS_CONFIG,2500,max_batch=12);print(hashlib.sha256(e.run(s,out_dtype=np.float32).tobytes()).hexdigest())"
    outs.append(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ)).stdout.strip())
print("bit-identical across builds:", len(set(outs)) == 1, [o[:10] for o in outs])
PY
for lib in hip swzgen noswz; do
rm -rf gpurun_out/ct
DSX_LIB=$L/libdsx_$lib.so DSX_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ct -- python3 bench.py --shape 1600x2000 --steps 1 --warmup 1 --cpu-planes 0 --settle 0 --no-verify > /dev/null 2>&1
python3 - $lib <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("gpurun_out/ct/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsx::k_rowfilter<36" in r["Kernel_Name"] or "dsx::k_rowfilter<18, 1, 4" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
for s,e,n in rows[len(rows)//2:]: print("%s alone: %8.1f us %s" % (sys.argv[1], (e-s)/1e3, n))
PY
done
rm -rf gpurun_out/ct
OUT=gpurun_out/r4_swz_ab.txt; : > $OUT
for round in 1 2 3; do
  for shape in "1600x2000" "1600x2000 --shading"; do
    for lib in hip swzgen noswz; do
      r=$(DSX_LIB=$L/libdsx_$lib.so timeout -k 10 200 python bench.py --shape $shape --steps 100 --warmup 20 --cpu-planes 0 --settle 0.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])")
      echo "[$shape $lib] $r" | tee -a $OUT
    done
  done
  r=$(DSX_STREAMS=1 DSX_LIB=$L/libdsx_hip.so timeout -k 10 200 python bench.py --shape 1600x2000 --steps 50 --warmup 10 --cpu-planes 0 --settle 0.5 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); echo "[1600x2000 hip DSX_STREAMS=1] $r" | tee -a $OUT
  r=$(DSX_STREAMS=1 DSX_LIB=$L/libdsx_noswz.so timeout -k 10 200 python bench.py --shape 1600x2000 --steps 50 --warmup 10 --cpu-planes 0 --settle 0.5 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); echo "[1600x2000 noswz DSX_STREAMS=1] $r" | tee -a $OUT
done
