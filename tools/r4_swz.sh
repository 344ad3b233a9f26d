#!/bin/bash
# bank-swizzled LDS addressing of the power-of-two row-filter plans (product) against -DDSX_SWZ=0 (tools/build_variant.sh noswz -DDSX_SWZ=0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_zarr_chunk_map.py -m gpu -x -q > gpurun_out/r4_swz_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 gpurun_out/r4_swz_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
L=$PWD/aind_smartspim_destripe_amd/_lib
for lib in hip noswz; do
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
    for lib in hip noswz; do
      r=$(DSX_LIB=$L/libdsx_$lib.so timeout -k 10 200 python bench.py --shape $shape --steps 100 --warmup 20 --cpu-planes 0 --settle 0.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])")
      echo "[$shape $lib] $r" | tee -a $OUT
    done
  done
done
