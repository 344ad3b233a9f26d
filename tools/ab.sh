#!/bin/bash
# A/B of the product library against variants built with tools/build_variant.sh:
#   tools/ab.sh <tag> <variant> [<variant> ...]   (interleaved, 3 rounds; 4-stream and 1-stream lines)
cd $GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_ab.txt; : > $OUT
L=aind_smartspim_destripe_amd/_lib
for round in 1 2 3; do
  for v in hip "$@"; do
    for s in 4 1; do
      r=$(DSX_LIB=$PWD/$L/libdsx_$v.so DSX_STREAMS=$s python bench.py --steps 100 --warmup 20 --cpu-planes 0 --settle 0 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
      echo "$v streams=$s $r" | tee -a $OUT
    done
  done
done
