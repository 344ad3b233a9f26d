#!/bin/bash
# A/B of library builds x environment settings, interleaved: tools/ab_env.sh <tag> "<lib> VAR=.. VAR=.." ...
cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/${TAG}_abenv.txt; : > $OUT
L=$PWD/aind_smartspim_destripe_amd/_lib
for round in 1 2 3; do
  for spec in "$@"; do
    lib=${spec%% *}; envs=""; [ "$spec" != "$lib" ] && envs=${spec#* }
    r=$(env DSX_LIB=$L/libdsx_$lib.so $envs python bench.py --steps 100 --warmup 20 --cpu-planes 0 --settle 0 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$spec -> $r" | tee -a $OUT
  done
done
