#!/bin/bash
# like ab_env.sh with extra bench arguments: tools/ab_env2.sh <tag> "<bench args>" "<lib> VAR=.." ...
cd $GRAFT_REPO_ROOT
TAG=$1; ARGS=$2; shift; shift
OUT=gpurun_out/${TAG}_abenv.txt; : > $OUT
L=$PWD/aind_smartspim_destripe_amd/_lib
for round in 1 2 3 4; do
  for spec in "$@"; do
    lib=${spec%% *}; envs=""; [ "$spec" != "$lib" ] && envs=${spec#* }
    r=$(env DSX_LIB=$L/libdsx_$lib.so $envs python bench.py --steps 100 --warmup 20 --cpu-planes 0 --settle 0 --no-verify $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "[$ARGS] $spec -> $r" | tee -a $OUT
  done
done
