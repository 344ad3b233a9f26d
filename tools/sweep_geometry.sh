#!/bin/bash
# launch-geometry switches of the library against the headline bench (one variable at a time, two rounds)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2}_geometry.txt; : > $OUT
run() {
  r=$(env "$@" python bench.py --steps 100 --warmup 20 --cpu-planes 0 --settle 0 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$* -> $r" | tee -a $OUT
}
for round in 1 2; do
  run DSX_X=0
  for w in 2048 3072 6144 8192; do run DSX_MARCH_WAVES=$w; done
  for h in 16 24 48 64; do run DSX_HIST_ROWS=$h; done
  for w in 4 6 7; do run DSX_ROW_WPB=$w; done
  run DSX_FWD_WPB=4
  run DSX_INV_WPB=4
  run DSX_FWD_WPB=4 DSX_INV_WPB=4
done
