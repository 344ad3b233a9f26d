cd $GRAFT_REPO_ROOT
OUT=gpurun_out/wpb_sweep.txt; : > $OUT
for round in 1 2 3 4 5; do
  for v in "" "DSX_ROW_WPB=4" "DSX_ROW_WPB=2" "DSX_STREAMS=1" "DSX_STREAMS=1 DSX_ROW_WPB=4" "DSX_STREAMS=1 DSX_ROW_WPB=2"; do
    r=$(env $v timeout -k 10 120 python bench.py --steps 100 --warmup 20 --cpu-planes 0 --settle 0.3 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "[$v] $r" | tee -a $OUT
  done
done
