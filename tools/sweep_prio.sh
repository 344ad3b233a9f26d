#!/bin/bash
# headline bench against stream priorities of the four sub-cohort streams (DSX_PRIO, lower = more urgent)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-sweep}_prio.txt; : > $OUT
python - <<'PY' | tee -a $OUT
import ctypes
h = ctypes.CDLL("libamdhip64.so"); a = ctypes.c_int(); b = ctypes.c_int()
h.hipDeviceGetStreamPriorityRange(ctypes.byref(a), ctypes.byref(b)); print("priority range least", a.value, "greatest", b.value)
PY
for round in 1 2; do
for pr in "0,0,0,0" "-1,0,0,0" "-1,-1,0,0" "-1,0,-1,0" "-1,0,1,1" "1,0,0,-1" "-1,-1,-1,0"; do
  r=$(DSX_PRIO=$pr python bench.py --steps 100 --warmup 20 --cpu-planes 0 --settle 0 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "prio=$pr $r" | tee -a $OUT
done
done
