cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multistream or fused_rowfinal" > gpurun_out/q1_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/q1_tests.log
OUT=gpurun_out/q1_sweep.txt; : > $OUT
for round in 1 2 3; do
  for v in "" "DSX_NO_QUANT=1" "DSX_STREAMS=1" "DSX_STREAMS=1 DSX_NO_QUANT=1"; do
    r=$(env $v timeout -k 10 120 python bench.py --steps 100 --warmup 20 --cpu-planes 0 --settle 0.3 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "[$v] $r" | tee -a $OUT
  done
done
for v in "" "DSX_NO_QUANT=1"; do
env $v DSX_STREAMS=1 timeout -k 10 200 python bench.py --cpu-planes 0 --steps 5 --warmup 2 --settle 0.2 --no-verify --kernel-breakdown 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print('[$v]', 'value', d['value'], ' '.join('%s=%.3f' % (n[2:].replace('_march','').replace('(',':').rstrip(')'), v['ms']) for n, v in k.items()))" | tee -a $OUT
done
