cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_rowfinal or timing_switches or multistream or baseline_shapes" > gpurun_out/rf1_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/rf1_tests.log
for i in 1 2 3; do
for m in 0 1; do
  r=$(DSX_NO_FUSE_RF=$m timeout -k 10 200 python bench.py --steps 100 --warmup 20 --cpu-planes 0 --settle 0.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])")
  echo "no_fuse_rf=$m streams=4 $r" | tee -a gpurun_out/rf1_ab.txt
done
done
for m in 0 1; do
DSX_NO_FUSE_RF=$m DSX_STREAMS=1 timeout -k 10 200 python bench.py --cpu-planes 0 --steps 5 --warmup 2 --settle 0.2 --no-verify --kernel-breakdown 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print('no_fuse_rf=$m', 'value', d['value'], ' '.join('%s=%.3f' % (n[2:].replace('_march','').replace('(',':').rstrip(')'), v['ms']) for n, v in k.items()))" | tee -a gpurun_out/rf1_ab.txt
done
