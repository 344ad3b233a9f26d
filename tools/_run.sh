cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_rowfinal or multistream or baseline_shapes" > gpurun_out/rf6_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/rf6_tests.log
for i in 1 2; do
for shp in 1600x2000 1800x1800; do
for m in 0 1; do
  r=$(DSX_NO_FUSE_RF=$m timeout -k 10 200 python bench.py --shape $shp --steps 100 --warmup 20 --cpu-planes 0 --settle 0.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])")
  echo "$shp no_fuse_rf=$m $r" | tee -a gpurun_out/rf6_ab.txt
done
done
done
