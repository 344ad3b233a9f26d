cd $GRAFT_REPO_ROOT
L=$PWD/aind_smartspim_destripe_amd/_lib
for v in c0 cw8 cw4 hip; do echo "== $v"; DSX_LIB=$L/libdsx_$v.so timeout -k 10 200 python tools/latency_small.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['planes'], d['latency_us'], d['back_to_back_us'], d['planes_per_s'])"; done | tee gpurun_out/c3_latency.txt
