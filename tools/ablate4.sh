#!/bin/bash
# 4-stream throughput under row-filter ablations (diagnosis only; results are wrong with DSX_ABLATE != 0)
# needs a -DDSX_DIAG build:  tools/build_variant.sh diag -DDSX_DIAG  (the product library ignores DSX_ABLATE)
export DSX_LIB=${DSX_LIB:-$(dirname "$0")/../aind_smartspim_destripe_amd/_lib/libdsx_diag.so}
for A in 0 1 2 3 7; do
  DSX_ABLATE=$A python bench.py --cpu-planes 0 --steps 40 --warmup 5 --settle 0.3 --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ablate', $A, '4-stream', d['value'], d['ms_per_step'])"
done
