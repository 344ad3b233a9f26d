#!/bin/bash
# build an experiment variant of the library next to the product one: tools/build_variant.sh <name> [-DFLAG ...]
# -> aind_smartspim_destripe_amd/_lib/libdsx_<name>.so (select with DSX_LIB=... ; travels to the GPU box)
# The timing-only switches (DSX_ABLATE, DSX_SKIP_HIST / _ROW / _COARSE: WRONG pixels) only exist in a variant built with
# -DDSX_DIAG, e.g.  tools/build_variant.sh diag -DDSX_DIAG ; DSX_LIB=.../libdsx_diag.so DSX_ABLATE=16 python bench.py
N=$1; shift
cd "$(dirname "$0")/../aind_smartspim_destripe_amd/csrc" && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -shared -fPIC "$@" -o ../_lib/libdsx_$N.so dsx.hip -lz -lpthread
