#!/bin/bash
# HBM bytes per bench step from the PMC counters (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and
# WRITE_SIZE in separate passes, no tracing; FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads),
# units of 1 KB.  usage: bash tools/traffic.sh <round-tag> [bench args, e.g. --shape 1800x1800]
# -> gpurun_out/<tag>_traffic.json (copy to profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r2}; shift
EXTRA="$@"
STEPS=2; WARM=1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$C
  rocprofv3 --pmc $C -d gpurun_out/pmc_$C --output-format csv -- python3 bench.py --steps $STEPS --warmup $WARM --settle 0 --no-verify --cpu-planes 0 $EXTRA > /dev/null 2> gpurun_out/pmc_$C.err
done
python3 - "$TAG" "$EXTRA" <<PY
import csv, glob, json, subprocess, sys
tag, extra = sys.argv[1], sys.argv[2]
tot, per_kernel = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    s = 0.0
    for f in glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "dsx::" in r["Kernel_Name"]:
                v = float(r["Counter_Value"])
                s += v
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                per_kernel.setdefault(k, {}).setdefault(c, 0.0)
                per_kernel[k][c] += v / ($STEPS + $WARM)
    tot[c] = s / ($STEPS + $WARM)
shape = "2048x2048"
if "--shape" in extra:
    shape = extra.split("--shape")[1].split()[0]
try:
    build_hash = open("aind_smartspim_destripe_amd/_lib/libdsx_hip.so.srchash").read().strip()
except Exception:
    build_hash = None
build = "measured " + __import__("time").strftime("%Y-%m-%d %H:%M") + " on the library stamped " + str(build_hash)[:12]
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/traffic.sh) on python3 bench.py --steps 2 --warmup 1 --settle 0 --no-verify --cpu-planes 0 " + extra,
    "build": build,
    "build_hash": build_hash,
    "shape": shape,
    "shading": "--shading" in extra,
    "correction": "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is; units KB -> bytes x1000",
    "fetch_size_kb_per_step": tot["FETCH_SIZE"],
    "write_size_kb_per_step": tot["WRITE_SIZE"],
    "hbm_bytes_per_step": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1000.0,
    "planes_per_step": 256,
    "mb_per_plane": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / 256.0 / 1000.0,
    "per_kernel_mb_per_plane": {k: round((2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) / 256.0 / 1000.0, 3) for k, v in sorted(per_kernel.items())},
}
json.dump(out, open("gpurun_out/%s_traffic.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
