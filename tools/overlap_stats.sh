#!/bin/bash
# How long does each big kernel take depending on what runs beside it?  Kernel trace of the 4-stream bench; for every
# instance of a big kernel: its duration and the share of it that each other big kernel type was running too.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ov
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ov -- python3 bench.py --steps 30 --warmup 5 --cpu-planes 0 --settle 0 --no-verify > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
import numpy as np
rows = []
for f in glob.glob("gpurun_out/ov/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsx::" in r["Kernel_Name"]:
            n = r["Kernel_Name"].replace("void dsx::", "").replace("dsx::", "").split("(")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) // 5
rows = [r for r in rows if r[0] >= t0]
big = {"k_fwd_march<0, true, 8>": "F", "k_inv_march<0, true, 8>": "X", "k_rowfilter<18, 4, 1, 0, 1>": "R1",
       "k_rowfilter<18, 2, 1, 1, 2>": "R2"}
inst = [(s, e, big[n]) for s, e, n in rows if n in big]
kinds = ["F", "X", "R1", "R2"]
for k in kinds:
    X, y = [], []
    for s, e, n in inst:
        if n != k: continue
        ov = dict.fromkeys(kinds, 0.0)
        for s2, e2, n2 in inst:
            if e2 <= s or s2 >= e or (s2 == s and e2 == e and n2 == n): continue
            ov[n2] += (min(e, e2) - max(s, s2)) / (e - s)
        X.append([ov[q] for q in kinds]); y.append((e - s) / 1e3)
    X = np.array(X); y = np.array(y)
    A = np.hstack([np.ones((len(y), 1)), X])
    coef, *_ = np.linalg.lstsq(A, y, rcond=None)
    print("%-3s n=%3d  mean %6.1f us (min %6.1f max %6.1f); mean overlap shares F %.2f X %.2f R1 %.2f R2 %.2f" % (
        k, len(y), y.mean(), y.min(), y.max(), *X.mean(0)))
    print("      fit: duration = %.0f us  + %.0f F + %.0f X + %.0f R1 + %.0f R2   (us per unit of overlap share)" % tuple(coef))
PY
rm -rf gpurun_out/ov
