#!/bin/bash
# Per-stream view of the 4-stream headline bench: for every HIP stream (queue) the share of the steady window in which
# one of its kernels is running, the gaps between consecutive kernels, and one step's launch sequence with start / end
# offsets.  -> stdout (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 6 --warmup 3 --cpu-planes 0 --settle 0 --no-verify > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True):
    rd = csv.DictReader(open(f))
    for r in rd:
        if "dsx::" in r["Kernel_Name"]:
            n = r["Kernel_Name"].replace("void dsx::", "").replace("dsx::", "").split("(")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Stream_Id", "?"), r.get("Queue_Id", "?")))
rows.sort()
t_lo, t_hi = rows[0][0], rows[-1][1]
w0 = t_lo + (t_hi - t_lo) // 3
sel = [r for r in rows if r[0] >= w0]
win = sel[-1][1] - sel[0][0]
by = collections.defaultdict(list)
for r in sel:
    by[(r[3], r[4])].append(r)
print("steady window %.2f ms; streams (stream id, queue id):" % (win / 1e6))
for k, v in sorted(by.items()):
    busy = sum(e - s for s, e, *_ in v)
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    gaps_pos = [g for g in gaps if g > 0]
    print("  %s: %4d kernels, running %5.1f %% of the window, gaps: median %.1f us, mean %.1f us, sum %.2f ms, overlapping launches %d" % (
        k, len(v), 100.0 * busy / win, sorted(gaps_pos)[len(gaps_pos) // 2] / 1e3 if gaps_pos else 0, sum(gaps_pos) / max(1, len(gaps_pos)) / 1e3,
        sum(gaps_pos) / 1e6, sum(1 for g in gaps if g <= 0)))
# one chain of the busiest stream
k0 = max(by, key=lambda k: len(by[k]))
v = by[k0]
starts = [i for i, r in enumerate(v) if r[2].startswith("k_zero3")]
if len(starts) >= 2:
    a, b = starts[0], starts[1]
    t0 = v[a][0]
    print("one chain on stream %s:" % (k0,))
    for r in v[a:b + 1]:
        print("   %9.1f us  + %8.1f us  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2]))
PY
rm -rf gpurun_out/tl
