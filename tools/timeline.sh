#!/bin/bash
# 4-stream timeline of the headline bench: how busy is the device, how many kernels run side by side, and how much
# longer each kernel type takes inside the mix than alone (tools/chain_trace.sh).  -> stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 6 --warmup 3 --cpu-planes 0 --settle 0 --no-verify > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsx::" in r["Kernel_Name"]:
            n = r["Kernel_Name"].replace("void dsx::", "").replace("dsx::", "").split("(")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
t_lo, t_hi = rows[0][0], rows[-1][1]
span = t_hi - t_lo
# steady window: the last two thirds
w0 = t_lo + span // 3
sel = [r for r in rows if r[0] >= w0]
ev = []
for s, e, n, q in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; conc_time = collections.Counter(); cur = 0; last = ev[0][0]
for t, d in ev:
    if cur > 0: busy += t - last
    conc_time[cur] += t - last
    cur += d; last = t
win = ev[-1][0] - ev[0][0]
print("window %.2f ms, device busy %.1f %%" % (win / 1e6, 100.0 * busy / win))
for c in sorted(conc_time):
    print("  %d kernels side by side: %5.1f %% of the time" % (c, 100.0 * conc_time[c] / win))
agg = collections.defaultdict(list)
for s, e, n, q in sel:
    agg[n].append(e - s)
tot = sum(sum(v) for v in agg.values())
print("kernel time summed over streams: %.2f ms per ms of wall = %.2f" % (tot / 1e6, tot / win))
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("  %-34s calls %4d  avg %8.1f us  share of summed time %5.1f %%" % (n, len(v), sum(v) / len(v) / 1e3, 100.0 * sum(v) / tot))
PY
rm -rf gpurun_out/tl
