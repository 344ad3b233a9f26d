#!/usr/bin/env python3
"""Static instruction mix of the device code: python tools/isa_mix.py [kernel-name-substring ...]
(compiles csrc/dsx.hip to gfx950 assembly, counts instruction classes per kernel; a static count, loops are not weighted)"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = "/tmp/dsx_isa.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "--cuda-device-only", "-S", "-o", S,
                "dsx.hip"] + [a for a in sys.argv[1:] if a.startswith("-D")], cwd=os.path.join(ROOT, "aind_smartspim_destripe_amd", "csrc"), check=True,
               stderr=subprocess.DEVNULL)
want = [a for a in sys.argv[1:] if not a.startswith("-D")] or ["k_rowfinalILi18", "k_fwd_marchILi0ELb1ELi8", "k_rowfilterILi18ELi2ELi1ELi1ELi2", "k_hist"]
cur = None; mix = {}
for line in open(S):
    m = re.match(r"(_Z\w+):", line)
    if m:
        cur = m.group(1); mix[cur] = collections.Counter(); continue
    if line.startswith(".Lfunc_end"):
        cur = None
    if cur and line.startswith("\t") and not line.lstrip().startswith((".", ";")):
        mix[cur][line.split()[0]] += 1
for name, c in mix.items():
    if not any(w in name for w in want): continue
    tot = sum(c.values())
    grp = lambda p: sum(n for k, n in c.items() if k.startswith(p))
    print("%s\n   total %d  valu %d (pk %d, fma/mac %d, cndmask %d, mov %d, cmp %d)  salu %d  lds %d  vmem %d  waitcnt %d" % (
        name[:90], tot, grp("v_"), grp("v_pk_"), grp("v_fma") + grp("v_fmac") + grp("v_pk_fma"), grp("v_cndmask"), grp("v_mov") + grp("v_accvgpr"), grp("v_cmp"),
        grp("s_") - grp("s_waitcnt") - grp("s_nop"), grp("ds_"), grp("buffer_") + grp("global_") + grp("scratch_"), grp("s_waitcnt")))
    print("   ", ", ".join("%s %d" % kv for kv in c.most_common(18)))
