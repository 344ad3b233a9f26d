#!/usr/bin/env python3
"""NumPy model of the exact row median of k_rowfilter / k_rowfinal (value-domain bracketing, csrc/dsx_kernels.h): the loop
of round 2 (`kernel_median`: regula falsi, key midpoint on every fourth step) and of round 3 (`kernel_median2`: Illinois
variant, key midpoint on every eighth step), float32 arithmetic as on the device.  Run on the level-1 / level-2 rows of
synthetic planes (through the oracle) and on adversarial rows (ties, constants, up to 90 % zeros, heavy tails, odd and even
lengths 12 ... 1027); prints the number of count passes per row and checks every median against np.median.
usage: python tools/median_model.py            (CPU only; test infrastructure, nothing in the product imports it)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from aind_smartspim_destripe_amd import synth
from oracle import destripe_oracle as orc

def f32_key(v):
    b = np.float32(v).view(np.uint32)
    return np.uint32(~b) if (b & np.uint32(0x80000000)) else np.uint32(b | np.uint32(0x80000000))
def key_f32(k):
    k = np.uint32(k)
    b = (k & np.uint32(0x7FFFFFFF)) if (k & np.uint32(0x80000000)) else np.uint32(~k)
    return np.uint32(b).view(np.float32)

def kernel_median(x, thr, strategy="cur"):
    """x: float32 row (masked entries already zeroed); returns (median, number of count passes)"""
    x = x.astype(np.float32); N = x.size
    k1 = (N - 1) >> 1; even = (N & 1) == 0
    ncount = 0
    def C(t, le=False):
        nonlocal ncount; ncount += 1
        return int((x <= t).sum()) if le else int((x < t).sum())
    lt = C(np.float32(0)); le = C(np.float32(0), True)
    thr = np.float32(thr)
    thr_up = (thr.view(np.uint32) + np.uint32(1)).view(np.float32)
    done = False
    if k1 < lt: lo, clo, hi, chi = -thr, 0, np.float32(0), lt
    elif k1 < le: lo, clo, hi, chi = np.float32(0), lt, np.uint32(1).view(np.float32), le; done = True
    else: lo, clo, hi, chi = np.uint32(1).view(np.float32), le, thr_up, N
    same = 0; last = 0
    it = 0
    while not done and it < 200:
        kl, kh = f32_key(lo), f32_key(hi)
        if chi - clo <= 1 or int(kh) - int(kl) <= 1: break
        frac = np.float32((np.float32(k1 - clo) + np.float32(0.5)) / np.float32(chi - clo))
        if strategy == "cur":
            tt = np.float32(lo + np.float32(hi - lo) * frac)
            tm = key_f32(np.uint32(int(kl) + ((int(kh) - int(kl)) >> 1)))
            if same >= 3 or (it & 3) == 3 or not (tt > lo and tt < hi): tt = tm; same = 0
        t = tt
        c = C(t)
        side = 1 if c <= k1 else 2
        if side == 1: lo, clo = t, c
        else: hi, chi = t, c
        same = same + 1 if side == last else 1
        last = side
        it += 1
    sk = x[x >= lo].min()
    sk1 = sk
    if even:
        nxt = x[x >= hi].min() if (x >= hi).any() else np.float32(np.inf)
        sk1 = sk if (k1 + 1 < chi) else nxt
    med = np.float32(0.5) * (sk + sk1) if even else sk
    return med, ncount

if __name__ == "__main__":  # pragma: no cover
    from parity_util import oracle_plane
    tot = {}
    for k in (1, 4):
        plane = synth.synthetic_plane(k, 2048, 2048)
        which, fore, back, ref, stages = oracle_plane(plane)
        for lv in (0, 1):
            st = stages[lv]
            ch = st["ch"].astype(np.float32); thr = np.float32(st["threshold"])
            bg = np.where(np.abs(ch) > thr, np.float32(0), ch)
            cnts = []; bad = 0; nomask = 0
            for r in range(0, ch.shape[0], 7):
                row = bg[r]
                if not (np.abs(ch[r]) > thr).any(): nomask += 1
                m, n = kernel_median(row, thr)
                if m != np.float32(np.median(row.astype(np.float32))): bad += 1
                cnts.append(n)
            cnts = np.array(cnts)
            print("plane", k, "cfg", which, "level", lv + 1, "thr %.4f" % thr, "rows", len(cnts), "mean counts %.2f" % cnts.mean(), "hist", np.bincount(cnts)[:20], "bad", bad, "rows without mask", nomask, "mask frac %.4f" % (np.abs(ch) > thr).mean(), "sigma %.4f" % ch.std())

def kernel_median2(x, thr, mode="illinois", safety=8):
    x = x.astype(np.float32); N = x.size
    k1 = (N - 1) >> 1; even = (N & 1) == 0
    ncount = 0
    def C(t, le=False):
        nonlocal ncount; ncount += 1
        return int((x <= t).sum()) if le else int((x < t).sum())
    lt = C(np.float32(0)); le = C(np.float32(0), True)
    thr = np.float32(thr)
    thr_up = (thr.view(np.uint32) + np.uint32(1)).view(np.float32)
    done = False
    if k1 < lt: lo, clo, hi, chi = -thr, 0, np.float32(0), lt
    elif k1 < le: lo, clo, hi, chi = np.float32(0), lt, np.uint32(1).view(np.float32), le; done = True
    else: lo, clo, hi, chi = np.uint32(1).view(np.float32), le, thr_up, N
    target = np.float32(k1) + np.float32(0.5)
    flo = np.float32(clo) - target; fhi = np.float32(chi) - target
    last = 0; it = 0
    while not done and it < 200:
        kl, kh = f32_key(lo), f32_key(hi)
        if chi - clo <= 1 or int(kh) - int(kl) <= 1: break
        tt = np.float32((lo * fhi - hi * flo) / (fhi - flo))
        if (safety and (it % safety) == safety - 1) or not (tt > lo and tt < hi):
            tt = key_f32(np.uint32(int(kl) + ((int(kh) - int(kl)) >> 1)))
        c = C(tt)
        f = np.float32(c) - target
        if c <= k1:
            lo, clo, flo = tt, c, f
            if last == 1: fhi = np.float32(fhi * 0.5)
            last = 1
        else:
            hi, chi, fhi = tt, c, f
            if last == 2: flo = np.float32(flo * 0.5)
            last = 2
        it += 1
    sk = x[x >= lo].min()
    sk1 = sk
    if even:
        nxt = x[x >= hi].min() if (x >= hi).any() else np.float32(np.inf)
        sk1 = sk if (k1 + 1 < chi) else nxt
    med = np.float32(0.5) * (sk + sk1) if even else sk
    return med, ncount


def kernel_median3(x, thr, extra=0):
    """Round-3 loop as shipped: Illinois false position, adjacency of lo / hi only tested on the fallback path, and a
    finished row keeps stepping while its partner is not finished (`extra` more steps here) -- the invariant
    C(lo) <= k < C(hi) must survive that."""
    x = x.astype(np.float32); N = x.size
    k1 = (N - 1) >> 1; even = (N & 1) == 0
    ncount = 0
    def C(t, le=False):
        nonlocal ncount; ncount += 1
        return int((x <= t).sum()) if le else int((x < t).sum())
    lt = C(np.float32(0)); le = C(np.float32(0), True)
    thr = np.float32(thr)
    thr_up = (thr.view(np.uint32) + np.uint32(1)).view(np.float32)
    adj = False
    if k1 < lt: lo, clo, hi, chi = -thr, 0, np.float32(0), lt
    elif k1 < le: lo, clo, hi, chi = np.float32(0), lt, np.uint32(1).view(np.float32), le; adj = True
    else: lo, clo, hi, chi = np.uint32(1).view(np.float32), le, thr_up, N
    target = np.float32(k1) + np.float32(0.5)
    flo = np.float32(clo) - target; fhi = np.float32(chi) - target
    last = 0; it = 0; extra_left = extra
    with np.errstate(all="ignore"):
        while it < 256:
            if adj or chi - clo <= 1:
                if extra_left == 0: break
                extra_left -= 1
            tt = np.float32(np.float32(lo * fhi - hi * flo) * (np.float32(1) / np.float32(fhi - flo)))
            if (it & 7) == 7 or not (tt > lo and tt < hi):
                kl, kh = f32_key(lo), f32_key(hi)
                if int(kh) - int(kl) <= 1: adj = True
                tt = key_f32(np.uint32(int(kl) + ((int(kh) - int(kl)) >> 1)))
            c = C(tt)
            f = np.float32(c) - target
            if c <= k1:
                lo, clo, flo = tt, c, f
                if last == 1: fhi = np.float32(fhi * 0.5)
                last = 1
            else:
                hi, chi, fhi = tt, c, f
                if last == 2: flo = np.float32(flo * 0.5)
                last = 2
            it += 1
    assert clo <= k1 < chi
    sk = x[x >= lo].min()
    sk1 = sk
    if even:
        nxt = x[x >= hi].min() if (x >= hi).any() else np.float32(np.inf)
        sk1 = sk if (k1 + 1 < chi) else nxt
    med = np.float32(0.5) * (sk + sk1) if even else sk
    return med, ncount
