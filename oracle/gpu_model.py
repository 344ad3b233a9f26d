"""NumPy model of the *GPU formulation* of the destripe path  --  TEST INFRASTRUCTURE ONLY.

Where ``destripe_oracle.py`` restates the reference literally, this file restates the algebra the
HIP kernels use, in float32, so that the reformulation itself can be checked
against the oracle on the CPU before/independently of any GPU run:

* Delta-pyramid: only ``aa``/``da`` are analysed; the inverse transform is
  applied to ``Delta_l = -(1 - mask_l) * LP_l(inpainted_l)`` and ``out = (1 + x) * exp(c0) + 1``.
* Two real rows per complex FFT, and the fftpack packed-index gain quirk written as
  ``W[k] = ep[k] * Z[k] + em[k] * Z[(N - k) % N]``.
* Mixed-radix Stockham autosort passes with the same index formulas as ``csrc/dsx_kernels.hip``.
* k-th smallest by bitwise bisection on monotone uint32 keys.
"""

import numpy as np

from oracle import destripe_oracle as orc

F32 = np.float32


# ----------------------------------------------------------------------------------------------
# FFT plan + Stockham passes
# ----------------------------------------------------------------------------------------------
def factorize(n, allowed=(4, 2, 3, 5)):
    """Radix list for length n: 4s, then 2, 3, 5, then remaining primes ascending."""
    radices = []
    for r in allowed:
        while n % r == 0 and n > 1:
            radices.append(r)
            n //= r
    p = 7
    while n > 1:
        while n % p == 0:
            radices.append(p)
            n //= p
        p += 2
    return radices


def stockham_fft(z, radices):
    """Forward DFT (e^{-2 pi i jk/N}) of complex64 rows via Stockham autosort passes.

    Pass with radix r on sub-length n (m = n / r) and stride s:
      y[q + s (r p + k)] = w_n^{p k} * sum_j x[q + s (p + m j)] * w_r^{j k},  p < m, q < s, k < r.
    """
    n_total = z.shape[-1]
    tw = np.exp(-2j * np.pi * np.arange(n_total) / n_total).astype(np.complex64)
    x = z.astype(np.complex64)
    n, s = n_total, 1
    for r in radices:
        m = n // r
        y = np.zeros_like(x)
        p = np.arange(m)[:, None, None]
        q = np.arange(s)[None, :, None]
        k = np.arange(r)[None, None, :]
        dst = (q + s * (r * p + k)).reshape(-1)
        acc = np.zeros(x.shape[:-1] + (m, s, r), dtype=np.complex64)
        for j in range(r):
            src = np.broadcast_to(q + s * (p + m * j), (m, s, 1)).reshape(-1)
            xj = x[..., src].reshape(x.shape[:-1] + (m, s, 1))
            wr = tw[((n_total // r) * ((j * k) % r)) % n_total]
            acc = acc + xj * wr
        wn = tw[((n_total // n) * (p * k)) % n_total]
        acc = acc * wn
        y[..., dst] = acc.reshape(x.shape[:-1] + (-1,))
        x = y
        n, s = m, s * r
    return x


def gain_tables(n, s_rows):
    """ep/em of DESIGN.md: LP gains e[j] = exp(-j^2 / (2 s^2)) in fftpack packed order.

    ep[k] = (e[2k-1] + e[2k]) / 2, em[k] = (e[2k-1] - e[2k]) / 2 for 1 <= k < n/2, mirrored to
    n - k; ep[0] = 1; even n: ep[n/2] = e[n-1], em[n/2] = 0.
    """
    j = np.arange(n, dtype=np.float64)
    e = np.exp(-(j**2) / (2.0 * s_rows**2))
    ep = np.zeros(n)
    em = np.zeros(n)
    ep[0] = e[0]
    for k in range(1, (n - 1) // 2 + 1):
        ea, eb = e[2 * k - 1], e[2 * k]
        ep[k] = ep[n - k] = 0.5 * (ea + eb)
        em[k] = em[n - k] = 0.5 * (ea - eb)
    if n % 2 == 0 and n >= 2:
        ep[n // 2] = e[n - 1]
        em[n // 2] = 0.0
    return ep.astype(F32), em.astype(F32)


def lowpass_rows(inp, s_rows):
    """LP(inp) for every row, two rows per complex transform (odd row count: last pairs with 0)."""
    h, n = inp.shape
    radices = factorize(n)
    ep, em = gain_tables(n, s_rows)
    hp = h + (h & 1)
    buf = np.zeros((hp, n), dtype=F32)
    buf[:h] = inp
    z = (buf[0::2] + 1j * buf[1::2]).astype(np.complex64)
    zf = stockham_fft(z, radices)
    rev = (n - np.arange(n)) % n
    w = ep * zf + em * zf[:, rev]
    # inverse through the forward kernel: swap re/im in, swap re/im out, scale 1/n
    ws = (w.imag + 1j * w.real).astype(np.complex64)
    ys = stockham_fft(ws, radices)
    y = (ys.imag + 1j * ys.real) * F32(1.0 / n)
    out = np.empty((hp, n), dtype=F32)
    out[0::2] = y.real
    out[1::2] = y.imag
    return out[:h]


# ----------------------------------------------------------------------------------------------
# median by bitwise bisection on monotone keys
# ----------------------------------------------------------------------------------------------
def float_key(v):
    b = v.astype(F32).view(np.uint32)
    return np.where(b & np.uint32(0x80000000), ~b, b | np.uint32(0x80000000)).astype(np.uint32)


def key_float(k):
    k = np.asarray(k, dtype=np.uint32)
    b = np.where(k & np.uint32(0x80000000), k & np.uint32(0x7FFFFFFF), ~k).astype(np.uint32)
    return b.view(F32)


def kth_smallest_key(keys, k):
    """Largest T with #{key < T} <= k  ==  k-th smallest key (0-based)."""
    res = np.uint32(0)
    for bit in range(31, -1, -1):
        trial = np.uint32(res | np.uint32(1 << bit))
        if int((keys < trial).sum()) <= k:
            res = trial
    return res


def row_median_bisect(row):
    n = row.shape[0]
    keys = float_key(row)
    k1 = (n - 1) // 2
    t1 = kth_smallest_key(keys, k1)
    v1 = key_float(np.array([t1]))[0]
    if n % 2 == 1:
        return v1
    c_le = int((keys <= t1).sum())
    if c_le > k1 + 1:
        v2 = v1
    else:
        v2 = key_float(np.array([keys[keys > t1].min()]))[0]
    return F32(0.5) * (v1 + v2)


# ----------------------------------------------------------------------------------------------
# Otsu in the GPU regime: float32 histogram, float64 class statistics
# ----------------------------------------------------------------------------------------------
def otsu_f32(q):
    q = q.astype(F32)
    qmin, qmax = q.min(), q.max()
    if qmin == qmax:
        return float(qmin)
    counts, edges = orc.histogram256(q)
    return float(orc.otsu_from_histogram(counts, edges.astype(np.float64)))


# ----------------------------------------------------------------------------------------------
# the whole path in the GPU formulation
# ----------------------------------------------------------------------------------------------
def dwt_aa_da(x):
    a0 = orc.dwt_axis(x, orc.DB3_DEC_LO, 0)
    d0 = orc.dwt_axis(x, orc.DB3_DEC_HI, 0)
    return orc.dwt_axis(a0, orc.DB3_DEC_LO, 1), orc.dwt_axis(d0, orc.DB3_DEC_LO, 1)


def idwt_delta(c, delta, out_h, out_w):
    """c_{l-1} = IDWT2(aa = c_l, da = Delta_l, ad = dd = 0), trimmed to (out_h, out_w)."""
    zero = np.zeros_like(delta)
    a0 = orc.idwt_axis(c, zero, orc.DB3_REC_LO, orc.DB3_REC_HI, 1)
    d0 = orc.idwt_axis(delta, zero, orc.DB3_REC_LO, orc.DB3_REC_HI, 1)
    full = orc.idwt_axis(a0, d0, orc.DB3_REC_LO, orc.DB3_REC_HI, 0)
    return full[:out_h, :out_w]


def destripe_plane_model(image, level, sigma, max_threshold, use_bisect_median=False):
    """float32 GPU-formulation result for one plane; returns (out f32[H', W'], per-level dict list)."""
    image = np.asarray(image)
    h0, w0 = image.shape
    levels = orc.resolve_level(image.shape, level)
    x = np.log(F32(1.0) + image.astype(F32)).astype(F32)
    dims = [(h0, w0)]
    das = []
    a = x
    for _ in range(levels):
        a, da = dwt_aa_da(a)
        das.append(da)
        dims.append(da.shape)
    width_fraction = sigma / min(h0, w0)
    info = []
    c = None
    for lv in range(levels, 0, -1):
        ch = das[lv - 1]
        hh, ww = ch.shape
        otsu = otsu_f32(ch * ch)
        thr = F32(min(max_threshold, np.sqrt(otsu)))
        mask = np.abs(ch) > thr
        bg = np.where(mask, F32(0.0), ch)
        if use_bisect_median:
            med = np.array([row_median_bisect(r) for r in bg], dtype=F32)
        else:
            med = np.median(bg, axis=-1).astype(F32)
        inp = np.where(mask, med[:, None], bg).astype(F32)
        lp = lowpass_rows(inp, hh * width_fraction)
        delta = np.where(mask, F32(0.0), -lp).astype(F32)
        info.append({"otsu": otsu, "threshold": float(thr), "mask_count": int(mask.sum()), "median": med})
        if c is None:
            c = np.zeros_like(delta)
        out_h, out_w = dims[lv - 1]
        if lv == 1:
            out_h += out_h & 1
            out_w += out_w & 1
        c = idwt_delta(c[:hh, :ww], delta, out_h, out_w)
    if levels == 0:
        return image.astype(F32) + F32(2.0), info
    hp, wp = c.shape
    # an odd plane grows by one row/col (waverec2): the extra sample of the reconstructed log image
    # is the half-sample symmetric extension, i.e. the replicated edge pixel (checked vs pywt)
    yy = np.minimum(np.arange(hp), h0 - 1)
    xx = np.minimum(np.arange(wp), w0 - 1)
    xin = image.astype(F32)[np.ix_(yy, xx)]
    out = (F32(1.0) + xin) * np.exp(c) + F32(1.0)
    return out, info
