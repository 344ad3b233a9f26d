// dsx_fft_core.h -- butterflies and Stockham index algebra shared by the HIP row-filter kernel
// and a host-side unit test (tests/test_fft_core_host.py builds this header with g++).
//
// Forward transform convention: X[k] = sum_j x[j] exp(-2 pi i j k / N)  (scipy.fftpack.rfft,
// reference call site filtering.py:206).  The inverse is run through the same passes with
// re/im swapped on the way in and out.
//
// One Stockham autosort pass of radix R on a length-M buffer, sub-length n = M / s, m = n / R:
//     y[q + s (R p + k)] = w_n^{p k} * sum_j x[q + s (p + m j)] w_R^{j k},   p < m, q < s, k < R
// With the butterfly index b = p s + q (0 <= b < M / R) this becomes
//     sources       x[b + j (M / R)]
//     destinations  y[R b - (R - 1) q + s k]
//     twiddles      tw[(b - q) k],  tw[t] = exp(-2 pi i t / M)          ((b - q) k < M)
// so only q = b mod s is needed per butterfly (dsx_split_b).
#ifndef DSX_FFT_CORE_H
#define DSX_FFT_CORE_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DSX_HD __host__ __device__ __forceinline__
typedef float2 dsx_c32;
#else
#define DSX_HD inline
struct dsx_c32 {
  float x, y;
};
#endif

DSX_HD dsx_c32 dsx_mk(float re, float im) {
  dsx_c32 r;
  r.x = re;
  r.y = im;
  return r;
}
DSX_HD dsx_c32 dsx_add(dsx_c32 a, dsx_c32 b) { return dsx_mk(a.x + b.x, a.y + b.y); }
DSX_HD dsx_c32 dsx_sub(dsx_c32 a, dsx_c32 b) { return dsx_mk(a.x - b.x, a.y - b.y); }
DSX_HD dsx_c32 dsx_mul(dsx_c32 a, dsx_c32 b) {
  return dsx_mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * (-i)
DSX_HD dsx_c32 dsx_mul_mi(dsx_c32 a) { return dsx_mk(a.y, -a.x); }
DSX_HD dsx_c32 dsx_scale(dsx_c32 a, float s) { return dsx_mk(a.x * s, a.y * s); }

// q = b mod s, via a float reciprocal (exact for b, s < 2^20: the quotient is never within
// 0.5 / s of an integer, and the relative error of the product is < 2^-22).
DSX_HD int dsx_mod_s(int b, int s, float inv_s) {
  int p = (int)(((float)b + 0.5f) * inv_s);
  return b - p * s;
}

// Output-centric decode for the generic (any radix) pass: o = q + s (R p + k).
DSX_HD void dsx_generic_decode(int o, int s, float inv_s, int R, float inv_R, int* q, int* k,
                               int* p) {
  int t = (int)(((float)o + 0.5f) * inv_s);  // o / s
  *q = o - t * s;
  int pp = (int)(((float)t + 0.5f) * inv_R);  // t / R
  *k = t - pp * R;
  *p = pp;
}

template <int R>
struct dsx_bfly;

template <>
struct dsx_bfly<2> {
  DSX_HD static void run(dsx_c32* v) {
    dsx_c32 a = v[0], b = v[1];
    v[0] = dsx_add(a, b);
    v[1] = dsx_sub(a, b);
  }
};

template <>
struct dsx_bfly<3> {
  DSX_HD static void run(dsx_c32* v) {
    const float S = 0.86602540378443864676f;  // sin(2 pi / 3)
    dsx_c32 t = dsx_add(v[1], v[2]);
    dsx_c32 d = dsx_sub(v[1], v[2]);
    dsx_c32 u = dsx_mk(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
    dsx_c32 w = dsx_scale(dsx_mul_mi(d), S);  // -i sin(2pi/3) (x1 - x2)
    v[0] = dsx_add(v[0], t);
    v[1] = dsx_add(u, w);
    v[2] = dsx_sub(u, w);
  }
};

template <>
struct dsx_bfly<4> {
  DSX_HD static void run(dsx_c32* v) {
    dsx_c32 t0 = dsx_add(v[0], v[2]);
    dsx_c32 t1 = dsx_sub(v[0], v[2]);
    dsx_c32 t2 = dsx_add(v[1], v[3]);
    dsx_c32 t3 = dsx_mul_mi(dsx_sub(v[1], v[3]));
    v[0] = dsx_add(t0, t2);
    v[1] = dsx_add(t1, t3);
    v[2] = dsx_sub(t0, t2);
    v[3] = dsx_sub(t1, t3);
  }
};

template <>
struct dsx_bfly<5> {
  DSX_HD static void run(dsx_c32* v) {
    const float C1 = 0.30901699437494742410f;   // cos(2 pi / 5)
    const float C2 = -0.80901699437494742410f;  // cos(4 pi / 5)
    const float S1 = 0.95105651629515357212f;   // sin(2 pi / 5)
    const float S2 = 0.58778525229247312917f;   // sin(4 pi / 5)
    dsx_c32 a1 = dsx_add(v[1], v[4]), b1 = dsx_sub(v[1], v[4]);
    dsx_c32 a2 = dsx_add(v[2], v[3]), b2 = dsx_sub(v[2], v[3]);
    dsx_c32 x0 = v[0];
    v[0] = dsx_mk(x0.x + a1.x + a2.x, x0.y + a1.y + a2.y);
    dsx_c32 p1 = dsx_mk(x0.x + C1 * a1.x + C2 * a2.x, x0.y + C1 * a1.y + C2 * a2.y);
    dsx_c32 p2 = dsx_mk(x0.x + C2 * a1.x + C1 * a2.x, x0.y + C2 * a1.y + C1 * a2.y);
    // -i (S1 b1 + S2 b2) and -i (S2 b1 - S1 b2)
    dsx_c32 q1 = dsx_mul_mi(dsx_mk(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y));
    dsx_c32 q2 = dsx_mul_mi(dsx_mk(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y));
    v[1] = dsx_add(p1, q1);
    v[4] = dsx_sub(p1, q1);
    v[2] = dsx_add(p2, q2);
    v[3] = dsx_sub(p2, q2);
  }
};

// ---- per-butterfly pieces of a pass (used verbatim by k_rowfilter and by the host unit test) ----
template <int R>
DSX_HD void dsx_bfly_load(const dsx_c32* buf, int b, int nb, dsx_c32* v) {
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int j = 0; j < R; ++j) v[j] = buf[b + j * nb];
}

// butterfly + twiddles + autosort scatter of butterfly b (sub-transform stride s)
template <int R>
DSX_HD void dsx_bfly_store(dsx_c32* buf, const dsx_c32* tw, int b, int s, float inv_s, dsx_c32* v) {
  const int q = (s == 1) ? 0 : dsx_mod_s(b, s, inv_s);
  dsx_bfly<R>::run(v);
  const int ps = b - q;
  const int dst = R * b - (R - 1) * q;
  buf[dst] = v[0];
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 1; k < R; ++k) buf[dst + s * k] = dsx_mul(v[k], tw[ps * k]);
}

// one output element o of a generic radix-R pass (any R dividing M / s)
DSX_HD dsx_c32 dsx_generic_output(const dsx_c32* buf, const dsx_c32* tw, int o, int M, int s,
                                  float inv_s, int R) {
  const int m = M / (s * R);
  const float inv_R = 1.0f / (float)R;
  const int step = M / R;
  const int sm = s * m;
  int q, k, p;
  dsx_generic_decode(o, s, inv_s, R, inv_R, &q, &k, &p);
  int src = q + s * p;
  int widx = 0;
  const int wstep = k * step;
  dsx_c32 sum = dsx_mk(0.f, 0.f);
  for (int j = 0; j < R; ++j) {
    const dsx_c32 x = buf[src];
    const dsx_c32 w = tw[widx];
    sum.x += x.x * w.x - x.y * w.y;
    sum.y += x.x * w.y + x.y * w.x;
    src += sm;
    widx += wstep;
    if (widx >= M) widx -= M;
  }
  return dsx_mul(sum, tw[s * p * k]);
}

#endif  // DSX_FFT_CORE_H
