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

#ifndef DSX_SCALAR_FMA
#define DSX_SCALAR_FMA 1  // the multiply-adds of pk_fma / the complex helpers as scalar v_fma_f32 (see __graft_entry__.build)
#endif
#ifndef DSX_SCALAR_CMUL
#define DSX_SCALAR_CMUL DSX_SCALAR_FMA
#endif
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DSX_HD __host__ __device__ __forceinline__
// (re, im) as a 2-vector: element-wise arithmetic maps onto the packed FP32 instructions
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).  Written with scalar members, hipcc's SLP vectoriser
// pairs unrelated values and shuffles them together with v_mov (20 % of the FFT instructions).
typedef float dsx_c32 __attribute__((ext_vector_type(2)));
DSX_HD dsx_c32 dsx_mk(float re, float im) { return dsx_c32{re, im}; }
DSX_HD dsx_c32 dsx_add(dsx_c32 a, dsx_c32 b) { return a + b; }
DSX_HD dsx_c32 dsx_sub(dsx_c32 a, dsx_c32 b) { return a - b; }
// a * (-i)
DSX_HD dsx_c32 dsx_mul_mi(dsx_c32 a) { return dsx_c32{a.y, -a.x}; }
// (a - b) * (-i) = (a.y - b.y, b.x - a.x) in one packed add (half selection and signs on the operands)
DSX_HD dsx_c32 dsx_sub_mi(dsx_c32 a, dsx_c32 b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DSX_NO_ASM_CMUL) && !DSX_SCALAR_FMA
  dsx_c32 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
#else
  return dsx_c32{a.y - b.y, b.x - a.x};
#endif
}
// complex product in two packed instructions: the half selection (op_sel) and the sign (neg_lo) ride on the
// operands, where the compiler builds {-b.y, b.x} with a v_xor and a v_mov first
//   t = (-a.y b.y, a.y b.x);   d = (a.x b.x + t.lo, a.x b.y + t.hi)
DSX_HD dsx_c32 dsx_mul(dsx_c32 a, dsx_c32 b) {
#if defined(__HIP_DEVICE_COMPILE__) && DSX_SCALAR_CMUL
  // the same four roundings as the packed pair below, as two multiplies and two multiply-adds
  const float tx = -(a.y * b.y), ty = a.y * b.x;
  return dsx_c32{__builtin_fmaf(a.x, b.x, tx), __builtin_fmaf(a.x, b.y, ty)};
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(DSX_NO_ASM_CMUL)
  dsx_c32 t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(t) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(t));
  return d;
#else
  return a.xx * b + a.yy * dsx_c32{-b.y, b.x};
#endif
}
DSX_HD dsx_c32 dsx_scale(dsx_c32 a, float s) { return a * s; }
// a * s + b
#if DSX_SCALAR_FMA
DSX_HD dsx_c32 dsx_fma_s(dsx_c32 a, float s, dsx_c32 b) { return dsx_c32{__builtin_fmaf(a.x, s, b.x), __builtin_fmaf(a.y, s, b.y)}; }
#else
DSX_HD dsx_c32 dsx_fma_s(dsx_c32 a, float s, dsx_c32 b) { return a * s + b; }
#endif
#else
#define DSX_HD inline
struct dsx_c32 {
  float x, y;
};
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
DSX_HD dsx_c32 dsx_sub_mi(dsx_c32 a, dsx_c32 b) { return dsx_mk(a.y - b.y, b.x - a.x); }
DSX_HD dsx_c32 dsx_scale(dsx_c32 a, float s) { return dsx_mk(a.x * s, a.y * s); }
// a * s + b
DSX_HD dsx_c32 dsx_fma_s(dsx_c32 a, float s, dsx_c32 b) { return dsx_mk(a.x * s + b.x, a.y * s + b.y); }
#endif

// Index products: every factor is below 2^12 (M <= 4096), so the 24-bit integer multiplier applies
// (v_mul_u32_u24 / v_mad_u32_u24: full rate; the 32-bit v_mul_lo_u32 and v_mad_u64_u32 the compiler picks
// without range knowledge run at a quarter of it and were 4 % of the row filter's instructions).
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define DSX_IMUL(a, b) __mul24((a), (b))
#else
#define DSX_IMUL(a, b) ((a) * (b))
#endif

// q = b mod s, via a float reciprocal (exact for b, s < 2^20: the quotient is never within
// 0.5 / s of an integer, and the relative error of the product is < 2^-22).
DSX_HD int dsx_mod_s(int b, int s, float inv_s) {
  int p = (int)(((float)b + 0.5f) * inv_s);
  return b - DSX_IMUL(p, s);
}

// Output-centric decode for the generic (any radix) pass: o = q + s (R p + k).
DSX_HD void dsx_generic_decode(int o, int s, float inv_s, int R, float inv_R, int* q, int* k,
                               int* p) {
  int t = (int)(((float)o + 0.5f) * inv_s);  // o / s
  *q = o - DSX_IMUL(t, s);
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
    dsx_c32 dm = dsx_sub_mi(v[1], v[2]);  // -i (x1 - x2)
    dsx_c32 u = dsx_fma_s(t, -0.5f, v[0]);
    v[0] = dsx_add(v[0], t);
    v[1] = dsx_fma_s(dm, S, u);   // u - i sin(2pi/3) (x1 - x2)
    v[2] = dsx_fma_s(dm, -S, u);
  }
};

template <>
struct dsx_bfly<4> {
  DSX_HD static void run(dsx_c32* v) {
    dsx_c32 t0 = dsx_add(v[0], v[2]);
    dsx_c32 t1 = dsx_sub(v[0], v[2]);
    dsx_c32 t2 = dsx_add(v[1], v[3]);
    dsx_c32 t3 = dsx_sub_mi(v[1], v[3]);
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
    v[0] = dsx_add(x0, dsx_add(a1, a2));
    dsx_c32 p1 = dsx_fma_s(a2, C2, dsx_fma_s(a1, C1, x0));
    dsx_c32 p2 = dsx_fma_s(a2, C1, dsx_fma_s(a1, C2, x0));
    // -i (S1 b1 + S2 b2) and -i (S2 b1 - S1 b2)
    dsx_c32 q1 = dsx_mul_mi(dsx_fma_s(b2, S2, dsx_scale(b1, S1)));
    dsx_c32 q2 = dsx_mul_mi(dsx_fma_s(b2, -S1, dsx_scale(b1, S2)));
    v[1] = dsx_add(p1, q1);
    v[4] = dsx_sub(p1, q1);
    v[2] = dsx_add(p2, q2);
    v[3] = dsx_sub(p2, q2);
  }
};

// ---- odd-prime register butterflies (P = 7, 11, 13, 17, 19): symmetric form, (P-1)^2 real FMAs -----
// X[k], X[P-k] = A -/+ iB,  A = x0 + sum_j cos(2 pi jk/P) (x_j + x_{P-j}),  B = sum_j sin(2 pi jk/P) (x_j - x_{P-j})
DSX_HD constexpr float dsx_root_cos(int P, int j) {
  switch (P) {
    case 6: {
      constexpr float t[6] = {1.0f, 0.50000000000000011f, -0.49999999999999978f, -1.0f, -0.50000000000000044f, 0.50000000000000011f};
      return t[j];
    }
    case 8: {
      constexpr float t[8] = {1.0f, 0.70710678118654757f, 6.123233995736766e-17f, -0.70710678118654746f, -1.0f, -0.70710678118654768f, -1.8369701987210297e-16f, 0.70710678118654735f};
      return t[j];
    }
    case 9: {
      constexpr float t[9] = {1.0f, 0.76604444311897801f, 0.17364817766693041f, -0.49999999999999978f, -0.93969262078590832f, -0.93969262078590843f, -0.50000000000000044f, 0.17364817766692997f, 0.76604444311897779f};
      return t[j];
    }
    case 10: {
      constexpr float t[10] = {1.0f, 0.80901699437494745f, 0.30901699437494745f, -0.30901699437494734f, -0.80901699437494734f, -1.0f, -0.80901699437494756f, -0.30901699437494756f, 0.30901699437494723f, 0.80901699437494734f};
      return t[j];
    }
    case 12: {
      constexpr float t[12] = {1.0f, 0.86602540378443871f, 0.50000000000000011f, 6.123233995736766e-17f, -0.49999999999999978f, -0.86602540378443871f, -1.0f, -0.86602540378443882f, -0.50000000000000044f, -1.8369701987210297e-16f, 0.50000000000000011f, 0.86602540378443837f};
      return t[j];
    }
    case 15: {
      constexpr float t[15] = {1.0f, 0.91354545764260087f, 0.66913060635885824f, 0.30901699437494745f, -0.10452846326765333f, -0.49999999999999978f, -0.80901699437494734f, -0.97814760073380569f, -0.97814760073380569f, -0.80901699437494756f, -0.50000000000000044f, -0.10452846326765423f, 0.30901699437494723f, 0.66913060635885846f, 0.91354545764260098f};
      return t[j];
    }
    case 16: {
      constexpr float t[16] = {1.0f, 0.92387953251128674f, 0.70710678118654757f, 0.38268343236508984f, 6.123233995736766e-17f, -0.38268343236508973f, -0.70710678118654746f, -0.92387953251128674f, -1.0f, -0.92387953251128685f, -0.70710678118654768f, -0.38268343236509034f, -1.8369701987210297e-16f, 0.38268343236509f, 0.70710678118654735f, 0.92387953251128652f};
      return t[j];
    }
    case 20: {
      constexpr float t[20] = {1.0f, 0.95105651629515353f, 0.80901699437494745f, 0.58778525229247314f, 0.30901699437494745f, 6.123233995736766e-17f, -0.30901699437494734f, -0.58778525229247303f, -0.80901699437494734f, -0.95105651629515353f, -1.0f, -0.95105651629515375f, -0.80901699437494756f, -0.58778525229247325f, -0.30901699437494756f, -1.8369701987210297e-16f, 0.30901699437494723f, 0.58778525229247292f, 0.80901699437494734f, 0.95105651629515353f};
      return t[j];
    }
    case 25: {
      constexpr float t[25] = {1.0f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f, 0.53582679497899655f, 0.30901699437494745f, 0.062790519529313527f, -0.1873813145857246f, -0.42577929156507272f, -0.63742398974868975f, -0.80901699437494734f, -0.92977648588825135f, -0.99211470131447776f, -0.99211470131447788f, -0.92977648588825146f, -0.80901699437494778f, -0.63742398974868952f, -0.42577929156507216f, -0.18738131458572463f, 0.062790519529312833f, 0.30901699437494723f, 0.53582679497899677f, 0.72896862742141122f, 0.87630668004386314f, 0.96858316112863097f};
      return t[j];
    }
    case 7: {
      constexpr float t[7] = {1.0f, 0.62348980185873359f, -0.22252093395631434f, -0.90096886790241903f, -0.90096886790241915f, -0.22252093395631459f, 0.62348980185873337f};
      return t[j];
    }
    case 11: {
      constexpr float t[11] = {1.0f, 0.84125353283118121f, 0.41541501300188644f, -0.142314838273285f, -0.65486073394528499f, -0.95949297361449737f, -0.95949297361449748f, -0.65486073394528521f, -0.14231483827328523f, 0.41541501300188605f, 0.84125353283118121f};
      return t[j];
    }
    case 13: {
      constexpr float t[13] = {1.0f, 0.88545602565320991f, 0.56806474673115592f, 0.12053668025532301f, -0.35460488704253545f, -0.74851074817110119f, -0.97094181742605201f, -0.97094181742605212f, -0.7485107481711013f, -0.3546048870425359f, 0.1205366802553232f, 0.56806474673115481f, 0.88545602565321002f};
      return t[j];
    }
    case 17: {
      constexpr float t[17] = {1.0f, 0.93247222940435581f, 0.73900891722065909f, 0.44573835577653831f, 0.092268359463302016f, -0.27366299007208289f, -0.60263463637925629f, -0.85021713572961399f, -0.98297309968390179f, -0.98297309968390179f, -0.8502171357296141f, -0.60263463637925718f, -0.27366299007208311f, 0.092268359463302432f, 0.4457383557765377f, 0.73900891722065853f, 0.93247222940435581f};
      return t[j];
    }
    case 19: {
      constexpr float t[19] = {1.0f, 0.94581724170063464f, 0.78914050939639357f, 0.54694815812242692f, 0.24548548714079924f, -0.082579345472332269f, -0.40169542465296942f, -0.67728157162574087f, -0.87947375120648896f, -0.98636130340272232f, -0.98636130340272243f, -0.8794737512064893f, -0.6772815716257411f, -0.40169542465296904f, -0.082579345472332741f, 0.24548548714079879f, 0.54694815812242659f, 0.78914050939639391f, 0.94581724170063464f};
      return t[j];
    }
    default: return 0.f;
  }
}
DSX_HD constexpr float dsx_root_sin(int P, int j) {
  switch (P) {
    case 6: {
      constexpr float t[6] = {0.0f, 0.8660254037844386f, 0.86602540378443871f, 1.2246467991473532e-16f, -0.86602540378443837f, -0.8660254037844386f};
      return t[j];
    }
    case 8: {
      constexpr float t[8] = {0.0f, 0.70710678118654746f, 1.0f, 0.70710678118654757f, 1.2246467991473532e-16f, -0.70710678118654746f, -1.0f, -0.70710678118654768f};
      return t[j];
    }
    case 9: {
      constexpr float t[9] = {0.0f, 0.64278760968653925f, 0.98480775301220802f, 0.86602540378443871f, 0.34202014332566888f, -0.34202014332566866f, -0.86602540378443837f, -0.98480775301220813f, -0.64278760968653958f};
      return t[j];
    }
    case 10: {
      constexpr float t[10] = {0.0f, 0.58778525229247314f, 0.95105651629515353f, 0.95105651629515364f, 0.58778525229247325f, 1.2246467991473532e-16f, -0.58778525229247303f, -0.95105651629515353f, -0.95105651629515364f, -0.58778525229247336f};
      return t[j];
    }
    case 12: {
      constexpr float t[12] = {0.0f, 0.49999999999999994f, 0.8660254037844386f, 1.0f, 0.86602540378443871f, 0.49999999999999994f, 1.2246467991473532e-16f, -0.49999999999999972f, -0.86602540378443837f, -1.0f, -0.8660254037844386f, -0.50000000000000044f};
      return t[j];
    }
    case 15: {
      constexpr float t[15] = {0.0f, 0.40673664307580015f, 0.74314482547739413f, 0.95105651629515353f, 0.9945218953682734f, 0.86602540378443871f, 0.58778525229247325f, 0.20791169081775931f, -0.20791169081775907f, -0.58778525229247303f, -0.86602540378443837f, -0.99452189536827329f, -0.95105651629515364f, -0.74314482547739402f, -0.40673664307580015f};
      return t[j];
    }
    case 16: {
      constexpr float t[16] = {0.0f, 0.38268343236508978f, 0.70710678118654746f, 0.92387953251128674f, 1.0f, 0.92387953251128674f, 0.70710678118654757f, 0.38268343236508989f, 1.2246467991473532e-16f, -0.38268343236508967f, -0.70710678118654746f, -0.92387953251128652f, -1.0f, -0.92387953251128663f, -0.70710678118654768f, -0.38268343236509039f};
      return t[j];
    }
    case 20: {
      constexpr float t[20] = {0.0f, 0.3090169943749474f, 0.58778525229247314f, 0.80901699437494745f, 0.95105651629515353f, 1.0f, 0.95105651629515364f, 0.80901699437494745f, 0.58778525229247325f, 0.30901699437494751f, 1.2246467991473532e-16f, -0.3090169943749469f, -0.58778525229247303f, -0.80901699437494734f, -0.95105651629515353f, -1.0f, -0.95105651629515364f, -0.80901699437494756f, -0.58778525229247336f, -0.30901699437494762f};
      return t[j];
    }
    case 25: {
      constexpr float t[25] = {0.0f, 0.24868988716485479f, 0.48175367410171532f, 0.68454710592868862f, 0.84432792550201508f, 0.95105651629515353f, 0.99802672842827156f, 0.98228725072868872f, 0.90482705246601947f, 0.77051324277578925f, 0.58778525229247325f, 0.36812455268467814f, 0.12533323356430454f, -0.12533323356430429f, -0.36812455268467792f, -0.58778525229247269f, -0.77051324277578936f, -0.9048270524660198f, -0.98228725072868872f, -0.99802672842827156f, -0.95105651629515364f, -0.84432792550201496f, -0.68454710592868895f, -0.4817536741017161f, -0.24868988716485535f};
      return t[j];
    }
    case 7: {
      constexpr float t[7] = {0.0f, 0.7818314824680298f, 0.97492791218182362f, 0.43388373911755823f, -0.43388373911755801f, -0.97492791218182362f, -0.78183148246802991f};
      return t[j];
    }
    case 11: {
      constexpr float t[11] = {0.0f, 0.54064081745559756f, 0.90963199535451833f, 0.9898214418809328f, 0.75574957435425827f, 0.28173255684142967f, -0.28173255684142939f, -0.75574957435425816f, -0.98982144188093268f, -0.90963199535451855f, -0.54064081745559744f};
      return t[j];
    }
    case 13: {
      constexpr float t[13] = {0.0f, 0.46472317204376851f, 0.82298386589365635f, 0.99270887409805397f, 0.93501624268541483f, 0.66312265824079519f, 0.23931566428755768f, -0.23931566428755743f, -0.66312265824079497f, -0.93501624268541472f, -0.99270887409805397f, -0.82298386589365702f, -0.4647231720437684f};
      return t[j];
    }
    case 17: {
      constexpr float t[17] = {0.0f, 0.36124166618715292f, 0.67369564364655721f, 0.89516329135506234f, 0.99573417629503447f, 0.96182564317281904f, 0.7980172272802396f, 0.52643216287735606f, 0.18374951781657037f, -0.18374951781657012f, -0.52643216287735584f, -0.79801722728023894f, -0.96182564317281904f, -0.99573417629503447f, -0.89516329135506256f, -0.67369564364655776f, -0.36124166618715303f};
      return t[j];
    }
    case 19: {
      constexpr float t[19] = {0.0f, 0.32469946920468346f, 0.61421271268966782f, 0.83716647826252855f, 0.96940026593933037f, 0.99658449300666985f, 0.9157733266550574f, 0.73572391067313181f, 0.47594739303707367f, 0.16459459028073403f, -0.16459459028073378f, -0.47594739303707312f, -0.73572391067313159f, -0.91577332665505762f, -0.99658449300666985f, -0.96940026593933049f, -0.83716647826252877f, -0.61421271268966737f, -0.32469946920468373f};
      return t[j];
    }
    default: return 0.f;
  }
}

template <int P>
DSX_HD void dsx_bfly_odd(dsx_c32* v) {
  constexpr int HP = (P - 1) / 2;
  dsx_c32 sp[HP], sm[HP];
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int j = 1; j <= HP; ++j) {
    sp[j - 1] = dsx_add(v[j], v[P - j]);
    sm[j - 1] = dsx_sub_mi(v[j], v[P - j]);  // -i (x_j - x_{P-j})
  }
  const dsx_c32 x0 = v[0];
  dsx_c32 dc = x0;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int j = 0; j < HP; ++j) dc = dsx_add(dc, sp[j]);
  v[0] = dc;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 1; k <= HP; ++k) {
    dsx_c32 A = x0, B = dsx_mk(0.f, 0.f);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 1; j <= HP; ++j) {
      A = dsx_fma_s(sp[j - 1], dsx_root_cos(P, (j * k) % P), A);
      B = dsx_fma_s(sm[j - 1], dsx_root_sin(P, (j * k) % P), B);
    }
    v[k] = dsx_add(A, B);      // A - iB'  (B = -i B')
    v[P - k] = dsx_sub(A, B);  // A + iB'
  }
}
template <>
struct dsx_bfly<7> {
  DSX_HD static void run(dsx_c32* v) { dsx_bfly_odd<7>(v); }
};
template <>
struct dsx_bfly<11> {
  DSX_HD static void run(dsx_c32* v) { dsx_bfly_odd<11>(v); }
};
template <>
struct dsx_bfly<13> {
  DSX_HD static void run(dsx_c32* v) { dsx_bfly_odd<13>(v); }
};
template <>
struct dsx_bfly<17> {
  DSX_HD static void run(dsx_c32* v) { dsx_bfly_odd<17>(v); }
};
template <>
struct dsx_bfly<19> {
  DSX_HD static void run(dsx_c32* v) { dsx_bfly_odd<19>(v); }
};


// ---- composite register butterflies R = R1 * R2 (Cooley-Tukey inside the registers) ---------------
// n = R2 n1 + n2, k = k1 + R1 k2:
//   X[k1 + R1 k2] = sum_n2 W_R^{n2 k1} (sum_n1 x[R2 n1 + n2] W_R1^{n1 k1}) W_R2^{n2 k2}
// computed in place; X[k] ends up in v[R2 (k % R1) + k / R1]  (dsx_comp_pos).
template <int R>
struct dsx_comp {
  static constexpr int R1 = 0, R2 = 0;
};
#define DSX_COMP(R, A, B)            \
  template <>                        \
  struct dsx_comp<R> {               \
    static constexpr int R1 = A, R2 = B; \
  };
DSX_COMP(6, 2, 3)
DSX_COMP(8, 2, 4)
DSX_COMP(9, 3, 3)
DSX_COMP(10, 2, 5)
DSX_COMP(12, 3, 4)
DSX_COMP(15, 3, 5)
DSX_COMP(16, 4, 4)
DSX_COMP(20, 4, 5)
DSX_COMP(25, 5, 5)
#undef DSX_COMP

template <int R>
DSX_HD constexpr int dsx_comp_pos(int k) {
  return dsx_comp<R>::R2 * (k % dsx_comp<R>::R1) + k / dsx_comp<R>::R1;
}

template <int R>
DSX_HD void dsx_bfly_composite(dsx_c32* v) {
  constexpr int R1 = dsx_comp<R>::R1, R2 = dsx_comp<R>::R2;
  // stage 1: R2 butterflies of radix R1 over n1 (stride R2), then the inner twiddles W_R^{n2 k1}
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int n2 = 0; n2 < R2; ++n2) {
    dsx_c32 t[R1];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int n1 = 0; n1 < R1; ++n1) t[n1] = v[R2 * n1 + n2];
    dsx_bfly<R1>::run(t);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k1 = 0; k1 < R1; ++k1) {
      const int e = (n2 * k1) % R;
      if (e == 0) {
        v[R2 * k1 + n2] = t[k1];
      } else {
        const float c = dsx_root_cos(R, e), sn = -dsx_root_sin(R, e);  // exp(-2 pi i e / R)
        v[R2 * k1 + n2] = dsx_mul(t[k1], dsx_mk(c, sn));
      }
    }
  }
  // stage 2: R1 butterflies of radix R2 over n2 (contiguous)
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k1 = 0; k1 < R1; ++k1) dsx_bfly<R2>::run(v + R2 * k1);
}

// ---- where element i of the row buffer / the twiddle table lives in LDS ---------------------------------------------
// dsx_idx_plain: at i.  dsx_idx_swz (the power-of-two plans: 2 048 = 16 * 16 * 8, 1 024 = 16 * 8 * 8): the low four bits of
// the index are XORed with bits 4 ... 7.  A Stockham pass of a power-of-two radix scatters its outputs at a power-of-two
// stride -- the first pass of the 2 048-point plan at 16 complex values = 128 bytes: all 64 lanes of a store instruction
// in the same two of the 32 LDS banks (rocprofv3: two thirds of that kernel's LDS cycles were bank conflicts, the LDS pipe
// busy 55 % of the time).  With the swizzle consecutive sixteens of elements are permuted differently, so that sources
// b + j M / R, destinations R b - (R - 1) q + s k and twiddles (b - q) k of 64 consecutive butterflies b all spread over
// the banks evenly; no padding, the buffer keeps its size (lengths are multiples of 16).
struct dsx_idx_plain {
  DSX_HD static int at(int i) { return i; }
};
struct dsx_idx_swz {
  DSX_HD static int at(int i) { return i ^ ((i >> 4) & 15); }
};

// ---- per-butterfly pieces of a pass (used verbatim by k_rowfilter and by the host unit test) ----
// JK (odd-prime radices only): input pairs (x_j, x_{R-j}) with j > JK are known to be zero (band-limited
// spectrum in the first pass of the inverse transform) and are neither loaded nor accumulated.
template <int R, int JK = (R - 1) / 2, class IX = dsx_idx_plain>
DSX_HD void dsx_bfly_load(const dsx_c32* buf, int b, int nb, dsx_c32* v) {
  if constexpr (JK < (R - 1) / 2) {
    v[0] = buf[IX::at(b)];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 1; j <= JK; ++j) {
      v[j] = buf[IX::at(b + DSX_IMUL(j, nb))];
      v[R - j] = buf[IX::at(b + DSX_IMUL(R - j, nb))];
    }
  } else {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < R; ++j) v[j] = buf[IX::at(b + DSX_IMUL(j, nb))];
  }
}

// butterfly + twiddles + autosort scatter of butterfly b (sub-transform stride s)
// unit_tw: last pass of a transform (s * R == M, so p == 0 and every twiddle is 1)
// KO (odd-prime radices only): only the outputs X[0] and the pairs (X[k], X[R-k]), k <= KO, are computed and stored --
// the LAST pass of a forward transform whose consumer reads a band |bin| <= kcut only (outputs q + s k: the pairs
// beyond KO = (kcut + s - 1) / s lie outside the band); the other destinations keep what the pass before left there.
template <int R, int JK = (R - 1) / 2, int KO = (R - 1) / 2, class IX = dsx_idx_plain>
DSX_HD void dsx_bfly_store(dsx_c32* buf, const dsx_c32* tw, int b, int s, float inv_s, dsx_c32* v,
                           bool unit_tw = false) {
  const int q = (s == 1) ? 0 : dsx_mod_s(b, s, inv_s);
  const int ps = b - q;
  const int dst = DSX_IMUL(R, b) - DSX_IMUL(R - 1, q);
  if constexpr (R == 7 || R == 11 || R == 13 || R == 17 || R == 19) {
    // odd prime: outputs are produced and scattered pair by pair (X[k], X[R-k]) so that only the
    // R inputs (folded in place into sums / differences) stay live in registers
    constexpr int HP = (R - 1) / 2;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 1; j <= JK; ++j) {
      const dsx_c32 x = v[j], y = v[R - j];
      v[j] = dsx_add(x, y);
      v[R - j] = dsx_sub_mi(x, y);  // -i (x_j - x_{R-j})
    }
    const dsx_c32 x0 = v[0];
    dsx_c32 dc = x0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 1; j <= JK; ++j) dc = dsx_add(dc, v[j]);
    buf[IX::at(dst)] = dc;
    static_assert(KO >= 1 && KO <= HP, "output pairs of an odd-prime butterfly");
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 1; k <= KO; ++k) {
      dsx_c32 A = x0, B = dsx_mk(0.f, 0.f);
#if defined(__HIPCC__)
#pragma unroll
#endif
      for (int j = 1; j <= JK; ++j) {
        A = dsx_fma_s(v[j], dsx_root_cos(R, (j * k) % R), A);
        B = dsx_fma_s(v[R - j], dsx_root_sin(R, (j * k) % R), B);
      }
      if (unit_tw) {
        buf[IX::at(dst + DSX_IMUL(s, k))] = dsx_add(A, B);        // X[k]   = A - i B'   (B = -i B')
        buf[IX::at(dst + DSX_IMUL(s, R - k))] = dsx_sub(A, B);  // X[R-k] = A + i B'
      } else {
        buf[IX::at(dst + DSX_IMUL(s, k))] = dsx_mul(dsx_add(A, B), tw[IX::at(DSX_IMUL(ps, k))]);
        buf[IX::at(dst + DSX_IMUL(s, R - k))] = dsx_mul(dsx_sub(A, B), tw[IX::at(DSX_IMUL(ps, R - k))]);
      }
    }
  } else if constexpr (dsx_comp<R>::R1 != 0) {
    dsx_bfly_composite<R>(v);
    buf[IX::at(dst)] = v[0];
    if (unit_tw) {
#if defined(__HIPCC__)
#pragma unroll
#endif
      for (int k = 1; k < R; ++k) buf[IX::at(dst + DSX_IMUL(s, k))] = v[dsx_comp_pos<R>(k)];
    } else {
#if defined(__HIPCC__)
#pragma unroll
#endif
      for (int k = 1; k < R; ++k) buf[IX::at(dst + DSX_IMUL(s, k))] = dsx_mul(v[dsx_comp_pos<R>(k)], tw[IX::at(DSX_IMUL(ps, k))]);
    }
  } else {
    dsx_bfly<R>::run(v);
    buf[IX::at(dst)] = v[0];
    if (unit_tw) {
#if defined(__HIPCC__)
#pragma unroll
#endif
      for (int k = 1; k < R; ++k) buf[IX::at(dst + DSX_IMUL(s, k))] = v[k];
    } else {
#if defined(__HIPCC__)
#pragma unroll
#endif
      for (int k = 1; k < R; ++k) buf[IX::at(dst + DSX_IMUL(s, k))] = dsx_mul(v[k], tw[IX::at(DSX_IMUL(ps, k))]);
    }
  }
}

// ---- the passes of the power-of-two plans with the dsx_idx_swz addressing written out -------------------------------
// sw(i) = i ^ ((i >> 4) & 15) only touches the low four index bits, and for the sources b + j M / R and destinations
// R b - (R - 1) q + S k of ONE butterfly the XOR term takes one or two values: every access becomes "swizzled base +
// compile-time offset" (an LDS instruction's immediate) instead of a shift, a mask and an XOR per access -- the generic
// policy cost the 2 048-point row filter + 52 % vector instructions.  R is 16 or 8, M and S powers of two, M / R a
// multiple of 128.  Same butterflies, same twiddle values and products as dsx_bfly_load / dsx_bfly_store with dsx_idx_swz.
// bases of the accesses hi + j STEP (+ XOR term in the low four bits): the term (c + j STEP / 16) & 15 repeats with period
// 256 / (largest power of two dividing STEP, at most 256)
template <int STEP>
struct dsx_swz_period {
  static_assert(STEP % 16 == 0, "accesses of a butterfly must keep their low four index bits");
  static constexpr int LOW = (STEP & -STEP) > 256 ? 256 : (STEP & -STEP);
  static constexpr int P = 256 / LOW;  // 1 (STEP = 0 mod 256), 2 (128), 4 (64), ...
};
template <int R, int M>
DSX_HD void dsx_swz_load(const dsx_c32* buf, int b, dsx_c32* v) {
  constexpr int NB = M / R;
  constexpr int P = dsx_swz_period<NB>::P;
  const int r = b & 15, c = b >> 4, hi = b & ~15;
  int base[P];
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int m = 0; m < P; ++m) base[m] = hi | (r ^ ((c + m * (NB / 16)) & 15));
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int j = 0; j < R; ++j) v[j] = buf[base[j % P] + j * NB];
}

template <int R, int M, int S>
DSX_HD void dsx_swz_store(dsx_c32* buf, const dsx_c32* tw, int b, dsx_c32* v) {
  static_assert(dsx_comp<R>::R1 != 0 && (R == 16 || R == 8), "composite power-of-two radices only");
  dsx_bfly_composite<R>(v);
  if constexpr (S == 1) {
    // first pass: q = 0, destinations R b + k, twiddles tw[b k]
    static_assert(R == 16, "R b must be a multiple of 16");
    const int r = b & 15, base = 16 * b;
    buf[base + r] = v[0];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 1; k < R; ++k) {
      const int t = DSX_IMUL(b, k);
      buf[base + (k ^ r)] = dsx_mul(v[dsx_comp_pos<R>(k)], tw[t ^ ((t >> 4) & 15)]);
    }
  } else if constexpr (S * R == M) {
    // last pass: q = b (b < S), destinations b + S k, unit twiddles
    constexpr int P = dsx_swz_period<S>::P;
    const int r = b & 15, c = b >> 4, hi = b & ~15;
    int base[P];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int m = 0; m < P; ++m) base[m] = hi | (r ^ ((c + m * (S / 16)) & 15));
    buf[base[0]] = v[0];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 1; k < R; ++k) buf[base[k % P] + k * S] = v[dsx_comp_pos<R>(k)];
  } else {
    // middle pass, S = 16: q = b & 15, p = b >> 4; destinations 16 R p + q + 16 k, twiddles tw[16 p k]
    static_assert(S == 16, "middle pass of 16 * R * R'");
    const int q = b & 15, p = b >> 4;
    const int base = 16 * R * p;
    const int x0 = DSX_IMUL(R, p) & 15;  // ((dst >> 4) & 15) = (R p + k) & 15: k for R = 16, k ^ (8 (p & 1)) for R = 8 (k < 8)
    buf[base + (q ^ x0)] = v[0];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 1; k < R; ++k) {
      const int m = DSX_IMUL(p, k);
      buf[base + 16 * k + (q ^ ((x0 + k) & 15))] = dsx_mul(v[dsx_comp_pos<R>(k)], tw[16 * m + (m & 15)]);
    }
  }
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
    sum = dsx_add(sum, dsx_mul(x, w));
    src += sm;
    widx += wstep;
    if (widx >= M) widx -= M;
  }
  return dsx_mul(sum, tw[s * p * k]);
}

#endif  // DSX_FFT_CORE_H
