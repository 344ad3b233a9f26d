// dsx_wavelet.h -- analysis / synthesis levels for ANY even-length perfect-reconstruction filter bank.
//
// The reference takes the wavelet from its config dict and hands it to pywt.wavedec2 / pywt.waverec2
// (/root/reference/code/aind_smartspim_destripe/filtering.py:176, 221).  Production runs db3
// (run_capsule.py:374-390) and that is what the marching kernels of dsx_kernels.h are specialised for (6 taps in
// registers, fused levels 1 + 2).  Every other wavelet of PyWavelets' discrete families (haar, db, sym, coif,
// bior, rbio; 2 ... 102 taps) takes the two kernels below: same sub-bands, same workspace layout, same statistics,
// so the histogram, Otsu and row-filter kernels behind them do not know the difference.
//
//   k_fwd_gen<IN_KIND>   one analysis level: log(1 + x) (pixel planes), aa and da (= cH) only, fg/bg statistic
//                        (level 1) and min / max of cH^2.  A block owns a tile of 16 x 32 coefficients: the
//                        (32 + F - 2) x (64 + F - 2) input patch goes through LDS once, column pass, row pass.
//   k_inv_gen<MODE>      one synthesis level of the Delta pyramid (cV = cD = 0, so the row pass runs rec_lo only);
//                        the last level applies (1 + x) exp(c0) + 1, the shading correction and the cast.
//
// Conventions (PyWavelets mode 'symmetric'):
//   analysis   out[i] = sum_t f[t] x[refl(2 i + 1 - t)],            i < (N + F - 1) / 2
//   synthesis  out[n] = sum_j in[(n >> 1) + j] rec[F - 2 + (n & 1) - 2 j],  j < F / 2,  n < 2 M - F + 2
#ifndef DSX_WAVELET_H
#define DSX_WAVELET_H

#include "dsx_kernels.h"

namespace dsx {

constexpr int kMaxTaps = 104;  // coif17 has 102
constexpr int kGenTH = 16, kGenTW = 32;    // coefficients per block of k_fwd_gen
constexpr int kGenOH = 32, kGenOW = 64;    // results per block of k_inv_gen

struct GenFwdArgs {
  const void* in;  // IN_KIND 0 / 1: pixels [B][H][W]
  long long in_plane_stride;
  float* ws;
  long long ws_plane_stride;
  long long in_off;  // IN_KIND 2: aa_{l-1} inside a plane's workspace
  int H, W, ldin;
  long long aa_off, da_off;
  int h, w, ld, lda;
  unsigned* minmax;  // [B][L][2]
  int lvl, L;
  PlaneStats* stats;
  float fg_cutoff;
  int F;
  int shared;  // stack mode: plane 0's min / max slots for every plane (Fwd1Args::shared)
  float lo[kMaxTaps], hi[kMaxTaps];  // dec_lo, dec_hi
};

// LDS floats of k_fwd_gen for F taps
inline __host__ __device__ int gen_fwd_cols(int F) { return 2 * kGenTW + F - 2 + 1; }  // + 1: odd pitch
inline __host__ __device__ int gen_fwd_rows(int F) { return 2 * kGenTH + F - 2; }
inline size_t gen_fwd_lds_bytes(int F) {
  return sizeof(float) * (size_t)(gen_fwd_rows(F) + 2 * kGenTH) * gen_fwd_cols(F);
}

template <int IN_KIND>
__global__ __launch_bounds__(256) void k_fwd_gen(GenFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float gen_smem[];
  const int F = a.F;
  const int R = gen_fwd_rows(F), C = gen_fwd_cols(F), CV = C - 1;
  float* s_in = gen_smem;             // [R][C]
  float* s_lo = s_in + R * C;         // [TH][C]
  float* s_hi = s_lo + kGenTH * C;    // [TH][C]
  const int tid = threadIdx.x;
  const int plane = blockIdx.z;
  const int i0 = blockIdx.y * kGenTH, j0 = blockIdx.x * kGenTW;
  const int gr0 = 2 * i0 + 1 - (F - 1), gc0 = 2 * j0 + 1 - (F - 1);  // first input row / column of the patch

  // ---- patch -> LDS (symmetric extension by index reflection); statistic on the pixels this block owns ----
  double s_all = 0.0, s_fg = 0.0;
  unsigned cnt = 0, bad = 0;
  for (int e = tid; e < R * CV; e += 256) {
    const int r = e / CV, c = e - r * CV;
    const int gr = gr0 + r, gc = gc0 + c;
    const int sr = reflect_idx(gr, a.H), sc = reflect_idx(gc, a.W);
    float v;
    if (IN_KIND == 2) {
      v = a.ws[plane * a.ws_plane_stride + a.in_off + (long long)sr * a.ldin + sc];
    } else {
      const long long off = plane * a.in_plane_stride + (long long)sr * a.W + sc;
      const float x = (IN_KIND == 0) ? (float)((const uint16_t*)a.in)[off] : ((const float*)a.in)[off];
      const bool own = gr >= 2 * i0 && gr < 2 * i0 + 2 * kGenTH && gr >= 0 && gr < a.H &&
                       gc >= 2 * j0 && gc < 2 * j0 + 2 * kGenTW && gc >= 0 && gc < a.W;
      if (own) {
        if (IN_KIND == 1) bad |= (x > -1.0f && x <= 3.402823466e38f) ? 0u : 1u;
        s_all += (double)x;
        if (x >= a.fg_cutoff) { s_fg += (double)x; cnt++; }
      }
      v = logf(1.0f + x);
    }
    s_in[r * C + c] = v;
  }
  __syncthreads();

  // ---- axis 0 (columns): lo / hi rows of the tile ----
  for (int e = tid; e < kGenTH * CV; e += 256) {
    const int oi = e / CV, c = e - oi * CV;
    const float* col = s_in + (2 * oi + F - 1) * C + c;  // input row 2 (i0 + oi) + 1 - t  <->  local 2 oi + F - 1 - t
    float lo = 0.f, hi = 0.f;
    for (int t = 0; t < F; ++t) {
      const float x = col[-t * C];
      lo = fmaf(a.lo[t], x, lo);
      hi = fmaf(a.hi[t], x, hi);
    }
    s_lo[oi * C + c] = lo;
    s_hi[oi * C + c] = hi;
  }
  __syncthreads();

  // ---- axis 1 (rows), low-pass of both: aa and da; min / max of da^2 ----
  float qmin = __builtin_huge_valf(), qmax = -1.0f;
  float* aa = a.ws + plane * a.ws_plane_stride + a.aa_off;
  float* da = a.ws + plane * a.ws_plane_stride + a.da_off;
  for (int e = tid; e < kGenTH * kGenTW; e += 256) {
    const int oi = e / kGenTW, oj = e - oi * kGenTW;
    const int i = i0 + oi, j = j0 + oj;
    if (i >= a.h || j >= a.w) continue;
    const float* rl = s_lo + oi * C + 2 * oj + F - 1;
    const float* rh = s_hi + oi * C + 2 * oj + F - 1;
    float va = 0.f, vd = 0.f;
    for (int t = 0; t < F; ++t) {
      va = fmaf(a.lo[t], rl[-t], va);
      vd = fmaf(a.lo[t], rh[-t], vd);
    }
    aa[(long long)i * a.lda + j] = va;
    da[(long long)i * a.ld + j] = vd;
    const float q = vd * vd;
    qmin = fminf(qmin, q);
    qmax = fmaxf(qmax, q);
  }
  qmin = wave_min_f32(qmin);
  qmax = wave_max_f32(qmax);
  if ((tid & 63) == 0 && qmin <= qmax) {
    unsigned* mm = a.minmax + ((long long)(a.shared ? 0 : plane) * a.L + a.lvl) * 2;
    atomicMax(&mm[0], ~as_u32(qmin));
    atomicMax(&mm[1], as_u32(qmax));
  }
  if (IN_KIND != 2) {
    s_all = wave_sum_f64(s_all);
    s_fg = wave_sum_f64(s_fg);
    cnt = __reduce_add_sync(~0ull, cnt);
    const bool bad_any = __any(bad != 0u) != 0;
    if ((tid & 63) == 0) {
      PlaneStats* ps = a.stats + plane;
      if (s_all != 0.0) atomicAdd(&ps->sum_all, s_all);
      if (cnt != 0) {
        atomicAdd(&ps->sum_fg, s_fg);
        atomicAdd(&ps->cnt_fg, (unsigned long long)cnt);
      }
      if (IN_KIND == 1 && bad_any) atomicOr(&ps->flags, 1ull);
    }
  }
}

struct GenInvArgs {
  float* ws;
  long long ws_plane_stride;
  long long c_off, d_off;
  int hc, wc, ldc, ldd;  // coefficient shape (of Delta_l; c_l is trimmed to it), pitches
  int has_c, has_pyr;
  const void* img;
  long long img_plane_stride;
  int H, W;
  void* out;
  long long out_plane_stride;
  int hout, wout;
  int out_dtype;  // 0 = uint16, 1 = float32
  const float* flat;
  const float* dark;
  int dark_ld;
  long long out_off;  // MODE 2: c_{l-1} destination inside the workspace
  int ldout;
  int F;
  float lo[kMaxTaps], hi[kMaxTaps];  // rec_lo, rec_hi
};

inline __host__ __device__ int gen_inv_krows(int F) { return kGenOH / 2 + F / 2 - 1; }
inline __host__ __device__ int gen_inv_kcols(int F) { return kGenOW / 2 + F / 2 - 1; }
inline size_t gen_inv_lds_bytes(int F) {
  return sizeof(float) * 2 * (size_t)gen_inv_krows(F) * (gen_inv_kcols(F) + 1 + kGenOW + 1);
}

// MODE 0 / 1: last level with uint16 / float32 pixels; MODE 2: pyramid level (writes c_{l-1} into the workspace)
template <int MODE>
__global__ __launch_bounds__(256) void k_inv_gen(GenInvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float gen_smem[];
  const int F = a.F, F2 = F >> 1;
  const int KR = gen_inv_krows(F), KC = gen_inv_kcols(F), PC = KC + 1, PW = kGenOW + 1;
  float* s_c = gen_smem;            // [KR][PC]  c_l
  float* s_d = s_c + KR * PC;       // [KR][PC]  Delta_l
  float* s_a0 = s_d + KR * PC;      // [KR][PW]  row synthesis of c_l
  float* s_d0 = s_a0 + KR * PW;     // [KR][PW]  row synthesis of Delta_l
  const int tid = threadIdx.x;
  const int plane = blockIdx.z;
  const int m0 = blockIdx.y * kGenOH, n0 = blockIdx.x * kGenOW;
  const int k0 = m0 >> 1, q0 = n0 >> 1;
  const bool pyr = a.has_pyr != 0;

  if (pyr) {
    const float* cbase = a.ws + plane * a.ws_plane_stride + a.c_off;
    const float* dbase = a.ws + plane * a.ws_plane_stride + a.d_off;
    for (int e = tid; e < KR * KC; e += 256) {
      const int r = e / KC, c = e - r * KC;
      const int k = k0 + r, q = q0 + c;
      const bool in = k < a.hc && q < a.wc;
      s_c[r * PC + c] = (in && a.has_c) ? cbase[(long long)k * a.ldc + q] : 0.f;
      s_d[r * PC + c] = in ? dbase[(long long)k * a.ldd + q] : 0.f;
    }
    __syncthreads();
    // ---- axis 1: both sub-bands through rec_lo (their high-pass partners cV, cD carry no correction) ----
    for (int e = tid; e < KR * kGenOW; e += 256) {
      const int r = e / kGenOW, x = e - r * kGenOW;
      const float* pc = s_c + r * PC + (x >> 1);
      const float* pd = s_d + r * PC + (x >> 1);
      const int t0 = F - 2 + (x & 1);
      float va = 0.f, vd = 0.f;
      for (int j = 0; j < F2; ++j) {
        const float tap = a.lo[t0 - 2 * j];
        va = fmaf(pc[j], tap, va);
        vd = fmaf(pd[j], tap, vd);
      }
      s_a0[r * PW + x] = va;
      s_d0[r * PW + x] = vd;
    }
    __syncthreads();
  }

  // ---- axis 0 + store / finish ----
  for (int e = tid; e < kGenOH * kGenOW; e += 256) {
    const int y = e / kGenOW, x = e - y * kGenOW;
    const int m = m0 + y, n = n0 + x;
    if (m >= a.hout || n >= a.wout) continue;
    float c0 = 0.f;
    if (pyr) {
      const float* pa = s_a0 + (y >> 1) * PW + x;
      const float* pd = s_d0 + (y >> 1) * PW + x;
      const int t0 = F - 2 + (y & 1);
      for (int j = 0; j < F2; ++j) {
        c0 = fmaf(pa[j * PW], a.lo[t0 - 2 * j], c0);
        c0 = fmaf(pd[j * PW], a.hi[t0 - 2 * j], c0);
      }
    }
    if (MODE == 2) {
      a.ws[plane * a.ws_plane_stride + a.out_off + (long long)m * a.ldout + n] = c0;
      continue;
    }
    // an odd plane grows by one replicated row / column
    const long long ioff = plane * a.img_plane_stride + (long long)min(m, a.H - 1) * a.W + min(n, a.W - 1);
    const float px = (MODE == 0) ? (float)((const uint16_t*)a.img)[ioff] : ((const float*)a.img)[ioff];
    float v = fmaf(1.0f + px, expf(c0), 1.0f);  // exp(log(1 + x) + c0) + 1  (filtering.py:222)
    if (a.flat != nullptr) {                      // flatfield_correction, filtering.py:399-412
      const float dark = a.dark[(long long)m * a.dark_ld + n], flat = a.flat[(long long)m * a.wout + n];
      v = (v > dark) ? (v - dark) : 0.f;
      v = v / flat;
      v = fminf(fmaxf(v, 0.f), 65535.f);
    }
    const long long o = plane * a.out_plane_stride + (long long)m * a.wout + n;
    if (a.out_dtype == 0) ((uint16_t*)a.out)[o] = (uint16_t)(unsigned)fminf(fmaxf(v, 0.f), 65535.f);
    else ((float*)a.out)[o] = v;
  }
}

}  // namespace dsx
#endif  // DSX_WAVELET_H
