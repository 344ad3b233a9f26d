// dsx_plan.h -- host-side planning: level geometry, workspace layout, FFT factorisation and the
// spectral gain tables of the row filter (DESIGN.md sections 2 and 3.4).  Pure C++ (no HIP), so
// that tests/test_plan_host.py can exercise it with g++.
//
// Reference semantics restated here:
//   level selection          pywt.wavedec2 / dwt_max_level        (call site filtering.py:176)
//   width_fraction, s        filtering.py:180, 213
//   notch gains g = 1 - e    filtering.py:91-115, applied in fftpack packed order (:206-215)
#ifndef DSX_PLAN_H
#define DSX_PLAN_H

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

namespace dsx {

constexpr int kPlanMaxLevels = 16;
constexpr int kPlanMaxPasses = 16;
constexpr int kFilterLen = 6;           // db3 (the default filter length of build_plan)
constexpr int kMaxFftLen = 64 * 36;     // largest row-filter kernel instantiation (CPL = 36)
constexpr int kMaxWideLen = 512 * 36;   // k_rowfilter_wide (one block per row pair): levels too long for one wave

struct HostCfg {
  int level;  // -1 == maximum
  double sigma;
  double max_threshold;
};

struct C32 {
  float re, im;
};

struct LevelPlan {
  int hin, win, ldin;  // input of the analysis level (level 0: the plane itself)
  int h, w, ld;        // coefficient shape and row pitch of da (multiple of 4 floats, >= w + 4)
  int lda;             // row pitch of aa: 4 margin columns left + w + 8 right (symmetric extension)
  long long aa_off, da_off;
  // row filter
  int M, K;
  int npass;
  int radix[kPlanMaxPasses];
  long long tw_off;    // C32 offset of the twiddles in the constant blob
  long long g_off[2];  // C32 offset of G1[M] (followed by G2[M]) per config
  int kcut[2];         // per config: G1[k] = G2[k] = 0 (below 1e-9 of the unit DC gain) for kcut < k < M - kcut
};

struct Plan {
  int H = 0, W = 0, Hout = 0, Wout = 0;
  int L = 0;            // levels run (max over both configs)
  int cfg_levels[2] = {0, 0};
  LevelPlan lv[kPlanMaxLevels];
  long long plane_floats = 0;  // workspace floats per plane
  std::vector<C32> consts;     // twiddles + gain tables
};

// pywt.dwt_max_level(data_len, filter_len)
inline int dwt_max_level(int n, int filter_len = kFilterLen) {
  if (filter_len < 2 || n < filter_len - 1) return 0;
  int l = (int)floor(log2((double)(n / (filter_len - 1))));
  return l < 0 ? 0 : l;
}

// kernel instantiations of k_rowfilter: complex values per lane (register budget / occupancy)
inline int cpl_class(int m) {
  const int cpl = (m + 63) / 64;
  for (int c : {2, 4, 6, 10, 18, 36}) if (cpl <= c) return c;
  return 1 << 30;
}

// Register-butterfly radices of k_rowfilter and the VALU instructions of one "round" of a pass
// (64 butterflies, one per lane; measured on the gfx950 code of the CPL = 18 instantiation, rocm 7.2).
// A pass over M values runs ceil(M / R / 64) rounds, so a radix whose butterfly count is just above
// a multiple of 64 pays a whole round for a few lanes (R = 16 at M = 1152: 72 butterflies, 2 rounds).
constexpr int kRegRadix[] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 19, 20, 25};
inline double round_cost(int r) {
  switch (r) {
    case 2: return 28;
    case 3: return 37;
    case 4: return 64;
    case 5: return 84;
    case 6: return 82;
    case 7: return 109;
    case 8: return 133;
    case 9: return 151;
    case 10: return 214;
    case 11: return 191;
    case 12: return 251;
    case 13: return 238;
    case 15: return 352;
    case 16: return 337;
    case 17: return 359;
    case 19: return 408;
    case 20: return 520;
    case 25: return 708;
    default: return 64.0 * (8.0 + 10.0 * r);  // generic pass: O(R) per value
  }
}
// cost of one pass of radix r inside a length-m transform (+ a fixed part per pass: two wave-level LDS
// fences and the exposed LDS latency)
inline double pass_cost(int m, int r) {
  constexpr double kPerPass = 60.0;
  bool reg = false;
  for (int q : kRegRadix) reg = reg || (q == r);
  if (!reg) return kPerPass + round_cost(r) * ((m + 63) / 64) / (double)r;  // one value per lane and round
  return kPerPass + round_cost(r) * (double)((m / r + 63) / 64);
}

// Cheapest decomposition of m into passes: dynamic programme over the divisors of m (the cost of a
// pass depends on m and its radix only, not on the order).  Prime factors above 19 go through the
// generic pass.
inline void best_passes(int m, int rest, std::vector<int>& cur, double cost, std::vector<int>& best,
                        double& best_cost) {
  if (cost >= best_cost || (int)cur.size() > kPlanMaxPasses) return;
  if (rest == 1) { best = cur; best_cost = cost; return; }
  bool any = false;
  const int last = cur.empty() ? 1 << 30 : cur.back();
  for (int i = (int)(sizeof(kRegRadix) / sizeof(int)) - 1; i >= 0; --i) {
    const int r = kRegRadix[i];
    if (rest % r) continue;
    any = true;
    if (r > last) continue;  // non-increasing order: every multiset is visited once
    cur.push_back(r);
    best_passes(m, rest / r, cur, cost + pass_cost(m, r), best, best_cost);
    cur.pop_back();
  }
  if (!any) {  // a prime factor without a register butterfly
    int pr = rest;
    for (int q = 2; q * q <= rest; ++q) if (rest % q == 0) { pr = q; break; }
    cur.push_back(pr);
    best_passes(m, rest / pr, cur, cost + pass_cost(m, pr), best, best_cost);
    cur.pop_back();
  }
}

// Radix list of a length, in pass order.
inline std::vector<int> factorize(int n, double* cost_out = nullptr) {
  std::vector<int> r, cur;
  double best = 1e300;
  if (n > 1) best_passes(n, n, cur, 0.0, r, best);
  else best = 0.0;
  // Pass order = LDS bank behaviour: the first pass scatters with stride R (buf[R b + k]) and gathers
  // twiddles tw[b k]; a power-of-two R puts whole lane groups on one bank (16-way conflicts for
  // R = 16).  Radices with an odd factor go first, pure powers of two last (their stride is then
  // multiplied by the odd product of the earlier passes).
  std::stable_sort(r.begin(), r.end(), [](int a, int b) {
    const bool pa = (a & (a - 1)) == 0, pb = (b & (b - 1)) == 0;
    return !pa && pb;
  });
  if (cost_out) *cost_out = best;
  return r;
}

inline double fft_cost(int m) {
  double c = 0;
  factorize(m, &c);
  // the 36-per-lane instantiation runs at one wave per SIMD: roughly 3x slower per instruction
  const double occ = (cpl_class(m) > 18) ? 3.0 : 1.0;
  // everything outside the passes (keys, median, in-paint, spectral step, row I/O) scales with m
  return (c + 30.0 * ((m + 63) / 64)) * occ;
}

// LP gains in fftpack packed order, folded to the complex bins of a length-n transform:
//   e[j] = exp(-j^2 / (2 s^2));  ep[k] = (e[2k-1] + e[2k]) / 2,  em[k] = (e[2k-1] - e[2k]) / 2
//   (1 <= k <= (n-1)/2, mirrored to n-k);  ep[0] = e[0] = 1;  even n: ep[n/2] = e[n-1], em = 0.
inline void packed_gains(int n, double s, std::vector<double>& ep, std::vector<double>& em) {
  ep.assign(n, 0.0);
  em.assign(n, 0.0);
  auto e = [&](int j) { return exp(-((double)j * (double)j) / (2.0 * s * s)); };
  ep[0] = e(0);
  for (int k = 1; k <= (n - 1) / 2; ++k) {
    const double ea = e(2 * k - 1), eb = e(2 * k);
    ep[k] = ep[n - k] = 0.5 * (ea + eb);
    em[k] = em[n - k] = 0.5 * (ea - eb);
  }
  if (n % 2 == 0 && n >= 2) {
    ep[n / 2] = e(n - 1);
    em[n / 2] = 0.0;
  }
}

// real even sequence -> real even inverse DFT:  h[d] = (1/n) sum_k g[k] cos(2 pi k d / n)
inline std::vector<double> idft_even(const std::vector<double>& g) {
  const int n = (int)g.size();
  std::vector<double> c(n), h(n);
  for (int t = 0; t < n; ++t) c[t] = cos(2.0 * M_PI * t / n);
  for (int d = 0; d < n; ++d) {
    double acc = 0;
    for (int k = 0; k < n; ++k) acc += g[k] * c[(int)(((long long)k * d) % n)];
    h[d] = acc / n;
  }
  return h;
}

// Build the plan.  Returns "" or an error text.
// filter_len: taps of the wavelet's filters (6 = db3; other lengths run the kernels of dsx_wavelet.h)
inline std::string build_plan(int H, int W, const HostCfg cfg[2], Plan& p, int filter_len = kFilterLen) {
  if (H < 1 || W < 1) return "plane must be at least 1x1";
  if (filter_len < 2 || (filter_len & 1)) return "wavelet filters must have an even number (>= 2) of taps";
  for (int c = 0; c < 2; ++c) {
    if (!(cfg[c].sigma > 0)) return "sigma must be positive";
    if (cfg[c].level < -1) return "Level value is too low . Minimum level is 0.";
  }
  p = Plan();
  p.H = H;
  p.W = W;
  const int max_level = std::min(dwt_max_level(H, filter_len), dwt_max_level(W, filter_len));
  for (int c = 0; c < 2; ++c) p.cfg_levels[c] = cfg[c].level < 0 ? max_level : cfg[c].level;
  p.L = std::max(p.cfg_levels[0], p.cfg_levels[1]);
  if (p.L > kPlanMaxLevels) return "too many decomposition levels";
  if ((H & 1) || (W & 1)) {
    // an odd plane grows by one row/column iff at least one level runs; both configs must agree
    if ((p.cfg_levels[0] == 0) != (p.cfg_levels[1] == 0))
      return "odd plane: both configs must either run levels or none (result shapes differ)";
  }
  p.Hout = (p.L > 0) ? H + (H & 1) : H;
  p.Wout = (p.L > 0) ? W + (W & 1) : W;

  long long off = 0;
  int hin = H, win = W, ldin = W;
  const double min_hw = (double)std::min(H, W);
  for (int l = 0; l < p.L; ++l) {
    LevelPlan& lp = p.lv[l];
    lp.hin = hin; lp.win = win; lp.ldin = ldin;
    lp.h = (hin + filter_len - 1) / 2;
    lp.w = (win + filter_len - 1) / 2;
    // da rows: >= 4 spare columns so that the synthesis kernels may load 4 coefficients past the end;
    // aa rows: logical column j lives at aa_off + i * lda + j, j in [-4, w + 8): the analysis kernel
    // stores the half-sample symmetric extension there, so the next level loads aligned groups only
    lp.ld = (lp.w + 4 + 3) & ~3;
    lp.lda = (lp.w + 12 + 3) & ~3;
    lp.aa_off = off + 4; off += (long long)lp.h * lp.lda + 8;
    lp.da_off = off; off += (long long)lp.h * lp.ld + 8;
    hin = lp.h; win = lp.w; ldin = lp.lda;

    // ---- row filter plan for this level -----------------------------------------------------
    const int n = lp.w;
    std::vector<double> ep[2], em[2], h1[2], h2[2];
    for (int c = 0; c < 2; ++c) {
      const double s = lp.h * (cfg[c].sigma / min_hw);  // filtering.py:180, 213
      packed_gains(n, s, ep[c], em[c]);
    }
    // Candidates: (a) the direct length-n transform (generic O(R) passes for prime factors > 19);
    // (b) the exact embedding of the length-n circular operator in a 19-smooth length
    //     M >= n + 2K + 1 with a periodic halo of K = floor(n / 2) samples on both sides.
    //     (The spatial kernels of ep / em decay only like 1/d^2 -- the packed-index gains are not
    //      smooth at k = 0 -- so the halo cannot be truncated below n / 2.)
    int best_m = n, best_k = 0;
    double best_cost = fft_cost(n);
    // transforms longer than one wave holds run k_rowfilter_wide (one block per row pair; priced like the CPL = 36 class
    // by fft_cost): rows of planes wider than ~4 600 px, and the embedding of a shorter row whose length has a large prime
    // factor (n = 1283 at level 2 of a 5120-wide plane: a generic radix-1283 pass would cost 400 x the passes of 2 601)
    const int max_len = kMaxWideLen;
    if (n > max_len) best_cost = 1e300;
    if (n >= 8) {
      const int K = n / 2;
      const int need = n + 2 * K + 1;
      for (int m = need; m <= std::min(2 * need, max_len); ++m) {
        int t = m;
        for (int q : {2, 3, 5, 7, 11, 13, 17, 19}) while (t % q == 0) t /= q;
        if (t != 1) continue;  // only lengths whose passes all have register butterflies
        const double c = fft_cost(m);
        if (c < best_cost) { best_cost = c; best_m = m; best_k = K; }
      }
    }
    // measurement hook (tools/plan_direct_vs_embedded.sh): DSX_PLAN_DIRECT_LEVEL=<l> forces the direct
    // length-n transform at level index l whatever the cost model says -- another transform, other roundings: only in
    // a -DDSX_DIAG build (tools/build_variant.sh), the product has no such switch
#ifdef DSX_DIAG
    if (const char* fd = getenv("DSX_PLAN_DIRECT_LEVEL")) {
      if (atoi(fd) == l && n <= kMaxFftLen) { best_m = n; best_k = 0; }
    }
#endif
    if (best_k > 0) {
      for (int c = 0; c < 2; ++c) {
        h1[c] = idft_even(ep[c]);
        h2[c] = idft_even(em[c]);
      }
    }
    if (best_cost >= 1e300) return "plane too wide for the row-filter kernels";
    lp.M = best_m;
    lp.K = best_k;
    const std::vector<int> rad = factorize(best_m);
    if ((int)rad.size() > kPlanMaxPasses) return "too many FFT passes";
    lp.npass = (int)rad.size();
    for (int i = 0; i < lp.npass; ++i) lp.radix[i] = rad[i];
    if (lp.npass == 0) { lp.npass = 0; }  // M == 1: no pass, transform is the identity

    const int M = lp.M;
    lp.tw_off = (long long)p.consts.size();
    for (int t = 0; t < M; ++t) {
      const double a = -2.0 * M_PI * t / M;
      p.consts.push_back(C32{(float)cos(a), (float)sin(a)});
    }
    for (int c = 0; c < 2; ++c) {
      lp.g_off[c] = (long long)p.consts.size();
      std::vector<C32> g1(M), g2(M);
      if (lp.K == 0) {  // direct: G1 = ep, G2 = em (real)
        for (int k = 0; k < M; ++k) {
          g1[k] = C32{(float)ep[c][k], 0.f};
          g2[k] = C32{(float)em[c][k], 0.f};
        }
      } else {
        const int Kc = lp.K, cshift = n + 2 * Kc;
        std::vector<double> cs(M), sn(M);
        for (int t = 0; t < M; ++t) { cs[t] = cos(2.0 * M_PI * t / M); sn[t] = sin(2.0 * M_PI * t / M); }
        for (int k = 0; k < M; ++k) {
          double a1 = h1[c][0], a2 = h2[c][0];
          for (int d = 1; d <= Kc; ++d) {
            // h[-d] = h[n - d] = h[d]; for even n the lag n/2 is one circular lag: half weight each side
            const double wgt = (2 * d == n) ? 1.0 : 2.0;
            const double cc = wgt * cs[(int)(((long long)d * k) % M)];
            a1 += h1[c][d] * cc;
            a2 += h2[c][d] * cc;
          }
          const int t = (int)(((long long)cshift * k) % M);
          g1[k] = C32{(float)a1, 0.f};
          g2[k] = C32{(float)(a2 * cs[t]), (float)(-a2 * sn[t])};  // * exp(-2 pi i c k / M)
        }
      }
      // band limit of the low-pass: bins whose gains are below 1e-9 (of a unit DC gain) are written as
      // exact zeros by the spectral step without reading anything
      int kc = 0;
      for (int k = 0; k < M; ++k) {
        const double mag = std::max(fabs((double)g1[k].re), sqrt((double)g2[k].re * g2[k].re + (double)g2[k].im * g2[k].im));
        if (mag > 1e-9) kc = std::max(kc, std::min(k, M - k));
      }
      lp.kcut[c] = kc;
      p.consts.insert(p.consts.end(), g1.begin(), g1.end());
      p.consts.insert(p.consts.end(), g2.begin(), g2.end());
    }
  }
  p.plane_floats = (off + 3) & ~3LL;
  if (p.plane_floats == 0) p.plane_floats = 4;
  return "";
}

}  // namespace dsx
#endif  // DSX_PLAN_H
