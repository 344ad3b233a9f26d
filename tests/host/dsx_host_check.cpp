// Host-side (g++) checks of the code the HIP kernels share with the CPU:
//   fft  <M>                     Stockham passes built from dsx_fft_core.h vs a naive DFT
//   rows <H> <W> <sigma> <level> <lvl>  emulates k_rowfilter's spectral pipeline for one level with the
//                                planned (M, K, radices, G tables) and compares with the exact
//                                length-w circular operator of the reference (packed-index gains)
//   plan <H> <W> <s0> <l0> <s1> <l1>    prints the plan (levels, dims, M, K, radices)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <complex>
#include <vector>

#include "../../aind_smartspim_destripe_amd/csrc/dsx_fft_core.h"
#include "../../aind_smartspim_destripe_amd/csrc/dsx_plan.h"

typedef std::complex<double> cd;

static void run_passes(std::vector<dsx_c32>& buf, const std::vector<dsx_c32>& tw, const int* radix, int npass) {
  const int M = (int)buf.size();
  int s = 1;
  for (int pi = 0; pi < npass; ++pi) {
    const int R = radix[pi];
    const float inv_s = 1.0f / (float)s;
    const int nb = M / R;
    if (R <= 20 || R == 25) {
      // read phase for every butterfly, then compute + scatter phase (what a wave does)
      std::vector<dsx_c32> regs((size_t)nb * R);
      for (int b = 0; b < nb; ++b) {
        switch (R) {
          case 2: dsx_bfly_load<2>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 3: dsx_bfly_load<3>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 4: dsx_bfly_load<4>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 5: dsx_bfly_load<5>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 7: dsx_bfly_load<7>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 11: dsx_bfly_load<11>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 13: dsx_bfly_load<13>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 17: dsx_bfly_load<17>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 19: dsx_bfly_load<19>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 6: dsx_bfly_load<6>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 8: dsx_bfly_load<8>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 9: dsx_bfly_load<9>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 10: dsx_bfly_load<10>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 12: dsx_bfly_load<12>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 15: dsx_bfly_load<15>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 16: dsx_bfly_load<16>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 20: dsx_bfly_load<20>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
          case 25: dsx_bfly_load<25>(buf.data(), b, nb, &regs[(size_t)b * R]); break;
        }
      }
      for (int b = 0; b < nb; ++b) {
        switch (R) {
          case 2: dsx_bfly_store<2>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R]); break;
          case 3: dsx_bfly_store<3>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R]); break;
          case 4: dsx_bfly_store<4>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R]); break;
          case 5: dsx_bfly_store<5>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R]); break;
          case 7: dsx_bfly_store<7>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R]); break;
          case 11: dsx_bfly_store<11>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R]); break;
          case 13: dsx_bfly_store<13>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R]); break;
          case 17: dsx_bfly_store<17>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R]); break;
          case 19: dsx_bfly_store<19>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R]); break;
          case 6: dsx_bfly_store<6>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R], s * R == M); break;
          case 8: dsx_bfly_store<8>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R], s * R == M); break;
          case 9: dsx_bfly_store<9>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R], s * R == M); break;
          case 10: dsx_bfly_store<10>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R], s * R == M); break;
          case 12: dsx_bfly_store<12>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R], s * R == M); break;
          case 15: dsx_bfly_store<15>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R], s * R == M); break;
          case 16: dsx_bfly_store<16>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R], s * R == M); break;
          case 20: dsx_bfly_store<20>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R], s * R == M); break;
          case 25: dsx_bfly_store<25>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R], s * R == M); break;
        }
      }
    } else {
      std::vector<dsx_c32> acc(M);
      for (int o = 0; o < M; ++o) acc[o] = dsx_generic_output(buf.data(), tw.data(), o, M, s, inv_s, R);
      buf = acc;
    }
    s *= R;
  }
}

static std::vector<dsx_c32> twiddles(int M);

// The forward plan of the level-1 row filter at 2048 columns (StaticFft<1>::run_forward in csrc/dsx_kernels.h): passes
// 6, 9 in full, then the radix-19 pass with only the output pairs k <= KO -- every bin |k| <= kcut must be the DFT.
template <int KO>
static double pruned_1026(int kcut) {
  const int M = 1026;
  std::vector<dsx_c32> buf(M), tw = twiddles(M);
  std::vector<cd> x(M);
  srand(1026 + KO);
  for (int i = 0; i < M; ++i) {
    x[i] = cd(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
    buf[i] = dsx_mk((float)x[i].real(), (float)x[i].imag());
  }
  const int first[2] = {6, 9};
  // (run_passes derives the sub-transform stride from the radices it is given: 1, then 6)
  {
    int s = 1;
    for (int pi = 0; pi < 2; ++pi) {
      const int R = first[pi], nb = M / R;
      const float inv_s = 1.0f / (float)s;
      std::vector<dsx_c32> regs((size_t)nb * R);
      for (int b = 0; b < nb; ++b) {
        if (R == 6) dsx_bfly_load<6>(buf.data(), b, nb, &regs[(size_t)b * R]);
        else dsx_bfly_load<9>(buf.data(), b, nb, &regs[(size_t)b * R]);
      }
      for (int b = 0; b < nb; ++b) {
        if (R == 6) dsx_bfly_store<6>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R], false);
        else dsx_bfly_store<9>(buf.data(), tw.data(), b, s, inv_s, &regs[(size_t)b * R], false);
      }
      s *= R;
    }
  }
  {
    const int R = 19, nb = M / R, s = 54;
    std::vector<dsx_c32> regs((size_t)nb * R);
    for (int b = 0; b < nb; ++b) dsx_bfly_load<19>(buf.data(), b, nb, &regs[(size_t)b * R]);
    for (int b = 0; b < nb; ++b) dsx_bfly_store<19, 9, KO>(buf.data(), tw.data(), b, s, 1.0f / 54.0f, &regs[(size_t)b * R], true);
  }
  double err = 0, nrm = 0;
  for (int k = 0; k < M; ++k) {
    if (k > kcut && k < M - kcut) continue;
    cd acc = 0;
    for (int j = 0; j < M; ++j) acc += x[j] * std::polar(1.0, -2.0 * M_PI * (double)((long long)j * k % M) / M);
    err = fmax(err, std::abs(acc - cd(buf[k].x, buf[k].y)));
    nrm = fmax(nrm, std::abs(acc));
  }
  return err / nrm;
}

// The power-of-two plans with bank-swizzled addressing (dsx_idx_swz on the row buffer AND the twiddle table, as
// StaticFft<3> / <5> run them): the data is placed at sw(i), the twiddles at sw(t), every pass addresses through the
// policy, and the result is read back from sw(k) -- it must be the DFT exactly as without the swizzle.
template <int R>
static void swz_pass(std::vector<dsx_c32>& buf, const std::vector<dsx_c32>& tw, int M, int s) {
  const int nb = M / R;
  std::vector<dsx_c32> regs((size_t)nb * R);
  for (int b = 0; b < nb; ++b) dsx_bfly_load<R, (R - 1) / 2, dsx_idx_swz>(buf.data(), b, nb, &regs[(size_t)b * R]);
  for (int b = 0; b < nb; ++b)
    dsx_bfly_store<R, (R - 1) / 2, (R - 1) / 2, dsx_idx_swz>(buf.data(), tw.data(), b, s, 1.0f / (float)s, &regs[(size_t)b * R], s * R == M);
}

// the same passes with the addressing written out (dsx_swz_load / dsx_swz_store): must give the very same bits
template <int R, int M, int S>
static void swz_pass_fast(std::vector<dsx_c32>& buf, const std::vector<dsx_c32>& tw) {
  const int nb = M / R;
  std::vector<dsx_c32> regs((size_t)nb * R);
  for (int b = 0; b < nb; ++b) dsx_swz_load<R, M>(buf.data(), b, &regs[(size_t)b * R]);
  for (int b = 0; b < nb; ++b) dsx_swz_store<R, M, S>(buf.data(), tw.data(), b, &regs[(size_t)b * R]);
}

static double swizzled(int M) {
  std::vector<dsx_c32> buf(M), plain = twiddles(M), tw(M);
  std::vector<cd> x(M);
  srand(M + 7);
  for (int i = 0; i < M; ++i) {
    x[i] = cd(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
    buf[dsx_idx_swz::at(i)] = dsx_mk((float)x[i].real(), (float)x[i].imag());
    tw[dsx_idx_swz::at(i)] = plain[i];
  }
  std::vector<dsx_c32> fast = buf;
  if (M == 2048) { swz_pass<16>(buf, tw, M, 1); swz_pass<16>(buf, tw, M, 16); swz_pass<8>(buf, tw, M, 256); }
  else { swz_pass<16>(buf, tw, M, 1); swz_pass<8>(buf, tw, M, 16); swz_pass<8>(buf, tw, M, 128); }
  if (M == 2048) { swz_pass_fast<16, 2048, 1>(fast, tw); swz_pass_fast<16, 2048, 16>(fast, tw); swz_pass_fast<8, 2048, 256>(fast, tw); }
  else { swz_pass_fast<16, 1024, 1>(fast, tw); swz_pass_fast<8, 1024, 16>(fast, tw); swz_pass_fast<8, 1024, 128>(fast, tw); }
  for (int i = 0; i < M; ++i)
    if (memcmp(&fast[i], &buf[i], sizeof(dsx_c32)) != 0) return 1.0;  // written-out addressing differs from the policy
  double err = 0, nrm = 0;
  for (int k = 0; k < M; ++k) {
    cd acc = 0;
    for (int j = 0; j < M; ++j) acc += x[j] * std::polar(1.0, -2.0 * M_PI * (double)((long long)j * k % M) / M);
    const dsx_c32 y = buf[dsx_idx_swz::at(k)];
    err = fmax(err, std::abs(acc - cd(y.x, y.y)));
    nrm = fmax(nrm, std::abs(acc));
  }
  return err / nrm;
}

static int cmd_swz() {
  printf("{\"m2048\": %.3e, \"m1024\": %.3e}\n", swizzled(2048), swizzled(1024));
  return 0;
}

static int cmd_pruned() {
  // kcut of the production configs at 2048 columns: 103 (cells) / 206 (no cells); 107 / 215 are the last bins 2 / 4 pairs reach
  printf("{\"ko2_kcut103\": %.3e, \"ko2_kcut107\": %.3e, \"ko4_kcut206\": %.3e, \"ko4_kcut215\": %.3e, \"ko9_full\": %.3e}\n",
         pruned_1026<2>(103), pruned_1026<2>(107), pruned_1026<4>(206), pruned_1026<4>(215), pruned_1026<9>(512));
  return 0;
}

static std::vector<dsx_c32> twiddles(int M) {
  std::vector<dsx_c32> tw(M);
  for (int t = 0; t < M; ++t) {
    const double a = -2.0 * M_PI * t / M;
    tw[t] = dsx_mk((float)cos(a), (float)sin(a));
  }
  return tw;
}

static int cmd_fft(int M) {
  std::vector<int> rad = dsx::factorize(M);
  std::vector<dsx_c32> buf(M);
  std::vector<cd> x(M);
  srand(M);
  for (int i = 0; i < M; ++i) {
    x[i] = cd(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
    buf[i] = dsx_mk((float)x[i].real(), (float)x[i].imag());
  }
  run_passes(buf, twiddles(M), rad.data(), (int)rad.size());
  double err = 0, nrm = 0;
  for (int k = 0; k < M; ++k) {
    cd acc = 0;
    for (int j = 0; j < M; ++j) acc += x[j] * std::polar(1.0, -2.0 * M_PI * (double)((long long)j * k % M) / M);
    err = fmax(err, std::abs(acc - cd(buf[k].x, buf[k].y)));
    nrm = fmax(nrm, std::abs(acc));
  }
  printf("{\"M\": %d, \"npass\": %d, \"rel_err\": %.3e}\n", M, (int)rad.size(), err / nrm);
  return 0;
}

static int cmd_plan(int H, int W, double s0, int l0, double s1, int l1, dsx::Plan& p, bool print) {
  dsx::HostCfg cfg[2] = {{l0, s0, 12.0}, {l1, s1, 3.0}};
  std::string e = dsx::build_plan(H, W, cfg, p);
  if (!e.empty()) {
    printf("{\"error\": \"%s\"}\n", e.c_str());
    return 1;
  }
  if (!print) return 0;
  printf("{\"H\": %d, \"W\": %d, \"Hout\": %d, \"Wout\": %d, \"L\": %d, \"plane_floats\": %lld, \"levels\": [", p.H, p.W,
         p.Hout, p.Wout, p.L, p.plane_floats);
  for (int l = 0; l < p.L; ++l) {
    const dsx::LevelPlan& lp = p.lv[l];
    printf("%s{\"h\": %d, \"w\": %d, \"ld\": %d, \"M\": %d, \"K\": %d, \"radix\": [", l ? ", " : "", lp.h, lp.w, lp.ld, lp.M,
           lp.K);
    for (int i = 0; i < lp.npass; ++i) printf("%s%d", i ? ", " : "", lp.radix[i]);
    printf("]}");
  }
  printf("]}\n");
  return 0;
}

// Emulate the spectral pipeline of k_rowfilter for two random rows at one level and compare with the
// exact operator  LP(x) = irfft_packed(rfft_packed(x) * e),  e[j] = exp(-j^2 / (2 s^2)).
static int cmd_rows(int H, int W, double sigma, int level, int lvl) {
  dsx::Plan p;
  if (cmd_plan(H, W, sigma, level, sigma, level, p, false)) return 1;
  if (lvl >= p.L) { printf("{\"error\": \"level out of range\"}\n"); return 1; }
  const dsx::LevelPlan& lp = p.lv[lvl];
  const int N = lp.w, M = lp.M, K = lp.K;
  const double s = lp.h * (sigma / (double)std::min(H, W));
  std::vector<double> xa(N), xb(N);
  srand(N * 7 + lvl);
  for (int n = 0; n < N; ++n) {
    xa[n] = rand() / (double)RAND_MAX - 0.3;
    xb[n] = rand() / (double)RAND_MAX - 0.6;
  }
  // exact reference operator in double: Y[k] = ep X[k] + em X[N-k]
  std::vector<double> ep, em;
  dsx::packed_gains(N, s, ep, em);
  auto exact = [&](const std::vector<double>& x) {
    std::vector<cd> X(N), Y(N);
    for (int k = 0; k < N; ++k) {
      cd acc = 0;
      for (int j = 0; j < N; ++j) acc += x[j] * std::polar(1.0, -2.0 * M_PI * (double)((long long)j * k % N) / N);
      X[k] = acc;
    }
    for (int k = 0; k < N; ++k) Y[k] = ep[k] * X[k] + em[k] * X[(N - k) % N];
    std::vector<double> y(N);
    for (int n = 0; n < N; ++n) {
      cd acc = 0;
      for (int k = 0; k < N; ++k) acc += Y[k] * std::polar(1.0, 2.0 * M_PI * (double)((long long)n * k % N) / N);
      y[n] = acc.real() / N;
    }
    return y;
  };
  std::vector<double> ya = exact(xa), yb = exact(xb);
  // kernel pipeline in float
  std::vector<dsx_c32> buf(M, dsx_mk(0.f, 0.f));
  for (int n = 0; n < N; ++n) {
    dsx_c32 z = dsx_mk((float)xa[n], (float)xb[n]);
    buf[K + n] = z;
    if (K > 0) {
      if (n <= K) buf[K + N + n] = z;
      if (n >= N - K) buf[n - (N - K)] = z;
    }
  }
  std::vector<dsx_c32> tw(M);
  for (int t = 0; t < M; ++t) tw[t] = dsx_mk(p.consts[lp.tw_off + t].re, p.consts[lp.tw_off + t].im);
  run_passes(buf, tw, lp.radix, lp.npass);
  const dsx::C32* g1 = &p.consts[lp.g_off[1]];
  const dsx::C32* g2 = g1 + M;
  std::vector<dsx_c32> v(M);
  for (int k = 0; k < M; ++k) {
    dsx_c32 u = buf[k], ur = buf[k == 0 ? 0 : M - k];
    dsx_c32 r = dsx_add(dsx_mul(dsx_mk(g1[k].re, g1[k].im), u), dsx_mul(dsx_mk(g2[k].re, g2[k].im), ur));
    v[k] = dsx_mk(r.y, r.x);
  }
  run_passes(v, tw, lp.radix, lp.npass);
  double err = 0, nrm = 0;
  for (int n = 0; n < N; ++n) {
    const double la = v[K + n].y / M, lb = v[K + n].x / M;
    err = fmax(err, fmax(fabs(la - ya[n]), fabs(lb - yb[n])));
    nrm = fmax(nrm, fmax(fabs(ya[n]), fabs(yb[n])));
  }
  printf("{\"N\": %d, \"M\": %d, \"K\": %d, \"rel_err\": %.3e}\n", N, M, K, err / nrm);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 3 && !strcmp(argv[1], "fft")) return cmd_fft(atoi(argv[2]));
  if (argc >= 8 && !strcmp(argv[1], "plan")) {
    dsx::Plan p;
    return cmd_plan(atoi(argv[2]), atoi(argv[3]), atof(argv[4]), atoi(argv[5]), atof(argv[6]), atoi(argv[7]), p, true);
  }
  if (argc >= 7 && !strcmp(argv[1], "rows"))
    return cmd_rows(atoi(argv[2]), atoi(argv[3]), atof(argv[4]), atoi(argv[5]), atoi(argv[6]));
  if (argc >= 2 && !strcmp(argv[1], "pruned")) return cmd_pruned();
  if (argc >= 2 && !strcmp(argv[1], "swz")) return cmd_swz();
  fprintf(stderr, "usage: pruned | swz | fft M | plan H W s0 l0 s1 l1 | rows H W sigma level lvl\n");
  return 2;
}
