// Mutation check of the chunk / plane codecs restated in csrc/dsx_io.h (Blosc frames, PNG scanline filters): random
// corruptions of a good frame must be rejected or decoded, never read or write out of bounds.  Built by
// tests/test_blosc.py with -fsanitize=address,undefined (CPU only) and run for a few seconds.
#include "../../aind_smartspim_destripe_amd/csrc/dsx_io.h"
#include <random>
#include <cstdio>
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  std::mt19937 rng(123);
  size_t n = 200000;
  std::vector<unsigned char> raw(n);
  for (size_t i = 0; i < n; ++i) raw[i] = (unsigned char)((i / 3) & 0xFF) ^ (unsigned char)(rng() % 4);
  std::vector<unsigned char> frame;
  std::string e = dsx::blosc_encode(raw.data(), n, 2, 3, true, frame);
  if (!e.empty()) { printf("encode: %s\n", e.c_str()); return 1; }
  std::vector<unsigned char> out(n);
  e = dsx::blosc_decode(frame.data(), frame.size(), out.data(), n);
  if (!e.empty() || out != raw) { printf("roundtrip failed %s\n", e.c_str()); return 1; }
  int errs = 0, oks = 0;
  for (int it = 0; it < iters; ++it) {
    std::vector<unsigned char> f = frame;
    int k = 1 + rng() % 4;
    for (int j = 0; j < k; ++j) {
      size_t pos = (rng() % 3 == 0) ? rng() % 64 : rng() % f.size();
      f[pos] = (unsigned char)rng();
    }
    if (rng() % 5 == 0) f.resize(rng() % f.size());
    std::string r = dsx::blosc_decode(f.data(), f.size(), out.data(), n);
    if (r.empty()) ++oks; else ++errs;
  }
  printf("mutations: %d rejected, %d decoded\n", errs, oks);
  // the same frames read as blosclz streams (inner codec 0), half of them with the bit-shuffle flag: the zstd bytes are
  // then a random blosclz instruction stream -- literal runs, near / far matches, overlong lengths, truncations
  int lz_errs = 0, lz_oks = 0;
  for (int it = 0; it < iters / 2; ++it) {
    std::vector<unsigned char> f = frame;
    f[2] = (unsigned char)((f[2] & 0x1F) | ((rng() & 1) ? 0x4 : 0x0));
    if (rng() % 4 == 0) f[3] = (unsigned char)(1 + rng() % 9);  // type size (split count, bit rows)
    int k = rng() % 4;
    for (int j = 0; j < k; ++j) f[16 + rng() % (f.size() - 16)] = (unsigned char)rng();
    if (rng() % 5 == 0) f.resize(16 + rng() % (f.size() - 16));
    std::string r = dsx::blosc_decode(f.data(), f.size(), out.data(), n);
    if (r.empty()) ++lz_oks; else ++lz_errs;
  }
  // and the stream decoder alone: random instruction streams into buffers of random size
  std::vector<unsigned char> lz(600), lzout(5000);
  for (int it = 0; it < iters; ++it) {
    const size_t len = rng() % lz.size(), cap = rng() % lzout.size();
    for (size_t i = 0; i < len; ++i) lz[i] = (rng() % 3 == 0) ? (unsigned char)(0xE0 | (rng() & 31)) : (unsigned char)rng();
    std::vector<unsigned char> in(lz.begin(), lz.begin() + len), o(cap);  // exact-size buffers: ASan sees one byte too many
    const size_t got = dsx::blosclz_decompress(in.data(), len, o.data(), cap);
    if (got > cap) { printf("blosclz produced more than its buffer\n"); return 1; }
  }
  printf("blosclz mutations: %d rejected, %d decoded\n", lz_errs, lz_oks);
  // png unfilter with random filter bytes
  std::vector<unsigned char> rows(100 * 301);
  for (int it = 0; it < 2000; ++it) {
    for (auto& b : rows) b = (unsigned char)rng();
    for (int y = 0; y < 100; ++y) rows[y * 301] = (unsigned char)(rng() % 6);
    (void)dsx::png_unfilter(rows.data(), 100, 300, 1 + rng() % 8);
  }
  printf("ok\n");
  return 0;
}
