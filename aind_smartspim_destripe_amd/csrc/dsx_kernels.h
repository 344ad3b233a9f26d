// dsx_kernels.h -- hand-written gfx950 kernels of the destripe path.
//
//   k_fwd_march  log(1+x) + one 2-D db3 analysis level, only the aa / da (= cH) subbands;
//                fused: fg/bg statistic (level 1) and min/max of cH^2          
//   k_hist       256-bin numpy-rule histogram of cH^2 per plane and level       (K2b)
//   k_otsu       config decision + Otsu arg-max + threshold                      (K3)
//   k_rowfilter  mask, exact row median, in-paint, FFT low-pass with the packed-index gain
//                quirk, Delta = -(1 - mask) LP(inpainted); two rows per complex FFT (K4)
//   k_inv_march  one synthesis level of the Delta pyramid; the last level fuses
//                (1 + x) exp(c0) + 1, flat/dark correction and the output cast    (K5, K6)
//
// Reference semantics: /root/reference/code/aind_smartspim_destripe/filtering.py:139-224
// (log_space_fft_filtering), :54-88 (fg/bg statistic), :338-414 (flatfield_correction).
#ifndef DSX_KERNELS_H
#define DSX_KERNELS_H

#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsx_fft_core.h"

// ---- build-time switches (defaults = the product; tools/build_variant.sh builds A/B variants with -D...) ----------
// Streaming (non-temporal) accesses for intermediates that the next kernel reads only after hundreds of MB of
// other traffic (da_1 out of the forward kernel, cH into the histogram, Delta out of the row filter).
// (Not for the Delta_1 loads of the final kernel: +2 % there.)
#ifndef DSX_NT
#define DSX_NT 1
#endif
#ifndef DSX_MEDIAN_BALLOT
#define DSX_MEDIAN_BALLOT 1
#endif
#ifndef DSX_FWD_STEADY
#define DSX_FWD_STEADY 1  // bit 0: steady-state row loop in interior strips, bit 1: in edge strips too (spills)
#endif
#ifndef DSX_RF_XCD
// k_rowfinal: blocks that share rows on the same compute die.  Measured and left OFF: 1.6 MB per plane less HBM traffic
// (25.0 -> 23.4), but 4 % SLOWER (61.9 k against 64.5 k planes/s, profiles/r3_fused_rowfinal_ab.txt): with runs of blocks
// per die the eight dies stream eight distant regions of a plane instead of one front.
#define DSX_RF_XCD 0
#endif
#ifndef DSX_INV_MINW
#define DSX_INV_MINW 1  // waves per SIMD the fused uint16 final kernel is compiled for
#endif
#ifndef DSX_FWD_PRUNE
// level-1 row filter of 2048-wide planes: forward passes in the order 6, 9, 19 with the last (radix-19) pass pruned to the
// output pairs inside the low-pass band (StaticFft<1>::run_forward); 0 = the order 19, 9, 6, every bin computed
#define DSX_FWD_PRUNE 1
#endif
#ifndef DSX_SWZ
// bank-swizzled LDS addressing (dsx_fft_core.h: dsx_idx_swz) for the power-of-two row-filter plans 2 048 = 16 * 16 * 8 and
// 1 024 = 16 * 8 * 8 (levels 1 and 2 of 2000-wide planes); 0 = plain addressing, 2 = the swizzle through the generic index
// policy instead of the written-out addressing (A/B builds)
#define DSX_SWZ 1
#endif
#ifndef DSX_FWD_MINW
#define DSX_FWD_MINW 4  // waves per SIMD the fused uint16 forward kernel is compiled for (register cap 128)
#endif


// Timing-only switches (DSX_ABLATE bits: kernels skip a phase or a store path and return WRONG pixels) exist in
// -DDSX_DIAG builds only (tools/build_variant.sh); in the product build the tests fold to `false`.
#ifdef DSX_DIAG
#define DSX_ABL(a, bits) (((a).ablate & (bits)) != 0)
#else
#define DSX_ABL(a, bits) false
#endif

namespace dsx {

constexpr int kMaxLevels = 16;
constexpr int kMaxPasses = 16;
constexpr int kWave = 64;

// db3 filter bank (PyWavelets Wavelet('db3')); rec_lo = reversed dec_lo, rec_hi = reversed dec_hi
#define DSX_DEC_LO                                                                              \
  {0.03522629188570953f, -0.08544127388202666f, -0.13501102001025458f, 0.45987750211849154f,   \
   0.8068915093110925f,  0.33267055295008263f}
#define DSX_DEC_HI                                                                              \
  {-0.33267055295008263f, 0.8068915093110925f,  -0.45987750211849154f, -0.13501102001025458f,  \
   0.08544127388202666f,  0.03522629188570953f}
#define DSX_REC_LO                                                                              \
  {0.33267055295008263f,  0.8068915093110925f,   0.45987750211849154f, -0.13501102001025458f,  \
   -0.08544127388202666f, 0.03522629188570953f}
#define DSX_REC_HI                                                                              \
  {0.03522629188570953f, 0.08544127388202666f, -0.13501102001025458f, -0.45987750211849154f,   \
   0.8068915093110925f,  -0.33267055295008263f}

typedef unsigned dsx_u32x2 __attribute__((ext_vector_type(2)));
typedef float dsx_f2 __attribute__((ext_vector_type(2)));  // arithmetic on these is packed FP32 (v_pk_fma_f32 ...)
typedef float dsx_f4 __attribute__((ext_vector_type(4)));
typedef unsigned dsx_u32x4 __attribute__((ext_vector_type(4)));

// Raw buffer resource over [p, p + 4 GB): the marching kernels address rows as "descriptor (SGPRs) + wave-uniform row
// offset (one SGPR) + per-lane byte offset (one VGPR)".  With plain pointers the compiler hoists "base + lane offset"
// out of the row loop as a 64-bit VGPR pair per stream and adds the row offset with 64-bit vector adds -- registers
// and issue slots the loop does not have.  (word 3 = 0x00020000: 32-bit data format, no swizzle, gfx9 layout.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dsx_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, -1, 0x00020000);
}
constexpr int kBufNT = 2;  // aux bits of a streaming (non-temporal) buffer access on gfx94x / gfx950

// Packed multiply-add with a FIXED evaluation order.  (a * b + c * d + e * f written with operators leaves
// the choice of which product is rounded first to the compiler, and two inlined copies of one expression --
// a segment's prologue and its steady-state loop -- did pick differently: results then depended on where
// the row segments of a launch start, i.e. on the batch size.)
#if DSX_SCALAR_FMA
__device__ __forceinline__ dsx_f2 pk_fma(dsx_f2 a, dsx_f2 b, dsx_f2 c) { return dsx_f2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ dsx_f2 pk_fma(dsx_f2 a, float b, dsx_f2 c) { return dsx_f2{fmaf(a.x, b, c.x), fmaf(a.y, b, c.y)}; }
__device__ __forceinline__ dsx_f2 pk_fma(float a, dsx_f2 b, dsx_f2 c) { return dsx_f2{fmaf(a, b.x, c.x), fmaf(a, b.y, c.y)}; }
#else
__device__ __forceinline__ dsx_f2 pk_fma(dsx_f2 a, dsx_f2 b, dsx_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ dsx_f2 pk_fma(dsx_f2 a, float b, dsx_f2 c) {
  const dsx_f2 bb = {b, b};
  return __builtin_elementwise_fma(a, bb, c);
}
__device__ __forceinline__ dsx_f2 pk_fma(float a, dsx_f2 b, dsx_f2 c) {
  const dsx_f2 aa = {a, a};
  return __builtin_elementwise_fma(aa, b, c);
}
#endif

struct PlaneStats {
  double sum_fg;               // sum of pixels in the foreground class (>= cut-off)
  double sum_all;              // sum of all pixels
  unsigned long long cnt_fg;   // pixels in the foreground class
  // bit 0: a float32 pixel was NaN, infinite or <= -1 (log(1 + x) is not finite: the reference dies in
  // numpy.histogram, filtering.py:188 -> skimage threshold_otsu); dsx_run_host turns it into DSX_EVALUE, the
  // device-buffer entry points fold it into the context's sticky flag word (k_otsu), read by dsx_sync & co.
  unsigned long long flags;
  unsigned pad_[56];
};
static_assert(sizeof(PlaneStats) == 256, "control block layout");

// Half-sample symmetric extension index (np.pad 'symmetric'), any distance.
__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i >= 0 && i < n) return i;
  int p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return i < n ? i : p - 1 - i;
}

__device__ __forceinline__ float as_f32(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ unsigned as_u32(float f) { return __float_as_uint(f); }

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_min_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}

// min of two floats that are never NaN, as ONE v_min_f32.  fminf() on a value that comes out of memory (or a select of
// one) makes the compiler quiet a possible signalling NaN first (v_max_f32 x, x, x per operand: 94 of them in the median's
// last pass before this).
__device__ __forceinline__ float min_no_nan(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max_no_nan(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// Wave minimum of values that are never NaN; the result is wave-uniform.  Four DPP steps inside the rows of 16 lanes
// (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: every lane then holds its row's minimum), then the four
// rows through v_readlane -- no LDS crossbar (__shfl_xor is ds_bpermute_b32 + a wait per step).
__device__ __forceinline__ float wave_min_no_nan(float v) {
  v = min_no_nan(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = min_no_nan(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = min_no_nan(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  v = min_no_nan(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fminf(fminf(r0, r1), fminf(r2, r3));
}

// Compiler-level + wavefront-scope ordering of LDS traffic inside ONE wave (no s_barrier):
// the hardware executes a wave's LDS instructions in order; this keeps the compiler from
// forwarding or reordering accesses across the point where other lanes' data is exchanged.
// value of the neighbouring lane (lane ^ 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ unsigned swap_adjacent(unsigned v) {
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}

// Bit e of a slot mask as a select mask (all ones / zero: one v_bfe_i32) and the select itself as one v_bfi_b32.  Written
// as "bit ? a : b" the compiler emits v_and + v_cmp + v_cndmask: 13.8 cycles of a SIMD per select against 6.2
// (tools/micro/valu_rate.hip).
template <typename M>
__device__ __forceinline__ int slot_mask(M m, int e) {
  const unsigned w = (sizeof(M) == 8 && e >= 32) ? (unsigned)((unsigned long long)m >> 32) : (unsigned)m;
  return (int)(w << (31 - (e & 31))) >> 31;
}
// An opaque copy of a slot mask: the select masks of a later stage are then extracted again (one instruction each) instead
// of being kept in registers across the transforms in between (36 of them).
template <typename M>
__device__ __forceinline__ M opaque_mask(M m) {
  if constexpr (sizeof(M) == 8) {
    unsigned lo = (unsigned)m, hi = (unsigned)((unsigned long long)m >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return (M)(((unsigned long long)hi << 32) | lo);
  } else {
    unsigned lo = (unsigned)m;
    asm volatile("" : "+v"(lo));
    return (M)lo;
  }
}
// (as assembly: from the C expression the compiler keeps ~m in a register and emits v_and + v_and_or, or reads a uniform
//  operand from an SGPR -- 6.2 / 5.5 cycles where this is 3.1)
__device__ __forceinline__ unsigned bit_select(int m, unsigned a, unsigned b) {
  unsigned r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
}
// all ones on odd lanes, zero on even ones, opaque to the optimiser (see slot_mask): the lane-pair exchanges select with it
__device__ __forceinline__ int odd_lane_mask(int lane) {
  int m = -(lane & 1);
  asm("" : "+v"(m));
  return m;
}
__device__ __forceinline__ float bit_select(int m, float a, float b) {  // m ? a : b
  return as_f32(bit_select(m, as_u32(a), as_u32(b)));
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef DSX_SCALAR_FMA
#define DSX_SCALAR_FMA 1  // the multiply-adds of pk_fma / the complex helpers as scalar v_fma_f32 (see __graft_entry__.build)
#endif
#ifndef DSX_FWD_PAIR
#define DSX_FWD_PAIR 1  // interior strips of the fused uint16 forward kernel load pixel rows by lane pairs (16 bytes per lane)
#endif
#ifndef DSX_FWD_SPLIT
#define DSX_FWD_SPLIT 1
#endif
#ifndef DSX_DPP_X
#define DSX_DPP_X 1
#endif
// Value of the same register in lane + 1 (DPP wave_shl:1); lane 63 reads 0.  All 64 lanes must be active.
__device__ __forceinline__ float lane_above(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}

// One launch instead of three hipMemsetAsync calls (each ~27 us of fill-kernel + launch latency) to zero
// the control block of a cohort part: plane statistics, min/max words, histograms (16-byte units).
__global__ __launch_bounds__(256) void k_zero3(uint4* p0, int n0, uint4* p1, int n1, uint4* p2, int n2) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n0 + n1 + n2; i += gridDim.x * 256) {
    if (i < n0) p0[i] = z;
    else if (i < n0 + n1) p1[i - n0] = z;
    else p2[i - n0 - n1] = z;
  }
}

// ================================================================================================
// K1: forward transform level, "marching" form
//
// One WAVE owns a strip of 256 input columns (4 per lane, one 8/16-byte load per lane and row) and
// streams down the rows with a 6-row sliding window in registers, so the column (axis-0) filter
// never touches LDS; only the freshly filtered row pair (a0, d0) goes through a 2 KiB per-wave LDS
// row to be low-passed along the row (axis 1).  No block barriers, no input halo re-reads along y.
// Strips advance by 252 input columns = 126 coefficient columns.
// ================================================================================================
struct Fwd1Args {
  const void* in;            // IN_KIND 0/1: pixels [B][H][W]
  long long in_plane_stride;
  float* ws;
  long long ws_plane_stride;
  long long in_off;          // IN_KIND 2: offset of aa_{l-1} inside a plane's workspace
  int H, W, ldin;            // input rows, valid columns, row pitch (elements)
  long long aa_off, da_off;
  int h, w, ld, lda;  // coefficient shape, pitch of da, pitch of aa (with extension margins)
  unsigned* minmax;  // [B][L][2]
  int lvl, L;
  PlaneStats* stats;
  float fg_cutoff;
  unsigned fg_cutoff_u16;  // integer pixels: pixel >= fg_cutoff_u16  <=>  (float)pixel >= fg_cutoff
  int nstrips, nseg, rows_per_seg;
  // FUSE (k_fwd_march<IN_KIND, true>): the level-2 analysis runs in the same wave, aa_1 never leaves the
  // chip.  Strips are then counted in level-2 columns (kFuseOut per wave), segments in level-2 rows.
  long long aa2_off, da2_off;
  int h2, w2, ld2, lda2;
  int ablate;  // diagnosis only (DSX_ABLATE): 16 = the fused forward kernel stores nothing (timing of its store path)
  // stack mode (the reference's 3-D input, filtering.py:188): ONE Otsu threshold per level for all planes of the call --
  // every plane's min / max and histogram go to plane 0's slots (k_hist, k_otsu read them there too)
  int shared;
};

constexpr int kMarchCols = 256;                 // input columns per wave
constexpr int kMarchOut = (kMarchCols - 4) / 2;  // 126 coefficient columns per wave
constexpr int kFuseOut = (kMarchOut - 4) / 2;    // 61 level-2 columns per wave of the fused kernel
constexpr int kRingRows = 8;                     // aa_1 rows a wave keeps in LDS (6 are needed)
constexpr int kRingPitch = 128;                  // floats per ring row (126 columns + margin positions)
constexpr int kX2Pitch = 136;                    // (lo, hi) pairs of the level-2 row exchange (reads run to 2*63+5)

template <int IN_KIND>
struct MarchStats {
  // uint16 pixels: exact integer partial sums (a lane sees < 2^15 pixels of < 2^16);
  // float32 pixels: double partial sums (exact for integer-valued pixels)
  unsigned cnt = 0;
  unsigned isum_all = 0, isum_fg = 0;
  double fsum_all = 0.0, fsum_fg = 0.0;
  unsigned bad = 0;  // float32 pixels: a pixel whose log(1 + x) is not finite was seen
  __device__ __forceinline__ void add(float f, float cutoff) {
    if (IN_KIND != 0) bad |= (f > -1.0f && f <= 3.402823466e38f) ? 0u : 1u;  // false for NaN as well
    if (IN_KIND == 0) {
      const unsigned u = (unsigned)f;
      isum_all += u;
      if (f >= cutoff) { isum_fg += u; cnt++; }
    } else {
      fsum_all += (double)f;
      if (f >= cutoff) { fsum_fg += (double)f; cnt++; }
    }
  }
};

// Raw (not yet converted) values of one row for this lane's 4 columns.
struct MarchRaw {
  float4 f;  // IN_KIND 1/2: 4 x float32; IN_KIND 0: .x/.y carry the bits of 4 x uint16
};

// Column addressing of a lane: 4 consecutive columns gc0 .. gc0+3.
struct MarchCol {
  int gc0;
  int base;   // first column of the vector load
  bool vec;   // one aligned 8/16-byte load (possibly of the mirrored group, see rev)
  bool rev;   // the group lies entirely in the symmetric extension: mirrored aligned group, reversed
  bool dead;  // no output of this wave depends on these columns: nothing is loaded
  bool own;   // this lane accounts these pixels in the fg/bg statistic
  unsigned own_mask[2];  // uint16 planes: AND masks of the owned elements {px0|px1<<16, px2|px3<<16}
  int rc[4];  // reflected column per element (scalar path)
};

// ext: the source rows store the symmetric extension for columns [-4, W + 8) (aa_{l-1} buffers)
__device__ __forceinline__ MarchCol march_col(int gc0, int W, int ld, int w_out, bool lane_owns, bool ext) {
  MarchCol c;
  c.gc0 = gc0;
  c.own = lane_owns;
  c.dead = gc0 > 2 * w_out - 1;  // coefficient j needs input columns 2j-4 .. 2j+1 only
  c.rev = false;
  c.base = gc0;
  const bool aligned = (ld & 3) == 0;
  c.vec = aligned && gc0 >= 0 && gc0 + 3 < W;
  if (ext && aligned && gc0 >= -4 && gc0 + 3 < W + 8) c.vec = true;
  if (!c.vec && !ext && aligned && (W & 3) == 0) {
    // W % 4 == 0: a group never straddles the edge, the mirrored group is aligned as well
    if (gc0 + 3 < 0 && -gc0 - 1 < W) { c.base = -gc0 - 4; c.rev = true; c.vec = true; }
    else if (gc0 >= W && 2 * W - 4 - gc0 >= 0) { c.base = 2 * W - 4 - gc0; c.rev = true; c.vec = true; }
  }
  if (c.dead) { c.base = 0; c.rev = false; }
  c.own_mask[0] = c.own_mask[1] = 0u;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    c.rc[e] = reflect_idx(gc0 + e, W);
    if (lane_owns && gc0 + e >= 0 && gc0 + e < W) c.own_mask[e >> 1] |= (e & 1) ? 0xFFFF0000u : 0x0000FFFFu;
  }
  return c;
}

// Source rows of a wave: pointer for the element-wise path, buffer descriptor + lane byte offset for FAST waves.
// rs is based BIAS elements before src (aa_{l-1} rows carry 4 margin columns on the left, c.base >= -4), so
// that the lane offset is unsigned.
template <int IN_KIND>
struct MarchSrc {
  static constexpr int ES = (IN_KIND == 0) ? 2 : 4;
  static constexpr int BIAS = (IN_KIND == 2) ? 4 : 0;
  const void* src;
  __amdgpu_buffer_rsrc_t rs;
  unsigned voff;
  __device__ __forceinline__ MarchSrc(const void* p, const MarchCol& c)
      : src(p), rs(dsx_rsrc((const char*)p - BIAS * ES)), voff((unsigned)(c.base + BIAS) * ES) {}
  // one row of a FAST wave; soff = row * pitch in bytes (wave-uniform)
  __device__ __forceinline__ MarchRaw load(unsigned soff) const {
    MarchRaw r;
    if (IN_KIND == 0) {
      const dsx_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
      r.f = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);
    } else {
      const dsx_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
      r.f = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
    }
    return r;
  }
};

// Issue the loads of one row (prefetch).  all_vec (wave-uniform): every lane of the wave is either
// vector-loadable or dead, so the load is unconditional and branch-free (a per-lane branch around
// a prefetch load makes the compiler drain it before the other path may write the registers).
template <int IN_KIND, bool FAST>
__device__ __forceinline__ MarchRaw march_issue(const MarchSrc<IN_KIND>& ms, int ld, int H, int gr_raw,
                                                const MarchCol& c) {
  MarchRaw r;
  if (FAST)  // every lane of the wave is vector-loadable (or dead, with a valid dummy address)
    return ms.load((unsigned)(reflect_idx(gr_raw, H) * ld) * MarchSrc<IN_KIND>::ES);
  const void* src = ms.src;
  const long long row = (long long)reflect_idx(gr_raw, H) * ld;
  r.f = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c.dead) return r;
  if (IN_KIND == 0) {
    const uint16_t* p = (const uint16_t*)src + row;
    if (c.vec) {
      const uint2 u = *(const uint2*)(p + c.base);
      r.f.x = __uint_as_float(u.x);
      r.f.y = __uint_as_float(u.y);
    } else {
      const unsigned u0 = p[c.rc[0]], u1 = p[c.rc[1]], u2 = p[c.rc[2]], u3 = p[c.rc[3]];
      r.f.x = __uint_as_float(u0 | (u1 << 16));
      r.f.y = __uint_as_float(u2 | (u3 << 16));
    }
  } else {
    const float* p = (const float*)src + row;
    if (c.vec) r.f = *(const float4*)(p + c.base);
    else r.f = make_float4(p[c.rc[0]], p[c.rc[1]], p[c.rc[2]], p[c.rc[3]]);
  }
  return r;
}

// Six consecutive raw rows g0 .. g0+5 (one prefetch group) of a FAST wave.  Interior groups -- all but the
// first / last of a plane -- take one row offset per group and add the pitch per row: the row index arithmetic
// (symmetric reflection = an integer division on the scalar unit) was as many scalar instructions per step as
// the kernel has vector ones.
template <int IN_KIND>
__device__ __forceinline__ void march_issue6(const MarchSrc<IN_KIND>& ms, int ld, int H, int g0, const MarchCol& c,
                                             MarchRaw (&out)[6]) {
  constexpr int ES = MarchSrc<IN_KIND>::ES;
  if (g0 >= 0 && g0 + 5 < H) {  // wave-uniform
    const unsigned pitch = (unsigned)ld * ES;
    unsigned soff = (unsigned)g0 * pitch;
#pragma unroll
    for (int r = 0; r < 6; ++r, soff += pitch) out[r] = ms.load(soff);
  } else {
#pragma unroll
    for (int r = 0; r < 6; ++r) out[r] = march_issue<IN_KIND, true>(ms, ld, H, g0 + r, c);
  }
}

// Convert a raw row.  Pixel planes: fg/bg statistic on owned pixels, then log2(1 + x) with the bare
// v_log_f32 (inputs are >= 1; the ln 2 factor is folded into the axis-0 filter taps).  aa_{l-1}: identity.
template <int IN_KIND, bool INSIDE = false>
__device__ __forceinline__ void march_consume(const Fwd1Args& a, const MarchRaw& r, int gr_raw, const MarchCol& c,
                                              bool row_in_seg, bool any_rev, MarchStats<IN_KIND>& st,
                                              float (&x)[4]) {
  // INSIDE: the caller knows 0 <= gr_raw < H (steady-state rows of the fused kernel)
  if (IN_KIND == 0) {
    unsigned u0 = __float_as_uint(r.f.x), u1 = __float_as_uint(r.f.y);
    if (any_rev) {  // wave-uniform (scalar branch): only edge strips hold mirrored groups
      if (c.rev) {  // mirrored group: reverse the four 16-bit elements
        const unsigned t = u0;
        u0 = (u1 >> 16) | (u1 << 16);
        u1 = (t >> 16) | (t << 16);
      }
    }
    if (row_in_seg && (INSIDE || (gr_raw >= 0 && gr_raw < a.H))) {  // wave-uniform: this row is accounted by this segment
      // ownership per element as AND masks (zeroed pixels add nothing and are below the cut-off)
      const unsigned o0 = u0 & c.own_mask[0], o1 = u1 & c.own_mask[1];
      // sum of the four pixels by two SADs against zero; foreground pixels (>= 384) are rare: a packed
      // 16-bit max decides whether the per-pixel path is needed at all (integer pixels: f >= cut-off
      // <=> pixel >= ceil(cut-off))
      st.isum_all = __builtin_amdgcn_sad_u16(o1, 0u, __builtin_amdgcn_sad_u16(o0, 0u, st.isum_all));
      typedef unsigned short dsx_u16x2 __attribute__((ext_vector_type(2)));
      union { unsigned u; dsx_u16x2 v; } pa, pb, pm;
      pa.u = o0; pb.u = o1;
      pm.v = __builtin_elementwise_max(pa.v, pb.v);
      const unsigned m = max(pm.u & 0xFFFFu, pm.u >> 16);
      if (m >= a.fg_cutoff_u16) {
        const unsigned px[4] = {o0 & 0xFFFFu, o0 >> 16, o1 & 0xFFFFu, o1 >> 16};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (px[e] >= a.fg_cutoff_u16) { st.isum_fg += px[e]; st.cnt++; }
      }
    }
    // 1 + x as a packed add on the converted pairs (one v_cvt with a 16-bit source select per pixel + half a v_pk_add;
    // written with scalar adds the compiler adds 1 in the integer domain first: two instructions per pixel)
    const dsx_f2 one = {1.0f, 1.0f};
    const dsx_f2 p01 = dsx_f2{(float)(u0 & 0xFFFFu), (float)(u0 >> 16)} + one;
    const dsx_f2 p23 = dsx_f2{(float)(u1 & 0xFFFFu), (float)(u1 >> 16)} + one;
    x[0] = __builtin_amdgcn_logf(p01.x);
    x[1] = __builtin_amdgcn_logf(p01.y);
    x[2] = __builtin_amdgcn_logf(p23.x);
    x[3] = __builtin_amdgcn_logf(p23.y);
    return;
  }
  float v[4] = {r.f.x, r.f.y, r.f.z, r.f.w};
  if (any_rev && c.vec && c.rev) {
    const float t0 = v[0], t1 = v[1];
    v[0] = v[3]; v[1] = v[2]; v[2] = t1; v[3] = t0;
  }
  if (IN_KIND == 2) {
    x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3];
    return;
  }
  const bool own_row = row_in_seg && c.own && (INSIDE || (gr_raw >= 0 && gr_raw < a.H));
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (own_row && c.gc0 + e >= 0 && c.gc0 + e < a.W) st.add(v[e], a.fg_cutoff);
    x[e] = __builtin_amdgcn_logf(1.0f + v[e]);
  }
}

// Body of k_fwd_march for one wave.  FAST: every lane's column group is vector-loadable, the loads
// are unconditional (two instantiations instead of one: if the scalar path shared the registers of
// the prefetch loads the compiler would drain vmcnt before every one of them).
// Geometry of a wave of the fused kernel.  A strip produces kFuseOut level-2 columns starting at o2; the
// LAST strip is shifted left so that it is full (it recomputes columns the strip before it owns: the
// mirror sources of the right-hand extension are then always inside the strip).  Ownership -- who
// stores a coefficient, who accounts a pixel in the statistic -- stays a partition:
// strip s owns level-2 columns [61 s, ..), level-1 columns [122 s, ..), pixels [244 s, ..).
struct FuseGeom {
  int o2;        // first level-2 column of the wave
  int j0;        // first level-1 column (lane 0): 2 o2 - 4, may be -4
  int own2_lo;   // level-2 columns >= own2_lo are stored by this strip
  int own1_lo, own1_hi;  // level-1 (da_1) columns stored by this strip
};
__device__ __forceinline__ FuseGeom fuse_geom(const Fwd1Args& a, int strip) {
  FuseGeom g;
  g.o2 = min(kFuseOut * strip, max(0, a.w2 - kFuseOut));
  g.j0 = 2 * g.o2 - 4;
  g.own2_lo = kFuseOut * strip;
  g.own1_lo = 2 * kFuseOut * strip;
  g.own1_hi = (strip == a.nstrips - 1) ? a.w : 2 * kFuseOut * (strip + 1);
  return g;
}

// EDGE = false (fused kernel only): the strip touches neither plane edge and is not the (shifted) last one -- no
// mirrored column groups, no margin columns to write, every level-1 / level-2 column of the wave exists, and the
// ownership ranges start and end on even columns.  The row loop of such a strip has no per-lane branches left.
template <int IN_KIND, bool FAST, bool FUSE, bool EDGE = true>
__device__ __forceinline__ void fwd_march_body(const Fwd1Args& a, float (*s_row)[2][2][kMarchCols / 2],
                                               float (*s_ring)[kRingRows][kRingPitch], float2 (*s_x2)[kX2Pitch],
                                               int lane, int wave, int strip, int seg, int plane,
                                               const MarchCol& col, bool any_rev_in) {
  const bool any_rev = EDGE && any_rev_in;
  constexpr float LO[6] = DSX_DEC_LO;
  constexpr float HI[6] = DSX_DEC_HI;
  // pixel planes enter as log2(1 + x): the axis-0 taps carry the factor ln 2
  constexpr float KS = (IN_KIND == 2) ? 1.0f : 0.69314718055994530942f;
  constexpr float LOV[6] = {LO[0] * KS, LO[1] * KS, LO[2] * KS, LO[3] * KS, LO[4] * KS, LO[5] * KS};
  constexpr float HIV[6] = {HI[0] * KS, HI[1] * KS, HI[2] * KS, HI[3] * KS, HI[4] * KS, HI[5] * KS};
  // FUSE: the segment is counted in level-2 rows [i2b, i2e); it runs the level-1 rows those need
  // (4 rows of overlap with the segment above, which owns them) and owns level-1 rows >= own_row_lo
  const FuseGeom fg = FUSE ? fuse_geom(a, strip) : FuseGeom();
  const int i2b = seg * a.rows_per_seg;
  const int i2e = FUSE ? min(a.h2, i2b + a.rows_per_seg) : 0;
  // The last level-2 rows of a plane mirror into level-1 rows h - 5 ... h - 1 (half-sample symmetric extension): a LAST
  // segment of one or two level-2 rows starts above them as well, or its ring would not hold them (round 3: heights like
  // 2390, 601 level-2 rows in 25-row segments + 1; tests::test_short_last_march_segment)
  const int i_first = (FUSE && i2e == a.h2) ? min(2 * i2b - 4, (a.h - 6) & ~1) : 2 * i2b - 4;
  const int i_begin = FUSE ? max(0, i_first) : seg * a.rows_per_seg;
  const int i_end = FUSE ? min(a.h, 2 * i2e) : min(a.h, i_begin + a.rows_per_seg);
  const int own_row_lo = FUSE ? 2 * i2b : i_begin;
  // steady-state level-1 rows [steady_lo, steady_hi): owned, raw rows 2i, 2i+1 inside the plane, level-2 source rows
  // 2 i2 - 4 .. 2 i2 + 1 all present without reflection
  const int steady_lo = max(own_row_lo, 6);
  const int steady_hi = min(min(i_end, a.h - 2), (a.H - 2) >> 1);
  const int j0 = FUSE ? fg.j0 : kMarchOut * strip;
  int next2 = i2b;  // next level-2 row to emit
  float q2min = __builtin_huge_valf(), q2max = 0.f;
  float* aa2 = FUSE ? a.ws + plane * a.ws_plane_stride + a.aa2_off : nullptr;
  float* da2 = FUSE ? a.ws + plane * a.ws_plane_stride + a.da2_off : nullptr;
  const void* src;
  if (IN_KIND == 0) src = (const uint16_t*)a.in + plane * a.in_plane_stride;
  else if (IN_KIND == 1) src = (const float*)a.in + plane * a.in_plane_stride;
  else src = a.ws + plane * a.ws_plane_stride + a.in_off;
  const MarchSrc<IN_KIND> ms(src, col);
  // PAIRLD (interior strips of the fused uint16 kernel): pixel rows are read by lane PAIRS -- the even lane loads 16 bytes
  // (the pair's 8 columns) of raw row g, the odd lane those of row g + 1, and the halves are exchanged over the DPP
  // crossbar when the group is consumed: half as many load instructions, and 16-byte accesses (8-byte ones top out at
  // 3.9 TB/s on this chip, which is where this kernel ran alone; tools/bw_access_width.py, inv_march_body<.., PAIR>)
  constexpr bool PAIRLD = DSX_FWD_PAIR && FUSE && FAST && !EDGE && IN_KIND == 0;
  const bool odd_lane = (lane & 1) != 0;
  const int oddm = odd_lane_mask(lane);
  const unsigned pitch_b = (unsigned)a.ldin * 2u;
  const unsigned voff_pair0 = (unsigned)(col.base - (odd_lane ? 4 : 0)) * 2u;   // the pair's first column (interior: >= 0)
  const unsigned voff_pair = voff_pair0 + (odd_lane ? pitch_b : 0u);
  auto issue_pairs = [&](int g0, MarchRaw (&out)[6]) {  // raw rows g0 .. g0 + 5 as three pair loads -> out[0], out[2], out[4]
    if (g0 >= 0 && g0 + 5 < a.H) {  // wave-uniform
      unsigned soff = (unsigned)g0 * pitch_b;
#pragma unroll
      for (int p = 0; p < 3; ++p, soff += 2u * pitch_b) {
        const dsx_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(ms.rs, voff_pair, soff, 0);
        out[2 * p].f = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
      }
    } else {  // first / last groups of a plane: reflected rows, one row offset per lane
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const int re = reflect_idx(g0 + 2 * p, a.H), ro = reflect_idx(g0 + 2 * p + 1, a.H);
        const dsx_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(ms.rs, voff_pair0 + (unsigned)(odd_lane ? ro : re) * pitch_b, 0u, 0);
        out[2 * p].f = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
      }
    }
  };
  float* aa = a.ws + plane * a.ws_plane_stride + a.aa_off;
  float* da = a.ws + plane * a.ws_plane_stride + a.da_off;
  // fused kernel: coefficient rows are stored through buffer descriptors (row offset in an SGPR)
  const __amdgpu_buffer_rsrc_t rs_da = dsx_rsrc(da), rs_aa2 = dsx_rsrc(aa2), rs_da2 = dsx_rsrc(da2);
  float2* sE[2] = {(float2*)s_row[wave][0][0], (float2*)s_row[wave][1][0]};
  float2* sO[2] = {(float2*)s_row[wave][0][1], (float2*)s_row[wave][1][1]};

  MarchStats<IN_KIND> st;
  float win[6][4];  // sliding window: rows 2i-4 .. 2i+1 of this lane's 4 columns
  // prologue: rows 2 i_begin - 4 .. 2 i_begin - 1 (owned by the previous segment / outside the plane)
  {
    MarchRaw pr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pr[r] = march_issue<IN_KIND, FAST>(ms, a.ldin, a.H, 2 * i_begin - 4 + r, col);
#pragma unroll
    for (int r = 0; r < 4; ++r) march_consume<IN_KIND>(a, pr[r], 2 * i_begin - 4 + r, col, false, any_rev, st, win[r]);
  }

  float qmin = __builtin_huge_valf(), qmax = 0.f;
  const int jj0 = 2 * lane;  // this lane's two output columns (lanes 0..62)
  const bool out_lane = lane < kMarchOut / 2;
  const bool edge_strip = EDGE && ((j0 <= 0) || (j0 + kMarchOut >= a.w - 8));

  // FUSE: everything about a lane's columns is loop-invariant -- predicates and byte offsets are taken
  // once, so that the row loop carries no per-iteration compares and its stores are "uniform row base +
  // 32-bit lane offset" (no 64-bit vector address arithmetic)
  const int jl = j0 + jj0;                                   // level-1 columns jl, jl + 1 of this lane
  const bool c_v0 = FUSE && (!EDGE || (jl >= 0 && jl < a.w)), c_v1 = FUSE && (!EDGE || (jl + 1 >= 0 && jl + 1 < a.w));
  const bool c_s0 = FUSE && out_lane && jl >= fg.own1_lo && jl < fg.own1_hi;
  const bool c_s1 = EDGE ? (FUSE && out_lane && jl + 1 >= fg.own1_lo && jl + 1 < fg.own1_hi) : c_s0;
  // DSX_ABLATE bit 64 (diagnosis, wrong results): every strip's da_1 row piece starts on a 512-byte boundary of the
  // row -- what the store path would gain from aligned strip boundaries (they sit at multiples of 488 bytes)
  const unsigned off_da = (unsigned)max(jl, 0) * 4u + ((FUSE && DSX_ABL(a, 64)) ? 24u * (unsigned)strip : 0u);
  const int jo2 = fg.o2 + lane;                              // level-2 column of this lane
  const bool l2_valid = FUSE && lane < kFuseOut && (!EDGE || jo2 < a.w2);
  const bool l2_store = l2_valid && jo2 >= fg.own2_lo;
  const bool l2_edge = FUSE && EDGE && (fg.o2 == 0 || fg.o2 + kFuseOut >= a.w2 - 8);  // wave-uniform
  const unsigned off_2 = (unsigned)jo2 * 4u;

  // ---- FUSE: one level-2 row from the aa_1 ring (rows 2 i2 - 4 .. 2 i2 + 1, half-sample symmetric) ----
  // steady_c (std::true_type): all six source rows exist (2 i2 - 4 >= 0, 2 i2 + 1 < h)
  auto l2_step = [&](auto steady_c, int i2) {
    constexpr bool STEADY = decltype(steady_c)::value;
    dsx_f2 c[6];
#if DSX_DPP_X
    if (EDGE) wave_sync();  // edge strips: the ring's mirrored columns were written by other lanes
#endif
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      int r = 2 * i2 - 4 + k;
      if (!STEADY) r = r < 0 ? -1 - r : (r >= a.h ? 2 * a.h - 1 - r : r);
      c[k] = *(const dsx_f2*)&s_ring[wave][r & (kRingRows - 1)][jj0];
    }
    // axis 0: out = sum_k f[k] x[2 i2 + 1 - k] = sum_k f[k] c[5 - k]; both columns of the lane at once
    dsx_f2 lo2 = LO[0] * c[5], hi2 = HI[0] * c[5];
#pragma unroll
    for (int k = 1; k < 6; ++k) {
      lo2 = pk_fma(LO[k], c[5 - k], lo2);
      hi2 = pk_fma(HI[k], c[5 - k], hi2);
    }
    dsx_f4 pk = {lo2.x, hi2.x, lo2.y, hi2.y};  // (lo, hi) per column: the row pass filters both at once
    // axis 1, low-pass only: out[o2 + t] = sum_k LO[k] x[2 t + 5 - k] (local columns), t = lane
#if DSX_DPP_X
    const dsx_f4 p0 = pk;  // columns 2t, 2t+1 are the lane's own; 2t+2.. come from lanes t+1, t+2 (DPP)
    const dsx_f4 p1 = {lane_above(p0.x), lane_above(p0.y), lane_above(p0.z), lane_above(p0.w)};
    const dsx_f4 p2 = {lane_above(p1.x), lane_above(p1.y), lane_above(p1.z), lane_above(p1.w)};
#else
    *(dsx_f4*)&s_x2[wave][jj0] = pk;
    wave_sync();
    const dsx_f4 p0 = *(const dsx_f4*)&s_x2[wave][2 * lane];
    const dsx_f4 p1 = *(const dsx_f4*)&s_x2[wave][2 * lane + 2];
    const dsx_f4 p2 = *(const dsx_f4*)&s_x2[wave][2 * lane + 4];
#endif
    dsx_f2 v = LO[5] * p0.xy;
    v = pk_fma(LO[4], p0.zw, v);
    v = pk_fma(LO[3], p1.xy, v);
    v = pk_fma(LO[2], p1.zw, v);
    v = pk_fma(LO[1], p2.xy, v);
    v = pk_fma(LO[0], p2.zw, v);
#if !DSX_DPP_X
    wave_sync();
#endif
    const float q = v.y * v.y;
    // (running extrema as bare v_min / v_max: fminf / fmaxf quiet a possible signalling NaN of every operand first, one
    //  more 5-cycle instruction each -- a NaN coefficient means a flagged plane, PlaneStats::flags)
    q2min = min_no_nan(q2min, l2_valid ? q : __builtin_huge_valf());
    q2max = max_no_nan(q2max, l2_valid ? q : 0.f);
    if (l2_store && !DSX_ABL(a, 16 | 128)) {  // 128: diagnosis, only the level-2 stores are left out
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.x), rs_aa2, off_2, (unsigned)(i2 * a.lda2) * 4u, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.y), rs_da2, off_2, (unsigned)(i2 * a.ld2) * 4u, 0);
      if (l2_edge) {  // extension margins of aa_2, read by the next level's aligned vector loads
        float* arow = aa2 + (long long)i2 * a.lda2;
        if (jo2 < 4) arow[-1 - jo2] = v.x;
        if (jo2 >= a.w2 - 8) arow[2 * a.w2 - 1 - jo2] = v.x;
      }
    }
  };

  // One output row i: the raw rows 2i, 2i+1 become window slots (r4, r5); r0..r5 = oldest..newest.
  // steady_c (std::true_type, fused kernel): the row is owned by the segment, its raw rows lie inside the plane,
  // and exactly the level-2 row (i - 1) / 2 becomes complete at odd i -- none of that is tested again
  auto step = [&](auto steady_c, int i, const MarchRaw& raw0, const MarchRaw& raw1, float (&r0)[4], float (&r1)[4],
                  float (&r2)[4], float (&r3)[4], float (&r4)[4], float (&r5)[4]) {
    constexpr bool STEADY = decltype(steady_c)::value;
    const bool own_row = STEADY || !FUSE || i >= own_row_lo;  // wave-uniform
    march_consume<IN_KIND, STEADY>(a, raw0, 2 * i, col, own_row, any_rev, st, r4);
    march_consume<IN_KIND, STEADY>(a, raw1, 2 * i + 1, col, own_row, any_rev, st, r5);
    // axis 0: out = sum_k f[k] * x[2i + 1 - k] = sum_k f[k] * r(5 - k)
    float lo[4], hi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lo[e] = LOV[0] * r5[e];
      hi[e] = HIV[0] * r5[e];
      lo[e] = fmaf(LOV[1], r4[e], lo[e]); hi[e] = fmaf(HIV[1], r4[e], hi[e]);
      lo[e] = fmaf(LOV[2], r3[e], lo[e]); hi[e] = fmaf(HIV[2], r3[e], hi[e]);
      lo[e] = fmaf(LOV[3], r2[e], lo[e]); hi[e] = fmaf(HIV[3], r2[e], hi[e]);
      lo[e] = fmaf(LOV[4], r1[e], lo[e]); hi[e] = fmaf(HIV[4], r1[e], hi[e]);
      lo[e] = fmaf(LOV[5], r0[e], lo[e]); hi[e] = fmaf(HIV[5], r0[e], hi[e]);
    }
#if DSX_DPP_X
    // axis 1, low-pass only.  A lane's two outputs need its own four columns and the first four of the lane
    // above: those come over the DPP crossbar (wave_shl:1, lane L reads lane L + 1) -- no LDS round trip.
    // out[jj] = LO0 x[2jj+5] + LO1 x[2jj+4] + ... + LO5 x[2jj], x = (lo[0..3], next lane's lo[0..3])
    // The DPP operand rides on the multiply itself (v_mul/v_fmac_f32_dpp): the exchange costs no instruction.
    // Per output the terms are added in the order LO0 .. LO5, as everywhere else in this file.
    float res[2][2];
    {
      float v0l, v1l, v0h, v1h;
      const float c0 = LO[0], c1 = LO[1], c2 = LO[2], c3 = LO[3];
#define DSX_DPP_UP " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      asm("s_nop 1\n"  // VALU write -> DPP read of lo / hi: 2 wait states (the assembler does not see in here)
          "v_mul_f32_dpp %0, %5, %12" DSX_DPP_UP   // v0 = LO0 n[1]
          "v_mul_f32_dpp %1, %7, %12" DSX_DPP_UP   // v1 = LO0 n[3]
          "v_mul_f32_dpp %2, %9, %12" DSX_DPP_UP
          "v_mul_f32_dpp %3, %11, %12" DSX_DPP_UP
          "v_fmac_f32_dpp %0, %4, %13" DSX_DPP_UP  // v0 += LO1 n[0]
          "v_fmac_f32_dpp %1, %6, %13" DSX_DPP_UP  // v1 += LO1 n[2]
          "v_fmac_f32_dpp %2, %8, %13" DSX_DPP_UP
          "v_fmac_f32_dpp %3, %10, %13" DSX_DPP_UP
          "v_fmac_f32_dpp %1, %5, %14" DSX_DPP_UP  // v1 += LO2 n[1]
          "v_fmac_f32_dpp %3, %9, %14" DSX_DPP_UP
          "v_fmac_f32_dpp %1, %4, %15" DSX_DPP_UP  // v1 += LO3 n[0]
          "v_fmac_f32_dpp %3, %8, %15" DSX_DPP_UP
          : "=&v"(v0l), "=&v"(v1l), "=&v"(v0h), "=&v"(v1h)
          : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]),
            "v"(c0), "v"(c1), "v"(c2), "v"(c3));
#undef DSX_DPP_UP
      v0l = fmaf(LO[2], lo[3], v0l); v0l = fmaf(LO[3], lo[2], v0l); v0l = fmaf(LO[4], lo[1], v0l); v0l = fmaf(LO[5], lo[0], v0l);
      v1l = fmaf(LO[4], lo[3], v1l); v1l = fmaf(LO[5], lo[2], v1l);
      v0h = fmaf(LO[2], hi[3], v0h); v0h = fmaf(LO[3], hi[2], v0h); v0h = fmaf(LO[4], hi[1], v0h); v0h = fmaf(LO[5], hi[0], v0h);
      v1h = fmaf(LO[4], hi[3], v1h); v1h = fmaf(LO[5], hi[2], v1h);
      res[0][0] = v0l; res[0][1] = v1l; res[1][0] = v0h; res[1][1] = v1h;
    }
#else
    sE[0][lane] = make_float2(lo[0], lo[2]);
    sO[0][lane] = make_float2(lo[1], lo[3]);
    sE[1][lane] = make_float2(hi[0], hi[2]);
    sO[1][lane] = make_float2(hi[1], hi[3]);
    wave_sync();
    // axis 1, low-pass only: out[jj] = LO0 O[jj+2] + LO1 E[jj+2] + LO2 O[jj+1] + LO3 E[jj+1] + LO4 O[jj] + LO5 E[jj]
    float res[2][2];
    if (out_lane) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float2 e01 = sE[b][lane], e23 = sE[b][lane + 1];
        const float2 o01 = sO[b][lane], o23 = sO[b][lane + 1];
        float v0 = LO[0] * o23.x;
        v0 = fmaf(LO[1], e23.x, v0); v0 = fmaf(LO[2], o01.y, v0); v0 = fmaf(LO[3], e01.y, v0);
        v0 = fmaf(LO[4], o01.x, v0); v0 = fmaf(LO[5], e01.x, v0);
        float v1 = LO[0] * o23.y;
        v1 = fmaf(LO[1], e23.y, v1); v1 = fmaf(LO[2], o23.x, v1); v1 = fmaf(LO[3], e23.x, v1);
        v1 = fmaf(LO[4], o01.y, v1); v1 = fmaf(LO[5], e01.y, v1);
        res[b][0] = v0;
        res[b][1] = v1;
      }
    }
    wave_sync();
#endif
    if (FUSE) {
      if (out_lane) {
        const int j = jl;
        const float q0 = res[1][0] * res[1][0], q1 = res[1][1] * res[1][1];
        if (own_row && !DSX_ABL(a, 16 | 256)) {  // (256: diagnosis, only the da_1 stores are left out)  da_1: owned rows and columns only (overlap rows / columns belong to a neighbour)
          // (64: diagnosis, with the 512-byte strip pieces above: a row pitch of 33 cache lines -- no partial line at all)
          const unsigned soff = DSX_ABL(a, 64) ? (unsigned)i * 4224u : (unsigned)(i * a.ld) * 4u;
          constexpr int aux = DSX_NT ? kBufNT : 0;
          if (c_s0 && c_s1) {
            const dsx_u32x2 dv = {__float_as_uint(res[1][0]), __float_as_uint(res[1][1])};
            __builtin_amdgcn_raw_buffer_store_b64(dv, rs_da, off_da, soff, aux);
          } else {
            if (c_s0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res[1][0]), rs_da, off_da, soff, aux);
            if (c_s1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res[1][1]), rs_da, off_da + 4u, soff, aux);
          }
        }
        qmin = min_no_nan(qmin, min_no_nan(c_v0 ? q0 : __builtin_huge_valf(), c_v1 ? q1 : __builtin_huge_valf()));
        qmax = max_no_nan(qmax, max_no_nan(c_v0 ? q0 : 0.f, c_v1 ? q1 : 0.f));
        // aa_1 row -> ring; edge strips also write the half-sample symmetric extension:
        // aa[-1-k] = aa[k] (k < 4), aa[w + k] = aa[w - 1 - k] (k < 6)
        float* ring = s_ring[wave][i & (kRingRows - 1)];
        if (!edge_strip) {
          *(float2*)(ring + jj0) = make_float2(res[0][0], res[0][1]);
        } else {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int je = j + e;
            if (je >= 0 && je < a.w) {
              ring[jj0 + e] = res[0][e];
              if (je < 4 && -1 - je - j0 >= 0) ring[-1 - je - j0] = res[0][e];
              const int t = 2 * a.w - 1 - je - j0;
              if (je >= a.w - 6 && t < kRingPitch) ring[t] = res[0][e];
            }
          }
        }
      }
      // level-2 rows whose six source rows are now in the ring (top rows need reflected row 3)
      if (STEADY) {
        if (2 * next2 + 1 <= i) {
          l2_step(std::true_type(), next2);
          ++next2;
        }
      } else {
        while (next2 < i2e) {
          const int need = min(max(2 * next2 + 1, 3 - 2 * next2), a.h - 1);
          if (need > i) break;
          l2_step(std::false_type(), next2);
          ++next2;
        }
      }
    } else
    if (out_lane) {
      const int j = j0 + jj0;
      const long long o = (long long)i * a.ld + j;
      const long long oa = (long long)i * a.lda + j;
      float* arow = aa + (long long)i * a.lda;
      if (j + 1 < a.w) {
        *(float2*)(aa + oa) = make_float2(res[0][0], res[0][1]);
        *(float2*)(da + o) = make_float2(res[1][0], res[1][1]);
        const float q0 = res[1][0] * res[1][0], q1 = res[1][1] * res[1][1];
        qmin = min_no_nan(qmin, min_no_nan(q0, q1));
        qmax = max_no_nan(qmax, max_no_nan(q0, q1));
      } else if (j < a.w) {
        aa[oa] = res[0][0];
        da[o] = res[1][0];
        const float q0 = res[1][0] * res[1][0];
        qmin = min_no_nan(qmin, q0);
        qmax = max_no_nan(qmax, q0);
      }
      // half-sample symmetric extension of aa into the row margins: aa[-1-k] = aa[k] (k < 4),
      // aa[w + k] = aa[w - 1 - k] (k < 8); read by the next level's aligned vector loads
      if (edge_strip) {  // wave-uniform: only the first / last strips own margin columns
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int je = j + e;
          if (je < a.w) {
            if (je < 4) arow[-1 - je] = res[0][e];
            if (je >= a.w - 8) arow[2 * a.w - 1 - je] = res[0][e];
          }
        }
      }
    }
  };

  // software prefetch: the 6 rows of the NEXT group(s) of 3 output rows are in flight while the
  // current group is filtered (the symmetric extension keeps every prefetched address valid).
  // uint16 planes (2 registers per row) keep TWO groups in flight: the kernel is bound by memory
  // latency, not by registers or VALU.
  constexpr int DEPTH = (IN_KIND == 0) ? 2 : 1;
  MarchRaw nxt[DEPTH][6];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    if (PAIRLD) {
      issue_pairs(2 * (i_begin + 3 * d), nxt[d]);
    } else if (FAST) {
      march_issue6<IN_KIND>(ms, a.ldin, a.H, 2 * (i_begin + 3 * d), col, nxt[d]);
    } else {
#pragma unroll
      for (int r = 0; r < 6; ++r)
        nxt[d][r] = march_issue<IN_KIND, FAST>(ms, a.ldin, a.H, 2 * (i_begin + 3 * d) + r, col);
    }
  }
  // One group of three output rows.  The steady groups run in a loop of their own (DSX_FWD_SPLIT): with both kinds of
  // group in one loop body the two paths merge every iteration, and the compiler reconciles their register assignments
  // with ~70 copies per group (the window, the prefetched rows, the level-2 state) -- a sixth of the loop's vector work.
  auto group = [&](auto steady_c, int i) {
    constexpr bool STEADY = decltype(steady_c)::value;
    MarchRaw cur[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) cur[r] = nxt[0][r];
    if (PAIRLD) {
      // pair p: this lane's 16 bytes of row 2p + (lane & 1); the own half of the own row stays, the other row's own half
      // comes from the neighbour
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const unsigned x = __float_as_uint(nxt[0][2 * p].f.x), y = __float_as_uint(nxt[0][2 * p].f.y);
        const unsigned z = __float_as_uint(nxt[0][2 * p].f.z), w = __float_as_uint(nxt[0][2 * p].f.w);
        const unsigned gx = swap_adjacent(bit_select(oddm, x, z)), gy = swap_adjacent(bit_select(oddm, y, w));
        cur[2 * p].f.x = __uint_as_float(bit_select(oddm, gx, x));      // row 2p, own 4 columns
        cur[2 * p].f.y = __uint_as_float(bit_select(oddm, gy, y));
        cur[2 * p + 1].f.x = __uint_as_float(bit_select(oddm, z, gx));  // row 2p + 1
        cur[2 * p + 1].f.y = __uint_as_float(bit_select(oddm, w, gy));
      }
    }
#pragma unroll
    for (int d = 0; d + 1 < DEPTH; ++d)
#pragma unroll
      for (int r = 0; r < 6; ++r) nxt[d][r] = nxt[d + 1][r];
    if (i + 3 * DEPTH < i_end) {
      if (PAIRLD) {
        issue_pairs(2 * (i + 3 * DEPTH), nxt[DEPTH - 1]);
      } else if (FAST) {
        march_issue6<IN_KIND>(ms, a.ldin, a.H, 2 * (i + 3 * DEPTH), col, nxt[DEPTH - 1]);
      } else {
#pragma unroll
        for (int r = 0; r < 6; ++r)
          nxt[DEPTH - 1][r] = march_issue<IN_KIND, FAST>(ms, a.ldin, a.H, 2 * (i + 3 * DEPTH) + r, col);
      }
    }
    if (STEADY) {
      step(std::true_type(), i, cur[0], cur[1], win[0], win[1], win[2], win[3], win[4], win[5]);
      step(std::true_type(), i + 1, cur[2], cur[3], win[2], win[3], win[4], win[5], win[0], win[1]);
      step(std::true_type(), i + 2, cur[4], cur[5], win[4], win[5], win[0], win[1], win[2], win[3]);
    } else {
      step(std::false_type(), i, cur[0], cur[1], win[0], win[1], win[2], win[3], win[4], win[5]);
      if (i + 1 < i_end) step(std::false_type(), i + 1, cur[2], cur[3], win[2], win[3], win[4], win[5], win[0], win[1]);
      if (i + 2 < i_end) step(std::false_type(), i + 2, cur[4], cur[5], win[4], win[5], win[0], win[1], win[2], win[3]);
    }
  };
  const bool steady_on = FUSE && DSX_FWD_STEADY && (EDGE ? (DSX_FWD_STEADY & 2) != 0 : true);
#if DSX_FWD_SPLIT
  {
    // groups start at i_begin + 3 g; the steady ones are those with steady_lo <= i and i + 3 <= steady_hi: a run
    int i = i_begin;
    int run_lo = i_end, run_hi = i_end;  // first steady group, first group after the run
    if (steady_on) {
      run_lo = i_begin + 3 * ((max(steady_lo - i_begin, 0) + 2) / 3);
      run_hi = (steady_hi - run_lo >= 3) ? run_lo + 3 * ((steady_hi - run_lo) / 3) : run_lo;
      if (run_hi == run_lo) run_lo = run_hi = i_end;
    }
#pragma nounroll
    for (int part = 0; part < 2; ++part) {  // one copy of the general group serves the rows above and below the run
      const int stop = part == 0 ? min(run_lo, i_end) : i_end;
      for (; i < stop; i += 3) group(std::false_type(), i);
      if (part == 0)
        for (; i < run_hi; i += 3) group(std::true_type(), i);
    }
  }
#else
  for (int i = i_begin; i < i_end; i += 3) {
    if (steady_on && i >= steady_lo && i + 3 <= steady_hi) group(std::true_type(), i);  // wave-uniform
    else group(std::false_type(), i);
  }
#endif

  qmin = wave_min_f32(qmin);
  qmax = wave_max_f32(qmax);
  if (lane == 0 && qmin <= qmax) {
    unsigned* mm = a.minmax + ((long long)(a.shared ? 0 : plane) * a.L + a.lvl) * 2;
    atomicMax(&mm[0], ~as_u32(qmin));
    atomicMax(&mm[1], as_u32(qmax));
  }
  if (FUSE) {
    q2min = wave_min_f32(q2min);
    q2max = wave_max_f32(q2max);
    if (lane == 0 && q2min <= q2max) {
      unsigned* mm = a.minmax + ((long long)(a.shared ? 0 : plane) * a.L + a.lvl + 1) * 2;
      atomicMax(&mm[0], ~as_u32(q2min));
      atomicMax(&mm[1], as_u32(q2max));
    }
  }
  if (IN_KIND != 2) {
    double s_all, s_fg;
    if (IN_KIND == 0) {
      s_all = (double)st.isum_all;
      s_fg = (double)st.isum_fg;
    } else {
      s_all = st.fsum_all;
      s_fg = st.fsum_fg;
    }
    s_all = wave_sum_f64(s_all);
    s_fg = wave_sum_f64(s_fg);
    const unsigned cnt = __reduce_add_sync(~0ull, st.cnt);
    const bool bad_any = (IN_KIND == 1) && __any(st.bad != 0u) != 0;
    if (lane == 0) {
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

// IN_KIND: 0 = uint16 pixels (log + statistic fused), 1 = float32 pixels (same), 2 = float32 aa_{l-1}
// WPB: waves per block (consecutive strips of one row segment: a block reads WPB x 512 contiguous bytes per row)
template <int IN_KIND, bool FUSE = false, int WPB = 4>
__global__ __launch_bounds__(64 * WPB, (FUSE && IN_KIND == 0) ? DSX_FWD_MINW : 1) void k_fwd_march(Fwd1Args a) {
  __shared__ __attribute__((aligned(16))) float s_row[WPB][2][2][kMarchCols / 2];  // [wave][lo|hi][parity][col/2]
  __shared__ __attribute__((aligned(16))) float s_ring[FUSE ? WPB : 1][kRingRows][FUSE ? kRingPitch : 4];
  __shared__ __attribute__((aligned(16))) float2 s_x2[FUSE ? WPB : 1][FUSE ? kX2Pitch : 2];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps row math on the SALU
  const int item = blockIdx.x * WPB + wave;
  if (item >= a.nstrips * a.nseg) return;
  const int strip = item % a.nstrips, seg = item / a.nstrips;
  const int plane = blockIdx.y;
  if (seg * a.rows_per_seg >= (FUSE ? a.h2 : a.h)) return;
  if (FUSE) {
    // host guarantees W % 4 == 0 and ldin % 4 == 0: every lane is vector-loadable (mirrored groups at the
    // plane edges are aligned as well) or dead
    const FuseGeom fg = fuse_geom(a, strip);
    const int gc0 = 2 * fg.j0 - 4 + 4 * lane;
    const bool owns = gc0 >= 4 * kFuseOut * strip && (strip == a.nstrips - 1 || gc0 < 4 * kFuseOut * (strip + 1));
    MarchCol col = march_col(gc0, a.W, a.ldin, a.w, owns, false);
    if (!col.vec) { col.dead = true; col.base = 0; col.rev = false; }  // left of the mirrored region: feeds nothing valid
    const bool any_rev = __any(col.rev) != 0;
    // interior strip (wave-uniform): see fwd_march_body, EDGE = false
    const bool interior = strip > 0 && strip < a.nstrips - 1 && fg.j0 > 0 && fg.j0 + kMarchOut + 2 < a.w - 8 &&
                          fg.o2 + kFuseOut < a.w2 - 8 && __all(col.vec && !col.rev && !col.dead) != 0;
    if (interior)
      fwd_march_body<IN_KIND, true, true, false>(a, s_row, (float (*)[kRingRows][kRingPitch])s_ring,
                                                 (float2 (*)[kX2Pitch])s_x2, lane, wave, strip, seg, plane, col, false);
    else
      fwd_march_body<IN_KIND, true, true, true>(a, s_row, (float (*)[kRingRows][kRingPitch])s_ring,
                                                (float2 (*)[kX2Pitch])s_x2, lane, wave, strip, seg, plane, col, any_rev);
    return;
  }
  // lane 0 re-reads the last 4 columns of the previous strip and does not account them
  const MarchCol col = march_col(2 * kMarchOut * strip - 4 + 4 * lane, a.W, a.ldin, a.w, lane >= 1, IN_KIND == 2);
  const bool all_vec = __all(col.vec || (col.dead && a.W >= 4 && (a.ldin & 3) == 0)) != 0;
  const bool any_rev = __any(col.rev) != 0;
  if (all_vec) fwd_march_body<IN_KIND, true, false>(a, s_row, nullptr, nullptr, lane, wave, strip, seg, plane, col, any_rev);
  else fwd_march_body<IN_KIND, false, false>(a, s_row, nullptr, nullptr, lane, wave, strip, seg, plane, col, any_rev);
}

// ================================================================================================
// K2: histogram of q = cH^2, numpy.histogram(bins=256) rule in float32
// ================================================================================================
// One launch covers SEVERAL levels (round 3: the eight histograms of a chain were eight launches, six of them a few
// microseconds of work behind a launch gap each): block x of the grid belongs to the first entry i with x < blk_end[i].
struct HistArgs {
  const float* ws;
  long long ws_plane_stride;
  const unsigned* minmax;
  unsigned* hist;  // [B][L][256]
  int L;
  int shared;  // stack mode: plane 0's min / max and histogram slots for every plane (Fwd1Args::shared)
  int nlev;    // entries below
  int lvl[kMaxLevels];             // level index of the entry
  long long da_off[kMaxLevels];
  int h[kMaxLevels], w[kMaxLevels], ld[kMaxLevels];
  int rows_per_block[kMaxLevels];
  int blk_end[kMaxLevels];         // cumulative block counts
};

// Bin of q = largest i with edges[i] <= q (numpy's estimate-then-correct rule ends there too).
// The edges are numpy.linspace's of the reference's pinned NumPy 1.26.4: float32 end points are
// promoted to float64, edges[i] = i * ((max - min) / 256) + min in float64 (separate multiply and
// add), then rounded to float32.  The estimate (q - qmin) * scale (the subtraction first: exact for q near qmin,
// so the estimate stays within 1e-4 bins however narrow the range) is corrected against the actual edges.
struct HistBins {
  float qmin, qmax, scale;
  double first64, step64;
  __device__ __forceinline__ HistBins(float lo, float hi) : qmin(lo), qmax(hi) {
    scale = 256.0f / (qmax - qmin);
    first64 = (double)qmin;
    step64 = ((double)qmax - (double)qmin) / 256.0;
  }
  __device__ __forceinline__ float edge(int i) const { return (float)__dadd_rn(__dmul_rn((double)i, step64), first64); }
};

// Round 3: branch-free binning.  Round 2's kernel spent 24 vector instructions and ~10 exec-mask branches per value
// (an "is the estimate near an edge" test in front of a float64 edge evaluation, byte-packed counters for the four
// lowest bins, LDS atomics under a branch for the rest).  Now the 257 float32 edges of the plane are built ONCE per
// block (one per thread, the same float64 arithmetic as before) into LDS, and every value takes the same path:
//     i0 = min(int((q - qmin) * scale), 255);  (e_lo, e_hi) = edges[i0], edges[i0 + 1];  (one ds_read2_b32; most
//     idx = i0 - (q < e_lo) + (q >= e_hi)       lanes read the same pair: broadcast, no bank conflict)
// which is numpy's estimate-then-correct rule with the correction always applied (the estimate is within 1e-4 bins,
// the correction moves it by at most one), and one conflict-free LDS atomic: every lane owns a 16-bit counter per bin
// (256 bins x 32 dwords, lane pair per dword, the kHistWaves waves of the block share them -- at most kHistWaves x the
// values of a thread, far below 65536).  10 vector + 2 LDS instructions per value; no data-dependent branch.
constexpr int kHistLoads = 8;  // 16-byte loads in flight per lane

constexpr int kHistWaves = 8;  // waves per block: they share one 32 KB counter array (LDS is what the 4-stream mix runs short of)

__global__ __launch_bounds__(64 * kHistWaves) void k_hist(HistArgs args) {
  __shared__ __attribute__((aligned(16))) unsigned s_cnt[256 * 32];
  __shared__ float s_edge[264];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = blockIdx.y;
  // the entry of this block (block-uniform, scalar unit)
  int ent = 0;
  while (ent + 1 < args.nlev && (int)blockIdx.x >= args.blk_end[ent]) ++ent;
  struct {
    const float* ws; long long ws_plane_stride, da_off; int h, w, ld, lvl, L, rows_per_block, shared;
    const unsigned* minmax; unsigned* hist;
  } a = {args.ws, args.ws_plane_stride, args.da_off[ent], args.h[ent], args.w[ent], args.ld[ent], args.lvl[ent], args.L,
         args.rows_per_block[ent], args.shared, args.minmax, args.hist};
  const int bx = (int)blockIdx.x - (ent > 0 ? args.blk_end[ent - 1] : 0);
  const int splane = a.shared ? 0 : plane;
  const unsigned* mm = a.minmax + ((long long)splane * a.L + a.lvl) * 2;
  const float qmin = as_f32(~mm[0]), qmax = as_f32(mm[1]);
  if (!(qmin < qmax)) return;  // constant cH^2: Otsu early-out, no histogram (block-uniform)
  const HistBins hb(qmin, qmax);
  {
    uint4* z = (uint4*)s_cnt;
#pragma unroll
    for (int i = 0; i < 2048 / (64 * kHistWaves); ++i) z[tid + 64 * kHistWaves * i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 256) s_edge[tid] = hb.edge(tid);
    else if (tid < 264) s_edge[tid] = __builtin_huge_valf();  // edges[256]: the last bin is closed (idx stays 255)
  }
  __syncthreads();
  const unsigned lane_off = (unsigned)(lane >> 1) * 4u;
  const unsigned inc = (lane & 1) ? 0x10000u : 1u;
  const float scale = hb.scale;
  auto tally = [&](float x) {
    const float q = x * x;
    const int i0 = min((int)((q - qmin) * scale), 255);
    const float e_lo = s_edge[i0], e_hi = s_edge[i0 + 1];
    const int idx = i0 - (q < e_lo ? 1 : 0) + (q >= e_hi ? 1 : 0);
    atomicAdd((unsigned*)((char*)s_cnt + ((unsigned)idx * 128u + lane_off)), inc);
  };
  const float* da = a.ws + plane * a.ws_plane_stride + a.da_off;
  const int r0 = bx * a.rows_per_block;
  const int nrows = min(a.h, r0 + a.rows_per_block) - r0;
  // ---- full 256-column chunks: item = (row, chunk), dealt round robin to the four waves, kHistLoads at a time ----
  const int gf = a.w >> 8;
  const int nitems = nrows * gf;
  for (int it0 = wave; it0 < nitems; it0 += kHistWaves * kHistLoads) {
    dsx_f4 v[kHistLoads];
#pragma unroll
    for (int k = 0; k < kHistLoads; ++k) {
      const int it = min(it0 + kHistWaves * k, nitems - 1);  // clamped: what a clamped load returns is not counted
      const int row = it / gf, g = it - row * gf;   // wave-uniform (scalar unit)
      const float* p = da + (long long)(r0 + row) * a.ld + 256 * g + 4 * lane;
#if DSX_NT
      v[k] = __builtin_nontemporal_load((const dsx_f4*)p);
#else
      v[k] = *(const dsx_f4*)p;
#endif
    }
#pragma unroll
    for (int k = 0; k < kHistLoads; ++k) {
      if (it0 + kHistWaves * k < nitems) {  // wave-uniform
        tally(v[k].x); tally(v[k].y); tally(v[k].z); tally(v[k].w);
      }
    }
  }
  // ---- the last (w mod 256) columns of every row: one value per thread and step ----
  const int tw = a.w - (gf << 8);
  for (int e = tid; e < nrows * tw; e += 64 * kHistWaves) {
    const int row = e / tw, c = (gf << 8) + (e - row * tw);
    tally(da[(long long)(r0 + row) * a.ld + c]);
  }
  __syncthreads();
  // bin tid: 32 dwords, read in a rotated order (thread t starts at column t: no two threads of a group on one bank)
  if (tid < 256) {
    unsigned n = 0;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
      const unsigned v = s_cnt[tid * 32 + ((j + tid) & 31)];
      n += (v & 0xFFFFu) + (v >> 16);
    }
    if (n) atomicAdd(&a.hist[((long long)splane * a.L + a.lvl) * 256 + tid], n);
  }
}

// ================================================================================================
// K3: config decision + Otsu threshold (one wave per plane and level)
// ================================================================================================
struct OtsuArgs {
  const PlaneStats* stats;
  double npix;
  double high_int;
  const unsigned* minmax;
  const unsigned* hist;
  float* thr;    // [B][L]
  float* otsu;   // [B][L]
  int* cfg;      // [B]
  double* means; // [B][2]
  float max_thr[2];
  int L;
  unsigned* sticky;  // host-mapped flag word of the context (dsx_ctx::h_sticky), may be null
  int shared;        // stack mode: every plane takes its thresholds from plane 0's min / max and histogram slots
};

// Otsu value of one plane and level from its 256-bin histogram, by ONE wave (64 lanes): returns the value on every
// lane.  scratch: 6 x 256 doubles of LDS.  The caller applies sqrt / the threshold cap.
__device__ __forceinline__ double otsu_from_hist(float q_lo, float q_hi, const unsigned* h, double* scratch, int lane) {
  // The class statistics are accumulated sequentially in numpy's order (cumsum forward for class 1,
  // cumsum over the reversed arrays for class 2): empty bins then give bit-identical variances on both
  // sides, and "first maximum" picks the same bin as np.argmax.  Only the two running sums are
  // sequential (lane 0 forward, lane 1 backward, additions only); products, quotients, variances and
  // the arg-max run on all lanes.
  double* s_cnt = scratch;
  double* s_cb = scratch + 256;
  double (*s_w)[256] = (double (*)[256])(scratch + 512);
  double (*s_s)[256] = (double (*)[256])(scratch + 1024);
  if (!(q_lo < q_hi)) return (double)q_lo;  // all values equal: threshold_otsu returns that value
  // float32 bin edges exactly as k_hist builds them (numpy.linspace of NumPy 1.26.4), float32 bin
  // centres (edges[:-1] + edges[1:]) / 2 as in skimage's threshold_otsu on a float32 image; the class
  // statistics below are float64 (counts.astype(float))
  const double first = (double)q_lo, last = (double)q_hi;
  const double step = (last - first) / 256.0;
  auto edge = [&](int i) { return (i == 256) ? q_hi : (float)__dadd_rn(__dmul_rn((double)i, step), first); };
  auto centre = [&](int g) { return __fmul_rn(__fadd_rn(edge(g), edge(g + 1)), 0.5f); };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int g = lane + 64 * i;
    const double c = (double)h[g];
    s_cnt[g] = c;
    s_cb[g] = c * (double)centre(g);
  }
  wave_sync();
  if (lane < 2) {
    // 256 dependent additions per running sum; what made this loop slow was not them but an LDS round trip per step
    // (the stores of one step may alias the loads of the next as far as the compiler knows): 16 values are loaded back
    // to back, summed in registers, stored back to back -- 18 -> 2 us per plane and level
    double w = 0.0, sacc = 0.0;
    for (int k0 = 0; k0 < 256; k0 += 16) {
      double c[16], b[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int g = lane ? 255 - (k0 + j) : k0 + j;
        c[j] = s_cnt[g];
        b[j] = s_cb[g];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        w += c[j];
        sacc += b[j];
        c[j] = w;
        b[j] = sacc;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int g = lane ? 255 - (k0 + j) : k0 + j;
        s_w[lane][g] = c[j];
        s_s[lane][g] = b[j];
      }
    }
  }
  wave_sync();
  // variance12[g] = weight1[g] weight2[g + 1] (mean1[g] - mean2[g + 1])^2, g = 0 .. 254; first maximum
  double best = -1.0;
  int best_g = 0;
#pragma unroll
  for (int i = 3; i >= 0; --i) {
    const int g = lane + 64 * i;
    if (g < 255) {
      const double d = s_s[0][g] / s_w[0][g] - s_s[1][g + 1] / s_w[1][g + 1];
      const double var = (s_w[0][g] * s_w[1][g + 1]) * (d * d);
      if (var >= best) { best = var; best_g = g; }  // descending g: ties keep the lower bin
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ob = __shfl_xor(best, o);
    const int og = __shfl_xor(best_g, o);
    if (ob > best || (ob == best && og < best_g)) { best = ob; best_g = og; }
  }
  return (double)centre(best_g);
}

__global__ __launch_bounds__(64) void k_otsu(OtsuArgs a) {
  const int lvl = blockIdx.x, plane = blockIdx.y, lane = threadIdx.x;
  // decision of filtering.py:459-462 (empty class -> mean 0.0)
  const PlaneStats st = a.stats[plane];
  const double cnt_fg = (double)st.cnt_fg, cnt_bg = a.npix - cnt_fg;
  const double fore = cnt_fg > 0 ? st.sum_fg / cnt_fg : 0.0;
  const double back = cnt_bg > 0 ? (st.sum_all - st.sum_fg) / cnt_bg : 0.0;
  const int cfg = (fore > back && fore > a.high_int) ? 1 : 0;
  if (lvl == 0 && lane == 0) {
    // (same value from every writer: a plain store, no read-modify-write across the host link)
    if (st.flags != 0ull && a.sticky != nullptr) *(volatile unsigned*)a.sticky = 1u;
    a.cfg[plane] = cfg;
    a.means[2 * plane] = fore;
    a.means[2 * plane + 1] = back;
  }
  const long long pl = (long long)plane * a.L + lvl;
  const long long spl = (long long)(a.shared ? 0 : plane) * a.L + lvl;
  const unsigned* mm = a.minmax + spl * 2;
  const float q_lo = as_f32(~mm[0]), q_hi = as_f32(mm[1]);
  __shared__ double s_scratch[6 * 256];
  const double otsu = otsu_from_hist(q_lo, q_hi, a.hist + spl * 256, s_scratch, lane);
  if (lane == 0) {
    const double t = fmin(otsu >= 0 ? sqrt(otsu) : 0.0, (double)a.max_thr[cfg]);
    a.otsu[pl] = (float)otsu;
    a.thr[pl] = (float)t;
  }
}

// ================================================================================================
// K4: row filter
// ================================================================================================
struct RowArgs {
  float* ws;
  long long ws_plane_stride;
  long long da_off;
  int h, w, ld;
  const float* thr;  // [B][L]
  const int* cfg;    // [B]
  int lvl, L;
  int lvl_active[2];  // number of levels each config filters (levels >= that get Delta = 0)
  int M, K;
  int npass;
  int radix[kMaxPasses];
  const float2* tw;    // [M] exp(-2 pi i t / M)
  const float2* g[2];  // per config: G1[M] then G2[M]
  int kcut[2];         // per config: the gains vanish for kcut < k < M - kcut
  float inv_M;
  int ablate;          // diagnosis only (DSX_ABLATE): 1 = no median, 2 = no FFT passes, 4 = no spectral step
};

__device__ __forceinline__ unsigned f32_key(float v) {
  const unsigned b = as_u32(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) {
  return as_f32((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

template <int R, int CPL, int JK = (R - 1) / 2, int KO = (R - 1) / 2, class IX = dsx_idx_plain>
__device__ __forceinline__ void fft_pass(float2* buf, const float2* tw, int M, int s, float inv_s,
                                         int lane) {
  constexpr int MAXB = (CPL + R - 1) / R;
  const int nb = M / R;
  const bool unit_tw = (s * R == M);  // last pass: all twiddles are 1 (wave-uniform)
  dsx_c32 v[MAXB][R];
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    const int b = lane + kWave * i;
    if (b < nb) dsx_bfly_load<R, JK, IX>((const dsx_c32*)buf, b, nb, v[i]);
  }
  // LDS operations of one wave execute in program order: every read above precedes the writes below
  wave_sync();
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    const int b = lane + kWave * i;
    if (b < nb) dsx_bfly_store<R, JK, KO, IX>((dsx_c32*)buf, (const dsx_c32*)tw, b, s, inv_s, v[i], unit_tw);
    // keep the unrolled butterflies from being interleaved: their temporaries would all be live
    // at once (215+ VGPRs at 18 values per lane) for no gain -- other waves hide the latency
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_sync();
}

template <int CPL>
__device__ __forceinline__ void fft_pass_generic(float2* buf, const float2* tw, int M, int s,
                                                 float inv_s, int R, int lane) {
  float2 acc[CPL];
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int o = lane + kWave * i;
    acc[i] = make_float2(0.f, 0.f);
    if (o < M) {
      const dsx_c32 r = dsx_generic_output((const dsx_c32*)buf, (const dsx_c32*)tw, o, M, s, inv_s, R);
      acc[i] = make_float2(r.x, r.y);
    }
  }
  wave_sync();
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int o = lane + kWave * i;
    if (o < M) buf[o] = acc[i];
  }
  wave_sync();
}

template <int CPL>
__device__ __forceinline__ void fft_run(float2* buf, const float2* tw, const RowArgs& a, int lane_in) {
  int s = 1;
  for (int pi = 0; pi < a.npass; ++pi) {
    const int R = a.radix[pi];
    const float inv_s = 1.0f / (float)s;
    // Opaque copies per iteration: otherwise LICM hoists every case's per-lane address arithmetic
    // out of this loop and keeps it live across all passes (+25 VGPRs per radix case).
    int lane = lane_in, M = a.M;
    asm volatile("" : "+v"(lane), "+s"(M));
    switch (R) {
      case 2: fft_pass<2, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 3: fft_pass<3, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 4: fft_pass<4, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 5: fft_pass<5, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 6: fft_pass<6, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 8: fft_pass<8, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 9: fft_pass<9, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 10: fft_pass<10, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 12: fft_pass<12, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 15: fft_pass<15, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 16: fft_pass<16, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 20: fft_pass<20, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 25: fft_pass<25, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 7: fft_pass<7, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 11: fft_pass<11, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 13: fft_pass<13, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 17: fft_pass<17, CPL>(buf, tw, M, s, inv_s, lane); break;
      case 19: fft_pass<19, CPL>(buf, tw, M, s, inv_s, lane); break;
      default: fft_pass_generic<CPL>(buf, tw, M, s, inv_s, R, lane); break;
    }
    s *= R;
  }
}

// Compile-time pass lists for the hot lengths: M, the strides and the butterfly counts become
// constants, so LDS offsets turn into immediates and the b < nb guards fold away.
template <int CPL, int M_, int S_, int R, int... REST>
__device__ __forceinline__ void fft_static_passes(float2* buf, const float2* tw, int lane_in) {
  int lane = lane_in;
  asm volatile("" : "+v"(lane));  // per-pass copy: keeps one pass's address arithmetic out of the next
  fft_pass<R, CPL>(buf, tw, M_, S_, 1.0f / (float)S_, lane);
  if constexpr (sizeof...(REST) > 0) fft_static_passes<CPL, M_, S_ * R, REST...>(buf, tw, lane_in);
}
// the same with the bank swizzle of dsx_idx_swz on the row buffer AND the twiddle table (power-of-two plans), the
// addressing written out as "swizzled base + immediate" (dsx_swz_load / dsx_swz_store)
template <int R, int CPL, int M_, int S_>
__device__ __forceinline__ void fft_pass_swz(float2* buf, const float2* tw, int lane) {
  constexpr int MAXB = (CPL + R - 1) / R;
  constexpr int NB = M_ / R;
  dsx_c32 v[MAXB][R];
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    const int b = lane + kWave * i;
    if (kWave * i + kWave <= NB || b < NB) dsx_swz_load<R, M_>((const dsx_c32*)buf, b, v[i]);
  }
  wave_sync();
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    const int b = lane + kWave * i;
    if (kWave * i + kWave <= NB || b < NB) dsx_swz_store<R, M_, S_>((dsx_c32*)buf, (const dsx_c32*)tw, b, v[i]);
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_sync();
}
template <int CPL, int M_, int S_, int R, int... REST>
__device__ __forceinline__ void fft_static_passes_swz(float2* buf, const float2* tw, int lane_in) {
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
#if DSX_SWZ == 2
  fft_pass<R, CPL, (R - 1) / 2, (R - 1) / 2, dsx_idx_swz>(buf, tw, M_, S_, 1.0f / (float)S_, lane);  // the generic policy (A/B)
#else
  fft_pass_swz<R, CPL, M_, S_>(buf, tw, lane);
#endif
  if constexpr (sizeof...(REST) > 0) fft_static_passes_swz<CPL, M_, S_ * R, REST...>(buf, tw, lane_in);
}
// PLAN_ ids of k_rowfilter: 0 = passes from RowArgs; the others must match dsx.hip's dispatch.
template <int PLAN_>
struct StaticFft {
  static constexpr int M = 0;
  static constexpr bool kSwz = false;  // dsx_idx_swz addressing of the row buffer and the twiddle table (rf_pair_body)
};
template <>
struct StaticFft<1> {  // level 1 of a 2048-wide plane
  static constexpr int M = 1026;
  static constexpr bool kSwz = false;
  static constexpr int kRadix[3] = {19, 9, 6};
  template <int CPL>
  static __device__ __forceinline__ void run(float2* buf, const float2* tw, int lane) {
    fft_static_passes<CPL, 1026, 1, 19, 9, 6>(buf, tw, lane);
  }
#if DSX_FWD_PRUNE
  // Forward transform whose consumer (the spectral step) only reads the bins k <= kcut and k >= M - kcut (the low-pass
  // is an exact zero beyond, dsx_plan.h: kcut = 103 / 206 of 513 for the production configs): the passes run in the
  // order 6, 9, 19, so that the LAST pass is the expensive radix-19 one -- outputs q + 54 k, q < 54 -- and only its
  // pairs (k, 19 - k) with k <= (kcut + 53) / 54 are computed (2 / 4 of 9: 72 / 144 of its 324 multiply-adds); the
  // bins it leaves stale are the ones the spectral step overwrites with zeros.  The mirror image of run_inverse.
  template <int CPL, int KO>
  static __device__ __forceinline__ void run_last_pruned(float2* buf, const float2* tw, int lane_in) {
    fft_static_passes<CPL, 1026, 1, 6, 9>(buf, tw, lane_in);
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    fft_pass<19, CPL, 9, KO>(buf, tw, 1026, 54, 1.0f / 54.0f, lane);
  }
  template <int CPL>
  static __device__ __forceinline__ void run_forward(float2* buf, const float2* tw, int lane, int kcut) {
    const int ko = (kcut + 53) / 54;
    if (ko <= 2) run_last_pruned<CPL, 2>(buf, tw, lane);
    else if (ko <= 4) run_last_pruned<CPL, 4>(buf, tw, lane);
    else run<CPL>(buf, tw, lane);
  }
#else
  template <int CPL>
  static __device__ __forceinline__ void run_forward(float2* buf, const float2* tw, int lane, int) { run<CPL>(buf, tw, lane); }
#endif
  // Inverse transform of a spectrum that vanishes for kcut < k < M - kcut: the first pass (radix 19,
  // sources b + 54 j) only sees non-zero input pairs (j, 19 - j) for j <= (kcut + 53) / 54.
  template <int CPL, int JK>
  static __device__ __forceinline__ void run_first_pruned(float2* buf, const float2* tw, int lane_in) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    fft_pass<19, CPL, JK>(buf, tw, 1026, 1, 1.0f, lane);
    fft_static_passes<CPL, 1026, 19, 9, 6>(buf, tw, lane_in);
  }
  template <int CPL>
  static __device__ __forceinline__ void run_inverse(float2* buf, const float2* tw, int lane, int kcut) {
    const int jk = (kcut + 53) / 54;
    if (jk <= 2) run_first_pruned<CPL, 2>(buf, tw, lane);
    else if (jk <= 4) run_first_pruned<CPL, 4>(buf, tw, lane);
    else run<CPL>(buf, tw, lane);
  }
};
template <>
struct StaticFft<2> {  // level 2 of a 2048-wide plane: 515 values embedded in 1071
  static constexpr int M = 1071;
  static constexpr bool kSwz = false;
  static constexpr int kRadix[3] = {17, 9, 7};
  template <int CPL>
  static __device__ __forceinline__ void run(float2* buf, const float2* tw, int lane) {
    fft_static_passes<CPL, 1071, 1, 17, 9, 7>(buf, tw, lane);
  }
  template <int CPL>
  static __device__ __forceinline__ void run_forward(float2* buf, const float2* tw, int lane, int) { run<CPL>(buf, tw, lane); }
  template <int CPL>
  static __device__ __forceinline__ void run_inverse(float2* buf, const float2* tw, int lane, int) {
    run<CPL>(buf, tw, lane);  // the embedded operator is not band-limited
  }
};

// Further compile-time plans: the wide levels of the production tile (1600 x 2000) and of 1800 x 1800 planes.
#define DSX_STATIC_FFT(ID, LEN, R0, R1, R2, SWZ)                                                   \
  template <>                                                                                      \
  struct StaticFft<ID> {                                                                           \
    static constexpr int M = LEN;                                                                  \
    static constexpr bool kSwz = SWZ;                                                              \
    template <int CPL>                                                                             \
    static __device__ __forceinline__ void run(float2* buf, const float2* tw, int lane) {          \
      if constexpr (SWZ) fft_static_passes_swz<CPL, LEN, 1, R0, R1, R2>(buf, tw, lane);            \
      else fft_static_passes<CPL, LEN, 1, R0, R1, R2>(buf, tw, lane);                              \
    }                                                                                              \
    template <int CPL>                                                                             \
    static __device__ __forceinline__ void run_forward(float2* buf, const float2* tw, int lane, int) { \
      run<CPL>(buf, tw, lane);                                                                     \
    }                                                                                              \
    template <int CPL>                                                                             \
    static __device__ __forceinline__ void run_inverse(float2* buf, const float2* tw, int lane, int) { \
      run<CPL>(buf, tw, lane);                                                                     \
    }                                                                                              \
  };
// (power-of-two lengths: their passes scatter at power-of-two strides -- bank-swizzled addressing, dsx_fft_core.h: dsx_idx_swz)
DSX_STATIC_FFT(3, 2048, 16, 16, 8, DSX_SWZ != 0)  // 1002 values (level 1 of a 2000-wide plane) embedded in 2048
DSX_STATIC_FFT(4, 1815, 15, 11, 11, false)         // 902 values (level 1 of an 1800-wide plane) embedded in 1815
DSX_STATIC_FFT(5, 1024, 16, 8, 8, DSX_SWZ != 0)   // 503 values (level 2 of a 2000-wide plane) embedded in 1024
DSX_STATIC_FFT(6, 960, 15, 8, 8, false)            // 453 values (level 2 of an 1800-wide plane) embedded in 960
#undef DSX_STATIC_FFT

typedef short dsx_s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ dsx_s16x2 as_s16x2(unsigned u) {
  union { unsigned u; dsx_s16x2 v; } c;
  c.u = u;
  return c.v;
}
__device__ __forceinline__ unsigned as_u32(dsx_s16x2 v) {
  union { unsigned u; dsx_s16x2 v; } c;
  c.v = v;
  return c.u;
}

// ---- element slots of a lane ------------------------------------------------------------------------
// A row of N values is split into Gf = N / 256 full groups (slot 4 g + i of a lane = element
// 256 g + 4 lane + i: one 16-byte access per group and row) and T = ceil((N % 256) / 64) lane-strided
// tail slots (slot 4 GV + k = element 256 Gf + 64 k + lane).  Only Gf groups and T tail slots are
// executed (wave-uniform guards), so a row of 1026 values costs 17 slots per lane and one of 515
// values 9 -- not the 20 the register arrays are dimensioned for.
template <int CPL>
struct RowSlots {
  static constexpr int GV = CPL / 4;     // vector groups the class can hold (N <= 64 CPL)
  static constexpr int E = 4 * GV + 4;   // register slots: vector groups + 4 tail slots
};
template <int E>
struct SlotMask { typedef unsigned long long type; };
template <> struct SlotMask<4> { typedef unsigned type; };
template <> struct SlotMask<8> { typedef unsigned type; };
template <> struct SlotMask<12> { typedef unsigned type; };
template <> struct SlotMask<20> { typedef unsigned type; };

// f(e) for every active slot e (e is a compile-time constant after unrolling)
template <int GV, typename F>
__device__ __forceinline__ void for_slots(int gf, int nt, F&& f) {
#pragma unroll
  for (int g = 0; g < GV; ++g) {
    if (g < gf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) f(4 * g + i);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k < nt) f(4 * GV + k);
  }
}

// Per-lane count of packed 16-bit values below a packed threshold, both halves at once:
// x and t are in signed order (value ^ 0x8000); returns {#(x.lo < t.lo), #(x.hi < t.hi)} packed.
template <int GV, int E>
__device__ __forceinline__ unsigned count_below_pk16(const unsigned (&x)[E], unsigned t, int gf, int nt) {
  const dsx_s16x2 tv = as_s16x2(t);
  dsx_s16x2 acc = as_s16x2(0u);
  for_slots<GV>(gf, nt, [&](int e) {
    const dsx_s16x2 d = __builtin_elementwise_sub_sat(as_s16x2(x[e]), tv);  // < 0  <=>  x < t
    acc -= (d >> 15);                                                       // -1 per hit
  });
  return as_u32(acc);
}

// largest 16-bit T (per half) with #{x < T} <= rank (per half); x in signed order
template <int GV, int E>
__device__ __forceinline__ unsigned bisect_pk16(const unsigned (&x)[E], unsigned rank_a, unsigned rank_b,
                                                int gf, int nt) {
  unsigned ra = 0, rb = 0;
  for (int bit = 15; bit >= 0; --bit) {
    const unsigned ta = ra | (1u << bit), tb = rb | (1u << bit);
    unsigned c = count_below_pk16<GV, E>(x, (ta | (tb << 16)) ^ 0x80008000u, gf, nt);
    c = __reduce_add_sync(~0ull, c);
    if ((c & 0xFFFFu) <= rank_a) ra = ta;
    if ((c >> 16) <= rank_b) rb = tb;
  }
  return ra | (rb << 16);
}

// Register budget per instantiation (second __launch_bounds__ argument = waves per SIMD): the LDS
// footprint (5 M complex per block) admits that many blocks per CU anyway.
template <int CPL>
constexpr int row_waves_per_simd() {
  return CPL <= 18 ? 4 : 1;
}
constexpr int kRowMaxWaves = 8;  // waves (row pairs) per block; they share one twiddle table

// One wave per pair of rows.  CPL = complex values per lane = ceil(M / 64).
// GF_ / NT_ >= 0: the slot structure of the row (full groups, tail slots) is a compile-time constant
// (the hot shapes get their own instantiation without the per-group guards); -1: taken from a.w.
// HALO_: 0 = direct transform (K == 0), 1 = periodic halo (K > 0), -1 = decided at run time.
// PLAN_: compile-time FFT plan (StaticFft), 0 = the pass list of RowArgs.
// TO_LDS (k_rowfinal): Delta is not stored to the workspace but left in the wave's FFT buffer as two planar float rows
// (row 2 pair at float 0, row 2 pair + 1 at float rf_lds_row_b(M)); pairs outside the plane and levels the plane's
// config does not filter leave rows of zeros there.  The function contains ONE block-wide barrier (twiddle staging)
// that every wave of the block must reach: it returns (never exits the kernel) on every path.
__host__ __device__ __forceinline__ constexpr int rf_lds_row_b(int M) { return (M + 1) & ~1; }

template <int CPL, int GF_, int NT_, int HALO_, int PLAN_, bool TO_LDS>
__device__ __forceinline__ void rf_pair_body(const RowArgs& a, float2* s_tw, float2* buf, int tid, int nthreads, int lane,
                                             int pair, int plane) {
  const int M = (PLAN_ > 0) ? StaticFft<PLAN_>::M : a.M;
  // P(i): where element i of the row buffer / twiddle table lives (bank swizzle of the power-of-two plans, else i)
  constexpr bool SWZ = StaticFft<PLAN_>::kSwz;
  auto P = [](int i) { return SWZ ? dsx_idx_swz::at(i) : i; };
  const int N = a.w, K = (HALO_ == 0) ? 0 : a.K;
  const bool halo = (HALO_ >= 0) ? (HALO_ != 0) : (K > 0);
  const int npairs = (a.h + 1) >> 1;
  const bool live = pair < npairs;  // (waves past the last pair still help loading the twiddles)
  const int r0 = live ? 2 * pair : 0;
  const bool has_b = (r0 + 1) < a.h;
  const int cfg = a.cfg[plane];
  float* rowa = a.ws + plane * a.ws_plane_stride + a.da_off + (long long)r0 * a.ld;
  float* rowb = rowa + a.ld;

  constexpr int GV = RowSlots<CPL>::GV;
  constexpr int E = RowSlots<CPL>::E;
  typedef typename SlotMask<E>::type mask_t;
  const int gf = (GF_ >= 0) ? GF_ : (N >> 8);                   // full 256-element groups
  const int tail0 = gf << 8;                                    // first tail element
  const int nt = (NT_ >= 0) ? NT_ : ((N - tail0 + 63) >> 6);    // lane-strided tail slots
  const int tn = tail0 + lane;                 // this lane's element of tail slot 0 (+ 64 per slot)

  const bool inactive = a.lvl >= a.lvl_active[cfg];  // this config does not filter this level: Delta = 0 (block-uniform)
  auto zero_rows = [&]() {
    if (live) {
      for (int n = 4 * lane; n < N; n += 4 * kWave) {
        *(float4*)(rowa + n) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_b) *(float4*)(rowb + n) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  if (!TO_LDS && inactive) {  // (block-uniform: no wave of the block reaches the barrier below)
    zero_rows();
    return;
  }
  const bool active = live && !inactive;
  const float thr = a.thr[(long long)plane * a.L + a.lvl];

  // ---- load both rows; background = masked entries zeroed (filtering.py:195-197) -------------
  // issue the row loads, THEN stage the twiddles: the two global latencies overlap
  float4 ra4[GV > 0 ? GV : 1], rb4[GV > 0 ? GV : 1];
  float rta[4], rtb[4];
#pragma unroll
  for (int g = 0; g < GV; ++g) {
    ra4[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    rb4[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active && g < gf) {
      ra4[g] = *(const float4*)(rowa + 256 * g + 4 * lane);
      if (has_b) rb4[g] = *(const float4*)(rowb + 256 * g + 4 * lane);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    rta[k] = 0.f;
    rtb[k] = 0.f;
    if (active && k < nt && tn + 64 * k < N) {
      rta[k] = rowa[tn + 64 * k];
      if (has_b) rtb[k] = rowb[tn + 64 * k];
    }
  }
  for (int i = tid; i < M; i += nthreads) s_tw[P(i)] = a.tw[i];
  __syncthreads();  // the only block-wide barrier: afterwards every wave works on its own rows
  float* const lds_a = (float*)buf;
  float* const lds_b = (float*)buf + rf_lds_row_b(M);
  if (!active) {
    if (TO_LDS) {
      for (int n = 2 * lane; n < N; n += 2 * kWave) {  // N is even for the plans that take this path; see k_rowfinal
        *(float2*)(lds_a + n) = make_float2(0.f, 0.f);
        *(float2*)(lds_b + n) = make_float2(0.f, 0.f);
      }
    }
    return;
  }

  // Background values (masked entries zeroed, filtering.py:195-197) stay in registers as floats; slots past
  // the row end hold +inf (never below a threshold, never a minimum).
  float va[E], vb[E];
  mask_t maska = 0, maskb = 0;
  auto take = [&](int e, float xa, float xb, bool valid) {
    const bool ma = valid && fabsf(xa) > thr, mb = valid && fabsf(xb) > thr;
    if (ma) maska |= ((mask_t)1 << e);
    if (mb) maskb |= ((mask_t)1 << e);
    va[e] = valid ? (ma ? 0.f : xa) : __builtin_huge_valf();
    vb[e] = valid ? (mb ? 0.f : xb) : __builtin_huge_valf();
  };
#pragma unroll
  for (int g = 0; g < GV; ++g) {
    if (g < gf) {
      take(4 * g + 0, ra4[g].x, rb4[g].x, true);
      take(4 * g + 1, ra4[g].y, rb4[g].y, true);
      take(4 * g + 2, ra4[g].z, rb4[g].z, true);
      take(4 * g + 3, ra4[g].w, rb4[g].w, true);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k < nt) take(4 * GV + k, rta[k], rtb[k], tn + 64 * k < N);
  }
  if (!has_b) {  // odd row count: the partner row is all zeros
    for_slots<GV>(gf, nt, [&](int e) { if (vb[e] != __builtin_huge_valf()) vb[e] = 0.f; });
  }

  // ---- exact row medians (np.median, filtering.py:201) ------------------------------------------
  // k-th smallest by bracketing in the VALUE domain: C(t) = #{x < t} is one compare + one add per value
  // and row; [lo, hi) with C(lo) <= k < C(hi) shrinks by regula falsi on (value, count) alternating with
  // the midpoint of the order-preserving integer keys (which bounds the number of steps), until exactly
  // one element -- or one value, for ties -- is left; the answer is then min{x >= lo}.  The zero spike
  // of the masked entries is probed first (C(0) and #{x <= 0}).  Both rows run in lockstep; everything
  // but the per-value compares is wave-uniform.  (Round 1 selected on packed 16-bit key halves by 16-step
  // bisection plus a bucket walk: ~2.1 k vector instructions per row pair with the key packing and
  // unpacking around it; this takes ~0.9 k.)
  float meda = 0.f, medb = 0.f;
  // the medians only enter through the masked positions: skip them for a mask-free pair of rows
  const bool any_mask = __ballot((maska | maskb) != (mask_t)0) != 0ull;
  if (any_mask && !DSX_ABL(a, 1)) {
    const unsigned k1 = (unsigned)(N - 1) >> 1;
    const bool even = (N & 1) == 0;
    auto count2 = [&](float ta, float tb, bool le) -> unsigned {  // (#a < ta) | (#b < tb) << 16, wave totals
#if DSX_MEDIAN_BALLOT
      // one vector compare per value; the wave total is a scalar population count of the compare mask
      unsigned ca = 0, cb = 0;
      for_slots<GV>(gf, nt, [&](int e) {
        ca += (unsigned)__popcll(__ballot(le ? (va[e] <= ta) : (va[e] < ta)));
        cb += (unsigned)__popcll(__ballot(le ? (vb[e] <= tb) : (vb[e] < tb)));
      });
      return ca | (cb << 16);
#else
      unsigned ca = 0, cb = 0;
      for_slots<GV>(gf, nt, [&](int e) {
        ca += (le ? (va[e] <= ta) : (va[e] < ta)) ? 1u : 0u;
        cb += (le ? (vb[e] <= tb) : (vb[e] < tb)) ? 1u : 0u;
      });
      return (unsigned)__builtin_amdgcn_readfirstlane((int)__reduce_add_sync(~0ull, ca | (cb << 16)));
#endif
    };
    // bracket state per row r (0 = a, 1 = b), wave-uniform
    float lo[2], hi[2];
    unsigned clo[2], chi[2];
    bool done[2];
    // f = C(t) - (k + 1/2) at the two ends (Illinois variant of regula falsi: an end that survives two updates of the
    // other one in a row has its f halved, which keeps the false position from creeping towards one side)
    float flo[2], fhi[2];
    int last[2] = {0, 0};
    const float target = (float)k1 + 0.5f;
    const float thr_up = as_f32(as_u32(thr) + 1u);  // next float above thr (thr >= 0): C(thr_up) = N
    {
      const unsigned c0 = count2(0.f, 0.f, false), c0e = count2(0.f, 0.f, true);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const unsigned lt = r ? (c0 >> 16) : (c0 & 0xFFFFu), le = r ? (c0e >> 16) : (c0e & 0xFFFFu);
        done[r] = false;
        if (k1 < lt) {            // the statistic is negative
          lo[r] = -thr; clo[r] = 0u; hi[r] = 0.f; chi[r] = lt;
        } else if (k1 < le) {     // inside the zero spike: [0, denorm_min) holds the value 0 only
          lo[r] = 0.f; clo[r] = lt; hi[r] = as_f32(1u); chi[r] = le; done[r] = true;
        } else {                  // positive: C(denorm_min) = #{x <= 0}
          lo[r] = as_f32(1u); clo[r] = le; hi[r] = thr_up; chi[r] = (unsigned)N;
        }
        flo[r] = (float)clo[r] - target;
        fhi[r] = (float)chi[r] - target;
      }
    }
    // A row is finished when one element -- or, for ties, one value -- is left: C(hi) - C(lo) <= 1, or lo and hi are
    // adjacent floats.  The second test needs the integer keys and is only made where they are computed anyway: on the
    // fallback path (a probe strictly inside (lo, hi) proves that the two are not adjacent).  A finished row simply
    // keeps running until its partner is finished too -- every step preserves C(lo) <= k < C(hi) -- so the loop body
    // carries no per-row "done" predicates (they were most of its ~240 scalar instructions per step).
    bool adj[2] = {done[0], done[1]};
    for (int it = 0; it < 256; ++it) {
      const bool fin0 = adj[0] || chi[0] - clo[0] <= 1u, fin1 = adj[1] || chi[1] - clo[1] <= 1u;
      if (fin0 && fin1) break;
      float t[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        // false position aimed at rank k + 1/2 (the probe only has to lie inside the bracket: the bare v_rcp_f32 instead
        // of an IEEE division -- 12 instructions each -- changes which values are probed, never the result); on every
        // eighth step, and whenever it leaves the bracket, the midpoint of the integer keys halves the number of
        // representable values left (at most 8 x 32 steps in all).  Round 2's plain regula falsi with a key midpoint on
        // every fourth step took 9.75 counts per masked row of the synthetic planes, this takes 8.96 (NumPy model of both
        // loops against np.median on real and adversarial rows: tools/median_model.py)
        float tt = (lo[r] * fhi[r] - hi[r] * flo[r]) * __builtin_amdgcn_rcpf(fhi[r] - flo[r]);
        if ((it & 7) == 7 || !(tt > lo[r] && tt < hi[r])) {  // wave-uniform
          const unsigned kl = f32_key(lo[r]), kh = f32_key(hi[r]);
          if (kh - kl <= 1u) adj[r] = true;
          tt = key_f32(kl + ((kh - kl) >> 1));
        }
        t[r] = tt;
      }
      const unsigned c = count2(t[0], t[1], false);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const unsigned cr = r ? (c >> 16) : (c & 0xFFFFu);
        const float f = (float)cr - target;
        if (cr <= k1) {
          lo[r] = t[r]; clo[r] = cr; flo[r] = f;
          if (last[r] == 1) fhi[r] *= 0.5f;
          last[r] = 1;
        } else {
          hi[r] = t[r]; chi[r] = cr; fhi[r] = f;
          if (last[r] == 2) flo[r] *= 0.5f;
          last[r] = 2;
        }
      }
    }
    // s_k = min{x >= lo};  even N: s_{k+1} = s_k if more than k + 1 values lie below hi, else min{x >= hi}
    float m_lo[2] = {__builtin_huge_valf(), __builtin_huge_valf()}, m_hi[2] = {__builtin_huge_valf(), __builtin_huge_valf()};
    for_slots<GV>(gf, nt, [&](int e) {
      m_lo[0] = min_no_nan(m_lo[0], va[e] >= lo[0] ? va[e] : __builtin_huge_valf());
      m_lo[1] = min_no_nan(m_lo[1], vb[e] >= lo[1] ? vb[e] : __builtin_huge_valf());
      if (even) {
        m_hi[0] = min_no_nan(m_hi[0], va[e] >= hi[0] ? va[e] : __builtin_huge_valf());
        m_hi[1] = min_no_nan(m_hi[1], vb[e] >= hi[1] ? vb[e] : __builtin_huge_valf());
      }
    });
    float sk[2], sk1[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      sk[r] = wave_min_no_nan(m_lo[r]);
      sk1[r] = sk[r];
      if (even) {
        const float nxt = wave_min_no_nan(m_hi[r]);
        sk1[r] = (k1 + 1u < chi[r]) ? sk[r] : nxt;
      }
    }
    meda = even ? 0.5f * (sk[0] + sk1[0]) : sk[0];
    medb = even ? 0.5f * (sk[1] + sk1[1]) : sk[1];
  }  // any_mask
  // The medians are wave-uniform (SGPR); ROCm 7.2's instruction selection crashes when they flow
  // into the selects / LDS stores below, so pin them into VGPRs.
  asm volatile("" : "+v"(meda), "+v"(medb));

  // ---- in-painted rows -> complex buffer u[m] = x[(m - K) mod N], m in [0, N + 2K]; zero above -----
  for_slots<GV>(gf, nt, [&](int e) {
    const int n = (e < 4 * GV) ? 256 * (e >> 2) + 4 * lane + (e & 3) : tn + 64 * (e - 4 * GV);
    if (e < 4 * GV || n < N) {
      const float xa = bit_select(slot_mask(maska, e), meda, va[e]);
      const float xb = bit_select(slot_mask(maskb, e), medb, vb[e]);
      const float2 z = make_float2(xa, xb);
      buf[P(K + n)] = z;
      if (halo) {
        if (n <= K) buf[P(K + N + n)] = z;
        if (n >= N - K) buf[P(n - (N - K))] = z;
      }
    }
  });
  if (halo) {
    for (int m = N + 2 * K + 1 + lane; m < M; m += kWave) buf[P(m)] = make_float2(0.f, 0.f);
  }
  wave_sync();

  if (!DSX_ABL(a, 2)) {
    if constexpr (PLAN_ > 0) StaticFft<PLAN_>::template run_forward<CPL>(buf, s_tw, lane, a.kcut[cfg]);
    else fft_run<CPL>(buf, s_tw, a, lane);
  }

  // ---- V[k] = G1[k] U[k] + G2[k] U[M - k], in place on the pair (k, M - k) ---------------------
  // G1 is real and even, G2[M - k] = conj(G2[k]) (both modes, dsx_plan.h).  The result is stored
  // re/im-swapped: the inverse transform runs through the forward passes.
  if (!DSX_ABL(a, 4)) {
    const float2* g1 = a.g[cfg];
    const float2* g2 = g1 + M;
    const int kcut = a.kcut[cfg];
    for (int k = lane; k <= kcut; k += kWave) {
      const int kr = (k == 0) ? 0 : M - k;
      const float2 u = buf[P(k)], ur = buf[P(kr)];
      const float ga = g1[k].x;
      const float2 gb = g2[k];
      const float2 v = make_float2(ga * u.x + gb.x * ur.x - gb.y * ur.y, ga * u.y + gb.x * ur.y + gb.y * ur.x);
      const float2 vr = make_float2(ga * ur.x + gb.x * u.x + gb.y * u.y, ga * ur.y + gb.x * u.y - gb.y * u.x);
      buf[P(k)] = make_float2(v.y, v.x);
      if (kr != k) buf[P(kr)] = make_float2(vr.y, vr.x);
    }
    // beyond the band limit of the low-pass the product is an exact zero
    for (int k = kcut + 1 + lane; k < M - kcut; k += kWave) buf[P(k)] = make_float2(0.f, 0.f);
    wave_sync();
  }

  if (!DSX_ABL(a, 2)) {
    if constexpr (PLAN_ > 0) StaticFft<PLAN_>::template run_inverse<CPL>(buf, s_tw, lane, a.kcut[cfg]);
    else fft_run<CPL>(buf, s_tw, a, lane);
  }

  // ---- buf = swap(M * LP): row a <- .y, row b <- .x ; Delta = -(1 - mask) LP (filtering.py:215-217)
  maska = opaque_mask(maska);
  maskb = opaque_mask(maskb);
  if (TO_LDS) {
    // every value is read out of the interleaved buffer before the first planar value overwrites it
    float oa[E], ob[E];
    const int no_b = has_b ? 0 : -1;  // odd row count: the partner row of the last pair is all zeros
    for_slots<GV>(gf, nt, [&](int e) {
      const int n = (e < 4 * GV) ? 256 * (e >> 2) + 4 * lane + (e & 3) : tn + 64 * (e - 4 * GV);
      oa[e] = 0.f;
      ob[e] = 0.f;
      if (e < 4 * GV || n < N) {
        const float2 y = buf[P(K + n)];
        oa[e] = bit_select(slot_mask(maska, e), 0.f, -y.y * a.inv_M);
        ob[e] = bit_select(slot_mask(maskb, e) | no_b, 0.f, -y.x * a.inv_M);
      }
    });
    wave_sync();
#pragma unroll
    for (int g = 0; g < GV; ++g) {
      if (g < gf) {
        const int nb0 = 256 * g + 4 * lane;  // (row b starts 8-byte aligned only: 8-byte stores)
        *(float2*)(lds_a + nb0) = make_float2(oa[4 * g], oa[4 * g + 1]);
        *(float2*)(lds_a + nb0 + 2) = make_float2(oa[4 * g + 2], oa[4 * g + 3]);
        *(float2*)(lds_b + nb0) = make_float2(ob[4 * g], ob[4 * g + 1]);
        *(float2*)(lds_b + nb0 + 2) = make_float2(ob[4 * g + 2], ob[4 * g + 3]);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int n = tn + 64 * k;
      if (k < nt && n < N) {
        lds_a[n] = oa[4 * GV + k];
        lds_b[n] = ob[4 * GV + k];
      }
    }
    return;
  }
#pragma unroll
  for (int g = 0; g < GV; ++g) {
    if (g < gf) {
      const int nb0 = 256 * g + 4 * lane;
      float da_[4], db_[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = 4 * g + i;
        const float2 y = buf[P(K + nb0 + i)];
        da_[i] = bit_select(slot_mask(maska, e), 0.f, -y.y * a.inv_M);
        db_[i] = bit_select(slot_mask(maskb, e), 0.f, -y.x * a.inv_M);
      }
#if DSX_NT
      const dsx_f4 oa = {da_[0], da_[1], da_[2], da_[3]}, ob = {db_[0], db_[1], db_[2], db_[3]};
      __builtin_nontemporal_store(oa, (dsx_f4*)(rowa + nb0));
      if (has_b) __builtin_nontemporal_store(ob, (dsx_f4*)(rowb + nb0));
#else
      *(float4*)(rowa + nb0) = make_float4(da_[0], da_[1], da_[2], da_[3]);
      if (has_b) *(float4*)(rowb + nb0) = make_float4(db_[0], db_[1], db_[2], db_[3]);
#endif
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int n = tn + 64 * k;
    if (k < nt && n < N) {
      const int e = 4 * GV + k;
      const float2 y = buf[P(K + n)];
      rowa[n] = bit_select(slot_mask(maska, e), 0.f, -y.y * a.inv_M);
      if (has_b) rowb[n] = bit_select(slot_mask(maskb, e), 0.f, -y.x * a.inv_M);
    }
  }
}

// One wave per pair of rows, kRowMaxWaves pairs per block (they share one twiddle table).
template <int CPL, int GF_ = -1, int NT_ = -1, int HALO_ = -1, int PLAN_ = 0>
__global__ __launch_bounds__(64 * kRowMaxWaves, row_waves_per_simd<CPL>()) void k_rowfilter(RowArgs a) {
  extern __shared__ __attribute__((aligned(16))) float2 dsx_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = (PLAN_ > 0) ? StaticFft<PLAN_>::M : a.M;
  rf_pair_body<CPL, GF_, NT_, HALO_, PLAN_, false>(a, dsx_smem, dsx_smem + (long long)M * (1 + wave), tid, blockDim.x, lane,
                                                   blockIdx.x * (blockDim.x >> 6) + wave, blockIdx.y);
}

// The row filters of SEVERAL coarse levels in one launch (round 3): levels 3 ... 8 of a 2048^2 plane are six launches of
// a few hundred to a few thousand row pairs, each behind a launch gap; their rows all fit the CPL = 6 class (M <= 384),
// so one grid carries them all and every block picks the RowArgs of its level from the block index.
constexpr int kRowMultiMax = 8;
struct RowMultiArgs {
  int nlev;
  int blk_end[kRowMultiMax];  // cumulative block counts
  RowArgs lv[kRowMultiMax];
};

template <int CPL>
__global__ __launch_bounds__(64 * kRowMaxWaves, row_waves_per_simd<CPL>()) void k_rowfilter_multi(RowMultiArgs args) {
  extern __shared__ __attribute__((aligned(16))) float2 dsx_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int ent = 0;
  while (ent + 1 < args.nlev && (int)blockIdx.x >= args.blk_end[ent]) ++ent;  // block-uniform, scalar unit
  const RowArgs& a = args.lv[ent];
  const int bx = (int)blockIdx.x - (ent > 0 ? args.blk_end[ent - 1] : 0);
  rf_pair_body<CPL, -1, -1, -1, 0, false>(a, dsx_smem, dsx_smem + (long long)a.M * (1 + wave), tid, blockDim.x, lane,
                                          bx * (blockDim.x >> 6) + wave, blockIdx.y);
}

// ---- K4w: the row filter for rows longer than one wave holds (levels wider than 64 x 36 = 2 304 coefficients: planes
// wider than ~4 600 pixels; round 3) -------------------------------------------------------------------------------
// One BLOCK of kWideThreads lanes per pair of rows; the complex row buffer fills the block's LDS (up to 144 KiB), the
// twiddles are read from global memory; the passes are those of k_rowfilter with the butterflies dealt to the whole
// block and a block barrier where the wave kernel has a wave barrier.  The exact median is a 32-step bisection of the
// order-preserving integer keys (both rows in lockstep, one block-wide count per step): simple and bounded; these
// shapes are rare (no SmartSPIM camera is that wide) and the kernel is written for the reference's "any width"
// (filtering.py:206), not for speed.  Same arithmetic as rf_pair_body everywhere else.
constexpr int kWideThreads = 512;
constexpr int kWideCpl = 36;                              // complex values per lane
constexpr int kWideMaxLen = kWideThreads * kWideCpl;      // 18 432: transform lengths the kernel takes

__device__ __forceinline__ unsigned wide_block_sum(unsigned v, unsigned* s_red, int& turn) {
  v = __reduce_add_sync(~0ull, v);
  unsigned* slot = s_red + ((turn & 1) ? (kWideThreads / 64) : 0);  // alternating halves: one barrier per reduction
  if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned t = 0;
#pragma unroll
  for (int w = 0; w < kWideThreads / 64; ++w) t += slot[w];
  ++turn;
  return t;
}
__device__ __forceinline__ unsigned wide_block_min(unsigned v, unsigned* s_red, int& turn) {
  v = __reduce_min_sync(~0ull, v);
  unsigned* slot = s_red + ((turn & 1) ? (kWideThreads / 64) : 0);
  if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned t = ~0u;
#pragma unroll
  for (int w = 0; w < kWideThreads / 64; ++w) t = min(t, slot[w]);
  ++turn;
  return t;
}

template <int R, int CPL, int JK = (R - 1) / 2>
__device__ __forceinline__ void fft_pass_block(float2* buf, const float2* tw, int M, int s, float inv_s, int tid) {
  constexpr int MAXB = (CPL + R - 1) / R;
  const int nb = M / R;
  const bool unit_tw = (s * R == M);
  dsx_c32 v[MAXB][R];
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    const int b = tid + kWideThreads * i;
    if (b < nb) dsx_bfly_load<R, JK>((const dsx_c32*)buf, b, nb, v[i]);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    const int b = tid + kWideThreads * i;
    if (b < nb) dsx_bfly_store<R, JK>((dsx_c32*)buf, (const dsx_c32*)tw, b, s, inv_s, v[i], unit_tw);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
}
template <int CPL>
__device__ __forceinline__ void fft_pass_generic_block(float2* buf, const float2* tw, int M, int s, float inv_s, int R,
                                                       int tid) {
  float2 acc[CPL];
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int o = tid + kWideThreads * i;
    acc[i] = make_float2(0.f, 0.f);
    if (o < M) {
      const dsx_c32 r = dsx_generic_output((const dsx_c32*)buf, (const dsx_c32*)tw, o, M, s, inv_s, R);
      acc[i] = make_float2(r.x, r.y);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int o = tid + kWideThreads * i;
    if (o < M) buf[o] = acc[i];
  }
  __syncthreads();
}
template <int CPL>
__device__ __forceinline__ void fft_run_block(float2* buf, const float2* tw, const RowArgs& a, int tid_in) {
  int s = 1;
  for (int pi = 0; pi < a.npass; ++pi) {
    const int R = a.radix[pi];
    const float inv_s = 1.0f / (float)s;
    int tid = tid_in, M = a.M;
    asm volatile("" : "+v"(tid), "+s"(M));  // (see fft_run)
    switch (R) {
      case 2: fft_pass_block<2, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 3: fft_pass_block<3, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 4: fft_pass_block<4, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 5: fft_pass_block<5, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 6: fft_pass_block<6, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 8: fft_pass_block<8, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 9: fft_pass_block<9, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 10: fft_pass_block<10, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 12: fft_pass_block<12, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 15: fft_pass_block<15, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 16: fft_pass_block<16, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 20: fft_pass_block<20, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 25: fft_pass_block<25, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 7: fft_pass_block<7, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 11: fft_pass_block<11, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 13: fft_pass_block<13, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 17: fft_pass_block<17, CPL>(buf, tw, M, s, inv_s, tid); break;
      case 19: fft_pass_block<19, CPL>(buf, tw, M, s, inv_s, tid); break;
      default: fft_pass_generic_block<CPL>(buf, tw, M, s, inv_s, R, tid); break;
    }
    s *= R;
  }
}

__global__ __launch_bounds__(kWideThreads) void k_rowfilter_wide(RowArgs a) {
  extern __shared__ __attribute__((aligned(16))) float2 dsx_smem[];  // [M]
  __shared__ unsigned s_red[2 * (kWideThreads / 64)];
  constexpr int E = kWideCpl;
  float2* const buf = dsx_smem;
  const int tid = threadIdx.x, plane = blockIdx.y, pair = blockIdx.x;
  const int M = a.M, N = a.w, K = a.K;
  const bool halo = K > 0;
  const int r0 = 2 * pair;
  const bool has_b = (r0 + 1) < a.h;
  const int cfg = a.cfg[plane];
  float* rowa = a.ws + plane * a.ws_plane_stride + a.da_off + (long long)r0 * a.ld;
  float* rowb = rowa + a.ld;
  if (a.lvl >= a.lvl_active[cfg]) {  // this config does not filter this level: Delta = 0 (block-uniform)
    for (int n = tid; n < N; n += kWideThreads) {
      rowa[n] = 0.f;
      if (has_b) rowb[n] = 0.f;
    }
    return;
  }
  const float thr = a.thr[(long long)plane * a.L + a.lvl];
  // background values (masked entries zeroed, filtering.py:195-197) as order-preserving keys; slots past the row: ~0
  unsigned ka[E], kb[E];
  unsigned long long maska = 0ull, maskb = 0ull;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int n = tid + kWideThreads * e;
    ka[e] = kb[e] = ~0u;
    if (n < N) {
      const float xa = rowa[n], xb = has_b ? rowb[n] : 0.f;
      const bool ma = fabsf(xa) > thr, mb = has_b && fabsf(xb) > thr;
      if (ma) maska |= 1ull << e;
      if (mb) maskb |= 1ull << e;
      ka[e] = f32_key(ma ? 0.f : xa);
      kb[e] = f32_key(mb ? 0.f : xb);
    }
  }
  int turn = 0;
  float meda = 0.f, medb = 0.f;
  const bool any_mask = wide_block_sum((maska | maskb) != 0ull ? 1u : 0u, s_red, turn) != 0u;
  if (any_mask) {  // (block-uniform) exact medians, np.median (filtering.py:201)
    const unsigned k1 = (unsigned)(N - 1) >> 1;
    auto count_below = [&](unsigned ta, unsigned tb) {  // (#a < ta) | (#b < tb) << 16 over the block (N < 65 536)
      unsigned c = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) c += (ka[e] < ta ? 1u : 0u) + (kb[e] < tb ? 0x10000u : 0u);
      return wide_block_sum(c, s_red, turn);
    };
    unsigned pa = 0u, pb = 0u;  // largest T with #{key < T} <= k1: the key of the element of rank k1
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned ta = pa | (1u << bit), tb = pb | (1u << bit);
      const unsigned c = count_below(ta, tb);
      if ((c & 0xFFFFu) <= k1) pa = ta;
      if ((c >> 16) <= k1) pb = tb;
    }
    float ska = key_f32(pa), skb = key_f32(pb);
    if ((N & 1) == 0) {  // even length: mean of the elements of rank k1 and k1 + 1
      const unsigned cle = count_below(pa + 1u, pb + 1u);  // #{key <= P} (a valid key is < ~0)
      unsigned na = ~0u, nb_ = ~0u;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if (ka[e] > pa) na = min(na, ka[e]);
        if (kb[e] > pb) nb_ = min(nb_, kb[e]);
      }
      na = wide_block_min(na, s_red, turn);
      nb_ = wide_block_min(nb_, s_red, turn);
      const float sa1 = ((cle & 0xFFFFu) >= k1 + 2u) ? ska : key_f32(na);
      const float sb1 = ((cle >> 16) >= k1 + 2u) ? skb : key_f32(nb_);
      meda = 0.5f * (ska + sa1);
      medb = 0.5f * (skb + sb1);
    } else {
      meda = ska;
      medb = skb;
    }
  }
  // ---- in-painted rows -> complex buffer u[m] = x[(m - K) mod N], m in [0, N + 2K]; zero above ----------------------
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int n = tid + kWideThreads * e;
    if (n < N) {
      const float2 z = make_float2(((maska >> e) & 1ull) ? meda : key_f32(ka[e]), ((maskb >> e) & 1ull) ? medb : key_f32(kb[e]));
      buf[K + n] = z;
      if (halo) {
        if (n <= K) buf[K + N + n] = z;
        if (n >= N - K) buf[n - (N - K)] = z;
      }
    }
  }
  if (halo) {
    for (int m = N + 2 * K + 1 + tid; m < M; m += kWideThreads) buf[m] = make_float2(0.f, 0.f);
  }
  __syncthreads();
  fft_run_block<E>(buf, a.tw, a, tid);
  {  // V[k] = G1[k] U[k] + G2[k] U[M - k], stored re/im-swapped (see rf_pair_body)
    const float2* g1 = a.g[cfg];
    const float2* g2 = g1 + M;
    const int kcut = a.kcut[cfg];
    for (int k = tid; k <= kcut; k += kWideThreads) {
      const int kr = (k == 0) ? 0 : M - k;
      const float2 u = buf[k], ur = buf[kr];
      const float ga = g1[k].x;
      const float2 gb = g2[k];
      const float2 v = make_float2(ga * u.x + gb.x * ur.x - gb.y * ur.y, ga * u.y + gb.x * ur.y + gb.y * ur.x);
      const float2 vr = make_float2(ga * ur.x + gb.x * u.x + gb.y * u.y, ga * ur.y + gb.x * u.y - gb.y * u.x);
      buf[k] = make_float2(v.y, v.x);
      if (kr != k) buf[kr] = make_float2(vr.y, vr.x);
    }
    for (int k = kcut + 1 + tid; k < M - kcut; k += kWideThreads) buf[k] = make_float2(0.f, 0.f);
    __syncthreads();
  }
  fft_run_block<E>(buf, a.tw, a, tid);
  // ---- Delta = -(1 - mask) LP (filtering.py:215-217): row a <- .y, row b <- .x ----------------------------------------
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int n = tid + kWideThreads * e;
    if (n < N) {
      const float2 y = buf[K + n];
      rowa[n] = ((maska >> e) & 1ull) ? 0.f : -y.y * a.inv_M;
      if (has_b) rowb[n] = ((maskb >> e) & 1ull) ? 0.f : -y.x * a.inv_M;
    }
  }
}

// ================================================================================================
// K5/K6: synthesis level, "marching" form.  One wave owns 256 result columns (4 per lane)
// and streams down the coefficient rows p: the row (axis-1) synthesis of c_1 / Delta_1 is done
// in registers (3 taps), a 3-row register window feeds the column (axis-0) synthesis, and the
// finish  (1 + x) exp(c0) + 1  (+ shading, cast) is fused.  No LDS, no barriers.
// ================================================================================================
struct FinalArgs {
  const float* ws;
  long long ws_plane_stride;
  long long c_off, d_off;
  int hc, wc, ldc, ldd;  // coefficient shape, pitch of c (aa buffer), pitch of Delta (da buffer)
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
  int nstrips, nseg, rows_per_seg;  // segments in coefficient rows p
  float* ws_out;      // MODE 2 (pyramid level): c_{l-1} destination = ws_out + plane stride + out_off
  long long out_off;
  int ldout;
  // FUSE (k_inv_march<IN_KIND, true>): c_1 is synthesised from level 2 inside the wave (it never touches
  // memory): c2 / Delta_2 buffers of level 2; the c_off buffer is not read
  long long c2_off, d2_off;
  int hc2, wc2, ldc2, ldd2, has_c2;
  int ablate;  // diagnosis only (DSX_ABLATE): 32 = the final kernel stores nothing
  int pair_io;  // FUSE, uint16 planes: 16-byte pixel loads / result stores by lane pairs (see inv_march_body)
};
constexpr int kC1Pitch = 136;  // floats per c_1 ring row: 33 lanes x 4 columns (130 are needed)

struct FinalRawC {  // raw coefficients q .. q+3 of one row of c and Delta
  float2 c01, c23, d01, d23;
};
template <int IN_KIND>
struct FinalRawI {  // raw pixels x0 .. x0+3 of one plane row
  uint2 u;
  float4 f;
  uint4 q;  // PAIR: 8 pixels of row 2p + (lane & 1) starting at the lane pair's first column
};

__device__ __forceinline__ void final_coeff_row(const FinalArgs& a, const float* base, int pitch, int p, int q,
                                                bool on, bool vec, float2& lo, float2& hi) {
  lo = make_float2(0.f, 0.f);
  hi = make_float2(0.f, 0.f);
  if (!on || p >= a.hc) return;
  const float* row = base + (long long)p * pitch;
  if (vec) {
    lo = *(const float2*)(row + q);
    hi = *(const float2*)(row + q + 2);
  } else {
    if (q < a.wc) lo.x = row[q];
    if (q + 1 < a.wc) lo.y = row[q + 1];
    if (q + 2 < a.wc) hi.x = row[q + 2];
    if (q + 3 < a.wc) hi.y = row[q + 3];
  }
}

// row synthesis of 4 result columns x = 4t .. 4t+3 from coefficients q .. q+3 (q = 2t):
// out[2 qq + b] = in[qq] rl[4+b] + in[qq+1] rl[2+b] + in[qq+2] rl[b]; columns (0,1) and (2,3) as packed pairs
__device__ __forceinline__ void final_xsynth(float2 lo, float2 hi, dsx_f2 (&o)[2]) {
  constexpr float RL[6] = DSX_REC_LO;
  const dsx_f2 t45 = {RL[4], RL[5]}, t23 = {RL[2], RL[3]}, t01 = {RL[0], RL[1]};
  o[0] = pk_fma(lo.x, t45, pk_fma(lo.y, t23, hi.x * t01));
  o[1] = pk_fma(lo.y, t45, pk_fma(hi.x, t23, hi.y * t01));
}
// a0 t0 + a1 t1 + a2 t2 + d0 u0 + d1 u1 + d2 u2, accumulated left to right
__device__ __forceinline__ dsx_f2 pk_dot6(dsx_f2 a0, float t0, dsx_f2 a1, float t1, dsx_f2 a2, float t2,
                                          dsx_f2 d0, float u0, dsx_f2 d1, float u1, dsx_f2 d2, float u2) {
  dsx_f2 v = a0 * t0;
  v = pk_fma(a1, t1, v);
  v = pk_fma(a2, t2, v);
  v = pk_fma(d0, u0, v);
  v = pk_fma(d1, u1, v);
  v = pk_fma(d2, u2, v);
  return v;
}

// Two results as packed uint16: v_cvt_u32_f32 saturates below at 0 (negative, NaN) and v_cvt_pk_u16_u32 above at 65535 --
// the clip of the reference's uint16 assignment without a compare per pixel.
__device__ __forceinline__ unsigned pack_u16_sat(float lo, float hi) {
  typedef unsigned short dsx_u16x2 __attribute__((ext_vector_type(2)));
  union { dsx_u16x2 v; unsigned u; } c;
  c.v = __builtin_amdgcn_cvt_pk_u16((unsigned)lo, (unsigned)hi);
  return c.u;
}

// c0l = c0 * log2(e) (the factor is folded into the axis-0 synthesis taps of the last level)
template <bool SHADE>
__device__ __forceinline__ float final_px(const FinalArgs& a, float c0l, float x, float dark, float flat) {
  float v = fmaf(1.0f + x, __builtin_amdgcn_exp2f(c0l), 1.0f);  // exp(log(1 + x) + c0) + 1  (filtering.py:222)
  if (SHADE) {                                                    // flatfield_correction, filtering.py:399-412
    v = (v > dark) ? (v - dark) : 0.f;
    // v / flat as reciprocal, product, one residual correction and the fix-up instruction of the division sequence
    // (zero / infinite / NaN operands come out as the IEEE quotient): 5 instructions where the full sequence has 11,
    // which was a fifth of this epilogue.  Differs from the correctly rounded quotient by at most one ulp, rarely.
    const float rc = __builtin_amdgcn_rcpf(flat);
    const float q = v * rc;
    v = __builtin_amdgcn_div_fixupf(fmaf(fmaf(-q, flat, v), rc, q), flat, v);
    v = fminf(fmaxf(v, 0.f), 65535.f);
  }
  return v;
}

// Body of k_inv_march for one wave.  FAST: every lane loads its coefficients / pixels with aligned
// vector loads, unconditionally (row indices are clamped: rows past the end only feed result rows
// that are never stored) -- see fwd_march_body for why this is a separate instantiation.
// PAIR (uint16 pixels, width a multiple of 8): pixel rows are read and result rows written by lane PAIRS -- the even
// lane moves 16 bytes of row 2p, the odd lane 16 bytes of row 2p + 1, and the halves are exchanged with one DPP
// swap each way.  A streaming kernel with 8-byte accesses per lane tops out at 3.9 TB/s on this chip, with 16-byte
// accesses at 6.3 TB/s (tools/bw_access_width.py), and this kernel's time is proportional to its bytes.
template <int IN_KIND, bool FAST, bool SHADE, bool FUSE = false, bool PAIR = false>
__device__ __forceinline__ void inv_march_body(const FinalArgs& a, int lane, int strip, int seg, int plane,
                                               float (*s_c1)[kC1Pitch] = nullptr) {
  constexpr float RL0[6] = DSX_REC_LO;
  constexpr float RH0[6] = DSX_REC_HI;
  // last level: the result feeds exp2(), so the axis-0 taps carry the factor log2(e)
  constexpr float KS = (IN_KIND == 2) ? 1.0f : 1.44269504088896340736f;
  constexpr float RL[6] = {RL0[0] * KS, RL0[1] * KS, RL0[2] * KS, RL0[3] * KS, RL0[4] * KS, RL0[5] * KS};
  constexpr float RH[6] = {RH0[0] * KS, RH0[1] * KS, RH0[2] * KS, RH0[3] * KS, RH0[4] * KS, RH0[5] * KS};
  const int np = (a.hout + 1) >> 1;  // coefficient rows that produce result rows
  const int p_begin = seg * a.rows_per_seg;
  const int p_end = min(np, p_begin + a.rows_per_seg);
  const int x0 = kMarchCols * strip + 4 * lane;
  const int q = x0 >> 1;
  // FUSE keeps the lanes right of the plane in the wave: their loads use a clamped (valid) address and
  // their results are never stored
  const int xl = FUSE ? min(x0, a.W - 4) : x0;
  const int ql = xl >> 1;
  const float* cbase = a.ws + plane * a.ws_plane_stride + a.c_off;
  const float* dbase = a.ws + plane * a.ws_plane_stride + a.d_off;
  const bool pyr = a.has_pyr != 0;
  const bool has_c = pyr && a.has_c;
  const bool vec_c = q + 3 < a.wc;
  const bool vec_in = (IN_KIND != 2) && ((a.W & 3) == 0) && (x0 + 3 < a.W);
  const int out_pitch = (IN_KIND == 2) ? a.ldout : a.wout;
  // pyramid levels write into padded aa rows: a 16-byte store may run into the margin columns
  const bool vec_out = ((out_pitch & 3) == 0) && ((IN_KIND == 2) ? (x0 + 3 < a.wout + 8) : (x0 + 3 < a.wout));
  const long long img_plane = plane * a.img_plane_stride;
  // flat / dark rows as aligned float4 loads when the pitches and base addresses allow it (wave-uniform)
  // FAST waves address rows through buffer descriptors: wave-uniform row offset + 32-bit lane offset (see dsx_rsrc)
  const __amdgpu_buffer_rsrc_t rs_c = dsx_rsrc(cbase), rs_d = dsx_rsrc(dbase);
  const unsigned vo_c = (unsigned)q * 4u, vo_d = (unsigned)ql * 4u;
  const int out_es = (a.out_dtype == 0) ? 2 : 4;
  const __amdgpu_buffer_rsrc_t rs_out =
      dsx_rsrc(IN_KIND == 2 ? (const char*)(a.ws_out + plane * a.ws_plane_stride + a.out_off)
                            : (const char*)a.out + plane * a.out_plane_stride * out_es);
  const unsigned vo_out = (unsigned)x0 * (IN_KIND == 2 ? 4u : (unsigned)out_es);
  constexpr int IES = (IN_KIND == 0) ? 2 : 4;
  const __amdgpu_buffer_rsrc_t rs_img = dsx_rsrc(IN_KIND == 2 ? (const char*)a.ws : (const char*)a.img + img_plane * IES);
  const unsigned vo_img = (unsigned)xl * IES;
  const int xs = min(x0, max(a.wout - 4, 0));
  const __amdgpu_buffer_rsrc_t rs_dark = dsx_rsrc(SHADE ? a.dark : a.ws), rs_flat = dsx_rsrc(SHADE ? a.flat : a.ws);
  const unsigned vo_shade = (unsigned)xs * 4u;
  const bool shade_vec = SHADE && (a.wout & 3) == 0 && (a.dark_ld & 3) == 0 && a.wout >= 4 &&
                         (((uintptr_t)a.dark | (uintptr_t)a.flat) & 15) == 0;

  // ---- FUSE: c_1 rows from level 2, lanes 0..32 (4 columns each), into the per-wave LDS ring ----------
  // One step P loads level-2 row P + 2 and emits c_1 rows 2P, 2P + 1 (plain taps: c_1 is a log-image
  // correction of level 1, the exp2 scaling only enters in the last synthesis step).
  const int q2 = (kMarchCols / 4) * strip + 2 * lane;  // first level-2 coefficient column of the lane
  const bool l2_lane = FUSE && lane < 33 && (kMarchCols / 2) * strip + 4 * lane < a.wc + 4;
  const float* c2base = FUSE ? a.ws + plane * a.ws_plane_stride + a.c2_off : nullptr;
  const float* d2base = FUSE ? a.ws + plane * a.ws_plane_stride + a.d2_off : nullptr;
  const __amdgpu_buffer_rsrc_t rs_c2 = dsx_rsrc(FUSE ? c2base : a.ws), rs_d2 = dsx_rsrc(FUSE ? d2base : a.ws);
  const unsigned vo_2 = (unsigned)max(q2, 0) * 4u;
  dsx_f2 A2[3][2], D2[3][2];
  FinalRawC n2;
  int next_P = 0, c1_ready = 0;  // c_1 rows < c1_ready are in the ring
  auto l2_load = [&](int P2) {
    FinalRawC r;
    r.c01 = r.c23 = r.d01 = r.d23 = make_float2(0.f, 0.f);
    if (l2_lane) {
      const int pr = min(P2, a.hc2 - 1);  // rows past the end only feed c_1 rows that are never used
      if (a.has_c2) {
        const dsx_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_c2, vo_2, (unsigned)(pr * a.ldc2) * 4u, 0);
        r.c01 = make_float2(__uint_as_float(u.x), __uint_as_float(u.y));
        r.c23 = make_float2(__uint_as_float(u.z), __uint_as_float(u.w));
      }
      const dsx_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_d2, vo_2, (unsigned)(pr * a.ldd2) * 4u, 0);
      r.d01 = make_float2(__uint_as_float(u.x), __uint_as_float(u.y));
      r.d23 = make_float2(__uint_as_float(u.z), __uint_as_float(u.w));
    }
    return r;
  };
  auto l2_step = [&](int P) {
    final_xsynth(n2.c01, n2.c23, A2[2]);
    final_xsynth(n2.d01, n2.d23, D2[2]);
    n2 = l2_load(P + 3);
    dsx_f2 ev[2], od[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      ev[h] = pk_dot6(A2[0][h], RL0[4], A2[1][h], RL0[2], A2[2][h], RL0[0], D2[0][h], RH0[4], D2[1][h], RH0[2],
                      D2[2][h], RH0[0]);
      od[h] = pk_dot6(A2[0][h], RL0[5], A2[1][h], RL0[3], A2[2][h], RL0[1], D2[0][h], RH0[5], D2[1][h], RH0[3],
                      D2[2][h], RH0[1]);
      A2[0][h] = A2[1][h]; A2[1][h] = A2[2][h];
      D2[0][h] = D2[1][h]; D2[1][h] = D2[2][h];
    }
    if (lane < 33) {
      *(float4*)&s_c1[(2 * P) & (kRingRows - 1)][4 * lane] = make_float4(ev[0].x, ev[0].y, ev[1].x, ev[1].y);
      *(float4*)&s_c1[(2 * P + 1) & (kRingRows - 1)][4 * lane] = make_float4(od[0].x, od[0].y, od[1].x, od[1].y);
    }
    wave_sync();
  };
  // c_1 coefficients q .. q+3 of row p for the row synthesis (FUSE: from the ring)
  auto ring_c = [&](int p, FinalRawC& r) {
    while (c1_ready <= p) {  // wave-uniform
      l2_step(next_P);
      ++next_P;
      c1_ready += 2;
    }
    const float* row = s_c1[p & (kRingRows - 1)] + 2 * lane;
    r.c01 = *(const float2*)row;
    r.c23 = *(const float2*)(row + 2);
  };
  if (FUSE) {
    const int P0 = p_begin >> 1;  // the host makes segments start at even rows
    next_P = P0;
    c1_ready = 2 * P0;
    const FinalRawC w0 = l2_load(P0), w1 = l2_load(P0 + 1);
    final_xsynth(w0.c01, w0.c23, A2[0]);
    final_xsynth(w0.d01, w0.d23, D2[0]);
    final_xsynth(w1.c01, w1.c23, A2[1]);
    final_xsynth(w1.d01, w1.d23, D2[1]);
    n2 = l2_load(P0 + 2);
  }

  auto issue_c = [&](int p) {
    FinalRawC r;
    if (FAST) {
      // rows are padded (>= 4 spare columns): coefficients past the end only feed discarded results
      const int pr = min(p, a.hc - 1);
      r.c01 = r.c23 = make_float2(0.f, 0.f);
      if (has_c && !FUSE) {  // uniform: the coarsest level has no approximation correction
        const dsx_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_c, vo_c, (unsigned)(pr * a.ldc) * 4u, 0);
        r.c01 = make_float2(__uint_as_float(u.x), __uint_as_float(u.y));
        r.c23 = make_float2(__uint_as_float(u.z), __uint_as_float(u.w));
      }
      // one 16-byte access at an 8-byte aligned address (coefficients q .. q+3, q even)
      const dsx_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_d, vo_d, (unsigned)(pr * a.ldd) * 4u, 0);
      r.d01 = make_float2(__uint_as_float(u.x), __uint_as_float(u.y));
      r.d23 = make_float2(__uint_as_float(u.z), __uint_as_float(u.w));
      return r;
    }
    final_coeff_row(a, cbase, a.ldc, p, q, has_c, vec_c, r.c01, r.c23);
    final_coeff_row(a, dbase, a.ldd, p, q, pyr, vec_c, r.d01, r.d23);
    return r;
  };
  const bool odd_lane = (lane & 1) != 0;
  const int oddm = odd_lane_mask(lane);
  const int xb = min(kMarchCols * strip + 8 * (lane >> 1), a.W - 8);  // PAIR: first column of the lane pair (clamped)
  // PAIR lane offsets: the odd lane of a pair takes the row below (pixels) / stores the row below (results)
  const unsigned vo_pair0 = (unsigned)xb * 2u, vo_pair = vo_pair0 + (odd_lane ? (unsigned)a.W * 2u : 0u);
  const unsigned vo_pair_out = (unsigned)xb * 2u + (odd_lane ? (unsigned)a.wout * 2u : 0u);
  auto issue_i = [&](int gy) {
    FinalRawI<IN_KIND> r;
    r.u = make_uint2(0u, 0u);
    r.f = make_float4(0.f, 0.f, 0.f, 0.f);
    r.q = make_uint4(0u, 0u, 0u, 0u);
    if (PAIR) {
      if ((gy & 1) == 0) {  // one load per row pair, issued with the even row
        const int ge = min(gy, a.H - 1);                       // wave-uniform
        const unsigned vo = (ge + 1 < a.H) ? vo_pair : vo_pair0;  // the odd lane's row exists (scalar condition)
        const dsx_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_img, vo, (unsigned)(ge * a.W) * 2u, kBufNT);
        r.q = make_uint4(u.x, u.y, u.z, u.w);
      }
      return r;
    }
    if (IN_KIND != 2 && FAST) {
      const unsigned so = (unsigned)(min(gy, a.H - 1) * a.W) * IES;
      if (IN_KIND == 0) {
        const dsx_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(rs_img, vo_img, so, kBufNT);
        r.u = make_uint2(u.x, u.y);
      } else {
        const dsx_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_img, vo_img, so, 0);
        r.f = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
      }
    } else if (IN_KIND != 2 && vec_in) {
      const long long off = img_plane + (long long)min(gy, a.H - 1) * a.W + xl;
      if (IN_KIND == 0) {
        const dsx_u32x2 u = __builtin_nontemporal_load((const dsx_u32x2*)((const uint16_t*)a.img + off));
        r.u = make_uint2(u.x, u.y);
      }
      else r.f = *(const float4*)((const float*)a.img + off);
    }
    return r;
  };

  dsx_f2 A[3][2], D[3][2];  // row-synthesised c and Delta rows p, p+1, p+2 (window), columns as packed pairs
  {
    FinalRawC r0 = issue_c(p_begin), r1 = issue_c(p_begin + 1);
    if (FUSE) {
      ring_c(p_begin, r0);
      ring_c(p_begin + 1, r1);
    }
    final_xsynth(r0.c01, r0.c23, A[0]);
    final_xsynth(r0.d01, r0.d23, D[0]);
    final_xsynth(r1.c01, r1.c23, A[1]);
    final_xsynth(r1.d01, r1.d23, D[1]);
  }

  auto emit_row = [&](int gy, const FinalRawI<IN_KIND>& raw, const dsx_f2 (&c0p)[2], unsigned (*pk_out)[2] = nullptr) {
    const float c0[4] = {c0p[0].x, c0p[0].y, c0p[1].x, c0p[1].y};
    if (gy >= a.hout && pk_out == nullptr) return;
    if (IN_KIND == 2) {
      float* dst = a.ws_out + plane * a.ws_plane_stride + a.out_off + (long long)gy * a.ldout + x0;
      if (vec_out) {
        const dsx_u32x4 o4 = {__float_as_uint(c0[0]), __float_as_uint(c0[1]), __float_as_uint(c0[2]), __float_as_uint(c0[3])};
        __builtin_amdgcn_raw_buffer_store_b128(o4, rs_out, vo_out, (unsigned)(gy * a.ldout) * 4u, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (x0 + e < a.wout) dst[e] = c0[e];
      }
      return;
    }
    float px[4];
    if (FAST || vec_in) {
      if (IN_KIND == 0) {
        px[0] = (float)(raw.u.x & 0xFFFFu); px[1] = (float)(raw.u.x >> 16);
        px[2] = (float)(raw.u.y & 0xFFFFu); px[3] = (float)(raw.u.y >> 16);
      } else {
        px[0] = raw.f.x; px[1] = raw.f.y; px[2] = raw.f.z; px[3] = raw.f.w;
      }
    } else {
      const int sy = min(gy, a.H - 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int sx = min(x0 + e, a.W - 1);  // an odd plane grows by one replicated column / row
        const long long off = img_plane + (long long)sy * a.W + sx;
        px[e] = (IN_KIND == 0) ? (float)((const uint16_t*)a.img)[off] : ((const float*)a.img)[off];
      }
    }
    float dk[4] = {0.f, 0.f, 0.f, 0.f}, fl[4] = {1.f, 1.f, 1.f, 1.f};
    if (SHADE) {
      if (shade_vec) {  // one 16-byte load per plane and lane (clamped address for lanes right of the plane)
        const dsx_u32x4 d4 = __builtin_amdgcn_raw_buffer_load_b128(rs_dark, vo_shade, (unsigned)(gy * a.dark_ld) * 4u, 0);
        const dsx_u32x4 f4 = __builtin_amdgcn_raw_buffer_load_b128(rs_flat, vo_shade, (unsigned)(gy * a.wout) * 4u, 0);
        dk[0] = __uint_as_float(d4.x); dk[1] = __uint_as_float(d4.y); dk[2] = __uint_as_float(d4.z); dk[3] = __uint_as_float(d4.w);
        fl[0] = __uint_as_float(f4.x); fl[1] = __uint_as_float(f4.y); fl[2] = __uint_as_float(f4.z); fl[3] = __uint_as_float(f4.w);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int gx = min(x0 + e, a.wout - 1);
          dk[e] = a.dark[(long long)gy * a.dark_ld + gx];
          fl[e] = a.flat[(long long)gy * a.wout + gx];
        }
      }
    }
    float r[4];
    if (SHADE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = final_px<SHADE>(a, c0[e], px[e], dk[e], fl[e]);  // all four: stores are masked
    } else {
      // final_px for pixel pairs: (1 + x) and the multiply-add as packed FP32 (same roundings as the scalar form)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const dsx_f2 one = {1.0f, 1.0f};
        const dsx_f2 x1 = dsx_f2{px[2 * h], px[2 * h + 1]} + one;
        const dsx_f2 ex = {__builtin_amdgcn_exp2f(c0[2 * h]), __builtin_amdgcn_exp2f(c0[2 * h + 1])};
        const dsx_f2 v = pk_fma(x1, ex, one);
        r[2 * h] = v.x;
        r[2 * h + 1] = v.y;
      }
    }
    if (pk_out != nullptr) {  // PAIR: hand the packed uint16 row back, the caller stores row pairs
      (*pk_out)[0] = pack_u16_sat(r[0], r[1]);
      (*pk_out)[1] = pack_u16_sat(r[2], r[3]);
      return;
    }
    const long long o = plane * a.out_plane_stride + (long long)gy * a.wout + x0;
    if (DSX_ABL(a, 32)) {
      if (r[0] + r[1] + r[2] + r[3] == -12345.f) ((float*)a.out)[0] = 0.f;  // keeps the arithmetic alive
    } else if (a.out_dtype == 0) {
      if (vec_out) {
        dsx_u32x2 pk;
        pk.x = pack_u16_sat(r[0], r[1]);
        pk.y = pack_u16_sat(r[2], r[3]);
        // written once, not read again here
        __builtin_amdgcn_raw_buffer_store_b64(pk, rs_out, vo_out, (unsigned)(gy * a.wout) * 2u, kBufNT);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (x0 + e < a.wout) ((uint16_t*)a.out)[o + e] = (uint16_t)(unsigned)fminf(r[e], 65535.f);
      }
    } else {
      if (vec_out) {
        const dsx_u32x4 o4 = {__float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2]), __float_as_uint(r[3])};
        __builtin_amdgcn_raw_buffer_store_b128(o4, rs_out, vo_out, (unsigned)(gy * a.wout) * 4u, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (x0 + e < a.wout) ((float*)a.out)[o + e] = r[e];
      }
    }
  };

  // one coefficient row p -> result rows 2p, 2p+1; (A0, A1, A2) = window rows p, p+1, p+2
  auto step = [&](int p, const FinalRawC& rc_in, const FinalRawI<IN_KIND>& ri0, const FinalRawI<IN_KIND>& ri1,
                  dsx_f2 (&A0)[2], dsx_f2 (&A1)[2], dsx_f2 (&A2)[2], dsx_f2 (&D0)[2], dsx_f2 (&D1)[2],
                  dsx_f2 (&D2)[2]) {
    FinalRawC rc = rc_in;
    if (FUSE) ring_c(p + 2, rc);
    final_xsynth(rc.c01, rc.c23, A2);
    final_xsynth(rc.d01, rc.d23, D2);
    dsx_f2 even[2], odd[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // out[2p + b] = sum_j A[p+j] rl[4 - 2j + b] + D[p+j] rh[4 - 2j + b], two columns per instruction
      even[h] = pk_dot6(A0[h], RL[4], A1[h], RL[2], A2[h], RL[0], D0[h], RH[4], D1[h], RH[2], D2[h], RH[0]);
      odd[h] = pk_dot6(A0[h], RL[5], A1[h], RL[3], A2[h], RL[1], D0[h], RH[5], D1[h], RH[3], D2[h], RH[1]);
    }
    if (PAIR) {
      // ri0.q: this lane's 16 bytes of row 2p + (lane & 1); the own half of the own row stays, the other
      // row's own half comes from the neighbour
      const unsigned sx = bit_select(oddm, ri0.q.x, ri0.q.z), sy = bit_select(oddm, ri0.q.y, ri0.q.w);  // what the neighbour needs
      const unsigned gx = swap_adjacent(sx), gyv = swap_adjacent(sy);
      FinalRawI<IN_KIND> r0 = ri0, r1 = ri0;
      r0.u = make_uint2(bit_select(oddm, gx, ri0.q.x), bit_select(oddm, gyv, ri0.q.y));      // row 2p, own 4 columns
      r1.u = make_uint2(bit_select(oddm, ri0.q.z, gx), bit_select(oddm, ri0.q.w, gyv));      // row 2p + 1
      if (a.out_dtype == 0 && !DSX_ABL(a, 32)) {
        unsigned pe[2], po[2];
        emit_row(2 * p, r0, even, &pe);
        emit_row(2 * p + 1, r1, odd, &po);
        // even lane stores row 2p: [own | neighbour's] even-row pack; odd lane row 2p + 1: [neighbour's | own]
        const unsigned tx = swap_adjacent(bit_select(oddm, pe[0], po[0])), ty = swap_adjacent(bit_select(oddm, pe[1], po[1]));
        dsx_u32x4 o4;
        o4.x = bit_select(oddm, tx, pe[0]);
        o4.y = bit_select(oddm, ty, pe[1]);
        o4.z = bit_select(oddm, po[0], tx);
        o4.w = bit_select(oddm, po[1], ty);
        const int gyl = 2 * p + (odd_lane ? 1 : 0);
        if (gyl < a.hout && x0 < a.wout)
          __builtin_amdgcn_raw_buffer_store_b128(o4, rs_out, vo_pair_out, (unsigned)(2 * p * a.wout) * 2u, kBufNT);
      } else {
        emit_row(2 * p, r0, even);
        emit_row(2 * p + 1, r1, odd);
      }
    } else {
      emit_row(2 * p, ri0, even);
      emit_row(2 * p + 1, ri1, odd);
    }
  };

  // software prefetch with rotating registers: slot k of (nc, ni) holds coefficient row p+2+k and plane rows
  // 2(p+k), 2(p+k)+1 of step p+k; as soon as the step has consumed its slot, the slot's loads for the step three
  // rows further are issued into the same registers.  (A double-buffered "current / next group" scheme keeps the
  // same three steps in flight with twice the registers -- 24 more, which cost the fourth wave per SIMD.)
  // Out-of-range rows load a clamped row; their results are never stored.
  FinalRawC nc[3];
  FinalRawI<IN_KIND> ni[6];
#pragma unroll
  for (int r = 0; r < 3; ++r) nc[r] = issue_c(p_begin + 2 + r);
#pragma unroll
  for (int r = 0; r < 6; ++r) ni[r] = issue_i(2 * p_begin + r);
  for (int p = p_begin; p < p_end; p += 3) {
    const bool more = p + 3 < p_end;  // wave-uniform
    step(p, nc[0], ni[0], ni[1], A[0], A[1], A[2], D[0], D[1], D[2]);
    if (more) {
      nc[0] = issue_c(p + 5);
      ni[0] = issue_i(2 * (p + 3));
      ni[1] = issue_i(2 * (p + 3) + 1);
    }
    if (p + 1 < p_end) step(p + 1, nc[1], ni[2], ni[3], A[1], A[2], A[0], D[1], D[2], D[0]);
    if (more) {
      nc[1] = issue_c(p + 6);
      ni[2] = issue_i(2 * (p + 3) + 2);
      ni[3] = issue_i(2 * (p + 3) + 3);
    }
    if (p + 2 < p_end) step(p + 2, nc[2], ni[4], ni[5], A[2], A[0], A[1], D[2], D[0], D[1]);
    if (more) {
      nc[2] = issue_c(p + 7);
      ni[4] = issue_i(2 * (p + 3) + 4);
      ni[5] = issue_i(2 * (p + 3) + 5);
    }
  }
}

// IN_KIND: 0 = last level, uint16 pixels; 1 = last level, float32 pixels;
//          2 = pyramid level: writes c_{l-1} (float32, hout x wout, pitch ldout) into the workspace
template <int IN_KIND, bool FUSE = false, int WPB = 4>
__global__ __launch_bounds__(64 * WPB, (FUSE && IN_KIND == 0) ? DSX_INV_MINW : 1) void k_inv_march(FinalArgs a) {
  __shared__ __attribute__((aligned(16))) float s_c1[FUSE ? WPB : 1][FUSE ? kRingRows : 1][FUSE ? kC1Pitch : 4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps row math on the SALU
  const int item = blockIdx.x * WPB + wave;
  if (item >= a.nstrips * a.nseg) return;
  const int strip = item % a.nstrips, seg = item / a.nstrips;
  const int plane = blockIdx.y;
  if (seg * a.rows_per_seg >= ((a.hout + 1) >> 1)) return;
  const int x0 = kMarchCols * strip + 4 * lane;
  if (FUSE) {
    // host guarantees W % 4 == 0 and even coefficient pitches; lanes right of the plane stay in the wave
    // (lanes 0..32 synthesise c_1 for everybody) and simply store nothing
    float (*ring)[kC1Pitch] = (float (*)[kC1Pitch])s_c1[wave];
    const bool pair = IN_KIND == 0 && a.pair_io != 0;  // host: uint16 planes, W % 8 == 0, 16-byte aligned bases
    if (IN_KIND == 0 && pair) {
      if (a.flat != nullptr) inv_march_body<IN_KIND, true, true, true, IN_KIND == 0>(a, lane, strip, seg, plane, ring);
      else inv_march_body<IN_KIND, true, false, true, IN_KIND == 0>(a, lane, strip, seg, plane, ring);
      return;
    }
    if (a.flat != nullptr) inv_march_body<IN_KIND, true, true, true>(a, lane, strip, seg, plane, ring);
    else inv_march_body<IN_KIND, true, false, true>(a, lane, strip, seg, plane, ring);
    return;
  }
  if (x0 >= a.wout) return;
  // coefficient rows are padded, so every lane can load 4 coefficients; pixel rows are not
  const bool lane_fast = (a.has_pyr != 0) && ((a.ldc & 1) == 0) && ((a.ldd & 1) == 0) &&
                         (IN_KIND == 2 || (((a.W & 3) == 0) && (x0 + 3 < a.W)));
  const bool fast = __all(lane_fast) != 0;
  if (IN_KIND != 2 && a.flat != nullptr) {
    if (fast) inv_march_body<IN_KIND, true, IN_KIND != 2>(a, lane, strip, seg, plane);
    else inv_march_body<IN_KIND, false, IN_KIND != 2>(a, lane, strip, seg, plane);
  } else {
    if (fast) inv_march_body<IN_KIND, true, false>(a, lane, strip, seg, plane);
    else inv_march_body<IN_KIND, false, false>(a, lane, strip, seg, plane);
  }
}

// ================================================================================================
// K4 + K5/K6 fused (round 3): level-1 row filter AND final synthesis in one kernel -- Delta_1 never touches HBM.
//
// Round 2's chain wrote Delta_1 (4.2 MB per 2048^2 plane) out of k_rowfilter and read it back in the final kernel: 8.4
// of the 61.6 MB a plane moved, on a chain whose 4-stream mix runs at the HBM ceiling of its access pattern.  Here a
// block of kRfWaves waves first filters kRfWaves row pairs of cH_1 exactly as k_rowfilter does (rf_pair_body, same
// code, same bits) but leaves the 2 kRfWaves Delta_1 rows in its FFT buffers (LDS, planar).  After ONE more block
// barrier the same waves turn into the strips of the final kernel: wave s owns result columns [256 s, 256 s + 256) and
// runs the 2 kRfWaves - 2 coefficient rows p whose three source rows p .. p + 2 the block holds (the last row pair
// is filtered again by the next block: + 1 / 7 of the level-1 FFT work, vector instructions the memory-bound mix has
// to spare).  The c_1 rows come from level 2 (c_2, Delta_2) as in k_inv_march<.., FUSE>, but every lane synthesises
// the four c_1 columns it consumes itself from level-2 coefficients Q .. Q + 3, Q = 64 s + lane (overlapping 16-byte
// loads at 4-byte granularity, L2 hits): no LDS ring, no exchange between lanes, nothing but loads, arithmetic and
// stores in the synthesis phase.  Pixel rows are read and result rows written by lane pairs (16 bytes per lane) as in
// inv_march_body<.., PAIR>.  The arithmetic of both phases is the unfused chain's, operation for operation: results
// are bit-identical (tests/test_gpu_parity.py::test_fused_rowfinal_is_bit_identical_to_the_unfused_chain).
//
// Shapes: uint16 planes, W % 8 == 0, H even, <= kRfWaves strips (W <= 2048), and a level-1 row-filter plan with a
// compile-time instantiation: 1026 = 19 * 9 * 6 direct (2048-wide planes; two blocks per CU), 1002 values embedded in
// 2048 (2000-wide: the production tile) and 902 in 1815 (1800-wide) -- one block of 8 waves per CU for those two,
// against ONE block of 4 waves for their k_rowfilter.  Everything else takes k_rowfilter + k_inv_march.
// ================================================================================================
constexpr int kRfWaves = 8;
constexpr int kRfRows = 2 * kRfWaves - 2;  // coefficient rows p per block

struct RowFinalArgs {
  RowArgs r;
  FinalArgs f;
};

// Loads of the synthesis phase that do not depend on Delta_1: issued BEFORE the block barrier between the two phases, so
// that a wave that is done with its row pair has its first pixel rows and level-2 rows in flight while it waits.
struct RowFinalPre {
  uint4 ni[2];         // pixel rows of the first two steps (a third costs the registers that make the kernel spill)

};

template <bool SHADE, bool PREFETCH_ONLY = false>
__device__ __forceinline__ void rowfinal_synth(const FinalArgs& a, const float2* smem, int M, int lane, int strip, int plane,
                                               int p_begin, int p_end, int row0, RowFinalPre& pre) {
  constexpr float RL0[6] = DSX_REC_LO;
  constexpr float RH0[6] = DSX_REC_HI;
  constexpr float KS = 1.44269504088896340736f;  // the result feeds exp2(): the axis-0 taps of level 1 carry log2(e)
  constexpr float RL[6] = {RL0[0] * KS, RL0[1] * KS, RL0[2] * KS, RL0[3] * KS, RL0[4] * KS, RL0[5] * KS};
  constexpr float RH[6] = {RH0[0] * KS, RH0[1] * KS, RH0[2] * KS, RH0[3] * KS, RH0[4] * KS, RH0[5] * KS};
  const int x0 = kMarchCols * strip + 4 * lane;
  const int xl = min(x0, a.W - 4);  // lanes right of the plane: clamped (valid) addresses, results never stored
  const int ql = xl >> 1;           // first Delta_1 / c_1 column of the lane
  const bool odd_lane = (lane & 1) != 0;
  const int oddm = odd_lane_mask(lane);
  const int xb = min(kMarchCols * strip + 8 * (lane >> 1), a.W - 8);  // first column of the lane pair (clamped)
  const __amdgpu_buffer_rsrc_t rs_img = dsx_rsrc((const char*)a.img + plane * a.img_plane_stride * 2);
  const __amdgpu_buffer_rsrc_t rs_out = dsx_rsrc((const char*)a.out + plane * a.out_plane_stride * 2);
  const unsigned vo_pair0 = (unsigned)xb * 2u, vo_pair = vo_pair0 + (odd_lane ? (unsigned)a.W * 2u : 0u);
  const unsigned vo_pair_out = (unsigned)xb * 2u + (odd_lane ? (unsigned)a.wout * 2u : 0u);
  const int xs = min(x0, max(a.wout - 4, 0));
  const __amdgpu_buffer_rsrc_t rs_dark = dsx_rsrc(SHADE ? a.dark : a.ws), rs_flat = dsx_rsrc(SHADE ? a.flat : a.ws);
  const unsigned vo_shade = (unsigned)xs * 4u;
  const bool shade_vec = SHADE && (a.wout & 3) == 0 && (a.dark_ld & 3) == 0 && a.wout >= 4 &&
                         (((uintptr_t)a.dark | (uintptr_t)a.flat) & 15) == 0;

  // ---- level 2 -> c_1, per lane: coefficients Q .. Q + 3 give c_1 columns 2 Q .. 2 Q + 3 = ql .. ql + 3 -------------
  const int Q = ql >> 1;
  const float* c2base = a.ws + plane * a.ws_plane_stride + a.c2_off;
  const float* d2base = a.ws + plane * a.ws_plane_stride + a.d2_off;
  const __amdgpu_buffer_rsrc_t rs_c2 = dsx_rsrc(c2base), rs_d2 = dsx_rsrc(d2base);
  const unsigned vo_2 = (unsigned)Q * 4u;
  auto l2_load = [&](int P2) {
    FinalRawC r;
    r.c01 = r.c23 = make_float2(0.f, 0.f);
    const int pr = min(P2, a.hc2 - 1);  // rows past the end only feed c_1 rows that are never used
    if (a.has_c2) {
      const dsx_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_c2, vo_2, (unsigned)(pr * a.ldc2) * 4u, 0);
      r.c01 = make_float2(__uint_as_float(u.x), __uint_as_float(u.y));
      r.c23 = make_float2(__uint_as_float(u.z), __uint_as_float(u.w));
    }
    const dsx_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_d2, vo_2, (unsigned)(pr * a.ldd2) * 4u, 0);
    r.d01 = make_float2(__uint_as_float(u.x), __uint_as_float(u.y));
    r.d23 = make_float2(__uint_as_float(u.z), __uint_as_float(u.w));
    return r;
  };
  auto issue_i = [&](int gy) {  // 16 bytes of pixel row gy + (lane & 1), one load per row pair
    const int ge = min(gy, a.H - 1);                          // wave-uniform
    const unsigned vo = (ge + 1 < a.H) ? vo_pair : vo_pair0;  // the odd lane's row exists (scalar condition)
    const dsx_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_img, vo, (unsigned)(ge * a.W) * 2u, kBufNT);
    return make_uint4(u.x, u.y, u.z, u.w);
  };
  int next_P = p_begin >> 1, c1_ready = p_begin;  // c_1 rows < c1_ready have been produced (p_begin is even)
  if (PREFETCH_ONLY) {
#pragma unroll
    for (int r = 0; r < 2; ++r) pre.ni[r] = issue_i(2 * (p_begin + r));
    return;
  }
  dsx_f2 A2[3][2], D2[3][2];
  dsx_f2 c1e[2], c1o[2];  // c_1 rows 2 P, 2 P + 1 of the last level-2 step (columns ql .. ql + 3 as packed pairs)
  {
    const FinalRawC w0 = l2_load(next_P), w1 = l2_load(next_P + 1);
    final_xsynth(w0.c01, w0.c23, A2[0]);
    final_xsynth(w0.d01, w0.d23, D2[0]);
    final_xsynth(w1.c01, w1.c23, A2[1]);
    final_xsynth(w1.d01, w1.d23, D2[1]);
  }
  FinalRawC n2 = l2_load(next_P + 2);
  auto l2_step = [&]() {  // level-2 rows P .. P + 2 -> c_1 rows 2 P, 2 P + 1 (plain taps: a log-image correction)
    final_xsynth(n2.c01, n2.c23, A2[2]);
    final_xsynth(n2.d01, n2.d23, D2[2]);
    n2 = l2_load(next_P + 3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      c1e[h] = pk_dot6(A2[0][h], RL0[4], A2[1][h], RL0[2], A2[2][h], RL0[0], D2[0][h], RH0[4], D2[1][h], RH0[2],
                       D2[2][h], RH0[0]);
      c1o[h] = pk_dot6(A2[0][h], RL0[5], A2[1][h], RL0[3], A2[2][h], RL0[1], D2[0][h], RH0[5], D2[1][h], RH0[3],
                       D2[2][h], RH0[1]);
      A2[0][h] = A2[1][h]; A2[1][h] = A2[2][h];
      D2[0][h] = D2[1][h]; D2[1][h] = D2[2][h];
    }
    ++next_P;
    c1_ready += 2;
  };
  // row-synthesised c_1 row p (rows are asked for in increasing order, each once)
  auto c1_row = [&](int p, dsx_f2 (&out)[2]) {
    if (c1_ready <= p) l2_step();  // wave-uniform
    const dsx_f2 lo = (p & 1) ? c1o[0] : c1e[0], hi = (p & 1) ? c1o[1] : c1e[1];
    final_xsynth(make_float2(lo.x, lo.y), make_float2(hi.x, hi.y), out);
  };
  // row-synthesised Delta_1 row r out of the block's FFT buffers (planar rows, see rf_pair_body<.., TO_LDS>)
  auto d1_row = [&](int r, dsx_f2 (&out)[2]) {
    const int rr = r - row0;  // 0 .. 2 kRfWaves - 1 (wave-uniform)
    const float* row = (const float*)(smem + (long long)M * (1 + (rr >> 1))) + ((rr & 1) ? rf_lds_row_b(M) : 0) + ql;
    const float2 lo = *(const float2*)row, hi = *(const float2*)(row + 2);
    final_xsynth(lo, hi, out);
  };

  dsx_f2 A[3][2], D[3][2];  // windows: rows p, p + 1, p + 2
  uint4 ni[3] = {pre.ni[0], pre.ni[1], issue_i(2 * (p_begin + 2))};
  c1_row(p_begin, A[0]);
  d1_row(p_begin, D[0]);
  c1_row(p_begin + 1, A[1]);
  d1_row(p_begin + 1, D[1]);

  auto px_row = [&](int gy, uint2 u, const dsx_f2 (&c0p)[2], unsigned (&pk)[2]) {
    const float c0[4] = {c0p[0].x, c0p[0].y, c0p[1].x, c0p[1].y};
    const float px[4] = {(float)(u.x & 0xFFFFu), (float)(u.x >> 16), (float)(u.y & 0xFFFFu), (float)(u.y >> 16)};
    float r[4];
    if (SHADE) {
      float dk[4], fl[4];
      const int gys = min(gy, a.hout - 1);
      if (shade_vec) {
        const dsx_u32x4 d4 = __builtin_amdgcn_raw_buffer_load_b128(rs_dark, vo_shade, (unsigned)(gys * a.dark_ld) * 4u, 0);
        const dsx_u32x4 f4 = __builtin_amdgcn_raw_buffer_load_b128(rs_flat, vo_shade, (unsigned)(gys * a.wout) * 4u, 0);
        dk[0] = __uint_as_float(d4.x); dk[1] = __uint_as_float(d4.y); dk[2] = __uint_as_float(d4.z); dk[3] = __uint_as_float(d4.w);
        fl[0] = __uint_as_float(f4.x); fl[1] = __uint_as_float(f4.y); fl[2] = __uint_as_float(f4.z); fl[3] = __uint_as_float(f4.w);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int gx = min(x0 + e, a.wout - 1);
          dk[e] = a.dark[(long long)gys * a.dark_ld + gx];
          fl[e] = a.flat[(long long)gys * a.wout + gx];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = final_px<SHADE>(a, c0[e], px[e], dk[e], fl[e]);
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const dsx_f2 one = {1.0f, 1.0f};
        const dsx_f2 x1 = dsx_f2{px[2 * h], px[2 * h + 1]} + one;
        const dsx_f2 ex = {__builtin_amdgcn_exp2f(c0[2 * h]), __builtin_amdgcn_exp2f(c0[2 * h + 1])};
        const dsx_f2 v = pk_fma(x1, ex, one);
        r[2 * h] = v.x;
        r[2 * h + 1] = v.y;
      }
    }
    pk[0] = pack_u16_sat(r[0], r[1]);
    pk[1] = pack_u16_sat(r[2], r[3]);
  };
  // one coefficient row p -> result rows 2p, 2p + 1; (A0, A1, A2) = window rows p, p + 1, p + 2
  auto step = [&](int p, const uint4& ri, dsx_f2 (&A0)[2], dsx_f2 (&A1)[2], dsx_f2 (&A2w)[2], dsx_f2 (&D0)[2],
                  dsx_f2 (&D1)[2], dsx_f2 (&D2w)[2]) {
    c1_row(p + 2, A2w);
    d1_row(p + 2, D2w);
    dsx_f2 even[2], odd[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      even[h] = pk_dot6(A0[h], RL[4], A1[h], RL[2], A2w[h], RL[0], D0[h], RH[4], D1[h], RH[2], D2w[h], RH[0]);
      odd[h] = pk_dot6(A0[h], RL[5], A1[h], RL[3], A2w[h], RL[1], D0[h], RH[5], D1[h], RH[3], D2w[h], RH[1]);
    }
    // ri: this lane's 16 bytes of pixel row 2p + (lane & 1); the own half of the own row stays, the other row's
    // own half comes from the neighbour (see inv_march_body<.., PAIR>)
    const unsigned sx = bit_select(oddm, ri.x, ri.z), sy = bit_select(oddm, ri.y, ri.w);
    const unsigned gx = swap_adjacent(sx), gyv = swap_adjacent(sy);
    const uint2 r0 = make_uint2(bit_select(oddm, gx, ri.x), bit_select(oddm, gyv, ri.y));  // row 2p, own 4 columns
    const uint2 r1 = make_uint2(bit_select(oddm, ri.z, gx), bit_select(oddm, ri.w, gyv));  // row 2p + 1
    unsigned pe[2], po[2];
    px_row(2 * p, r0, even, pe);
    px_row(2 * p + 1, r1, odd, po);
    const unsigned tx = swap_adjacent(bit_select(oddm, pe[0], po[0])), ty = swap_adjacent(bit_select(oddm, pe[1], po[1]));
    dsx_u32x4 o4;
    o4.x = bit_select(oddm, tx, pe[0]);
    o4.y = bit_select(oddm, ty, pe[1]);
    o4.z = bit_select(oddm, po[0], tx);
    o4.w = bit_select(oddm, po[1], ty);
    const int gyl = 2 * p + (odd_lane ? 1 : 0);
    if (gyl < a.hout && x0 < a.wout)
      __builtin_amdgcn_raw_buffer_store_b128(o4, rs_out, vo_pair_out, (unsigned)(2 * p * a.wout) * 2u, kBufNT);
  };
  for (int p = p_begin; p < p_end; p += 3) {
    const bool more = p + 3 < p_end;  // wave-uniform
    step(p, ni[0], A[0], A[1], A[2], D[0], D[1], D[2]);
    if (more) ni[0] = issue_i(2 * (p + 3));
    if (p + 1 < p_end) step(p + 1, ni[1], A[1], A[2], A[0], D[1], D[2], D[0]);
    if (more) ni[1] = issue_i(2 * (p + 4));
    if (p + 2 < p_end) step(p + 2, ni[2], A[2], A[0], A[1], D[2], D[0], D[1]);
    if (more) ni[2] = issue_i(2 * (p + 5));
  }
}

// PLAN_ / CPL / GF_ / NT_ / HALO_ as in k_rowfilter; the block's row pairs are [kRfWaves - 1) * blockIdx.x + wave
template <int CPL, int GF_, int NT_, int HALO_, int PLAN_>
__global__ __launch_bounds__(64 * kRfWaves, CPL <= 18 ? 4 : 2) void k_rowfinal(RowFinalArgs a) {
  extern __shared__ __attribute__((aligned(16))) float2 dsx_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int M = StaticFft<PLAN_>::M;
  const int plane = blockIdx.y;
  // DSX_RF_XCD (off, see the switch): workgroup i of a launch runs on compute die i mod 8 (each die has its own L2).
  // Neighbouring blocks of a plane share rows -- the row pair both filter, 2 of the 10 level-2 rows -- so the blocks
  // of one die are made neighbours: die d takes the d-th run of gridDim.x / 8 consecutive blocks.
#if DSX_RF_XCD
  const int per_die = gridDim.x >> 3;  // the host launches a multiple of 8 blocks per plane
  const int bx = (blockIdx.x & 7) * per_die + (blockIdx.x >> 3);
#else
  const int bx = blockIdx.x;
#endif
  const int pair0 = (kRfWaves - 1) * bx;  // first row pair of the block; its coefficient rows start at 2 pair0
  rf_pair_body<CPL, GF_, NT_, HALO_, PLAN_, true>(a.r, dsx_smem, dsx_smem + (long long)M * (1 + wave), tid, 64 * kRfWaves, lane,
                                                  pair0 + wave, plane);
  const int np = (a.f.hout + 1) >> 1;
  const int p_begin = 2 * pair0, p_end = min(np, p_begin + kRfRows);
  const bool synth = wave < a.f.nstrips && p_begin < p_end;  // wave-uniform
  RowFinalPre pre;
  asm volatile("" ::: "memory");  // keep the prefetch loads (and their registers) out of the row-filter phase
  __builtin_amdgcn_sched_barrier(0);
  if (synth) rowfinal_synth<false, true>(a.f, dsx_smem, M, lane, wave, plane, p_begin, p_end, 2 * pair0, pre);
  __syncthreads();  // all 2 kRfWaves Delta_1 rows of the block are in LDS
  if (!synth) return;
  if (a.f.flat != nullptr) rowfinal_synth<true>(a.f, dsx_smem, M, lane, wave, plane, p_begin, p_end, 2 * pair0, pre);
  else rowfinal_synth<false>(a.f, dsx_smem, M, lane, wave, plane, p_begin, p_end, 2 * pair0, pre);
}

}  // namespace dsx
#endif  // DSX_KERNELS_H
