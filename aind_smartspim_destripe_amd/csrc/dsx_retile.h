// dsx_retile.h -- the data-format steps either side of the stripe filter, gfx950.
//
//   k_bricks_to_planes / k_planes_to_bricks : Zarr chunk ("brick") order <-> dense [Z, H, W] planes, uint16.
//       The reference gathers (1,1,64,128,128) chunks into a block through zarr's NumPy indexing and
//       scatters the filtered block back the same way (zarr_destriper.py:1066-1074, 336); here the
//       decompressed chunks are uploaded as they lie in the store and re-tiled in HBM.
//   k_downsample2 : one 2x2x2 windowed-mean level of the multiscale pyramid, uint16 -> uint16
//       (compute_pyramid, zarr_destriper.py:365-407; xarray_multiscale.reducers.windowed_mean with
//       preserve_dtype=True: float64 mean, then astype(uint16) == floor(sum / 8)).
//
// All three are pure HBM streaming: one workgroup per row segment, 16-byte accesses, no LDS.  Row / slab
// indices come from blockIdx (scalar registers), so the only per-lane integer division is x / cx.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsx {

struct BrickArgs {
  const uint16_t* src;
  uint16_t* dst;
  int Z, H, W;        // dense stack
  int cz, cy, cx;     // brick shape
  int nbz, nby, nbx;  // bricks per axis (ceil)
  int z0;             // first z of the dense stack inside the brick grid (multiple of cz not required)
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int VEC>
struct U16Vec;
template <>
struct U16Vec<8> { using type = u32x4; };
template <>
struct U16Vec<4> { using type = u32x2; };
template <>
struct U16Vec<2> { using type = uint32_t; };
template <>
struct U16Vec<1> { using type = uint16_t; };

__device__ __forceinline__ size_t brick_offset(const BrickArgs& a, int z, int y, int x) {
  const int bz = z / a.cz, iz = z - bz * a.cz;
  const int by = y / a.cy, iy = y - by * a.cy;
  const int bx = x / a.cx, ix = x - bx * a.cx;
  const size_t brick = ((size_t)bz * a.nby + by) * a.nbx + bx;
  return ((brick * a.cz + iz) * a.cy + iy) * (size_t)a.cx + ix;
}

// grid: (ceil(W / VEC / 256), H, Z); one thread per VEC pixels of a dense row.
template <int VEC>
__global__ __launch_bounds__(256) void k_bricks_to_planes(BrickArgs a) {
  using V = typename U16Vec<VEC>::type;
  const int x = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (x >= a.W) return;
  const int y = blockIdx.y, z = blockIdx.z;
  const size_t s = brick_offset(a, z + a.z0, y, x);
  const size_t d = ((size_t)z * a.H + y) * a.W + x;
  *reinterpret_cast<V*>(a.dst + d) = __builtin_nontemporal_load(reinterpret_cast<const V*>(a.src + s));
}

// grid: (ceil(nbx * cx / VEC / 256), nby * cy, Zb) with Zb = number of brick-grid z rows covered.
// Brick positions outside the dense stack get the fill value 0 (what zarr stores for partial edge chunks).
template <int VEC>
__global__ __launch_bounds__(256) void k_planes_to_bricks(BrickArgs a) {
  using V = typename U16Vec<VEC>::type;
  const int x = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (x >= a.nbx * a.cx) return;
  const int y = blockIdx.y, zb = blockIdx.z;  // zb: z inside the brick grid
  const int z = zb - a.z0;
  V v = (V)0;
  if (z >= 0 && z < a.Z && y < a.H && x < a.W)
    v = *reinterpret_cast<const V*>(a.src + ((size_t)z * a.H + y) * a.W + x);
  __builtin_nontemporal_store(v, reinterpret_cast<V*>(a.dst + brick_offset(a, zb, y, x)));
}

struct DownArgs {
  const uint16_t* src;
  uint16_t* dst;
  int Z, Y, X;     // source
  int Zo, Yo, Xo;  // destination = floor(source / 2)
};

__device__ __forceinline__ uint32_t pair_sum(uint32_t w) { return (w & 0xffffu) + (w >> 16); }

// grid: (ceil(Xo / 4 / 256), Yo, Zo); one thread per 4 output voxels = 2 x 2 x 8 source voxels.
// VEC8 == true needs X % 8 == 0 (16-byte aligned rows); otherwise scalar loads.
template <bool VEC8>
__global__ __launch_bounds__(256) void k_downsample2(DownArgs a) {
  const int xo = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (xo >= a.Xo) return;
  const int yo = blockIdx.y, zo = blockIdx.z;
  const size_t row = (size_t)a.X, slab = (size_t)a.Y * a.X;
  const uint16_t* p = a.src + (size_t)(2 * zo) * slab + (size_t)(2 * yo) * row + 2 * xo;
  uint32_t s[4] = {0, 0, 0, 0};
  if (VEC8 && xo + 4 <= a.Xo) {
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + dz * slab + dy * row));
        s[0] += pair_sum(v.x); s[1] += pair_sum(v.y); s[2] += pair_sum(v.z); s[3] += pair_sum(v.w);
      }
    u32x2 o;
    o.x = (s[0] >> 3) | ((s[1] >> 3) << 16);
    o.y = (s[2] >> 3) | ((s[3] >> 3) << 16);
    __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(a.dst + ((size_t)zo * a.Yo + yo) * a.Xo + xo));
  } else {
    const int n = min(4, a.Xo - xo);
    for (int i = 0; i < n; ++i) {
      uint32_t t = 0;
      for (int dz = 0; dz < 2; ++dz)
        for (int dy = 0; dy < 2; ++dy) t += (uint32_t)p[dz * slab + dy * row + 2 * i] + p[dz * slab + dy * row + 2 * i + 1];
      a.dst[((size_t)zo * a.Yo + yo) * a.Xo + xo + i] = (uint16_t)(t >> 3);
    }
  }
}

// ---- a11 as a stand-alone call: flatfield_correction() of one plane (filtering.py:338-414) ------------
// x = x > dark ? x - dark : 0 (assigned back into the image, so integer planes truncate, :400-403);
// x / flat - baseline; clip to [0, 65535]; truncate to uint16.  Inside the filter the same arithmetic
// is fused into the last synthesis kernel; this kernel serves direct callers of the reference function.
struct ShadeArgs {
  const void* src;
  const float* flat;
  const float* dark;  // [dark_h][dark_w], cropped to the plane by indexing
  uint16_t* dst;
  int H, W, dark_w;
  float baseline;
  const float* baseline_rows;  // nullable: one value per plane row (a 2-D plane broadcasts baseline[:, None], :393-398)
};

template <bool U16>
__global__ __launch_bounds__(256) void k_shade(ShadeArgs a) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= a.W) return;
  const int y = blockIdx.y;
  const size_t i = (size_t)y * a.W + x;
  const float v = U16 ? (float)reinterpret_cast<const uint16_t*>(a.src)[i] : reinterpret_cast<const float*>(a.src)[i];
  const float d = a.dark[(size_t)y * a.dark_w + x];
  float t = v > d ? v - d : 0.f;
  if (U16) t = truncf(t);
  float c = t / a.flat[i] - (a.baseline_rows ? a.baseline_rows[y] : a.baseline);
  c = fminf(fmaxf(c, 0.f), 65535.f);
  a.dst[i] = (uint16_t)c;
}

// ---- a2 as a stand-alone call: get_foreground_background_mean() (filtering.py:54-88) ------------------
// mask = sigmoid(float16((float16(x) - 400) / 20)) > threshold.  Everything after float16(x) is monotone,
// so the host turns the threshold into the smallest float16 pixel value that passes (`cutoff`) and the
// kernel compares float16(x) >= cutoff (round-to-nearest-even conversion, as NumPy's astype).
// acc[0..1] = sum of foreground / background pixels (double), cnt[0..1] = their counts.
struct FgBgArgs {
  const void* src;
  uint8_t* mask;  // nullable
  double* acc;
  unsigned long long* cnt;
  size_t n;
  float cutoff;
};

template <bool U16>
__global__ __launch_bounds__(256) void k_fgbg(FgBgArgs a) {
  double s[2] = {0.0, 0.0};
  unsigned long long c[2] = {0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
    const float v = U16 ? (float)reinterpret_cast<const uint16_t*>(a.src)[i] : reinterpret_cast<const float*>(a.src)[i];
    const bool fg = __half2float(__float2half_rn(v)) >= a.cutoff;
    if (a.mask) a.mask[i] = fg ? 1 : 0;
    s[fg ? 0 : 1] += (double)v;
    c[fg ? 0 : 1] += 1;
  }
  __shared__ double ss[2][4];
  __shared__ unsigned long long sc[2][4];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    for (int o = 32; o > 0; o >>= 1) {
      s[k] += __shfl_down(s[k], o);
      c[k] += __shfl_down(c[k], o);
    }
    if ((threadIdx.x & 63) == 0) { ss[k][threadIdx.x >> 6] = s[k]; sc[k][threadIdx.x >> 6] = c[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int k = threadIdx.x;
    atomicAdd(&a.acc[k], ss[k][0] + ss[k][1] + ss[k][2] + ss[k][3]);
    atomicAdd(&a.cnt[k], sc[k][0] + sc[k][1] + sc[k][2] + sc[k][3]);
  }
}

}  // namespace dsx
