// Store bandwidth of row pieces as the march kernels write them: every wave owns a piece of PIECE floats of each row
// (pieces of neighbouring waves are adjacent: piece s starts at float PIECE * s, 8 waves per block write 8 adjacent
// pieces) and walks down ROWS rows; the piece is written with 4-, 8- or 16-byte stores per lane.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/store_width tools/micro/store_width.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int W>  // floats per lane and store: 1, 2, 4
__global__ __launch_bounds__(512) void k(float* out, int piece, int ld, int rows, int rows_per_seg, long long plane_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x * 8 + wave, seg = blockIdx.y, plane = blockIdx.z;
  float* base = out + plane * plane_stride + (long long)strip * piece;
  const int r0 = seg * rows_per_seg, r1 = min(rows, r0 + rows_per_seg);
  const int c = lane * W;
  float v = (float)(strip + lane);
  for (int r = r0; r < r1; ++r) {
    float* p = base + (long long)r * ld + c;
    v += 1.0f;
    if (c + W <= piece) {
      if (W == 1) p[0] = v;
      if (W == 2) { float2 t = {v, v}; __builtin_nontemporal_store(t.x, p); __builtin_nontemporal_store(t.y, p + 1); }
      if (W == 4) { p[0] = v; p[1] = v; p[2] = v; p[3] = v; }
    } else {
      for (int e = 0; c + e < piece && e < W; ++e) p[e] = v;
    }
  }
}
// (the compiler merges the element stores of a lane into one wide store where the address alignment it can prove allows;
//  to pin the width the variants below use the buffer intrinsics)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7FFFFFFF, 0x00020000); }

template <int W>
__global__ __launch_bounds__(512) void kb(float* out, int piece, int ld, int rows, int rows_per_seg, long long plane_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x * 8 + wave, seg = blockIdx.y, plane = blockIdx.z;
  float* base = out + plane * plane_stride + (long long)strip * piece;
  const __amdgpu_buffer_rsrc_t rs = rsrc(base);
  const int r0 = seg * rows_per_seg, r1 = min(rows, r0 + rows_per_seg);
  const int c = lane * W;
  unsigned v = (unsigned)(strip + lane);
  const bool full = c + W <= piece;
  const int tail = piece - c;  // floats of a partial lane
  for (int r = r0; r < r1; ++r) {
    const unsigned soff = (unsigned)r * (unsigned)ld * 4u;
    v += 1u;
    if (W == 1) {  // 64 floats per instruction: as many instructions as the piece needs
      for (int cc = lane; cc < piece; cc += 64) __builtin_amdgcn_raw_buffer_store_b32(v, rs, cc * 4u, soff, 0);
      continue;
    }
    if (full) {
      if (W == 1) __builtin_amdgcn_raw_buffer_store_b32(v, rs, c * 4u, soff, 0);
      if (W == 2) { u32x2 t = {v, v}; __builtin_amdgcn_raw_buffer_store_b64(t, rs, c * 4u, soff, 0); }
      if (W == 4) { u32x4 t = {v, v, v, v}; __builtin_amdgcn_raw_buffer_store_b128(t, rs, c * 4u, soff, 0); }
    } else if (tail > 0) {
      for (int e = 0; e < tail; ++e) __builtin_amdgcn_raw_buffer_store_b32(v, rs, (c + e) * 4u, soff, 0);
    }
  }
}

template <int W>
void run(const char* name, float* d, int piece, int strips, int rows, int planes) {
  const int ld = (piece * strips + 4 + 3) & ~3;
  const long long plane_stride = (long long)ld * rows + 64;
  const int rows_per_seg = 128, nseg = (rows + rows_per_seg - 1) / rows_per_seg;
  dim3 grid((strips + 7) / 8, nseg, planes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kb<W>, grid, dim3(512), 0, 0, d, piece, ld, rows, rows_per_seg, plane_stride);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(kb<W>, grid, dim3(512), 0, 0, d, piece, ld, rows, rows_per_seg, plane_stride);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 10.0 * planes * (double)rows * piece * strips * 4.0;
  printf("%-44s piece %3d floats (%4d B): %7.3f ms  %7.1f GB/s\n", name, piece, piece * 4, ms / 10, bytes / (ms * 1e6));
}

int main() {
  float* d;
  const int planes = 128;
  (void)hipMalloc(&d, (size_t)planes * (1100ll * 1030 + 64) * 4 + (1 << 20));
  // da_1 of a 2048^2 plane: 1026 rows, 8 strips of 122 floats (+ the last, shorter one: left out)
  run<1>("da1-like, 4-byte stores", d, 122, 8, 1026, planes);
  run<2>("da1-like, 8-byte stores (the kernel's)", d, 122, 8, 1026, planes);
  run<4>("da1-like, 16-byte stores", d, 122, 8, 1026, planes);
  run<2>("da1-like, 8-byte stores, 128-float pieces", d, 128, 8, 1026, planes);
  run<4>("da1-like, 16-byte stores, 128-float pieces", d, 128, 8, 1026, planes);
  // aa_2 / da_2: 515 rows, 8 strips of 61 floats
  run<1>("level-2-like, 4-byte stores (the kernel's)", d, 61, 8, 515, planes);
  run<2>("level-2-like, 8-byte stores", d, 61, 8, 515, planes);
  run<4>("level-2-like, 16-byte stores", d, 61, 8, 515, planes);
  run<1>("level-2-like, 4-byte stores, 64-float pieces", d, 64, 8, 515, planes);
  run<4>("level-2-like, 16-byte stores, 64-float pieces", d, 64, 8, 515, planes);
  return 0;
}
