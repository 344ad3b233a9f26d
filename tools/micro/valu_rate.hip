// Issue rate of FP32 vector instructions on gfx950, plain against packed: each kernel runs ITER x 16 independent
// instructions of one kind per wave; blocks x waves fill the chip at a chosen number of waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/valu_rate tools/micro/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  f2 a[16];
  const float t = seed + threadIdx.x * 1e-6f;
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = f2{t + i, t - i};
  const f2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {  // 32 v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a[i].x), "+v"(a[i].y) : "v"(m.x), "v"(c.x));
      REP16(X)
#undef X
    } else if (KIND == 1) {  // 16 v_pk_fma_f32 (32 fmas)
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      REP16(X)
#undef X
    } else if (KIND == 2) {  // 16 v_pk_mul_f32
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      REP16(X)
#undef X
    } else if (KIND == 3) {  // 16 v_pk_add_f32
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (KIND == 4) {  // 16 v_exp_f32
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i].x));
      REP16(X)
#undef X
    } else if (KIND == 5) {  // 16 v_mov_b64
#define X(i) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
      REP16(X)
#undef X
    } else if (KIND == 6) {  // 16 v_cndmask_b32
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(c.x) : "vcc");
      REP16(X)
#undef X
    } else if (KIND == 8) {  // v_cndmask_b32_e64 with an SGPR-pair condition
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i].x) : "v"(c.x));
      REP16(X)
#undef X
    } else if (KIND == 9) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
      REP16(X)
#undef X
    } else if (KIND == 10) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
      REP16(X)
#undef X
    } else if (KIND == 11) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i].x), "v"(c.x) : "vcc");
      REP16(X)
#undef X
    } else if (KIND == 12) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i].x) : "v"(a[(i + 1) & 15].y));
      REP16(X)
#undef X
    } else if (KIND == 13) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[i].x) : "v"(a[(i + 1) & 15].y));
      REP16(X)
#undef X
    } else if (KIND == 14) {
#define X(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i].x));
      REP16(X)
#undef X
    } else if (KIND == 15) {
#define X(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i].x));
      REP16(X)
#undef X
    } else if (KIND == 16) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i].x));
      REP16(X)
#undef X
    } else if (KIND == 17) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
      REP16(X)
#undef X
    } else if (KIND == 18) {  // v_fma_f32 with an SGPR operand
#define X(i) asm volatile("v_fma_f32 %0, %0, s12, %1" : "+v"(a[i].x) : "v"(c.x));
      REP16(X)
#undef X
    } else if (KIND == 19) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
      REP16(X)
#undef X
    } else if (KIND == 20) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, s[12:13], %1" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (KIND == 21) {
#define X(i) asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(a[i].x) : "v"(c.x));
      REP16(X)
#undef X
    } else if (KIND == 22) {
#define X(i) asm volatile("v_mul_f32 %0, s12, %0" : "+v"(a[i].x));
      REP16(X)
#undef X
    } else if (KIND == 23) {
#define X(i) asm volatile("v_fma_f32 %0, %0, 1.0, %1" : "+v"(a[i].x) : "v"(c.x));
      REP16(X)
#undef X
    } else if (KIND == 24) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
      REP16(X)
#undef X
    } else if (KIND == 25) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(c.x));
      REP16(X)
#undef X
    } else if (KIND == 26) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i].x) : "v"(a[(i + 5) & 15].y), "v"(m.x), "v"(c.x));
      REP16(X)
#undef X
    } else if (KIND == 27) {
#define X(i) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(a[i].x));
      REP16(X)
#undef X
    } else if (KIND == 28) {
#define X(i) asm volatile("v_fmac_f32_e32 %0, 0x3f0f2dee, %1" : "+v"(a[i].x) : "v"(a[(i + 5) & 15].y));
      REP16(X)
#undef X
    } else if (KIND == 29) {
#define X(i) asm volatile("v_fmamk_f32 %0, %1, 0x3f0f2dee, %0" : "+v"(a[i].x) : "v"(a[(i + 5) & 15].y));
      REP16(X)
#undef X
    } else if (KIND == 30) {
#define X(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i].x) : "v"(a[(i + 5) & 15].y), "v"(a[(i + 9) & 15].y));
      REP16(X)
#undef X
    } else if (KIND == 31) {
#define X(i) asm volatile("v_fmac_f32_e32 %0, s12, %1" : "+v"(a[i].x) : "v"(a[(i + 5) & 15].y));
      REP16(X)
#undef X
    } else if (KIND == 32) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, s12, %0" : : "v"(a[i].x) : "vcc");
      REP16(X)
#undef X
    } else if (KIND == 33) {
#define X(i) asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1" : : "v"(a[i].x), "v"(c.x) : "s10", "s11");
      REP16(X)
#undef X
    } else if (KIND == 34) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_bcnt1_i32_b64 s10, vcc\n s_add_i32 s11, s11, s10" : : "v"(a[i].x), "v"(c.x) : "vcc", "s10", "s11", "scc");
      REP16(X)
#undef X
    } else if (KIND == 35) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, s12, %0\n s_bcnt1_i32_b64 s10, vcc\n s_add_i32 s11, s11, s10" : : "v"(a[i].x) : "vcc", "s10", "s11", "scc");
      REP16(X)
#undef X
    } else if (KIND == 7) {  // 16 v_pk_fma_f32 with op_sel_hi broadcast of a scalar pair (the form the compiler emits for taps)
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(m), "v"(c));
      REP16(X)
#undef X
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
  if (s == 12345.678f) out[0] = s;
}

template <int KIND>
double run(const char* name, int instr_per_iter, int waves_per_simd, float* d) {
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 SIMDs x waves_per_simd waves) / 4 waves per block
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)iters * instr_per_iter * waves_per_simd;  // wave-instructions per SIMD
  const double ns_per_instr = ms * 1e6 / per_simd;
  printf("%-34s waves/SIMD %d: %8.3f ms  %6.3f ns per wave-instruction per SIMD (= %.2f clk at 2.4 GHz)\n", name, waves_per_simd, ms,
         ns_per_instr, ns_per_instr * 2.4);
  return ns_per_instr;
}

int main() {
  float* d;
  hipMalloc(&d, 1024);
  for (int w : {4}) {
    run<0>("v_fma_f32", 32, w, d);
    run<1>("v_pk_fma_f32", 16, w, d);
    run<7>("v_pk_fma_f32 op_sel_hi:[1,0,1]", 16, w, d);
    run<2>("v_pk_mul_f32", 16, w, d);
    run<3>("v_pk_add_f32", 16, w, d);
    run<4>("v_exp_f32", 16, w, d);
    run<5>("v_mov_b64", 16, w, d);
    run<32>("v_cmp_lt_f32 vcc, sgpr, v", 16, w, d);
    run<33>("v_cmp_lt_f32_e64 sgpr pair, v, v", 16, w, d);
    run<34>("v_cmp vcc,v,v + s_bcnt1 + s_add", 16, w, d);
    run<35>("v_cmp vcc,s,v + s_bcnt1 + s_add", 16, w, d);
    run<28>("v_fmac_f32_e32 with literal", 16, w, d);
    run<29>("v_fmamk_f32 (literal)", 16, w, d);
    run<30>("v_fmac_f32_e32 3 vgprs", 16, w, d);
    run<31>("v_fmac_f32_e32 with sgpr", 16, w, d);
    run<20>("v_pk_fma_f32 with sgpr pair", 16, w, d);
    run<21>("v_fma_f32 with inline 2.0", 16, w, d);
    run<23>("v_fma_f32 with inline 1.0", 16, w, d);
    run<26>("v_fma_f32 d = a*b+c (4 regs)", 16, w, d);
    run<22>("v_mul_f32 with sgpr", 16, w, d);
    run<27>("v_mul_f32 with inline 2.0", 16, w, d);
    run<24>("v_max_f32", 16, w, d);
    run<25>("v_cndmask_b32_e64 vcc (no clobber)", 16, w, d);
    run<6>("v_cndmask_b32 (vcc)", 16, w, d);
    run<8>("v_cndmask_b32_e64 (sgpr pair)", 16, w, d);
    run<9>("v_mul_f32", 16, w, d);
    run<17>("v_fmac_f32", 16, w, d);
    run<18>("v_fma_f32 with sgpr", 16, w, d);
    run<19>("v_min_f32", 16, w, d);
    run<10>("v_add_u32", 16, w, d);
    run<11>("v_cmp_lt_f32 -> vcc", 16, w, d);
    run<12>("v_mov_b32", 16, w, d);
    run<13>("v_mov_b32_dpp quad_perm", 16, w, d);
    run<14>("v_cvt_f32_u32", 16, w, d);
    run<15>("v_log_f32", 16, w, d);
    run<16>("v_rcp_f32", 16, w, d);
  }
  return 0;
}
