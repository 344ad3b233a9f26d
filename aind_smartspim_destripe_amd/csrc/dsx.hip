// dsx.hip -- C ABI of the MI355X destripe engine (see include/dsx.h) and the launch pipeline.
//
// Pipeline per cohort part (a cohort of <= max_batch planes is split into parts that run on separate
// HIP streams; no host synchronisation inside):
//   k_zero3  ->  k_fwd_march x L  ->  k_hist x L  ->  k_otsu  ->  k_rowfilter x L
//   ->  k_inv_march<pyramid> x (L-1)  ->  k_inv_march<final>
// A wavelet other than db3 (dsx_set_wavelet) swaps k_fwd_gen / k_inv_gen (dsx_wavelet.h) in for the marching kernels.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl.so is dlopen'ed by dsx_comm_init
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/dsx.h"
#include "dsx_kernels.h"
#include "dsx_retile.h"
#include "dsx_wavelet.h"
#include "dsx_plan.h"
#include "dsx_io.h"

namespace {

thread_local std::string g_init_error;

enum KernelClass { KC_FWD1 = 0, KC_FWD, KC_HIST, KC_OTSU, KC_ROW, KC_INV, KC_FINAL, KC_COUNT };
const char* kClassNames[KC_COUNT] = {"k_fwd_march(level1)", "k_fwd_march(coarse)", "k_hist", "k_otsu",
                                     "k_rowfilter",       "k_inv_march(pyramid)", "k_inv_march(final)"};

struct ProfRec {
  int cls;
  hipEvent_t e0, e1;
};

}  // namespace

// Launch-geometry switches of a context, read from the environment ONCE PER CONTEXT by dsx_init (INTEGRATION.md
// section 6).  None of them may change a result: tools/fuzz_parity.py draws them per case (a fresh context each) and
// tests/test_gpu_fuzz.py holds 30 such cases to the parity statement; out-of-range values are clamped here.
struct DsxTuning {
  int row_wpb = 0;            // DSX_ROW_WPB: waves per k_rowfilter block (0 = chosen per transform length)
  int march_waves = 256 * 16; // DSX_MARCH_WAVES: waves a march launch aims at
  bool no_quant = false;      // DSX_NO_QUANT: segment counts without the launch-quantisation score
  int seg_min_rows = 4;       // DSX_SEG_MIN_ROWS: least rows per march segment of the plain level kernels
  bool no_fuse = false;       // DSX_NO_FUSE: one forward launch per level (aa_1 materialised)
  bool no_fuse_inv = false;   // DSX_NO_FUSE_INV: one inverse launch per level
  bool no_pair = false;       // DSX_NO_PAIR: 8-byte pixel / result accesses in the final kernels
  int hist_rows = 0;          // DSX_HIST_ROWS: rows per k_hist block (0 = chosen per cohort size)
  int fwd_wpb = 8, inv_wpb = 8;  // DSX_FWD_WPB / DSX_INV_WPB: waves per block of the fused march kernels (4 or 8)
  bool no_row_multi = false;  // DSX_NO_ROW_MULTI: one row-filter launch per coarse level
  int row_multi_alone = 48;   // DSX_ROW_MULTI_ALONE: largest unsplit cohort that takes the merged coarse row filter
  int helper = -1;            // DSX_HELPER: -1 = helper stream for unsplit cohorts only, 0 / 1 = never / always
  bool no_pipeline = false;   // DSX_NO_PIPELINE: join the sub-cohort streams at every call
#ifdef DSX_DIAG
  int skip_hist = 0, skip_row = 0, skip_from = 1000;  // timing-only: WRONG results (DSX_SKIP_HIST / _ROW / _COARSE)
#endif
};

struct dsx_ctx {
  int device = 0;
  DsxTuning tune;
  hipStream_t stream_ = nullptr;  // the context ("compute") stream; entry points reach it through use_main()
  std::string err;
  bool planned = false;
  dsx::Plan plan;
  dsx::HostCfg cfg[2];
  double high_int = 2700.0;
  int max_batch = 0;
  // sigmoid(float16((x - 400) / 20)) > 0.3 of filtering.py:78-80  <=>  float16(x) >= 383.25  <=>
  // x > 383.125 (round-to-nearest-even), checked against every float16 pattern in tests/golden
  float fg_cutoff = 383.12503f;
  // device buffers
  float* d_ws = nullptr;          // [max_batch][plane_floats]
  char* d_ctl = nullptr;          // control block, zeroed per cohort (stats | minmax | hist)
  size_t ctl_zero_bytes = 0;
  dsx::PlaneStats* d_stats = nullptr;
  unsigned* d_minmax = nullptr;
  unsigned* d_hist = nullptr;
  float* d_thr = nullptr;
  float* d_otsu = nullptr;
  int* d_cfg = nullptr;
  double* d_means = nullptr;
  dsx::C32* d_consts = nullptr;
  size_t consts_bytes = 0;
  float* d_flat = nullptr;
  float* d_dark = nullptr;
  bool own_shading = false;
  int dark_h = 0, dark_w = 0;
  // staging for dsx_run_host
  void* d_stage_in = nullptr;
  void* d_stage_out = nullptr;
  size_t stage_in_bytes = 0, stage_out_bytes = 0;
  // last cohort (debug hooks)
  int last_n = 0;
  int stop_after = 0;
  // timing
  hipEvent_t t0 = nullptr, t1 = nullptr;
  bool profiling = false;
  std::vector<ProfRec> prof;
  size_t workspace_bytes = 0;
  // filter bank given by dsx_set_wavelet (wl_len == 0: db3, the specialised marching kernels)
  int wl_len = 0;          // bank in use by the current plan
  int wl_bank_len = 0;     // bank given by the last dsx_set_wavelet
  bool wl_set = false;
  float wl[4][dsx::kMaxTaps] = {};       // dec_lo, dec_hi, rec_lo, rec_hi of the current plan
  float wl_bank[4][dsx::kMaxTaps] = {};  // ... of the last dsx_set_wavelet
  // Sticky flag word in pinned host memory, written by k_otsu when a plane's PlaneStats::flags is set (a float32
  // pixel whose log(1 + x) is not finite).  The asynchronous device-buffer path (dsx_run_device) cannot raise at
  // the call; the next synchronising entry point (dsx_sync, dsx_event_sync, dsx_stream_sync, dsx_get_stats) reports
  // DSX_EVALUE instead and clears the word.
  unsigned* h_sticky = nullptr;
  unsigned* d_sticky = nullptr;
  // dsx_set_stack_mode: the planes of a call share ONE Otsu threshold per level (the reference's 3-D input mode)
  bool stack_mode = false;
  bool no_fuse_rf = false;  // DSX_NO_FUSE_RF=1: k_rowfilter + k_inv_march instead of k_rowfinal (same bits)
  bool fuse_rf_wide = false;  // DSX_FUSE_RF_WIDE=1: k_rowfinal for 2000- / 1800-wide planes too (slower there; rowfinal_plan)
  // DSX_ABLATE environment variable (timing-only switches that return WRONG pixels): read only by a library built
  // with -DDSX_DIAG (tools/build_variant.sh); the product build has no such switch
  int ablate = 0;
  // sub-cohort streams: a cohort is split into parts that run their launch chains concurrently, so that
  // latency-bound (march) and compute-bound (row filter) kernels of different parts overlap on the chip
  static constexpr int kMaxStreams = 8;
  int n_streams = 4;
  hipStream_t aux[kMaxStreams] = {};
  hipEvent_t ev_fork = nullptr, ev_join[kMaxStreams] = {};
  // Cross-call pipelining of the sub-cohort streams: the joins of a split cohort are deferred until something
  // else needs the context stream, so that an identical run_device call that follows (the next batch of a
  // steady stream of batches) starts each part right behind the same part of the call before it.
  // Helper stream per part: the wide levels' histogram / row filter run beside the coarse levels' launches
  // (small grids that leave most of the chip idle), see run_cohort
  hipStream_t helper[kMaxStreams] = {};
  hipEvent_t ev_h[kMaxStreams][4] = {};
  int joins_pending = 0;              // parts (incl. part 0) of a split cohort the context stream has not joined yet
  unsigned long long main_ops = 0;    // operations entry points put on the context stream (use_main calls)
  struct {
    const void* in = nullptr; void* out = nullptr; const void* cfg = nullptr;
    int n = 0, in_dtype = 0, out_dtype = 0, parts = 0;
    unsigned long long main_ops = 0;
  } last_split;
  // multi-GPU: RCCL communicator of the one collective of the path (dsx_comm_*)
  void* rccl = nullptr;  // dlopen handle of librccl.so
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  double* d_red = nullptr;  // scratch of dsx_comm_allreduce_f64
  // copy streams of the overlapped chunk map (dsx_memcpy_*_async): 1 = upload, 2 = download
  hipStream_t copy_stream[2] = {};
  hipEvent_t ev_xs = nullptr;  // cross-stream ordering (dsx_stream_wait)
  static constexpr int kEventSlots = 8;
  hipEvent_t ev_slot[kEventSlots] = {};  // host-visible completion marks (dsx_event_record / dsx_event_sync)
  // HIP graphs of unsplit cohorts (run_cohort_split), OPT-IN with DSX_GRAPH=1: the launch chain of a (planes, result,
  // count, types) tuple is captured the second time the tuple is seen and replayed from then on -- one graph launch
  // instead of ~30 kernel launches and 4 cross-stream events for the per-slice calls of the reference's API
  // (filter_stripes: 1 plane).  Measured on ROCm 7.2 / MI355X (tools/latency_small.py, profiles/r2_latency_small.txt):
  // bit-identical results, but 5-8 % SLOWER than the eager launches (1 plane of 2048^2: 515 against 473 us per call,
  // the enqueue takes 130 us either way), so eager stays the default.
  struct GraphEntry {
    const void* in; void* out; const void* cfg;
    int n, in_dtype, out_dtype, seen;
    hipGraphExec_t exec;
    unsigned long long last_use;
  };
  static constexpr int kGraphSlots = 8;
  std::vector<GraphEntry> graphs;
  unsigned long long graph_clock = 0;
  int graph_mode = 0;  // DSX_GRAPH=1 turns graphs on; back to 0 when a capture fails on this runtime
  unsigned long long graph_launches = 0, graph_captures = 0;
};

namespace {
// Buffers of one part of a cohort (planes [po, po + nb) of the workspace) and the stream it runs on.
struct CohortView {
  hipStream_t stream;
  hipStream_t helper;   // nullptr: everything on `stream`
  hipEvent_t* ev;       // 4 events of this part (helper fork / join, twice)
  bool alone;           // the cohort runs as ONE part: its big kernels have the chip to themselves (march_segments)
  float* ws;
  dsx::PlaneStats* stats;
  unsigned* minmax;
  unsigned* hist;
  float* thr;
  float* otsu;
  int* cfg;
  double* means;
};
}  // namespace

namespace {

// the tuning of the context whose launch chain this thread is enqueueing (set by run_cohort_split / the stand-alone
// entry points; the launch helpers below have no context argument)
thread_local const DsxTuning* t_tune = nullptr;
const DsxTuning& tune() {
  static const DsxTuning defaults;
  return t_tune ? *t_tune : defaults;
}

int env_int(const char* name, int fallback) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : fallback;
}

void read_tuning(DsxTuning& t) {
  t.row_wpb = env_int("DSX_ROW_WPB", 0);
  if (t.row_wpb != 0) t.row_wpb = std::max(4, std::min(t.row_wpb, (int)dsx::kRowMaxWaves));
  t.march_waves = std::max(64, env_int("DSX_MARCH_WAVES", 256 * 16));
  t.no_quant = env_int("DSX_NO_QUANT", 0) != 0;
  t.seg_min_rows = std::max(2, env_int("DSX_SEG_MIN_ROWS", 4));
  t.no_fuse = env_int("DSX_NO_FUSE", 0) != 0;
  t.no_fuse_inv = env_int("DSX_NO_FUSE_INV", 0) != 0;
  t.no_pair = env_int("DSX_NO_PAIR", 0) != 0;
  t.hist_rows = std::max(0, env_int("DSX_HIST_ROWS", 0));
  t.fwd_wpb = env_int("DSX_FWD_WPB", 8) == 4 ? 4 : 8;
  t.inv_wpb = env_int("DSX_INV_WPB", 8) == 4 ? 4 : 8;
  t.no_row_multi = env_int("DSX_NO_ROW_MULTI", 0) != 0;
  t.row_multi_alone = std::max(0, env_int("DSX_ROW_MULTI_ALONE", 48));
  t.helper = getenv("DSX_HELPER") ? (env_int("DSX_HELPER", 0) != 0 ? 1 : 0) : -1;
  t.no_pipeline = env_int("DSX_NO_PIPELINE", 0) != 0;
#ifdef DSX_DIAG
  t.skip_hist = env_int("DSX_SKIP_HIST", 0);
  t.skip_row = env_int("DSX_SKIP_ROW", 0);
  t.skip_from = env_int("DSX_SKIP_COARSE", 1000);
#endif
}

int fail(dsx_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  else g_init_error = msg;  // context-free calls (dsx_io_*): dsx_last_error(NULL) reports it
  return code;
}

// The context stream for an entry point: first waits for the sub-cohort streams of a split cohort that is
// still running (deferred joins), and notes that the stream has been used.
hipStream_t use_main(dsx_ctx* c) {
  for (int i = 1; i < c->joins_pending; ++i) (void)hipStreamWaitEvent(c->stream_, c->ev_join[i], 0);
  c->joins_pending = 0;
  ++c->main_ops;
  return c->stream_;
}

#define DSX_HIP(call)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(ctx, DSX_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));     \
  } while (0)

void drop_graphs(dsx_ctx* c) {
  for (auto& g : c->graphs)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  c->graphs.clear();
}

void free_plan_buffers(dsx_ctx* c) {
  drop_graphs(c);  // they hold the addresses of the buffers freed below
  auto fr = [](void* p) { if (p) (void)hipFree(p); };
  fr(c->d_ws); c->d_ws = nullptr;
  fr(c->d_ctl); c->d_ctl = nullptr;
  fr(c->d_thr); c->d_thr = nullptr;
  fr(c->d_otsu); c->d_otsu = nullptr;
  fr(c->d_cfg); c->d_cfg = nullptr;
  fr(c->d_means); c->d_means = nullptr;
  fr(c->d_consts); c->d_consts = nullptr;
  if (c->own_shading) { fr(c->d_flat); fr(c->d_dark); }
  c->d_flat = c->d_dark = nullptr;
  c->own_shading = false;
  fr(c->d_stage_in); c->d_stage_in = nullptr; c->stage_in_bytes = 0;
  fr(c->d_stage_out); c->d_stage_out = nullptr; c->stage_out_bytes = 0;
  c->planned = false;
}

struct LaunchScope {
  dsx_ctx* c;
  bool on;
  ProfRec r;
  LaunchScope(dsx_ctx* c_, int cls) : c(c_), on(c_->profiling) {
    if (on) {
      r.cls = cls;
      (void)hipEventCreate(&r.e0);
      (void)hipEventCreate(&r.e1);
      (void)hipEventRecord(r.e0, c->stream_);
    }
  }
  ~LaunchScope() {
    if (on) {
      (void)hipEventRecord(r.e1, c->stream_);
      c->prof.push_back(r);
    }
  }
};

// Waves (row pairs) per block: they share one twiddle table in LDS; pick the count that puts the
// most waves on a CU (160 KiB LDS, at most 16 waves at 4 waves per SIMD).
int rowfilter_waves_per_block(int M) {
  int best_w = 4, best_total = 0;
  for (int w = 4; w <= dsx::kRowMaxWaves; ++w) {
    const size_t bytes = (size_t)M * (w + 1) * sizeof(float2);
    const int blocks = (int)std::min<size_t>(160 * 1024 / bytes, 8);
    const int total = std::min(blocks * w, 16);
    if (total > best_total) { best_total = total; best_w = w; }
  }
  return best_w;
}

template <int CPL, int GF = -1, int NT = -1, int HALO = -1, int PLAN = 0>
hipError_t launch_rowfilter(const dsx::RowArgs& a_in, int npairs, int nb, hipStream_t s) {
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)dsx::k_rowfilter<CPL, GF, NT, HALO, PLAN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev & 63] = true;
  }
  int force_wpb = tune().row_wpb;
  if (force_wpb && (size_t)a_in.M * (force_wpb + 1) * sizeof(float2) > 160 * 1024) force_wpb = 0;  // would not fit the LDS
  // M >= 1024 (level 2 of the hot shapes, 77 KB per 8-wave block): 4-wave blocks of 43 KB.  Alone the kernel does not
  // care; beside the 74 KB blocks of k_rowfinal and the small coarse-level blocks three of them still find room on a CU
  // where two of the big ones do not (+0.6 % 4-stream, +1.3 % single-stream, gpurun_out/wpb_sweep.txt)
  const int wpb = force_wpb ? force_wpb : ((CPL > 18 || a_in.M >= 1024) ? 4 : rowfilter_waves_per_block(a_in.M));
  const size_t smem = (size_t)a_in.M * (wpb + 1) * sizeof(float2);
  const dsx::RowArgs& a = a_in;
  const dim3 grid((npairs + wpb - 1) / wpb, nb);
  hipLaunchKernelGGL((dsx::k_rowfilter<CPL, GF, NT, HALO, PLAN>), grid, dim3(64 * wpb), smem, s, a);
  return hipGetLastError();
}

// rows longer than one wave holds: one 512-lane block per row pair, the row buffer is the block's LDS
hipError_t launch_rowfilter_wide(const dsx::RowArgs& a, int npairs, int nb, hipStream_t s) {
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)dsx::k_rowfilter_wide, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(dsx::kWideMaxLen * sizeof(float2)));
    if (e != hipSuccess) return e;
    attr_set[dev & 63] = true;
  }
  if (a.M > dsx::kWideMaxLen || a.w >= 65536) return hipErrorInvalidValue;  // (the plan refuses these lengths)
  hipLaunchKernelGGL(dsx::k_rowfilter_wide, dim3(npairs, nb), dim3(dsx::kWideThreads), (size_t)a.M * sizeof(float2), s, a);
  return hipGetLastError();
}

hipError_t dispatch_rowfilter(const dsx::RowArgs& a, int npairs, int nb, hipStream_t s) {
  const int cpl = (a.M + 63) / 64;
  if (cpl > 36) return launch_rowfilter_wide(a, npairs, nb, s);
  // slot structure of the row (full 256-value groups, 64-value tail slots) and pass list: the wide levels of
  // 2048-, 2000- and 1800-wide planes have instantiations with all of it as compile-time constants
  const int gf = a.w >> 8, nt = (a.w - (gf << 8) + 63) >> 6;
  auto plan_is = [&](int m, int r0, int r1, int r2) {
    return a.M == m && a.npass == 3 && a.radix[0] == r0 && a.radix[1] == r1 && a.radix[2] == r2;
  };
  if (cpl <= 2) return launch_rowfilter<2>(a, npairs, nb, s);
  if (cpl <= 4) return launch_rowfilter<4>(a, npairs, nb, s);
  if (cpl <= 6) return launch_rowfilter<6>(a, npairs, nb, s);
  if (cpl <= 10) return launch_rowfilter<10>(a, npairs, nb, s);
  if (cpl <= 18) {
    if (gf == 4 && nt == 1 && a.K == 0) {
      if (plan_is(1026, 19, 9, 6))
        return launch_rowfilter<18, 4, 1, 0, 1>(a, npairs, nb, s);
      return launch_rowfilter<18, 4, 1, 0>(a, npairs, nb, s);
    }
    if (gf == 2 && nt == 1 && a.K > 0) {
      if (plan_is(1071, 17, 9, 7))
        return launch_rowfilter<18, 2, 1, 1, 2>(a, npairs, nb, s);
      return launch_rowfilter<18, 2, 1, 1>(a, npairs, nb, s);
    }
    if (gf == 1 && nt == 4 && a.K > 0) {
      if (plan_is(1024, 16, 8, 8)) return launch_rowfilter<18, 1, 4, 1, 5>(a, npairs, nb, s);
      if (plan_is(960, 15, 8, 8)) return launch_rowfilter<18, 1, 4, 1, 6>(a, npairs, nb, s);
    }
    return launch_rowfilter<18>(a, npairs, nb, s);
  }
  if (a.K > 0 && gf == 3) {
    if (nt == 4 && plan_is(2048, 16, 16, 8)) return launch_rowfilter<36, 3, 4, 1, 3>(a, npairs, nb, s);
    if (nt == 3 && plan_is(1815, 15, 11, 11)) return launch_rowfilter<36, 3, 3, 1, 4>(a, npairs, nb, s);
  }
  return launch_rowfilter<36>(a, npairs, nb, s);
}

// Level-1 row filter + final synthesis in one kernel (k_rowfinal): Delta_1 stays in LDS.  Shapes that take it: see the
// kernel's header.  DSX_NO_FUSE_RF=1 (read by dsx_init) keeps the two kernels: A/B runs and the bit-identity test.
// Returns the StaticFft plan id of the instantiation (1, 3, 4) or 0.
int rowfinal_plan(const dsx::Plan& p, const dsx::LevelPlan& lp, int in_dtype, int out_dtype, bool pair_io, bool wide) {
  if (!pair_io || in_dtype != DSX_U16 || out_dtype != DSX_U16) return 0;
  if ((lp.w & 1) != 0 || (p.Wout + dsx::kMarchCols - 1) / dsx::kMarchCols > dsx::kRfWaves) return 0;
  const int gf = lp.w >> 8, nt = (lp.w - (gf << 8) + 63) >> 6;
  auto plan_is = [&](int m, int r0, int r1, int r2) {
    return lp.M == m && lp.npass == 3 && lp.radix[0] == r0 && lp.radix[1] == r1 && lp.radix[2] == r2;
  };
  // wide (DSX_FUSE_RF_WIDE=1, read by dsx_init; OFF by default): the embedded plans of 2000- and 1800-wide planes.  Their
  // FFT buffers are 16 KB per wave: one 8-wave block per CU, the two phases of a block cannot hide behind another
  // block's, and the fused kernel LOSES 10-12 % against k_rowfilter + k_inv_march there (1600 x 2000: 53.8 k against
  // 59.8 k planes/s, 1800^2: 54.8 k against 62.8 k; profiles/r3_fused_rowfinal_ab.txt).  Bit-identical all the same
  // (tests/test_gpu_parity.py::test_fused_rowfinal_is_bit_identical_to_the_unfused_chain runs them with the switch on).
  if (gf == 4 && nt == 1 && lp.K == 0 && plan_is(1026, 19, 9, 6)) return 1;
  if (wide && lp.K > 0 && gf == 3 && nt == 4 && plan_is(2048, 16, 16, 8)) return 3;
  if (wide && lp.K > 0 && gf == 3 && nt == 3 && plan_is(1815, 15, 11, 11)) return 4;
  return 0;
}

template <int CPL, int GF, int NT, int HALO, int PLAN>
hipError_t launch_rowfinal_t(const dsx::RowFinalArgs& a, int nb, hipStream_t s) {
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto kern = dsx::k_rowfinal<CPL, GF, NT, HALO, PLAN>;
  if (!attr_set[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev & 63] = true;
  }
  const size_t smem = (size_t)a.r.M * (dsx::kRfWaves + 1) * sizeof(float2);
  const int np = (a.f.hout + 1) / 2;
  int nblk = (np + dsx::kRfRows - 1) / dsx::kRfRows;
#if DSX_RF_XCD
  nblk = (nblk + 7) / 8 * 8;  // the kernel deals runs of consecutive blocks to the 8 compute dies
#endif
  const dim3 grid(nblk, nb);
  hipLaunchKernelGGL(kern, grid, dim3(64 * dsx::kRfWaves), smem, s, a);
  return hipGetLastError();
}

hipError_t launch_rowfinal(int plan, const dsx::RowFinalArgs& a, int nb, hipStream_t s) {
  if (plan == 1) return launch_rowfinal_t<18, 4, 1, 0, 1>(a, nb, s);
  if (plan == 3) return launch_rowfinal_t<36, 3, 4, 1, 3>(a, nb, s);
  if (plan == 4) return launch_rowfinal_t<36, 3, 3, 1, 4>(a, nb, s);
  return hipErrorInvalidValue;
}

// Levels whose row filters all fit the CPL = 6 class, in ONE launch (k_rowfilter_multi).
hipError_t launch_rowfilter_multi(const dsx::RowMultiArgs& a_in, const int* npairs, int nb, hipStream_t s) {
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)dsx::k_rowfilter_multi<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev & 63] = true;
  }
  const int wpb = dsx::kRowMaxWaves;
  dsx::RowMultiArgs a = a_in;
  int blocks = 0, max_m = 1;
  for (int i = 0; i < a.nlev; ++i) {
    blocks += (npairs[i] + wpb - 1) / wpb;
    a.blk_end[i] = blocks;
    max_m = std::max(max_m, a.lv[i].M);
  }
  const size_t smem = (size_t)max_m * (wpb + 1) * sizeof(float2);
  hipLaunchKernelGGL(dsx::k_rowfilter_multi<6>, dim3(blocks, nb), dim3(64 * wpb), smem, s, a);
  return hipGetLastError();
}

// Row segmentation of the marching kernels: enough waves to fill the chip for small cohorts,
// one segment per strip for large ones (each extra segment re-reads a 4-row halo).
// wpb > 0 (the fused level-1 kernels, wpb waves per block, 16 waves per CU): the segment count is also chosen against
// the QUANTISATION of the launch -- blocks run in rounds of (CUs x 16 / wpb); 576 equal blocks on 512 slots take two
// rounds, 512 take one.  Among the counts around the target the one with the smallest rounds x (rows per segment +
// halo_rows) wins (halo_rows: the rows a segment recomputes, in the units of `rows`).
void march_segments(int nb, int nstrips, int rows, int* nseg, int* rows_per_seg, int wpb = 0, int halo_rows = 0,
                    bool coarse = false) {
  const int target_waves = tune().march_waves;
  const bool no_quant = tune().no_quant;
  int want = (target_waves + nb * nstrips - 1) / (nb * nstrips);
  // Rows per segment, at least: 24 until late in round 3.  The coarse levels (260 rows and fewer) then ran as a few
  // hundred waves marching 24+ rows each -- launches that are short on parallelism, not on work: levels 3 ... 8 hold 6 % of
  // the coefficients and cost 19 % of the run (timing-only DSX_SKIP_COARSE, profiles/r3_coarse_levels.txt).  4 rows per
  // segment (2 more of halo) gives them 3-6 x the waves and a march a sixth as long: +0.8 % on the whole run.  Only the
  // plain level kernels take it (coarse = true); the fused level-1 + 2 kernels keep 24: at the cohort sizes that matter
  // their segment count is set by the wave target, not by this floor.  (The first attempt gave them 4 rows too and failed
  // the GPU suite -- through last segments of one or two level-2 rows, the bug fwd_march_body's i_first now fixes.)
  const int seg_min_coarse = tune().seg_min_rows;
  const int seg_min_rows = coarse ? seg_min_coarse : 24;
  const int max_seg = std::max(1, rows / seg_min_rows);
  want = std::max(1, std::min(want, max_seg));
  if (wpb > 0 && !no_quant) {
    static int cus = 0;
    if (cus == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
      if (cus <= 0) cus = 256;
    }
    const long long slots = (long long)cus * (16 / wpb);
    double best = 1e300;
    int best_n = want;
    for (int n = std::max(1, want / 2); n <= std::min(max_seg, 2 * want + 1); ++n) {
      const int rps = (rows + n - 1) / n;
      const int ns = (rows + rps - 1) / rps;
      const long long blocks = (long long)nb * ((nstrips * ns + wpb - 1) / wpb);
      const long long rounds = (blocks + slots - 1) / slots;
      // a launch that leaves most of the chip empty is no bargain either: count at least half a round
      const double fill = std::max(0.5, (double)blocks / (double)slots);
      const double cost = std::max((double)rounds, fill) * (double)(rps + halo_rows);
      if (cost < best * 0.999) { best = cost; best_n = n; }
    }
    want = best_n;
  }
  int rps = (rows + want - 1) / want;
  rps = std::max(1, std::min(rps, 4096));  // uint32 partial sums of k_fwd1_march
  *rows_per_seg = rps;
  *nseg = (rows + rps - 1) / rps;
}

// One cohort of nb planes through the whole chain (asynchronous).
int run_cohort(dsx_ctx* ctx, const CohortView& v, const void* d_in, int in_dtype, int nb, void* d_out,
               int out_dtype, int32_t* d_cfg_used) {
  const dsx::Plan& p = ctx->plan;
  hipStream_t s = v.stream;
  const int L = p.L;
  const int Lc = L > 0 ? L : 1;
  t_tune = &ctx->tune;  // the launch helpers of this thread follow this context's switches
  {
    // stats: 32 B per plane; minmax: 8 Lc B per plane (parts start at even planes -> 16-byte aligned
    // except for odd Lc * po, handled by rounding the count up inside the part's own slice);
    // hist: 1 KiB per plane and level
    const int n0 = (int)(sizeof(dsx::PlaneStats) * nb / 16);
    const int n2 = (int)(sizeof(unsigned) * 256 * Lc * (size_t)nb / 16);
    const size_t mm_bytes = sizeof(unsigned) * 2 * Lc * (size_t)nb;
    if ((mm_bytes % 16) == 0 && (((uintptr_t)v.minmax) % 16) == 0) {
      const int n1 = (int)(mm_bytes / 16);
      const int total = n0 + n1 + n2;
      hipLaunchKernelGGL(dsx::k_zero3, dim3(std::min((total + 255) / 256, 1024)), dim3(256), 0, s, (uint4*)v.stats, n0,
                         (uint4*)v.minmax, n1, (uint4*)v.hist, n2);
      DSX_HIP(hipGetLastError());
    } else {
      DSX_HIP(hipMemsetAsync(v.stats, 0, sizeof(dsx::PlaneStats) * nb, s));
      DSX_HIP(hipMemsetAsync(v.minmax, 0, mm_bytes, s));
      DSX_HIP(hipMemsetAsync(v.hist, 0, sizeof(unsigned) * 256 * Lc * nb, s));
    }
  }

  // ---- forward transform ------------------------------------------------------------------
  // levels 1 + 2 in one kernel when the plane allows it (aa_1 never leaves the chip)
  const bool no_fuse = ctx->tune.no_fuse;
  // any wavelet but db3: tap-count-generic level kernels, one launch per level, nothing fused
  const bool generic = ctx->wl_len > 0 && L > 0;
  const bool fuse12 = !generic && !no_fuse && L >= 2 && (p.W % 4) == 0 && (p.lv[0].ldin % 4) == 0 && p.lv[0].h >= 16 &&
                      p.lv[0].w >= 16;
  const bool no_fuse_inv = ctx->tune.no_fuse_inv;
  const bool fuse21 = fuse12 && !no_fuse_inv;
  // uint16 planes whose rows and bases allow 16-byte pixel / result accesses by lane pairs (inv_march_body<.., PAIR>)
  const bool no_pair = ctx->tune.no_pair;
  const bool pair_io = !no_pair && in_dtype == DSX_U16 && (p.W % 8) == 0 && (p.H % 2) == 0 && p.Wout == p.W &&
                       (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0 && ((size_t)p.H * p.W * 2) % 16 == 0;
  // level-1 row filter inside the final kernel (k_rowfinal): not for the staged debug runs, which read Delta_1 back
  const int rf_plan = (fuse21 && !ctx->no_fuse_rf && ctx->stop_after == 0) ? rowfinal_plan(p, p.lv[0], in_dtype, out_dtype, pair_io, ctx->fuse_rf_wide) : 0;
  const bool fuse_rf = rf_plan != 0;
  // The wide levels (1, 2) hold 94 % of the coefficients; the coarse levels are chains of small launches that
  // leave most of the chip idle.  With the fused forward kernel the wide levels' data is complete early, so
  // their histograms (and, below, their row filters) run on the part's helper stream BESIDE the coarse
  // levels' launches instead of after them.
  const bool split = v.helper != nullptr && fuse12 && L >= 3;
  // DSX_SKIP_HIST / DSX_SKIP_ROW: bit masks of level indices whose histogram / row-filter launch is left out, and
  // DSX_SKIP_COARSE=k: no launch for levels with index >= k (what the chain would gain if those kernels were free).
  // The results are WRONG: the switches only exist in a -DDSX_DIAG build.
#ifdef DSX_DIAG
  const int skip_hist = ctx->tune.skip_hist, skip_row = ctx->tune.skip_row, skip_from = ctx->tune.skip_from;
#else
  constexpr int skip_hist = 0, skip_row = 0, skip_from = 1000;
#endif
  // histograms of levels [l0, l1) in ONE launch (k_hist finds its level from the block index)
  auto hist_levels = [&](int l0, int l1, hipStream_t hs) -> int {
    dsx::HistArgs a;
    memset(&a, 0, sizeof(a));
    a.ws = v.ws;
    a.ws_plane_stride = p.plane_floats;
    a.minmax = v.minmax;
    a.hist = v.hist;
    a.L = L;
    a.shared = ctx->stack_mode ? 1 : 0;
    int blocks = 0;
    for (int l = l0; l < l1 && l < skip_from; ++l) {
      if ((skip_hist >> l) & 1) continue;
      const dsx::LevelPlan& lp = p.lv[l];
      const int i = a.nlev++;
      a.lvl[i] = l;
      a.da_off[i] = lp.da_off;
      a.h[i] = lp.h; a.w[i] = lp.w; a.ld[i] = lp.ld;
      // rows per block: each block zeroes and folds 32 KB of counters, so big cohorts take tall blocks (128 rows: + 1-2 %
      // in the 4-stream run against 32), small ones keep enough blocks to spread over the chip (~512 per launch)
      const int hist_rows = ctx->tune.hist_rows;
      const int auto_rows = std::max(32, std::min(128, (int)((long long)lp.h * nb / 512)));
      // k_hist counts in 16-bit per-lane counters shared by the waves of a block: a lane sees 4 values per 16-byte load
      // and (w >> 8) + 1 loads per row at most, and in the worst case they all fall into ONE bin (bin 0 of a plane with
      // one outlier) -- the rows of a block are capped so that even then no counter carries into its neighbour
      // (w = 18 432, the widest level k_rowfilter_wide admits: 227 rows)
      const int rows_cap = std::min(256, 65535 / (4 * (lp.w >> 8) + 4));
      a.rows_per_block[i] = std::max(1, std::min(hist_rows > 0 ? hist_rows : auto_rows, rows_cap));
      blocks += (lp.h + a.rows_per_block[i] - 1) / a.rows_per_block[i];
      a.blk_end[i] = blocks;
    }
    if (a.nlev == 0) return DSX_OK;
    LaunchScope ls(ctx, KC_HIST);
    hipLaunchKernelGGL(dsx::k_hist, dim3(blocks, nb), dim3(64 * dsx::kHistWaves), 0, hs, a);
    DSX_HIP(hipGetLastError());
    return DSX_OK;
  };
  for (int l = 0; l < L; ++l) {
    if (fuse12 && l == 1) continue;
    if (l >= skip_from) continue;
    const dsx::LevelPlan& lp = p.lv[l];
    if (generic) {
      dsx::GenFwdArgs g;
      memset(&g, 0, sizeof(g));
      g.in = d_in;
      g.in_plane_stride = (long long)p.H * p.W;
      g.ws = v.ws;
      g.ws_plane_stride = p.plane_floats;
      g.in_off = (l > 0) ? p.lv[l - 1].aa_off : 0;
      g.H = lp.hin; g.W = lp.win; g.ldin = lp.ldin;
      g.aa_off = lp.aa_off; g.da_off = lp.da_off;
      g.h = lp.h; g.w = lp.w; g.ld = lp.ld; g.lda = lp.lda;
      g.minmax = v.minmax;
      g.lvl = l; g.L = L;
      g.stats = v.stats;
      g.fg_cutoff = ctx->fg_cutoff;
      g.F = ctx->wl_len;
      g.shared = ctx->stack_mode ? 1 : 0;
      memcpy(g.lo, ctx->wl[0], sizeof(float) * ctx->wl_len);
      memcpy(g.hi, ctx->wl[1], sizeof(float) * ctx->wl_len);
      const dim3 gg((lp.w + dsx::kGenTW - 1) / dsx::kGenTW, (lp.h + dsx::kGenTH - 1) / dsx::kGenTH, nb);
      const size_t smem = dsx::gen_fwd_lds_bytes(g.F);
      LaunchScope ls(ctx, l == 0 ? KC_FWD1 : KC_FWD);
      if (l > 0) hipLaunchKernelGGL(dsx::k_fwd_gen<2>, gg, dim3(256), smem, s, g);
      else if (in_dtype == DSX_U16) hipLaunchKernelGGL(dsx::k_fwd_gen<0>, gg, dim3(256), smem, s, g);
      else hipLaunchKernelGGL(dsx::k_fwd_gen<1>, gg, dim3(256), smem, s, g);
      DSX_HIP(hipGetLastError());
      continue;
    }
    dsx::Fwd1Args f;
    memset(&f, 0, sizeof(f));
    f.in = d_in;
    f.in_plane_stride = (long long)p.H * p.W;
    f.ws = v.ws;
    f.ws_plane_stride = p.plane_floats;
    f.in_off = (l > 0) ? p.lv[l - 1].aa_off : 0;
    f.H = lp.hin; f.W = lp.win; f.ldin = lp.ldin;
    f.aa_off = lp.aa_off; f.da_off = lp.da_off;
    f.h = lp.h; f.w = lp.w; f.ld = lp.ld; f.lda = lp.lda;
    f.minmax = v.minmax;
    f.lvl = l; f.L = L;
    f.stats = v.stats;
    f.fg_cutoff = ctx->fg_cutoff;
    f.ablate = ctx->ablate;
    f.shared = ctx->stack_mode ? 1 : 0;
    f.fg_cutoff_u16 = (unsigned)std::min(65536.0, std::max(0.0, ceil((double)ctx->fg_cutoff)));
    f.nstrips = (lp.w + dsx::kMarchOut - 1) / dsx::kMarchOut;
    march_segments(nb, f.nstrips, lp.h, &f.nseg, &f.rows_per_seg, 0, 0, /*coarse=*/l >= 2);
    if (fuse12 && l == 0) {
      const dsx::LevelPlan& l2 = p.lv[1];
      f.aa2_off = l2.aa_off; f.da2_off = l2.da_off;
      f.h2 = l2.h; f.w2 = l2.w; f.ld2 = l2.ld; f.lda2 = l2.lda;
      f.nstrips = (l2.w + dsx::kFuseOut - 1) / dsx::kFuseOut;
      // (a cohort split over the streams shares the chip with the other parts' kernels: rounds mean nothing there --
      //  measured 65.6 k with against 66.2 k planes/s without; alone: forward kernel 1.39 -> 1.14 ms per 256 planes)
      march_segments(nb, f.nstrips, l2.h, &f.nseg, &f.rows_per_seg, v.alone ? 8 : 0, 2);
    }
    dim3 grid((f.nstrips * f.nseg + 3) / 4, nb);
    LaunchScope ls(ctx, l == 0 ? KC_FWD1 : KC_FWD);
    // 8 waves per block for the fused kernels: a block covers 8 consecutive strips of one row segment (4 KB of
    // contiguous pixels per row at 2048 columns), and at most 8 march waves sit on a CU next to the other
    // streams' row-filter blocks: +4 % in the 4-stream run (DSX_FWD_WPB / DSX_INV_WPB = 4 restores 4)
    const int fwd_wpb = ctx->tune.fwd_wpb;
    if (fuse12 && l == 0 && fwd_wpb == 8) {
      const dim3 g8((f.nstrips * f.nseg + 7) / 8, nb);
      if (in_dtype == DSX_U16) hipLaunchKernelGGL((dsx::k_fwd_march<0, true, 8>), g8, dim3(512), 0, s, f);
      else hipLaunchKernelGGL((dsx::k_fwd_march<1, true, 8>), g8, dim3(512), 0, s, f);
    } else if (fuse12 && l == 0) {
      if (in_dtype == DSX_U16) hipLaunchKernelGGL((dsx::k_fwd_march<0, true>), grid, dim3(256), 0, s, f);
      else hipLaunchKernelGGL((dsx::k_fwd_march<1, true>), grid, dim3(256), 0, s, f);
    } else if (l > 0) {
      hipLaunchKernelGGL(dsx::k_fwd_march<2>, grid, dim3(256), 0, s, f);
    } else if (in_dtype == DSX_U16) {
      hipLaunchKernelGGL(dsx::k_fwd_march<0>, grid, dim3(256), 0, s, f);
    } else {
      hipLaunchKernelGGL(dsx::k_fwd_march<1>, grid, dim3(256), 0, s, f);
    }
    DSX_HIP(hipGetLastError());
    if (split && l == 0) {  // levels 1 and 2 are complete: their histograms start now, on the helper stream
      DSX_HIP(hipEventRecord(v.ev[0], s));
      DSX_HIP(hipStreamWaitEvent(v.helper, v.ev[0], 0));
      if (int rc = hist_levels(0, 2, v.helper)) return rc;
      DSX_HIP(hipEventRecord(v.ev[1], v.helper));
    }
  }

  // ---- thresholds -----------------------------------------------------------------------------
  if (int rc = hist_levels(split ? 2 : 0, L, s)) return rc;
  if (split) DSX_HIP(hipStreamWaitEvent(s, v.ev[1], 0));  // histograms of levels 1, 2 (helper stream)
  if (L > 0) {
    dsx::OtsuArgs a;
    a.stats = v.stats;
    a.npix = (double)p.H * (double)p.W;
    a.high_int = ctx->high_int;
    a.minmax = v.minmax;
    a.hist = v.hist;
    a.thr = v.thr;
    a.otsu = v.otsu;
    a.cfg = v.cfg;
    a.means = v.means;
    a.max_thr[0] = (float)ctx->cfg[0].max_threshold;
    a.max_thr[1] = (float)ctx->cfg[1].max_threshold;
    a.L = L;
    a.sticky = ctx->d_sticky;
    a.shared = ctx->stack_mode ? 1 : 0;
    LaunchScope ls(ctx, KC_OTSU);
    hipLaunchKernelGGL(dsx::k_otsu, dim3(L, nb), dim3(64), 0, s, a);
    DSX_HIP(hipGetLastError());
    if (d_cfg_used)
      DSX_HIP(hipMemcpyAsync(d_cfg_used, v.cfg, sizeof(int32_t) * nb, hipMemcpyDeviceToDevice, s));
  } else if (d_cfg_used) {
    DSX_HIP(hipMemsetAsync(d_cfg_used, 0, sizeof(int32_t) * nb, s));
  }
  if (ctx->stop_after == 1) return DSX_OK;

  // ---- row filter -----------------------------------------------------------------------------
  // split: levels 1, 2 on the helper stream (1.5 ms of the 5.3 ms chain per 256 planes), the coarse levels'
  // row filters AND the coarse inverse levels on the part's own stream beside them; joined before the final kernel
  const bool split_inv = split && fuse21;  // the fused final kernel consumes Delta_1, Delta_2 and c_2 only
  if (split_inv) {
    DSX_HIP(hipEventRecord(v.ev[2], s));                  // thresholds are known
    DSX_HIP(hipStreamWaitEvent(v.helper, v.ev[2], 0));
  }
  dsx::RowArgs row1;  // level 1, for k_rowfinal
  memset(&row1, 0, sizeof(row1));
  // coarse levels (M <= 384) that run on the part's own stream: one launch for all of them (DSX_NO_ROW_MULTI=1: one each)
  const bool no_multi = ctx->tune.no_row_multi;
  dsx::RowMultiArgs multi;
  memset(&multi, 0, sizeof(multi));
  int multi_pairs[dsx::kRowMultiMax] = {};
  for (int l = 0; l < L && l < skip_from; ++l) {
    hipStream_t rs = (split_inv && l < 2) ? v.helper : s;
    if ((skip_row >> l) & 1) continue;
    const dsx::LevelPlan& lp = p.lv[l];
    dsx::RowArgs a;
    memset(&a, 0, sizeof(a));
    a.ws = v.ws;
    a.ws_plane_stride = p.plane_floats;
    a.da_off = lp.da_off;
    a.h = lp.h; a.w = lp.w; a.ld = lp.ld;
    a.thr = v.thr;
    a.cfg = v.cfg;
    a.lvl = l; a.L = L;
    a.lvl_active[0] = p.cfg_levels[0];
    a.lvl_active[1] = p.cfg_levels[1];
    a.M = lp.M; a.K = lp.K;
    a.npass = lp.npass;
    for (int i = 0; i < lp.npass; ++i) a.radix[i] = lp.radix[i];
    a.tw = (const float2*)(ctx->d_consts + lp.tw_off);
    a.g[0] = (const float2*)(ctx->d_consts + lp.g_off[0]);
    a.g[1] = (const float2*)(ctx->d_consts + lp.g_off[1]);
    a.kcut[0] = lp.kcut[0];
    a.kcut[1] = lp.kcut[1];
    a.inv_M = 1.0f / (float)lp.M;
    a.ablate = ctx->ablate;
    if (fuse_rf && l == 0) {
      row1 = a;
      continue;
    }
    const int npairs = (lp.h + 1) / 2;
    // (only for cohorts split over the streams, where the launch gaps of six small kernels cost more than the
    //  specialised CPL = 2 / 4 instantiations save: +2.4 % there, -5 % for a cohort that runs alone -- its coarse chain
    //  is the critical path beside the level-2 row filter on the helper stream; profiles/r3_merged_small_launches_ab.txt)
    // ... and small cohorts that run alone are launch-bound again: one plane 418 -> 358 us per call, 32 planes + 7 %,
    // 64 planes even, 128 planes - 2.7 % (profiles/r3_merged_small_launches_ab.txt); DSX_ROW_MULTI_ALONE = the limit
    const int multi_alone = ctx->tune.row_multi_alone;
    if (!no_multi && (!v.alone || nb <= multi_alone) && rs == s && a.M <= 6 * 64 && multi.nlev < dsx::kRowMultiMax) {
      multi_pairs[multi.nlev] = npairs;
      multi.lv[multi.nlev++] = a;
      continue;
    }
    LaunchScope ls(ctx, KC_ROW);
    DSX_HIP(dispatch_rowfilter(a, npairs, nb, rs));
  }
  if (multi.nlev == 1) {
    LaunchScope ls(ctx, KC_ROW);
    DSX_HIP(dispatch_rowfilter(multi.lv[0], multi_pairs[0], nb, s));
  } else if (multi.nlev > 1) {
    LaunchScope ls(ctx, KC_ROW);
    DSX_HIP(launch_rowfilter_multi(multi, multi_pairs, nb, s));
  }
  if (split_inv) DSX_HIP(hipEventRecord(v.ev[3], v.helper));
  if (ctx->stop_after == 2) return DSX_OK;

  // ---- inverse transform of the Delta pyramid + finish ---------------------------------------
  // level-2 synthesis inside the final kernel when the plane allows it (c_1 never leaves the chip): fuse21
  for (int l = L - 1; l >= (L > 0 ? 0 : -1); --l) {
    if (fuse21 && l == 1) continue;
    if (l >= skip_from) continue;
    if (generic) {
      dsx::GenInvArgs g;
      memset(&g, 0, sizeof(g));
      const dsx::LevelPlan& lp = p.lv[l];
      g.ws = v.ws;
      g.ws_plane_stride = p.plane_floats;
      g.c_off = lp.aa_off; g.d_off = lp.da_off;
      g.hc = lp.h; g.wc = lp.w; g.ldc = lp.lda; g.ldd = lp.ld;
      g.has_c = (l < L - 1) ? 1 : 0;
      g.has_pyr = 1;
      g.F = ctx->wl_len;
      memcpy(g.lo, ctx->wl[2], sizeof(float) * ctx->wl_len);
      memcpy(g.hi, ctx->wl[3], sizeof(float) * ctx->wl_len);
      const bool last_g = (l == 0);
      if (!last_g) {
        const dsx::LevelPlan& lo = p.lv[l - 1];
        g.out_off = lo.aa_off;
        g.hout = lo.h; g.wout = lo.w; g.ldout = lo.lda;
      } else {
        g.img = d_in;
        g.img_plane_stride = (long long)p.H * p.W;
        g.H = p.H; g.W = p.W;
        g.out = d_out;
        g.out_plane_stride = (long long)p.Hout * p.Wout;
        g.hout = p.Hout; g.wout = p.Wout;
        g.out_dtype = (out_dtype == DSX_U16) ? 0 : 1;
        g.flat = ctx->d_flat;
        g.dark = ctx->d_dark;
        g.dark_ld = ctx->dark_w;
      }
      const dim3 gg((g.wout + dsx::kGenOW - 1) / dsx::kGenOW, (g.hout + dsx::kGenOH - 1) / dsx::kGenOH, nb);
      const size_t smem = dsx::gen_inv_lds_bytes(g.F);
      LaunchScope ls(ctx, last_g ? KC_FINAL : KC_INV);
      if (!last_g) hipLaunchKernelGGL(dsx::k_inv_gen<2>, gg, dim3(256), smem, s, g);
      else if (in_dtype == DSX_U16) hipLaunchKernelGGL(dsx::k_inv_gen<0>, gg, dim3(256), smem, s, g);
      else hipLaunchKernelGGL(dsx::k_inv_gen<1>, gg, dim3(256), smem, s, g);
      DSX_HIP(hipGetLastError());
      continue;
    }
    dsx::FinalArgs f;
    memset(&f, 0, sizeof(f));
    f.ws = v.ws;
    f.ws_plane_stride = p.plane_floats;
    if (l >= 0) {
      const dsx::LevelPlan& lp = p.lv[l];
      f.c_off = lp.aa_off;
      f.d_off = lp.da_off;
      f.hc = lp.h; f.wc = lp.w; f.ldc = lp.lda; f.ldd = lp.ld;
      f.has_c = (l < L - 1) ? 1 : 0;
      f.has_pyr = 1;
    }
    const bool last = (l <= 0);
    if (!last) {
      const dsx::LevelPlan& lo = p.lv[l - 1];
      f.ws_out = v.ws;
      f.out_off = lo.aa_off;
      f.hout = lo.h; f.wout = lo.w; f.ldout = lo.lda;
    } else {
      f.img = d_in;
      f.img_plane_stride = (long long)p.H * p.W;
      f.H = p.H; f.W = p.W;
      f.out = d_out;
      f.out_plane_stride = (long long)p.Hout * p.Wout;
      f.hout = p.Hout; f.wout = p.Wout;
      f.out_dtype = (out_dtype == DSX_U16) ? 0 : 1;
      f.flat = ctx->d_flat;
      f.dark = ctx->d_dark;
      f.dark_ld = ctx->dark_w;
    }
    f.ablate = ctx->ablate;
    f.nstrips = (f.wout + dsx::kMarchCols - 1) / dsx::kMarchCols;
    march_segments(nb, f.nstrips, (f.hout + 1) / 2, &f.nseg, &f.rows_per_seg, 0, 0, /*coarse=*/l >= 2);
    const bool fused = fuse21 && last;
    if (last && split_inv) DSX_HIP(hipStreamWaitEvent(s, v.ev[3], 0));  // Delta_1, Delta_2 (helper stream)
    if (fused) {
      const dsx::LevelPlan& l2 = p.lv[1];
      f.c2_off = l2.aa_off; f.d2_off = l2.da_off;
      f.hc2 = l2.h; f.wc2 = l2.w; f.ldc2 = l2.lda; f.ldd2 = l2.ld;
      f.has_c2 = (L > 2) ? 1 : 0;
      f.has_c = 1;
      f.pair_io = pair_io ? 1 : 0;
      if (f.rows_per_seg & 1) {  // segments start at even level-1 rows
        f.rows_per_seg += 1;
        f.nseg = ((f.hout + 1) / 2 + f.rows_per_seg - 1) / f.rows_per_seg;
      }
    }
    dim3 grid((f.nstrips * f.nseg + 3) / 4, nb);
    LaunchScope ls(ctx, last ? KC_FINAL : KC_INV);
    const int inv_wpb = ctx->tune.inv_wpb;
    if (fused && fuse_rf) {
      dsx::RowFinalArgs rf;
      rf.r = row1;
      rf.f = f;
      DSX_HIP(launch_rowfinal(rf_plan, rf, nb, s));
    } else if (fused && inv_wpb == 8) {
      const dim3 g8((f.nstrips * f.nseg + 7) / 8, nb);
      if (in_dtype == DSX_U16) hipLaunchKernelGGL((dsx::k_inv_march<0, true, 8>), g8, dim3(512), 0, s, f);
      else hipLaunchKernelGGL((dsx::k_inv_march<1, true, 8>), g8, dim3(512), 0, s, f);
    } else if (fused) {
      if (in_dtype == DSX_U16) hipLaunchKernelGGL((dsx::k_inv_march<0, true>), grid, dim3(256), 0, s, f);
      else hipLaunchKernelGGL((dsx::k_inv_march<1, true>), grid, dim3(256), 0, s, f);
    } else if (!last) {
      hipLaunchKernelGGL(dsx::k_inv_march<2>, grid, dim3(256), 0, s, f);
    } else if (in_dtype == DSX_U16) {
      hipLaunchKernelGGL(dsx::k_inv_march<0>, grid, dim3(256), 0, s, f);
    } else {
      hipLaunchKernelGGL(dsx::k_inv_march<1>, grid, dim3(256), 0, s, f);
    }
    DSX_HIP(hipGetLastError());
  }
  return DSX_OK;
}

size_t elem_size(int dtype) { return dtype == DSX_U16 ? 2 : 4; }

// helper: 1 = this part may use its helper stream (run_cohort).  Measured at 2048^2: a cohort that runs as ONE
// part gains 6 % (52.5 k against 49.5 k planes/s), a cohort split over 4 streams LOSES 2-8 % (the other parts
// already fill the chip, the extra streams only add contention), so only unsplit cohorts use it; DSX_HELPER=0 / 1
// forces it off / on.
CohortView make_view(dsx_ctx* ctx, int po, hipStream_t stream, int part = 0, bool helper = false) {
  const int Lc = ctx->plan.L > 0 ? ctx->plan.L : 1;
  CohortView v;
  v.stream = stream;
  const int force_helper = ctx->tune.helper;
  const bool use = force_helper >= 0 ? force_helper != 0 : helper;
  v.helper = (!use || ctx->profiling || ctx->stop_after != 0) ? nullptr : ctx->helper[part];
  v.alone = helper;
  v.ev = ctx->ev_h[part];
  v.ws = ctx->d_ws + (size_t)po * ctx->plan.plane_floats;
  v.stats = ctx->d_stats + po;
  v.minmax = ctx->d_minmax + (size_t)po * Lc * 2;
  v.hist = ctx->d_hist + (size_t)po * Lc * 256;
  v.thr = ctx->d_thr + (size_t)po * Lc;
  v.otsu = ctx->d_otsu + (size_t)po * Lc;
  v.cfg = ctx->d_cfg + po;
  v.means = ctx->d_means + 2 * (size_t)po;
  return v;
}

// One cohort (<= max_batch planes), split into parts on the context's streams.
//
// Part 0 runs on the context stream, parts 1.. on the auxiliary streams behind a fork event.  The context
// stream does NOT wait for the other parts here: use_main() does that when an entry point needs the stream
// next.  If the call that follows is another split cohort with the same planes pointer, result pointer,
// count and element types (and nothing has been put on the context stream in between), every part is
// simply queued behind the same part of this call -- it reads the input nobody writes and rewrites its
// own slice of the workspace, control block and result -- so consecutive batches overlap instead of
// draining the chip at every call boundary.  Anything else joins first, then forks again.
int run_cohort_split(dsx_ctx* ctx, const void* d_in, int in_dtype, int nb, void* d_out, int out_dtype,
                     int32_t* d_cfg_used, size_t in_plane, size_t out_plane) {
  int parts = ctx->n_streams;
  if (ctx->profiling || ctx->stop_after != 0 || ctx->stack_mode) parts = 1;  // stack mode: one control block, one chain
  while (parts > 1 && nb / parts < 16) --parts;  // keep every part big enough to fill the chip
  if (parts <= 1) {
    hipStream_t main = use_main(ctx);  // joins whatever split cohort is still running
    const CohortView view = make_view(ctx, 0, main, 0, true);
    if (ctx->graph_mode == 0 || ctx->profiling || ctx->stop_after != 0 || ctx->ablate != 0)
      return run_cohort(ctx, view, d_in, in_dtype, nb, d_out, out_dtype, d_cfg_used);
    // ---- graph replay of a call seen before (same buffers, count and types under the current plan) ----
    dsx_ctx::GraphEntry* hit = nullptr;
    for (auto& g : ctx->graphs)
      if (g.in == d_in && g.out == d_out && g.cfg == (const void*)d_cfg_used && g.n == nb && g.in_dtype == in_dtype &&
          g.out_dtype == out_dtype) { hit = &g; break; }
    if (!hit) {
      if ((int)ctx->graphs.size() >= dsx_ctx::kGraphSlots) {  // evict the least recently used tuple
        size_t lru = 0;
        for (size_t i = 1; i < ctx->graphs.size(); ++i)
          if (ctx->graphs[i].last_use < ctx->graphs[lru].last_use) lru = i;
        if (ctx->graphs[lru].exec) (void)hipGraphExecDestroy(ctx->graphs[lru].exec);
        ctx->graphs.erase(ctx->graphs.begin() + (long)lru);
      }
      ctx->graphs.push_back({d_in, d_out, (const void*)d_cfg_used, nb, in_dtype, out_dtype, 0, nullptr, 0});
      hit = &ctx->graphs.back();
    }
    hit->last_use = ++ctx->graph_clock;
    if (hit->exec) {
      ++ctx->graph_launches;
      DSX_HIP(hipGraphLaunch(hit->exec, main));
      return DSX_OK;
    }
    if (hit->seen++ == 0)  // first sight: eager (one-off calls never pay for a capture; function attributes get set)
      return run_cohort(ctx, view, d_in, in_dtype, nb, d_out, out_dtype, d_cfg_used);
    // second sight: capture the chain (the helper stream joins the capture through its fork event and is joined
    // back before the final kernel), instantiate, launch
    if (hipStreamBeginCapture(main, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();
      ctx->graph_mode = 0;
      return run_cohort(ctx, view, d_in, in_dtype, nb, d_out, out_dtype, d_cfg_used);
    }
    const int rc = run_cohort(ctx, view, d_in, in_dtype, nb, d_out, out_dtype, d_cfg_used);
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(main, &graph);
    hipGraphExec_t exec = nullptr;
    if (rc == DSX_OK && ce == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
      (void)hipGraphDestroy(graph);
      hit->exec = exec;
      ++ctx->graph_captures;
      ++ctx->graph_launches;
      DSX_HIP(hipGraphLaunch(exec, main));
      return DSX_OK;
    }
    // capture is not usable here: nothing has run yet -- fall back to eager launches for good
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    ctx->graph_mode = 0;
    if (rc != DSX_OK) return rc;
    return run_cohort(ctx, view, d_in, in_dtype, nb, d_out, out_dtype, d_cfg_used);
  }
  const bool no_pipe = ctx->tune.no_pipeline;
  auto& ls = ctx->last_split;
  const bool in_out_disjoint = (const char*)d_in + nb * in_plane <= (const char*)d_out ||
                               (const char*)d_out + nb * out_plane <= (const char*)d_in;
  const bool same_call = !no_pipe && ctx->joins_pending == parts && ls.in == d_in && ls.out == d_out && ls.n == nb &&
                         ls.in_dtype == in_dtype && ls.out_dtype == out_dtype && ls.parts == parts &&
                         ls.cfg == (const void*)d_cfg_used && ls.main_ops == ctx->main_ops && in_out_disjoint;
  if (!same_call) {
    hipStream_t main = use_main(ctx);
    DSX_HIP(hipEventRecord(ctx->ev_fork, main));
  }
  const int per = (nb + parts - 1) / parts;
  for (int i = 0; i < parts; ++i) {
    const int po = i * per, n = std::min(per, nb - po);
    if (n <= 0) break;
    hipStream_t st = (i == 0) ? ctx->stream_ : ctx->aux[i];
    if (i > 0 && !same_call) DSX_HIP(hipStreamWaitEvent(st, ctx->ev_fork, 0));
    const int rc = run_cohort(ctx, make_view(ctx, po, st, i), (const char*)d_in + po * in_plane, in_dtype, n,
                              (char*)d_out + po * out_plane, out_dtype, d_cfg_used ? d_cfg_used + po : nullptr);
    if (rc != DSX_OK) return rc;
    if (i > 0) DSX_HIP(hipEventRecord(ctx->ev_join[i], st));
  }
  ctx->joins_pending = parts;
  ls.in = d_in; ls.out = d_out; ls.cfg = d_cfg_used;
  ls.n = nb; ls.in_dtype = in_dtype; ls.out_dtype = out_dtype; ls.parts = parts;
  ls.main_ops = ctx->main_ops;
  return DSX_OK;
}

// Value errors of planes that went through the asynchronous device-buffer path (see dsx_ctx::h_sticky): reported once,
// by the first synchronising entry point after the kernels that found them.
int check_sticky(dsx_ctx* ctx) {
  if (!ctx->h_sticky) return DSX_OK;
  // one atomic exchange: a kernel of another stream may be storing its 1 right now -- read-then-clear could lose it
  const unsigned f = __atomic_exchange_n(ctx->h_sticky, 0u, __ATOMIC_ACQ_REL);
  if (f == 0u) return DSX_OK;
  return fail(ctx, DSX_EVALUE, "a plane of an earlier dsx_run_device call: autodetected range of [nan, nan] is not finite "
                               "(a float32 pixel is NaN, infinite or <= -1); that plane's result is not valid");
}

}  // namespace

extern "C" {

int dsx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* dsx_last_error(const dsx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int dsx_init(int device, dsx_ctx** out_ctx) {
  if (!out_ctx) { g_init_error = "out_ctx is NULL"; return DSX_EINVAL; }
  *out_ctx = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_init_error = std::string("no HIP device available: ") + hipGetErrorString(e);
    return DSX_EHIP;
  }
  if (device < 0 || device >= n) { g_init_error = "device index out of range"; return DSX_EINVAL; }
  e = hipSetDevice(device);
  if (e != hipSuccess) { g_init_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return DSX_EHIP; }
  dsx_ctx* c = new dsx_ctx();
  c->device = device;
  read_tuning(c->tune);
#ifdef DSX_DIAG
  if (const char* ab = getenv("DSX_ABLATE")) c->ablate = atoi(ab);
#endif
  if (const char* gm = getenv("DSX_GRAPH")) c->graph_mode = atoi(gm) != 0 ? 1 : 0;
  if (const char* nf = getenv("DSX_NO_FUSE_RF")) c->no_fuse_rf = atoi(nf) != 0;
  if (const char* fw = getenv("DSX_FUSE_RF_WIDE")) c->fuse_rf_wide = atoi(fw) != 0;
  if (const char* ns = getenv("DSX_STREAMS")) c->n_streams = std::max(1, std::min(atoi(ns), (int)dsx_ctx::kMaxStreams));
  // DSX_PRIO=p0,p1,...: stream priority per sub-cohort stream (experiment hook; default: all equal)
  int prio[dsx_ctx::kMaxStreams] = {};
  if (const char* pr = getenv("DSX_PRIO")) {
    int i = 0;
    for (const char* q = pr; *q && i < dsx_ctx::kMaxStreams; ++i) {
      prio[i] = atoi(q);
      while (*q && *q != ',') ++q;
      if (*q == ',') ++q;
    }
  }
  e = hipStreamCreateWithPriority(&c->stream_, hipStreamNonBlocking, prio[0]);
  for (int i = 1; i < dsx_ctx::kMaxStreams && e == hipSuccess; ++i) {
    e = hipStreamCreateWithPriority(&c->aux[i], hipStreamNonBlocking, prio[i]);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipStreamCreateWithFlags(&c->copy_stream[i], hipStreamNonBlocking);
  for (int i = 0; i < dsx_ctx::kMaxStreams && e == hipSuccess; ++i) {
    e = hipStreamCreateWithFlags(&c->helper[i], hipStreamNonBlocking);
    for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&c->ev_h[i][k], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_xs, hipEventDisableTiming);
  for (int i = 0; i < dsx_ctx::kEventSlots && e == hipSuccess; ++i)
    e = hipEventCreateWithFlags(&c->ev_slot[i], hipEventDisableTiming);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_sticky, 64, hipHostMallocMapped);
  if (e == hipSuccess) {
    *c->h_sticky = 0u;
    e = hipHostGetDevicePointer((void**)&c->d_sticky, c->h_sticky, 0);
  }
  if (e == hipSuccess) e = hipEventCreate(&c->t0);
  if (e == hipSuccess) e = hipEventCreate(&c->t1);
  if (e != hipSuccess) {
    g_init_error = std::string("stream/event creation: ") + hipGetErrorString(e);
    delete c;
    return DSX_EHIP;
  }
  *out_ctx = c;
  return DSX_OK;
}

void dsx_destroy(dsx_ctx* ctx) {
  if (!ctx) return;
  if (t_tune == &ctx->tune) t_tune = nullptr;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(use_main(ctx));
  (void)dsx_comm_destroy(ctx);
  for (auto& r : ctx->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  free_plan_buffers(ctx);
  if (ctx->t0) (void)hipEventDestroy(ctx->t0);
  if (ctx->t1) (void)hipEventDestroy(ctx->t1);
  for (int i = 1; i < dsx_ctx::kMaxStreams; ++i) {
    if (ctx->aux[i]) { (void)hipStreamSynchronize(ctx->aux[i]); (void)hipStreamDestroy(ctx->aux[i]); }
    if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  for (int i = 0; i < dsx_ctx::kMaxStreams; ++i) {
    if (ctx->helper[i]) { (void)hipStreamSynchronize(ctx->helper[i]); (void)hipStreamDestroy(ctx->helper[i]); }
    for (int k = 0; k < 4; ++k) if (ctx->ev_h[i][k]) (void)hipEventDestroy(ctx->ev_h[i][k]);
  }
  for (int i = 0; i < 2; ++i)
    if (ctx->copy_stream[i]) { (void)hipStreamSynchronize(ctx->copy_stream[i]); (void)hipStreamDestroy(ctx->copy_stream[i]); }
  if (ctx->h_sticky) (void)hipHostFree(ctx->h_sticky);
  if (ctx->ev_xs) (void)hipEventDestroy(ctx->ev_xs);
  for (int i = 0; i < dsx_ctx::kEventSlots; ++i)
    if (ctx->ev_slot[i]) (void)hipEventDestroy(ctx->ev_slot[i]);
  if (ctx->stream_) (void)hipStreamDestroy(ctx->stream_);
  delete ctx;
}

int dsx_plan(dsx_ctx* ctx, int height, int width, int max_batch, const dsx_cfg* cells_config,
             const dsx_cfg* no_cells_config, double microscope_high_int, const float* flat,
             const float* dark, int dark_h, int dark_w) {
  if (!ctx) return DSX_EINVAL;
  if (!cells_config || !no_cells_config) return fail(ctx, DSX_EINVAL, "config is NULL");
  if (max_batch < 1) return fail(ctx, DSX_EINVAL, "max_batch must be >= 1");
  if ((flat == nullptr) != (dark == nullptr))
    return fail(ctx, DSX_EINVAL, "flatfield and darkfield must be given together");
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  free_plan_buffers(ctx);

  const dsx_cfg* src[2] = {no_cells_config, cells_config};  // index 1 == cells_config
  for (int c = 0; c < 2; ++c) {
    // both configs run through ONE decomposition (the config is only decided once the level-1 statistic is known)
    if (src[c]->wavelet != DSX_WAVELET_DB3 && src[c]->wavelet != DSX_WAVELET_BANK)
      return fail(ctx, DSX_EINVAL, "unknown wavelet id");
    if (src[c]->wavelet != src[0]->wavelet)
      return fail(ctx, DSX_EINVAL, "cells_config and no_cells_config must name the same wavelet");
    ctx->cfg[c].level = src[c]->level;
    ctx->cfg[c].sigma = src[c]->sigma;
    ctx->cfg[c].max_threshold = src[c]->max_threshold;
  }
  const bool bank = src[0]->wavelet == DSX_WAVELET_BANK;
  if (bank && !ctx->wl_set) return fail(ctx, DSX_EINVAL, "DSX_WAVELET_BANK without dsx_set_wavelet");
  ctx->wl_len = bank ? ctx->wl_bank_len : 0;
  if (bank) {
    memcpy(ctx->wl, ctx->wl_bank, sizeof(ctx->wl));
    // the level kernels stage their patches in dynamic LDS: up to 110 KB at 104 taps
    const void* fns[6] = {(const void*)dsx::k_fwd_gen<0>, (const void*)dsx::k_fwd_gen<1>, (const void*)dsx::k_fwd_gen<2>,
                          (const void*)dsx::k_inv_gen<0>, (const void*)dsx::k_inv_gen<1>, (const void*)dsx::k_inv_gen<2>};
    for (const void* fn : fns)
      DSX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  const std::string perr = dsx::build_plan(height, width, ctx->cfg, ctx->plan, bank ? ctx->wl_bank_len : dsx::kFilterLen);
  if (!perr.empty()) {
    const bool limit = perr.find("too") != std::string::npos;
    return fail(ctx, limit ? DSX_ELIMIT : DSX_EINVAL, perr);
  }
  const dsx::Plan& p = ctx->plan;
  ctx->high_int = microscope_high_int;
  ctx->max_batch = max_batch;
  const int L = p.L, Lc = L > 0 ? L : 1;

  const size_t ws_bytes = (size_t)max_batch * p.plane_floats * sizeof(float);
  DSX_HIP(hipMalloc((void**)&ctx->d_ws, ws_bytes));
  const size_t stats_b = sizeof(dsx::PlaneStats) * max_batch;
  const size_t minmax_b = sizeof(unsigned) * 2 * Lc * max_batch;
  const size_t hist_b = sizeof(unsigned) * 256 * Lc * max_batch;
  ctx->ctl_zero_bytes = stats_b + minmax_b + hist_b;
  DSX_HIP(hipMalloc((void**)&ctx->d_ctl, ctx->ctl_zero_bytes));
  ctx->d_stats = (dsx::PlaneStats*)ctx->d_ctl;
  ctx->d_minmax = (unsigned*)(ctx->d_ctl + stats_b);
  ctx->d_hist = (unsigned*)(ctx->d_ctl + stats_b + minmax_b);
  DSX_HIP(hipMalloc((void**)&ctx->d_thr, sizeof(float) * Lc * max_batch));
  DSX_HIP(hipMalloc((void**)&ctx->d_otsu, sizeof(float) * Lc * max_batch));
  DSX_HIP(hipMalloc((void**)&ctx->d_cfg, sizeof(int) * max_batch));
  DSX_HIP(hipMalloc((void**)&ctx->d_means, sizeof(double) * 2 * max_batch));
  DSX_HIP(hipMemset(ctx->d_cfg, 0, sizeof(int) * max_batch));
  DSX_HIP(hipMemset(ctx->d_means, 0, sizeof(double) * 2 * max_batch));
  DSX_HIP(hipMemset(ctx->d_thr, 0, sizeof(float) * Lc * max_batch));
  DSX_HIP(hipMemset(ctx->d_otsu, 0, sizeof(float) * Lc * max_batch));
  ctx->consts_bytes = std::max<size_t>(p.consts.size(), 1) * sizeof(dsx::C32);
  DSX_HIP(hipMalloc((void**)&ctx->d_consts, ctx->consts_bytes));
  if (!p.consts.empty())
    DSX_HIP(hipMemcpy(ctx->d_consts, p.consts.data(), p.consts.size() * sizeof(dsx::C32), hipMemcpyHostToDevice));
  ctx->workspace_bytes = ws_bytes + ctx->ctl_zero_bytes + ctx->consts_bytes;

  if (flat) {
    if (dark_h < p.Hout || dark_w < p.Wout)
      return fail(ctx, DSX_EINVAL, "Please, check the shape of the darkfield.");
    const size_t fb = sizeof(float) * (size_t)p.Hout * p.Wout;
    const size_t db = sizeof(float) * (size_t)dark_h * dark_w;
    DSX_HIP(hipMalloc((void**)&ctx->d_flat, fb));
    DSX_HIP(hipMalloc((void**)&ctx->d_dark, db));
    ctx->own_shading = true;
    DSX_HIP(hipMemcpy(ctx->d_flat, flat, fb, hipMemcpyHostToDevice));
    DSX_HIP(hipMemcpy(ctx->d_dark, dark, db, hipMemcpyHostToDevice));
    ctx->dark_h = dark_h;
    ctx->dark_w = dark_w;
    ctx->workspace_bytes += fb + db;
  }
  ctx->planned = true;
  ctx->last_n = 0;
  return DSX_OK;
}

int dsx_set_wavelet(dsx_ctx* ctx, const double* dec_lo, const double* dec_hi, const double* rec_lo,
                    const double* rec_hi, int len) {
  if (!ctx) return DSX_EINVAL;
  if (!dec_lo || !dec_hi || !rec_lo || !rec_hi) return fail(ctx, DSX_EINVAL, "filter pointer is NULL");
  if (len < 2 || (len & 1)) return fail(ctx, DSX_EINVAL, "wavelet filters must have an even number (>= 2) of taps");
  if (len > dsx::kMaxTaps) return fail(ctx, DSX_ELIMIT, "wavelet filters are longer than the kernels take");
  const double* src[4] = {dec_lo, dec_hi, rec_lo, rec_hi};
  for (int f = 0; f < 4; ++f)
    for (int t = 0; t < len; ++t) {
      if (!(fabs(src[f][t]) < 1e30)) return fail(ctx, DSX_EINVAL, "wavelet filter coefficient is not finite");
      ctx->wl_bank[f][t] = (float)src[f][t];
    }
  ctx->wl_bank_len = len;
  ctx->wl_set = true;
  return DSX_OK;  // takes effect at the next dsx_plan whose configs carry DSX_WAVELET_BANK
}

int dsx_set_shading_device(dsx_ctx* ctx, const float* d_flat, const float* d_dark, int dark_h,
                           int dark_w) {
  if (!ctx) return DSX_EINVAL;
  if (!ctx->planned) return fail(ctx, DSX_ENOPLAN, "dsx_plan has not been called");
  if ((d_flat == nullptr) != (d_dark == nullptr))
    return fail(ctx, DSX_EINVAL, "flatfield and darkfield must be given together");
  if (d_flat && (dark_h < ctx->plan.Hout || dark_w < ctx->plan.Wout))
    return fail(ctx, DSX_EINVAL, "Please, check the shape of the darkfield.");
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  drop_graphs(ctx);  // captured final kernels carry the old shading addresses
  if (ctx->own_shading) { (void)hipFree(ctx->d_flat); (void)hipFree(ctx->d_dark); }
  ctx->own_shading = false;
  ctx->d_flat = const_cast<float*>(d_flat);
  ctx->d_dark = const_cast<float*>(d_dark);
  ctx->dark_h = dark_h;
  ctx->dark_w = dark_w;
  return DSX_OK;
}

int dsx_graph_stats(const dsx_ctx* ctx, uint64_t* launches, uint64_t* captures) {
  if (!ctx || !launches || !captures) return DSX_EINVAL;
  *launches = ctx->graph_launches;
  *captures = ctx->graph_captures;
  return DSX_OK;
}

int dsx_constants_device(const dsx_ctx* ctx, void** d_ptr, size_t* bytes) {
  if (!ctx || !ctx->planned || !d_ptr || !bytes) return DSX_EINVAL;
  *d_ptr = ctx->d_consts;
  *bytes = ctx->plan.consts.size() * sizeof(dsx::C32);
  return DSX_OK;
}

int dsx_plan_info(const dsx_ctx* ctx, dsx_plan_info_t* info) {
  if (!ctx || !info) return DSX_EINVAL;
  if (!ctx->planned) return DSX_ENOPLAN;
  memset(info, 0, sizeof(*info));
  const dsx::Plan& p = ctx->plan;
  info->height = p.H; info->width = p.W;
  info->out_height = p.Hout; info->out_width = p.Wout;
  info->levels = p.L;
  for (int l = 0; l < p.L && l < 16; ++l) {
    info->level_h[l] = p.lv[l].h;
    info->level_w[l] = p.lv[l].w;
    info->fft_len[l] = p.lv[l].M;
    info->fft_halo[l] = p.lv[l].K;
  }
  info->max_batch = ctx->max_batch;
  info->workspace_bytes = ctx->workspace_bytes;
  return DSX_OK;
}

int dsx_run_device(dsx_ctx* ctx, const void* d_in, int in_dtype, int n, void* d_out, int out_dtype,
                   int32_t* d_cfg_used) {
  if (!ctx) return DSX_EINVAL;
  if (!ctx->planned) return fail(ctx, DSX_ENOPLAN, "dsx_plan has not been called");
  if (n < 0 || (n > 0 && (!d_in || !d_out))) return fail(ctx, DSX_EINVAL, "bad plane pointers / count");
  if ((in_dtype != DSX_U16 && in_dtype != DSX_F32) || (out_dtype != DSX_U16 && out_dtype != DSX_F32))
    return fail(ctx, DSX_EINVAL, "unknown element type");
  DSX_HIP(hipSetDevice(ctx->device));
  if (ctx->stack_mode && n > ctx->max_batch)
    return fail(ctx, DSX_ELIMIT, "stack mode: all planes of the stack must fit one cohort (plan with max_batch >= n)");
  const dsx::Plan& p = ctx->plan;
  const size_t in_plane = (size_t)p.H * p.W * elem_size(in_dtype);
  const size_t out_plane = (size_t)p.Hout * p.Wout * elem_size(out_dtype);
  for (int start = 0; start < n; start += ctx->max_batch) {
    const int nb = std::min(ctx->max_batch, n - start);
    const int rc = run_cohort_split(ctx, (const char*)d_in + start * in_plane, in_dtype, nb,
                                    (char*)d_out + start * out_plane, out_dtype,
                                    d_cfg_used ? d_cfg_used + start : nullptr, in_plane, out_plane);
    if (rc != DSX_OK) return rc;
    ctx->last_n = nb;
  }
  return DSX_OK;
}

int dsx_sync(dsx_ctx* ctx) {
  if (!ctx) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  return check_sticky(ctx);
}

int dsx_run_host(dsx_ctx* ctx, const void* in, int in_dtype, int n, void* out, int out_dtype,
                 int32_t* cfg_used) {
  if (!ctx) return DSX_EINVAL;
  if (!ctx->planned) return fail(ctx, DSX_ENOPLAN, "dsx_plan has not been called");
  if (n < 0 || (n > 0 && (!in || !out))) return fail(ctx, DSX_EINVAL, "bad plane pointers / count");
  if ((in_dtype != DSX_U16 && in_dtype != DSX_F32) || (out_dtype != DSX_U16 && out_dtype != DSX_F32))
    return fail(ctx, DSX_EINVAL, "unknown element type");
  DSX_HIP(hipSetDevice(ctx->device));
  const dsx::Plan& p = ctx->plan;
  const size_t in_plane = (size_t)p.H * p.W * elem_size(in_dtype);
  const size_t out_plane = (size_t)p.Hout * p.Wout * elem_size(out_dtype);
  const int B = ctx->max_batch;
  if (ctx->stack_mode && n > B)
    return fail(ctx, DSX_ELIMIT, "stack mode: all planes of the stack must fit one cohort (plan with max_batch >= n)");
  const size_t need_in = in_plane * B, need_out = out_plane * B + sizeof(int32_t) * B;
  if (ctx->stage_in_bytes < need_in) {
    if (ctx->d_stage_in) (void)hipFree(ctx->d_stage_in);
    ctx->d_stage_in = nullptr; ctx->stage_in_bytes = 0;
    DSX_HIP(hipMalloc(&ctx->d_stage_in, need_in));
    ctx->stage_in_bytes = need_in;
  }
  if (ctx->stage_out_bytes < need_out) {
    if (ctx->d_stage_out) (void)hipFree(ctx->d_stage_out);
    ctx->d_stage_out = nullptr; ctx->stage_out_bytes = 0;
    DSX_HIP(hipMalloc(&ctx->d_stage_out, need_out));
    ctx->stage_out_bytes = need_out;
  }
  int32_t* d_cfg = (int32_t*)((char*)ctx->d_stage_out + out_plane * B);
  // A value error of an EARLIER asynchronous dsx_run_device call on this context that nobody has synchronised with yet
  // is reported now (this call synchronises) instead of being wiped by this call's own bookkeeping below.
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  if (int rc = check_sticky(ctx)) return rc;
  for (int start = 0; start < n; start += B) {
    const int nb = std::min(B, n - start);
    DSX_HIP(hipMemcpyAsync(ctx->d_stage_in, (const char*)in + start * in_plane, in_plane * nb,
                           hipMemcpyHostToDevice, use_main(ctx)));
    const int rc = run_cohort_split(ctx, ctx->d_stage_in, in_dtype, nb, ctx->d_stage_out, out_dtype, d_cfg,
                                    in_plane, out_plane);
    if (rc != DSX_OK) return rc;
    ctx->last_n = nb;
    if (ctx->stop_after == 0)
      DSX_HIP(hipMemcpyAsync((char*)out + start * out_plane, ctx->d_stage_out, out_plane * nb,
                             hipMemcpyDeviceToHost, use_main(ctx)));
    if (cfg_used)
      DSX_HIP(hipMemcpyAsync(cfg_used + start, d_cfg, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, use_main(ctx)));
    DSX_HIP(hipStreamSynchronize(use_main(ctx)));
    // float32 planes: the forward kernel flags pixels whose log(1 + x) is not finite (PlaneStats::flags); with at
    // least one decomposition level the reference raises ValueError for such a plane (numpy.histogram inside
    // threshold_otsu: "autodetected range of [nan, nan] is not finite")
    if (p.L > 0) {
      std::vector<dsx::PlaneStats> hs((size_t)nb);
      DSX_HIP(hipMemcpy(hs.data(), ctx->d_stats, sizeof(dsx::PlaneStats) * nb, hipMemcpyDeviceToHost));
      // everything before this call was synchronised and reported above: what the word holds now is this cohort's own,
      // reported right here, per plane
      if (ctx->h_sticky) (void)__atomic_exchange_n(ctx->h_sticky, 0u, __ATOMIC_ACQ_REL);
      for (int k = 0; k < nb && in_dtype == DSX_F32; ++k)
        if (hs[(size_t)k].flags & 1ull)
          return fail(ctx, DSX_EVALUE, "plane " + std::to_string(start + k) +
                                           ": autodetected range of [nan, nan] is not finite (a pixel is NaN, "
                                           "infinite or <= -1)");
    }
  }
  return DSX_OK;
}

/* ---- memory + timing helpers ------------------------------------------------------------------ */
int dsx_malloc(dsx_ctx* ctx, size_t bytes, void** d_ptr) {
  if (!ctx || !d_ptr) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
  if (e != hipSuccess) return fail(ctx, DSX_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
  return DSX_OK;
}
int dsx_free(dsx_ctx* ctx, void* d_ptr) {
  if (!ctx) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  DSX_HIP(hipFree(d_ptr));
  return DSX_OK;
}
int dsx_memcpy_h2d(dsx_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
  if (!ctx) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, use_main(ctx)));
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  return DSX_OK;
}
int dsx_memcpy_d2h(dsx_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
  if (!ctx) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, use_main(ctx)));
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  return DSX_OK;
}
int dsx_memcpy_d2d(dsx_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
  if (!ctx) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, use_main(ctx)));
  return DSX_OK;
}
/* pinned host memory + copies on their own streams: the overlapped chunk map (zarr_destriper.py) */
namespace {
hipStream_t pick_stream(dsx_ctx* ctx, int id) {
  if (id == DSX_STREAM_UPLOAD) return ctx->copy_stream[0];
  if (id == DSX_STREAM_DOWNLOAD) return ctx->copy_stream[1];
  return use_main(ctx);
}
bool stream_id_ok(int id) { return id == DSX_STREAM_COMPUTE || id == DSX_STREAM_UPLOAD || id == DSX_STREAM_DOWNLOAD; }
}  // namespace
int dsx_malloc_host(dsx_ctx* ctx, size_t bytes, void** h_ptr) {
  if (!ctx || !h_ptr) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  hipError_t e = hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) return fail(ctx, DSX_ENOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
  return DSX_OK;
}
int dsx_free_host(dsx_ctx* ctx, void* h_ptr) {
  if (!ctx) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipHostFree(h_ptr));
  return DSX_OK;
}
int dsx_memcpy_h2d_async(dsx_ctx* ctx, void* d_dst, const void* src, size_t bytes, int stream_id) {
  if (!ctx || !stream_id_ok(stream_id)) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, pick_stream(ctx, stream_id)));
  return DSX_OK;
}
int dsx_memcpy_d2h_async(dsx_ctx* ctx, void* dst, const void* d_src, size_t bytes, int stream_id) {
  if (!ctx || !stream_id_ok(stream_id)) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, pick_stream(ctx, stream_id)));
  return DSX_OK;
}
int dsx_stream_wait(dsx_ctx* ctx, int waiter, int signaller) {
  if (!ctx || !stream_id_ok(waiter) || !stream_id_ok(signaller)) return DSX_EINVAL;
  if (waiter == signaller) return DSX_OK;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipEventRecord(ctx->ev_xs, pick_stream(ctx, signaller)));
  DSX_HIP(hipStreamWaitEvent(pick_stream(ctx, waiter), ctx->ev_xs, 0));
  return DSX_OK;
}
int dsx_event_record(dsx_ctx* ctx, int slot, int stream_id) {
  if (!ctx || slot < 0 || slot >= dsx_ctx::kEventSlots || !stream_id_ok(stream_id)) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipEventRecord(ctx->ev_slot[slot], pick_stream(ctx, stream_id)));
  return DSX_OK;
}
int dsx_event_sync(dsx_ctx* ctx, int slot) {
  if (!ctx || slot < 0 || slot >= dsx_ctx::kEventSlots) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipEventSynchronize(ctx->ev_slot[slot]));
  return check_sticky(ctx);
}
int dsx_stream_sync(dsx_ctx* ctx, int stream_id) {
  if (!ctx || !stream_id_ok(stream_id)) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipStreamSynchronize(pick_stream(ctx, stream_id)));
  return check_sticky(ctx);
}
int dsx_timer_start(dsx_ctx* ctx) {
  if (!ctx) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipEventRecord(ctx->t0, use_main(ctx)));
  return DSX_OK;
}
int dsx_timer_stop(dsx_ctx* ctx, float* ms) {
  if (!ctx || !ms) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipEventRecord(ctx->t1, use_main(ctx)));
  DSX_HIP(hipEventSynchronize(ctx->t1));
  DSX_HIP(hipEventElapsedTime(ms, ctx->t0, ctx->t1));
  return DSX_OK;
}
int dsx_profile_enable(dsx_ctx* ctx, int on) {
  if (!ctx) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  for (auto& r : ctx->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  ctx->prof.clear();
  ctx->profiling = on != 0;
  return DSX_OK;
}
int dsx_profile_read(dsx_ctx* ctx, int max_classes, float* ms, int32_t* launches, const char** names,
                     int* n_classes) {
  if (!ctx || !ms || !launches || !n_classes) return DSX_EINVAL;
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  const int nc = std::min<int>(max_classes, KC_COUNT);
  for (int i = 0; i < nc; ++i) { ms[i] = 0.f; launches[i] = 0; if (names) names[i] = kClassNames[i]; }
  for (auto& r : ctx->prof) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess && r.cls < nc) { ms[r.cls] += t; launches[r.cls]++; }
  }
  *n_classes = nc;
  return DSX_OK;
}

/* ---- f1 / f3: Zarr brick re-tiling and the 2x2x2 pyramid level (dsx_retile.h) --------------- */
namespace {
int retile_vec(const void* a, const void* b, int W, int cx) {
  int v = 8;
  while (v > 1 && (W % v || cx % v || (uintptr_t)a % (2 * v) || (uintptr_t)b % (2 * v))) v >>= 1;
  return v;
}
int brick_args(dsx_ctx* ctx, dsx::BrickArgs& a, int Z, int H, int W, int cz, int cy, int cx, int z0) {
  if (Z <= 0 || H <= 0 || W <= 0 || cz <= 0 || cy <= 0 || cx <= 0 || z0 < 0)
    return fail(ctx, DSX_EINVAL, "brick re-tiling: shapes must be positive");
  a.Z = Z; a.H = H; a.W = W; a.cz = cz; a.cy = cy; a.cx = cx; a.z0 = z0;
  a.nbz = (z0 + Z + cz - 1) / cz; a.nby = (H + cy - 1) / cy; a.nbx = (W + cx - 1) / cx;
  if ((long long)a.nby * cy > 65535 || (long long)a.nbz * cz > 65535)
    return fail(ctx, DSX_ELIMIT, "brick re-tiling: more than 65535 rows or planes per call");
  return DSX_OK;
}
}  // namespace

int dsx_bricks_to_planes_u16(dsx_ctx* ctx, const void* d_bricks, void* d_planes, int Z, int H, int W, int cz,
                             int cy, int cx, int z0) {
  if (!ctx || !d_bricks || !d_planes) return DSX_EINVAL;
  dsx::BrickArgs a;
  if (int rc = brick_args(ctx, a, Z, H, W, cz, cy, cx, z0)) return rc;
  a.src = (const uint16_t*)d_bricks; a.dst = (uint16_t*)d_planes;
  DSX_HIP(hipSetDevice(ctx->device));
  const int v = retile_vec(d_bricks, d_planes, W, cx);
  const dim3 grid((W / v + 255) / 256, H, Z);
  switch (v) {
    case 8: hipLaunchKernelGGL(dsx::k_bricks_to_planes<8>, grid, dim3(256), 0, use_main(ctx), a); break;
    case 4: hipLaunchKernelGGL(dsx::k_bricks_to_planes<4>, grid, dim3(256), 0, use_main(ctx), a); break;
    case 2: hipLaunchKernelGGL(dsx::k_bricks_to_planes<2>, grid, dim3(256), 0, use_main(ctx), a); break;
    default: hipLaunchKernelGGL(dsx::k_bricks_to_planes<1>, grid, dim3(256), 0, use_main(ctx), a); break;
  }
  DSX_HIP(hipGetLastError());
  return DSX_OK;
}

int dsx_planes_to_bricks_u16(dsx_ctx* ctx, const void* d_planes, void* d_bricks, int Z, int H, int W, int cz,
                             int cy, int cx, int z0) {
  if (!ctx || !d_bricks || !d_planes) return DSX_EINVAL;
  dsx::BrickArgs a;
  if (int rc = brick_args(ctx, a, Z, H, W, cz, cy, cx, z0)) return rc;
  a.src = (const uint16_t*)d_planes; a.dst = (uint16_t*)d_bricks;
  DSX_HIP(hipSetDevice(ctx->device));
  const int v = retile_vec(d_bricks, d_planes, W, cx);
  const dim3 grid((a.nbx * cx / v + 255) / 256, a.nby * cy, a.nbz * cz);
  switch (v) {
    case 8: hipLaunchKernelGGL(dsx::k_planes_to_bricks<8>, grid, dim3(256), 0, use_main(ctx), a); break;
    case 4: hipLaunchKernelGGL(dsx::k_planes_to_bricks<4>, grid, dim3(256), 0, use_main(ctx), a); break;
    case 2: hipLaunchKernelGGL(dsx::k_planes_to_bricks<2>, grid, dim3(256), 0, use_main(ctx), a); break;
    default: hipLaunchKernelGGL(dsx::k_planes_to_bricks<1>, grid, dim3(256), 0, use_main(ctx), a); break;
  }
  DSX_HIP(hipGetLastError());
  return DSX_OK;
}

int dsx_downsample2_u16(dsx_ctx* ctx, const void* d_src, void* d_dst, int Z, int Y, int X) {
  if (!ctx || !d_src || !d_dst) return DSX_EINVAL;
  if (Z < 2 || Y < 2 || X < 2) return fail(ctx, DSX_EINVAL, "downsample: every axis needs at least 2 voxels");
  dsx::DownArgs a;
  a.src = (const uint16_t*)d_src; a.dst = (uint16_t*)d_dst;
  a.Z = Z; a.Y = Y; a.X = X; a.Zo = Z / 2; a.Yo = Y / 2; a.Xo = X / 2;
  if (a.Yo > 65535 || a.Zo > 65535) return fail(ctx, DSX_ELIMIT, "downsample: more than 65535 rows or planes");
  DSX_HIP(hipSetDevice(ctx->device));
  const dim3 grid(((a.Xo + 3) / 4 + 255) / 256, a.Yo, a.Zo);
  const bool vec = X % 8 == 0 && (uintptr_t)d_src % 16 == 0 && (uintptr_t)d_dst % 8 == 0;
  if (vec) hipLaunchKernelGGL(dsx::k_downsample2<true>, grid, dim3(256), 0, use_main(ctx), a);
  else hipLaunchKernelGGL(dsx::k_downsample2<false>, grid, dim3(256), 0, use_main(ctx), a);
  DSX_HIP(hipGetLastError());
  return DSX_OK;
}

int dsx_flatfield_correction(dsx_ctx* ctx, const void* d_img, int in_dtype, int H, int W, const float* d_flat,
                             const float* d_dark, int dark_h, int dark_w, float baseline, void* d_out) {
  return dsx_flatfield_correction_rows(ctx, d_img, in_dtype, H, W, d_flat, d_dark, dark_h, dark_w, baseline, nullptr, d_out);
}

int dsx_flatfield_correction_rows(dsx_ctx* ctx, const void* d_img, int in_dtype, int H, int W, const float* d_flat,
                                  const float* d_dark, int dark_h, int dark_w, float baseline,
                                  const float* d_baseline_rows, void* d_out) {
  if (!ctx || !d_img || !d_flat || !d_dark || !d_out) return DSX_EINVAL;
  if (in_dtype != DSX_U16 && in_dtype != DSX_F32) return fail(ctx, DSX_EINVAL, "unknown element type");
  if (H <= 0 || W <= 0 || H > 65535) return fail(ctx, DSX_EINVAL, "flatfield_correction: bad plane shape");
  if (dark_h < H || dark_w < W)
    return fail(ctx, DSX_EINVAL, "Please, check the shape of the darkfield (smaller than the image)");
  dsx::ShadeArgs a;
  a.src = d_img; a.flat = d_flat; a.dark = d_dark; a.dst = (uint16_t*)d_out;
  a.H = H; a.W = W; a.dark_w = dark_w; a.baseline = baseline;
  a.baseline_rows = d_baseline_rows;
  DSX_HIP(hipSetDevice(ctx->device));
  const dim3 grid((W + 255) / 256, H);
  if (in_dtype == DSX_U16) hipLaunchKernelGGL(dsx::k_shade<true>, grid, dim3(256), 0, use_main(ctx), a);
  else hipLaunchKernelGGL(dsx::k_shade<false>, grid, dim3(256), 0, use_main(ctx), a);
  DSX_HIP(hipGetLastError());
  return DSX_OK;
}

int dsx_foreground_background(dsx_ctx* ctx, const void* d_img, int in_dtype, size_t n, float cutoff,
                              double* fore_mean, double* back_mean, void* d_mask) {
  if (!ctx || !d_img || !fore_mean || !back_mean) return DSX_EINVAL;
  if (in_dtype != DSX_U16 && in_dtype != DSX_F32) return fail(ctx, DSX_EINVAL, "unknown element type");
  if (n == 0) return fail(ctx, DSX_EINVAL, "empty image");
  DSX_HIP(hipSetDevice(ctx->device));
  char* d_acc = nullptr;
  DSX_HIP(hipMalloc(&d_acc, 32));
  int rc = DSX_OK;
  do {
    if (hipMemsetAsync(d_acc, 0, 32, use_main(ctx)) != hipSuccess) { rc = DSX_EHIP; break; }
    dsx::FgBgArgs a;
    a.src = d_img; a.mask = (uint8_t*)d_mask; a.acc = (double*)d_acc; a.cnt = (unsigned long long*)(d_acc + 16);
    a.n = n; a.cutoff = cutoff;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    if (in_dtype == DSX_U16) hipLaunchKernelGGL(dsx::k_fgbg<true>, dim3(blocks), dim3(256), 0, use_main(ctx), a);
    else hipLaunchKernelGGL(dsx::k_fgbg<false>, dim3(blocks), dim3(256), 0, use_main(ctx), a);
    char h[32];
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(h, d_acc, 32, hipMemcpyDeviceToHost, use_main(ctx)) != hipSuccess ||
        hipStreamSynchronize(use_main(ctx)) != hipSuccess) { rc = DSX_EHIP; break; }
    const double* acc = (const double*)h;
    const unsigned long long* cnt = (const unsigned long long*)(h + 16);
    // an empty class has mean 0.0 (filtering.py:84-85)
    *fore_mean = cnt[0] ? acc[0] / (double)cnt[0] : 0.0;
    *back_mean = cnt[1] ? acc[1] / (double)cnt[1] : 0.0;
  } while (0);
  (void)hipFree(d_acc);
  if (rc != DSX_OK) return fail(ctx, rc, "foreground / background statistic failed");
  return DSX_OK;
}


/* ---- multi-GPU: RCCL communicator (one process per GPU) ------------------ */
namespace {
struct RcclApi {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclBroadcast) broadcast = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};
RcclApi g_rccl;

// librccl.so is loaded on first use only: single-GPU users of the library never pay for it.
int load_rccl(dsx_ctx* ctx) {
  if (ctx->rccl) return DSX_OK;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return fail(ctx, DSX_EHIP, std::string("dlopen(librccl.so): ") + dlerror());
#define DSX_SYM(field, name)                                                            \
  g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);                                \
  if (!g_rccl.field) return fail(ctx, DSX_EHIP, std::string("librccl.so lacks ") + name)
  DSX_SYM(get_unique_id, "ncclGetUniqueId");
  DSX_SYM(comm_init_rank, "ncclCommInitRank");
  DSX_SYM(comm_destroy, "ncclCommDestroy");
  DSX_SYM(broadcast, "ncclBroadcast");
  DSX_SYM(all_reduce, "ncclAllReduce");
  DSX_SYM(error_string, "ncclGetErrorString");
#undef DSX_SYM
  ctx->rccl = h;
  return DSX_OK;
}

#define DSX_NCCL(call)                                                                          \
  do {                                                                                          \
    ncclResult_t r_ = (call);                                                                   \
    if (r_ != ncclSuccess)                                                                      \
      return fail(ctx, DSX_ECOMM, std::string(#call) + ": " + g_rccl.error_string(r_));         \
  } while (0)
}  // namespace

int dsx_comm_unique_id(dsx_ctx* ctx, char* id, size_t id_bytes) {
  if (!ctx || !id) return DSX_EINVAL;
  if (id_bytes < DSX_COMM_ID_BYTES) return fail(ctx, DSX_EINVAL, "unique id buffer too small");
  static_assert(DSX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (int rc = load_rccl(ctx)) return rc;
  DSX_HIP(hipSetDevice(ctx->device));
  ncclUniqueId u;
  DSX_NCCL(g_rccl.get_unique_id(&u));
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return DSX_OK;
}

int dsx_comm_init(dsx_ctx* ctx, const char* id, size_t id_bytes, int rank, int world) {
  if (!ctx || !id) return DSX_EINVAL;
  if (id_bytes < DSX_COMM_ID_BYTES || world < 1 || rank < 0 || rank >= world)
    return fail(ctx, DSX_EINVAL, "bad unique id / rank / world size");
  if (ctx->comm) return fail(ctx, DSX_EINVAL, "communicator already initialised");
  if (int rc = load_rccl(ctx)) return rc;
  DSX_HIP(hipSetDevice(ctx->device));
  ncclUniqueId u;
  memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  DSX_NCCL(g_rccl.comm_init_rank(&ctx->comm, world, u, rank));
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  DSX_HIP(hipMalloc((void**)&ctx->d_red, sizeof(double) * 64));
  return DSX_OK;
}

int dsx_comm_destroy(dsx_ctx* ctx) {
  if (!ctx) return DSX_EINVAL;
  if (ctx->comm) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(use_main(ctx));
    (void)g_rccl.comm_destroy(ctx->comm);
    ctx->comm = nullptr;
  }
  if (ctx->d_red) { (void)hipFree(ctx->d_red); ctx->d_red = nullptr; }
  ctx->comm_rank = 0;
  ctx->comm_world = 1;
  return DSX_OK;
}

int dsx_comm_broadcast(dsx_ctx* ctx, void* d_buf, size_t bytes, int root) {
  if (!ctx || (!d_buf && bytes)) return DSX_EINVAL;
  if (!ctx->comm) return fail(ctx, DSX_ECOMM, "dsx_comm_init has not been called");
  if (root < 0 || root >= ctx->comm_world) return fail(ctx, DSX_EINVAL, "root out of range");
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_NCCL(g_rccl.broadcast(d_buf, d_buf, bytes, ncclUint8, root, ctx->comm, use_main(ctx)));
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  return DSX_OK;
}

int dsx_comm_allreduce_f64(dsx_ctx* ctx, double* values, int n, int op) {
  if (!ctx || !values || n < 1 || n > 64) return DSX_EINVAL;
  if (!ctx->comm) return fail(ctx, DSX_ECOMM, "dsx_comm_init has not been called");
  const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
  if (op < 0 || op > 2) return fail(ctx, DSX_EINVAL, "reduction: 0 sum, 1 max, 2 min");
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipMemcpyAsync(ctx->d_red, values, sizeof(double) * n, hipMemcpyHostToDevice, use_main(ctx)));
  DSX_NCCL(g_rccl.all_reduce(ctx->d_red, ctx->d_red, n, ncclDouble, ops[op], ctx->comm, use_main(ctx)));
  DSX_HIP(hipMemcpyAsync(values, ctx->d_red, sizeof(double) * n, hipMemcpyDeviceToHost, use_main(ctx)));
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  return DSX_OK;
}

/* ---- chunk files <-> staging memory on native threads (dsx_io.h) -------------------------------- */
int dsx_io_read_chunks(dsx_ctx* ctx, const char* const* paths, void* const* dst, const size_t* bytes, int n,
                       int threads, int codec, uint16_t fill_value) {
  if (n < 0 || (n > 0 && (!paths || !dst || !bytes))) return DSX_EINVAL;
  if (codec < DSX_CODEC_RAW || codec > DSX_CODEC_BLOSC) return fail(ctx, DSX_EINVAL, "unknown chunk codec");
  const std::string e = dsx::io_read_chunks(paths, dst, bytes, n, threads, codec, fill_value);
  if (!e.empty()) return fail(ctx, DSX_EIO, e);
  return DSX_OK;
}
int dsx_io_write_chunks(dsx_ctx* ctx, const char* const* paths, const void* const* src, const size_t* bytes, int n,
                        int threads, int zlib_level) {
  if (n < 0 || (n > 0 && (!paths || !src || !bytes))) return DSX_EINVAL;
  const std::string e = dsx::io_write_chunks(paths, src, bytes, n, threads, zlib_level);
  if (!e.empty()) return fail(ctx, DSX_EIO, e);
  return DSX_OK;
}

int dsx_io_write_chunks_blosc(dsx_ctx* ctx, const char* const* paths, const void* const* src, const size_t* bytes,
                              int n, int threads, int clevel, int typesize, int shuffle) {
  if (n < 0 || (n > 0 && (!paths || !src || !bytes))) return DSX_EINVAL;
  if (typesize < 1 || typesize > 255 || clevel < 0 || clevel > 9) return fail(ctx, DSX_EINVAL, "bad Blosc parameters");
  const std::string e = dsx::io_write_chunks(paths, src, bytes, n, threads, clevel, typesize, shuffle != 0);
  if (!e.empty()) return fail(ctx, DSX_EIO, e);
  return DSX_OK;
}
int dsx_blosc_decode(const void* frame, size_t frame_bytes, void* dst, size_t dst_bytes) {
  if (!frame || (!dst && dst_bytes)) return DSX_EINVAL;
  const std::string e = dsx::blosc_decode((const unsigned char*)frame, frame_bytes, dst, dst_bytes);
  if (!e.empty()) return fail(nullptr, DSX_EIO, e);
  return DSX_OK;
}
int dsx_blosc_encode(const void* src, size_t bytes, int typesize, int clevel, int shuffle, void* frame,
                     size_t frame_capacity, size_t* frame_bytes) {
  if ((!src && bytes) || !frame || !frame_bytes) return DSX_EINVAL;
  if (typesize < 1 || typesize > 255 || clevel < 0 || clevel > 9) return fail(nullptr, DSX_EINVAL, "bad Blosc parameters");
  std::vector<unsigned char> out;
  const std::string e = dsx::blosc_encode(src, bytes, typesize, clevel, shuffle != 0, out);
  if (!e.empty()) return fail(nullptr, DSX_EIO, e);
  if (out.size() > frame_capacity) return fail(nullptr, DSX_EINVAL, "frame buffer too small (bytes + 16 always fits)");
  memcpy(frame, out.data(), out.size());
  *frame_bytes = out.size();
  return DSX_OK;
}

int dsx_png_unfilter(void* rows, int height, int stride, int bytes_per_pixel) {
  if (!rows && height > 0) return DSX_EINVAL;
  const std::string e = dsx::png_unfilter((unsigned char*)rows, height, stride, bytes_per_pixel);
  if (!e.empty()) return fail(nullptr, DSX_EIO, e);
  return DSX_OK;
}

/* ---- debug hooks ---------------------------------------------------------------------------- */
int dsx_set_stack_mode(dsx_ctx* ctx, int on) {
  if (!ctx) return DSX_EINVAL;
  const bool want = on != 0;
  // the mode is baked into captured launch chains (Fwd1Args / HistArgs / OtsuArgs::shared) and is not part of the
  // graph cache's key: a graph captured under the other mode must not be replayed
  if (want != ctx->stack_mode) drop_graphs(ctx);
  ctx->stack_mode = want;
  return DSX_OK;
}
int dsx_set_stop_after(dsx_ctx* ctx, int stage) {
  if (!ctx || stage < 0 || stage > 2) return DSX_EINVAL;
  ctx->stop_after = stage;
  return DSX_OK;
}
int dsx_get_stats(dsx_ctx* ctx, int plane, double* fore_mean, double* back_mean, int32_t* cfg_used) {
  if (!ctx) return DSX_EINVAL;
  if (!ctx->planned) return fail(ctx, DSX_ENOPLAN, "dsx_plan has not been called");
  if (plane < 0 || plane >= ctx->last_n) return fail(ctx, DSX_EINVAL, "plane index outside the last cohort");
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  if (int rc = check_sticky(ctx)) return rc;
  double m[2];
  int c = 0;
  DSX_HIP(hipMemcpy(m, ctx->d_means + 2 * plane, sizeof(m), hipMemcpyDeviceToHost));
  DSX_HIP(hipMemcpy(&c, ctx->d_cfg + plane, sizeof(int), hipMemcpyDeviceToHost));
  if (fore_mean) *fore_mean = m[0];
  if (back_mean) *back_mean = m[1];
  if (cfg_used) *cfg_used = c;
  return DSX_OK;
}
int dsx_get_thresholds(dsx_ctx* ctx, int plane, int level, float* otsu, float* threshold) {
  if (!ctx) return DSX_EINVAL;
  if (!ctx->planned) return fail(ctx, DSX_ENOPLAN, "dsx_plan has not been called");
  if (plane < 0 || plane >= ctx->last_n || level < 0 || level >= ctx->plan.L)
    return fail(ctx, DSX_EINVAL, "plane / level index out of range");
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  const size_t i = (size_t)plane * ctx->plan.L + level;
  if (otsu) DSX_HIP(hipMemcpy(otsu, ctx->d_otsu + i, sizeof(float), hipMemcpyDeviceToHost));
  if (threshold) DSX_HIP(hipMemcpy(threshold, ctx->d_thr + i, sizeof(float), hipMemcpyDeviceToHost));
  return DSX_OK;
}
int dsx_get_level(dsx_ctx* ctx, int plane, int level, int stage, float* out) {
  if (!ctx || !out) return DSX_EINVAL;
  if (!ctx->planned) return fail(ctx, DSX_ENOPLAN, "dsx_plan has not been called");
  if (plane < 0 || plane >= ctx->last_n || level < 0 || level >= ctx->plan.L)
    return fail(ctx, DSX_EINVAL, "plane / level index out of range");
  DSX_HIP(hipSetDevice(ctx->device));
  DSX_HIP(hipStreamSynchronize(use_main(ctx)));
  const dsx::LevelPlan& lp = ctx->plan.lv[level];
  const long long off = (stage == DSX_STAGE_APPROX) ? lp.aa_off : lp.da_off;
  const int pitch = (stage == DSX_STAGE_APPROX) ? lp.lda : lp.ld;
  const float* src = ctx->d_ws + (size_t)plane * ctx->plan.plane_floats + off;
  DSX_HIP(hipMemcpy2D(out, sizeof(float) * lp.w, src, sizeof(float) * pitch, sizeof(float) * lp.w, lp.h,
                      hipMemcpyDeviceToHost));
  return DSX_OK;
}

}  // extern "C"
