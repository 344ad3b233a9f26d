/*
 * dsx.h  --  C ABI of the MI355X destripe engine (libdsx_hip.so).
 *
 * Drop-in boundary for ONE hot path of AllenNeuralDynamics/aind-smartspim-destripe: the per-plane
 * stripe filter.  Every entry point below replaces a piece of the reference's Python interface
 * (file:line relative to /root/reference/code/aind_smartspim_destripe/):
 *
 *   dsx_plan()        <- the arguments of filter_stripes()            filtering.py:417-424
 *                        (cells_config / no_cells_config dicts splatted into
 *                         log_space_fft_filtering(), filtering.py:139-145, 464, 467;
 *                         microscope_high_int, filtering.py:423, 462;
 *                         shadow_correction {flatfield, darkfield}, filtering.py:470-489)
 *   dsx_run_host()    <- the z-loop over planes of execute_worker()    zarr_destriper.py:319-327
 *                        and read_filter_save()                        destriper.py:194-200
 *   dsx_run_device()  <- same, operands already resident in HBM (what bench.py times)
 *   dsx_get_*()       <- per-level internals for parity tests (Otsu threshold filtering.py:190-193,
 *                        row medians filtering.py:200-202, cH / filtered cH filtering.py:186,217)
 *
 * Plain C types only; no torch / numpy types cross this boundary.  One context per GPU per
 * process; no global state; every call is synchronous unless stated.  All functions return 0
 * on success or a negative DSX_E* code; dsx_last_error() gives the message.
 *
 * Reference-side binding: see INTEGRATION.md (ctypes stub for filtering.py / zarr_destriper.py).
 */
#ifndef DSX_H
#define DSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSX_VERSION 100

/* error codes */
#define DSX_OK 0
#define DSX_EINVAL (-1)   /* bad argument (reference raises ValueError)            */
#define DSX_ENOPLAN (-2)  /* dsx_run_* before dsx_plan                              */
#define DSX_EHIP (-3)     /* HIP runtime error                                      */
#define DSX_ENOMEM (-4)   /* device or host allocation failed                       */
#define DSX_ELIMIT (-5)   /* plane exceeds an implementation limit                  */
#define DSX_ECOMM (-6)    /* RCCL error / communicator not initialised              */
#define DSX_EIO (-7)      /* chunk file could not be read / written                  */
#define DSX_EVALUE (-8)   /* a float32 pixel is NaN, infinite or <= -1: the reference raises ValueError from
                             numpy.histogram inside threshold_otsu (filtering.py:188); dsx_run_host only      */

/* plane element types */
#define DSX_U16 0 /* uint16 pixels (TIFF path, destriper.py:172-200)                 */
#define DSX_F32 1 /* float32 pixels (Zarr path, zarr_destriper.py:1049)              */

/* wavelet ids: db3 is the production setting (run_capsule.py:374-390) and has its own kernels; any other
 * wavelet the reference's config may name (pywt.wavedec2(..., wavelet=...), filtering.py:176) is handed over
 * as its filter bank with dsx_set_wavelet and selected with DSX_WAVELET_BANK                              */
#define DSX_WAVELET_DB3 3
#define DSX_WAVELET_BANK 0

/* stage buffers for dsx_get_level() */
#define DSX_STAGE_APPROX 0 /* aa_l (before the inverse pass overwrites it with c_l)   */
#define DSX_STAGE_DETAIL 1 /* da_l == cH_l, or Delta_l once the row filter has run    */

typedef struct dsx_ctx dsx_ctx;

/* One config dict of the reference: {"wavelet","level","sigma","max_threshold"}. */
typedef struct dsx_cfg {
  int32_t wavelet;     /* DSX_WAVELET_DB3 or DSX_WAVELET_BANK; both configs the same   */
  int32_t level;       /* -1 == None (maximum level), 0 == filter is the identity + 2 */
  float sigma;         /* > 0                                                         */
  float max_threshold; /* upper bound of the Otsu threshold                           */
} dsx_cfg;

/* Geometry of a planned plane (filled by dsx_plan_info). */
typedef struct dsx_plan_info_t {
  int32_t height, width;         /* input plane                                       */
  int32_t out_height, out_width; /* result plane: H + (H & 1) when any level runs     */
  int32_t levels;                /* decomposition depth actually run (max of both cfg) */
  int32_t level_h[16], level_w[16]; /* cH shape per level, index 0 == finest           */
  int32_t fft_len[16];           /* transform length used by the row filter per level  */
  int32_t fft_halo[16];          /* periodic halo K (0 == direct length-w transform)   */
  int32_t max_batch;             /* planes per cohort                                  */
  uint64_t workspace_bytes;      /* device workspace                                   */
} dsx_plan_info_t;

/* ---- context ------------------------------------------------------------------------------ */
int dsx_init(int device, dsx_ctx** out_ctx);
void dsx_destroy(dsx_ctx* ctx);
const char* dsx_last_error(const dsx_ctx* ctx); /* ctx may be NULL: last dsx_init error */
int dsx_device_count(void);

/* ---- plan: everything filter_stripes() takes besides the plane itself --------------------- */
/* flat / dark: host float32 planes (flat[H*W] row-major, dark[dark_h*dark_w] row-major, cropped
 * to the plane as flatfield_correction() does, filtering.py:377) or NULL for no shading.      */
int dsx_plan(dsx_ctx* ctx, int height, int width, int max_batch, const dsx_cfg* cells_config,
             const dsx_cfg* no_cells_config, double microscope_high_int, const float* flat,
             const float* dark, int dark_h, int dark_w);
int dsx_plan_info(const dsx_ctx* ctx, dsx_plan_info_t* info);
/* The four filters of a pywt.Wavelet (dec_lo, dec_hi, rec_lo, rec_hi; `len` taps each, even, 2 ... 104) for
 * plans whose configs carry DSX_WAVELET_BANK: what the reference passes by name to pywt.wavedec2 / waverec2
 * (filtering.py:176, 221).  Mode 'symmetric', perfect-reconstruction banks only (the engine reconstructs the
 * correction, not the plane).  Takes effect at the next dsx_plan.                                          */
int dsx_set_wavelet(dsx_ctx* ctx, const double* dec_lo, const double* dec_hi, const double* rec_lo,
                    const double* rec_hi, int len);
/* Same, with the shading planes already in device memory (e.g. after an RCCL broadcast). */
int dsx_set_shading_device(dsx_ctx* ctx, const float* d_flat, const float* d_dark, int dark_h,
                           int dark_w);
/* Device address + size of the constant blob (twiddles, gain tables, shading) so that a host
 * program can broadcast it between ranks (RCCL) instead of rebuilding it per rank.              */
int dsx_constants_device(const dsx_ctx* ctx, void** d_ptr, size_t* bytes);

/* ---- run ---------------------------------------------------------------------------------- */
/* n planes, C-order [n, H, W] in, [n, H', W'] out.  out_dtype DSX_F32: exp(y)+1 (after shading
 * if planned, before the integer cast); DSX_U16: clip to [0, 65535] and truncate.
 * cfg_used (nullable): per plane 1 == cells_config, 0 == no_cells_config (0 throughout when neither
 * config runs a decomposition level: the result is image + 2 either way and no statistic is made). */
int dsx_run_host(dsx_ctx* ctx, const void* in, int in_dtype, int n, void* out, int out_dtype,
                 int32_t* cfg_used);
/* Device pointers; asynchronous on the context stream (call dsx_sync).  d_cfg_used nullable. */
int dsx_run_device(dsx_ctx* ctx, const void* d_in, int in_dtype, int n, void* d_out,
                   int out_dtype, int32_t* d_cfg_used);
int dsx_sync(dsx_ctx* ctx);
/* With DSX_GRAPH=1 in the environment, cohorts that run as ONE part (fewer than 32 planes: the per-slice calls of
 * filter_stripes, filtering.py:417, and small batches) are replayed as a HIP graph from the third call with the
 * same buffers, count and element types on (first call eager, second captured): one graph launch instead of ~30
 * kernel launches.  Off by default: measured 5-8 % slower than eager launches on ROCm 7.2.
 * Counters for tests / diagnosis.                                                                           */
int dsx_graph_stats(const dsx_ctx* ctx, uint64_t* launches, uint64_t* captures);

/* ---- device memory + timing helpers for host programs without a GPU array library --------- */
int dsx_malloc(dsx_ctx* ctx, size_t bytes, void** d_ptr);
int dsx_free(dsx_ctx* ctx, void* d_ptr);
int dsx_memcpy_h2d(dsx_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int dsx_memcpy_d2h(dsx_ctx* ctx, void* dst, const void* d_src, size_t bytes);
int dsx_memcpy_d2d(dsx_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
/* Pinned host staging + copies on their own streams, so that a caller can overlap the upload of
 * block k+1 and the download of block k-1 with the kernels of block k -- the role the reference's
 * producer / consumer queue plays on CPU cores (zarr_destriper.py:797-906, 1138-1172).
 * stream ids: DSX_STREAM_COMPUTE is the context stream every dsx_run_device / re-tiling call uses. */
#define DSX_STREAM_COMPUTE 0
#define DSX_STREAM_UPLOAD 1
#define DSX_STREAM_DOWNLOAD 2
int dsx_malloc_host(dsx_ctx* ctx, size_t bytes, void** h_ptr);
int dsx_free_host(dsx_ctx* ctx, void* h_ptr);
int dsx_memcpy_h2d_async(dsx_ctx* ctx, void* d_dst, const void* src, size_t bytes, int stream_id);
int dsx_memcpy_d2h_async(dsx_ctx* ctx, void* dst, const void* d_src, size_t bytes, int stream_id);
/* Work submitted to `waiter` after this call starts only when everything submitted to `signaller`
 * before it has finished (event record + stream wait; nothing blocks the host).                 */
int dsx_stream_wait(dsx_ctx* ctx, int waiter, int signaller);
int dsx_stream_sync(dsx_ctx* ctx, int stream_id);
/* Host-visible completion marks, slots 0..7: record on a stream, later block the host until everything
 * submitted to that stream before the record has finished (staging buffer k may be refilled / written out). */
int dsx_event_record(dsx_ctx* ctx, int slot, int stream_id);
int dsx_event_sync(dsx_ctx* ctx, int slot);
/* HIP events on the context stream: start, stop -> elapsed milliseconds. */
int dsx_timer_start(dsx_ctx* ctx);
int dsx_timer_stop(dsx_ctx* ctx, float* ms);
/* Per-kernel-class device time of the runs since the last reset (HIP events around every
 * launch; slows the pipeline down, for diagnosis only).  names: NUL-separated list.           */
int dsx_profile_enable(dsx_ctx* ctx, int on);
int dsx_profile_read(dsx_ctx* ctx, int max_classes, float* ms, int32_t* launches,
                     const char** names, int* n_classes);

/* ---- data formats either side of the filter ------------- */
/* Zarr chunk ("brick") order <-> dense planes, uint16, device pointers, asynchronous on the
 * context stream.  Replaces the NumPy gather of (1,1,64,128,128) chunks into a block and the
 * scatter of the filtered block back into chunks (zarr_destriper.py:1066-1074 and :336).
 * d_bricks: [nbz][nby][nbx][cz][cy][cx], every brick full-sized as zarr stores it
 * (nb* = ceil over the axis; nbz = ceil((z0 + Z) / cz)); d_planes: dense [Z][H][W]; z0: position
 * of plane 0 inside the brick grid.  planes_to_bricks writes 0 (the fill value) where a brick
 * sticks out of the stack.                                                                    */
int dsx_bricks_to_planes_u16(dsx_ctx* ctx, const void* d_bricks, void* d_planes, int Z, int H,
                             int W, int cz, int cy, int cx, int z0);
int dsx_planes_to_bricks_u16(dsx_ctx* ctx, const void* d_planes, void* d_bricks, int Z, int H,
                             int W, int cz, int cy, int cx, int z0);
/* One 2x2x2 windowed-mean pyramid level, uint16 [Z,Y,X] -> [Z/2,Y/2,X/2] (odd trailing voxels are
 * cropped), value = floor(sum of 8 / 8): compute_pyramid(), zarr_destriper.py:365-407, i.e.
 * xarray_multiscale.reducers.windowed_mean + preserve_dtype.  Asynchronous on the context stream. */
int dsx_downsample2_u16(dsx_ctx* ctx, const void* d_src, void* d_dst, int Z, int Y, int X);

/* Host side of the chunk map: n Zarr chunk files <-> memory (normally the pinned staging buffers) on
 * `threads` native threads -- what zarr / numcodecs do under the reference's worker processes
 * (zarr_destriper.py:336, 1042-1074).  codec: raw chunks, zlib streams, or Blosc frames -- the production
 * arrays are Blosc(cname="zstd", clevel=3, shuffle=SHUFFLE) (zarr_destriper.py:1066-1074); the c-blosc 1.x
 * container is restated in csrc/dsx_io.h (zstd / lz4 / blosclz / zlib inside, byte or bit shuffle; libzstd.so.1
 * and liblz4.so.1 are dlopen'ed), pinned by frames of the real c-blosc 1.21.0 (tests/golden/blosc_frames.npz) and by
 * decoding this writer's frames with that library (tests/test_blosc.py).  A missing chunk reads as the
 * 16-bit fill value; writes go to "<path>.tmp" and are renamed.  bytes[i] is the decompressed chunk size.
 * Synchronous; no GPU involved (ctx may be NULL: the message of a failure is then read with
 * dsx_last_error(NULL)).                                                                              */
#define DSX_CODEC_RAW 0
#define DSX_CODEC_ZLIB 1
#define DSX_CODEC_BLOSC 2
int dsx_io_read_chunks(dsx_ctx* ctx, const char* const* paths, void* const* dst, const size_t* bytes,
                       int n, int threads, int codec, uint16_t fill_value);
/* zlib_level < 0: raw chunks, otherwise zlib streams of that level */
int dsx_io_write_chunks(dsx_ctx* ctx, const char* const* paths, const void* const* src,
                        const size_t* bytes, int n, int threads, int zlib_level);
/* Blosc frames with zstd inside: clevel 0 ... 9 (Blosc's scale), typesize = element size, shuffle 0 / 1 */
int dsx_io_write_chunks_blosc(dsx_ctx* ctx, const char* const* paths, const void* const* src,
                              const size_t* bytes, int n, int threads, int clevel, int typesize, int shuffle);
/* One frame in memory (what numcodecs.Blosc.decode / .encode do for one chunk); errors: dsx_last_error(NULL).
 * A frame never exceeds bytes + 16.                                                                    */
int dsx_blosc_decode(const void* frame, size_t frame_bytes, void* dst, size_t dst_bytes);
int dsx_blosc_encode(const void* src, size_t bytes, int typesize, int clevel, int shuffle, void* frame,
                     size_t frame_capacity, size_t* frame_bytes);

/* PNG scanline reconstruction for the directory mode's reader (imageio's iio.imread, readers.py:86-87): `height`
 * rows of one filter-type byte + `stride` bytes, un-filtered in place (Sub / Up / Average / Paeth).  Host only.  */
int dsx_png_unfilter(void* rows, int height, int stride, int bytes_per_pixel);

/* flatfield_correction() of one plane as a stand-alone call (filtering.py:338-414): dark subtraction
 * (integer planes truncate, :400-403), division by the flat, baseline, clip, uint16.  dark is
 * [dark_h][dark_w] >= the plane and is cropped to it (:377).  Device pointers, asynchronous.   */
/* Stack mode = the 3-D input mode of log_space_fft_filtering() (filtering.py:182-183, 188, 210-211): the planes of
 * ONE dsx_run_* call are a stack that shares one Otsu threshold per decomposition level (min / max of cH^2 and the
 * 256-bin histogram are taken over all planes; row medians and the FFT stay per plane row).  The whole stack must
 * fit one cohort (n <= max_batch of dsx_plan, else DSX_ELIMIT).  Off by default: planes are independent.       */
int dsx_set_stack_mode(dsx_ctx* ctx, int on);

int dsx_flatfield_correction(dsx_ctx* ctx, const void* d_img, int in_dtype, int H, int W,
                             const float* d_flat, const float* d_dark, int dark_h, int dark_w,
                             float baseline, void* d_out);
/* The same with one baseline value per plane ROW (d_baseline_rows: H floats on the device, NULL = the scalar): what
 * the reference's broadcast baseline[:, np.newaxis] does for a 2-D plane and a baseline of length H
 * (filtering.py:393-398, 409).                                                                              */
int dsx_flatfield_correction_rows(dsx_ctx* ctx, const void* d_img, int in_dtype, int H, int W,
                                  const float* d_flat, const float* d_dark, int dark_h, int dark_w,
                                  float baseline, const float* d_baseline_rows, void* d_out);

/* get_foreground_background_mean() as a stand-alone call (filtering.py:54-88): a pixel is foreground
 * when float16(pixel) >= cutoff (the host derives cutoff from threshold_mask, 383.25 for the default
 * 0.3); means in double, 0.0 for an empty class; d_mask (nullable) gets 1 / 0 per pixel.
 * Synchronous.  Inside dsx_run_* the same statistic is fused into the first analysis kernel.      */
int dsx_foreground_background(dsx_ctx* ctx, const void* d_img, int in_dtype, size_t n, float cutoff,
                              double* fore_mean, double* back_mean, void* d_mask);

/* ---- multi-GPU: one process per GPU, z-ranges per rank, ONE collective ---- */
/* The reference parallelises over chunks with OS processes and a queue (zarr_destriper.py:1138-1172)
 * and never communicates between workers; planes are independent (:319-327).  Here the only exchange
 * is a broadcast (root 0) of the constant blob / shading planes before the data path, straight on
 * RCCL over xGMI (librccl.so is dlopen'ed by the first dsx_comm_* call; no torch).  Rank 0 creates
 * the 128-byte unique id and hands it to the other ranks through any host channel (file, env,
 * socket: aind_smartspim_destripe_amd/distributed.py has a file rendezvous), then every rank calls
 * dsx_comm_init -- which is collective.  Broadcast / all-reduce run on the context stream and are
 * synchronous; an all-reduce doubles as a barrier.                                               */
#define DSX_COMM_ID_BYTES 128
int dsx_comm_unique_id(dsx_ctx* ctx, char* id, size_t id_bytes);
int dsx_comm_init(dsx_ctx* ctx, const char* id, size_t id_bytes, int rank, int world);
int dsx_comm_destroy(dsx_ctx* ctx);
/* In place on device memory: root's bytes replace everybody else's. */
int dsx_comm_broadcast(dsx_ctx* ctx, void* d_buf, size_t bytes, int root);
/* Host doubles (n <= 64) reduced over the ranks: op 0 = sum, 1 = max, 2 = min. */
int dsx_comm_allreduce_f64(dsx_ctx* ctx, double* values, int n, int op);

/* ---- parity / debug hooks (state of the LAST cohort of the last run) ----------------------- */
/* Per plane of the last cohort: fore/back means and chosen config (filtering.py:459-462). */
int dsx_get_stats(dsx_ctx* ctx, int plane, double* fore_mean, double* back_mean,
                  int32_t* cfg_used);
/* Per plane and level (0 == finest): Otsu value (of cH^2) and the threshold actually used. */
int dsx_get_thresholds(dsx_ctx* ctx, int plane, int level, float* otsu, float* threshold);
/* Copy a level buffer [h, w] (float32, dense) of a plane to the host. */
int dsx_get_level(dsx_ctx* ctx, int plane, int level, int stage, float* out);
/* Stop the pipeline after a stage (0 == run everything, 1 == after the forward transform +
 * thresholds, 2 == after the row filter) so that dsx_get_level can read cH / Delta.           */
int dsx_set_stop_after(dsx_ctx* ctx, int stage);

#ifdef __cplusplus
}
#endif
#endif /* DSX_H */
