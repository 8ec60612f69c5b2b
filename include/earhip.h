/* earhip.h — C ABI of the MI355X-native ADM render DSP path.
 *
 * This is the drop-in boundary: a plain-C shared library (libearhip.so) whose
 * entry points are what a binding for libear's DSP hot path would call.  Each
 * group below names the libear interface it replaces (paths relative to the
 * libear tree).  The C++14 classes in libear_amd/host/ear/dsp/ wrap these entry
 * points behind libear's own class names and signatures and map the status
 * codes back to libear's exception types.
 *
 * Conventions (libear's, include/ear/dsp/ptr_adapter.hpp:10-40):
 *   - audio is planar float32: `const float *const *in` is an array of channel
 *     pointers, each to contiguous samples;
 *   - the caller owns every buffer; the library reads/writes host pointers only
 *     during the call and never retains them;
 *   - `*_device` entry points take device pointers in the same planar layout
 *     (channel c at base + c * stride) and enqueue on the context's stream
 *     without synchronising;
 *   - no entry point allocates host or device memory in a `process` call: buffers
 *     are made at create and where curves are committed (earhip_render_commit; a
 *     process call that finds uncommitted curves commits them first).
 *
 * Errors: every function returns an int status.  Nothing throws across this
 * boundary.  earhip_last_error() returns the message of the calling thread's
 * last failure.
 */
#ifndef EARHIP_H
#define EARHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EARHIP_VERSION 100 /* 0.1.0 */

/* status codes; the C++ shim maps 1 -> ear::invalid_argument, 2,3 -> ear::internal_error,
 * 4 -> ear::not_implemented, 5 -> ear::unknown_layout, 6 -> ear::adm_error
 * (include/ear/exceptions.hpp:8-43) */
#define EARHIP_OK 0
#define EARHIP_INVALID_ARGUMENT 1
#define EARHIP_INTERNAL_ERROR 2
#define EARHIP_DEVICE_ERROR 3
#define EARHIP_NOT_IMPLEMENTED 4 /* ear::not_implemented: a case libear itself refuses */
#define EARHIP_UNKNOWN_LAYOUT 5  /* ear::unknown_layout (an invalid-argument kind: src/bs2051.cpp:19) */
#define EARHIP_ADM_ERROR 6       /* ear::adm_error: invalid ADM metadata (an invalid-argument kind) */

int earhip_version(void);
const char *earhip_last_error(void);
int earhip_device_count(int *count);

/* ------------------------------------------------------------------------
 * Context: one HIP device + one stream.  Not thread-safe (libear's stateful
 * DSP objects are single-owner as well).
 * ---------------------------------------------------------------------- */
typedef struct earhip_ctx earhip_ctx;

/* hip_stream: a hipStream_t to enqueue on (e.g. the caller's current stream),
 * or NULL to let the context create its own. */
int earhip_ctx_create(int device, void *hip_stream, earhip_ctx **out);
int earhip_ctx_destroy(earhip_ctx *ctx);
int earhip_ctx_synchronize(earhip_ctx *ctx);
/* strict != 0: gain kernels reproduce libear's float arithmetic exactly
 * (un-contracted `in * ((1-p)*s + p*e)`, inputs accumulated in channel order),
 * so M->N results are bit-identical to the CPU path.  Default 0: fused
 * multiply-adds and tree accumulation (faster, within 1e-6 relative RMS). */
int earhip_ctx_set_strict(earhip_ctx *ctx, int strict);
/* Run-time configuration (libear has none: its only knob is the FFT plugin; SURVEY 5 asks for "a small runtime config").
 * Every tuning knob is an option of the context: key = one of the names below (case-insensitive, an "EARHIP_" prefix is
 * accepted), value = a decimal integer as text, or NULL / "" to return the option to its default ("the library decides").
 * At earhip_ctx_create every option is taken ONCE from the environment variable EARHIP_<KEY>; after that only this call
 * changes it — no process call reads the environment.  An option takes effect for calls / objects made after it is
 * set.  Unknown key, or a value that is not an integer ("abc", "1x"): EARHIP_INVALID_ARGUMENT, nothing changed (a context
 * is not created from an environment that holds such a value).
 *   gain stage:   MFMA (0 VALU | 1 exact-f32 MFMA | 3 default | 4 grid kernel | 5 piece lists | 6 hinge kernel),
 *                 H2_TILE, P2_TILE, HG_TILE (256 | 512), P2_PAIRS, HINGE (0 | 1), H2_WGS, P2_WGS (workgroups of the launch;
 *                 P2_WGS 0: one per tile), H2_PAIR (1: the 8-wave grid kernel's plain and wide forms as a pair of launches instead of one kernel that
 *                 branches on the device's mode word: rounds 1-5), H2_RUNS (1: a workgroup of the grid kernel takes a contiguous run of tiles instead of
 *                 every n-th: measured slower, kept as a knob), HG_ROBUST (default 1: a call whose levels spread beyond the hinge
 *                 kernel's packed-f16 kink products runs that kernel's f32 form; 0: it is handed to the piece lists standing by,
 *                 rounds 4-5), BUILD_TPW, HBUILD_TPW (1 2 4 8), BUILD_2K (1: the list builders as two kernels — classify every (object,
 *                 tile) pair object-major into a staging matrix, place tile-major: the same lists bit for bit; 0: one pass; default:
 *                 two kernels for the paired piece lists only, where they are faster),
 *                 SPL (2 | 4), WAVES (1..8), TPW (1..8), NRT (4 | 8), XSCALE (log2 of a fixed input prescale), PROBE_RUNS
 *   renderer:     K2_WG, K2_OWN_BLOCK (0 | 1), RUN (odd blocks per decorrelator run), GSPLIT (1..32) — read by
 *                 earhip_render_create; TAILCUT (v = 0..7, default 2: a stream call of k rounds of tiles plus at most v / 8
 *                 of a round runs as two consecutive calls, earhip_render_last_tail_blocks; longer tails measured slower cut)
 *   host pointers: HOST_CHUNK_MB (MB of inputs per time chunk of a long earhip_render_process call: default 32 from device-reachable
 *                 rows, 16 from ordinary ones; <= 0: no pipeline, one transfer), HOST_THREADS (staging threads: default min(8, the CPUs the process may use — affinity mask, cgroup quota — less 2)),
 *                 HOST_BIND (1: the staging threads run on the NUMA node that holds the caller's rows — found with move_pages(2), the
 *                 node's CPUs from /sys; a thread remote to both the rows and the pinned staging buffer gathers at 38 GB/s where any
 *                 other placement reaches 46-48: tools/host_stream_numa.py; default 0: the scheduler's placement, 1.4 % faster
 *                 where it is good), HOST_NT (default 1: the gather writes the staging buffer with streaming stores; 0: memcpy),
 *                 HOST_FIRST (1: the first chunk of staged rows a quarter of the others; measured level, default 0)
 *   diagnostics:  BLOCK_GROUPS, DEBUG_TIMING
 * (K2_WG, K2_OWN_BLOCK, DEBUG_TIMING are "on" for any value other than 0 — rounds 1-4 read the mere presence of the environment
 * variable as "on": EARHIP_K2_WG=0 now means off.) */
int earhip_ctx_set_option(earhip_ctx *ctx, const char *key, const char *value);
int earhip_ctx_get_option(const earhip_ctx *ctx, const char *key, int *is_set, int *value);
/* Host memory the device reaches directly (pinned and mapped).  libear's interfaces take `float **` channel
 * pointers into the caller's own memory (include/ear/dsp/ptr_adapter.hpp:17-24: the columns of a matrix);
 * when those buffers come from earhip_host_alloc, or were registered once with earhip_host_register
 * (hipHostRegister: the range must stay allocated until released), AND the channel pointers of a call are
 * evenly spaced — a column-major matrix — the short-call path of earhip_render_process copies them with
 * one strided DMA instead of gathering them into a staging buffer first, and writes the outputs in place
 * (block mode: 0.12 -> 0.09 ms per 512-sample call at the headline shape); long calls (16 MB of inputs and more) move
 * their time chunks by strided DMA in both directions (0.91-0.95 of the bus's own rate, against 0.83-0.88 through the staging
 * threads).  Any other pointers work as before.  earhip_host_release frees / unregisters a range (by its start);
 * earhip_ctx_destroy the rest. */
int earhip_host_alloc(earhip_ctx *ctx, size_t bytes, void **out);
int earhip_host_register(earhip_ctx *ctx, void *ptr, size_t bytes);
int earhip_host_release(earhip_ctx *ctx, void *ptr);
/* tuning aid: enqueue a 1-thread kernel that writes {shader-cycle counter,
 * constant-rate counter} (2 x uint64) to device memory */
int earhip_debug_clock_probe(earhip_ctx *ctx, void *out_dev);
/* measurement aid (bench.py's `roofline.peak_measured`): average duration in ms, over `reps`
 * launches timed with HIP events on the context's stream, of two kernels that only READ
 * in_dev [rows][stride]: ms[0] a linear stream over rows * stride floats, ms[1] the gain stage's
 * access pattern (each workgroup a 1 KB piece of every row) over rows * nsamples floats.
 * stride and nsamples: multiples of 256, nsamples <= stride; in_dev 16-byte aligned. */
int earhip_debug_read_bandwidth(earhip_ctx *ctx, const float *in_dev, size_t rows, size_t stride,
                                size_t nsamples, int reps, double ms[2]);
/* measurement aid (bench.py's `host_stream`): what the bus gives a plain copy of `bytes` bytes between `host` (pinned or
 * pageable, the caller's) and a device buffer of the call's own — average ms over `reps` copies of ms[0] host -> device and
 * ms[1] device -> host, each between HIP events on the context's stream: the 100 % mark of the host-pointer entry points. */
int earhip_debug_copy_bandwidth(earhip_ctx *ctx, void *host, size_t bytes, int reps, double ms[2]);
/* tuning aid, diagnostic builds (-DEARHIP_K2_PROF) only: the shader-clock stamps wave 0 of two workgroups of the last decorrelator
 * launch left at its phase boundaries, out64[2][32] (tools/k2_phases.py); an ordinary build answers EARHIP_INVALID_ARGUMENT */
int earhip_debug_k2_prof(earhip_ctx *ctx, unsigned long long *out64);
/* the same for the list builders (k_piece_build / k_hinge_build; -DEARHIP_BUILD_PROF; tools/build_phases.py) */
int earhip_debug_build_prof(earhip_ctx *ctx, unsigned long long *out64);
/* the same for the hinge kernel (k_gain_mix_hg; -DEARHIP_HG_PROF; tools/hg_phases.py): per wave of two workgroups the cycles summed
 * over its chunk loop's phases, out128[2][8][8] */
int earhip_debug_hg_prof(earhip_ctx *ctx, unsigned long long *out128);

/* ------------------------------------------------------------------------
 * (A) Interpolation policies — replaces LinearInterpSingle / LinearInterpVector
 * / LinearInterpMatrix ::apply_interp and ::apply_constant
 * (include/ear/dsp/gain_interpolator.hpp:187-208, 214-241, 250-299).
 * n_in x n_out = 1x1 (Single), 1xN (Vector), MxN (Matrix).  Points are dense
 * row-major [n_in][n_out] (libear's Matrix point is vector<vector<float>>
 * indexed [in][out], gain_interpolator.hpp:247).  Writes out[o][range_start ..
 * range_end) for every output o; in and out must not alias.
 * ---------------------------------------------------------------------- */
int earhip_interp_apply_interp(earhip_ctx *ctx, int n_in, int n_out,
                               const float *const *in, float *const *out,
                               int64_t range_start, int64_t range_end,
                               int64_t block_start, int64_t start, int64_t end,
                               const float *start_point, const float *end_point);
int earhip_interp_apply_constant(earhip_ctx *ctx, int n_in, int n_out,
                                 const float *const *in, float *const *out,
                                 int64_t range_start, int64_t range_end,
                                 const float *point);

/* ------------------------------------------------------------------------
 * (A') Whole-curve GainInterpolator with device-resident points — replaces
 * GainInterpolator<InterpType>::process (gain_interpolator.hpp:53-87) for
 * callers that keep the curve fixed across calls.  values: [npoints][n_in]
 * [n_out]; times must be sorted (duplicates = step).
 * ---------------------------------------------------------------------- */
typedef struct earhip_gain_interp earhip_gain_interp;
int earhip_gain_interp_create(earhip_ctx *ctx, int n_in, int n_out,
                              earhip_gain_interp **out);
int earhip_gain_interp_destroy(earhip_gain_interp *gi);
int earhip_gain_interp_set_points(earhip_gain_interp *gi, int npoints,
                                  const int64_t *times, const float *values);
int earhip_gain_interp_process(earhip_gain_interp *gi, int64_t block_start,
                               size_t nsamples, const float *const *in,
                               float *const *out);
/* in_dev: [n_in][in_stride], out_dev: [n_out][out_stride], device memory */
int earhip_gain_interp_process_device(earhip_gain_interp *gi, int64_t block_start,
                                      size_t nsamples, const float *in_dev,
                                      size_t in_stride, float *out_dev,
                                      size_t out_stride);

/* ------------------------------------------------------------------------
 * (B) FFT plugin — an r2c/c2r transform with libear's FFTPlan contract
 * (include/ear/fft.hpp:27-50): forward n_fft reals -> n_fft/2+1 unpacked
 * complex bins; reverse the inverse; both un-normalised.  Like libear's kissfft
 * plan (src/fft_kiss.cpp:104-107) any even n_fft is taken — here up to 8192
 * (mixed radix 4/2/3/5 + kissfft's generic butterfly for every other prime; the
 * powers of two from 64 have their own kernels).  Cost: a prime factor p > 5 costs
 * n_fft * p complex multiply-adds per transform, as in kissfft — n_fft = 2 * 4093
 * is ~33 M of them, thousands of times a neighbouring size: correct, not fast.
 * Host pointers.
 * ---------------------------------------------------------------------- */
typedef struct earhip_fft_plan earhip_fft_plan;
int earhip_fft_plan_create(earhip_ctx *ctx, size_t n_fft, earhip_fft_plan **out);
int earhip_fft_plan_destroy(earhip_fft_plan *plan);
int earhip_fft_forward(earhip_fft_plan *plan, const float *in, float *out_complex);
int earhip_fft_reverse(earhip_fft_plan *plan, const float *in_complex, float *out);

/* ------------------------------------------------------------------------
 * (C) BlockConvolver — replaces ear::dsp::block_convolver::{Context, Filter,
 * BlockConvolver} (include/ear/dsp/block_convolver.hpp:28-112; behaviour of
 * src/dsp/block_convolver_impl.cpp:10-243).  block_size in [1, 4096], any
 * factorisation (480, 960, 1920, 441, primes ... as well as the powers of two;
 * large prime factors at the cost stated under (B)).
 *
 * libear's Context takes an FFTImpl<float> plugin by reference (block_convolver.hpp:34,
 * include/ear/fft.hpp:54-62) and runs ITS transforms.  Here the transform is part of
 * the device convolver (earhip_conv_ctx_create takes no plugin); the C++ mirror's
 * Context(size_t, FFTImpl<float>&) accepts the plugin that is this transform —
 * ear::get_fft_hip(), which ear::get_fft_kiss<float>() also returns — and REFUSES any
 * other plugin with ear::invalid_argument at construction: a caller's host FFT cannot
 * run inside a device kernel, and routing the convolver through host transforms would
 * be the CPU path this library does not have (tests/cpp/test_dropin.cpp,
 * "foreign FFTImpl is refused").  Callers with their own FFTImpl keep libear's
 * BlockConvolver for that object; results agree to the convolver tests' 1e-6.
 * ---------------------------------------------------------------------- */
typedef struct earhip_conv_ctx earhip_conv_ctx;
typedef struct earhip_conv_filter earhip_conv_filter;
typedef struct earhip_conv earhip_conv;

int earhip_conv_ctx_create(earhip_ctx *ctx, size_t block_size,
                           earhip_conv_ctx **out);
int earhip_conv_ctx_destroy(earhip_conv_ctx *cctx);
int earhip_conv_filter_create(earhip_conv_ctx *cctx, size_t n, const float *taps,
                              earhip_conv_filter **out);
/* Drops the creator's reference.  A convolver keeps its own reference for every queue slot
 * that points at the filter (libear's queue holds shared_ptrs,
 * src/dsp/block_convolver_impl.hpp:154-167), so a filter may be destroyed while it is still
 * in use or fading out; it is freed when the last slot lets go of it. */
int earhip_conv_filter_destroy(earhip_conv_filter *filter);
size_t earhip_conv_filter_num_blocks(const earhip_conv_filter *filter);
/* filter may be NULL (then num_blocks must be > 0); num_blocks 0 = take the
 * partition count from the filter (block_convolver.hpp:67-77) */
int earhip_conv_create(earhip_conv_ctx *cctx, const earhip_conv_filter *filter,
                       size_t num_blocks, earhip_conv **out);
int earhip_conv_destroy(earhip_conv *conv);
/* in may be NULL = a block of silence (block_convolver_impl.cpp:145-147,156) */
int earhip_conv_process(earhip_conv *conv, const float *in, float *out);
/* filter NULL = fade_down() / unset_filter() (src/dsp/block_convolver.cpp:27-37) */
int earhip_conv_crossfade_filter(earhip_conv *conv, const earhip_conv_filter *filter);
int earhip_conv_set_filter(earhip_conv *conv, const earhip_conv_filter *filter);

/* ------------------------------------------------------------------------
 * (D) DelayBuffer — replaces ear::dsp::DelayBuffer
 * (include/ear/dsp/delay_buffer.hpp:12-30, src/dsp/delay_buffer_impl.cpp:19-43)
 * ---------------------------------------------------------------------- */
typedef struct earhip_delay earhip_delay;
int earhip_delay_create(earhip_ctx *ctx, size_t nchannels, size_t nsamples,
                        earhip_delay **out);
int earhip_delay_destroy(earhip_delay *d);
int earhip_delay_process(earhip_delay *d, size_t nsamples, const float *const *in,
                         float *const *out);
int earhip_delay_get_delay(const earhip_delay *d);

/* ------------------------------------------------------------------------
 * (E) VariableBlockSizeAdapter — replaces ear::dsp::VariableBlockSizeAdapter
 * (include/ear/dsp/variable_block_size.hpp:17-40,
 * src/dsp/variable_block_size_impl.cpp:27-85).  Host-side FIFO re-blocking
 * around a user callback; adds block_size samples of delay.
 * ---------------------------------------------------------------------- */
typedef struct earhip_vbs earhip_vbs;
typedef int (*earhip_process_func)(const float *const *in, float *const *out,
                                   void *user); /* returns an earhip status */
int earhip_vbs_create(size_t block_size, size_t num_channels_in,
                      size_t num_channels_out, earhip_process_func process_func,
                      void *user, earhip_vbs **out);
/* the same with the adapter's two FIFO buffers in device-reachable host memory of `ctx` (earhip_host_alloc): a
 * renderer called from process_func then takes its no-staging path (the FIFO rows are evenly spaced channel
 * buffers).  Destroy the adapter before the context. */
int earhip_vbs_create_pinned(earhip_ctx *ctx, size_t block_size, size_t num_channels_in,
                             size_t num_channels_out, earhip_process_func process_func, void *user,
                             earhip_vbs **out);
int earhip_vbs_destroy(earhip_vbs *v);
int earhip_vbs_process(earhip_vbs *v, size_t nsamples, const float *const *in,
                       float *const *out);
int earhip_vbs_get_delay(const earhip_vbs *v);

/* ------------------------------------------------------------------------
 * (G) Decorrelator design (setup path, host code) — replaces designDecorrelators
 * <float> and decorrelatorCompensationDelay (include/ear/decorrelate.hpp:16-34,
 * src/decorrelate.cpp:31-97).  channel_names: the layout's channel names in
 * layout order (the filter id of a channel is the rank of its name);
 * out: [n_channels][512].
 * ---------------------------------------------------------------------- */
int earhip_decorrelator_size(void);               /* 512 */
int earhip_decorrelator_compensation_delay(void); /* 255 */
int earhip_design_decorrelator_basic(int decorrelator_id, int size, double *out);
int earhip_design_decorrelators(int n_channels, const char *const *channel_names, float *out);

/* ------------------------------------------------------------------------
 * (H) ITU-R BS.2051 loudspeaker layouts (setup path, host data) — replaces
 * ear::loadLayouts / ear::getLayout (include/ear/bs2051.hpp:8-11, src/bs2051.cpp:11-22,
 * table src/bs2051_layouts.cpp): what the render path needs of a Layout — channel
 * names in layout order (decorrelator ids), nominal positions (the gain producers) and
 * the LFE flags (zero-gain columns; Layout::withoutLfe, include/ear/layout.hpp).
 * An unknown layout name is EARHIP_UNKNOWN_LAYOUT (libear throws unknown_layout).
 * ---------------------------------------------------------------------- */
int earhip_layout_count(void);
const char *earhip_layout_name(int index); /* NULL when out of range */
int earhip_layout_num_channels(const char *layout, int *n_channels);
/* any output pointer may be NULL; *name points at static storage */
int earhip_layout_channel(const char *layout, int index, const char **name, double *azimuth,
                          double *elevation, int *is_lfe);
/* the ranges BS.2051 allows a real loudspeaker of this channel (Channel::azimuthRange / elevationRange,
 * include/ear/layout.hpp:41-42, table src/bs2051_layouts.cpp): azimuth from [0] anticlockwise to [1],
 * elevation from [0] up to [1], degrees; either pointer may be NULL */
int earhip_layout_channel_ranges(const char *layout, int index, double azimuth_range[2],
                                 double elevation_range[2]);
/* designDecorrelators<float>(getLayout(layout)) — or of getLayout(layout).withoutLfe() —
 * out: [channels kept][512] (include/ear/decorrelate.hpp:26-28) */
int earhip_design_decorrelators_for_layout(const char *layout, int without_lfe, float *out);

/* ------------------------------------------------------------------------
 * (I) Gain-vector producer for Objects content — replaces ear::GainCalculatorObjects
 * (include/ear/gain_calculators.hpp:45-56, src/object_based/gain_calculator_objects.cpp:24-57)
 * for point sources: the polar point-source panner (src/common/point_source_panner.cpp:
 * triplets, quads, virtual n-gons, extra height loudspeakers and their downmix, the 0+2+0
 * stereo downmix), zero gains on the LFE channels, gain, and the sqrt(1 - diffuse) /
 * sqrt(diffuse) split into the direct and diffuse vectors that feed (F).  A BATCH of
 * positions per call (one device thread each, double precision): what a renderer needs
 * per (object, metadata block).  Not implemented, as in libear's own calculate(): Cartesian
 * positions, divergence, channel lock, zone exclusion, screen scaling.
 * The _extent forms add libear's polar extent panner (src/object_based/polar_extent.cpp:
 * 12-302 with its core, polar_extent_scalar.cpp:25-108): width and height in degrees, depth in
 * distance units, each may be NULL (0); one wave per position sums the point source panner's
 * gains over 1652 points on the sphere weighted by the extent's shape (float, as in libear).
 * The float sums run in a different order from libear's scalar core: results agree to 1e-5 of
 * a gain vector's norm, the bound libear's tests put between its own cores
 * (tests/extent_tests.cpp:140-169).  With all three NULL they are the plain forms.
 * layout: an ITU-R BS.2051 name (group H).  Positions are polar: azimuth, elevation in
 * degrees (ADM convention), distance; distance / gain / diffuse may be NULL (1, 1, 0).
 * direct, diffuse_out: [npos][n_channels] float, LFE columns zero.
 * ---------------------------------------------------------------------- */
typedef struct earhip_panner earhip_panner;
int earhip_panner_create(earhip_ctx *ctx, const char *layout, earhip_panner **out);
/* the same for loudspeakers that do not stand at their nominal positions (Channel::polarPosition,
 * include/ear/layout.hpp:32-40): azimuth / elevation in degrees, one per channel of the full layout
 * (LFE channels included, ignored); both NULL (n_channels 0): nominal.  As in libear the layer logic
 * and the triangulation follow the nominal positions and the geometry the real ones
 * (point_source_panner.cpp:256-349, :431-476); M+SC / M-SC outside 5..25 or 35..60 degrees is
 * EARHIP_INVALID_ARGUMENT, wider than 25 degrees EARHIP_NOT_IMPLEMENTED (:558-577). */
int earhip_panner_create_positions(earhip_ctx *ctx, const char *layout, int n_channels,
                                   const double *azimuth, const double *elevation,
                                   earhip_panner **out);
int earhip_panner_destroy(earhip_panner *p);
int earhip_panner_num_channels(const earhip_panner *p, int *n_channels);
/* host pointers: H2D, kernel, D2H, synchronise */
int earhip_panner_calculate(earhip_panner *p, size_t npos, const double *azimuth,
                            const double *elevation, const double *distance, const double *gain,
                            const double *diffuse, float *direct, float *diffuse_out);
/* device pointers: enqueues on the context's stream, does not synchronise */
int earhip_panner_calculate_device(earhip_panner *p, size_t npos, const double *azimuth,
                                   const double *elevation, const double *distance,
                                   const double *gain, const double *diffuse, float *direct,
                                   float *diffuse_out);

int earhip_panner_calculate_extent(earhip_panner *p, size_t npos, const double *azimuth,
                                   const double *elevation, const double *distance,
                                   const double *width, const double *height, const double *depth,
                                   const double *gain, const double *diffuse, float *direct,
                                   float *diffuse_out);
int earhip_panner_calculate_extent_device(earhip_panner *p, size_t npos, const double *azimuth,
                                          const double *elevation, const double *distance,
                                          const double *width, const double *height,
                                          const double *depth, const double *gain,
                                          const double *diffuse, float *direct, float *diffuse_out);
/* positions of the LAST *_device call that no region of the layout took (libear dereferences an empty
 * optional there, src/object_based/gain_calculator_objects.cpp:46; the host-pointer forms turn it into
 * EARHIP_INTERNAL_ERROR themselves): their gain rows are zero.  Synchronises the stream. */
int earhip_panner_missed(earhip_panner *p, unsigned *count);

/* (I, HOA) Decode matrix for scene-based (HOA) content — replaces ear::GainCalculatorHOA
 * (include/ear/gain_calculators.hpp:58-70, src/hoa/gain_calculator_hoa.cpp:8-72,
 * src/hoa/hoa.hpp:16-182): the AllRAD design over the layout's point source panner.  One
 * (order, degree) pair per input channel; normalization "SN3D", "N3D" or "FuMa" (an unknown
 * one is EARHIP_ADM_ERROR: libear throws adm_error); out: [n_channels][n_coef],
 * rows of LFE channels zero.  It is constant over time: feed its COLUMNS to (F) or (A')
 * as single-point gain curves (docs/dsp.rst:73-89).  screenRef and nfcRefDist are ignored
 * by libear (with a warning) and are not parameters here. */
int earhip_hoa_decode_matrix(earhip_ctx *ctx, const char *layout, int n_coef, const int *orders,
                             const int *degrees, const char *normalization, float *out);
/* with the loudspeakers' real positions (see earhip_panner_create_positions) */
int earhip_hoa_decode_matrix_positions(earhip_ctx *ctx, const char *layout, int n_channels,
                                       const double *azimuth, const double *elevation, int n_coef,
                                       const int *orders, const int *degrees,
                                       const char *normalization, float *out);

/* ------------------------------------------------------------------------
 * (F) Composed Objects render block — the chain libear documents but does not
 * implement (docs/dsp.rst:40-71, include/ear/gain_calculators.hpp:45-56):
 *   per object: interpolated direct and diffuse gain vectors (a
 *   GainInterpolator<LinearInterpVector> each) summed into a direct and a
 *   diffuse loudspeaker bus; diffuse bus -> one BlockConvolver per loudspeaker
 *   (decorrelator FIRs); direct bus -> DelayBuffer(delay); out = sum.
 * It has the shape of VariableBlockSizeAdapter::ProcessFunc
 * (variable_block_size.hpp:19) and processes `nblocks` consecutive blocks per
 * call ("stream mode"), which is what lets the device path approach its
 * roofline.
 * ---------------------------------------------------------------------- */
typedef struct earhip_render earhip_render;

typedef struct earhip_render_config {
  int n_objects;  /* M: input channels handled by this instance (this GPU's shard) */
  int n_out;      /* N: loudspeakers */
  int block_size; /* B in [16, 4096], any factorisation (as libear's kissfft); the tuned
                     kernels are the powers of two from 64 (512 and 1024 above all) */
  int n_buses;    /* 1: direct bus only, written straight to the output
                     2: direct + diffuse with decorrelation, delay and mix */
  /* n_buses == 2: decorrelator FIRs [n_out][n_taps] (designDecorrelators,
   * include/ear/decorrelate.hpp:26-28); FIRs longer than a block are partitioned like
   * libear's Filter (src/dsp/block_convolver_impl.cpp:16-41), up to 64 partitions */
  const float *decorrelators;
  int n_taps;
  int delay;      /* compensation delay on the direct bus in samples
                     (decorrelatorCompensationDelay() = 255); 0 = none */
  int max_blocks; /* capacity T: largest nblocks of one process call */
} earhip_render_config;

int earhip_render_create(earhip_ctx *ctx, const earhip_render_config *cfg,
                         earhip_render **out);
int earhip_render_destroy(earhip_render *r);
/* Replace one object's gain curve: times[npoints] sorted; direct and diffuse
 * [npoints][n_out] (diffuse ignored / may be NULL when n_buses == 1).  Same
 * semantics as GainInterpolator::interp_points (gain_interpolator.hpp:27-43). */
int earhip_render_set_object_points(earhip_render *r, int object, int npoints,
                                    const int64_t *times, const float *direct,
                                    const float *diffuse);
/* Upload pending curve changes now (otherwise done at the next process), and make everything a call on the new curves
 * can need: the scratch of the list kernels (sized from the curves) and, for curves the hinge kernel is planned for, its
 * kink rows — for calls of max_blocks, max_blocks / 2 and one block.  This is where the library allocates and
 * synchronises when curves outgrow what is there; a process call on committed curves does neither.  What a call keeps per
 * CONTEXT — the level and mode words of the split-operand kernels, the per-object levels of the level probe, the grid
 * kernel's per-tile words — is made by earhip_render_create for max_blocks and n_objects of that renderer.  (The counted
 * exceptions: a process call whose launch plan no commit foresaw — an option changed in between — grows the scratch itself,
 * and a call that finds one of the context's buffers smaller than it needs — none of its renderers announced that size —
 * grows that; earhip_render_scratch_regrows counts the process calls of this renderer that did either: 0 in the
 * library's own tests and benchmarks.) */
int earhip_render_commit(earhip_render *r);
int earhip_render_scratch_regrows(const earhip_render *r, long *count);
/* Zero the DSP state (convolver tails, delay line) and set the sample clock. */
int earhip_render_reset(earhip_render *r, int64_t sample_time);
/* Process nblocks blocks from device memory: in_dev [n_objects][in_stride],
 * out_dev [n_out][out_stride], nblocks*block_size samples per channel.
 * Enqueues on the context's stream; does not synchronise. */
int earhip_render_process_device(earhip_render *r, size_t nblocks,
                                 const float *in_dev, size_t in_stride,
                                 float *out_dev, size_t out_stride);
/* Same from host channel pointers — libear's own calling convention (src/dsp/variable_block_size_impl.cpp:44-81) —: H2D,
 * kernels, D2H, synchronise.  Calls of 16 MB of inputs and more run as a PIPELINE of time chunks (a few blocks each) on three
 * streams: chunk c + 1 on its way to the device while chunk c's kernels run and chunk c - 1's outputs come back; from ordinary
 * pointers persistent staging threads (started at the first such call) gather the next chunk meanwhile.  Each chunk is an
 * ordinary process call of its blocks (the DSP state carries over): the results are those of consecutive calls, and the
 * last-call queries describe the last chunk. */
int earhip_render_process(earhip_render *r, size_t nblocks, const float *const *in,
                          float *const *out);
/* Kernel timing (HIP events on the context's stream around each launch).
 * enable != 0 starts collecting and zeroes the counters; enable = n > 1 times
 * every n-th process call only, starting with the next one (each timed call
 * records six events, which costs the GPU about 20 us of idle time). */
int earhip_render_enable_timing(earhip_render *r, int enable);
/* Sums since enable: [0] gain_mix kernel ms, [1] its launches, [2] decorrelate/
 * delay/mix kernel ms, [3] its launches, [4] segment-prep kernel ms, [5] its
 * launches.  Synchronises the stream. */
int earhip_render_get_timing(earhip_render *r, double out[6]);
/* Which gain kernel the last process call used: 0 = VALU with libear's exact
 * arithmetic (strict mode), 1 = f32 MFMA over slot lists, 2 = f32 MFMA on the tile grid (option
 * MFMA = 1 with every curve point on the 512-sample grid of a call of whole tiles), 3 = f16x2-split MFMA (all
 * curve points on the kernel's tile boundaries), 4 = f16x2-split MFMA over piece lists
 * (metadata that ignores the tile grid), 5 = f16x2-split MFMA with hinges (curves that ramp
 * all the time off the tile grid; the piece lists stand by: see below); -1 before the first
 * call. */
int earhip_render_gain_kernel(const earhip_render *r, int *kind);
/* Kernel 5 keeps 1e-6 for inputs down to 16 binades below the call's level (kernel 4: 21), so a
 * call it is planned for is decided ON THE DEVICE, from the level probe of the call's inputs: the
 * hinge kernel or the piece lists launched behind it.  *standby = 1 when the last call of this
 * renderer was planned for kernel 5 and the piece lists did it, else 0.  Synchronises the stream
 * (for benchmarks and tests that must name the kernel they measured).  The kernel that does a call
 * leaves a copy of the context's decision word in the renderer's own device slot: the answer stays
 * valid until the next process call of THIS renderer, whatever other renderers or gain stages of the
 * context do in between. */
int earhip_render_hinge_standby(earhip_render *r, int *standby);
/* Round 6: by default such a call is no longer handed over — kernel 5 has a third form of its body whose kink products are made
 * in f32 (7 instructions per value instead of 3; inputs 21 binades below the call's level keep ~17 bits of their products),
 * picked by the same device-side word; earhip_render_hinge_standby then answers 0 and *robust = 1 tells that the last call
 * of this renderer ran that form (0: the packed-f16 products sufficed, or the call was not kernel 5's).  Option HG_ROBUST = 0
 * restores the hand-over to the piece lists.  Synchronises the stream; valid until this renderer's next process call. */
int earhip_render_hinge_robust(earhip_render *r, int *robust);
/* The split-operand kernels (3, 4, 5) have two forms of their body: plain, and wide (the low pieces of the inputs scaled so
 * that they stay normal f16 numbers 21 binades below the call's level instead of 11).  Long calls (two rounds of workgroups
 * and more) pick on the device, from the level probe; shorter ones run the wide form.  *wide = 1 / 0: the form the last call
 * of this renderer ran (-1: its kernel has no split operands).  Synchronises the stream; valid until the next process call
 * of this renderer, like earhip_render_hinge_standby.  After a call that ran as two spans (earhip_render_last_tail_blocks
 * > 0) both queries — like earhip_render_gain_kernel and earhip_render_last_plan — describe the MAIN span (the whole
 * rounds of tiles: where the call's time goes); the short tail behind it runs the wide form without a hand-over. */
int earhip_render_wide_form(earhip_render *r, int *wide);
/* The launch plan of the last process call: [0] gain kernel (as above), [1] samples per
 * workgroup tile of the gain kernel, [2] number of such tiles, [3] grid-level object splits.
 * For tests and benchmarks that must know which kernel instantiation they measured. */
int earhip_render_last_plan(const earhip_render *r, int out[4]);
/* Layout of the piece lists of the last call (kernel 4, or the lists standing by behind kernel 5): *paired = 1 paired
 * (an object's base and delta piece share one input request; curves that hold most of the time), 0 packed (every chunk
 * sums its products among itself before it touches the running total: curves that ramp most of the time ALWAYS get this
 * one, whatever their other statistics — the planner's rule, asserted by the tests), -1: the call built no piece lists. */
int earhip_render_last_list_layout(const earhip_render *r, int *paired);
/* A stream call whose tiles are whole rounds of the chip's workgroups plus a few (1025 blocks of 512 samples on 256 CUs)
 * is run as two consecutive calls — the whole rounds, then the few blocks behind them spread over the chip by object
 * splits — instead of paying a whole round for the few.  *blocks = the blocks of the last call that ran as such a tail
 * (0: the call was not cut).  Results are those of the two calls made by the caller. */
int earhip_render_last_tail_blocks(const earhip_render *r, int *blocks);
/* earhip_render_process from host channel pointers (libear's calling convention, variable_block_size_impl.cpp:44-81) runs a long
 * call — 16 MB of inputs and more — as a pipeline of time chunks on three streams (transfer in / kernels / transfer out; options
 * HOST_CHUNK_MB, HOST_THREADS, HOST_BIND).  *chunks = the chunks the last such call of this renderer ran as (0: one piece). */
int earhip_render_last_host_chunks(const earhip_render *r, int *chunks);
/* Bytes of device scratch (segment descriptors, slot / piece / hinge lists) the last process call needed: sized per call
 * from its launch plan and the curves (the piece lists from the most ramps any window of a tile's length overlaps, per
 * object), not for the worst case. */
int earhip_render_scratch_bytes(const earhip_render *r, size_t *bytes);

/* ------------------------------------------------------------------------
 * (J) Multi-GPU exchange — no libear counterpart (libear is single-device).  Objects are
 * sharded over the GPUs of one node, one process and one earhip_ctx per GPU; every rank
 * renders its shard completely with (F) — everything after the buses is linear and per
 * channel — and the partial outputs are summed by ONE RCCL reduce-scatter over the
 * channel axis on the context's stream (xGMI): rank r ends up owning rows
 * [r * per, (r + 1) * per) of the shared bus.
 *   earhip_comm_unique_id: rank 0 makes the 128-byte id; the caller hands it to every
 *     rank by its own means (MPI, torch.distributed, a socket, a file);
 *   earhip_comm_create: collective over all ranks (ncclCommInitRank);
 *   earhip_comm_channel_range: rows of the exchange buffers (n_out rounded up to a
 *     multiple of the ranks; the rows past n_out must be zero) and the channels [lo, hi)
 *     rank `rank` owns — ragged when the ranks do not divide n_out (10 channels on 4
 *     ranks: 3, 3, 3, 1);
 *   earhip_render_exchange_device: partial_dev [padded_rows][row_stride] ->
 *     owned_dev [rows_per_rank][row_stride].  Ordered behind everything enqueued on the
 *     context's stream so far (the render that wrote partial_dev) but run on the
 *     communicator's own stream, so that it overlaps the next render; does not
 *     synchronise the host.  slot (0 or 1) names one of two exchanges in flight
 *     (double-buffered outputs);
 *   earhip_comm_gather_device: the shared loudspeaker bus in ONE place — collects the
 *     owned slices into full_dev [padded_rows][row_stride], on every rank (root < 0: one
 *     all-gather) or on rank `root` only (the others send their slice straight to it,
 *     world - 1 transfers over world - 1 different links at once; full_dev may be NULL
 *     on the ranks that do not receive).  Runs on the communicator's stream behind the
 *     exchange of the same slot; the first n_out rows of full_dev are the bus;
 *   earhip_comm_wait(slot): work enqueued on the context's stream after this call runs
 *     after the last exchange / gather issued with that slot — call it before rendering
 *     into that slot's partial buffer again and before reading its owned / full buffer;
 *   earhip_comm_last_exchange_ms: what the collectives of a slot took on the
 *     communicator's stream (HIP events around them); waits for them.
 * ---------------------------------------------------------------------- */
typedef struct earhip_comm earhip_comm;
int earhip_comm_unique_id(void *id128);
int earhip_comm_create(earhip_ctx *ctx, int rank, int world, const void *id128, earhip_comm **out);
int earhip_comm_destroy(earhip_comm *comm);
int earhip_comm_channel_range(int n_out, int rank, int world, int *padded_rows, int *lo, int *hi);
int earhip_render_exchange_device(earhip_comm *comm, int slot, const float *partial_dev,
                                  float *owned_dev, size_t rows_per_rank, size_t row_stride);
int earhip_comm_gather_device(earhip_comm *comm, int slot, const float *owned_dev, float *full_dev,
                              size_t rows_per_rank, size_t row_stride, int root);
int earhip_comm_wait(earhip_comm *comm, int slot);
int earhip_comm_last_exchange_ms(earhip_comm *comm, int slot, double *ms);
/* What RCCL says about the communicator: info[0] ranks (ncclCommCount), [1] this rank (ncclCommUserRank), [2] the HIP device it
 * lives on (ncclCommCuDevice), [3] the RCCL version (ncclGetVersion) — a benchmark line shows with it that N ranks really met. */
int earhip_comm_info(earhip_comm *comm, int info[4]);
/* Measured rate of one link direction: every rank sends `bytes` bytes to rank + shift and receives as many from rank - shift
 * (one ncclSend / ncclRecv pair per rank, all at once: the pattern of the exchange's point-to-point steps), `reps` times between
 * HIP events on the communicator's stream.  *GBps = bytes a rank sent per second / 1e9 (0 with one rank).  Collective. */
int earhip_comm_link_probe(earhip_comm *comm, size_t bytes, int shift, int reps, double *GBps);

#ifdef __cplusplus
}
#endif
#endif /* EARHIP_H */
