// api_render.hip — (F) the composed Objects render block of include/earhip.h:
// K0 segment prep -> K1 gain_mix (direct + diffuse buses) -> K2
// decorrelate_delay_mix, `nblocks` blocks per call, state resident in HBM.
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>


#include "common.h"
#include "curves.h"
#include "host_gather.h"
#include "fft_kernels.h"
#include "render_kernels.h"

using namespace earhip;

namespace earhip {
// twiddle table exp(-2*pi*i*t/L), computed in double
std::vector<cf> make_twiddles(int L) {
  std::vector<cf> tw(L);
  const double pi = 3.14159265358979323846264338327950288;
  for (int t = 0; t < L; t++) {
    const double a = -2.0 * pi * (double)t / (double)L;
    tw[t] = cf_make((float)std::cos(a), (float)std::sin(a));
  }
  return tw;
}

template <int L>
static void launch_spectrum_t(const float *in, size_t stride, int n_valid, const cf *tw, cf *out,
                              int rows, hipStream_t s) {
  hipLaunchKernelGGL((k_spectrum_real<L>), dim3(rows), dim3(kFftThreads), 0, s, in, stride,
                     n_valid, tw, out);
}
void launch_spectrum(int L, const float *in, size_t stride, int n_valid, const cf *tw, cf *out,
                     int rows, hipStream_t s) {
  switch (L) {
    case 64: launch_spectrum_t<64>(in, stride, n_valid, tw, out, rows, s); break;
    case 128: launch_spectrum_t<128>(in, stride, n_valid, tw, out, rows, s); break;
    case 256: launch_spectrum_t<256>(in, stride, n_valid, tw, out, rows, s); break;
    case 512: launch_spectrum_t<512>(in, stride, n_valid, tw, out, rows, s); break;
    case 1024: launch_spectrum_t<1024>(in, stride, n_valid, tw, out, rows, s); break;
    case 2048: launch_spectrum_t<2048>(in, stride, n_valid, tw, out, rows, s); break;
    case 4096: launch_spectrum_t<4096>(in, stride, n_valid, tw, out, rows, s); break;
    case 8192: launch_spectrum_t<8192>(in, stride, n_valid, tw, out, rows, s); break;
    default: {  // any other size: mixed radix at run time
      FftShape S;
      if (!fft_make_shape(L, &S)) fail_invalid("FFT size must be in [4, 8192]");
      hipLaunchKernelGGL(k_spectrum_real_rt, dim3(rows), dim3(kFftThreads), fft_rt_lds(k_spectrum_real_rt, L), s, S,
                         in, stride, n_valid, tw, out);
    }
  }
  EARHIP_HIP(hipGetLastError());
}
}  // namespace earhip

template <int L>
static void launch_decor_t(const DecorParams &P, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((k_decorrelate_delay_mix<L>), grid, dim3(256), 0, s, P);
}
static void launch_decor(int L, const DecorParams &P, dim3 grid, hipStream_t s, bool force_wg) {
  switch (L) {
    case 128: launch_decor_t<128>(P, grid, s); break;
    case 256: launch_decor_t<256>(P, grid, s); break;
    case 512: launch_decor_t<512>(P, grid, s); break;
    case 1024:
      if (force_wg) {  // the workgroup-per-run kernel (FIRs of several partitions; option K2_WG)
        launch_decor_t<1024>(P, grid, s);
      } else {  // one wave per run, kDecorWaves runs per workgroup
        hipLaunchKernelGGL(k_decorrelate_wave, dim3((grid.x + kDecorWaves - 1) / kDecorWaves, grid.y),
                           dim3(64 * kDecorWaves), 0, s, P);
      }
      break;
    case 2048: launch_decor_t<2048>(P, grid, s); break;
    case 4096: launch_decor_t<4096>(P, grid, s); break;
    case 8192: launch_decor_t<8192>(P, grid, s); break;
    default: {  // any other block size: mixed-radix transforms at run time
      FftShape S;
      if (!fft_make_shape(L, &S)) fail_invalid("block_size must be in [16, 4096]");
      const size_t lds = fft_rt_lds(k_decorrelate_delay_mix_rt, L, sizeof(float) * (L / 2));
      hipLaunchKernelGGL(k_decorrelate_delay_mix_rt, grid, dim3(256), lds, s, P, S);
    }
  }
  EARHIP_HIP(hipGetLastError());
}

// Run length (odd, so that the warm-up block pairs with the first one) of k_decorrelate_wave for a
// call of T blocks on N loudspeakers.  A run of R blocks costs (R+1)/2 pair transforms, each
// workgroup puts one wave on every SIMD, and a CU holds three workgroups (LDS): the cost of a
// round of k = 1..3 resident workgroups per CU is pairs x c[k] with the measured pair times
// c = 6.9, 9.3, 12.2 us (latency-bound at this occupancy).  NOTES.md (round 2, section 4, K2).
static int wave_run_len(int T, int N, int num_cus) {
  const double c[4] = {0.0, 6.9, 9.3, 12.2};
  int best = 1;
  double best_cost = 1e30;
  for (int R = 1; R <= 31; R += 2) {
    const long runs = (T + R - 1) / R;
    const long wgs = (runs + earhip::kDecorWaves - 1) / earhip::kDecorWaves * N;
    const long full = wgs / (3L * num_cus), rem = wgs - full * 3L * num_cus;
    const double pairs = (R + 1) / 2;
    const double cost = pairs * (full * c[3] + c[(rem + num_cus - 1) / num_cus]);
    if (cost <= best_cost) best_cost = cost, best = R;  // (ties: the longer run does less warm-up work)
    if (runs == 1) break;
  }
  return best;
}

// Staging threads of the host-pointer entry point, started at the first long call and kept: a call
// neither creates threads nor allocates.  A long call is cut into TIME chunks (a few blocks each: ~8 MB of inputs); a job =
// gather every channel's samples of chunk g into the pinned buffer, chunk-major ([chunk][channel][samples of the chunk]: a
// chunk is one linear transfer), thread t taking every nthreads-th channel; done[g] counts the threads that have finished
// chunk g (the caller starts that chunk's transfer then, while the threads gather the next one).
struct GatherPool {
  static constexpr int kMaxGroups = 64;
  NumaMap numa;
  bool streaming = true;              // (option HOST_NT, read when a job is submitted)
  int job_node = -1;                  // the node this job's rows live on (-1: run anywhere)
  const cpu_set_t *job_cpus = nullptr;
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable go, finished_cv;
  uint64_t generation = 0;
  int finished = 0;
  bool quit = false;
  // the job
  const float *const *in = nullptr;
  float *dst = nullptr;
  size_t n = 0;                      // samples per channel in the call
  size_t cstart[kMaxGroups + 1] = {0};  // chunk g = samples [cstart[g], cstart[g + 1]) of every channel
  int M = 0, groups = 0;
  std::atomic<int> done[kMaxGroups];
  int nthreads() const { return (int)threads.size(); }
  size_t group_len(int g) const { return cstart[g + 1] - cstart[g]; }
  void start(int count) {
    for (int t = 0; t < count; t++)
      threads.emplace_back([this, t] {
        uint64_t seen = 0;
        int my_node = -1;  // the node this thread is bound to (-1: not bound)
        for (;;) {
          int want_node;
          const cpu_set_t *want_cpus;
          {
            std::unique_lock<std::mutex> lk(mu);
            go.wait(lk, [&] { return quit || generation != seen; });
            if (quit) return;
            seen = generation;
            want_node = job_node, want_cpus = job_cpus;
          }
          if (want_node != my_node) {  // (a failure leaves the thread where it is: placement is an optimisation)
            if (want_node >= 0 && want_cpus) {
              if (sched_setaffinity(0, sizeof(cpu_set_t), want_cpus) == 0) my_node = want_node;
            } else if (numa.ok && sched_setaffinity(0, sizeof(cpu_set_t), &numa.allowed) == 0) {
              my_node = -1;
            }
          }
          const int nt = nthreads();
          for (int g = 0; g < groups; g++) {
            const size_t len = group_len(g), at = cstart[g];
            float *base = dst + (size_t)M * at;
            if (streaming) {
              for (int m = t; m < M; m += nt) stream_copy(base + (size_t)m * len, in[m] + at, len);
#if defined(__x86_64__)
              _mm_sfence();  // (streaming stores are weakly ordered: globally visible before the chunk counts as gathered)
#endif
            } else {
              for (int m = t; m < M; m += nt) std::memcpy(base + (size_t)m * len, in[m] + at, sizeof(float) * len);
            }
            done[g].fetch_add(1, std::memory_order_release);
          }
          std::lock_guard<std::mutex> lk(mu);
          if (++finished == nt) finished_cv.notify_one();
        }
      });
  }
  void submit(const float *const *in_, float *dst_, size_t n_, const size_t *starts, int nchunks, int M_, bool bind, bool nt) {
    const int node = bind && numa.ok ? NumaMap::rows_node(in_, M_, n_) : -1;
    std::lock_guard<std::mutex> lk(mu);
    in = in_, dst = dst_, n = n_, M = M_;
    for (int g = 0; g <= nchunks; g++) cstart[g] = starts[g];
    streaming = nt;
    job_cpus = numa.cpus_of(node);
    job_node = job_cpus ? node : -1;
    groups = nchunks;
    for (auto &d : done) d.store(0);
    finished = 0;
    generation++;
    go.notify_all();
  }
  void wait_all() {
    std::unique_lock<std::mutex> lk(mu);
    finished_cv.wait(lk, [&] { return finished == nthreads(); });
  }
  ~GatherPool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    go.notify_all();
    for (auto &th : threads) th.join();
  }
};

// Copy streams and events of long host-pointer calls (made at the first such call and kept): the chunks of a call go
// H2D on `in`, through the kernels on the context's stream, and D2H on `out`, each stage ordered behind the one
// before it by the chunk's events — chunk c's kernels and the transfer of its outputs run beside the transfer of c + 1.
struct StreamPipe {
  hipStream_t in = nullptr, out = nullptr;
  hipEvent_t ev_in[GatherPool::kMaxGroups], ev_k[GatherPool::kMaxGroups], ev_out[GatherPool::kMaxGroups];
  bool made = false;
  void make() {
    if (made) return;
    EARHIP_HIP(hipStreamCreateWithFlags(&in, hipStreamNonBlocking));
    EARHIP_HIP(hipStreamCreateWithFlags(&out, hipStreamNonBlocking));
    for (int i = 0; i < GatherPool::kMaxGroups; i++) {
      EARHIP_HIP(hipEventCreateWithFlags(&ev_in[i], hipEventDisableTiming));
      EARHIP_HIP(hipEventCreateWithFlags(&ev_k[i], hipEventDisableTiming));
      EARHIP_HIP(hipEventCreateWithFlags(&ev_out[i], hipEventDisableTiming));
    }
    made = true;
  }
  ~StreamPipe() {
    if (!made) return;
    for (int i = 0; i < GatherPool::kMaxGroups; i++) {
      (void)hipEventDestroy(ev_in[i]);
      (void)hipEventDestroy(ev_k[i]);
      (void)hipEventDestroy(ev_out[i]);
    }
    (void)hipStreamDestroy(in);
    (void)hipStreamDestroy(out);
  }
};

struct earhip_render {
  earhip_ctx *ctx = nullptr;
  int M = 0, N = 0, B = 0, K = 1, D = 0, T = 0;
  // The decorrelators' own partition size Bk (and transform size Lk = 2 Bk).  A linear convolution does not
  // depend on how it is partitioned, so callers' blocks of 1024, 2048 ... samples run through 512-sample
  // partitions whenever the FIRs fit one of them: the wave kernel (k_decorrelate_wave) exists for that size
  // and is 2.5 times faster per sample than the workgroup kernel at 2048 points (BASELINE config 5: K2 0.125
  // -> 0.05 ms).  Otherwise Bk = B, libear's own partitioning (src/dsp/block_convolver_impl.cpp:16-41).
  int Bk = 0, Lk = 0;
  int NP = 1;  // partitions of the decorrelator FIRs (ceil(n_taps / Bk))
  std::unique_ptr<CurveSet> curves;
  int64_t t = 0;  // sample clock: absolute time of the next block
  int last_plan[3] = {0, 0, 0};  // tile samples, tiles, grid-level object splits of the last call
  int last_paired = -1;          // layout of its piece lists: 1 paired, 0 packed, -1 none built
  size_t last_scratch_bytes = 0;  // K0 / K1 scratch the last call needed
  long scratch_regrows = 0;       // process calls that had to grow the scratch themselves (none on committed curves)
  // What the last call's gain kernel decided on the device: the kernel that did the call left a copy of the context's mode
  // word in THIS renderer's own slot (rec[0]: the call, or the main span of a call cut in two; rec[1]: the tail span), so the
  // queries below stay valid whatever other renderers of the context do afterwards.
  DevBuf<unsigned> rec;
  bool last_gated = false;       // the last call (its main span) was planned for the hinge kernel behind a device-side gate
  bool last_device_form = false; // ... its split-operand kernel picked its form (plain / wide) on the device
  bool last_hg_robust = false;   // ... a gated call beyond the packed kink products' span runs the hinge kernel's robust form (no stand-by lists)
  int last_kind = -1;  // gain kernel of the last call: 0 VALU (strict), 1 f32 MFMA, 2 f32 MFMA on the tile grid, 3 f16x2 MFMA, 4 f16x2 MFMA over piece lists,
                       // 5 f16x2 MFMA with hinges (gain_hg.h)
  int run_len = 11;       // blocks per decorrelator run of the workgroup kernel
  bool run_len_set = false;  // EARHIP_RUN given: also fixes the run length of the wave kernel

  DevBuf<SegDesc> desc;
  DevBuf<float> bus;  // [gsplit][K*N][bus_stride], strides chosen per call
  int max_gsplit = 1;
  DevBuf<cf> H, tw;
  DevBuf<float> tail[2], dly[2], hist[2];  // tails [NP][N][Bk]; diffuse-bus history [N][(NP-1) Bk]
  DevBuf<float> ztail, zdly, zhist;  // all-zero state, never written: what the first call after a reset reads
  bool fresh = true;          // no call since create / reset: the state is zero
  int cur = 0;  // which state buffer holds the current state
  // host-pointer staging
  DevBuf<float> d_in, d_out;
  PinBuf<float> p_in, p_out;
  std::unique_ptr<GatherPool> gather;  // staging threads of long host-pointer calls
  StreamPipe pipe;                     // ... their copy streams and events
  int last_host_chunks = 0;            // time chunks the last host-pointer call ran as (0: one piece)
  // timing
  bool timing = false;
  bool last_timed = false;
  int timing_every = 1;      // time every n-th process call (the event records cost ~3 us of idle GPU each)
  long timing_calls = 0;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // (a call cut into a main part and a tail — process_device — is ONE call of the timing: its two sets of events add up)
  struct Pending { hipEvent_t e[6]; bool has_k2; bool continues; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> pool;
  double acc_ms[3] = {0, 0, 0};
  double acc_n[3] = {0, 0, 0};

  ~earhip_render() {
    for (auto &p : pending)
      for (auto e : p.e)
        if (e) (void)hipEventDestroy(e);
    for (auto e : pool) (void)hipEventDestroy(e);
  }

  hipEvent_t get_event() {
    if (!pool.empty()) {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
    hipEvent_t e;
    EARHIP_HIP(hipEventCreate(&e));
    return e;
  }

  void drain_timing() {
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &p : pending) {
      float ms = 0;
      const double one = p.continues ? 0.0 : 1.0;  // (the second part of a call cut in two adds time, not a call)
      EARHIP_HIP(hipEventElapsedTime(&ms, p.e[0], p.e[1]));
      acc_ms[2] += ms; acc_n[2] += one;
      EARHIP_HIP(hipEventElapsedTime(&ms, p.e[2], p.e[3]));
      acc_ms[0] += ms; acc_n[0] += one;
      if (p.has_k2) {
        EARHIP_HIP(hipEventElapsedTime(&ms, p.e[4], p.e[5]));
        acc_ms[1] += ms; acc_n[1] += one;
      }
      for (auto e : p.e)
        if (e) pool.push_back(e);
    }
    pending.clear();
  }

  // the launch plan of a call of nblocks blocks at the current sample clock
  MixLaunch plan_call(size_t nblocks, size_t in_stride) {
    const int nsamples = (int)(nblocks * (size_t)B);
    MixLaunch ml = plan_mix(ctx, curves->plan(), M, nsamples, ctx->strict, max_gsplit, curves->aligned_tile(t),
                            curves->ramp_share(), curves->gain_scale(), curves->point_density_all(), curves->pair_waste(256), curves->pair_waste(512),
                            curves->hinge_exact_share(in_stride, (size_t)nsamples), curves->tiles_aligned(kF32GridTile, t),
                            curves->deltas_per_pair(256));
    const size_t bus_stride = ((size_t)nsamples + 3) & ~(size_t)3;
    while (ml.gsplit > 1 && bus_stride * K * N * ml.gsplit > bus.n) ml.gsplit /= 2;
    return ml;
  }

  // Everything a call on the CURRENT curves can need beyond what earhip_render_create allocated — K0 / K1 scratch for
  // the plan of the call (piece or hinge lists are sized from the curves), the kink rows of the hinge kernel — is made
  // HERE, where curves change (earhip_render_commit, or the implicit commit of the first process call after
  // set_object_points), for the longest call (max_blocks), a call half as long and a single block, with a quarter of
  // headroom: process calls on committed curves neither allocate nor synchronise (include/earhip.h, conventions).
  size_t last_in_stride = 0;
  void reserve_for_curves() {
    size_t need = 0;
    bool hinge = false;
    const size_t lens[3] = {(size_t)T, (size_t)std::max(1, T / 2), (size_t)1};
    for (size_t nb : lens) {
      const MixLaunch ml = plan_call(nb, last_in_stride ? last_in_stride : nb * (size_t)B);
      need = std::max(need, scratch_units(*curves, ml, M));
      hinge = hinge || ml.hinge;
    }
    if (hinge) curves->ensure_kinks(ctx);
    if (need > desc.n) {
      EARHIP_HIP(hipStreamSynchronize(ctx->stream));  // (the old scratch may still be in use)
      desc.reserve(need + need / 4);
    }
  }

  // Workgroups of the gain kernel the chip holds at once for a plan: the tiles of a call are worked off in rounds of so many
  // (8-wave forms: one workgroup per CU; 4-wave forms: two)
  int resident_workgroups(const MixLaunch &ml) const {
    if (!(ml.split || ml.pieces || ml.hinge)) return 0;
    return ctx->num_cus * (ml.tile() >= 512 ? 1 : 2);
  }

  // A tile of the headline scene is 100 us of one CU: a call of k rounds of tiles PLUS A FEW (1025 blocks; the 513 of a
  // time-sharded rank: its share and a lead block) used to cost k + 1 rounds — 0.306 ms for 513 blocks where 512 take 0.235.
  // Such a call is cut in two: the whole rounds as they were, and the few tiles behind them as a short call of their own,
  // which the planner spreads over the chip by splitting the objects (as in block mode).  The cut costs a second K0, a
  // second K2 launch and the boundaries (~30 us): taken when the tail is at most a quarter of a round.  Results are those
  // of two consecutive calls (every call length is a valid call: the state carries over).
  void process_device(size_t nblocks, const float *in_dev, size_t in_stride, float *out_dev, size_t out_stride) {
    MixLaunch whole;  // the plan of the uncut call, made once (process_span takes it as it is)
    bool have_plan = false;
    if (!ctx->strict && ctx->get(OPT_TAILCUT, 1) != 0 && nblocks >= 2) {
      last_in_stride = in_stride;
      if (curves->dirty()) {
        curves->commit(ctx);
        reserve_for_curves();
      }
      const MixLaunch ml = whole = plan_call(nblocks, in_stride);
      have_plan = true;
      const int W = resident_workgroups(ml);
      if (W > 0 && ml.gsplit == 1 && ml.ntiles > W) {
        const size_t tile = (size_t)ml.tile();
        const size_t full = (size_t)ml.ntiles / (size_t)W, rest = (size_t)ml.ntiles % (size_t)W;
        const size_t main_samples = full * (size_t)W * tile;
        // (option TAILCUT = v: tails of up to v / 8 of a round are cut off; default 2, 0: never)
        const size_t eighths = (size_t)std::min(std::max(ctx->get(OPT_TAILCUT, 2), 0), 7);
        if (rest > 0 && rest <= (size_t)W * eighths / 8 && main_samples % (size_t)B == 0) {
          const size_t main_blocks = main_samples / (size_t)B;
          process_span(main_blocks, in_dev, in_stride, out_dev, out_stride, false);
          const int kind = last_kind, plan3[3] = {last_plan[0], last_plan[1], last_plan[2]};
          const size_t scratch = last_scratch_bytes;
          const bool gated = last_gated, device_form = last_device_form, hg_robust = last_hg_robust;
          const int paired = last_paired;
          process_span(nblocks - main_blocks, in_dev + main_samples, in_stride, out_dev + main_samples, out_stride, true);
          last_kind = kind;  // (what the call is reported as: its main part — kernel, plan and what it decided on the device)
          last_gated = gated, last_device_form = device_form, last_paired = paired, last_hg_robust = hg_robust;
          for (int i = 0; i < 3; i++) last_plan[i] = plan3[i];
          last_scratch_bytes = scratch;
          last_tail_blocks = (int)(nblocks - main_blocks);
          return;
        }
      }
    }
    last_tail_blocks = 0;
    process_span(nblocks, in_dev, in_stride, out_dev, out_stride, false, have_plan ? &whole : nullptr);
  }
  int last_tail_blocks = 0;  // blocks of the last call that ran as its tail part (0: the call was not cut)

  void process_span(size_t nblocks, const float *in_dev, size_t in_stride, float *out_dev,
                    size_t out_stride, bool continues, const MixLaunch *planned = nullptr) {
    const int nsamples = (int)(nblocks * (size_t)B);
    last_in_stride = in_stride;
    if (curves->dirty()) {
      curves->commit(ctx);
      reserve_for_curves();
    }
    const bool strict = ctx->strict;
    MixLaunch ml = planned ? *planned : plan_call(nblocks, in_stride);
    if (ml.hinge) curves->ensure_kinks(ctx);  // (already there unless an option changed the plan since the commit)

    last_kind = ml.f32grid ? 2 : ml.hinge ? 5 : ml.pieces ? 4 : ml.split ? 3 : ml.mfma ? 1 : 0;
    {
      // K0 / K1 scratch for THIS plan and THESE curves (round 3: 537 MB at the headline's size for any curves): reserved
      // when the curves were committed (reserve_for_curves).  The one exception, documented in earhip.h: a plan that no
      // commit foresaw (an option changed between commit and call, a call on a time grid the curves' phase statistics
      // did not predict) grows it here, geometrically, behind a synchronisation.
      const size_t need = scratch_units(*curves, ml, M);
      if (need > desc.n) {
        EARHIP_HIP(hipStreamSynchronize(ctx->stream));
        desc.reserve(need + need / 2);
        scratch_regrows++;
      }
      last_scratch_bytes = need * 16;
    }
    const size_t bus_stride = ((size_t)nsamples + 3) & ~(size_t)3;
    const size_t part_stride = bus_stride * K * N;
    // (the bus is sized for every plan plan_mix can make, earhip_render_create; should a tuning knob push a plan
    // beyond it, plan_call has taken fewer object splits, always a valid plan)
    if (part_stride * ml.gsplit > bus.n) fail_internal("bus buffer too small for this launch plan");
    last_paired = (ml.pieces || ml.hinge) ? (ml.paired ? 1 : 0) : -1;
    last_plan[0] = ml.tile();
    last_plan[1] = ml.ntiles;
    last_plan[2] = ml.gsplit;
    Pending pd;
    hipEvent_t *evp = nullptr;
    // (the second part of a call cut in two is timed when its first part was)
    const bool timed = timing && (continues ? last_timed : (timing_calls++ % timing_every) == 0);
    last_timed = timed;
    if (timed) {
      for (int i = 0; i < 6; i++) pd.e[i] = get_event();
      pd.has_k2 = K == 2;
      pd.continues = continues;
      evp = pd.e;
    }
    const long lazy_before = ctx->lazy_allocs;
    ctx->record_slot = rec.p + (continues ? 1 : 0);  // (consumed by the launch_gain_mix below)
    if (K == 1) {
      // direct bus only: K1 writes the output rows itself
      if (ml.gsplit == 1) {
        launch_gain_mix(ctx, *curves, ml, strict, t, nsamples, in_dev, in_stride, out_dev,
                        out_stride, 0, desc.p, evp);
      } else {
        launch_gain_mix(ctx, *curves, ml, strict, t, nsamples, in_dev, in_stride, bus.p,
                        bus_stride, part_stride, desc.p, evp);
        hipLaunchKernelGGL(k_sum_parts, dim3((nsamples + 255) / 256, N), dim3(256), 0,
                           ctx->stream, bus.p, part_stride, ml.gsplit, bus_stride, N, nsamples,
                           out_dev, out_stride);
        EARHIP_HIP(hipGetLastError());
      }
    } else {
      launch_gain_mix(ctx, *curves, ml, strict, t, nsamples, in_dev, in_stride, bus.p,
                      bus_stride, part_stride, desc.p, evp);
      DecorParams P;
      P.bus = bus.p;
      P.bus_stride = bus_stride;
      P.part_stride = part_stride;
      P.nparts = ml.gsplit;
      const bool wave_k2 = Lk == 1024 && NP == 1 && !ctx->get(OPT_K2_WG);
      const int kblocks = (int)(nblocks * (size_t)(B / Bk));  // the call in decorrelator partitions
      if (wave_k2 && ml.gsplit > 1) {
        // The wave kernel has one wave per run: summing the object splits there is a chain of
        // dependent loads on the call's critical path (block mode).  Sum them into slab 0 with the
        // whole chip first (K N rows; in place: a thread reads and writes its own sample only).
        if (evp) EARHIP_HIP(hipEventRecord(evp[4], ctx->stream));
        hipLaunchKernelGGL(k_sum_parts, dim3((nsamples + 255) / 256, K * N), dim3(256), 0, ctx->stream, bus.p,
                           part_stride, ml.gsplit, bus_stride, K * N, nsamples, bus.p, bus_stride);
        EARHIP_HIP(hipGetLastError());
        P.nparts = 1;
      }
      P.out = out_dev;
      P.out_stride = out_stride;
      P.tw = tw.p;
      P.dly_in = fresh ? zdly.p : dly[cur].p;
      P.dly_out = dly[cur ^ 1].p;
      P.N = N;
      P.T = kblocks;
      const int R = wave_k2 && !run_len_set ? wave_run_len(kblocks, N, ctx->num_cus) : run_len;
      P.R = R;
      P.D = D;
      P.hist_len = (NP - 1) * Bk;
      P.hist_in = fresh ? zhist.p : hist[cur].p;
      P.hist_out = hist[cur ^ 1].p;
      const dim3 grid((unsigned)((kblocks + R - 1) / R), N);
      if (evp && P.nparts == ml.gsplit) EARHIP_HIP(hipEventRecord(evp[4], ctx->stream));
      // one launch per partition of the FIRs: partition 0 writes (decorrelated + delayed direct), the
      // others add their share of the decorrelated signal (render_kernels.h)
      for (int part = 0; part < NP; part++) {
        P.H = H.p + (size_t)part * N * Lk;
        P.tail_in = (fresh ? ztail.p : tail[cur].p) + (fresh ? 0 : (size_t)part * N * Bk);
        P.tail_out = tail[cur ^ 1].p + (size_t)part * N * Bk;
        P.shift = part * Bk;
        P.accumulate = part > 0 ? 1 : 0;
        if (wave_k2) {
          hipLaunchKernelGGL(k_decorrelate_wave, dim3((grid.x + kDecorWaves - 1) / kDecorWaves, grid.y), dim3(64 * kDecorWaves),
                             0, ctx->stream, P);
          EARHIP_HIP(hipGetLastError());
        } else {
          launch_decor(Lk, P, grid, ctx->stream, true);
        }
      }
      if (evp) EARHIP_HIP(hipEventRecord(evp[5], ctx->stream));
      cur ^= 1;
      fresh = false;
    }
    if (timed) pending.push_back(pd);
    last_gated = ctx->last_gate_idx >= 0;  // (set by launch_gain_mix for this call)
    last_device_form = ctx->last_wide_idx >= 0;
    last_hg_robust = last_gated && ctx->last_hinge_robust;
    if (ctx->lazy_allocs != lazy_before) scratch_regrows++;  // (a buffer of the context no renderer had announced: earhip.h)
    t += nsamples;
  }
};

extern "C" {

int earhip_render_create(earhip_ctx *ctx, const earhip_render_config *cfg, earhip_render **out) {
  return guarded([&] {
    require(ctx != nullptr && cfg != nullptr && out != nullptr, "NULL argument");
    require(cfg->n_objects >= 1 && cfg->n_out >= 1, "n_objects and n_out must be >= 1");
    require(cfg->n_buses == 1 || cfg->n_buses == 2, "n_buses must be 1 or 2");
    // (any size, as libear's kissfft: 480, 960, 1920 ... through mixed-radix passes, other primes through the
    // generic butterfly; the tuned kernels are the power-of-two sizes from 64)
    FftShape shape;
    require(cfg->block_size >= 16 && cfg->block_size <= 4096 && fft_make_shape(2 * cfg->block_size, &shape),
            "block_size must be in [16, 4096]");
    require(cfg->max_blocks >= 1, "max_blocks must be >= 1");
    require((int64_t)cfg->max_blocks * cfg->block_size < ((int64_t)1 << 30),
            "max_blocks * block_size too large");
    require(cfg->delay >= 0, "delay must be >= 0");
    if (cfg->n_buses == 2) {
      require(cfg->decorrelators != nullptr && cfg->n_taps >= 1,
              "n_buses == 2 needs decorrelator filters");
      require((cfg->n_taps + cfg->block_size - 1) / cfg->block_size <= 64,
              "decorrelator filters of more than 64 blocks are not supported by the fused render "
              "(use BlockConvolver)");
    } else {
      require(cfg->delay == 0, "delay needs n_buses == 2");
    }
    ctx->use();
    std::unique_ptr<earhip_render> r(new earhip_render);
    r->ctx = ctx;
    r->M = cfg->n_objects;
    r->N = cfg->n_out;
    r->B = cfg->block_size;
    r->K = cfg->n_buses;
    r->D = cfg->delay;
    r->T = cfg->max_blocks;
    r->Bk = r->B;
    if (r->K == 2 && r->B > 512 && r->B % 512 == 0 && cfg->n_taps <= 512 && !ctx->get(OPT_K2_OWN_BLOCK)) r->Bk = 512;
    r->Lk = 2 * r->Bk;
    // workgroup decorrelator kernel: blocks per run.  Block 1024 (BASELINE config 5, 512 blocks x 24
    // loudspeakers): 7 -> K2 0.113 ms, 5 -> 0.117, 11 -> 0.128, 15 -> 0.140 (two rounds of workgroups that fill
    // the chip evenly beat one ragged round)
    if (r->Lk == 2048) r->run_len = 7;
    if (ctx->has(OPT_RUN)) {  // tuning knob: blocks per decorrelator run (odd)
      const int v = ctx->get(OPT_RUN);
      if (v >= 1 && v <= 255) r->run_len = v | 1, r->run_len_set = true;
    }
    r->curves.reset(new CurveSet(r->M, r->K * r->N, r->K, false));
    const size_t max_samples = (size_t)r->T * r->B;
    reserve_call_words(ctx, r->M, max_samples);  // (the context's words for calls of this size: no process call makes them)
    r->rec.alloc_zero(2, ctx->stream);
    // smallest tile any gain kernel of this context uses (f32 MFMA: 16 * nrt samples)
    const size_t min_tile = (size_t)std::min(16 * ctx->nrt, std::min(64 * ctx->spl, 256));
    const size_t max_tiles = (max_samples + min_tile - 1) / min_tile;
    // (the scratch of K0 / K1 — descriptors, slot, piece or hinge lists — is sized per call from the plan and the curves:
    // process_device; here only what the grid kernel needs for the longest call)
    r->desc.alloc((size_t)r->M * ((max_samples + 255) / 256) + 1);
    (void)max_tiles;  // (the piece-list kernel's tiles: 256 or 512 samples)
    // Buses: [gsplit][K*N][pad4(nsamples)] per call.  Grid-level object splits (gsplit > 1) are only
    // chosen for calls with few tiles: plan_mix doubles gsplit while gsplit * (ntiles / tpw) stays below
    // 2 * num_cus, so gsplit * ntiles < (4 * num_cus + gsplit) * tpw with tiles of at most 256 samples
    // (tpw = 1) or 16 * nrt samples (f32 MFMA kernel, tpw adjacent tiles per workgroup).
    r->max_gsplit = 16;  // block mode: more splits make K2 sum more partial slabs than K1 gains
    if (ctx->has(OPT_GSPLIT)) {  // tuning knob: grid-level splits of short calls
      const int v = ctx->get(OPT_GSPLIT);
      if (v >= 1 && v <= 32) r->max_gsplit = v;
    }
    r->bus.alloc_zero((size_t)r->K * r->N * bus_samples_bound(ctx, max_samples, r->max_gsplit), ctx->stream);
    if (r->K == 2) {
      const auto tw = make_twiddles(r->Lk);
      r->tw.alloc(r->Lk);
      EARHIP_HIP(hipMemcpy(r->tw.p, tw.data(), sizeof(cf) * r->Lk, hipMemcpyHostToDevice));
      // H[p] = DFT_L(zero-padded partition p of the FIR: taps [p B, (p + 1) B), Filter::Filter,
      // block_convolver_impl.cpp:16-41), computed with the device transform, then made exactly Hermitian so
      // that two real blocks separate cleanly
      r->NP = (cfg->n_taps + r->Bk - 1) / r->Bk;
      const size_t rows = (size_t)r->NP * r->N;
      std::vector<float> parts(rows * r->Bk, 0.0f);  // [NP][N][B]
      for (int n = 0; n < r->N; n++)
        for (int t = 0; t < cfg->n_taps; t++)
          parts[((size_t)(t / r->Bk) * r->N + n) * r->Bk + t % r->Bk] = cfg->decorrelators[(size_t)n * cfg->n_taps + t];
      DevBuf<float> taps;
      taps.alloc(parts.size());
      EARHIP_HIP(hipMemcpy(taps.p, parts.data(), sizeof(float) * parts.size(), hipMemcpyHostToDevice));
      r->H.alloc(rows * r->Lk);
      launch_spectrum(r->Lk, taps.p, r->Bk, r->Bk, r->tw.p, r->H.p, (int)rows, ctx->stream);
      EARHIP_HIP(hipStreamSynchronize(ctx->stream));
      std::vector<cf> h(rows * r->Lk);
      EARHIP_HIP(hipMemcpy(h.data(), r->H.p, sizeof(cf) * h.size(), hipMemcpyDeviceToHost));
      for (size_t n = 0; n < rows; n++) {
        cf *hn = h.data() + n * r->Lk;
        hn[0].y = 0.0f;
        hn[r->Bk].y = 0.0f;
        for (int k = 1; k < r->Bk; k++) hn[r->Lk - k] = cf_conj(hn[k]);
      }
      EARHIP_HIP(hipMemcpy(r->H.p, h.data(), sizeof(cf) * h.size(), hipMemcpyHostToDevice));
      const size_t hist_n = (size_t)r->N * std::max((r->NP - 1) * r->Bk, 1);
      for (int i = 0; i < 2; i++) {
        r->tail[i].alloc_zero(rows * r->Bk, ctx->stream);
        r->dly[i].alloc_zero((size_t)r->N * std::max(r->D, 1), ctx->stream);
        r->hist[i].alloc_zero(hist_n, ctx->stream);
        if (i == 0) {
          r->ztail.alloc_zero((size_t)r->N * r->Bk, ctx->stream);
          r->zdly.alloc_zero((size_t)r->N * std::max(r->D, 1), ctx->stream);
          r->zhist.alloc_zero(hist_n, ctx->stream);
        }
      }
    }
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    *out = r.release();
  });
}

int earhip_render_destroy(earhip_render *r) {
  return guarded([&] {
    if (!r) return;
    (void)hipSetDevice(r->ctx->device);
    (void)hipStreamSynchronize(r->ctx->stream);
    delete r;
  });
}

int earhip_render_set_object_points(earhip_render *r, int object, int npoints,
                                    const int64_t *times, const float *direct,
                                    const float *diffuse) {
  return guarded([&] {
    require(r != nullptr, "render must not be NULL");
    require(npoints >= 1, "interp_points must not be empty");
    require(times != nullptr && direct != nullptr, "times/direct must not be NULL");
    require(r->K == 1 || diffuse != nullptr, "diffuse gains must not be NULL when n_buses == 2");
    const int N = r->N, cols = r->K * N;
    std::vector<float> rows((size_t)npoints * cols);
    for (int k = 0; k < npoints; k++) {
      std::memcpy(&rows[(size_t)k * cols], direct + (size_t)k * N, sizeof(float) * N);
      if (r->K == 2)
        std::memcpy(&rows[(size_t)k * cols + N], diffuse + (size_t)k * N, sizeof(float) * N);
    }
    r->curves->set_object(object, npoints, times, rows.data());
  });
}

int earhip_render_commit(earhip_render *r) {
  return guarded([&] {
    require(r != nullptr, "render must not be NULL");
    r->ctx->use();
    r->curves->commit(r->ctx);
    r->reserve_for_curves();
  });
}

int earhip_render_reset(earhip_render *r, int64_t sample_time) {
  return guarded([&] {
    require(r != nullptr, "render must not be NULL");
    r->ctx->use();
    r->t = sample_time;
    r->fresh = true;  // the next call reads the all-zero state and rewrites its own pair completely
  });
}

int earhip_render_process_device(earhip_render *r, size_t nblocks, const float *in_dev,
                                 size_t in_stride, float *out_dev, size_t out_stride) {
  return guarded([&] {
    require(r != nullptr, "render must not be NULL");
    require(in_dev != nullptr && out_dev != nullptr, "device pointers must not be NULL");
    require(nblocks <= (size_t)r->T, "nblocks exceeds max_blocks");
    require(in_stride >= nblocks * r->B && out_stride >= nblocks * r->B, "stride too small");
    if (nblocks == 0) return;
    r->ctx->use();
    r->process_device(nblocks, in_dev, in_stride, out_dev, out_stride);
  });
}

int earhip_render_process(earhip_render *r, size_t nblocks, const float *const *in,
                          float *const *out) {
  return guarded([&] {
    require(r != nullptr, "render must not be NULL");
    require(in != nullptr && out != nullptr, "in and out must not be NULL");
    require(nblocks <= (size_t)r->T, "nblocks exceeds max_blocks");
    if (nblocks == 0) return;
    earhip_ctx *ctx = r->ctx;
    ctx->use();
    const size_t n = nblocks * r->B;
    const size_t cap = (size_t)r->T * r->B;
    r->p_in.reserve(cap * r->M);
    r->p_out.reserve(cap * r->N);
    r->d_in.reserve(cap * r->M);
    r->d_out.reserve(cap * r->N);
    const size_t in_bytes = sizeof(float) * n * r->M;
    const bool dbg = ctx->get(OPT_DEBUG_TIMING) != 0;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = dbg ? now() : 0.0;
    const bool short_call = in_bytes < ((size_t)16 << 20) || r->M < 16 || (ctx->has(OPT_HOST_CHUNK_MB) && ctx->get(OPT_HOST_CHUNK_MB) <= 0);
    // Channel pointers that are evenly spaced inside memory the device reaches (earhip_host_alloc /
    // earhip_host_register: the columns of a pinned matrix) need no staging copy: strided DMA in, and
    // the output rows written in place.  stride in floats; 0: not that shape.
    auto direct_stride = [&](const float *const *ch, int count) -> size_t {
      if (ctx->host_ranges.empty() || count < 1) return 0;
      // (addresses as integers: the channel pointers need not belong to one array as far as C++ knows)
      const uintptr_t a0 = (uintptr_t)ch[0];
      const uintptr_t step = count > 1 ? (uintptr_t)ch[1] - a0 : sizeof(float) * n;
      if ((a0 & 15) != 0 || step % 16 != 0 || step < sizeof(float) * n || step > ((uintptr_t)1 << 40)) return 0;
      for (int c = 1; c < count; c++)
        if ((uintptr_t)ch[c] - (uintptr_t)ch[c - 1] != step) return 0;
      return ctx->host_reachable(ch[0], step * (count - 1) + sizeof(float) * n) ? (size_t)(step / sizeof(float)) : 0;
    };
    const size_t in_st = direct_stride(in, r->M);
    const size_t out_st = r->NP <= 1 || !short_call ? direct_stride(out, r->N) : 0;
    r->last_host_chunks = 0;
    if (!short_call) {
      // Long calls (libear's calling convention for offline renders: host channel pointers, any length): the call is cut
      // into TIME chunks of a few blocks (option HOST_CHUNK_MB: ~8 MB of inputs each, at most 64 chunks) and the chunks run
      // through a three-stage pipeline on three streams: H2D of chunk c + 1 (its own copy stream) beside the kernels of
      // chunk c (the context's stream: each chunk is an ordinary process call of its blocks, the DSP state carries over)
      // beside the D2H of chunk c - 1 (a second copy stream).  The kernels are ~100 x faster than the bus, so the call
      // takes what its inputs take over PCIe plus one chunk's kernels and output transfer.  From ordinary (pageable)
      // pointers the staging threads gather chunk c + 1 into the pinned buffer (chunk-major: one linear transfer per
      // chunk) while chunk c is on the bus, and the outputs of finished chunks are handed back to the caller's rows while
      // later chunks are still in flight; from device-reachable rows (earhip_host_alloc / _register) neither copy exists.
      const size_t block_bytes = sizeof(float) * (size_t)r->B * r->M;
      // (rows in device-reachable memory go by strided DMA, which wants long row pieces: 32 MB chunks = 32 KB pieces at 1024
      // objects; staged rows: 16 MB — the pieces the staging threads copy are then 16 KB, and the first chunk's gather, which
      // nothing overlaps, stays short)
      const size_t want_bytes = (size_t)std::max(1, ctx->get(OPT_HOST_CHUNK_MB, in_st ? 32 : 16)) << 20;
      size_t cb = std::max<size_t>(1, want_bytes / block_bytes);
      cb = std::max(cb, (nblocks + GatherPool::kMaxGroups - 2) / (GatherPool::kMaxGroups - 1));
      size_t gran = 1;  // (chunks start on 16-byte boundaries of the staging rows: vector loads)
      while ((gran * (size_t)r->B) % 4 != 0) gran++;
      cb = (cb + gran - 1) / gran * gran;
      // (option HOST_FIRST = 1: from staged rows the first chunk a quarter of the others — its gather is the one nothing overlaps —;
      // measured level with whole chunks (2.906 vs 2.905 ms per 64-block call): the staging threads are through ALL chunks of such a
      // call after 0.66 ms, the call's time is the bus's plus ~19 us per chunk — 11 between two copies on a stream, 8 for the event
      // and the kernels' stream behind it, tools/experiments/h2d_chunks.hip — so the default keeps whole chunks)
      size_t cb0 = in_st || ctx->get(OPT_HOST_FIRST, 0) == 0 ? cb : std::max(gran, cb / 4 / gran * gran);
      if (cb0 >= nblocks) cb0 = cb;
      size_t cstart[GatherPool::kMaxGroups + 1];
      int nch = 0;
      for (size_t b = 0; b < nblocks; b += nch == 1 ? cb0 : cb) cstart[nch++] = b * r->B;  // (nch counts the chunk being opened)
      cstart[nch] = n;
      r->pipe.make();
      if (!in_st) {
        if (!r->gather) {
          r->gather.reset(new GatherPool);
          const int want = ctx->get(OPT_HOST_THREADS, 0);
          r->gather->start(want >= 1 && want <= 64 ? want : default_staging_threads());
        }
        r->gather->submit(in, r->p_in.p, n, cstart, nch, r->M, ctx->get(OPT_HOST_BIND, 0) != 0, ctx->get(OPT_HOST_NT, 1) != 0);
      }
      hipError_t err = hipSuccess;
      std::string fail;
      int scattered = 0;
      auto scatter_chunk = [&](int c) {
        const size_t at = cstart[c], len = cstart[c + 1] - at;
        const float *base = r->p_out.p + (size_t)r->N * at;
        for (int ch = 0; ch < r->N; ch++) std::memcpy(out[ch] + at, base + (size_t)ch * len, sizeof(float) * len);
      };
      for (int c = 0; c < nch; c++) {
        const size_t at = cstart[c], len = cstart[c + 1] - at;
        float *din = r->d_in.p + (size_t)r->M * at, *dout = r->d_out.p + (size_t)r->N * at;
        if (!in_st) {
          GatherPool &gp = *r->gather;
          while (gp.done[c].load(std::memory_order_acquire) < gp.nthreads()) {
            // (meanwhile: outputs of chunks whose transfer has landed go back to the caller's rows)
            if (!out_st && scattered < c && err == hipSuccess && hipEventQuery(r->pipe.ev_out[scattered]) == hipSuccess) scatter_chunk(scattered++);
            else std::this_thread::yield();
          }
        }
        if (err != hipSuccess || !fail.empty()) continue;  // (the staging threads still finish their job)
        if (in_st)
          err = hipMemcpy2DAsync(din, sizeof(float) * len, in[0] + at, sizeof(float) * in_st, sizeof(float) * len, r->M,
                                 hipMemcpyHostToDevice, r->pipe.in);
        else
          err = hipMemcpyAsync(din, r->p_in.p + (size_t)r->M * at, sizeof(float) * len * r->M, hipMemcpyHostToDevice, r->pipe.in);
        if (err == hipSuccess) err = hipEventRecord(r->pipe.ev_in[c], r->pipe.in);
        if (err == hipSuccess) err = hipStreamWaitEvent(ctx->stream, r->pipe.ev_in[c], 0);
        if (err != hipSuccess) continue;
        try {
          r->process_device(len / r->B, din, len, dout, len);
        } catch (const Error &e) {
          fail = e.msg;
          continue;
        }
        err = hipEventRecord(r->pipe.ev_k[c], ctx->stream);
        if (err == hipSuccess) err = hipStreamWaitEvent(r->pipe.out, r->pipe.ev_k[c], 0);
        if (err != hipSuccess) continue;
        if (out_st)
          err = hipMemcpy2DAsync(out[0] + at, sizeof(float) * out_st, dout, sizeof(float) * len, sizeof(float) * len, r->N,
                                 hipMemcpyDeviceToHost, r->pipe.out);
        else
          err = hipMemcpyAsync(r->p_out.p + (size_t)r->N * at, dout, sizeof(float) * len * r->N, hipMemcpyDeviceToHost, r->pipe.out);
        if (err == hipSuccess) err = hipEventRecord(r->pipe.ev_out[c], r->pipe.out);
      }
      if (!in_st) r->gather->wait_all();
      const double t_c = dbg ? now() : 0.0;
      // (everything queued is waited for whatever happened above: nothing of this call is in flight when it returns)
      (void)hipStreamSynchronize(r->pipe.in);
      (void)hipStreamSynchronize(ctx->stream);
      (void)hipStreamSynchronize(r->pipe.out);
      EARHIP_HIP(err);
      if (!fail.empty()) throw Error{EARHIP_INTERNAL_ERROR, fail};
      const double t_d = dbg ? now() : 0.0;
      if (!out_st)
        for (; scattered < nch; scattered++) scatter_chunk(scattered);
      r->last_host_chunks = nch;
      if (dbg)
        fprintf(stderr, "render_process: %d chunks of %zu blocks%s: enqueue incl. gather %.1f us, drain %.1f us, scatter of the rest %.1f us\n", nch,
                cb, in_st ? " (device-reachable rows)" : "", t_c - t_a, t_d - t_c, now() - t_d);
      return;
    }
    if (in_st) {
      EARHIP_HIP(hipMemcpy2DAsync(r->d_in.p, sizeof(float) * n, in[0], sizeof(float) * in_st, sizeof(float) * n, r->M,
                                  hipMemcpyHostToDevice, ctx->stream));
    } else {
      // short calls (block mode): one gather, one transfer.  Splitting a 2 MB block into groups whose
      // transfers overlap the gather was measured twice and loses (132 -> 150 us per call at the headline
      // shape: four DMA start-ups cost more than the 30 us of gather they hide).
      // ... in `groups` parts, the transfer of a part starting while the next is gathered (EARHIP_BLOCK_GROUPS;
      // default 2 above 1 MB: a second DMA start-up costs less than the half gather it hides, four cost more:
      // 0.134 / 0.126 / 0.138 ms per 512-sample call of 1024 objects with 1 / 2 / 4 groups, same box.  Enqueueing the
      // segment descriptors — the part of K0 that does not look at the inputs — ahead of the gather was measured
      // too and loses: 0.146 / 0.125 / 0.138: the launch is host time in front of the gather, and the probe it
      // leaves behind the transfer is one more kernel in the chain)
      const int groups_env = ctx->get(OPT_BLOCK_GROUPS);
      const int groups = groups_env >= 1 && groups_env <= 8 ? groups_env : (in_bytes >= ((size_t)1 << 20) ? 2 : 1);
      for (int g = 0; g < groups; g++) {
        const int m0 = (int)((int64_t)r->M * g / groups), m1 = (int)((int64_t)r->M * (g + 1) / groups);
        for (int m = m0; m < m1; m++) std::memcpy(r->p_in.p + (size_t)m * n, in[m], sizeof(float) * n);
        if (m1 > m0)
          EARHIP_HIP(hipMemcpyAsync(r->d_in.p + (size_t)m0 * n, r->p_in.p + (size_t)m0 * n, sizeof(float) * n * (m1 - m0),
                                    hipMemcpyHostToDevice, ctx->stream));
      }
    }
    const double t_b = dbg ? now() : 0.0;
    // Short calls: K2 writes the few output rows straight into host memory (the caller's own rows when they
    // are reachable, else the pinned staging buffer: no D2H copy to start and wait for, 126 -> 119 us per
    // 512-sample call at the headline shape).  The other direction does not pay — kernels that read their 2 MB
    // of inputs over PCIe themselves are slower than the copy engine plus kernels (129 us) — and neither does
    // spinning on a completion word behind one more launch.
    // (FIRs of several partitions accumulate into the output: that stays in device memory)
    const bool direct_out = r->NP <= 1;
    float *dst = out_st ? out[0] : direct_out ? r->p_out.p : r->d_out.p;
    r->process_device(nblocks, r->d_in.p, n, dst, out_st ? out_st : n);
    if (!direct_out)
      EARHIP_HIP(hipMemcpyAsync(r->p_out.p, r->d_out.p, sizeof(float) * n * r->N, hipMemcpyDeviceToHost, ctx->stream));
    const double t_c = dbg ? now() : 0.0;
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    const double t_d = dbg ? now() : 0.0;
    if (!out_st)
      for (int c = 0; c < r->N; c++) std::memcpy(out[c], r->p_out.p + c * n, sizeof(float) * n);
    if (dbg)
      fprintf(stderr, "render_process: gather+H2D enqueue %.1f us, launches %.1f us, wait %.1f us, scatter %.1f us\n", t_b - t_a,
              t_c - t_b, t_d - t_c, now() - t_d);
  });
}

/* diagnostic builds (-DEARHIP_K2_PROF) only: the s_memtime stamps wave 0 of two workgroups of the last k_decorrelate_wave launch
 * left at its phase boundaries, out[2][32]; an ordinary build reports "not built in" */
int earhip_debug_k2_prof(earhip_ctx *ctx, unsigned long long *out64) {
  return guarded([&] {
    require(ctx != nullptr && out64 != nullptr, "NULL argument");
#ifdef EARHIP_K2_PROF
    ctx->use();
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    EARHIP_HIP(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_k2_prof), sizeof(unsigned long long) * 64));
#else
    fail_invalid("this build has no K2 phase stamps (-DEARHIP_K2_PROF)");
#endif
  });
}

int earhip_render_enable_timing(earhip_render *r, int enable) {
  return guarded([&] {
    require(r != nullptr, "render must not be NULL");
    r->ctx->use();
    if (r->timing) r->drain_timing();
    r->timing = enable != 0;
    r->timing_every = enable > 1 ? enable : 1;
    r->timing_calls = 0;
    for (int i = 0; i < 3; i++) r->acc_ms[i] = r->acc_n[i] = 0;
  });
}

int earhip_render_get_timing(earhip_render *r, double out[6]) {
  return guarded([&] {
    require(r != nullptr && out != nullptr, "NULL argument");
    r->ctx->use();
    r->drain_timing();
    out[0] = r->acc_ms[0]; out[1] = r->acc_n[0];
    out[2] = r->acc_ms[1]; out[3] = r->acc_n[1];
    out[4] = r->acc_ms[2]; out[5] = r->acc_n[2];
  });
}

int earhip_render_gain_kernel(const earhip_render *r, int *kind) {
  return guarded([&] {
    require(r != nullptr && kind != nullptr, "NULL argument");
    *kind = r->last_kind;
  });
}

int earhip_render_hinge_standby(earhip_render *r, int *standby) {
  return guarded([&] {
    require(r != nullptr && standby != nullptr, "NULL argument");
    *standby = 0;
    earhip_ctx *ctx = r->ctx;
    // (the copy of the mode word the call's own gain kernel left in this renderer's slot: valid until this renderer's next call)
    if (r->last_kind != 5 || !r->last_gated || r->last_hg_robust) return;  // (robust form allowed: nobody stands by)
    ctx->use();
    unsigned word = 0;
    EARHIP_HIP(hipMemcpyAsync(&word, r->rec.p, sizeof(word), hipMemcpyDeviceToHost, ctx->stream));
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    if (!(word & kModeRecorded)) fail_internal("no gain kernel recorded the call's mode word");
    *standby = (word & kGateHingeUnsafe) ? 1 : 0;
  });
}

int earhip_render_hinge_robust(earhip_render *r, int *robust) {
  return guarded([&] {
    require(r != nullptr && robust != nullptr, "NULL argument");
    *robust = 0;
    earhip_ctx *ctx = r->ctx;
    if (r->last_kind != 5 || !r->last_hg_robust) return;
    ctx->use();
    unsigned word = 0;
    EARHIP_HIP(hipMemcpyAsync(&word, r->rec.p, sizeof(word), hipMemcpyDeviceToHost, ctx->stream));
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    if (!(word & kModeRecorded)) fail_internal("no gain kernel recorded the call's mode word");
    *robust = hinge_span_exceeded(word, r->M) ? 1 : 0;
  });
}

int earhip_render_wide_form(earhip_render *r, int *wide) {
  return guarded([&] {
    require(r != nullptr && wide != nullptr, "NULL argument");
    *wide = 1;
    earhip_ctx *ctx = r->ctx;
    if (r->last_kind < 3) *wide = -1;  // (no split operands at all)
    if (r->last_kind < 3 || !r->last_device_form) return;
    ctx->use();
    unsigned word = 0;
    EARHIP_HIP(hipMemcpyAsync(&word, r->rec.p, sizeof(word), hipMemcpyDeviceToHost, ctx->stream));
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    if (!(word & kModeRecorded)) fail_internal("no gain kernel recorded the call's mode word");
    *wide = (word & 1u) ? 1 : 0;
  });
}

int earhip_render_scratch_bytes(const earhip_render *r, size_t *bytes) {
  return guarded([&] {
    require(r != nullptr && bytes != nullptr, "NULL argument");
    *bytes = r->last_scratch_bytes;
  });
}

int earhip_render_scratch_regrows(const earhip_render *r, long *count) {
  return guarded([&] {
    require(r != nullptr && count != nullptr, "NULL argument");
    *count = r->scratch_regrows;
  });
}

int earhip_render_last_tail_blocks(const earhip_render *r, int *blocks) {
  return guarded([&] {
    require(r != nullptr && blocks != nullptr, "NULL argument");
    *blocks = r->last_tail_blocks;
  });
}

int earhip_render_last_host_chunks(const earhip_render *r, int *chunks) {
  return guarded([&] {
    require(r != nullptr && chunks != nullptr, "NULL argument");
    *chunks = r->last_host_chunks;
  });
}

int earhip_render_last_list_layout(const earhip_render *r, int *paired) {
  return guarded([&] {
    require(r != nullptr && paired != nullptr, "NULL argument");
    *paired = r->last_paired;
  });
}

int earhip_render_last_plan(const earhip_render *r, int out[4]) {
  return guarded([&] {
    require(r != nullptr && out != nullptr, "NULL argument");
    out[0] = r->last_kind;
    for (int i = 0; i < 3; i++) out[1 + i] = r->last_plan[i];
  });
}

}  // extern "C"
