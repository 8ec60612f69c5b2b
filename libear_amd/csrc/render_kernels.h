// render_kernels.h — K2 `decorrelate_delay_mix`: everything of the composed
// Objects render block (libear docs/dsp.rst:40-71) that follows the bus-forming
// gain stage, fused into one kernel:
//
//   diffuse bus --> BlockConvolver (static decorrelator FIR; overlap-add exactly as
//                   block_convolver_impl.cpp:143-237).  One launch handles ONE partition of the
//                   FIR (<= B taps); a FIR longer than a block (blocks of 64..256 with the 512-tap
//                   decorrelators) is P launches: partition p convolves the diffuse bus delayed by
//                   p blocks (the previous call's last P - 1 blocks come from the history state)
//                   with its own overlap-add tail and ADDS to the output — libear sums the
//                   partitions' spectra before one inverse transform (:193-215), here their
//                   time-domain results are summed; equal up to rounding
//   direct bus  --> DelayBuffer(D)            (delay_buffer_impl.cpp:19-40)
//   out = decorrelated + delayed              (docs/figures/objects.png)
//
// One 256-thread workgroup owns one loudspeaker and a run of R consecutive
// blocks.  Two real blocks t, t+1 ride through ONE complex FFT of length
// L = 2B as z = pad(x_t) + i*pad(x_{t+1}); because the FIR is real its spectrum
// H is Hermitian and  IDFT(H (.) DFT(z)) = y_t + i*y_{t+1}, so no untangling is
// needed.  The overlap-add tail of the block before the run is recomputed (one
// extra block per run) instead of being exchanged between workgroups, so the
// time axis is embarrassingly parallel; the tail of the block before the call
// and the delay line come from the persistent state, which the workgroup that
// owns the last block of the call rewrites (double-buffered).
#pragma once

#include <hip/hip_runtime.h>

#include "fft_lds.h"

namespace earhip {

struct DecorParams {
  const float *bus;    // [part][2N][bus_stride]; rows 0..N-1 direct, N..2N-1 diffuse
  size_t bus_stride;
  size_t part_stride;
  int nparts;
  float *out;          // [N][out_stride]
  size_t out_stride;
  const cf *H;         // [N][L] decorrelator spectra (full, Hermitian)
  const cf *tw;        // [L]
  const float *tail_in;  // [N][B] un-normalised OLA tail of the block before the call
  float *tail_out;
  const float *dly_in;   // [N][D] last D direct-bus samples before the call
  float *dly_out;
  int N, T, R, D;
  // partition pass (workgroup kernel only; the wave kernel handles single-partition FIRs)
  int shift = 0;               // this partition's delay in samples (p * B)
  int accumulate = 0;          // 1: out += decorrelated (no direct path, no delay-line update)
  int hist_len = 0;            // (P - 1) * B: diffuse-bus samples kept from before the call
  const float *hist_in = nullptr;  // [N][hist_len]
  float *hist_out = nullptr;       // written by the pass with accumulate == 0
};

// radix-4 butterfly with the three twiddles given (already conjugated for DIR > 0 by
// the caller's table): r4_core of fft_lds.h without its table reads
template <int DIR>
__device__ __forceinline__ void r4_core_w(cf v0, cf v1, cf v2, cf v3, cf *out, int Ns, int j,
                                          const cf (&w)[3]) {
  const int k = j & (Ns - 1);
  v1 = cf_mul(v1, DIR > 0 ? cf_conj(w[0]) : w[0]);
  v2 = cf_mul(v2, DIR > 0 ? cf_conj(w[1]) : w[1]);
  v3 = cf_mul(v3, DIR > 0 ? cf_conj(w[2]) : w[2]);
  const cf a0 = cf_add(v0, v2), a1 = cf_sub(v0, v2);
  const cf a2 = cf_add(v1, v3), a3 = cf_mul_i<DIR>(cf_sub(v1, v3));
  const int base = ((j - k) << 2) + k;
  out[base] = cf_add(a0, a2);
  out[base + Ns] = cf_add(a1, a3);
  out[base + 2 * Ns] = cf_sub(a0, a2);
  out[base + 3 * Ns] = cf_sub(a1, a3);
}

// the same butterfly with its four results returned instead of stored
template <int DIR>
__device__ __forceinline__ void r4_vals_w(cf v0, cf v1, cf v2, cf v3, const cf (&w)[3], cf (&o)[4]) {
  v1 = cf_mul(v1, DIR > 0 ? cf_conj(w[0]) : w[0]);
  v2 = cf_mul(v2, DIR > 0 ? cf_conj(w[1]) : w[1]);
  v3 = cf_mul(v3, DIR > 0 ? cf_conj(w[2]) : w[2]);
  const cf a0 = cf_add(v0, v2), a1 = cf_sub(v0, v2);
  const cf a2 = cf_add(v1, v3), a3 = cf_mul_i<DIR>(cf_sub(v1, v3));
  o[0] = cf_add(a0, a2);
  o[1] = cf_add(a1, a3);
  o[2] = cf_sub(a0, a2);
  o[3] = cf_sub(a1, a3);
}

// fft_run_passes of fft_lds.h for a power-of-four L with one butterfly per thread and
// the twiddles of this thread's butterflies held in registers (twr[p - 1] for pass p)
template <int L, int DIR, int NT>
__device__ __forceinline__ cf *fft_run_passes_w(cf *src, cf *dst, const cf (&twr)[fft_r4_passes(L) - 1][3],
                                                int first, int stop, int tid) {
  constexpr int NR4 = fft_r4_passes(L);
  static_assert(!fft_has_r2(L) && L / 4 == NT, "power-of-four length, one butterfly per thread");
  int Ns = 1;
  for (int p = 0; p < first; ++p) Ns *= 4;
#pragma unroll
  for (int p = 0; p < NR4; ++p) {
    if (p < first || p >= stop) continue;
    __syncthreads();
    if (p == 0) stockham_r4<L, DIR>(src, dst, nullptr, 1, tid);
    else r4_core_w<DIR>(src[tid], src[tid + L / 4], src[tid + L / 2], src[tid + 3 * L / 4], dst, Ns, tid, twr[p - 1]);
    cf *t = src;
    src = dst;
    dst = t;
    Ns *= 4;
  }
  return src;
}

// radix-4 passes [first, stop) of fft_run_passes (fft_lds.h) with table twiddles and several
// butterflies per thread; the trailing radix-2 pass of an odd log2(L) is left to the caller
template <int L, int DIR, int NT>
__device__ __forceinline__ cf *fft_r4_range(cf *src, cf *dst, const cf *tw, int first, int stop, int tid) {
  int Ns = 1;
  for (int p = 0; p < first; ++p) Ns *= 4;
#pragma unroll
  for (int p = 0; p < fft_r4_passes(L); ++p) {
    if (p < first || p >= stop) continue;
    __syncthreads();
    for (int j = tid; j < L / 4; j += NT) stockham_r4<L, DIR>(src, dst, tw, Ns, j);
    cf *t = src;
    src = dst;
    dst = t;
    Ns *= 4;
  }
  return src;
}

template <int L>
__global__ void __launch_bounds__(256) k_decorrelate_delay_mix(DecorParams P) {
  constexpr int B = L / 2;
  constexpr int NT = 256;
  constexpr int EPT = B / NT > 0 ? B / NT : 1;  // input / output samples per thread and block
  constexpr int BPT = L / 4 / NT > 0 ? L / 4 / NT : 1;  // radix-4 butterflies per thread and pass
  // The kernel is latency-bound (83% of its wave cycles in s_waitcnt/barriers), so
  // nothing on the per-pass critical path may come from global memory: twiddles
  // live in LDS (up to L = 2048), the channel's spectrum H in registers, and the
  // next pair's bus samples are requested before the current pair's passes start.
  // At one butterfly per thread (L = 1024) the twiddles of a thread's butterflies are
  // 12 constants: held in registers they take 30 % off the LDS traffic of a pass, which
  // is what six co-resident workgroups per CU are bound by.
  constexpr bool kTwInRegs = L == 4 * NT;
  // 2048 points: two radix-4 butterflies and four radix-2 butterflies per thread; the same three
  // pass boundaries stay in registers (table twiddles from LDS)
  constexpr bool kFused2k = L == 2048 && NT == 256;
  constexpr bool kTwInLds = L <= 2048 && !kTwInRegs;
  __shared__ __attribute__((aligned(16))) cf lds[2 * L + (kTwInLds ? L : 0)];
  __shared__ float tail[B];
  cf *a = lds, *b = lds + L;
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int first = blockIdx.x * P.R;
  const int last = min(first + P.R, P.T);
  const float norm = 1.0f / (float)(2 * B);  // block_convolver_impl.cpp:212
  const cf *H = P.H + (size_t)n * L;
  const float *direct = P.bus + (size_t)n * P.bus_stride;
  const float *diffuse = P.bus + (size_t)(P.N + n) * P.bus_stride;
  float *out = P.out + (size_t)n * P.out_stride;
  const cf *tw = P.tw;
  if (kTwInLds) {
    cf *twl = lds + 2 * L;
    for (int i = tid; i < L; i += NT) twl[i] = P.tw[i];
    tw = twl;  // visible after the first barrier below
  }

  cf twr[kTwInRegs ? fft_r4_passes(L) - 1 : 1][3];
  if constexpr (kTwInRegs) {
    int Ns = 4;
#pragma unroll
    for (int p = 1; p < fft_r4_passes(L); p++) {
      const int k = tid & (Ns - 1), step = L / (Ns * 4);
      twr[p - 1][0] = P.tw[k * step];
      twr[p - 1][1] = P.tw[2 * k * step];
      twr[p - 1][2] = P.tw[3 * k * step];
      Ns *= 4;
    }
  }

  // one bus sample = sum of the partial slabs of a short call's object splits, four loads
  // in flight at a time (a dependent chain of up to 16 loads cost block mode 14 us)
  auto bus_at = [&](const float *row, int s) {
    const float *q = row + s;
    float v = q[0];
    int p = 1;
    for (; p + 3 < P.nparts; p += 4) {
      const float a = q[(size_t)p * P.part_stride], b = q[(size_t)(p + 1) * P.part_stride];
      const float c = q[(size_t)(p + 2) * P.part_stride], d = q[(size_t)(p + 3) * P.part_stride];
      v += (a + b) + (c + d);
    }
    for (; p < P.nparts; p++) v += q[(size_t)p * P.part_stride];
    return v;
  };
  // direct bus delayed by D: sample s of the call (s may reach back into state)
  auto delayed = [&](int s) {
    const int sd = s - P.D;
    return sd >= 0 ? bus_at(direct, sd) : P.dly_in[(size_t)n * P.D + (sd + P.D)];
  };
  // diffuse bus delayed by this partition's shift: sample s of the call (s - shift may reach back into
  // the history state)
  auto diffuse_at = [&](int s) {
    const int sd = s - P.shift;
    return sd >= 0 ? bus_at(diffuse, sd) : P.hist_in[(size_t)n * P.hist_len + (sd + P.hist_len)];
  };
  // this thread's input samples of the pair (tb, tb+1): real part block tb, imaginary tb+1
  auto load_pair = [&](int tb, cf (&z)[EPT]) {
    const bool have_re = tb >= 0 && tb < last, have_im = tb + 1 < last;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      const int i = tid + e * NT;
      z[e] = cf_make(0.0f, 0.0f);
      if (i < B) {
        if (have_re) z[e].x = diffuse_at(tb * B + i);
        if (have_im) z[e].y = diffuse_at((tb + 1) * B + i);
      }
    }
  };

  // H of the butterflies this thread multiplies in the first inverse pass
  cf h[BPT][4];
#pragma unroll
  for (int q = 0; q < BPT; q++) {
    const int j = tid + q * NT;
#pragma unroll
    for (int r = 0; r < 4; r++) h[q][r] = j < L / 4 ? H[j + r * (L / 4)] : cf_make(0.0f, 0.0f);
  }

  if (first == 0)
    for (int i = tid; i < B; i += NT) tail[i] = P.tail_in[(size_t)n * B + i];

  cf z[EPT];
  load_pair(first - 1, z);
  for (int tb = first - 1; tb < last; tb += 2) {
    const bool have_re = tb >= 0, have_im = tb + 1 < last;
    __syncthreads();  // previous pair's reads of a/b are finished
    if constexpr (kTwInRegs) {
      // One butterfly per thread: the first forward pass takes this thread's own two
      // samples (inputs 2 and 3 of its butterfly are the zero padding) straight from
      // registers — no staging of the block in LDS, no padding writes.
      static_assert(!kTwInRegs || EPT == 2, "two samples per thread and block");
      r4_core<L, -1>(z[0], z[EPT - 1], cf_make(0.0f, 0.0f), cf_make(0.0f, 0.0f), b, tw, 1, tid);
    } else if constexpr (kFused2k) {
#pragma unroll
      for (int q = 0; q < 2; q++)  // butterfly tid + 256 q: samples tid + 256 q and + 512; the rest is padding
        r4_core<L, -1>(z[q], z[q + 2], cf_make(0.0f, 0.0f), cf_make(0.0f, 0.0f), b, tw, 1, tid + q * NT);
    } else {
#pragma unroll
      for (int e = 0; e < EPT; e++) {
        const int i = tid + e * NT;
        if (i < B) {
          a[i] = z[e];
          a[i + B] = cf_make(0.0f, 0.0f);  // zero padding of both blocks
        }
      }
    }
    // request the next pair and this pair's delayed direct-bus samples now; they
    // are consumed after the ten passes below
    load_pair(tb + 2, z);
    float dre[EPT], dim[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      const int i = tid + e * NT;
      dre[e] = (i < B && have_re && tb >= first && !P.accumulate) ? delayed(tb * B + i) : 0.0f;
      dim[e] = (i < B && have_im && !P.accumulate) ? delayed((tb + 1) * B + i) : 0.0f;
    }
    cf *Z, *W;
    if constexpr (kTwInRegs) {
      // The last forward pass produces bins tid + r*L/4, exactly the four inputs of this
      // thread's butterfly in the first inverse pass: multiply by H and go on in registers.
      constexpr int NR4 = fft_r4_passes(L);
      cf *src = fft_run_passes_w<L, -1, NT>(b, a, twr, 1, NR4 - 1, tid);
      __syncthreads();
      cf o[4];
      r4_vals_w<-1>(src[tid], src[tid + L / 4], src[tid + L / 2], src[tid + 3 * L / 4], twr[NR4 - 2], o);
      W = src == a ? b : a;
      Z = src;
      r4_core<L, +1>(cf_mul(o[0], h[0][0]), cf_mul(o[1], h[0][1]), cf_mul(o[2], h[0][2]),
                     cf_mul(o[3], h[0][3]), W, tw, 1, tid);
    } else if constexpr (kFused2k) {
      constexpr int NR4 = fft_r4_passes(L);
      cf *src = fft_r4_range<L, -1, NT>(b, a, tw, 1, NR4, tid);
      __syncthreads();
      cf o[4][2];  // radix-2 pass (Ns = L/2): bins tid + 256 q and + 1024
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int j = tid + q * NT;
        const cf v0 = src[j], v1 = cf_mul(src[j + L / 2], tw[j]);
        o[q][0] = cf_add(v0, v1);
        o[q][1] = cf_sub(v0, v1);
      }
      W = src == a ? b : a;
      Z = src;
#pragma unroll
      for (int q = 0; q < 2; q++)  // first inverse butterfly tid + 256 q: bins j, j + 512, j + 1024, j + 1536
        r4_core<L, +1>(cf_mul(o[q][0], h[q][0]), cf_mul(o[q + 2][0], h[q][1]), cf_mul(o[q][1], h[q][2]),
                       cf_mul(o[q + 2][1], h[q][3]), W, tw, 1, tid + q * NT);
    } else {
      Z = fft_run_passes<L, -1, NT>(a, b, tw, 0, tid);
      W = Z == a ? b : a;
      __syncthreads();
      // first inverse pass with the spectral multiply folded into its loads
#pragma unroll
      for (int q = 0; q < BPT; q++) {
        const int j = tid + q * NT;
        if (j < L / 4)
          r4_core<L, +1>(cf_mul(Z[j], h[q][0]), cf_mul(Z[j + L / 4], h[q][1]),
                         cf_mul(Z[j + L / 2], h[q][2]), cf_mul(Z[j + 3 * L / 4], h[q][3]), W, tw,
                         1, j);
      }
    }
    // real part = block tb, imaginary part = block tb+1
    auto overlap_add = [&](int e, cf y_lo, cf y_hi) {  // y[i], y[B + i] of sample i = tid + e*NT
      const int i = tid + e * NT;
      float tl = tail[i];
      if (have_re) {
        if (tb >= first) {
          const float dec = (y_lo.x + tl) * norm;  // :223-226
          out[tb * B + i] = P.accumulate ? out[tb * B + i] + dec : dec + dre[e];
        }
        tl = y_hi.x;  // :224
      }
      if (have_im) {
        const float dec = (y_lo.y + tl) * norm;
        out[(tb + 1) * B + i] = P.accumulate ? out[(tb + 1) * B + i] + dec : dec + dim[e];
        tl = y_hi.y;
      }
      tail[i] = tl;
    };
    if constexpr (kTwInRegs) {
      // ... and the last inverse pass hands its four results (samples tid, tid + 256 and
      // their tails at + B) to the overlap-add in registers instead of through LDS
      constexpr int NR4 = fft_r4_passes(L);
      cf *src = fft_run_passes_w<L, +1, NT>(W, Z, twr, 1, NR4 - 1, tid);
      __syncthreads();
      cf o[4];
      r4_vals_w<+1>(src[tid], src[tid + L / 4], src[tid + L / 2], src[tid + 3 * L / 4], twr[NR4 - 2], o);
      overlap_add(0, o[0], o[2]);
      overlap_add(1, o[1], o[3]);
    } else if constexpr (kFused2k) {
      constexpr int NR4 = fft_r4_passes(L);
      cf *src = fft_r4_range<L, +1, NT>(W, Z, tw, 1, NR4, tid);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; q++) {  // last (radix-2) pass: samples i = tid + 256 q and their tails at + B
        const int j = tid + q * NT;
        const cf v0 = src[j], v1 = cf_mul(src[j + L / 2], cf_conj(tw[j]));
        overlap_add(q, cf_add(v0, v1), cf_sub(v0, v1));
      }
    } else {
      cf *y = fft_run_passes<L, +1, NT>(W, Z, tw, 1, tid);
      __syncthreads();
#pragma unroll
      for (int e = 0; e < EPT; e++) {
        const int i = tid + e * NT;
        if (i >= B) continue;
        overlap_add(e, y[i], y[B + i]);
      }
    }
  }

  if (last == P.T) {  // this workgroup owns the end of the call: publish the state
    for (int i = tid; i < B; i += NT) P.tail_out[(size_t)n * B + i] = tail[i];
    const int total = P.T * B;
    if (!P.accumulate) {
      for (int j = tid; j < P.D; j += NT) P.dly_out[(size_t)n * P.D + j] = delayed(total + j);
      // the last hist_len diffuse samples (history | this call) for the next call's later partitions
      for (int j = tid; j < P.hist_len; j += NT) {
        const int sd = total - P.hist_len + j;
        P.hist_out[(size_t)n * P.hist_len + j] =
            sd >= 0 ? bus_at(diffuse, sd) : P.hist_in[(size_t)n * P.hist_len + (sd + P.hist_len)];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// K2 for block sizes without an instantiation above (anything that is not a power of two: 480, 960,
// 1920 ...): the same run / pair / overlap-add structure as k_decorrelate_delay_mix with the transform
// size taken from an FftShape (mixed radix, fft_lds.h), the spectral multiply as its own sweep and no
// register staging — the general path, not a tuned one.  Dynamic LDS: 2 L complex + B floats.
static __global__ void __launch_bounds__(256) k_decorrelate_delay_mix_rt(DecorParams P, FftShape S) {
  constexpr int NT = 256;
  const int L = S.L, B = L / 2;
  cf *a = reinterpret_cast<cf *>(fft_dyn_lds), *b = a + L;
  float *tail = reinterpret_cast<float *>(b + L);
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int first = blockIdx.x * P.R;
  const int last = min(first + P.R, P.T);
  const float norm = 1.0f / (float)(2 * B);  // block_convolver_impl.cpp:212
  const cf *H = P.H + (size_t)n * L;
  const float *direct = P.bus + (size_t)n * P.bus_stride;
  const float *diffuse = P.bus + (size_t)(P.N + n) * P.bus_stride;
  float *out = P.out + (size_t)n * P.out_stride;
  auto bus_at = [&](const float *row, int s) {
    const float *q = row + s;
    float v = q[0];
    for (int p = 1; p < P.nparts; p++) v += q[(size_t)p * P.part_stride];
    return v;
  };
  auto delayed = [&](int s) {  // direct bus delayed by D (may reach back into the state)
    const int sd = s - P.D;
    return sd >= 0 ? bus_at(direct, sd) : P.dly_in[(size_t)n * P.D + (sd + P.D)];
  };
  auto diffuse_at = [&](int s) {  // diffuse bus delayed by this partition's shift
    const int sd = s - P.shift;
    return sd >= 0 ? bus_at(diffuse, sd) : P.hist_in[(size_t)n * P.hist_len + (sd + P.hist_len)];
  };
  for (int i = tid; i < B; i += NT) tail[i] = first == 0 ? P.tail_in[(size_t)n * B + i] : 0.0f;

  // pairs of blocks through one complex transform: real part = block tb, imaginary part = block tb + 1
  // (H is exactly Hermitian); the pair before `first` only rebuilds the overlap-add tail
  for (int tb = first - 1; tb < last; tb += 2) {
    const bool have_re = tb >= 0, have_im = tb + 1 < last;
    __syncthreads();  // the previous pair's reads of a / b and tail are finished
    for (int i = tid; i < B; i += NT) {
      a[i] = cf_make(have_re ? diffuse_at(tb * B + i) : 0.0f, have_im ? diffuse_at((tb + 1) * B + i) : 0.0f);
      a[i + B] = cf_make(0.0f, 0.0f);
    }
    cf *Z = fft_run_shape<-1, NT>(a, b, P.tw, S, tid);
    __syncthreads();
    for (int i = tid; i < L; i += NT) Z[i] = cf_mul(Z[i], H[i]);
    cf *y = fft_run_shape<+1, NT>(Z, Z == a ? b : a, P.tw, S, tid);
    __syncthreads();
    for (int i = tid; i < B; i += NT) {
      const cf y_lo = y[i], y_hi = y[B + i];
      float tl = tail[i];
      if (have_re) {
        if (tb >= first) {
          const float dec = (y_lo.x + tl) * norm;  // :223-226
          out[tb * B + i] = P.accumulate ? out[tb * B + i] + dec : dec + delayed(tb * B + i);
        }
        tl = y_hi.x;  // :224
      }
      if (have_im) {
        const float dec = (y_lo.y + tl) * norm;
        out[(tb + 1) * B + i] = P.accumulate ? out[(tb + 1) * B + i] + dec : dec + delayed((tb + 1) * B + i);
        tl = y_hi.y;
      }
      tail[i] = tl;
    }
  }

  if (last == P.T) {  // this workgroup owns the end of the call: publish the state
    __syncthreads();
    for (int i = tid; i < B; i += NT) P.tail_out[(size_t)n * B + i] = tail[i];
    const int total = P.T * B;
    if (!P.accumulate) {
      for (int j = tid; j < P.D; j += NT) P.dly_out[(size_t)n * P.D + j] = delayed(total + j);
      for (int j = tid; j < P.hist_len; j += NT) {
        const int sd = total - P.hist_len + j;
        P.hist_out[(size_t)n * P.hist_len + j] =
            sd >= 0 ? bus_at(diffuse, sd) : P.hist_in[(size_t)n * P.hist_len + (sd + P.hist_len)];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// K2 for 512-sample blocks, one WAVE per (loudspeaker, run of blocks): the 1024-point
// transforms are Stockham passes of radix 16, 16 and 4 with 16 values per lane (indices
// lane + 64 m — the same layout on input and output, so forward, x H and inverse chain
// in registers), exchanged twice per transform through wave-private LDS.  No workgroup
// barriers, 4 instead of 8 LDS round trips per pair of blocks, and the overlap-add tail
// stays in registers.  (tools/experiments/wavefft.hip: 1.0e-7 from a double DFT.)
//
// Complex values live in aligned VGPR pairs (v2f) and every complex operation is ONE or TWO packed-f32
// instructions whose op_sel / neg modifiers do the swaps and sign changes (left to itself the
// compiler pairs real parts of different values and spends a v_mov per operand on it).
typedef float v2f __attribute__((ext_vector_type(2)));

// a * w (DIR < 0) or a * conj(w) (DIR > 0): t = (ai wi, ai wr); r = (ar wr -/+ t.lo, +/- ar wi + t.hi)
template <int DIR>
__device__ __forceinline__ v2f wave_cmul(v2f a, v2f w) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
  if (DIR < 0)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
// the same with a wave-uniform constant w held in an SGPR pair
template <int DIR>
__device__ __forceinline__ v2f wave_cmul_k(v2f a, v2f w) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
  if (DIR < 0)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
        : "=v"(r) : "v"(a), "s"(w), "v"(t));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=v"(r) : "v"(a), "s"(w), "v"(t));
  return r;
}
__device__ __forceinline__ v2f wave_add_i(v2f a, v2f t) {  // a + i t = (a.x - t.y, a.y + t.x)
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(t));
  return r;
}
__device__ __forceinline__ v2f wave_sub_i(v2f a, v2f t) {  // a - i t = (a.x + t.y, a.y - t.x)
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(t));
  return r;
}
// radix-4 butterfly; the forward transform multiplies the odd difference by -i, the inverse by +i
template <int DIR>
__device__ __forceinline__ void wave_r4(v2f &a, v2f &b, v2f &c, v2f &d) {
  const v2f a0 = a + c, a1 = a - c, a2 = b + d, t = b - d;
  a = a0 + a2;
  c = a0 - a2;
  b = DIR < 0 ? wave_sub_i(a1, t) : wave_add_i(a1, t);
  d = DIR < 0 ? wave_add_i(a1, t) : wave_sub_i(a1, t);
}

// 16-point DFT in registers, natural order in and out (n = 4 n1 + n2, k = k1 + 4 k2)
template <int DIR>
__device__ __forceinline__ void wave_dft16(v2f (&v)[16]) {
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r2 = 0.70710678118654752f;
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) wave_r4<DIR>(v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]);
  const v2f W1 = {c1, -s1}, W2 = {r2, -r2}, W3 = {s1, -c1}, W4 = {0.0f, -1.0f}, W6 = {-r2, -r2}, W9 = {-c1, s1};
  v[5] = wave_cmul_k<DIR>(v[5], W1);  // W16^{n2 k1}
  v[6] = wave_cmul_k<DIR>(v[6], W2);
  v[7] = wave_cmul_k<DIR>(v[7], W3);
  v[9] = wave_cmul_k<DIR>(v[9], W2);
  v[10] = wave_cmul_k<DIR>(v[10], W4);
  v[11] = wave_cmul_k<DIR>(v[11], W6);
  v[13] = wave_cmul_k<DIR>(v[13], W3);
  v[14] = wave_cmul_k<DIR>(v[14], W6);
  v[15] = wave_cmul_k<DIR>(v[15], W9);
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) wave_r4<DIR>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
  v2f t;
#define EARHIP_SW(a, b) t = v[a], v[a] = v[b], v[b] = t;
  EARHIP_SW(1, 4) EARHIP_SW(2, 8) EARHIP_SW(3, 12) EARHIP_SW(6, 9) EARHIP_SW(7, 13) EARHIP_SW(11, 14)
#undef EARHIP_SW
}

// ---- Round 6: the same transforms in fewer instructions (the kernel is VALU-bound at three waves per SIMD: 3100 of the 3459
// SIMD cycles a pair of transforms takes are its packed-f32 instructions, profiles/r05_k2_transform_cycles.txt).  Three
// moves, all algebra on the butterflies — the spectral multiply, the overlap-add and their order stay libear's
// (block_convolver_impl.cpp:194-226):
//   * a twiddle multiply that feeds a butterfly's sum is folded into it: a + w c is two fused multiply-adds on a (cmul_acc)
//     instead of a complex multiply (two instructions) and an addition, and the matching difference a - w c = 2 a - (a + w c)
//     is ONE more instead of another subtraction of a product that no longer exists on its own;
//   * the constant rotations of the 16-point kernel are not multiplied out: x (-+i) is a pair of operand modifiers on the
//     butterfly that takes it, x r2 (1 -+ i) one packed addition (x -+ i x) and the factor r2 inside the butterfly's fma;
//   * the forward transform's inputs are zero padded to twice the block (Filter / BlockConvolver: 2 B points for B samples): the
//     first radix-4 stage of its first 16-point kernel sees c = d = 0 and is four additions instead of eight.
// 562 -> 490 packed instructions in the kernel (static count), K2 0.0531 -> 0.0508 ms on the headline, 0.1615 -> 0.152 on config 3
// (same box, two runs each).  NOT THE DEFAULT: built this way, tests/test_gpu_render.py::test_contexts_on_concurrent_threads —
// four contexts rendering at once on four threads — gave one wrong block of one loudspeaker (2-4 % off) in 5 of 110 fresh
// processes, the round-5 form built from the same tree in 0 of 78; every other test (fixed inputs, one stream) passes, the
// instruction stream shows no unhandled hazard (every dependent pair of packed instructions has its s_nop), and a deterministic
// kernel that is wrong once in twenty runs is not shipped for 4 % of K2.  -DEARHIP_K2_FOLDED=1 builds it (NOTES.md, round 6).
#ifndef EARHIP_K2_ABL
#define EARHIP_K2_ABL 0  // (timing-only ablations of k_decorrelate_wave, NOTES round 6: 1 no output stores, 2 no bus loads, 3 neither)
#endif
#ifndef EARHIP_K2_FOLDED
#define EARHIP_K2_FOLDED 0
#endif
// acc + a w (DIR < 0) or acc + a conj(w) (DIR > 0): t = acc + (-+ a.y w.y, a.y w.x); r = t + (a.x w.x, +- a.x w.y)
template <int DIR>
__device__ __forceinline__ v2f wave_cmul_acc(v2f a, v2f w, v2f acc) {
  v2f t, r;
  if (DIR < 0) {
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(t) : "v"(a), "v"(w), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  } else {
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a), "v"(w), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  }
  return r;
}
template <int DIR>
__device__ __forceinline__ v2f wave_cmul_acc_k(v2f a, v2f w, v2f acc) {  // the same with a wave-uniform w in an SGPR pair
  v2f t, r;
  if (DIR < 0) {
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(t) : "v"(a), "s"(w), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
  } else {
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a), "s"(w), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
  }
  return r;
}
__device__ __forceinline__ v2f wave_twice_minus(v2f a, v2f b) {  // 2 a - b
  const v2f two = {2.0f, 2.0f};
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "s"(two), "v"(b));
  return r;
}
__device__ __forceinline__ v2f wave_fma_k(v2f u, v2f k, v2f a) {  // a + k u (k: the same constant in both halves, SGPR pair)
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(u), "s"(k), "v"(a));
  return r;
}
__device__ __forceinline__ v2f wave_fnma_k(v2f u, v2f k, v2f a) {  // a - k u
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(u), "s"(k), "v"(a));
  return r;
}
__device__ __forceinline__ v2f wave_fma_k_subi(v2f q, v2f k, v2f a) {  // a - i k q = (a.x + k q.y, a.y - k q.x)
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(q), "s"(k), "v"(a));
  return r;
}
__device__ __forceinline__ v2f wave_fma_k_addi(v2f q, v2f k, v2f a) {  // a + i k q = (a.x - k q.y, a.y + k q.x)
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(q), "s"(k), "v"(a));
  return r;
}
// the outputs of a radix-4 butterfly from its sums: a0 = a + c, a1 = a - c, a2 = b + d, t = b - d
template <int DIR>
__device__ __forceinline__ void wave_r4_out(v2f a0, v2f a1, v2f a2, v2f t, v2f &a, v2f &b, v2f &c, v2f &d) {
  a = a0 + a2;
  c = a0 - a2;
  b = DIR < 0 ? wave_sub_i(a1, t) : wave_add_i(a1, t);
  d = DIR < 0 ? wave_add_i(a1, t) : wave_sub_i(a1, t);
}
// radix-4 butterfly of w_a a, w_b b, w_c c, w_d d (TA: a carries a twiddle too): 14 (12) instructions instead of 16 (14)
template <int DIR, bool TA>
__device__ __forceinline__ void wave_r4_tw(v2f &a, v2f &b, v2f &c, v2f &d, v2f wa, v2f wb, v2f wc, v2f wd) {
  const v2f pa = TA ? wave_cmul<DIR>(a, wa) : a;
  const v2f a0 = wave_cmul_acc<DIR>(c, wc, pa), a1 = wave_twice_minus(pa, a0);
  const v2f pb = wave_cmul<DIR>(b, wb);
  const v2f a2 = wave_cmul_acc<DIR>(d, wd, pb), t = wave_twice_minus(pb, a2);
  wave_r4_out<DIR>(a0, a1, a2, t, a, b, c, d);
}
// 16-point DFT in registers, natural order in and out, as wave_dft16 with the constant rotations folded into the second
// stage's butterflies.  HALF: v[8..15] are zero on entry (the forward transform's padding).  STAGE1: the first stage is done
// here (false: the caller did it, with its own twiddles folded in)
template <int DIR, bool HALF, bool STAGE1>
__device__ __forceinline__ void wave_dft16_f(v2f (&v)[16]) {
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r2 = 0.70710678118654752f;
  if (STAGE1) {
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) {
      if (HALF) {  // c = d = 0: a0 = a1 = a, a2 = t = b
        const v2f a = v[n2], b = v[n2 + 4];
        v[n2] = a + b;
        v[n2 + 8] = a - b;
        v[n2 + 4] = DIR < 0 ? wave_sub_i(a, b) : wave_add_i(a, b);
        v[n2 + 12] = DIR < 0 ? wave_add_i(a, b) : wave_sub_i(a, b);
      } else {
        wave_r4<DIR>(v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]);
      }
    }
  }
  const v2f W1 = {c1, -s1}, W3 = {s1, -c1}, W9 = {-c1, s1}, R2 = {r2, r2};
  // k1 = 0: no twiddles
  wave_r4<DIR>(v[0], v[1], v[2], v[3]);
  {  // k1 = 1: b = W1 v5, c = W2 v6 = r2 (1 -+ i) v6, d = W3 v7
    const v2f u = DIR < 0 ? wave_sub_i(v[6], v[6]) : wave_add_i(v[6], v[6]);
    const v2f a0 = wave_fma_k(u, R2, v[4]), a1 = wave_fnma_k(u, R2, v[4]);
    const v2f pb = wave_cmul_k<DIR>(v[5], W1);
    const v2f a2 = wave_cmul_acc_k<DIR>(v[7], W3, pb), t = wave_twice_minus(pb, a2);
    wave_r4_out<DIR>(a0, a1, a2, t, v[4], v[5], v[6], v[7]);
  }
  {  // k1 = 2: b = W2 v9 = r2 (1 -+ i) v9, c = W4 v10 = -+i v10, d = W6 v11 = -r2 (1 +- i) v11
    const v2f u9 = DIR < 0 ? wave_sub_i(v[9], v[9]) : wave_add_i(v[9], v[9]);
    const v2f u11 = DIR < 0 ? wave_add_i(v[11], v[11]) : wave_sub_i(v[11], v[11]);
    const v2f a0 = DIR < 0 ? wave_sub_i(v[8], v[10]) : wave_add_i(v[8], v[10]);
    const v2f a1 = DIR < 0 ? wave_add_i(v[8], v[10]) : wave_sub_i(v[8], v[10]);
    const v2f sm = u9 - u11, q = u9 + u11;  // a2 = r2 sm, t = r2 q
    v[8] = wave_fma_k(sm, R2, a0);
    v[10] = wave_fnma_k(sm, R2, a0);
    v[9] = DIR < 0 ? wave_fma_k_subi(q, R2, a1) : wave_fma_k_addi(q, R2, a1);
    v[11] = DIR < 0 ? wave_fma_k_addi(q, R2, a1) : wave_fma_k_subi(q, R2, a1);
  }
  {  // k1 = 3: b = W3 v13, c = W6 v14 = -r2 (1 +- i) v14, d = W9 v15
    const v2f u = DIR < 0 ? wave_add_i(v[14], v[14]) : wave_sub_i(v[14], v[14]);
    const v2f a0 = wave_fnma_k(u, R2, v[12]), a1 = wave_fma_k(u, R2, v[12]);
    const v2f pb = wave_cmul_k<DIR>(v[13], W3);
    const v2f a2 = wave_cmul_acc_k<DIR>(v[15], W9, pb), t = wave_twice_minus(pb, a2);
    wave_r4_out<DIR>(a0, a1, a2, t, v[12], v[13], v[14], v[15]);
  }
  v2f t;
#define EARHIP_SW(a, b) t = v[a], v[a] = v[b], v[b] = t;
  EARHIP_SW(1, 4) EARHIP_SW(2, 8) EARHIP_SW(3, 12) EARHIP_SW(6, 9) EARHIP_SW(7, 13) EARHIP_SW(11, 14)
#undef EARHIP_SW
}

__device__ __forceinline__ int wave_pad(int i) { return i + (i >> 4); }  // 17-word rows: conflict-free exchanges

// v[m] <-> index lane + 64 m.  t1[r-1] = W256^{r (lane & 15)}, t2[256 (r-1) + j] = W1024^{r j} (both in LDS).
template <int DIR, bool HALF = false>
__device__ __forceinline__ void wave_fft1024(v2f (&v)[16], v2f *lds, const v2f *t1, const v2f *t2, int lane) {
#if EARHIP_K2_FOLDED
  wave_dft16_f<DIR, HALF, true>(v);  // radix 16, Ns = 1: out[16 lane + r]
#pragma unroll
  for (int r = 0; r < 16; r++) lds[wave_pad(16 * lane + r)] = v[r];
#pragma unroll
  for (int r = 0; r < 16; r++) v[r] = lds[wave_pad(lane + 64 * r)];  // radix 16, Ns = 16
  __builtin_amdgcn_sched_barrier(0);  // (keeps the twiddle reads from being hoisted: register pressure)
  // the twiddles W256^{r (lane & 15)} folded into the first stage of the second 16-point kernel (v[0] has none)
  wave_r4_tw<DIR, false>(v[0], v[4], v[8], v[12], v[0], t1[3], t1[7], t1[11]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int n2 = 1; n2 < 4; n2++) {
    wave_r4_tw<DIR, true>(v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12], t1[n2 - 1], t1[n2 + 3], t1[n2 + 7], t1[n2 + 11]);
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_dft16_f<DIR, false, false>(v);
  __builtin_amdgcn_sched_barrier(0);
  const int ob = (lane >> 4) * 256 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 16; r++) lds[wave_pad(ob + 16 * r)] = v[r];
#pragma unroll
  for (int q = 0; q < 4; q++)  // radix 4, Ns = 256: butterfly lane + 64 q
#pragma unroll
    for (int r = 0; r < 4; r++) v[q + 4 * r] = lds[wave_pad(lane + 64 * q + 256 * r)];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int j = lane + 64 * q;
    wave_r4_tw<DIR, false>(v[q], v[q + 4], v[q + 8], v[q + 12], v[q], t2[j], t2[256 + j], t2[512 + j]);
    __builtin_amdgcn_sched_barrier(0);
  }
#else
  wave_dft16<DIR>(v);  // radix 16, Ns = 1: out[16 lane + r]
#pragma unroll
  for (int r = 0; r < 16; r++) lds[wave_pad(16 * lane + r)] = v[r];
#pragma unroll
  for (int r = 0; r < 16; r++) v[r] = lds[wave_pad(lane + 64 * r)];  // radix 16, Ns = 16
  __builtin_amdgcn_sched_barrier(0);  // (keeps the twiddle reads from being hoisted: register pressure)
#pragma unroll
  for (int r = 1; r < 16; r++) {
    v[r] = wave_cmul<DIR>(v[r], t1[r - 1]);
    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  wave_dft16<DIR>(v);
  __builtin_amdgcn_sched_barrier(0);
  const int ob = (lane >> 4) * 256 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 16; r++) lds[wave_pad(ob + 16 * r)] = v[r];
#pragma unroll
  for (int q = 0; q < 4; q++)  // radix 4, Ns = 256: butterfly lane + 64 q
#pragma unroll
    for (int r = 0; r < 4; r++) v[q + 4 * r] = lds[wave_pad(lane + 64 * q + 256 * r)];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 4; q++) {
#pragma unroll
    for (int r = 1; r < 4; r++) v[q + 4 * r] = wave_cmul<DIR>(v[q + 4 * r], t2[256 * (r - 1) + lane + 64 * q]);
    wave_r4<DIR>(v[q], v[q + 4], v[q + 8], v[q + 12]);
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
}

constexpr int kDecorWaves = 4;  // waves (= runs) per workgroup of k_decorrelate_wave

// -DEARHIP_K2_PROF: wave 0 of two workgroups (the first and one from the middle of the grid) leaves s_memtime stamps at the
// kernel's phase boundaries (earhip_debug_k2_prof reads them): a diagnostic build, the marks cost the kernel a few per cent
#ifdef EARHIP_K2_PROF
static __device__ unsigned long long g_k2_prof[2][32];
#define EARHIP_K2_MARK(i)                                                                        \
  do {                                                                                           \
    if (prof_slot >= 0 && lane == 0 && (i) < 32) g_k2_prof[prof_slot][(i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define EARHIP_K2_MARK(i) do {} while (0)
#endif

// grid = (ceil(runs / kDecorWaves), N); same parameters and semantics as k_decorrelate_delay_mix<1024>
__global__ void __launch_bounds__(64 * kDecorWaves) __attribute__((amdgpu_waves_per_eu(3, 3))) k_decorrelate_wave(DecorParams P) {
  constexpr int L = 1024, B = 512;
  __shared__ v2f lds_all[kDecorWaves][L + L / 16];
  __shared__ v2f h_lds[L + L / 16];   // the loudspeaker's spectrum (all runs of a workgroup share it)
  __shared__ v2f t1_lds[16][17];    // W256^{r k}: row k = lane & 15, column r - 1 (r = 1..15)
  __shared__ v2f t2_lds[3 * 256];    // W1024^{r j}: [256 (r - 1) + j], r = 1..3, j < 256
  // (the wave index as a SCALAR: the run's blocks, its bus and output rows and every branch on them are then wave-uniform values
  // in SGPRs.  Round 6's phase stamps — tools/k2_phases.py — showed half of a pair's time BETWEEN its transforms: with `w` a lane
  // value the compiler kept 64-bit per-lane row pointers, spilled the output pointer, and the reload in front of a block's stores —
  // a vector-memory operation on the in-order counter — made every block's stores wait for the block before them to be
  // acknowledged by memory: `s_waitcnt vmcnt(0)` behind a scratch_load, twice per pair)
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f *lds = lds_all[w];
  auto to_v2f = [](cf c) { return v2f{c.x, c.y}; };
  const int n = blockIdx.y;
#ifdef EARHIP_K2_PROF
  const int prof_slot = w != 0 ? -1 : (blockIdx.x == 0 && blockIdx.y == 0) ? 0 : (blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2) ? 1 : -1;
  int prof_i = 2;
#endif
  EARHIP_K2_MARK(0);
  for (int i = threadIdx.x; i < L; i += 64 * kDecorWaves) h_lds[wave_pad(i)] = to_v2f(P.H[(size_t)n * L + i]);
  if (threadIdx.x < 240) {
    const int k = threadIdx.x / 15, r = threadIdx.x % 15 + 1;
    t1_lds[k][r - 1] = to_v2f(P.tw[(4 * r * k) & (L - 1)]);
  }
  for (int i = threadIdx.x; i < 3 * 256; i += 64 * kDecorWaves) t2_lds[i] = to_v2f(P.tw[((i >> 8) + 1) * (i & 255)]);
  __syncthreads();
  EARHIP_K2_MARK(1);
  const int first = (blockIdx.x * kDecorWaves + w) * P.R;
  if (first >= P.T) return;  // (no workgroup barriers below)
  const int last = min(first + P.R, P.T);
  const float norm = 1.0f / (float)(2 * B);  // block_convolver_impl.cpp:212
  const float *direct = P.bus + (size_t)n * P.bus_stride;
  const float *diffuse = P.bus + (size_t)(P.N + n) * P.bus_stride;
  float *out = P.out + (size_t)n * P.out_stride;

  const v2f *t1 = t1_lds[lane & 15];
  const v2f *t2 = t2_lds;

  auto bus_at = [&](const float *row, int s) {
    const float *q = row + s;
    float v = q[0];
    for (int p = 1; p < P.nparts; p++) v += q[(size_t)p * P.part_stride];
    return v;
  };
  // acc[m] = bus sample s0 + lane + 64 m summed over the object splits (8 loads in flight per split)
  auto bus_at8 = [&](const float *row, int s0, float (&acc)[8]) {
    const float *q = row + s0 + lane;
#if EARHIP_K2_ABL >= 2
#pragma unroll
    for (int m = 0; m < 8; m++) acc[m] = (float)(lane - m) * 1e-3f;
    return;
#endif
#pragma unroll
    for (int m = 0; m < 8; m++) acc[m] = __builtin_nontemporal_load(q + 64 * m);  // (read once)
    for (int p = 1; p < P.nparts; p++) {
      q += P.part_stride;
#pragma unroll
      for (int m = 0; m < 8; m++) acc[m] += __builtin_nontemporal_load(q + 64 * m);
    }
  };
  auto delayed = [&](int s) {  // direct bus delayed by D
    const int sd = s - P.D;
    return sd >= 0 ? bus_at(direct, sd) : P.dly_in[(size_t)n * P.D + (sd + P.D)];
  };
  auto delayed8 = [&](int tb, float (&acc)[8]) {  // block tb of the delayed direct bus
    if (tb * B >= P.D) {
      bus_at8(direct, tb * B - P.D, acc);
    } else {  // first block of a call: the head comes from the delay state
      const int s0 = tb * B - P.D + lane;
      const float *q = direct + s0;
#pragma unroll
      for (int m = 0; m < 8; m++)
        acc[m] = s0 + 64 * m < 0 ? P.dly_in[(size_t)n * P.D + (s0 + 64 * m + P.D)] : q[64 * m];
      for (int p = 1; p < P.nparts; p++) {
        q += P.part_stride;
#pragma unroll
        for (int m = 0; m < 8; m++)
          if (s0 + 64 * m >= 0) acc[m] += q[64 * m];
      }
    }
  };

  float tl[8];  // un-normalised overlap-add tail of the previous block, samples lane + 64 m
#pragma unroll
  for (int m = 0; m < 8; m++) tl[m] = first == 0 ? P.tail_in[(size_t)n * B + lane + 64 * m] : 0.0f;

  // real part = diffuse block tb, imaginary part = block tb+1
  // (straight into the register pairs the transform takes them in — real part block tb, imaginary part block tb + 1 —: loaded into
  // two float arrays, the compiler interleaved them with moves right behind the requests, i.e. waited for them at once)
  auto load_half = [&](int tb, v2f (&z)[8], auto im_tag) {
    constexpr bool IM = decltype(im_tag)::value;
    const float *q = diffuse + tb * B + lane;
#pragma unroll
    for (int m = 0; m < 8; m++) {
#if EARHIP_K2_ABL >= 2
      const float x = (float)(lane + m) * 1e-3f;  // (timing only: no bus loads)
#else
      const float x = __builtin_nontemporal_load(q + 64 * m);  // (read once)
#endif
      if (IM) z[m].y = x;
      else z[m].x = x;
    }
    for (int p = 1; p < P.nparts; p++) {
      q += P.part_stride;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const float x = __builtin_nontemporal_load(q + 64 * m);
        if (IM) z[m].y += x;
        else z[m].x += x;
      }
    }
  };
  auto load_pair = [&](int tb, v2f (&z)[8]) {
#pragma unroll
    for (int m = 0; m < 8; m++) z[m] = v2f{0.0f, 0.0f};
    if (tb >= 0 && tb < last) load_half(tb, z, std::false_type{});
    if (tb + 1 < last) load_half(tb + 1, z, std::true_type{});
  };
  // The vector-memory counter is IN ORDER and counts stores: whatever is waited for behind a block's stores waits for those
  // stores to be acknowledged by memory.  So a pair's iteration is laid out as: requests (the next pair's inputs, this pair's
  // delayed direct samples) -> transforms -> everything that consumes a loaded value (the outputs, made in the registers of the
  // delayed samples; the next pair's inputs moved into the transform registers) -> the stores, last.  (Round 5's order — stores,
  // then the next iteration picking up its prefetched inputs — waited for every pair's stores: `s_waitcnt vmcnt(0)` at the loop's
  // back edge, 5-10 k cycles of a pair's ~16 k: tools/k2_phases.py.)
  v2f v[16];
  {
    v2f z[8];
    load_pair(first - 1, z);
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = z[m];
  }
  for (int tb = first - 1; tb < last; tb += 2) {
    const bool have_re = tb >= 0, have_im = tb + 1 < last;
#pragma unroll
    for (int m = 0; m < 8; m++) v[m + 8] = v2f{0.0f, 0.0f};  // zero padding to 2 B
    // loads issued ahead of the transforms: the next pair, and this pair's delayed direct samples
    v2f z[8];
    load_pair(tb + 2, z);
    float dre[8], dim[8];
#pragma unroll
    for (int m = 0; m < 8; m++) dre[m] = dim[m] = 0.0f;
    if (have_re && tb >= first) delayed8(tb, dre);
    if (have_im) delayed8(tb + 1, dim);
#ifdef EARHIP_K2_PROF
    EARHIP_K2_MARK(prof_i);  // pair start: registers set, next pair's loads issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    EARHIP_K2_MARK(prof_i + 1);  // ... and arrived (a diagnostic wait: the kernel itself does not wait here)
#endif
    wave_fft1024<-1, true>(v, lds, t1, t2, lane);  // (v[8..15] = 0: the padding)
#ifdef EARHIP_K2_PROF
    asm volatile("" : "+v"(v[0]));
    EARHIP_K2_MARK(prof_i + 2);
#endif
#pragma unroll
    for (int m = 0; m < 16; m++) {
      v[m] = wave_cmul<-1>(v[m], h_lds[wave_pad(lane + 64 * m)]);
      if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
#ifdef EARHIP_K2_PROF
    asm volatile("" : "+v"(v[0]));
    EARHIP_K2_MARK(prof_i + 3);
#endif
    wave_fft1024<+1>(v, lds, t1, t2, lane);
#ifdef EARHIP_K2_PROF
    asm volatile("" : "+v"(v[0]));
    EARHIP_K2_MARK(prof_i + 4);
    prof_i += 5;
#endif
    // (the outputs are made whether the block exists or not — a block that does not is not stored —: a register that a load MAY
    // have been issued for has then been read on every path before the stores, and the compiler's wait-count pass, which merges
    // the paths conservatively, puts no wait in front of them)
#pragma unroll
    for (int m = 0; m < 8; m++) {  // :223-226
      dre[m] = (v[m].x + tl[m]) * norm + dre[m];
      asm volatile("" : "+v"(dre[m]) : : "memory");  // (made HERE, not sunk into the store's branch behind other stores)
      tl[m] = have_re ? v[m + 8].x : tl[m];  // :224
    }
#pragma unroll
    for (int m = 0; m < 8; m++) {
      dim[m] = (v[m].y + tl[m]) * norm + dim[m];
      asm volatile("" : "+v"(dim[m]) : : "memory");
      tl[m] = have_im ? v[m + 8].y : tl[m];
    }
#pragma unroll
    for (int m = 0; m < 8; m++) {  // the next pair's inputs (every load of this iteration has been waited for by now)
      v[m] = z[m];
      // (an opaque use, ordered against memory operations: left to itself the compiler coalesces v with the load's destination
      // and the first real use — hence the wait — lands in the NEXT iteration, behind the stores)
      asm volatile("" : "+v"(v[m]) : : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);  // nothing that reads a loaded value sinks below the stores
#if EARHIP_K2_ABL == 1 || EARHIP_K2_ABL == 3  // (timing-only ablation: no output stores — one store per wave and pair keeps the work alive)
    if (have_im && dre[0] + dim[0] + dre[7] + dim[7] == 123.456f) out[tb * B + lane] = dre[1] + dim[1];
#else
    if (have_re && tb >= first) {
#pragma unroll
      for (int m = 0; m < 8; m++) __builtin_nontemporal_store(dre[m], out + tb * B + lane + 64 * m);
    }
    if (have_im) {
#pragma unroll
      for (int m = 0; m < 8; m++) __builtin_nontemporal_store(dim[m], out + (tb + 1) * B + lane + 64 * m);
    }
#endif
  }
  EARHIP_K2_MARK(prof_i < 31 ? prof_i : 31);  // the run's last store issued
  if (last == P.T) {  // this wave owns the end of the call: publish the state
#pragma unroll
    for (int m = 0; m < 8; m++) P.tail_out[(size_t)n * B + lane + 64 * m] = tl[m];
    const int total = P.T * B;
    if (total >= P.D && P.D <= 256) {
      // the last D samples of the direct bus, four per lane, summed over the object splits with
      // 4 x 4 loads in flight (block mode: 16 splits, and this is on the call's critical path)
      const float *q = direct + (total - P.D) + lane;
      float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; k++) ok[k] = lane + 64 * k < P.D;
      int p = 0;
      for (; p + 3 < P.nparts; p += 4) {
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int k = 0; k < 4; k++) t[j][k] = ok[k] ? q[(size_t)(p + j) * P.part_stride + 64 * k] : 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) {  // same order as bus_at: part after part
          acc[k] = p == 0 ? t[0][k] : acc[k] + t[0][k];
          acc[k] += t[1][k];
          acc[k] += t[2][k];
          acc[k] += t[3][k];
        }
      }
      for (; p < P.nparts; p++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float v = ok[k] ? q[(size_t)p * P.part_stride + 64 * k] : 0.0f;
          acc[k] = p == 0 ? v : acc[k] + v;
        }
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (ok[k]) P.dly_out[(size_t)n * P.D + lane + 64 * k] = acc[k];
    } else {
      for (int j = lane; j < P.D; j += 64) P.dly_out[(size_t)n * P.D + j] = delayed(total + j);
    }
  }
}

// n_buses == 2 without decorrelators is not a libear configuration; n_buses == 1
// (direct bus only) never reaches this kernel: K1 writes the output directly.

}  // namespace earhip
