// fft_kernels.h — workgroup-level FFT kernels built on fft_lds.h:
//   k_spectrum_real   real rows (zero padded)        -> full complex spectra
//   k_ifft_real       half spectra (Hermitian ext.)  -> real rows, un-normalised
//   k_conv_forward    one BlockConvolver input block (optionally faded down/up,
//                     block_convolver_impl.cpp:127-141,162-187) -> half spectra
//   k_conv_mac        Y = sum_t H_t (.) X_t over the partition queue (:193-209)
//   k_conv_ifft_ola   inverse + overlap-add + tail update + 1/(2B) (:212-234)
//   k_delay           DelayBuffer::process (delay_buffer_impl.cpp:19-40)
// One 256-thread workgroup transforms one row entirely in LDS.  Every kernel comes as a template over a
// power-of-two size and as an _rt twin for any other size (FftShape: mixed radix, dynamic LDS of 2 L
// complex values) — libear's kissfft takes any block size (src/fft_kiss.cpp:104-107).
#pragma once

#include <hip/hip_runtime.h>

#include "fft_lds.h"

namespace earhip {

constexpr int kFftThreads = 256;

// in: [rows][in_stride] reals, first n_valid of each row used, rest = 0
// out: [rows][L] complex, natural order, un-normalised forward transform
template <int L>
__global__ void __launch_bounds__(kFftThreads)
k_spectrum_real(const float *in, size_t in_stride, int n_valid, const cf *tw, cf *out) {
  __shared__ __attribute__((aligned(16))) cf lds[2 * L];
  cf *a = lds, *b = lds + L;
  const int tid = threadIdx.x;
  const float *x = in + (size_t)blockIdx.x * in_stride;
  for (int i = tid; i < L; i += kFftThreads) a[i] = cf_make(i < n_valid ? x[i] : 0.0f, 0.0f);
  cf *res = fft_run_passes<L, -1, kFftThreads>(a, b, tw, 0, tid);
  __syncthreads();
  cf *o = out + (size_t)blockIdx.x * L;
  for (int i = tid; i < L; i += kFftThreads) o[i] = res[i];
}

// in: [rows][L/2+1] complex; out: [rows][L] real = Re(IDFT_L(Hermitian ext.)) * L
template <int L>
__global__ void __launch_bounds__(kFftThreads)
k_ifft_real(const cf *in, const cf *tw, float *out) {
  __shared__ __attribute__((aligned(16))) cf lds[2 * L];
  cf *a = lds, *b = lds + L;
  const int tid = threadIdx.x;
  const cf *X = in + (size_t)blockIdx.x * (L / 2 + 1);
  for (int i = tid; i < L; i += kFftThreads) a[i] = i <= L / 2 ? X[i] : cf_conj(X[L - i]);
  cf *res = fft_run_passes<L, +1, kFftThreads>(a, b, tw, 0, tid);
  __syncthreads();
  float *o = out + (size_t)blockIdx.x * L;
  for (int i = tid; i < L; i += kFftThreads) o[i] = res[i].x;
}

// BlockConvolver forward step.  x: B = L/2 samples.  fade == 0: X_new = r2c(pad(x)).
// fade != 0: X_old = r2c(pad(fade_down(x))), X_new = r2c(pad(fade_up(x))), both
// from ONE complex transform of z = down + i*up, untangled by Hermitian symmetry.
template <int L>
__global__ void __launch_bounds__(kFftThreads)
k_conv_forward(const float *x, int fade, const cf *tw, cf *X_old, cf *X_new) {
  __shared__ __attribute__((aligned(16))) cf lds[2 * L];
  constexpr int B = L / 2;
  cf *a = lds, *b = lds + L;
  const int tid = threadIdx.x;
  const float i_scale = 1.0f / (float)B;  // block_convolver_impl.cpp:133
  for (int i = tid; i < L; i += kFftThreads) {
    cf v = cf_make(0.0f, 0.0f);
    if (i < B) {
      const float s = x[i];
      if (fade) {
        const float av = (float)i * i_scale, bv = 1.0f - av;  // :136-139
        v = cf_make(bv * s, av * s);
      } else {
        v = cf_make(s, 0.0f);
      }
    }
    a[i] = v;
  }
  cf *Z = fft_run_passes<L, -1, kFftThreads>(a, b, tw, 0, tid);
  __syncthreads();
  for (int k = tid; k <= B; k += kFftThreads) {
    const cf zk = Z[k], zc = cf_conj(Z[(L - k) & (L - 1)]);
    if (fade) {
      X_old[k] = cf_make(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
      // (zk - zc) / (2i) = -i/2 * (zk - zc)
      X_new[k] = cf_make(0.5f * (zk.y - zc.y), -0.5f * (zk.x - zc.x));
    } else {
      X_new[k] = zk;
    }
  }
}

struct MacTerms {
  const cf *H[16];
  const cf *X[16];
  int n;
  int overwrite;  // first chunk: Y = sum, else Y += sum
};

// Y[k] (+)= sum_t H[t][k] * X[t][k], k in [0, nbins); terms in queue order
static __global__ void k_conv_mac(MacTerms T, int nbins, cf *Y) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nbins) return;
  cf acc = T.overwrite ? cf_make(0.0f, 0.0f) : Y[k];
  for (int t = 0; t < T.n; t++) acc = cf_add(acc, cf_mul(T.H[t][k], T.X[t][k]));
  Y[k] = acc;
}

// mode 0: y = c2r(Y); if (use_tail) y[0:B] += tail; tail = y[B:2B]; out = y[0:B]*norm
// mode 1: out = tail * norm (caller marks the tail as zero afterwards)
template <int L>
__global__ void __launch_bounds__(kFftThreads)
k_conv_ifft_ola(const cf *Y, const cf *tw, float *tail, int use_tail, int mode, float *out) {
  __shared__ __attribute__((aligned(16))) cf lds[2 * L];
  constexpr int B = L / 2;
  const int tid = threadIdx.x;
  const float norm = 1.0f / (float)(2 * B);  // block_convolver_impl.cpp:212
  if (mode == 1) {
    for (int i = tid; i < B; i += kFftThreads) out[i] = tail[i] * norm;
    return;
  }
  cf *a = lds, *b = lds + L;
  for (int i = tid; i < L; i += kFftThreads) a[i] = i <= B ? Y[i] : cf_conj(Y[L - i]);
  cf *y = fft_run_passes<L, +1, kFftThreads>(a, b, tw, 0, tid);
  __syncthreads();
  for (int i = tid; i < B; i += kFftThreads) {
    float v = y[i].x;
    if (use_tail) v = v + tail[i];
    tail[i] = y[B + i].x;
    out[i] = v * norm;
  }
}


// ---- run-time sizes (dynamic LDS: 2 L complex) ----------------------------------------
static __global__ void __launch_bounds__(kFftThreads)
k_spectrum_real_rt(FftShape S, const float *in, size_t in_stride, int n_valid, const cf *tw, cf *out) {
  const int L = S.L;
  cf *a = reinterpret_cast<cf *>(fft_dyn_lds), *b = a + L;
  const int tid = threadIdx.x;
  const float *x = in + (size_t)blockIdx.x * in_stride;
  for (int i = tid; i < L; i += kFftThreads) a[i] = cf_make(i < n_valid ? x[i] : 0.0f, 0.0f);
  cf *res = fft_run_shape<-1, kFftThreads>(a, b, tw, S, tid);
  __syncthreads();
  cf *o = out + (size_t)blockIdx.x * L;
  for (int i = tid; i < L; i += kFftThreads) o[i] = res[i];
}

static __global__ void __launch_bounds__(kFftThreads)
k_ifft_real_rt(FftShape S, const cf *in, const cf *tw, float *out) {
  const int L = S.L;
  cf *a = reinterpret_cast<cf *>(fft_dyn_lds), *b = a + L;
  const int tid = threadIdx.x;
  const cf *X = in + (size_t)blockIdx.x * (L / 2 + 1);
  for (int i = tid; i < L; i += kFftThreads) a[i] = i <= L / 2 ? X[i] : cf_conj(X[L - i]);
  cf *res = fft_run_shape<+1, kFftThreads>(a, b, tw, S, tid);
  __syncthreads();
  float *o = out + (size_t)blockIdx.x * L;
  for (int i = tid; i < L; i += kFftThreads) o[i] = res[i].x;
}

static __global__ void __launch_bounds__(kFftThreads)
k_conv_forward_rt(FftShape S, const float *x, int fade, const cf *tw, cf *X_old, cf *X_new) {
  const int L = S.L, B = L / 2;
  cf *a = reinterpret_cast<cf *>(fft_dyn_lds), *b = a + L;
  const int tid = threadIdx.x;
  const float i_scale = 1.0f / (float)B;  // block_convolver_impl.cpp:133
  for (int i = tid; i < L; i += kFftThreads) {
    cf v = cf_make(0.0f, 0.0f);
    if (i < B) {
      const float s = x[i];
      if (fade) {
        const float av = (float)i * i_scale, bv = 1.0f - av;  // :136-139
        v = cf_make(bv * s, av * s);
      } else {
        v = cf_make(s, 0.0f);
      }
    }
    a[i] = v;
  }
  cf *Z = fft_run_shape<-1, kFftThreads>(a, b, tw, S, tid);
  __syncthreads();
  for (int k = tid; k <= B; k += kFftThreads) {
    const cf zk = Z[k], zc = cf_conj(Z[k == 0 ? 0 : L - k]);
    if (fade) {
      X_old[k] = cf_make(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
      X_new[k] = cf_make(0.5f * (zk.y - zc.y), -0.5f * (zk.x - zc.x));
    } else {
      X_new[k] = zk;
    }
  }
}

static __global__ void __launch_bounds__(kFftThreads)
k_conv_ifft_ola_rt(FftShape S, const cf *Y, const cf *tw, float *tail, int use_tail, int mode, float *out) {
  const int L = S.L, B = L / 2;
  const int tid = threadIdx.x;
  const float norm = 1.0f / (float)(2 * B);  // block_convolver_impl.cpp:212
  if (mode == 1) {
    for (int i = tid; i < B; i += kFftThreads) out[i] = tail[i] * norm;
    return;
  }
  cf *a = reinterpret_cast<cf *>(fft_dyn_lds), *b = a + L;
  for (int i = tid; i < L; i += kFftThreads) a[i] = i <= B ? Y[i] : cf_conj(Y[L - i]);
  cf *y = fft_run_shape<+1, kFftThreads>(a, b, tw, S, tid);
  __syncthreads();
  for (int i = tid; i < B; i += kFftThreads) {
    float v = y[i].x;
    if (use_tail) v = v + tail[i];
    tail[i] = y[B + i].x;
    out[i] = v * norm;
  }
}

// dynamic LDS of an _rt kernel; above 64 KB the kernel has to be told once per process
template <typename K>
static size_t fft_rt_lds(K kernel, int L, size_t extra = 0) {
  const size_t bytes = 2 * (size_t)L * sizeof(cf) + extra;
  if (bytes > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bytes);
  return bytes;
}

// DelayBuffer: [mem; in] -> [out; mem'] per channel; mem double-buffered
static __global__ void k_delay(const float *in, size_t stride, int nsamples, int delay,
                        const float *mem_in, float *mem_out, float *out) {
  const int c = blockIdx.y;
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsamples + delay) return;
  const float v = s < delay ? mem_in[(size_t)c * delay + s]
                            : in[(size_t)c * stride + (s - delay)];
  if (s < nsamples) out[(size_t)c * stride + s] = v;
  else mem_out[(size_t)c * delay + (s - nsamples)] = v;
}

}  // namespace earhip
