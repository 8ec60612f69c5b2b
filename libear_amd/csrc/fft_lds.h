// fft_lds.h — power-of-two complex FFT passes for data held in LDS.
//
// Stockham auto-sort formulation: pass with radix R and Ns = product of the
// radices of the previous passes maps butterfly j in [0, L/R) as
//     v[r]   = in[j + r*L/R] * W_{Ns*R}^{r*(j mod Ns)}          r = 0..R-1
//     out[(j div Ns)*Ns*R + (j mod Ns) + r*Ns] = DFT_R(v)[r]
// so the output of the last pass is in natural order and no bit reversal is
// needed.  Reads are unit-stride across butterflies (conflict-free ds_read_b64).
// Twiddles come from one table tw[t] = exp(-2*pi*i*t/L), t in [0, L), computed
// in double on the host; the inverse transform conjugates them (DIR = +1).
//
// Everything here is __host__ __device__ so the index algebra is unit-tested
// on the CPU (tests/cpp/test_fft_passes.cpp) before it ever runs on a GPU.
#pragma once

#if defined(__HIPCC__)
#define EARHIP_HD __host__ __device__ __forceinline__
#else
#define EARHIP_HD inline
#endif

namespace earhip {

struct cf {
  float x, y;
};

EARHIP_HD cf cf_make(float x, float y) {
  cf r;
  r.x = x;
  r.y = y;
  return r;
}
EARHIP_HD cf cf_add(cf a, cf b) { return cf_make(a.x + b.x, a.y + b.y); }
EARHIP_HD cf cf_sub(cf a, cf b) { return cf_make(a.x - b.x, a.y - b.y); }
EARHIP_HD cf cf_mul(cf a, cf b) {
  return cf_make(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
EARHIP_HD cf cf_conj(cf a) { return cf_make(a.x, -a.y); }
// a * (DIR * i): forward butterflies need -i, inverse +i
template <int DIR>
EARHIP_HD cf cf_mul_i(cf a) {
  return DIR > 0 ? cf_make(-a.y, a.x) : cf_make(a.y, -a.x);
}
template <int DIR>
EARHIP_HD cf tw_load(const cf *tw, int t) {
  const cf w = tw[t];
  return DIR > 0 ? cf_conj(w) : w;
}

// DFT_4 of (v0..v3) after twiddling, scattered with stride Ns
template <int L, int DIR>
EARHIP_HD void r4_core(cf v0, cf v1, cf v2, cf v3, cf *out, const cf *tw, int Ns,
                       int j) {
  const int k = j & (Ns - 1);
  if (Ns > 1) {
    const int step = L / (Ns * 4);
    v1 = cf_mul(v1, tw_load<DIR>(tw, k * step));
    v2 = cf_mul(v2, tw_load<DIR>(tw, 2 * k * step));
    v3 = cf_mul(v3, tw_load<DIR>(tw, 3 * k * step));
  }
  const cf a0 = cf_add(v0, v2), a1 = cf_sub(v0, v2);
  const cf a2 = cf_add(v1, v3), a3 = cf_mul_i<DIR>(cf_sub(v1, v3));
  const int base = ((j - k) << 2) + k;
  out[base] = cf_add(a0, a2);
  out[base + Ns] = cf_add(a1, a3);
  out[base + 2 * Ns] = cf_sub(a0, a2);
  out[base + 3 * Ns] = cf_sub(a1, a3);
}

// one radix-4 butterfly of a pass, j in [0, L/4)
template <int L, int DIR>
EARHIP_HD void stockham_r4(const cf *in, cf *out, const cf *tw, int Ns, int j) {
  r4_core<L, DIR>(in[j], in[j + L / 4], in[j + L / 2], in[j + 3 * L / 4], out, tw,
                  Ns, j);
}

// one radix-2 butterfly of a pass, j in [0, L/2)
template <int L, int DIR>
EARHIP_HD void stockham_r2(const cf *in, cf *out, const cf *tw, int Ns, int j) {
  const int k = j & (Ns - 1);
  const cf v0 = in[j];
  cf v1 = in[j + L / 2];
  if (Ns > 1) v1 = cf_mul(v1, tw_load<DIR>(tw, k * (L / (Ns * 2))));
  const int base = ((j - k) << 1) + k;
  out[base] = cf_add(v0, v1);
  out[base + Ns] = cf_sub(v0, v1);
}

// number of radix-4 passes of an L-point transform (a trailing radix-2 pass is
// needed when log2(L) is odd)
constexpr int fft_log2(int n) { return n <= 1 ? 0 : 1 + fft_log2(n / 2); }
constexpr int fft_r4_passes(int L) { return fft_log2(L) / 2; }
constexpr bool fft_has_r2(int L) { return fft_log2(L) % 2 == 1; }
constexpr int fft_total_passes(int L) { return fft_r4_passes(L) + (fft_has_r2(L) ? 1 : 0); }

#if defined(__HIPCC__)
// Runs passes [first, total) of an L-point transform on LDS buffers a -> b ->
// a ...; `src` holds the input of pass `first`.  Every thread of the NT-thread
// workgroup must call this.  Returns the buffer holding the result.  A barrier
// is issued before each pass (so the caller's LDS writes are visible) but not
// after the last one.
template <int L, int DIR, int NT>
__device__ __forceinline__ cf *fft_run_passes(cf *src, cf *dst, const cf *tw,
                                              int first, int tid) {
  constexpr int NR4 = fft_r4_passes(L);
  int Ns = 1;
  for (int p = 0; p < first; ++p) Ns *= 4;
#pragma unroll
  for (int p = 0; p < NR4; ++p) {
    if (p < first) continue;
    __syncthreads();
    for (int j = tid; j < L / 4; j += NT) stockham_r4<L, DIR>(src, dst, tw, Ns, j);
    cf *t = src;
    src = dst;
    dst = t;
    Ns *= 4;
  }
  if (fft_has_r2(L) && first <= NR4) {
    __syncthreads();
    for (int j = tid; j < L / 2; j += NT) stockham_r2<L, DIR>(src, dst, tw, Ns, j);
    cf *t = src;
    src = dst;
    dst = t;
  }
  return src;
}
#endif

}  // namespace earhip
