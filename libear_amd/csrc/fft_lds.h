// fft_lds.h — complex FFT passes for data held in LDS: compile-time power-of-two sizes (the hot
// sizes) and run-time mixed-radix sizes (FftShape: any length up to 8192, every prime factor).
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


// ---------------------------------------------------------------------------
// Run-time sizes.  libear's FFT is kissfft (src/fft_kiss.cpp:104-107), which factorises ANY length
// (radix 4, 2, 3, 5 butterflies, a generic one for other primes: submodules/kissfft/kissfft.hh:34-51),
// so a BlockConvolver block size need not be a power of two (480 = 10 ms at 48 kHz, 960, 1920 ...).
// Same Stockham formulation as above with the radix of every pass taken from a shape: radix-4 passes
// first, then 2, 3, 5, then the remaining primes in ascending order through the generic butterfly
// (O(p^2) per butterfly: inputs re-read from LDS instead of a register array of run-time size).
struct FftShape {
  int L;
  int npass;
  int radix[14];  // L <= 8192: at most 13 prime factors
};
constexpr int kFftMaxLen = 8192;

// Any length in [2, 8192], like kissfft (kissfft.hh:36-51: 4s, then 2s, then the odd primes in ascending
// order, whatever they are — a prime p costs O(p) per output in the generic butterfly, there as here).
// false: L is out of range.
inline bool fft_make_shape(int L, FftShape *s) {
  s->L = L;
  s->npass = 0;
  if (L < 2 || L > kFftMaxLen) return false;
  int rest = L;
  while (rest % 4 == 0) s->radix[s->npass++] = 4, rest /= 4;
  for (int p = 2; p * p <= rest; p++)
    while (rest % p == 0) {
      if (s->npass == 14) return false;
      s->radix[s->npass++] = p;
      rest /= p;
    }
  if (rest > 1) {  // what is left is a prime
    if (s->npass == 14) return false;
    s->radix[s->npass++] = rest;
  }
  return true;
}
EARHIP_HD bool fft_radix_is_generic(int R) { return R > 5; }

// output q of the generic radix-R butterfly j (R any prime): X_q = sum_r (in_r W_{Ns R}^{k r}) W_R^{r q},
// W_R^{r q} = tw[(r q mod R) M] — terms added in the order r = 0, 1, ... (kissfft's kf_bfly_generic,
// kissfft.hh:300-340).  One (j, q) pair per call so that a pass with few, large butterflies (L = 2 p) still
// spreads over a whole workgroup.
template <int DIR>
EARHIP_HD void stockham_generic_q(const cf *in, cf *out, const cf *tw, int L, int R, int Ns, int j, int q) {
  const int k = j % Ns;
  const int M = L / R;
  const int step = L / (Ns * R);
  const int base = (j - k) * R + k;
  cf acc = in[j];
  int rq = 0, rk = 0;  // r q mod R and r k step (< L)
  for (int r = 1; r < R; r++) {
    rq += q;
    if (rq >= R) rq -= R;
    rk += k * step;
    cf v = in[j + r * M];
    if (Ns > 1) v = cf_mul(v, tw_load<DIR>(tw, rk));
    acc = cf_add(acc, cf_mul(v, tw_load<DIR>(tw, rq * M)));
  }
  out[base + q * Ns] = acc;
}

// one radix-R butterfly of a pass over L points, j in [0, L / R)
template <int DIR>
EARHIP_HD void stockham_any(const cf *in, cf *out, const cf *tw, int L, int R, int Ns, int j) {
  const int k = j % Ns;
  const int M = L / R;              // butterflies of the pass = input stride
  const int step = L / (Ns * R);    // W_{Ns R}^{k r} = tw[k r step]
  const int base = (j - k) * R + k;
  if (R == 4) {
    cf v0 = in[j], v1 = in[j + M], v2 = in[j + 2 * M], v3 = in[j + 3 * M];
    if (Ns > 1) {
      v1 = cf_mul(v1, tw_load<DIR>(tw, k * step));
      v2 = cf_mul(v2, tw_load<DIR>(tw, 2 * k * step));
      v3 = cf_mul(v3, tw_load<DIR>(tw, 3 * k * step));
    }
    const cf a0 = cf_add(v0, v2), a1 = cf_sub(v0, v2);
    const cf a2 = cf_add(v1, v3), a3 = cf_mul_i<DIR>(cf_sub(v1, v3));
    out[base] = cf_add(a0, a2);
    out[base + Ns] = cf_add(a1, a3);
    out[base + 2 * Ns] = cf_sub(a0, a2);
    out[base + 3 * Ns] = cf_sub(a1, a3);
  } else if (R == 2) {
    const cf v0 = in[j];
    cf v1 = in[j + M];
    if (Ns > 1) v1 = cf_mul(v1, tw_load<DIR>(tw, k * step));
    out[base] = cf_add(v0, v1);
    out[base + Ns] = cf_sub(v0, v1);
  } else if (R == 3) {
    const cf v0 = in[j];
    cf v1 = in[j + M], v2 = in[j + 2 * M];
    if (Ns > 1) {
      v1 = cf_mul(v1, tw_load<DIR>(tw, k * step));
      v2 = cf_mul(v2, tw_load<DIR>(tw, 2 * k * step));
    }
    // w = exp(DIR 2 pi i / 3) = -1/2 + DIR i sqrt(3)/2
    const float h = 0.86602540378443865f;
    const cf t = cf_add(v1, v2), d = cf_sub(v1, v2);
    const cf m = cf_make(v0.x - 0.5f * t.x, v0.y - 0.5f * t.y);
    const cf r = cf_mul_i<DIR>(cf_make(h * d.x, h * d.y));
    out[base] = cf_add(v0, t);
    out[base + Ns] = cf_add(m, r);
    out[base + 2 * Ns] = cf_sub(m, r);
  } else if (R == 5) {
    const cf v0 = in[j];
    cf v1 = in[j + M], v2 = in[j + 2 * M], v3 = in[j + 3 * M], v4 = in[j + 4 * M];
    if (Ns > 1) {
      v1 = cf_mul(v1, tw_load<DIR>(tw, k * step));
      v2 = cf_mul(v2, tw_load<DIR>(tw, 2 * k * step));
      v3 = cf_mul(v3, tw_load<DIR>(tw, 3 * k * step));
      v4 = cf_mul(v4, tw_load<DIR>(tw, 4 * k * step));
    }
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;  // cos(2 pi / 5), cos(4 pi / 5)
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;   // sin(2 pi / 5), sin(4 pi / 5)
    const cf a1 = cf_add(v1, v4), a2 = cf_add(v2, v3), b1 = cf_sub(v1, v4), b2 = cf_sub(v2, v3);
    const cf m1 = cf_make(v0.x + c1 * a1.x + c2 * a2.x, v0.y + c1 * a1.y + c2 * a2.y);
    const cf m2 = cf_make(v0.x + c2 * a1.x + c1 * a2.x, v0.y + c2 * a1.y + c1 * a2.y);
    const cf r1 = cf_mul_i<DIR>(cf_make(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));
    const cf r2 = cf_mul_i<DIR>(cf_make(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));
    out[base] = cf_add(v0, cf_add(a1, a2));
    out[base + Ns] = cf_add(m1, r1);
    out[base + 2 * Ns] = cf_add(m2, r2);
    out[base + 3 * Ns] = cf_sub(m2, r2);
    out[base + 4 * Ns] = cf_sub(m1, r1);
  } else {  // any other prime
    for (int q = 0; q < R; q++) stockham_generic_q<DIR>(in, out, tw, L, R, Ns, j, q);
  }
}

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

// the dynamic LDS of the run-time-size kernels (2 L complex, + what the kernel adds)
extern __shared__ __attribute__((aligned(16))) unsigned char fft_dyn_lds[];

// the same for a run-time shape (all passes)
template <int DIR, int NT>
__device__ __forceinline__ cf *fft_run_shape(cf *src, cf *dst, const cf *tw, const FftShape &S, int tid) {
  int Ns = 1;
  for (int p = 0; p < S.npass; ++p) {
    const int R = S.radix[p];
    __syncthreads();
    if (fft_radix_is_generic(R)) {  // one output per thread and step: L outputs of O(R) each
      const int M = S.L / R;
      for (int idx = tid; idx < S.L; idx += NT) stockham_generic_q<DIR>(src, dst, tw, S.L, R, Ns, idx % M, idx / M);
    } else {
      for (int j = tid; j < S.L / R; j += NT) stockham_any<DIR>(src, dst, tw, S.L, R, Ns, j);
    }
    cf *t = src;
    src = dst;
    dst = t;
    Ns *= R;
  }
  return src;
}
#endif

}  // namespace earhip
