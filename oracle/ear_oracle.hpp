// oracle/ear_oracle.hpp — CPU restatement of libear's per-block DSP hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
// in __graft_entry__.py and bench.py's cpu_baseline leg may build, link or call
// anything in oracle/.  The shipped path (libear_amd/) never includes this file
// and has no CPU fallback.
//
// Every function states the reference file:line (relative to the libear tree)
// whose behaviour it restates.  The arithmetic (operation order, float types,
// absence of FMA contraction — build with -ffp-contract=off) follows the
// reference exactly so that results are comparable bit-for-bit wherever the
// reference itself is deterministic.
//
// Parity pinning:
//   * FFT: checked bit-for-bit against the reference's vendored kissfft.hh,
//     compiled in place into oracle/_ref (oracle/Makefile target `ref`), and
//     against committed fixtures generated from it (tests/golden/).
//   * GainInterpolator: reference header cannot be compiled here without a
//     stand-in for its CMake-generated export header, so it is pinned by the
//     reference tests' closed-form expectations (tests/gain_interpolator_tests.cpp).
//   * BlockConvolver: pinned by the reference tests' brute-force convolution
//     oracle and all 15 scenarios (tests/block_convolver_tests.cpp), abs 1e-6.
//   * DelayBuffer / VariableBlockSizeAdapter: exact-shift properties (==).
//   * Decorrelator design: the 6 known-answer values of tests/decorrelate_tests.cpp.
//   * The composed Objects render block exists only as prose in the reference
//     (docs/dsp.rst:40-71); its composition is "parity unpinned" and is pinned
//     only transitively through its stages.
#pragma once

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace ear_oracle {

// ---------------------------------------------------------------------------
// errors — include/ear/exceptions.hpp:8-43, include/ear/helpers/assert.hpp:7-18
// ---------------------------------------------------------------------------
struct invalid_argument : std::invalid_argument {
  explicit invalid_argument(const std::string &w) : std::invalid_argument(w) {}
};
struct internal_error : std::runtime_error {
  explicit internal_error(const std::string &w)
      : std::runtime_error("internal error: " + w) {}
};
inline void check_internal(bool ok, const char *msg) {
  if (!ok) throw internal_error(msg);
}

using SampleIndex = long;  // include/ear/dsp/gain_interpolator.hpp:16

// ---------------------------------------------------------------------------
// Interpolation policies — include/ear/dsp/gain_interpolator.hpp:133-300
// ---------------------------------------------------------------------------

// ramp position of sample (block_start + i) on the curve [start, end);
// gain_interpolator.hpp:192-195 (the long -> float conversion is part of the
// reference arithmetic).
static inline float ramp_pos(SampleIndex block_start, SampleIndex i,
                             SampleIndex start, float scale) {
  return (float)((block_start + i) - start) * scale;
}
static inline float ramp_gain(float p, float s, float e) {
  return (1.0f - p) * s + p * e;  // gain_interpolator.hpp:195,226,273
}

// 1 -> 1, gain_interpolator.hpp:186-209
struct LinearInterpSingle {
  using Point = float;
  static bool constant_interp(const Point &a, const Point &b) { return a == b; }
  static void apply_interp(const float *const *in, float *const *out,
                           SampleIndex r0, SampleIndex r1,
                           SampleIndex block_start, SampleIndex start,
                           SampleIndex end, const Point &sp, const Point &ep) {
    const float scale = 1.0f / (end - start);
    for (SampleIndex i = r0; i < r1; i++) {
      const float p = ramp_pos(block_start, i, start, scale);
      out[0][i] = in[0][i] * ramp_gain(p, sp, ep);
    }
  }
  static void apply_constant(const float *const *in, float *const *out,
                             SampleIndex r0, SampleIndex r1, const Point &pt) {
    for (SampleIndex i = r0; i < r1; i++) out[0][i] = in[0][i] * pt;
  }
};

// 1 -> N, gain_interpolator.hpp:213-242
struct LinearInterpVector {
  using Point = std::vector<float>;
  static bool constant_interp(const Point &a, const Point &b) { return a == b; }
  static void apply_interp(const float *const *in, float *const *out,
                           SampleIndex r0, SampleIndex r1,
                           SampleIndex block_start, SampleIndex start,
                           SampleIndex end, const Point &sp, const Point &ep) {
    const float scale = 1.0f / (end - start);
    for (size_t c = 0; c < sp.size(); c++) {
      const float s = sp[c], e = ep[c];
      for (SampleIndex i = r0; i < r1; i++) {
        const float p = ramp_pos(block_start, i, start, scale);
        out[c][i] = in[0][i] * ramp_gain(p, s, e);
      }
    }
  }
  static void apply_constant(const float *const *in, float *const *out,
                             SampleIndex r0, SampleIndex r1, const Point &pt) {
    for (size_t c = 0; c < pt.size(); c++)
      for (SampleIndex i = r0; i < r1; i++) out[c][i] = in[0][i] * pt[c];
  }
};

// M -> N, point indexed [in][out]; gain_interpolator.hpp:248-300.  The output
// range is zeroed first and inputs are accumulated in channel order.
struct LinearInterpMatrix {
  using Point = std::vector<std::vector<float>>;
  static bool constant_interp(const Point &a, const Point &b) { return a == b; }
  static size_t n_out(const Point &p) { return p.empty() ? 0 : p[0].size(); }
  static void apply_interp(const float *const *in, float *const *out,
                           SampleIndex r0, SampleIndex r1,
                           SampleIndex block_start, SampleIndex start,
                           SampleIndex end, const Point &sp, const Point &ep) {
    const float scale = 1.0f / (end - start);
    for (size_t o = 0; o < n_out(sp); o++)
      for (SampleIndex i = r0; i < r1; i++) out[o][i] = 0.0;
    for (size_t m = 0; m < sp.size(); m++)
      for (size_t o = 0; o < sp[m].size(); o++) {
        const float s = sp[m][o], e = ep[m][o];
        for (SampleIndex i = r0; i < r1; i++) {
          const float p = ramp_pos(block_start, i, start, scale);
          out[o][i] += in[m][i] * ramp_gain(p, s, e);
        }
      }
  }
  static void apply_constant(const float *const *in, float *const *out,
                             SampleIndex r0, SampleIndex r1, const Point &pt) {
    for (size_t o = 0; o < n_out(pt); o++)
      for (SampleIndex i = r0; i < r1; i++) out[o][i] = 0.0;
    for (size_t m = 0; m < pt.size(); m++)
      for (size_t o = 0; o < pt[m].size(); o++)
        for (SampleIndex i = r0; i < r1; i++) out[o][i] += in[m][i] * pt[m][o];
  }
};

// ---------------------------------------------------------------------------
// GainInterpolator — include/ear/dsp/gain_interpolator.hpp:40-130
//
// Curve segment k (0..n) lies between point k-1 and point k; segment 0 is
// "before the first point", segment n is "after the last point".  A sample t
// belongs to segment k iff  time[k-1] <= t < time[k]  (block_cmp, :94-106).
// ---------------------------------------------------------------------------
template <typename Interp>
class GainInterpolator {
 public:
  std::vector<std::pair<SampleIndex, typename Interp::Point>> interp_points;

  // gain_interpolator.hpp:53-87
  void process(SampleIndex block_start, size_t nsamples, const float *const *in,
               float *const *out) {
    // The reference reads interp_points[-1] when the vector is empty (UB,
    // :71-72); the restatement (and the HIP path) define this as an error.
    if (interp_points.empty())
      throw invalid_argument("interp_points must not be empty");
    const SampleIndex block_end = block_start + (SampleIndex)nsamples;
    const size_t n = interp_points.size();
    SampleIndex cur = block_start;
    while (cur < block_end) {
      const size_t k = locate(cur);
      const SampleIndex seg_end =
          k == n ? block_end : std::min(interp_points[k].first, block_end);
      check_internal(cur < seg_end,
                     "found block ends before processed block starts");
      const bool flat =
          k == 0 || k == n ||
          Interp::constant_interp(interp_points[k - 1].second,
                                  interp_points[k].second);
      if (flat) {
        const size_t src = k == n ? k - 1 : k;
        Interp::apply_constant(in, out, cur - block_start, seg_end - block_start,
                               interp_points[src].second);
      } else {
        Interp::apply_interp(in, out, cur - block_start, seg_end - block_start,
                             block_start, interp_points[k - 1].first,
                             interp_points[k].first,
                             interp_points[k - 1].second,
                             interp_points[k].second);
      }
      cur = seg_end;
    }
  }

 private:
  size_t hint_ = 0;

  // -1 / 0 / +1: t lies before / inside / after segment k (:94-106)
  int side(size_t k, SampleIndex t) const {
    if (k > 0 && t < interp_points[k - 1].first) return -1;
    if (k < interp_points.size() && t >= interp_points[k].first) return 1;
    return 0;
  }

  // cached linear search; a change of direction means the times are not
  // sorted (:110-129)
  size_t locate(SampleIndex t) {
    if (hint_ > interp_points.size()) hint_ = 0;
    const int first = side(hint_, t);
    int dir = first;
    while (dir != 0) {
      hint_ += dir;
      if (dir != first)
        throw invalid_argument("interpolation points are not sorted");
      dir = side(hint_, t);
    }
    return hint_;
  }
};

// ---------------------------------------------------------------------------
// FFT — restatement of the vendored kissfft (submodules/kissfft/kissfft.hh,
// BSD-3-Clause, Mark Borgerding) as used by src/fft_kiss.cpp.  Mixed-radix
// decimation in time, radix 4 first then 2, then odd radices (3 and 5 with their
// own butterflies, any other prime through the generic one — block sizes that
// are not powers of two).  Written iteratively: digit-reversed load followed by the
// butterfly passes deepest-first, which performs exactly the butterflies of
// the recursive formulation (kissfft.hh:90-123) with identical operands.
// ---------------------------------------------------------------------------
template <typename T>
class KissLikeFFT {
 public:
  using cpx = std::complex<T>;

  KissLikeFFT(size_t n, bool inverse) : n_(n), inverse_(inverse) {
    // twiddles, kissfft.hh:28-32 (computed in T, not in double)
    tw_.resize(n_);
    const T phinc = (inverse_ ? 2 : -2) * std::acos((T)-1) / n_;
    for (size_t i = 0; i < n_; i++) tw_[i] = std::exp(cpx(0, i * phinc));
    // factorisation 4,4,..,2,3,5,7.. kissfft.hh:34-51
    size_t rem = n_, p = 4;
    do {
      while (rem % p) {
        if (p == 4) p = 2;
        else if (p == 2) p = 3;
        else p += 2;
        if (p * p > rem) p = rem;
      }
      rem /= p;
      radix_.push_back(p);
      remain_.push_back(rem);
    } while (rem > 1);
    // digit-reversal: output slot sum(q_s * remain_s) reads input slot
    // sum(q_s * prod(radix_0..s-1))  (leaf copies of kissfft.hh:98-102)
    perm_.assign(n_, 0);
    for (size_t o = 0; o < n_; o++) {
      size_t r = o, src = 0, stride = 1;
      for (size_t s = 0; s < radix_.size(); s++) {
        const size_t q = r / remain_[s];
        r -= q * remain_[s];
        src += q * stride;
        stride *= radix_[s];
      }
      perm_[o] = src;
    }
  }

  size_t size() const { return n_; }
  const std::vector<cpx> &twiddles() const { return tw_; }

  // complex DFT, un-normalised (kissfft.hh:90-123)
  void transform(const cpx *in, cpx *out) const {
    for (size_t o = 0; o < n_; o++) out[o] = in[perm_[o]];
    size_t fstride = n_;
    for (size_t s = radix_.size(); s-- > 0;) {
      const size_t p = radix_[s], m = remain_[s];
      fstride = n_ / (p * m);  // = prod(radix_0..s-1) = number of groups
      for (size_t g = 0; g < fstride; g++) {
        cpx *f = out + g * p * m;
        switch (p) {
          case 2: bfly2(f, fstride, m); break;
          case 3: bfly3(f, fstride, m); break;
          case 4: bfly4(f, fstride, m); break;
          case 5: bfly5(f, fstride, m); break;
          default: bfly_any(f, fstride, m, p); break;
        }
      }
    }
  }

  // real DFT of 2*n_ reals through this n_-point complex plan; DC in
  // dst[0].real, Nyquist packed in dst[0].imag (kissfft.hh:154-189)
  void transform_real(const T *src, cpx *dst) const {
    const size_t N = n_;
    if (N == 0) return;
    transform(reinterpret_cast<const cpx *>(src), dst);
    dst[0] = cpx(dst[0].real() + dst[0].imag(), dst[0].real() - dst[0].imag());
    const T pi = std::acos((T)-1);
    const T half_phi_inc = (inverse_ ? pi : -pi) / N;
    const cpx twiddle_mul = std::exp(cpx(0, half_phi_inc));
    for (size_t k = 1; 2 * k < N; ++k) {
      const cpx w = (T)0.5 * cpx(dst[k].real() + dst[N - k].real(),
                                 dst[k].imag() - dst[N - k].imag());
      const cpx z = (T)0.5 * cpx(dst[k].imag() + dst[N - k].imag(),
                                 -dst[k].real() + dst[N - k].real());
      const cpx twiddle = k % 2 == 0 ? tw_[k / 2] : tw_[k / 2] * twiddle_mul;
      dst[k] = w + twiddle * z;
      dst[N - k] = std::conj(w - twiddle * z);
    }
    if (N % 2 == 0) dst[N / 2] = std::conj(dst[N / 2]);
  }

 private:
  // kissfft.hh:193-200
  void bfly2(cpx *f, size_t fstride, size_t m) const {
    for (size_t k = 0; k < m; ++k) {
      const cpx t = f[m + k] * tw_[k * fstride];
      f[m + k] = f[k] - t;
      f[k] += t;
    }
  }
  // kissfft.hh:202-231
  void bfly3(cpx *f, size_t fstride, size_t m) const {
    const size_t m2 = 2 * m;
    const cpx epi3 = tw_[fstride * m];
    for (size_t k = 0; k < m; ++k) {
      const cpx s1 = f[k + m] * tw_[k * fstride];
      const cpx s2 = f[k + m2] * tw_[k * fstride * 2];
      const cpx s3 = s1 + s2;
      cpx s0 = s1 - s2;
      f[k + m] = f[k] - s3 * T(0.5);
      s0 *= epi3.imag();
      f[k] += s3;
      f[k + m2] = cpx(f[k + m].real() + s0.imag(), f[k + m].imag() - s0.real());
      f[k + m] += cpx(-s0.imag(), s0.real());
    }
  }
  // kissfft.hh:233-254
  void bfly4(cpx *f, size_t fstride, size_t m) const {
    const T sgn = (T)(inverse_ ? -1 : +1);
    for (size_t k = 0; k < m; ++k) {
      const cpx a1 = f[k + m] * tw_[k * fstride];
      const cpx a2 = f[k + 2 * m] * tw_[k * fstride * 2];
      const cpx a3 = f[k + 3 * m] * tw_[k * fstride * 3];
      const cpx d02 = f[k] - a2;
      f[k] += a2;
      const cpx s13 = a1 + a3;
      cpx d13 = a1 - a3;
      d13 = cpx(d13.imag() * sgn, -d13.real() * sgn);
      f[k + 2 * m] = f[k] - s13;
      f[k] += s13;
      f[k + m] = d02 + d13;
      f[k + 3 * m] = d02 - d13;
    }
  }
  // kissfft.hh:256-323
  void bfly5(cpx *f, size_t fstride, size_t m) const {
    const cpx ya = tw_[fstride * m], yb = tw_[fstride * 2 * m];
    for (size_t u = 0; u < m; ++u) {
      cpx *f0 = f + u, *f1 = f0 + m, *f2 = f0 + 2 * m, *f3 = f0 + 3 * m,
          *f4 = f0 + 4 * m;
      const cpx x0 = *f0;
      const cpx x1 = *f1 * tw_[u * fstride];
      const cpx x2 = *f2 * tw_[2 * u * fstride];
      const cpx x3 = *f3 * tw_[3 * u * fstride];
      const cpx x4 = *f4 * tw_[4 * u * fstride];
      const cpx s14 = x1 + x4, d14 = x1 - x4, s23 = x2 + x3, d23 = x2 - x3;
      *f0 += s14;
      *f0 += s23;
      const cpx a = x0 + cpx(s14.real() * ya.real() + s23.real() * yb.real(),
                             s14.imag() * ya.real() + s23.imag() * yb.real());
      const cpx b = cpx(d14.imag() * ya.imag() + d23.imag() * yb.imag(),
                        -d14.real() * ya.imag() - d23.real() * yb.imag());
      *f1 = a - b;
      *f4 = a + b;
      const cpx c = x0 + cpx(s14.real() * yb.real() + s23.real() * ya.real(),
                             s14.imag() * yb.real() + s23.imag() * ya.real());
      const cpx d = cpx(-d14.imag() * yb.imag() + d23.imag() * ya.imag(),
                        d14.real() * yb.imag() - d23.real() * ya.imag());
      *f2 = c + d;
      *f3 = c - d;
    }
  }

  // kissfft.hh:321-352: any other radix p, O(p^2) per butterfly; the twiddle index of output q1 walks
  // in steps of fstride * (u + q1 m) modulo n
  void bfly_any(cpx *f, size_t fstride, size_t m, size_t p) const {
    std::vector<cpx> x(p);
    for (size_t u = 0; u < m; ++u) {
      for (size_t q1 = 0; q1 < p; ++q1) x[q1] = f[u + q1 * m];
      for (size_t q1 = 0; q1 < p; ++q1) {
        const size_t k = u + q1 * m;
        size_t twidx = 0;
        f[k] = x[0];
        for (size_t q = 1; q < p; ++q) {
          twidx += fstride * k;
          if (twidx >= n_) twidx -= n_;
          f[k] += x[q] * tw_[twidx];
        }
      }
    }
  }

  size_t n_;
  bool inverse_;
  std::vector<cpx> tw_;
  std::vector<size_t> radix_, remain_, perm_;
};

// r2c / c2r plan with libear's packing — src/fft_kiss.cpp:52-99.  Forward is
// a half-length complex transform + untangle; reverse Hermitian-extends to a
// full complex inverse of n_fft points and keeps the real part.  Both
// un-normalised.
class RealFFT {
 public:
  using cpx = std::complex<float>;
  explicit RealFFT(size_t n_fft)
      : n_fft_(n_fft), fwd_(n_fft / 2, false), rev_(n_fft, true),
        tmp_in_(n_fft), tmp_out_(n_fft) {
    check_internal(n_fft % 2 == 0, "n_fft must be even");  // fft_kiss.cpp:105
  }
  size_t n_fft() const { return n_fft_; }
  // fft_kiss.cpp:61-71
  void forward(const float *in, cpx *out) const {
    fwd_.transform_real(in, out);
    out[n_fft_ / 2] = out[0].imag();
    out[0].imag(0.0);
  }
  // fft_kiss.cpp:73-88
  void reverse(const cpx *in, float *out) const {
    for (size_t i = 0; i < n_fft_; i++)
      tmp_in_[i] = i < n_fft_ / 2 + 1 ? in[i] : std::conj(in[n_fft_ - i]);
    rev_.transform(tmp_in_.data(), tmp_out_.data());
    for (size_t i = 0; i < n_fft_; i++) out[i] = tmp_out_[i].real();
  }

 private:
  size_t n_fft_;
  KissLikeFFT<float> fwd_, rev_;
  mutable std::vector<cpx> tmp_in_, tmp_out_;
};

// ---------------------------------------------------------------------------
// BlockConvolver — src/dsp/block_convolver_impl.{hpp,cpp}
// ---------------------------------------------------------------------------
namespace block_convolver {
using cpx = std::complex<float>;

// block_convolver_impl.cpp:10-14
struct Context {
  explicit Context(size_t block_size)
      : block_size(block_size), fft(2 * block_size), td_size(2 * block_size),
        fd_size(block_size + 1) {}
  size_t block_size;
  RealFFT fft;
  size_t td_size, fd_size;
};

// block_convolver_impl.cpp:16-41: partitions of block_size taps, each
// zero-padded to 2*block_size and transformed.
struct Filter {
  Filter(const std::shared_ptr<Context> &ctx, size_t n, const float *taps) {
    std::vector<float> td(ctx->td_size);
    for (size_t ofs = 0; ofs < n; ofs += ctx->block_size) {
      const size_t len = std::min(ctx->block_size, n - ofs);
      std::fill(td.begin(), td.end(), 0.0f);
      std::copy(taps + ofs, taps + ofs + len, td.begin());
      std::vector<cpx> fd(ctx->fd_size);
      ctx->fft.forward(td.data(), fd.data());
      blocks.push_back(std::move(fd));
    }
  }
  size_t num_blocks() const { return blocks.size(); }
  std::vector<std::vector<cpx>> blocks;
};

// A buffer that remembers whether it holds only zeros
// (block_convolver_impl.hpp:124-146).
template <typename V>
struct Tracked {
  explicit Tracked(size_t n) : data(n), zero(true) {}
  std::vector<V> data;
  bool zero;
  std::vector<V> &write() { zero = false; return data; }
  void clear() {
    if (!zero) { zero = true; std::fill(data.begin(), data.end(), V()); }
  }
};

class BlockConvolver {
 public:
  using FilterPtr = std::shared_ptr<const Filter>;

  // block_convolver_impl.cpp:43-60
  BlockConvolver(const std::shared_ptr<Context> &ctx, size_t num_blocks)
      : ctx_(ctx), P_(num_blocks), fq_(num_blocks + 1), f_ofs_(0), s_ofs_(0),
        tail_(ctx->block_size), td_old_(ctx->td_size), td_new_(ctx->td_size),
        acc_(ctx->fd_size), out_td_(ctx->td_size) {
    for (size_t i = 0; i < P_; i++) {
      sp_old_.emplace_back(ctx->fd_size);
      sp_new_.emplace_back(ctx->fd_size);
    }
  }
  // block_convolver_impl.cpp:63-69
  BlockConvolver(const std::shared_ptr<Context> &ctx, const FilterPtr &filter,
                 size_t num_blocks = 0)
      : BlockConvolver(ctx, num_blocks > 0 ? num_blocks : filter->num_blocks()) {
    set_filter(filter);
  }

  // block_convolver_impl.cpp:71-76
  void crossfade_filter(const FilterPtr &f) { check(f); filt(0) = f; }
  // block_convolver_impl.cpp:78-83
  void set_filter(const FilterPtr &f) {
    check(f);
    for (auto &q : fq_) q = f;
  }

  // block_convolver_impl.cpp:143-237; `in` may be null (= silence)
  void process(const float *in, float *out) {
    const size_t B = ctx_->block_size;
    bool silent = in == nullptr;
    if (!silent) {
      silent = true;
      for (size_t i = 0; i < B; i++)
        if (in[i] != 0.0f) { silent = false; break; }
    }
    if (silent) {  // :156-159
      sp_old(0).clear();
      sp_new(0).clear();
    } else if (filt(1) != filt(0)) {  // :162-176 filter changed: fade both
      std::vector<float> &dn = td_old_.write(), &up = td_new_.write();
      const float i_scale = 1.0f / (float)B;  // :133
      for (size_t i = 0; i < B; i++) {        // :135-140
        const float a = (float)i * i_scale, b = 1.0f - a;
        up[i] = a * in[i];
        dn[i] = b * in[i];
      }
      ctx_->fft.forward(dn.data(), sp_old(0).write().data());
      std::fill(dn.begin() + B, dn.end(), 0.0f);
      ctx_->fft.forward(up.data(), sp_new(0).write().data());
      std::fill(up.begin() + B, up.end(), 0.0f);
    } else {  // :177-187
      std::vector<float> &td = td_new_.write();
      std::copy(in, in + B, td.begin());
      ctx_->fft.forward(td.data(), sp_new(0).write().data());
      std::fill(td.begin() + B, td.end(), 0.0f);
      sp_old(0).clear();
    }

    // spectral multiply-accumulate over the partition queue, old before new,
    // ascending age (:193-209)
    acc_.clear();
    for (size_t i = 0; i < P_; i++) {
      const Filter *fo = filt(i + 1).get(), *fn = filt(i).get();
      if (fo && i < fo->blocks.size() && !sp_old(i).zero) mac(fo->blocks[i], sp_old(i).data);
      if (fn && i < fn->blocks.size() && !sp_new(i).zero) mac(fn->blocks[i], sp_new(i).data);
    }

    const float norm = 1.0f / (float)(2 * B);  // :212
    if (!acc_.zero) {  // :217-226
      std::vector<float> &y = out_td_.write();
      ctx_->fft.reverse(acc_.write().data(), y.data());
      if (!tail_.zero)
        for (size_t i = 0; i < B; i++) y[i] += tail_.data[i];
      std::vector<float> &t = tail_.write();
      for (size_t i = 0; i < B; i++) t[i] = y[B + i];
      for (size_t i = 0; i < B; i++) out[i] = y[i] * norm;
    } else if (!tail_.zero) {  // :227-230
      for (size_t i = 0; i < B; i++) out[i] = tail_.data[i] * norm;
      tail_.clear();
    } else {  // :231-234
      for (size_t i = 0; i < B; i++) out[i] = 0.0f;
    }

    // rotate_queues, :114-122
    s_ofs_ = (s_ofs_ + P_ - 1) % P_;
    f_ofs_ = (f_ofs_ + P_) % (P_ + 1);
    filt(0) = filt(1);
  }

 private:
  // block_convolver_impl.cpp:85-98
  void check(const FilterPtr &f) const {
    if (!f) return;
    for (auto &b : f->blocks)
      if (b.size() != ctx_->fd_size)
        throw invalid_argument(
            "Filter block size is not equal to BlockConvolver block size; "
            "was this created using the same context?");
    if (f->num_blocks() > P_)
      throw invalid_argument("too many blocks in given Filter");
  }
  void mac(const std::vector<cpx> &h, const std::vector<cpx> &x) {
    std::vector<cpx> &a = acc_.write();
    for (size_t k = 0; k < a.size(); k++) a[k] += h[k] * x[k];
  }
  FilterPtr &filt(size_t i) { return fq_[(f_ofs_ + i) % (P_ + 1)]; }
  Tracked<cpx> &sp_old(size_t i) { return sp_old_[(s_ofs_ + i) % P_]; }
  Tracked<cpx> &sp_new(size_t i) { return sp_new_[(s_ofs_ + i) % P_]; }

  std::shared_ptr<Context> ctx_;
  size_t P_;
  std::vector<FilterPtr> fq_;
  size_t f_ofs_, s_ofs_;
  std::vector<Tracked<cpx>> sp_old_, sp_new_;
  Tracked<float> tail_, td_old_, td_new_;
  Tracked<cpx> acc_;
  Tracked<float> out_td_;
};
}  // namespace block_convolver

// ---------------------------------------------------------------------------
// DelayBuffer — src/dsp/delay_buffer_impl.cpp:14-43
// ---------------------------------------------------------------------------
class DelayBuffer {
 public:
  DelayBuffer(size_t nchannels, size_t nsamples)
      : nch_(nchannels), delay_(nsamples), mem_(nchannels * nsamples, 0.0f) {}
  // [mem; input] -> [output; mem], per channel (:19-40)
  void process(size_t nsamples, const float *const *in, float *const *out) {
    for (size_t c = 0; c < nch_; c++) {
      float *mem = mem_.data() + c * delay_;
      for (size_t s = 0; s < nsamples + delay_; s++) {
        const float v = s < delay_ ? mem[s] : in[c][s - delay_];
        if (s < nsamples) out[c][s] = v;
        else mem[s - nsamples] = v;
      }
    }
  }
  int get_delay() const { return (int)delay_; }

 private:
  size_t nch_, delay_;
  std::vector<float> mem_;
};

// ---------------------------------------------------------------------------
// VariableBlockSizeAdapter — src/dsp/variable_block_size_impl.cpp:27-85
// ---------------------------------------------------------------------------
class VariableBlockSizeAdapter {
 public:
  using ProcessFunc = void(const float *const *in, float *const *out);
  VariableBlockSizeAdapter(size_t block_size, size_t n_in, size_t n_out,
                           std::function<ProcessFunc> fn)
      : fn_(std::move(fn)), B_(block_size), nin_(n_in), nout_(n_out),
        ibuf_(block_size * n_in, 0.0f), obuf_(block_size * n_out, 0.0f),
        fill_(0), iptr_(n_in), optr_(n_out) {
    for (size_t c = 0; c < n_in; c++) iptr_[c] = ibuf_.data() + c * B_;
    for (size_t c = 0; c < n_out; c++) optr_[c] = obuf_.data() + c * B_;
  }
  // :44-81
  void process(size_t nsamples, const float *const *in, float *const *out) {
    size_t done = 0;
    while (done < nsamples) {
      const size_t n = std::min(nsamples - done, B_ - fill_);
      for (size_t c = 0; c < nin_; c++)
        for (size_t i = 0; i < n; i++) iptr_[c][fill_ + i] = in[c][done + i];
      for (size_t c = 0; c < nout_; c++)
        for (size_t i = 0; i < n; i++) out[c][done + i] = optr_[c][fill_ + i];
      done += n;
      fill_ += n;
      const bool run = fill_ == B_;
      if (run) {
        fn_(iptr_.data(), optr_.data());
        fill_ = 0;
      }
      check_internal(run || n > 0, "no progress made");
    }
    check_internal(done == nsamples, "processed more samples than expected");
  }
  int get_delay() const { return (int)B_; }  // :85

 private:
  std::function<ProcessFunc> fn_;
  size_t B_, nin_, nout_;
  std::vector<float> ibuf_, obuf_;
  size_t fill_;
  std::vector<float *> iptr_, optr_;
};

// ---------------------------------------------------------------------------
// Decorrelator design — src/decorrelate.cpp:16-97
// ---------------------------------------------------------------------------
// :31-51; all-pass, random phase from mt19937(id), inverse DFT in double.
inline std::vector<double> design_decorrelator_basic(int id, int size) {
  const double PI = 3.14159265358979323846264338327950288;
  std::mt19937 eng(id);
  std::vector<std::complex<double>> fd(size);
  fd[0] = 1.0;
  for (int i = 0; i < size / 2 - 1; ++i) {
    const double u = eng() / static_cast<double>(0x100000000l);  // :18-21
    fd[i + 1] = std::exp(std::complex<double>(0.0, 2.0 * PI * u));
  }
  fd[size / 2] = 1.0;
  for (int i = 0; i < size / 2; ++i) fd[size / 2 + i] = std::conj(fd[size / 2 - i]);
  KissLikeFFT<double> ifft(size, true);
  std::vector<std::complex<double>> td(size);
  ifft.transform(fd.data(), td.data());
  std::vector<double> h(size);
  for (int i = 0; i < size; ++i) h[i] = td[i].real() / size;
  return h;
}
const int kDecorrelatorSize = 512;  // :53
// :55-68 — filter id = number of channel names lexicographically smaller
inline int decorrelator_id(const std::vector<std::string> &names, size_t idx) {
  int id = 0;
  for (auto &n : names)
    if (n < names.at(idx)) id++;
  return id;
}
// :70-90
inline std::vector<std::vector<float>> design_decorrelators(
    const std::vector<std::string> &names) {
  std::vector<std::vector<float>> out(names.size());
  for (size_t c = 0; c < names.size(); c++) {
    const auto h = design_decorrelator_basic(decorrelator_id(names, c), kDecorrelatorSize);
    out[c].resize(h.size());
    for (size_t i = 0; i < h.size(); i++) out[c][i] = (float)h[i];
  }
  return out;
}
inline int decorrelator_compensation_delay() { return (kDecorrelatorSize - 1) / 2; }  // :97

// ---------------------------------------------------------------------------
// Composed Objects render block — specified by docs/dsp.rst:40-71 and
// include/ear/gain_calculators.hpp:45-56; no code in the reference (composition
// parity unpinned).  One GainInterpolator<LinearInterpVector> per object and
// bus, temporaries summed into the buses in object order (the per-object form
// the docs describe); one BlockConvolver per loudspeaker on the diffuse bus;
// DelayBuffer(255) on the direct bus; out = decorrelated + delayed.
// ---------------------------------------------------------------------------
class ObjectsRenderer {
  size_t M_, N_, B_;

 public:
  using Interp = GainInterpolator<LinearInterpVector>;
  ObjectsRenderer(size_t n_objects, size_t n_out, size_t block_size,
                  const std::vector<std::vector<float>> &decorrelators,
                  size_t delay)
      : M_(n_objects), N_(n_out), B_(block_size), direct(n_objects),
        diffuse(n_objects), delay_(n_out, delay), dbus_(n_out * block_size),
        fbus_(n_out * block_size), tmp_(n_out * block_size),
        dec_(n_out * block_size), del_(n_out * block_size), t_(0) {
    auto ctx = std::make_shared<block_convolver::Context>(block_size);
    for (size_t c = 0; c < n_out; c++) {
      auto f = std::make_shared<block_convolver::Filter>(
          ctx, decorrelators[c].size(), decorrelators[c].data());
      conv_.emplace_back(new block_convolver::BlockConvolver(ctx, f));
    }
  }
  std::vector<Interp> direct, diffuse;  // per object; set interp_points
  // one block: in[M][B] -> out[N][B]
  void process(const float *const *in, float *const *out) {
    std::vector<float *> t(N_), db(N_), fb(N_), dc(N_), dl(N_);
    for (size_t c = 0; c < N_; c++) {
      t[c] = tmp_.data() + c * B_;
      db[c] = dbus_.data() + c * B_;
      fb[c] = fbus_.data() + c * B_;
      dc[c] = dec_.data() + c * B_;
      dl[c] = del_.data() + c * B_;
    }
    std::fill(dbus_.begin(), dbus_.end(), 0.0f);
    std::fill(fbus_.begin(), fbus_.end(), 0.0f);
    for (size_t m = 0; m < M_; m++) {
      direct[m].process(t_, B_, &in[m], t.data());
      for (size_t i = 0; i < N_ * B_; i++) dbus_[i] += tmp_[i];
      diffuse[m].process(t_, B_, &in[m], t.data());
      for (size_t i = 0; i < N_ * B_; i++) fbus_[i] += tmp_[i];
    }
    for (size_t c = 0; c < N_; c++) conv_[c]->process(fb[c], dc[c]);
    delay_.process(B_, db.data(), dl.data());
    for (size_t c = 0; c < N_; c++)
      for (size_t i = 0; i < B_; i++) out[c][i] = dc[c][i] + dl[c][i];
    t_ += (SampleIndex)B_;
  }

 private:
  std::vector<std::unique_ptr<block_convolver::BlockConvolver>> conv_;
  DelayBuffer delay_;
  std::vector<float> dbus_, fbus_, tmp_, dec_, del_;
  SampleIndex t_;
};

}  // namespace ear_oracle
