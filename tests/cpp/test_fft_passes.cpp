// CPU unit test of the Stockham pass index algebra in libear_amd/csrc/fft_lds.h.
// The same functions run per-thread on the GPU; here every butterfly index of a
// pass is executed in a loop (a pass boundary = the device's __syncthreads()).
// Build: g++ -std=c++17 -O2 test_fft_passes.cpp -o test_fft_passes
#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>

#include "../../libear_amd/csrc/fft_lds.h"

using namespace earhip;

template <int L, int DIR>
double run_case(unsigned seed) {
  std::vector<cf> a(L), b(L), tw(L);
  for (int t = 0; t < L; t++) {
    const double ang = -2.0 * M_PI * t / L;
    tw[t] = cf_make((float)std::cos(ang), (float)std::sin(ang));
  }
  std::mt19937 g(seed);
  std::uniform_real_distribution<float> u(-1, 1);
  std::vector<std::complex<double>> x(L);
  for (int i = 0; i < L; i++) {
    a[i] = cf_make(u(g), u(g));
    x[i] = {a[i].x, a[i].y};
  }
  cf *src = a.data(), *dst = b.data();
  int Ns = 1;
  for (int p = 0; p < fft_r4_passes(L); p++) {
    for (int j = 0; j < L / 4; j++) stockham_r4<L, DIR>(src, dst, tw.data(), Ns, j);
    std::swap(src, dst);
    Ns *= 4;
  }
  if (fft_has_r2(L)) {
    for (int j = 0; j < L / 2; j++) stockham_r2<L, DIR>(src, dst, tw.data(), Ns, j);
    std::swap(src, dst);
  }
  // reference DFT in double
  double err = 0, nrm = 0;
  for (int k = 0; k < L; k++) {
    std::complex<double> s = 0;
    for (int n = 0; n < L; n++) {
      const double ang = DIR * 2.0 * M_PI * (double)((long long)k * n % L) / L;
      s += x[n] * std::complex<double>(std::cos(ang), std::sin(ang));
    }
    const std::complex<double> got(src[k].x, src[k].y);
    err += std::norm(got - s);
    nrm += std::norm(s);
  }
  return std::sqrt(err / nrm);
}

// run-time shapes (mixed radix): the same check against a double DFT
template <int DIR>
double run_shape(int L, unsigned seed, bool *ok_shape) {
  FftShape S;
  *ok_shape = fft_make_shape(L, &S);
  if (!*ok_shape) return 0.0;
  std::vector<cf> a(L), b(L), tw(L);
  for (int t = 0; t < L; t++) {
    const double ang = -2.0 * M_PI * t / L;
    tw[t] = cf_make((float)std::cos(ang), (float)std::sin(ang));
  }
  std::mt19937 g(seed);
  std::uniform_real_distribution<float> u(-1, 1);
  std::vector<std::complex<double>> x(L);
  for (int i = 0; i < L; i++) {
    a[i] = cf_make(u(g), u(g));
    x[i] = {a[i].x, a[i].y};
  }
  cf *src = a.data(), *dst = b.data();
  int Ns = 1;
  for (int p = 0; p < S.npass; p++) {
    const int R = S.radix[p];
    for (int j = 0; j < L / R; j++) stockham_any<DIR>(src, dst, tw.data(), L, R, Ns, j);
    std::swap(src, dst);
    Ns *= R;
  }
  double err = 0, nrm = 0;
  for (int k = 0; k < L; k++) {
    std::complex<double> s = 0;
    for (int n = 0; n < L; n++) {
      const double ang = DIR * 2.0 * M_PI * (double)((long long)k * n % L) / L;
      s += x[n] * std::complex<double>(std::cos(ang), std::sin(ang));
    }
    const std::complex<double> got(src[k].x, src[k].y);
    err += std::norm(got - s);
    nrm += std::norm(s);
  }
  return std::sqrt(err / nrm);
}

int check_shape(int L, bool expect_ok = true) {
  bool s1, s2;
  const double ef = run_shape<-1>(L, L, &s1), ei = run_shape<+1>(L, L + 1, &s2);
  if (!expect_ok) {
    std::printf("L=%5d (run-time) rejected: %s\n", L, !s1 ? "ok" : "FAIL");
    return s1 ? 1 : 0;
  }
  // (a prime p is a sum of p float32 terms per output, in kissfft as here: the error grows like sqrt(p))
  int pmax = 2;
  {
    FftShape S;
    if (fft_make_shape(L, &S))
      for (int i = 0; i < S.npass; i++) pmax = S.radix[i] > pmax ? S.radix[i] : pmax;
  }
  const double tol = pmax <= 97 ? 6e-7 : 6e-7 * std::sqrt(pmax / 97.0);
  const bool ok = s1 && s2 && ef < tol && ei < tol;
  std::printf("L=%5d (run-time) fwd_relerr=%.3g inv_relerr=%.3g %s\n", L, ef, ei, ok ? "ok" : "FAIL");
  return ok ? 0 : 1;
}

template <int L>
int check() {
  const double ef = run_case<L, -1>(L), ei = run_case<L, +1>(L + 1);
  const bool ok = ef < 3e-7 && ei < 3e-7;
  std::printf("L=%5d passes=%d fwd_relerr=%.3g inv_relerr=%.3g %s\n", L,
              fft_total_passes(L), ef, ei, ok ? "ok" : "FAIL");
  return ok ? 0 : 1;
}

int main() {
  int bad = 0;
  bad += check<64>();
  bad += check<128>();
  bad += check<256>();
  bad += check<512>();
  bad += check<1024>();
  bad += check<2048>();
  bad += check<4096>();
  bad += check<8192>();
  // 2 x block sizes that are not powers of two: 3- and 5-smooth, other primes, powers of two again
  for (int L : {6, 10, 18, 30, 90, 96, 240, 882, 960, 1000, 1024, 1920, 2 * 1155, 3840, 6000, 2 * 97 * 31, 8192, 2 * 3 * 343, 2 * 101, 2 * 1031, 4 * 499})
    bad += check_shape(L);
  bad += check_shape(2 * 4093);          // the largest prime an even length up to 8192 can hold
  bad += check_shape(8194, false);       // out of range
  return bad;
}
