// search.h — point-time searches of the gain curves, host + device so that the index logic is
// unit-tested on the CPU (tests/cpp/test_search.cpp).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define EARHIP_SEARCH_HD __host__ __device__ __forceinline__
#else
#define EARHIP_SEARCH_HD inline
#endif

namespace earhip {

// number of points of object m with time <= t  (= libear's find_block result)
EARHIP_SEARCH_HD int upper_bound_time(const int64_t *t, int n, int64_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (t[mid] <= v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// Same result, started from where v would lie if the points were evenly spaced
// (metadata usually is): a bracket is grown around the guess in doubling steps and
// then bisected, so a good guess costs 3-4 dependent loads instead of log2(n) + 1.
EARHIP_SEARCH_HD int upper_bound_time_guess(const int64_t *t, int n, int64_t v) {
  if (n < 8) return upper_bound_time(t, n, v);
  const int64_t first = t[0], last = t[n - 1];
  if (v < first) return 0;
  if (v >= last) return n;
  // first <= v < last: the answer k is in [1, n-1] with t[k-1] <= v < t[k]
  int g = (int)((double)(v - first) / (double)(last - first) * (double)(n - 1));
  g = g < 0 ? 0 : (g > n - 2 ? n - 2 : g);
  int lo, hi;  // invariant: t[lo] <= v < t[hi]
  if (t[g] <= v) {
    lo = g;
    int step = 1;
    hi = g + 1;
    while (hi < n - 1 && t[hi] <= v) {
      lo = hi;
      step <<= 1;
      hi = hi + step > n - 1 ? n - 1 : hi + step;
    }
  } else {
    hi = g;
    int step = 1;
    lo = g - 1;
    while (lo > 0 && t[lo] > v) {
      hi = lo;
      step <<= 1;
      lo = lo - step < 0 ? 0 : lo - step;
    }
  }
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t[mid] <= v) lo = mid;
    else hi = mid;
  }
  return hi;
}

// The same result with ONE round of loads behind the end points when the points are evenly spaced, as metadata
// mostly is: eight neighbouring times around the guess are requested together (the loads do not depend on each
// other) and the answer is read off them; only when it lies outside the window does the bracketing search run.
// The list builder of the piece-list kernel does one such search per (object, tile): its dependent rounds of loads
// are what that kernel's time is made of.
EARHIP_SEARCH_HD int upper_bound_time_window(const int64_t *t, int n, int64_t v) {
  if (n < 16) return upper_bound_time(t, n, v);
  const int64_t first = t[0], last = t[n - 1];
  if (v < first) return 0;
  if (v >= last) return n;
  int g = (int)((double)(v - first) / (double)(last - first) * (double)(n - 1));
  int lo = g - 3;
  lo = lo < 0 ? 0 : (lo > n - 8 ? n - 8 : lo);
  int64_t w[8];
  for (int i = 0; i < 8; i++) w[i] = t[lo + i];
  if (w[0] <= v && v < w[7]) {
    int k = lo;
    for (int i = 0; i < 8; i++) k += w[i] <= v ? 1 : 0;
    return k;
  }
  return upper_bound_time_guess(t, n, v);
}

// The same search over the packed point RECORDS (16 bytes: time, 1 / segment length, flat bits), with the curve's end points
// handed in (an object's header, ObjHdr): the window of eight records around the guess is one or two cache lines — the lines
// whoever walks the curve from the result on reads next (the list builders: their walk then hits in cache instead of
// making further trips to memory).  REC: any struct whose first member is the int64 time.
template <typename REC>
EARHIP_SEARCH_HD int upper_bound_rec_window(const REC *rec, int n, int64_t first, int64_t last, int64_t v) {
  if (v < first) return 0;
  if (v >= last) return n;
  if (n < 16) {  // (few points: bisect — first <= v < last here)
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (rec[mid].time <= v) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  }
  int g = (int)((double)(v - first) / (double)(last - first) * (double)(n - 1));
  int lo = g - 3;
  lo = lo < 0 ? 0 : (lo > n - 8 ? n - 8 : lo);
  int64_t w[8];
  for (int i = 0; i < 8; i++) w[i] = rec[lo + i].time;
  if (w[0] <= v && v < w[7]) {
    int k = lo;
    for (int i = 0; i < 8; i++) k += w[i] <= v ? 1 : 0;
    return k;
  }
  // the window missed: bracket from the guess and bisect (upper_bound_time_guess on the records)
  g = g < 0 ? 0 : (g > n - 2 ? n - 2 : g);
  int blo, bhi;  // invariant: rec[blo].time <= v < rec[bhi].time
  if (rec[g].time <= v) {
    blo = g;
    int step = 1;
    bhi = g + 1;
    while (bhi < n - 1 && rec[bhi].time <= v) {
      blo = bhi;
      step <<= 1;
      bhi = bhi + step > n - 1 ? n - 1 : bhi + step;
    }
  } else {
    bhi = g;
    int step = 1;
    blo = g - 1;
    while (blo > 0 && rec[blo].time > v) {
      bhi = blo;
      step <<= 1;
      blo = blo - step < 0 ? 0 : blo - step;
    }
  }
  while (bhi - blo > 1) {
    const int mid = (blo + bhi) >> 1;
    if (rec[mid].time <= v) blo = mid;
    else bhi = mid;
  }
  return bhi;
}

}  // namespace earhip
