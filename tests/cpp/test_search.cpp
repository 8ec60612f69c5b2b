// CPU unit test of libear_amd/csrc/search.h: the guess-started, the windowed and the record-window search return exactly what the
// plain upper bound returns (= libear's find_block, gain_interpolator.hpp:110-129) for sorted
// times with duplicates, clusters, evenly spaced grids and queries before / inside / past them.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "search.h"

int main() {
  srand(1);
  long bad = 0, tests = 0;
  for (int it = 0; it < 20000; it++) {
    const int n = 1 + rand() % 60;
    std::vector<int64_t> t(n);
    const int mode = rand() % 5;
    int64_t cur = (rand() % 2000) - 1000;
    if (mode == 4) cur = (int64_t)1 << 61;  // huge times: the guess arithmetic must not matter
    for (int i = 0; i < n; i++) {
      const int64_t step = mode == 0 ? 512 : mode == 1 ? (rand() % 3 == 0 ? 0 : rand() % 1000)
                           : mode == 2 ? (i < n / 2 ? 1 : 100000) : rand() % 5;
      cur += step;
      t[i] = cur;
    }
    for (int q = 0; q < 40; q++) {
      int64_t v = t[0] - 50 + (int64_t)((double)rand() / RAND_MAX * (double)(t[n - 1] - t[0] + 100));
      if (q % 7 == 0) v = t[rand() % n];
      const int a = earhip::upper_bound_time(t.data(), n, v);
      const int b = earhip::upper_bound_time_guess(t.data(), n, v);
      const int c = earhip::upper_bound_time_window(t.data(), n, v);
      // ... and the search over packed records with the end points handed in (the list builders': an object's header)
      struct Rec { int64_t time; float scale; uint32_t flat; };
      std::vector<Rec> rec(n);
      for (int i = 0; i < n; i++) rec[i] = Rec{t[i], 0.0f, 0u};
      const int d = earhip::upper_bound_rec_window(rec.data(), n, t[0], t[n - 1], v);
      tests++;
      if (a != b || a != c || a != d) {
        if (bad < 5) printf("mismatch n=%d v=%lld a=%d b=%d\n", n, (long long)v, a, b);
        bad++;
      }
    }
  }
  printf("%ld mismatches of %ld\n", bad, tests);
  return bad != 0;
}
