// libear_amd/csrc/host_gather.h on the CPU: the streaming-store copy of the staging path against memcpy (every size, every
// misalignment of source and destination, guard floats around the destination), the staging threads' default count, and the
// NUMA helpers' behaviour on whatever host this runs on (they must answer or decline, never fail).
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "host_gather.h"

int main() {
  using namespace earhip;
  int bad = 0;
  std::mt19937 rng(7);
  std::vector<float> src(9000), dst(9000), ref(9000);
  for (size_t i = 0; i < src.size(); i++) src[i] = (float)(int)(rng() % 100000) * 0.25f;
  long cases = 0;
  for (int rep = 0; rep < 4000; rep++) {
    const size_t n = rep < 200 ? (size_t)rep : rng() % 4500;
    const size_t so = rng() % 37, d_o = 16 + rng() % 37;
    std::fill(dst.begin(), dst.end(), -1.0f);
    std::fill(ref.begin(), ref.end(), -1.0f);
    stream_copy(dst.data() + d_o, src.data() + so, n);
#if defined(__x86_64__)
    _mm_sfence();
#endif
    std::memcpy(ref.data() + d_o, src.data() + so, n * sizeof(float));
    if (std::memcmp(dst.data(), ref.data(), dst.size() * sizeof(float)) != 0) {
      if (bad < 5) printf("stream_copy differs: n=%zu src+%zu dst+%zu\n", n, so, d_o);
      bad++;
    }
    cases++;
  }
  const int th = default_staging_threads();
  if (th < 2 || th > 8) printf("default_staging_threads() = %d\n", th), bad++;
  NumaMap nm;
  std::vector<float> rows(1 << 16, 1.0f);  // (touched: its pages exist)
  const float *chan[3] = {rows.data(), rows.data() + 20000, rows.data() + 40000};
  const int node = NumaMap::rows_node(chan, 3, 20000);
  const int pn = NumaMap::page_node(rows.data());
  if (node < -1 || pn < -1) printf("node %d page node %d\n", node, pn), bad++;
  if (nm.ok && pn >= 0) {
    const cpu_set_t *cs = nm.cpus_of(pn);
    if (cs && CPU_COUNT(cs) < 1) printf("empty cpu set for node %d\n", pn), bad++;
    if (cs) {  // a node's CPUs lie inside the process's own mask
      for (int c = 0; c < CPU_SETSIZE; c++)
        if (CPU_ISSET(c, cs) && !CPU_ISSET(c, &nm.allowed)) {
          printf("cpu %d of node %d outside the allowed set\n", c, pn), bad++;
          break;
        }
    }
  }
  if (nm.cpus_of(-1) != nullptr || nm.cpus_of(100000) != nullptr) printf("cpus_of accepts a node that cannot exist\n"), bad++;
  printf("%ld copies checked, staging threads %d, rows on node %d (page node %d), numa map %s: %d problem(s)\n", cases, th, node, pn,
         nm.ok ? "ok" : "unavailable", bad);
  return bad ? 1 : 0;
}
