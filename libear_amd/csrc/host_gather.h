// host_gather.h — host-side helpers of the long host-pointer calls' staging path (api_render.hip): where the caller's rows live
// (NUMA), how many staging threads a process may usefully run (affinity mask, cgroup CPU quota), and the copy into the pinned staging
// buffer with streaming stores.  Plain C++17 + Linux interfaces, no HIP: unit-tested on the CPU (tests/cpp/test_host_gather.cpp).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <emmintrin.h>
#endif
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace earhip {

// Where a range of the caller's memory lives (NUMA node; -1: unknown, or its pages sit on several nodes) and which CPUs are that
// node's: option HOST_BIND = 1 runs the staging threads of long host-pointer calls on the node of the rows they read.
// A staging thread that runs on a node remote to BOTH the caller's pages and the pinned staging buffer moves 38 GB/s where any
// other placement moves 46-48 (two-socket host, `tools/host_stream_numa.py`).  Not the default: where the scheduler had the threads
// well placed already, binding them to the rows' node measured 1.4 % slower (`tools/host_stream_ab.py`).  Plain Linux interfaces: move_pages(2) with no target nodes only reports, /sys lists a node's CPUs.
struct NumaMap {
  cpu_set_t allowed;                 // the CPUs this process may use at all (cpusets, taskset)
  std::vector<cpu_set_t> node_cpus;  // per node: its CPUs among the allowed ones (empty set: unknown / none)
  std::vector<char> known;
  bool ok = false;
  NumaMap() {
    CPU_ZERO(&allowed);
    ok = sched_getaffinity(0, sizeof(allowed), &allowed) == 0;
    known.reserve(64), node_cpus.reserve(64);  // (the staging threads hold pointers to a node's set while a job runs)
  }
  static int page_node(const void *p) {
    void *pg = reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)4095);
    int status = -1;
    const long rc = syscall(SYS_move_pages, 0, 1ul, &pg, nullptr, &status, 0);
    return rc == 0 && status >= 0 ? status : -1;
  }
  // the node of the first, a middle and the last row of a call's inputs when they agree
  static int rows_node(const float *const *in, int M, size_t n) {
    const int a = page_node(in[0]), b = page_node(in[M / 2] + n / 2), c = page_node(in[M - 1] + (n ? n - 1 : 0));
    return a >= 0 && a == b && b == c ? a : -1;
  }
  const cpu_set_t *cpus_of(int node) {
    if (!ok || node < 0 || node >= 1024) return nullptr;
    if ((size_t)node >= known.size()) known.resize(node + 1, 0), node_cpus.resize(node + 1);
    if (!known[node]) {
      known[node] = 1;
      CPU_ZERO(&node_cpus[node]);
      char path[96];
      snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
      if (FILE *f = fopen(path, "r")) {
        int a = 0, b = 0;
        for (;;) {  // "0-63,128-191"
          if (fscanf(f, "%d", &a) != 1) break;
          b = a;
          int ch = fgetc(f);
          if (ch == '-') {
            if (fscanf(f, "%d", &b) != 1) break;
            ch = fgetc(f);
          }
          for (int c = a; c <= b && c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &allowed)) CPU_SET(c, &node_cpus[node]);
          if (ch != ',') break;
        }
        fclose(f);
      }
    }
    return CPU_COUNT(&node_cpus[node]) > 0 ? &node_cpus[node] : nullptr;
  }
};

// A row piece into the pinned staging buffer with NON-TEMPORAL stores: the buffer is written once and read by the copy engine,
// never by a CPU — ordinary stores first READ every destination line into the cache (16 KB pieces are far below the size at which
// memcpy switches to streaming stores itself), a third of the gather's memory traffic and the whole of its cache footprint.
// (SSE2: baseline x86-64.  The caller fences before it publishes the chunk.)
static inline void stream_copy(float *dst, const float *src, size_t nfloats) {
#if defined(__x86_64__)
  const size_t head = std::min(nfloats, (size_t)(((64 - (reinterpret_cast<uintptr_t>(dst) & 63)) & 63) / sizeof(float)));
  if (head) std::memcpy(dst, src, head * sizeof(float));
  dst += head, src += head, nfloats -= head;
  const size_t lines = nfloats / 16;
  const __m128i *s = reinterpret_cast<const __m128i *>(src);
  __m128i *d = reinterpret_cast<__m128i *>(dst);
  for (size_t i = 0; i < lines; i++) {
    const __m128i a = _mm_loadu_si128(s + 4 * i), b = _mm_loadu_si128(s + 4 * i + 1), c = _mm_loadu_si128(s + 4 * i + 2), e = _mm_loadu_si128(s + 4 * i + 3);
    _mm_stream_si128(d + 4 * i, a);
    _mm_stream_si128(d + 4 * i + 1, b);
    _mm_stream_si128(d + 4 * i + 2, c);
    _mm_stream_si128(d + 4 * i + 3, e);
  }
  const size_t done = lines * 16;
  if (nfloats > done) std::memcpy(dst + done, src + done, (nfloats - done) * sizeof(float));
#else
  std::memcpy(dst, src, nfloats * sizeof(float));
#endif
}

// How many staging threads a long host-pointer call gets by default: 8 — the gather is bound by its memory accesses' latency, not by
// cores: 8, 16 and 32 threads move the same bytes per second (tools/host_stream_ab.py: 0.895 / 0.885 / 0.866 of the bus) — but never
// more than the CPUs this process may actually USE less two (the calling thread polls while it waits, the HIP runtime has
// threads of its own): a container sees all of the host's cores (hardware_concurrency: 256) while its cgroup grants it 16 CPUs'
// worth of time, and threads beyond the grant get the whole group throttled in the middle of a call (32 threads: 0.62).
static inline int default_staging_threads() {
  double cpus = (double)std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) cpus = std::min(cpus, (double)CPU_COUNT(&set));
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota us | max> <period us>"
    char q[32] = {0};
    long period = 0;
    if (fscanf(f, "%31s %ld", q, &period) == 2 && period > 0 && q[0] != 'm') cpus = std::min(cpus, (double)atol(q) / (double)period);
    fclose(f);
  } else {  // cgroup v1
    long quota = -1, period = 0;
    if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      if (fscanf(g, "%ld", &quota) != 1) quota = -1;
      fclose(g);
    }
    if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (fscanf(g, "%ld", &period) != 1) period = 0;
      fclose(g);
    }
    if (quota > 0 && period > 0) cpus = std::min(cpus, (double)quota / (double)period);
  }
  return (int)std::max(2.0, std::min(8.0, cpus - 2.0));
}

}  // namespace earhip
