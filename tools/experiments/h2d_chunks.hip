// h2d_chunks.hip — what the chunked H2D of a staged host-pointer call loses against one plain copy: 128 MB from a pinned buffer to
// HBM (a) in one hipMemcpyAsync, (b) in 8 / 16 / 32 back-to-back copies on one stream, (c) with an event behind each copy that
// a second stream waits for (a kernel per chunk there), (d) while CPU threads write another part of the same pinned buffer,
// (e) each chunk freshly written by CPU threads (plain / streaming stores) right before its copy is enqueued.
//   hipcc -O3 --offload-arch=gfx950 -o h2d_chunks h2d_chunks.hip -lpthread && ./h2d_chunks
#include <hip/hip_runtime.h>
#include <emmintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__global__ void k_touch(const float *p, float *out, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i * 1024 < n) out[i & 65535] = p[i * 1024] + 1.0f;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void fill(float *dst, const float *src, size_t n, bool nt) {
  if (!nt) {
    std::memcpy(dst, src, n * 4);
    return;
  }
  const __m128i *s = reinterpret_cast<const __m128i *>(src);
  __m128i *d = reinterpret_cast<__m128i *>(dst);
  for (size_t i = 0; i < n / 4; i++) _mm_stream_si128(d + i, _mm_loadu_si128(s + i));
  _mm_sfence();
}

int main() {
  const size_t total = (size_t)128 << 20, nf = total / 4;
  hipStream_t s_in, s_k;
  CK(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
  float *host, *host2, *dev, *out;
  CK(hipHostMalloc(&host, 2 * total, hipHostMallocDefault));
  host2 = host + nf;
  CK(hipMalloc(&dev, total));
  CK(hipMalloc(&out, 65536 * 4));
  std::vector<float> src(nf);
  for (size_t i = 0; i < nf; i++) src[i] = (float)(i & 1023);
  std::memcpy(host, src.data(), total);
  std::vector<hipEvent_t> ev(64);
  for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto median = [&](const char *name, auto &&body) {
    std::vector<double> t;
    for (int it = 0; it < 12; it++) {
      const double a = now_us();
      body();
      CK(hipStreamSynchronize(s_in));
      CK(hipStreamSynchronize(s_k));
      const double b = now_us();
      if (it >= 2) t.push_back(b - a);
    }
    std::sort(t.begin(), t.end());
    printf("%-78s %8.1f us  %5.1f GB/s  (min %.1f max %.1f)\n", name, t[t.size() / 2], total / t[t.size() / 2] / 1e3, t.front(), t.back());
    return t[t.size() / 2];
  };
  median("(a) one copy of 128 MB", [&] { CK(hipMemcpyAsync(dev, host, total, hipMemcpyHostToDevice, s_in)); });
  for (int nch : {8, 16, 32}) {
    char name[128];
    snprintf(name, sizeof name, "(b) %d copies back to back on one stream", nch);
    median(name, [&] {
      for (int c = 0; c < nch; c++) CK(hipMemcpyAsync(dev + nf / nch * c, host + nf / nch * c, total / nch, hipMemcpyHostToDevice, s_in));
    });
    snprintf(name, sizeof name, "(c) %d copies, an event behind each, a kernel per chunk on a second stream", nch);
    median(name, [&] {
      for (int c = 0; c < nch; c++) {
        CK(hipMemcpyAsync(dev + nf / nch * c, host + nf / nch * c, total / nch, hipMemcpyHostToDevice, s_in));
        CK(hipEventRecord(ev[c], s_in));
        CK(hipStreamWaitEvent(s_k, ev[c], 0));
        hipLaunchKernelGGL(k_touch, dim3(128), dim3(256), 0, s_k, dev + nf / nch * c, out, nf / nch);
      }
    });
  }
  for (int nthreads : {4, 8}) {
    char name[128];
    snprintf(name, sizeof name, "(d) 8 copies while %d CPU threads write the OTHER half of the pinned buffer", nthreads);
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++)
      th.emplace_back([&, t] {
        const size_t per = nf / nthreads;
        while (!stop.load()) fill(host2 + per * t, src.data() + per * t, per, true);
      });
    median(name, [&] {
      for (int c = 0; c < 8; c++) CK(hipMemcpyAsync(dev + nf / 8 * c, host + nf / 8 * c, total / 8, hipMemcpyHostToDevice, s_in));
    });
    stop = true;
    for (auto &t : th) t.join();
  }
  for (int nt = 0; nt < 2; nt++)
    for (int nch : {8, 32}) {
      char name[160];
      snprintf(name, sizeof name, "(e) %d chunks, each written by 8 CPU threads (%s stores) right before its copy", nch, nt ? "streaming" : "plain");
      median(name, [&] {
        for (int c = 0; c < nch; c++) {
          const size_t per = nf / nch / 8;
          std::vector<std::thread> th;
          for (int t = 0; t < 8; t++) th.emplace_back([&, t] { fill(host + nf / nch * c + per * t, src.data() + nf / nch * c + per * t, per, nt != 0); });
          for (auto &t : th) t.join();
          CK(hipMemcpyAsync(dev + nf / nch * c, host + nf / nch * c, total / nch, hipMemcpyHostToDevice, s_in));
        }
      });
    }
  return 0;
}
