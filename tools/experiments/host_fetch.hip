// host_fetch.hip — block mode's transfer: 2 MB of device-reachable rows to HBM by the copy engine (hipMemcpy2DAsync, what
// earhip_render_process does for short calls) against a kernel that reads the rows over PCIe itself (a wave per row piece).
// Wall time of enqueue + hipStreamSynchronize per call, median of 200; with and without a dependent kernel behind it.
//   hipcc -O3 --offload-arch=gfx950 -o host_fetch host_fetch.hip && ./host_fetch
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);  \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid = (M / rows_per_wg), block = 64 * rows_per_wg: a wave = one row, n4 float4s of it
__global__ void k_fetch(const float *src, size_t src_stride, float *dst, size_t dst_stride, int n4, int M) {
  const int lane = threadIdx.x & 63, m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (m >= M) return;
  const f32x4 *s = reinterpret_cast<const f32x4 *>(src + (size_t)m * src_stride);
  f32x4 *d = reinterpret_cast<f32x4 *>(dst + (size_t)m * dst_stride);
  for (int i = lane; i < n4; i += 256) {
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (i + 64 * k < n4) v[k] = __builtin_nontemporal_load(s + i + 64 * k);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (i + 64 * k < n4) d[i + 64 * k] = v[k];
  }
}
// grid-stride over all float4s (rows contiguous in dst): many more requests in flight per CU
__global__ void k_fetch_flat(const float *src, size_t src_stride, float *dst, int n4, int M) {
  const size_t total = (size_t)n4 * M;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / n4, j = i - m * n4;
    reinterpret_cast<f32x4 *>(dst)[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + m * src_stride) + j);
  }
}
__global__ void k_touch(const float *p, float *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = p[i * 16] + 1.0f;
}

int main() {
  const int M = 1024, n = 512;
  const size_t strides[2] = {(size_t)n, (size_t)48000};
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *dev, *out;
  CK(hipMalloc(&dev, sizeof(float) * M * n));
  CK(hipMalloc(&out, sizeof(float) * 65536));
  for (size_t stride : strides) {
    float *host;
    CK(hipHostMalloc(&host, sizeof(float) * stride * M, hipHostMallocDefault));
    for (size_t i = 0; i < stride * M; i++) host[i] = (float)(i & 1023);
    auto run = [&](const char *name, auto &&enqueue) -> int {
      std::vector<double> t;
      for (int it = 0; it < 220; it++) {
        const auto a = std::chrono::steady_clock::now();
        enqueue();
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
        const auto b = std::chrono::steady_clock::now();
        if (it >= 20) t.push_back(std::chrono::duration<double, std::micro>(b - a).count());
      }
      std::sort(t.begin(), t.end());
      printf("row stride %6zu  %-44s median %7.1f us  p95 %7.1f us\n", stride, name, t[t.size() / 2], t[t.size() * 95 / 100]);
      return 0;
    };
    auto touch = [&] { hipLaunchKernelGGL(k_touch, dim3(128), dim3(256), 0, st, dev, out, 32768); };
    auto dma = [&] {
      (void)hipMemcpy2DAsync(dev, sizeof(float) * n, host, sizeof(float) * stride, sizeof(float) * n, M, hipMemcpyHostToDevice, st);
    };
    if (run("hipMemcpy2DAsync", dma)) return 1;
    if (run("hipMemcpy2DAsync + dependent kernel", [&] { dma(); touch(); })) return 1;
    if (stride == (size_t)n && run("hipMemcpyAsync (linear)", [&] { (void)hipMemcpyAsync(dev, host, sizeof(float) * n * M, hipMemcpyHostToDevice, st); })) return 1;
    for (int rows : {1, 4, 16}) {
      char name[96];
      snprintf(name, sizeof name, "k_fetch, %d row(s) per workgroup", rows);
      auto kf = [&] { hipLaunchKernelGGL(k_fetch, dim3(M / rows), dim3(64 * rows), 0, st, host, stride, dev, (size_t)n, n / 4, M); };
      if (run(name, kf)) return 1;
    }
    for (int wgs : {256, 1024}) {
      char name[96];
      snprintf(name, sizeof name, "k_fetch_flat, %d workgroups of 256", wgs);
      if (run(name, [&] { hipLaunchKernelGGL(k_fetch_flat, dim3(wgs), dim3(256), 0, st, host, stride, dev, n / 4, M); })) return 1;
    }
    if (run("k_fetch (4 rows) + dependent kernel", [&] {
          hipLaunchKernelGGL(k_fetch, dim3(M / 4), dim3(256), 0, st, host, stride, dev, (size_t)n, n / 4, M);
          touch();
        }))
      return 1;
    if (run("dependent kernel alone", touch)) return 1;
    CK(hipHostFree(host));
  }
  return 0;
}
