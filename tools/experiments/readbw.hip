// readbw.hip — achievable HBM read bandwidth on this box: (a) linear streaming read, (b) the
// gain stage's pattern: every workgroup reads a 1 KB piece of each of 1024 rows, rows 2 MB apart.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_linear(const f32x4 *in, float *out, size_t n4, int nt) {
  f32x4 acc = {0, 0, 0, 0};
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t step = (size_t)gridDim.x * 256;
  for (; i < n4; i += step) acc += nt ? __builtin_nontemporal_load(in + i) : in[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
// workgroup b owns samples [256 b, 256 b + 256) of every row; each wave reads 8 rows per step
// (lane: 16 lanes x 16 B = 256 B of 4 rows), `depth` steps in flight
template <int DEPTH>
__global__ void __launch_bounds__(256) k_rows(const float *in, float *out, int rows, size_t stride) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 15, kg = lane >> 4;
  const float *base = in + (size_t)blockIdx.x * 256 + w * 64 + li * 4;
  f32x4 acc = {0, 0, 0, 0};
  f32x4 ring[DEPTH][8];
  for (int d = 0; d < DEPTH; d++)
    for (int q = 0; q < 8; q++)
      ring[d][q] = __builtin_nontemporal_load((const f32x4 *)(base + (size_t)(d * 32 + kg * 8 + q) * stride));
  for (int r0 = 0; r0 < rows; r0 += 32 * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
#pragma unroll
      for (int q = 0; q < 8; q++) acc += ring[d][q];
      const int rn = r0 + 32 * DEPTH + d * 32;
      if (rn < rows)
#pragma unroll
        for (int q = 0; q < 8; q++)
          ring[d][q] = __builtin_nontemporal_load((const f32x4 *)(base + (size_t)(rn + kg * 8 + q) * stride));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipEventRecord(e0); for (int i = 0; i < 5; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  const int rows = 1024; const size_t stride = 524288; const size_t n = rows * stride;
  float *in, *out; hipMalloc(&in, n * 4); hipMalloc(&out, 4096 * 256 * 4 * 4); hipMemset(in, 0, n * 4);
  for (int nt = 0; nt < 2; nt++) for (int blocks : {2048, 8192}) {
    float ms = timeit([&] { k_linear<<<blocks, 256>>>((const f32x4 *)in, out, n / 4, nt); });
    printf("linear read, %5d workgroups, %s: %.3f ms = %.2f TB/s\n", blocks, nt ? "nt" : "default", ms, n * 4 / ms / 1e9);
  }
  // limited occupancy: 256 * k workgroups of 4 waves -> k workgroups per CU resident at a time,
  // each covering 2048 / (256 k) tiles in sequence would change the pattern; instead launch the full
  // grid but cap residency with dynamic LDS (160 KB / k per workgroup)
  for (int k : {1, 2, 4, 8}) {
    const size_t lds = (size_t)(150 * 1024) / k;
    hipFuncSetAttribute((const void *)k_rows<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void *)k_rows<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void *)k_rows<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float m1 = timeit([&] { k_rows<1><<<2048, 256, lds>>>(in, out, rows, stride); });
    float m2 = timeit([&] { k_rows<2><<<2048, 256, lds>>>(in, out, rows, stride); });
    float m4 = timeit([&] { k_rows<4><<<2048, 256, lds>>>(in, out, rows, stride); });
    printf("row pattern, %d workgroup(s) = %2d waves per CU: 8 KB/wave in flight %.2f TB/s, 16 KB %.2f, 32 KB %.2f\n", k, 4 * k,
           n * 4 / m1 / 1e9, n * 4 / m2 / 1e9, n * 4 / m4 / 1e9);
  }
  return 0;
}
