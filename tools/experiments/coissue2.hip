// coissue2.hip — how many independent VALU instructions hide behind one bf16 MFMA issued by the
// SAME wave (one wave per SIMD), for the 16x16x32 and 32x32x16 shapes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) short frag8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;

template <int SHAPE, int NV>
__global__ void __launch_bounds__(256) k(float *sink, int iters) {
  const int l = threadIdx.x & 63;
  u4 av = {0x3f803f80u + l, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  frag8 a = __builtin_bit_cast(frag8, av);
  f32x4 acc4[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
  f32x16 acc16[2];
  for (int i = 0; i < 16; i++) { acc16[0][i] = 0; acc16[1][i] = 0; }
  float v[8];
  for (int i = 0; i < 8; i++) v[i] = (float)(l + i);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (SHAPE == 16) acc4[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc4[c], 0, 0, 0);
      else if (c < 2) acc16[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc16[c], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV; i++) v[(c * NV + i) & 7] = __builtin_fmaf(v[(c * NV + i) & 7], 1.0000001f, 0.5f);
      // keep the VALU ops next to their MFMA
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float out = acc4[0][0] + acc4[1][1] + acc4[2][2] + acc4[3][3] + acc16[0][0] + acc16[1][5];
  for (int i = 0; i < 8; i++) out += v[i];
  sink[blockIdx.x * 256 + threadIdx.x] = out;
}
template <int SHAPE, int NV>
void run(float *sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  k<SHAPE, NV><<<256, 256>>>(sink, 10);
  hipEventRecord(e0);
  k<SHAPE, NV><<<256, 256>>>(sink, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double nm = SHAPE == 16 ? 4.0 * iters : 2.0 * iters;
  printf("mfma %dx%d, %d VALU per loop quarter: %.3f ms, %.2f ns per MFMA\n", SHAPE, SHAPE, NV, ms, ms * 1e6 / nm);
}
int main() {
  float *sink; hipMalloc(&sink, 256 * 256 * 4);
  run<16, 0>(sink); run<16, 1>(sink); run<16, 2>(sink); run<16, 3>(sink); run<16, 4>(sink); run<16, 6>(sink);
  run<32, 0>(sink); run<32, 2>(sink); run<32, 4>(sink); run<32, 6>(sink); run<32, 8>(sink);
  return 0;
}
