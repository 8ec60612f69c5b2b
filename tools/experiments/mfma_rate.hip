// mfma_rate.hip — issue-rate probe for v_mfma_f32_16x16x32_bf16 dependency patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) short frag8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// NCH independent accumulator chains, round-robin; ZERO: every 6th MFMA of a chain starts from C = 0
template <int NCH, bool ZERO>
__global__ void __launch_bounds__(256) k_chain(float *sink, int iters) {
  const int l = threadIdx.x;
  u4 av = {0x3f803f80u + l, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  u4 bv = {0x3f803f80u, 0x3f803f80u + l, 0x3f803f80u, 0x3f803f80u};
  frag8 a = __builtin_bit_cast(frag8, av), b = __builtin_bit_cast(frag8, bv);
  f32x4 acc[NCH], tot[NCH];
  const f32x4 zero = {0, 0, 0, 0};
  for (int c = 0; c < NCH; c++) { acc[c] = zero; tot[c] = zero; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 6; k++)
#pragma unroll
      for (int c = 0; c < NCH; c++)
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, (ZERO && k == 0) ? zero : acc[c], 0, 0, 0);
    if (ZERO) {
#pragma unroll
      for (int c = 0; c < NCH; c++) tot[c] += acc[c];
    }
  }
  float s = 0;
  for (int c = 0; c < NCH; c++) s += acc[c][0] + tot[c][1];
  sink[blockIdx.x * blockDim.x + l] = s;
}

template <typename K>
static void run(const char *name, K kern, int nch, float *sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves = 1; waves <= 2; waves++) {
    const int iters = 4000, blocks = 256;  // one WG per CU, 4*waves waves per CU
    kern<<<blocks, 256 * waves>>>(sink, 10);
    hipEventRecord(e0);
    kern<<<blocks, 256 * waves>>>(sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)waves * iters * 6 * nch;
    printf("%-28s %d wave(s)/SIMD: %.3f ms  %.1f ns per MFMA per SIMD (16 cyc = %.1f ns @2.3GHz)\n", name, waves, ms,
           ms * 1e6 / mfma_per_simd, 16 / 2.3);
  }
}
int main() {
  float *sink; CK(hipMalloc(&sink, 1024 * 512 * 4));
  run("12 chains", k_chain<12, false>, 12, sink);
  run("4 chains", k_chain<4, false>, 4, sink);
  run("2 chains", k_chain<2, false>, 2, sink);
  run("1 chain", k_chain<1, false>, 1, sink);
  run("4 chains, zero-start + fold", k_chain<4, true>, 4, sink);
  run("8 chains, zero-start + fold", k_chain<8, true>, 8, sink);
  return 0;
}
