// fma_mix_probe.hip — semantics check of v_fma_mixlo_f16 / v_fma_mixhi_f16 as the split-operand kernels use them:
// l = f16(fma(f32(h.half), k, s)) written to one half of a register, the other half kept.
// hipcc --offload-arch=gfx950 -O3 fma_mix_probe.hip -o fma_mix_probe && ./fma_mix_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mixlo(unsigned &d, unsigned h, int half, float k, float s) {
  if (half) asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(h), "s"(k), "v"(s));
  else asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(h), "s"(k), "v"(s));
}
__device__ __forceinline__ void mixhi(unsigned &d, unsigned h, int half, float k, float s) {
  if (half) asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(h), "s"(k), "v"(s));
  else asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(h), "s"(k), "v"(s));
}

__global__ void k_probe(const float *x, unsigned *out, float xs, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = x[2 * i] * xs, b = x[2 * i + 1] * xs;
  const unsigned H = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, h2));
  const float kk = -2048.0f;
  unsigned L = 0xdeadbeefu;
  mixlo(L, H, 0, kk, a * 2048.0f);
  mixhi(L, H, 1, kk, b * 2048.0f);
  out[2 * i] = H;
  out[2 * i + 1] = L;
}

int main() {
  const int n = 1 << 16;
  std::vector<float> x(2 * n);
  srand(3);
  for (auto &v : x) v = ldexpf((float)rand() / RAND_MAX * 2.0f - 1.0f, -(rand() % 24));
  float *dx;
  unsigned *dout;
  hipMalloc(&dx, sizeof(float) * 2 * n);
  hipMalloc(&dout, sizeof(unsigned) * 2 * n);
  hipMemcpy(dx, x.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice);
  const float xs = 128.0f;
  hipLaunchKernelGGL(k_probe, dim3(n / 256), dim3(256), 0, 0, dx, dout, xs, n);
  std::vector<unsigned> out(2 * n);
  hipMemcpy(out.data(), dout, sizeof(unsigned) * 2 * n, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n && bad < 10; i++) {
    for (int j = 0; j < 2; j++) {
      const float s = x[2 * i + j] * xs;
      const _Float16 h = (_Float16)s;
      const _Float16 l = (_Float16)((s - (float)h) * 2048.0f);
      const unsigned short hb = (unsigned short)(out[2 * i] >> (16 * j)), lb = (unsigned short)(out[2 * i + 1] >> (16 * j));
      unsigned short hw, lw;
      memcpy(&hw, &h, 2);
      memcpy(&lw, &l, 2);
      if (hb != hw || lb != lw) {
        printf("mismatch at %d/%d: s %g  h %04x/%04x  l %04x/%04x\n", i, j, s, hb, hw, lb, lw);
        bad++;
      }
    }
  }
  printf("fma_mix probe: %s\n", bad ? "FAIL" : "ok");
  return bad ? 1 : 0;
}
