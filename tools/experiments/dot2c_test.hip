#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2c(uint32_t h, uint32_t m, float acc) {
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(h), "v"(m));
  return acc;
}
__global__ void k(float *o, const float *a, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float a0 = a[2 * i], a1 = a[2 * i + 1];
  f2 v = {a0, a1};
  uint32_t H = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));
  float r0 = dot2c(H, 0x0000bf80u, a0);   // lo = -1.0, hi = 0
  float r1 = dot2c(H, 0xbf800000u, a1);   // lo = 0, hi = -1.0
  o[2 * i] = r0; o[2 * i + 1] = r1;
}
int main() {
  const int n = 1 << 20;
  float *a, *o; hipMallocManaged(&a, n * 8); hipMallocManaged(&o, n * 8);
  uint64_t st = 88172645463325252ull;
  for (int i = 0; i < 2 * n; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    uint32_t bits = (uint32_t)(st >> 32); bits &= 0xbfffffffu; // avoid inf/nan-ish huge exponents
    float f; memcpy(&f, &bits, 4); if (!std::isfinite(f)) f = 1.0f; a[i] = (i % 7 == 0) ? f : (float)((double)(st & 0xffffff) / 16777216.0 * 2 - 1); }
  k<<<n / 256, 256>>>(o, a, n); hipDeviceSynchronize();
  long bad = 0;
  for (int i = 0; i < 2 * n; i++) {
    uint32_t b; memcpy(&b, &a[i], 4);
    uint32_t r = b + 0x7fffu + ((b >> 16) & 1); r &= 0xffff0000u;  // RNE to bf16
    float hi; memcpy(&hi, &r, 4);
    float want = a[i] - hi;
    if (!(want == o[i]) && std::isfinite(want)) { if (bad < 5) printf("a=%g hi=%g want=%g got=%g\n", a[i], hi, want, o[i]); bad++; }
  }
  printf("mismatches: %ld of %d\n", bad, 2 * n);
  return 0;
}
