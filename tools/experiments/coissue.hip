// coissue.hip — do a VALU-only wave and an MFMA-only wave on the SAME SIMD overlap?
// 512-thread workgroups (8 waves, one workgroup per CU -> 2 waves per SIMD).  Each wave reads
// its SIMD id (HW_REG_HW_ID bits 5:4); the first arrival on a SIMD runs MFMAs, the second VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) short frag8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode bit0: MFMA waves work, bit1: VALU waves work; pairing 0: by SIMD id, 1: wave index >> 2
__global__ void __launch_bounds__(512) k(float *sink, int *ids, int iters_m, int iters_v, int mode, int pairing) {
  __shared__ int cnt[4];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
  __syncthreads();
  int slot = 0, simd = 0;
  if (l == 0) {
    simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;
    slot = atomicAdd(&cnt[simd], 1);
  }
  simd = __builtin_amdgcn_readfirstlane(simd);
  slot = __builtin_amdgcn_readfirstlane(slot);
  if (blockIdx.x == 0 && l == 0) { ids[w * 2] = simd; ids[w * 2 + 1] = slot; }
  const int role = pairing == 0 ? (slot & 1) : (w >> 2);
  float out = 0;
  if (role == 0) {
    if (mode & 1) {
      u4 av = {0x3f803f80u + l, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
      frag8 a = __builtin_bit_cast(frag8, av);
      f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
      for (int it = 0; it < iters_m; it++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc[c], 0, 0, 0);
      out = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    }
  } else {
    if (mode & 2) {
      float v[8];
      for (int i = 0; i < 8; i++) v[i] = (float)(l + i);
      for (int it = 0; it < iters_v; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = __builtin_fmaf(v[i], 1.0000001f, 0.5f);
      for (int i = 0; i < 8; i++) out += v[i];
    }
  }
  sink[blockIdx.x * 512 + threadIdx.x] = out;
}

int main() {
  float *sink; int *ids; CK(hipMalloc(&sink, 256 * 512 * 4)); CK(hipMalloc(&ids, 64));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int im = 20000, iv = 20000;  // 80k MFMAs (16 cyc each) vs 160k VALU (4 cyc each): ~equal alone
  for (int pairing = 0; pairing < 2; pairing++)
    for (int mode = 1; mode <= 3; mode++) {
      k<<<256, 512>>>(sink, ids, 10, 10, mode, pairing);
      hipEventRecord(e0);
      k<<<256, 512>>>(sink, ids, im, iv, mode, pairing);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("pairing %s mode %d (%s): %.3f ms\n", pairing == 0 ? "by SIMD id" : "wave>>2", mode,
             mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", ms);
    }
  int h[16]; CK(hipMemcpy(h, ids, 64, hipMemcpyDeviceToHost));
  printf("wave: (simd, slot):");
  for (int w = 0; w < 8; w++) printf(" %d:(%d,%d)", w, h[2 * w], h[2 * w + 1]);
  printf("\n");
  return 0;
}
