// k2_fft_probe.hip — what ONE pair transform of K2 (render_kernels.h: forward FFT-1024, x spectrum, inverse) costs a wave in
// shader cycles when nothing else is in its way: data in registers, twiddles and spectrum in LDS as in k_decorrelate_wave,
// 1 .. 3 waves per SIMD.  Tells how much of K2's ~14k cycles per pair and wave (6.9 us alone on a SIMD) is the transforms.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../libear_amd/csrc k2_fft_probe.hip -o k2_fft_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "common.h"
#include "fft_kernels.h"
#include "render_kernels.h"
using namespace earhip;

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_probe(const cf *tw, const cf *H, unsigned long long *cycles, float *sink, int iters, int what) {
  constexpr int L = 1024;
  __shared__ v2f lds_all[4][L + L / 16];
  __shared__ v2f h_lds[L + L / 16];
  __shared__ v2f t1_lds[16][17];
  __shared__ v2f t2_lds[3 * 256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  auto to_v2f = [](cf c) { return v2f{c.x, c.y}; };
  for (int i = threadIdx.x; i < L; i += 256) h_lds[wave_pad(i)] = to_v2f(H[i]);
  if (threadIdx.x < 240) {
    const int k = threadIdx.x / 15, r = threadIdx.x % 15 + 1;
    t1_lds[k][r - 1] = to_v2f(tw[(4 * r * k) & (L - 1)]);
  }
  for (int i = threadIdx.x; i < 3 * 256; i += 256) t2_lds[i] = to_v2f(tw[((i >> 8) + 1) * (i & 255)]);
  __syncthreads();
  v2f *lds = lds_all[w];
  const v2f *t1 = t1_lds[lane & 15];
  v2f v[16];
  for (int m = 0; m < 16; m++) v[m] = v2f{(float)(lane + m) * 1e-3f, (float)(lane - m) * 1e-3f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (what & 1) wave_fft1024<-1>(v, lds, t1, t2_lds, lane);
    if (what & 2) {
#pragma unroll
      for (int m = 0; m < 16; m++) {
        v[m] = wave_cmul<-1>(v[m], h_lds[wave_pad(lane + 64 * m)]);
        if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (what & 4) wave_fft1024<+1>(v, lds, t1, t2_lds, lane);
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] *= 1.0f / 1024.0f;
  }
  const unsigned long long t1c = __builtin_readcyclecounter();
  float s = 0;
  for (int m = 0; m < 16; m++) s += v[m].x + v[m].y;
  if (s == 123.456f) sink[0] = s;
  if (lane == 0) cycles[blockIdx.x * 4 + w] = t1c - t0;
}

int main() {
  const int L = 1024;
  std::vector<cf> tw(L), H(L);
  for (int t = 0; t < L; t++) {
    const double a = -2.0 * M_PI * t / L;
    tw[t] = cf_make((float)cos(a), (float)sin(a));
    H[t] = cf_make((float)cos(0.37 * t), (float)sin(0.37 * t));
  }
  cf *dtw, *dH;
  unsigned long long *dc;
  float *ds;
  hipMalloc(&dtw, sizeof(cf) * L);
  hipMalloc(&dH, sizeof(cf) * L);
  hipMalloc(&dc, sizeof(unsigned long long) * 4 * 256 * 3);
  hipMalloc(&ds, 64);
  hipMemcpy(dtw, tw.data(), sizeof(cf) * L, hipMemcpyHostToDevice);
  hipMemcpy(dH, H.data(), sizeof(cf) * L, hipMemcpyHostToDevice);
  const int iters = 200;
  for (int wgs_per_cu = 1; wgs_per_cu <= 3; wgs_per_cu++)
    for (int what : {1, 4, 7}) {
      const int blocks = 256 * wgs_per_cu;
      hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, dtw, dH, dc, ds, iters, what);
      hipDeviceSynchronize();
      std::vector<unsigned long long> c(4 * blocks);
      hipMemcpy(c.data(), dc, sizeof(unsigned long long) * 4 * blocks, hipMemcpyDeviceToHost);
      double sum = 0;
      for (auto x : c) sum += (double)x;
      printf("%d wave(s) per SIMD, %s: %.0f cycles per iteration and wave\n", wgs_per_cu,
             what == 1 ? "forward FFT-1024" : what == 4 ? "inverse FFT-1024" : "forward, x H, inverse (a pair transform)", sum / c.size() / iters);
    }
  return 0;
}
