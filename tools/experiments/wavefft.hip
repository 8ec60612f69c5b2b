// wavefft.hip — one 1024-point complex FFT per WAVE: Stockham radices 16 x 16 x 4, 16 values per
// lane (indices lane + 64 m, the same layout on input and output), two exchanges through
// wave-private LDS, no workgroup barriers.  Correctness vs a double DFT, and throughput.
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>

struct cf { float x, y; };
__host__ __device__ inline cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }
__host__ __device__ inline cf add(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
__host__ __device__ inline cf sub(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
__host__ __device__ inline cf mul(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <int DIR> __host__ __device__ inline cf muli(cf a) { return DIR > 0 ? mk(-a.y, a.x) : mk(a.y, -a.x); }  // * (DIR * i)
template <int DIR> __host__ __device__ inline cf tw(cf w) { return DIR > 0 ? mk(w.x, -w.y) : w; }  // table holds exp(-2 pi i t / N)

// radix-4 butterfly on v[s], v[s + st], v[s + 2 st], v[s + 3 st] (in place, natural order out)
template <int DIR>
__device__ __forceinline__ void r4(cf &a, cf &b, cf &c, cf &d) {
  const cf a0 = add(a, c), a1 = sub(a, c), a2 = add(b, d), a3 = muli<DIR>(sub(b, d));
  a = add(a0, a2); b = add(a1, a3); c = sub(a0, a2); d = sub(a1, a3);
}
// 16-point DFT of v[0..15] in registers: out[k] = sum_n v[n] W16^{nk}; result in natural order in v
template <int DIR>
__device__ __forceinline__ void dft16(cf (&v)[16]) {
  // n = 4 n1 + n2, k = k1 + 4 k2: first radix-4 over n1 (stride 4), twiddle W16^{n2 k1}, then radix-4 over n2
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r2 = 0.70710678118654752f;
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) r4<DIR>(v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]);  // -> index k1 at n2 + 4 k1
  // twiddles W16^{n2*k1} = exp(-2 pi i n2 k1 / 16) (DIR < 0)
  const cf W1 = mk(c1, -s1), W2 = mk(r2, -r2), W3 = mk(s1, -c1), W6 = mk(-r2, -r2), W9 = mk(-c1, s1);
  v[1 + 4] = mul(v[1 + 4], tw<DIR>(W1)); v[2 + 4] = mul(v[2 + 4], tw<DIR>(W2)); v[3 + 4] = mul(v[3 + 4], tw<DIR>(W3));
  v[1 + 8] = mul(v[1 + 8], tw<DIR>(W2)); v[2 + 8] = muli<DIR>(v[2 + 8]) /* W4 = -i */; v[3 + 8] = mul(v[3 + 8], tw<DIR>(W6));
  v[1 + 12] = mul(v[1 + 12], tw<DIR>(W3)); v[2 + 12] = mul(v[2 + 12], tw<DIR>(W6)); v[3 + 12] = mul(v[3 + 12], tw<DIR>(W9));
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) r4<DIR>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);  // -> k2 at 4 k1 + k2
  // now v[4 k1 + k2] = X[k1 + 4 k2]: transpose the 4x4 index to natural order
  cf t;
#define SW(a, b) t = v[a]; v[a] = v[b]; v[b] = t;
  SW(1, 4) SW(2, 8) SW(3, 12) SW(6, 9) SW(7, 13) SW(11, 14)
#undef SW
}

constexpr int PAD(int i) { return i + (i >> 4); }  // LDS index padding (17-word rows)

// forward (DIR = -1) or inverse (DIR = +1, unnormalised) transform of v (index lane + 64 m) via wave-private LDS
template <int DIR>
__device__ __forceinline__ void fft1024(cf (&v)[16], cf *lds, const cf (&t1)[15], const cf (&t2)[4][3], int lane) {
  // pass 0: radix 16, Ns = 1: butterfly j = lane, inputs in[lane + 64 r], outputs out[16 lane + r]
  dft16<DIR>(v);
#pragma unroll
  for (int r = 0; r < 16; r++) lds[PAD(16 * lane + r)] = v[r];
  // pass 1: radix 16, Ns = 16: inputs in[lane + 64 r] * W256^{r (lane & 15)}, outputs out[(lane >> 4) 256 + (lane & 15) + 16 r]
#pragma unroll
  for (int r = 0; r < 16; r++) v[r] = lds[PAD(lane + 64 * r)];
#pragma unroll
  for (int r = 1; r < 16; r++) v[r] = mul(v[r], tw<DIR>(t1[r - 1]));
  dft16<DIR>(v);
  const int ob = (lane >> 4) * 256 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 16; r++) lds[PAD(ob + 16 * r)] = v[r];
  // pass 2: radix 4, Ns = 256: butterflies j = lane + 64 q: inputs in[j + 256 r] * W1024^{r j}, outputs out[j + 256 r]
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int r = 0; r < 4; r++) v[q + 4 * r] = lds[PAD(lane + 64 * q + 256 * r)];  // out index lane + 64 (q + 4 r)
#pragma unroll
  for (int q = 0; q < 4; q++) {
#pragma unroll
    for (int r = 1; r < 4; r++) v[q + 4 * r] = mul(v[q + 4 * r], tw<DIR>(t2[q][r - 1]));
    r4<DIR>(v[q], v[q + 4], v[q + 8], v[q + 12]);
  }
}

__global__ void __launch_bounds__(256) k_fft(const cf *in, cf *out, const cf *table, int nfft, int reps, int inverse_too) {
  __shared__ cf lds_all[4][1024 + 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  cf *lds = lds_all[w];
  cf t1[15], t2[4][3];
  for (int r = 1; r < 16; r++) t1[r - 1] = table[(4 * r * (lane & 15)) & 1023];  // W256^{r k} = W1024^{4 r k}
  for (int q = 0; q < 4; q++)
    for (int r = 1; r < 4; r++) t2[q][r - 1] = table[(r * (lane + 64 * q)) & 1023];
  for (int f = blockIdx.x * 4 + w; f < nfft; f += gridDim.x * 4) {
    cf v[16];
    for (int m = 0; m < 16; m++) v[m] = in[(size_t)f * 1024 + lane + 64 * m];
    for (int rep = 0; rep < reps; rep++) {
      fft1024<-1>(v, lds, t1, t2, lane);
      if (inverse_too) {
        fft1024<+1>(v, lds, t1, t2, lane);
        for (int m = 0; m < 16; m++) v[m] = mk(v[m].x * (1.0f / 1024), v[m].y * (1.0f / 1024));
      }
    }
    for (int m = 0; m < 16; m++) out[(size_t)f * 1024 + lane + 64 * m] = v[m];
  }
}

int main() {
  const int N = 1024, nfft = 4096;
  std::vector<cf> h((size_t)nfft * N), table(N);
  for (int t = 0; t < N; t++) table[t] = mk((float)std::cos(-2 * M_PI * t / N), (float)std::sin(-2 * M_PI * t / N));
  unsigned s = 12345;
  for (auto &v : h) { s = s * 1664525u + 1013904223u; v.x = (float)(s >> 8) / 8388608.0f - 1; s = s * 1664525u + 1013904223u; v.y = (float)(s >> 8) / 8388608.0f - 1; }
  cf *din, *dout, *dt;
  hipMalloc(&din, h.size() * 8); hipMalloc(&dout, h.size() * 8); hipMalloc(&dt, N * 8);
  hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dt, table.data(), N * 8, hipMemcpyHostToDevice);
  k_fft<<<256, 256>>>(din, dout, dt, nfft, 1, 0);
  std::vector<cf> got(h.size());
  hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost);
  double num = 0, den = 0;
  for (int f = 0; f < 2; f++)
    for (int k = 0; k < N; k++) {
      std::complex<double> acc = 0;
      for (int n = 0; n < N; n++) acc += std::complex<double>(h[(size_t)f * N + n].x, h[(size_t)f * N + n].y) * std::polar(1.0, -2 * M_PI * n * k / N);
      const std::complex<double> g(got[(size_t)f * N + k].x, got[(size_t)f * N + k].y);
      num += std::norm(g - acc); den += std::norm(acc);
    }
  printf("forward rel err vs double DFT: %.3e\n", std::sqrt(num / den));
  k_fft<<<256, 256>>>(din, dout, dt, nfft, 1, 1);
  hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost);
  num = den = 0;
  for (size_t i = 0; i < (size_t)4 * N; i++) { num += (got[i].x - h[i].x) * (got[i].x - h[i].x) + (got[i].y - h[i].y) * (got[i].y - h[i].y); den += h[i].x * h[i].x + h[i].y * h[i].y; }
  printf("forward+inverse round trip rel err: %.3e\n", std::sqrt(num / den));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 50;
  k_fft<<<2048, 256>>>(din, dout, dt, 8192 > nfft ? nfft : 8192, 2, 1);
  hipEventRecord(e0);
  k_fft<<<1024, 256>>>(din, dout, dt, nfft, reps, 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%d x %d (forward + inverse) transforms: %.3f ms -> %.2f ns per pair of transforms (K2 needs 13.5k per step: %.1f us)\n",
         nfft, reps, ms, ms * 1e6 / ((double)nfft * reps), ms * 1e3 / ((double)nfft * reps) * 13536);
  return 0;
}
