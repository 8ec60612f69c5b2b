// pk_rate.hip — issue rate of the packed-f32 VALU instructions K2 (render_kernels.h, k_decorrelate_wave) is made of,
// against plain f32 instructions doing the same arithmetic: cycles per wave-instruction at 1 / 2 / 3 / 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(float *out, int iters, float seed) {
  // 8 independent chains per lane (dependent latency hidden), 32 instructions per loop iteration
  v2f a[8];
  float s[16];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = v2f{seed + i, seed - i};
#pragma unroll
  for (int i = 0; i < 16; i++) s[i] = seed + 0.5f * i;
  const v2f w = {0.999f, 0.001f};
  const float ws = 0.999f, wt = 0.001f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
      if (KIND == 0) {  // 8 v_pk_fma_f32 (plain operands)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(w));
      } else if (KIND == 1) {  // 16 v_fma_f32: the same flops
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(s[(i + 1) & 15]), "v"(ws));
      } else if (KIND == 2) {  // 8 v_pk_mul_f32 with op_sel (the complex multiply's first half)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(w));
      } else if (KIND == 3) {  // 8 v_pk_add_f32 with neg / op_sel (a + i t)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(w));
      } else if (KIND == 4) {  // 8 v_pk_fma_f32 with an SGPR-pair operand and modifiers (wave_cmul_k)
#pragma unroll
        for (int i = 0; i < 8; i++)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(w));
      } else if (KIND == 5) {  // 16 v_mul_f32
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s[i]) : "v"(s[(i + 1) & 15]), "v"(ws));
      } else if (KIND == 6) {  // 16 v_add_f32
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_add_f32 %0, %1, %2" : "=v"(s[i]) : "v"(s[(i + 1) & 15]), "v"(wt));
      } else if ((KIND >= 8 && KIND < 40) || KIND >= 42) {  // 16 of one plain-encoded instruction the operand split is made of (round 5: what a split costs)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          float &d = s[i];
          const float &b = s[(i + 1) & 15];
          if (KIND == 8) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(d) : "v"(b));
          if (KIND == 9) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(d) : "v"(b));
          if (KIND == 10) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(b), "v"(ws));
          if (KIND == 11) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(b), "v"(ws), "v"(wt));
          if (KIND == 12) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(d) : "v"(b), "v"(ws), "v"(wt));
          if (KIND == 13) asm volatile("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(d) : "v"(b), "v"(ws), "v"(wt));
          if (KIND == 14) asm volatile("v_and_b32 %0, %1, %2" : "=v"(d) : "v"(b), "v"(ws));
          if (KIND == 15) asm volatile("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(b), "v"(ws), "v"(wt));
          if (KIND == 16) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(d) : "v"(b), "v"(ws));
          if (KIND == 17) asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(ws), "v"(wt));
          if (KIND == 18) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(ws), "v"(wt));
          if (KIND == 19) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(d) : "v"(b), "v"(ws));
          if (KIND == 20) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(ws), "v"(wt));
          if (KIND == 21) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(d) : "v"(b));
          if (KIND == 22) asm volatile("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(d) : "v"(b), "v"(ws));
          if (KIND == 23) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(d) : "v"(b), "v"(ws));
          if (KIND == 24) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(ws), "v"(wt));
          if (KIND == 25) asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(b), "v"(ws));
          if (KIND == 26) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(d) : "v"(b), "v"(ws));
          if (KIND == 27) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(ws), "v"(wt));
          if (KIND == 28) asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(b));
          if (KIND == 29) asm volatile("v_or_b32 %0, %1, %2" : "=v"(d) : "v"(b), "v"(ws));
          if (KIND == 30) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(d) : "v"(b));
          if (KIND == 31) asm volatile("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(b), "v"(ws));
          if (KIND == 32) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(b), "v"(ws));
          if (KIND == 33) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(ws), "s"(0x5555aaaa5555aaaaull));
          if (KIND == 34) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(ws), "v"(wt));
          if (KIND == 35) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(b), "v"(ws) : "vcc");
          if (KIND == 36) asm volatile("s_mov_b64 vcc, exec\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(d) : "v"(b), "v"(ws) : "vcc");
          if (KIND == 37) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=&v"(d) : "v"(b), "v"(ws));
          if (KIND == 38) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=&v"(d) : "v"(b), "v"(ws) : "vcc");
          // a compare, N plain instructions, then the select on its vcc (3 / 6 / 10 / 18 instructions per item)
          if (KIND == 42) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_add_f32 %0, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=&v"(d) : "v"(b), "v"(ws) : "vcc");
          if (KIND == 43) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=&v"(d) : "v"(b), "v"(ws) : "vcc");
          if (KIND == 44) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=&v"(d) : "v"(b), "v"(ws) : "vcc");
          if (KIND == 45) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_add_f32 %0, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=&v"(d) : "v"(b), "v"(ws) : "vcc");
          // the compare's result in an SGPR pair, the select on it (VOP3 forms)
          if (KIND == 46) asm volatile("v_cmp_gt_f32_e64 s[20:21], %1, %2\n v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=&v"(d) : "v"(b), "v"(ws) : "s20", "s21");
          // one compare, four selects on its vcc
          if (KIND == 47) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %0, %2, %1, vcc\n v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %0, %2, %1, vcc" : "=&v"(d) : "v"(b), "v"(ws) : "vcc");
        }
      } else if (KIND == 40 || KIND == 41) {  // 8 64-bit integer instructions on register pairs
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if (KIND == 40) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(s[i]), "v"(s[i + 8]) : "vcc");
          if (KIND == 41) asm volatile("v_lshl_add_u64 %0, %1, 2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        }
      } else {  // 8 v_pk_add_f32 plain
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(w));
      }
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r += a[i].x + a[i].y;
#pragma unroll
  for (int i = 0; i < 16; i++) r += s[i];
  if (r == 123.456f) out[0] = r;
}

template <int KIND>
static void run(const char *name, int per_iter, float *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps++) {  // waves per SIMD: blocks of 256 threads = one wave per SIMD each
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // instructions per SIMD = wps * iters * 4 reps * per_iter; cycles at an assumed 2.4 GHz and 2.05 GHz
    const double n = (double)wps * iters * 4 * per_iter;
    printf("%-44s %d waves/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (%.2f cyc at 2.4 GHz)\n", name, wps, ms,
           ms * 1e6 / n, ms * 1e6 / n * 2.4);
  }
}

int main() {
  float *out;
  hipMalloc(&out, 64);
  run<1>("v_fma_f32 x16", 16, out);
  run<0>("v_pk_fma_f32 x8 (same flops)", 8, out);
  run<4>("v_pk_fma_f32 x8, SGPR operand + op_sel/neg", 8, out);
  run<2>("v_pk_mul_f32 x8, op_sel", 8, out);
  run<5>("v_mul_f32 x16", 16, out);
  run<3>("v_pk_add_f32 x8, op_sel/neg", 8, out);
  run<7>("v_pk_add_f32 x8 plain", 8, out);
  run<6>("v_add_f32 x16", 16, out);
  run<8>("v_cvt_f32_f16 x16", 16, out);
  run<9>("v_cvt_f32_f16_sdwa WORD_1 x16", 16, out);
  run<10>("v_cvt_pk_f16_f32 x16", 16, out);
  run<19>("v_cvt_pkrtz_f16_f32 x16", 16, out);
  run<11>("v_fma_mix_f32 (f16 lo, f32, f32) x16", 16, out);
  run<12>("v_fma_mixlo_f16 x16", 16, out);
  run<13>("v_fma_mixhi_f16 (c: f16) x16", 16, out);
  run<14>("v_and_b32 x16", 16, out);
  run<15>("v_fma_f32 clamp x16", 16, out);
  run<16>("v_pk_mul_f16 x16", 16, out);
  run<17>("v_pk_fma_f16 x16", 16, out);
  run<18>("v_perm_b32 x16", 16, out);
  run<20>("v_med3_f32 x16", 16, out);
  run<21>("v_cvt_f32_i32 x16", 16, out);
  run<22>("v_lshl_add_u32 x16", 16, out);
  run<23>("v_mul_lo_u32 x16", 16, out);
  run<24>("v_mad_u32_u24 x16", 16, out);
  run<25>("v_add_u32 x16", 16, out);
  run<26>("v_cndmask_b32 x16", 16, out);
  run<27>("v_add3_u32 x16", 16, out);
  run<28>("v_mov_b32 x16", 16, out);
  run<29>("v_or_b32 x16", 16, out);
  run<30>("v_lshlrev_b32 x16", 16, out);
  run<31>("v_max_f32 x16", 16, out);
  run<32>("v_sub_f32 x16", 16, out);
  run<33>("v_cndmask_b32_e64 (SGPR-pair mask) x16", 16, out);
  run<34>("v_bfi_b32 x16", 16, out);
  run<35>("v_cmp_gt_f32 -> vcc x16", 16, out);
  run<36>("s_mov vcc + v_cndmask_b32 x16 (2 instr each)", 32, out);
  run<37>("v_cndmask_b32 (vcc), distinct registers x16", 16, out);
  run<38>("v_cmp + v_cndmask pairs x16 (2 instr each)", 32, out);
  run<42>("v_cmp, 1 add, v_cndmask (3 instr per item) x16", 48, out);
  run<43>("v_cmp, 4 adds, v_cndmask (6 per item) x16", 96, out);
  run<44>("v_cmp, 8 adds, v_cndmask (10 per item) x16", 160, out);
  run<45>("v_cmp, 16 adds, v_cndmask (18 per item) x16", 288, out);
  run<46>("v_cmp_e64 -> SGPR pair, v_cndmask_e64 (2 per item) x16", 32, out);
  run<47>("v_cmp, 4 v_cndmask on its vcc (5 per item) x16", 80, out);
  run<40>("v_mad_u64_u32 x8", 8, out);
  run<41>("v_lshl_add_u64 x8", 8, out);
  return 0;
}
