// gain_f32g.h — K1 in EXACT f32 on the matrix cores for gain curves on the tile grid (round 5).
//
// The exact-f32 gain kernel of rounds 1-4 (gain_mfma.h) takes any curves: it folds libear's ramp into the INPUT side —
// two slots per ramping object, c(s) x(s) with c = 1 - p and c = p (gain_interpolator.hpp:272-274) — which costs a fused
// multiply-add and a multiply per sample, slot and wave on the VALU, and on this chip VALU and MFMA issue add up: 50.3 M
// MFMAs x 32 cycles are 72 % of its SIMD cycles, its 125 M VALU instructions most of the rest (NOTES.md, round 5).  When
// every curve point of the call lies on the 512-sample tile grid (block-aligned metadata, static gains: what gain_h2.h
// takes with split operands), a tile sees ONE segment per object, and libear's gain is a line in the sample index:
//
//     g(s) = B0 + (s - c) B1,    B0 = (1 - p_c) S + p_c E  (libear's own value at the tile's centre sample c),
//                                B1 = scale (E - S)         (slope per sample; 0 on constant segments)
//
// so  bus = sum_m x B0 + (s - c) sum_m x B1: two plain products on v_mfma_f32_16x16x4_f32 whose LEFT operand is the input
// sample itself — no operand split, no prescale, no level probe, no per-sample arithmetic in front of the matrix pipe —
// and the ramp factor once per tile on the two sets of accumulators.  Per 32 objects, 64 samples and 48 columns a wave
// issues 192 MFMAs (6144 cycles of the pipe) and about twenty other instructions: the kernel runs at the rate of the
// fp32 matrix pipe, 2 MACs per object, column and sample.
//
// One workgroup (8 waves, wave w = samples [64 w, 64 w + 64)) per 512-sample tile; a chunk = 32 objects between two
// barriers: the waves turn four objects each into the B fragments of the NEXT chunk (descriptor -> the two gain rows ->
// B0 / B1 in LDS, fragment layout of the MFMA's B operand) while the matrix pipe works on this one.  Descriptors as
// k_seg_prep leaves them (gain_kernels.h: one per object and tile).  Chosen by the launch plan for the exact-f32 setting
// (option MFMA = 1) when tiles_aligned(512) holds strictly and the call is whole tiles; everything else stays with
// gain_mfma.h.  Results differ from gain_mfma.h's by rounding (f32 products of the same operands, ramp applied per tile).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "gain_kernels.h"

namespace earhip {

typedef float f32g_x4 __attribute__((ext_vector_type(4)));

constexpr int kF32GridTile = 512;

// grid = (tiles, grid-level object splits, column super-groups), block = 512
template <int NCT>
__global__ void __launch_bounds__(512, 1) k_gain_mix_f32g(GainMixParams P, int zero_row) {
  constexpr int NW = 8, NRT = 4, TS = 16 * NRT, T = TS * NW, CH = 32, NQ = CH / NW;
  static_assert(T == kF32GridTile, "one workgroup per 512-sample tile");
  // B fragments of a chunk: [j = k-step][lane][2 ct + {B0, B1}] (+ 2 floats of padding: 32-byte entries, two b128 reads)
  __shared__ __attribute__((aligned(16))) float bfrag[2][8][2][64][4];  // (16-byte entries, lanes contiguous: no bank conflicts)
  __shared__ int rampflag[2][NW];  // per chunk buffer and wave: some object of the wave's share ramps (else the slope products are skipped)
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kg = lane >> 4;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int nparts = gridDim.y, part = blockIdx.y;
  const int col0 = blockIdx.z * 16 * NCT;
  const int wave_s0 = w * TS;
  const int tile_s0 = tile * T + wave_s0;  // first sample of this wave inside the call (whole tiles: the plan checks)
  const float *__restrict__ gain = P.ps.gain;
  const unsigned rowlen = (unsigned)P.ps.row;
  const SegDesc *__restrict__ desc = P.desc + (size_t)tile * P.M;

  f32g_x4 tot0[NRT][NCT], tot1[NRT][NCT];
#pragma unroll
  for (int r = 0; r < NRT; r++)
#pragma unroll
    for (int c = 0; c < NCT; c++) tot0[r][c] = tot1[r][c] = f32g_x4{0.0f, 0.0f, 0.0f, 0.0f};

  const int nchunks = (P.M + CH - 1) / CH;
  const int c_lo = (int)(((int64_t)nchunks * part) / nparts), c_hi = (int)(((int64_t)nchunks * (part + 1)) / nparts);

  // ---- the inputs of chunk c: object kg * 8 + q of the chunk, samples 4 li .. 4 li + 3 of this wave (objects past M: zeros,
  // so that nothing of theirs meets a gain — not even a NaN of the row that stands in)
  // (objects past M read sixteen bytes of the all-zero gain row instead: the mask is on the ADDRESS — a select on the loaded
  // value made the compiler wait for every request where it was issued: a trip to memory in the open per object group)
  const float *xbase = P.in + tile_s0 + 4 * li;
  const float *xzero = gain + (size_t)zero_row * rowlen + 4 * (li & 3);
  // (VEC: the rows are 16-byte aligned — decided once per kernel, outside the chunk loop: no branch around a request)
  auto load_x1 = [&](auto vec_tag, int c, int q) -> f32g_x4 {
    const int m = c * CH + kg * 8 + q;
    const float *px = m < P.M ? xbase + (size_t)m * P.in_stride : xzero;
    if constexpr (decltype(vec_tag)::value) return __builtin_nontemporal_load(reinterpret_cast<const f32g_x4 *>(px));
    else return f32g_x4{px[0], px[1], px[2], px[3]};
  };
  // ---- this wave's share of a chunk's B operand: object NQ w + (lane >> 4) of the chunk, column lane & 15 of every column tile
  struct Rows {
    float S[NCT], E[NCT];
    float p0, scale;
  };
  auto load_desc = [&](int c) -> SegDesc {
    const int m = c * CH + NQ * w + kg;
    SegDesc d;
    d.row = zero_row, d.d0 = 0, d.scale = 0.0f, d.info = 0;
    if (m < P.M && c < c_hi) d = desc[m];
    return d;
  };
  auto load_rows = [&](const SegDesc d, Rows &R) {
    const bool ramp = d.info & kSegRamp;
    // libear's ramp position (gain_interpolator.hpp:272) at the tile's centre sample; constant segments: 0
    R.p0 = ramp ? (float)(d.d0 + T / 2) * d.scale : 0.0f;
    R.scale = ramp ? d.scale : 0.0f;
    const float *gs = gain + (size_t)(unsigned)d.row * rowlen + col0 + li;
    const float *ge = gs + (ramp ? rowlen : 0u);
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {
      R.S[ct] = gs[16 * ct];
      R.E[ct] = ge[16 * ct];
    }
  };
  auto store_b = [&](const Rows &R, int buf) {
    const int q32 = NQ * w + kg;  // the object's place in the chunk: MFMA k-step q32 & 7, k index q32 >> 3
    float *f = &bfrag[buf][q32 & 7][0][(q32 >> 3) * 16 + li][0];
    float v[8];
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {
      // (constant segments: the row as it is — libear multiplies by it, gain_interpolator.hpp:293-296)
      v[2 * ct] = R.scale != 0.0f ? __builtin_fmaf(1.0f - R.p0, R.S[ct], R.p0 * R.E[ct]) : R.S[ct];  // (gain_h2.h's B0)
      v[2 * ct + 1] = R.scale != 0.0f ? R.scale * (R.E[ct] - R.S[ct]) : 0.0f;
    }
#pragma unroll
    for (int i = 2 * NCT; i < 8; i++) v[i] = 0.0f;
    *reinterpret_cast<f32g_x4 *>(f) = f32g_x4{v[0], v[1], v[2], v[3]};
    if (NCT > 2) *reinterpret_cast<f32g_x4 *>(f + 64 * 4) = f32g_x4{v[4], v[5], v[6], v[7]};
    const bool any = __ballot(R.scale != 0.0f) != 0ull;  // (wave-uniform)
    if (lane == 0) rampflag[buf][w] = any ? 1 : 0;
  };

  auto run = [&](auto vec_tag) __attribute__((always_inline)) {
    // ONE set of input registers: the inputs of object group j of the next chunk are requested into X[j] right behind the
    // MFMAs that read it — a chunk of 192 MFMAs is 3 us of the pipe, a trip to memory well under that
    f32g_x4 X[8];
    SegDesc dn;  // descriptor of this wave's object of the chunk after next
    {
      Rows R;
      load_rows(load_desc(c_lo), R);
      dn = load_desc(c_lo + 1);
#pragma unroll
      for (int q = 0; q < 8; q++) X[q] = load_x1(vec_tag, c_lo, q);
      store_b(R, c_lo & 1);
    }
    // chunk c: inputs in X, B fragments in bfrag[c & 1]; requests the rows and inputs of chunk c + 1 and the descriptor of
    // chunk c + 2, converts chunk c + 1 behind its MFMAs
#pragma unroll 1
    for (int c = c_lo; c < c_hi; c++) {
      __syncthreads();  // the fragments of chunk c are in place, those of chunk c - 1 no longer read
      Rows R;
      load_rows(dn, R);
      dn = load_desc(c + 2);
      const int buf = c & 1;
      // static gains, held gains: a chunk without a single ramp has no slope — half the MFMAs (workgroup-uniform)
      int anyramp = 0;
#pragma unroll
      for (int i = 0; i < NW; i++) anyramp |= rampflag[buf][i];
      const bool slopes = __builtin_amdgcn_readfirstlane(anyramp) != 0;
      const int cn = min(c + 1, c_hi - 1);  // (the last chunk requests its own inputs again: an unconditional request keeps
                                            // the compiler's count of what is in flight exact — behind a branch it waited for everything)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const f32g_x4 ba = *reinterpret_cast<const f32g_x4 *>(&bfrag[buf][j][0][lane][0]);
        f32g_x4 bb = f32g_x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (NCT > 2) bb = *reinterpret_cast<const f32g_x4 *>(&bfrag[buf][j][1][lane][0]);
        const float b[8] = {ba[0], ba[1], ba[2], ba[3], bb[0], bb[1], bb[2], bb[3]};
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) {
#pragma unroll
          for (int r = 0; r < NRT; r++) tot0[r][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(X[j][r], b[2 * ct], tot0[r][ct], 0, 0, 0);
          if (slopes) {
#pragma unroll
            for (int r = 0; r < NRT; r++) tot1[r][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(X[j][r], b[2 * ct + 1], tot1[r][ct], 0, 0, 0);
          }
        }
        X[j] = load_x1(vec_tag, cn, j);
      }
      store_b(R, buf ^ 1);
    }
  };
  if (c_hi > c_lo) {
    if (P.vec_ok) run(std::true_type{});
    else run(std::false_type{});
  }

  // ---- bus = tot0 + (s - centre) tot1.  D fragment of row tile r: rows 4 kg + e = samples 16 kg + 4 e + r of the wave: for a
  // fixed e the four row tiles are four consecutive samples
  float *op = P.out + (size_t)blockIdx.y * P.part_stride + tile_s0;
#pragma unroll
  for (int c = 0; c < NCT; c++) {
    const int col = col0 + c * 16 + li;
    if (col >= P.ncols) continue;
    float *o = op + (size_t)col * P.out_stride;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int s = kg * 16 + e * 4;
      f32g_x4 v;
#pragma unroll
      for (int r = 0; r < NRT; r++) v[r] = __builtin_fmaf((float)(wave_s0 + s + r - T / 2), tot1[r][c][e], tot0[r][c][e]);
      if (P.vec_ok) {
        *reinterpret_cast<f32g_x4 *>(o + s) = v;
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) o[s + i] = v[i];
      }
    }
  }
}

}  // namespace earhip
