// gain_h2.h — K1 on the f16 matrix cores with every fp32 operand split into TWO f16
// pieces after an exact power-of-two prescale ("f16x2").  The bus-forming contraction of
// gain_mfma.h,
//
//     bus[col][s] = sum_m x_m(s) * g_m,col(s),      g linear in s on a curve segment,
//
// with the ramp moved to the gain side.  For a workgroup tile starting at sample s0
//
//     g(s) = B0 + (s - s0) * B1,   B0 = (1-p0)*S + p0*E  (libear's gain AT s0,
//                                  gain_interpolator.hpp:272-274),
//                                  B1 = scale * (E - S)  (slope per sample)
//
// so  bus = sum_m x*B0 + (s - s0) * sum_m x*B1 : two plain products with the SAME left
// operand x, and (s - s0) is applied to the accumulator rows.  x is split once per sample
// and object; B0/B1 are split once per workgroup tile and object and shared by the
// workgroup's waves through LDS.  The f32-in MFMA runs at the vector rate (157 TFLOP/s:
// 0.70 ms for the headline scene, above its HBM floor of 0.33 ms); the f16 MFMA at 16x that:
//
//   * v * 2^k = h + l with f16 pieces (11 + 11 significand bits, RNE, residual exact in
//     fp32) is good to 2^-22 relative as long as l is a normal f16, i.e. over 2^16 / 2^-3
//     = 19 binades below the top of the f16 range; below that the error is 2^-25 ABSOLUTE
//     in scaled units.  The prescale puts the operands at the top of the range: the gains
//     by 2^14 / (largest |gain| of the curve set, rounded up to a power of two; known at
//     commit), the inputs of a call by 2^7 / (the level K0 probes in them; EARHIP_XSCALE: a fixed
//     scale);
//   * of the four partial products the three of order >= 2^-11 are kept: xl*bh, xh*bl,
//     xh*bh (relative error of the contraction on the headline scene: 7e-8, see
//     tests/test_gpu_render.py and DESIGN.md section 4 / NOTES.md);
//   * products of f16 pairs are exact in the fp32 accumulate of
//     v_mfma_f32_16x16x32_f16, so the chunks accumulate straight into two running sets of
//     accumulators (B0 terms and B1 terms: no per-chunk fold), in scaled units; the ramp
//     factor (s - s0) and the inverse scale are applied once per tile;
//   * an input outside the f16 range after the prescale (|x| * scale >= 65520, or a
//     non-finite sample) turns the tile's accumulators non-finite: the wave then
//     recomputes its tile with the exact f32 MFMA path, unscaled.  Correct for any input,
//     fast for audio.
//
// Fragment layout of v_mfma_f32_16x16x32_f16 (8 f16 = 4 VGPRs per operand):
//   A: lane l holds row l&15, k = 8*(l>>4) .. +7     B: column l&15, same k
//   D: lane l holds column l&15, rows 4*(l>>4) + e, e = 0..3
// k = the 32 objects of a chunk; row i of row tile r = sample 4*i + r of the wave's
// 64-sample tile (a lane's 4 inputs of one object are one 16-byte load and the D fragments
// of the 4 row tiles interleave into float4 stores); column j of column tile c = bus column
// col0 + 16*c + j.
//
// Objects with a curve point inside the workgroup tile, unaligned buffers and partial tiles
// use the exact f32 MFMA on the same accumulators (slow path, same arithmetic as
// gain_mfma.h).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "gain_kernels.h"
#include "gain_mfma.h"

namespace earhip {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kSplitTile = 256;   // samples per workgroup tile of the 4-wave kernel (descriptor tile)
constexpr int kSplitChunk = 32;   // objects per MFMA (k)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

// two floats -> packed f16 pair, round to nearest even (v_cvt_pk_f16_f32)
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ float f16_lo(uint32_t u) { return (float)__builtin_bit_cast(f16x2_t, u)[0]; }
__device__ __forceinline__ float f16_hi(uint32_t u) { return (float)__builtin_bit_cast(f16x2_t, u)[1]; }
// v - (the low / high half of u as a float): the exact residual of a split in ONE instruction (v_fma_mix_f32 converts its
// f16 operand on the way in; the conversion alone, v_cvt_f32_f16, issues at the same 4.3 cycles, and the subtraction came on
// top — profiles/r05_valu_issue_rates.txt)
// (written out: the compiler turns fma(half, -1, v) back into a conversion and a subtraction)
__device__ __forceinline__ float sub_f16_lo(float v, uint32_t u) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(u), "v"(v));
  return r;
}
__device__ __forceinline__ float sub_f16_hi(float v, uint32_t u) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(u), "v"(v));
  return r;
}

// Wide mode: the low piece of an input is kept as (residual x 2^11) and multiplied with (h x 2^-11) of the gain —
// both exact scalings, the same product — so that it is a normal f16 over 21 binades below the level the prescale
// aims at instead of 11 (measured relative RMS of the products: 6.5e-8 down to 2^-18, 1.4e-7 at 2^-20, 5e-7 at
// 2^-22; with the plain residual 2.5e-7 at 2^-10 and 1e-6 at 2^-12).
constexpr float kLowPieceScale = 2048.0f;
__device__ __forceinline__ uint32_t scale_f16x2_down(uint32_t h) {  // both halves x 2^-11 (exact unless subnormal)
  const f16x2_t k = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2_t, h) * k);
}

__device__ __forceinline__ f32x4 mfma_f16(const u32x4 &a, const u32x4 &b, const f32x4 &c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c,
                                                0, 0, 0);
}

// P.ntiles / P.desc refer to WORKGROUP tiles of kSplitTile samples.  x_scale: an exact power of two (see
// above); gcol: [row] per-COLUMN gain scales, powers of two that put the largest gain a column ever gets at 2^14
// (CurveSet::commit: a loudspeaker that only gets small gains — an object 120 dB down alone on it — keeps both
// f16 pieces of its gains normal; the inverse is applied to the column's output); zero_row: index of an all-zero
// gain row.
// WIDE: the low pieces of the inputs are kept scaled (kLowPieceScale) — 16 more multiplies and 6 more LDS reads per
// chunk: 4 % of this kernel on the headline scene (measured by bisection; neither reading the third piece a block
// ahead, nor making it from h in registers, nor v_fma_mix forms of the split bring that down).  So both forms
// exist and the level probe decides on the device which one works (k_seg_prep sets *wide_cur when some object lies
// more than kPlainBinades below the loudest): the 8-wave kernel carries both bodies and branches on the word (FORM 2,
// round 6), the 4-wave kernel is two instantiations launched back to back, of which the one not named returns at once
// (an empty grid: ~3-5 us).  Without a probe (wide_cur == NULL) only the wide form runs.  The mode words alternate
// between calls like the level words.
// NW waves per workgroup (4 or 8), each on 64 samples of the workgroup's tile of 64 NW samples: the
// conversion of a chunk's gains is shared by the whole workgroup, so the 512-sample tile halves that
// work (and the gain rows' L2 traffic) when no curve point falls inside 512-sample tiles.
// (the 4-wave forms with two or three column tiles: gain_h2_t1.h, a tile per workgroup)
constexpr bool h2_persistent(int nct, int nw) { return nw == 8 || nct == 1; }
// FORM: 0 the plain form, 1 the wide form (each returns at once when the probe's word names the other: the 4-wave kernel is launched
// as such a pair), 2: both forms in ONE kernel, the word picking with a wave-uniform branch (the 8-wave kernel: one workgroup per CU
// at either form's register count, and no launch whose only job is to return — ~5 us per call)
template <int NCT, int NW, int FORM>
__global__ void __launch_bounds__(64 * NW, NCT == 1 && NW == 8 ? 2 : (NW == 4 ? 2 : 1))
k_gain_mix_h2(GainMixParams P, int zero_row, float x_scale, const float *__restrict__ gcol, const unsigned *level_cur,
              unsigned *level_next, const unsigned *slow_cur, unsigned *slow_next, const unsigned *wide_cur, unsigned *wide_next) {
  if (FORM != 2 && wide_cur && ((*wide_cur & 1u) != 0u) != (FORM == 1)) return;  // the other form of this kernel works on this call
  constexpr int NRT = 4, TS = 16 * NRT, CH = kSplitChunk;
  constexpr int NQ = CH / NW;         // objects whose gains one wave converts per chunk
  constexpr int NFRAG = 2 * NCT * 3;  // {B0,B1} x column tiles x {h, l, h 2^-11 (wide mode)}
  __shared__ u32x4 bfrag[2][NFRAG + 6][64];  // + 6 never-read fragments: the lanes without a column write there
  // the wave's output tile of one column tile, [16 columns][64 samples (+ 4: bank spread)], on its way from the
  // D fragments (a lane: one column, 16-byte pieces 64 bytes apart) to stores of whole 256-byte rows
  constexpr int OP = TS + 4;
  __shared__ __attribute__((aligned(16))) float otile[NW][16 * OP];
  __shared__ float inv_gcol[16 * NCT];  // 1 / the gain scale of the workgroup's columns: read at the END of a tile, where a
                                        // round trip to memory would stand in the open
  // the workgroup's tiles that a wave has to redo exactly (bit k: its k-th tile; at most 64: the host's grid), kept in
  // LDS: set once in a blue moon, read once
  __shared__ unsigned long long redo_tiles[NW];
  auto body = [&](auto wide_tag) __attribute__((always_inline)) {
  constexpr bool WIDE = decltype(wide_tag)::value;
  if (threadIdx.x < 16 * NCT) inv_gcol[threadIdx.x] = 1.0f / gcol[blockIdx.z * 16 * NCT + threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kg = lane >> 4;
  // A workgroup works through the tiles vb = blockIdx.x, + gridDim.x, ... (< P.ntiles; the host launches as many
  // workgroups as are resident at once): ONE software pipeline runs through all of them, so the requests of a
  // tile's first chunks are in flight while the tile before is finished and stored.  (A workgroup per tile spent
  // 8-11 us per tile outside its chunk loop — filling the pipeline, draining it, storing — with nothing else on
  // the CU to cover it: 8 % of the kernel at 32 chunks per tile, a quarter at 4.)  vb % 8 = the XCD for every
  // tile of a workgroup when gridDim.x is a multiple of 8.
  const int ntl = P.ntiles, gx = gridDim.x;
  // P.tile_runs: a workgroup takes a contiguous RUN of tiles (the balanced partition of [0, ntl) over the workgroups) instead
  // of every gx-th one: its 64 row pieces of consecutive tiles are neighbours in memory (option H2_RUNS; config 2's shape)
  const bool runs = P.tile_runs != 0;
  const int run_first = (int)(((int64_t)ntl * blockIdx.x) / gx), run_end = (int)(((int64_t)ntl * (blockIdx.x + 1)) / gx);
  const int vb0 = runs ? run_first : (int)blockIdx.x, vstep = runs ? 1 : gx;
  const int vb_last = runs ? max(run_end - 1, run_first) : ntl - 1;
  const int my_tiles = runs ? run_end - run_first : (ntl - (int)blockIdx.x + gx - 1) / gx;
  auto tile_at = [&](int vb) { return runs ? min(vb, ntl - 1) : xcd_tile(vb, ntl); };
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    if (wide_next) *wide_next = 0u;
    record_mode(P, wide_cur != nullptr);
  }
  if (level_cur) {
    // input scale of THIS call from the level K0 probed: the largest magnitude seen, in [2^E, 2^(E+1)),
    // goes to [2^7, 2^8) — peaks up to 256x the probed maximum stay inside the f16 range (beyond:
    // exact fallback below), and samples down to 2^-11 of it keep a normal low piece (2^-22 relative;
    // below that 2^-33 of the maximum, absolute).  Nothing seen (silence), absurd or non-finite
    // levels: clamped.
    const unsigned lv = *level_cur;
    if (lv) {
      const int E = max(-60, min(20, (int)(lv >> 23) - 127));
      x_scale = __uint_as_float((unsigned)(127 + 7 - E) << 23);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *level_next = 0;
  }
  const int nparts = gridDim.y;
  const int part = blockIdx.y;
  const int m_lo = (int)(((int64_t)P.M * part) / nparts);
  const int m_hi = (int)(((int64_t)P.M * (part + 1)) / nparts);
  const int col0 = blockIdx.z * 16 * NCT;
  const float *__restrict__ gain = P.ps.gain;
  const unsigned rowlen = (unsigned)P.ps.row;
  // (per-tile values are made from the tile index where they are needed: wave-uniform, a few scalar instructions,
  // instead of registers held through the chunk loop)
  auto tile_first = [&](int t) { return t * (TS * NW) + w * TS; };  // first sample of this wave's part of tile t
  auto tile_length = [&](int t) { return max(0, min(TS, P.nsamples - tile_first(t))); };
  // the rare paths below make their lane numbers anew (from a value the compiler cannot see through) so that nothing
  // of theirs is kept in registers through the chunk loop
  auto opaque_lane = [&]() __attribute__((always_inline)) {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };

  // running totals in scaled units: bus = (tot0 + (s - s0) tot1) / (x_scale g_scale)
  f32x4 tot0[NRT][NCT], tot1[NRT][NCT];
  auto clear_totals = [&]() {
#pragma unroll
    for (int r = 0; r < NRT; r++)
#pragma unroll
      for (int c = 0; c < NCT; c++) tot0[r][c] = tot1[r][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  };
  clear_totals();

  // (objects the level probe found far below the call's level — kSegQuiet, set by k_seg_prep
  // — are treated like objects with a curve point inside the tile: zero row in the main loop,
  // this path afterwards)
  // ---- slow path: one object, all its pieces inside this wave's tile, exact f32 MFMA
  // with k = {a, b} of ONE object (k slots 2, 3 idle), accumulated into tot0 in units of 1 / (sx sg)
  // (the two scales are applied to the two operands: their product may not be a float)
  // khint >= 0: the segment index K0 found at the start of the WORKGROUP tile (the search then only
  // walks on from there: a few steps instead of log2 n dependent loads per object and wave)
  // (sg: the gains are scaled by their column's scale, like the split operands — or not at all)
  auto single_object = [&](int t, int m, float sx, bool sg, int khint = -1) __attribute__((always_inline)) {
    const int tile_s0 = tile_first(t), tile_len = tile_length(t);
    if (tile_len <= 0) return;
    const int64_t tile_t0 = P.t_call + tile_s0, tile_t1 = tile_t0 + tile_len;
    const int ol = opaque_lane(), li = ol & 15, kg = ol >> 4;
    float gsc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; c++) gsc[c] = sg ? gcol[col0 + c * 16 + li] : 1.0f;
    const int base = P.ps.off[m], n = P.ps.cnt[m];
    const float *row = P.in + (size_t)m * P.in_stride + tile_s0;
    const bool is_b = kg & 1;
    const bool slot0 = kg < 2;
    int k;
    if (khint >= 0) {
      k = min(khint, n);
      while (k < n && P.ps.time[base + k] <= tile_t0) k++;
    } else {
      k = upper_bound_time(P.ps.time + base, n, tile_t0);
    }
    int cur = 0;
    while (cur < tile_len) {
      const SegDesc dk = describe_segment(P.ps, base, n, k, tile_t0, tile_t1);
      const int r1 = min(seg_r1(dk.info), tile_len);
      if (r1 > cur) {  // duplicate times make empty segments (steps)
        const bool ramp = dk.info & kSegRamp;
        float a[NRT], gv[NCT];
#pragma unroll
        for (int r = 0; r < NRT; r++) {
          const int s = li * NRT + r;
          const float x = row[min(s, tile_len - 1)];
          const float p = (float)(dk.d0 + s) * dk.scale;  // gain_interpolator.hpp:272
          float coef = ramp ? (is_b ? p : 1.0f - p) : (is_b ? 0.0f : 1.0f);
          coef = (slot0 && s >= cur && s < r1) ? coef : 0.0f;
          a[r] = (x * coef) * sx;
        }
        const int grow = dk.row + ((ramp && is_b && slot0) ? 1 : 0);
        const float *gp = gain + (size_t)grow * rowlen + col0 + li;
#pragma unroll
        for (int c = 0; c < NCT; c++) gv[c] = gp[c * 16] * gsc[c];
#pragma unroll
        for (int r = 0; r < NRT; r++)
#pragma unroll
          for (int c = 0; c < NCT; c++)
            tot0[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], gv[c], tot0[r][c], 0, 0, 0);
        cur = r1;
      }
      if (!(dk.info & kSegMulti)) break;
      k++;
    }
  };

  const int nobj = m_hi - m_lo;
  const int nch = (P.vec_ok && nobj >= CH) ? (nobj + CH - 1) / CH : 0;
  // the tile is done: scale, apply the ramp factor, store (and start the next tile's totals from zero).  inv_x: the
  // inverse of the input scale the totals carry (exact: a power of two); col_scaled: they are in units of 1 / (the
  // column's gain scale) as well
  auto flush_tile = [&](int t, float inv_x, bool col_scaled) __attribute__((always_inline)) {
    const int tile_s0 = tile_first(t), tile_len = tile_length(t);
    if (tile_len > 0) {
      const int ol = opaque_lane(), li = ol & 15, kg = ol >> 4;
      // D fragment of row tile r: rows 4kg + e = samples 16kg + 4e + r: for fixed e the
      // four row tiles are 4 consecutive samples.  (s - s0) of the rows: sample 64w + 16kg + 4e + r.
      const float wf0 = (float)(w * TS + kg * 16 - (TS * NW) / 2);  // (s - c): the line is anchored at the tile's centre
      float inv_gc[NCT];  // inverse gain scale of the lane's column in each column tile (the D fragments' layout)
#pragma unroll
      for (int c = 0; c < NCT; c++) inv_gc[c] = col_scaled ? inv_gcol[c * 16 + li] : 1.0f;
      float *op = P.out + (size_t)blockIdx.y * P.part_stride + tile_s0;
      const bool whole = P.vec_ok && tile_len == TS;  // (wave-uniform)
#pragma unroll
      for (int c = 0; c < NCT; c++) {
        if (whole) {
          // Through wave-private LDS: written straight from the fragments, one store instruction covers 4 columns x
          // 64 bytes in 16-byte pieces (64 scattered pieces per instruction: the stores of the 100 MB of buses cost a
          // tenth of this kernel's time, a quarter at 256 objects); transposed, it covers 4 whole 256-byte rows.  The
          // rows are written past the caches (K2 reads them once, much later: headline K1 0.437 -> 0.412 ms).
          float *ot = otile[w];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < NRT; r++)
              v[r] = (__builtin_fmaf(wf0 + (float)(4 * e + r), tot1[r][c][e], tot0[r][c][e]) * inv_x) * inv_gc[c];
            *reinterpret_cast<f32x4 *>(ot + li * OP + kg * 16 + e * 4) = v;
          }
#pragma unroll
          for (int j = 0; j < 4; j++) {  // lane: column 4 j + (lane >> 4), samples 4 (lane & 15) .. + 3
            const int cl = 4 * j + kg, col = col0 + c * 16 + cl;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(ot + cl * OP + li * 4);
            if (col < P.ncols) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(op + (size_t)col * P.out_stride + li * 4));
          }
          continue;
        }
        const int col = col0 + c * 16 + li;
        if (col >= P.ncols) continue;
        float *o = op + (size_t)col * P.out_stride;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int s = kg * 16 + e * 4;
          f32x4 v;
#pragma unroll
          for (int r = 0; r < NRT; r++)
            v[r] = (__builtin_fmaf(wf0 + (float)(4 * e + r), tot1[r][c][e], tot0[r][c][e]) * inv_x) * inv_gc[c];
          if (P.vec_ok && s + 3 < tile_len) {
            *reinterpret_cast<f32x4 *>(o + s) = v;
          } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
              if (s + i < tile_len) o[s + i] = v[i];
          }
        }
      }

    }
    clear_totals();
  };

  if (lane == 0) redo_tiles[w] = 0ull;
  if (nch > 0) {
    // lane-constant part of the input address: byte offset of this lane's float4 (lanes past the
    // end of the call re-read the last vector; never stored).  All loads below are (wave-uniform
    // 64-bit base) + (32-bit lane offset).  The last chunk is moved back to end at m_hi (its objects
    // that the previous chunk already covered get the zero row).
    // (the tile enters as a wave-uniform offset: the requests of a chunk may belong to the tile after the one
    // whose totals are being accumulated)
    const int nvec = (P.nsamples + 3) & ~3;
    const unsigned xrow = (unsigned)(kg * 8) * (unsigned)P.in_stride * 4u;
    const int xsl = w * TS + li * NRT;  // the lane's first sample within a workgroup tile
    // where the pipeline stands: stage A = the chunk whose MFMAs run, B = the next one (its gains are requested
    // and converted during A), C = the one after (its descriptors and inputs are requested during A): tile index
    // and chunk within the tile of each; `left` = chunks this workgroup still has to do, A included (B is live
    // while left > 1, C while left > 2; stages past the end stay on the last tile: their requests are never used)
    struct Stage {
      int tile, c;
    };
    int left = my_tiles * nch;
    int vb_c = vb0;  // stage C's tile counter
    auto next_stage = [&](const Stage &s) __attribute__((always_inline)) {
      Stage n;
      const bool wrap = s.c + 1 == nch;
      n.c = wrap ? 0 : s.c + 1;
      if (wrap) vb_c = min(vb_c + vstep, vb_last);
      n.tile = wrap ? tile_at(vb_c) : s.tile;
      return n;
    };
    const unsigned bcol_e = (unsigned)(col0 + min(lane, 16 * NCT - 1));  // the lane's gain column
    const float g_scale = gcol[bcol_e];                                   // ... and that column's scale (a power of two)
    // fragment this lane fills: B0 pieces at bfr, bfr+1, B1 pieces NCT*2 further
    const int blane = (w * NQ / 8) * 16 + (lane & 15);
    const int bfr = lane < 16 * NCT ? (lane >> 4) * 3 : NFRAG;
    const int bfr1 = lane < 16 * NCT ? NCT * 3 : 3;
    auto chunk_base = [&](int c) { return min(m_lo + c * CH, m_hi - CH); };

    // inputs q0 .. q0 + n - 1 of chunk c.  Past the last chunk (the two-chunks-ahead requests of the
    // last two chunks: never used) every lane of every request reads the same 16 bytes instead of 64
    // lines per wave that would occupy the CU's miss slots for nothing (6 % of the input requests).
    auto load_x_part = [&](const Stage &st, bool live, f32x4 (&x)[8], int q0, int n) {
      const int ts0 = st.tile * (TS * NW);
      const char *bp = reinterpret_cast<const char *>(P.in + (live ? (size_t)chunk_base(st.c) * P.in_stride + ts0 : 0));
      const size_t rstride = live ? P.in_stride * sizeof(float) : 0;
      const unsigned xo = live ? xrow + 4u * (unsigned)min(xsl, nvec - 4 - ts0) : 0u;
#pragma unroll
      for (int q = 0; q < 8; q++)
        if (q >= q0 && q < q0 + n)
          x[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(bp + (size_t)q * rstride + xo));
    };
    auto load_x = [&](const Stage &st, bool live, f32x4 (&x)[8]) { load_x_part(st, live, x, 0, 8); };
    // What this WAVE converts for chunk c: objects chunk_base + NQ w + q.  Their descriptors are
    // wave-uniform: scalar loads (requested one chunk ahead), scalar row arithmetic, and the
    // gain rows come in as (scalar row pointer) + (the lane's column).
    // (read through the constant address space: the compiler cannot prove that nothing in the kernel
    // writes the descriptors — K0 wrote them, this kernel only reads — and would otherwise use vector
    // loads and vector row arithmetic for these uniform values)
    typedef const SegDesc __attribute__((address_space(4))) *ConstDesc;
    struct ChunkDesc {
      SegDesc d[NQ];
    };
    auto load_desc = [&](const Stage &st) {
      ConstDesc dp = (ConstDesc)(P.desc + (size_t)st.tile * P.M + chunk_base(st.c) + w * NQ);
      ChunkDesc D;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        D.d[q].row = dp[q].row;
        D.d[q].d0 = dp[q].d0;
        D.d[q].scale = dp[q].scale;
        D.d[q].info = dp[q].info;
      }
      return D;
    };
    struct ChunkCoef {
      float p0[NQ], scale[NQ];
    };
    auto load_gains = [&](const ChunkDesc &R, int cc, ChunkCoef &D, float (&S)[NQ], float (&E)[NQ]) {
      const int m0 = chunk_base(cc) + w * NQ;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const SegDesc d = R.d[q];
        // objects the previous chunk already covered (last chunk moved back) and objects with
        // curve points inside the tile (slow path) get the all-zero row
        const bool valid = m0 + q >= m_lo + cc * CH && !(d.info & (kSegMulti | kSegQuiet));
        const unsigned rs = (unsigned)(valid ? d.row : zero_row);
        // (a select between two uniform values: "rs + (ramp ? 1 : 0)" turns the uniform condition into a lane value and
        // the row products into eight quarter-rate vector multiplies per chunk)
        const unsigned re = (valid && (d.info & kSegRamp)) ? (unsigned)d.row + 1u : rs;
        // gain_interpolator.hpp:272 at the workgroup tile's CENTRE sample (constant segments: scale = 0): the line is anchored
        // there — (s - c) runs over +-half a tile, not over a whole one: half the weight on the slope totals' rounding
        D.p0[q] = d.scale != 0.0f ? (float)(d.d0 + (TS * NW) / 2) * d.scale : 0.0f;
        D.scale[q] = d.scale;             // constant segments: scale = 0, d0 = 0
        const float *rps = gain + (size_t)rs * rowlen, *rpe = gain + (size_t)re * rowlen;
        S[q] = rps[bcol_e];
        E[q] = rpe[bcol_e];
      }
    };
    // B0 (gain at the tile start, part 0) or B1 (slope, part 1) of the wave's NQ objects,
    // scaled and split -> LDS (k = 8 kgw + 2 i + j of the fragment entry of lane 16 kgw + column)
    auto store_b = [&](const ChunkCoef &D, const float (&S)[NQ], const float (&E)[NQ], int buf, int part) {
      uint32_t h[NQ / 2], l[NQ / 2], hs[NQ / 2];
#pragma unroll
      for (int i = 0; i < NQ / 2; i++) {
        float v[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int q = 2 * i + j;
          v[j] = (part == 0 ? __builtin_fmaf(1.0f - D.p0[q], S[q], D.p0[q] * E[q]) : D.scale[q] * (E[q] - S[q])) *
                 g_scale;
        }
        const uint32_t H = pack_f16(v[0], v[1]);
        h[i] = H;
        l[i] = pack_f16(sub_f16_lo(v[0], H), sub_f16_hi(v[1], H));  // residuals: exact in fp32
        hs[i] = WIDE ? scale_f16x2_down(H) : 0u;                // h 2^-11: the partner of the inputs' scaled low piece
      }
      u32x4 *f = &bfrag[buf][bfr][blane] + (part ? bfr1 * 64 : 0);
      if constexpr (NQ == 8) {
        f[0] = u32x4{h[0], h[1], h[2], h[3]};
        f[64] = u32x4{l[0], l[1], l[2], l[3]};
        if constexpr (WIDE) f[128] = u32x4{hs[0], hs[1], hs[2], hs[3]};
      } else {  // half an entry: words 2 (w & 1), + 1
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        u32x2 *g = reinterpret_cast<u32x2 *>(f) + (w & 1);
        g[0] = u32x2{h[0], h[1]};
        g[128] = u32x2{l[0], l[1]};
        if constexpr (WIDE) g[256] = u32x2{hs[0], hs[1]};
      }
    };

    // Inputs are requested TWO chunks ahead (a chunk is about 1 us of work for the wave, less
    // than the memory latency under load) into the registers the operand split has just
    // freed; the chunk loop is unrolled twice so that the two register sets need no moves.
    // The vector-memory counter is in order (a wait for one load waits for every older one),
    // so per chunk the gain rows of the next chunk are requested BEFORE the inputs and
    // converted while those are in flight.
    f32x4 X0[8], X1[8];
    ChunkDesc L;
    Stage sa, sb, sc;
    sa.tile = tile_at(vb_c);
    sa.c = 0;
    sb = next_stage(sa);
    sc = next_stage(sb);
    int ka = 0;  // stage A's tile is the workgroup's ka-th
    {
      float S[NQ], E[NQ];
      ChunkCoef D;
      L = load_desc(sa);
      load_gains(L, 0, D, S, E);
      load_x(sa, true, X0);
      load_x(sb, left > 1, X1);
      store_b(D, S, E, 0, 0);
      store_b(D, S, E, 0, 1);
      L = load_desc(sb);
    }
    // the chunk of stage A: inputs in xc, B fragments in bfrag[buf]
    auto chunk = [&](int buf, f32x4 (&xc)[8]) __attribute__((always_inline)) {
      __syncthreads();  // B fragments of this chunk are in bfrag[buf]; bfrag[buf^1] is free
      float S[NQ], E[NQ];
      ChunkCoef D;
      load_gains(L, sb.c, D, S, E);  // the next chunk (its descriptors were fetched one chunk ago)
      L = load_desc(sc);
      __builtin_amdgcn_sched_barrier(0);  // every gain row is requested before any input (see above)

      // A fragments: row tile r = sample 4*li + r of the 8 objects of this lane.  2 x 2
      // blocks: an f16 pair packs two OBJECTS (q, q+1) of one row tile, the scaling and the
      // exact residual subtractions pair two SAMPLES (r, r+1) of one object (neighbours in
      // the loaded float4: packed arithmetic without operand moves).
      u32x4 ah[NRT], al[NRT];
#pragma unroll
      for (int qp = 0; qp < 4; qp++)
#pragma unroll
        for (int rp = 0; rp < NRT; rp += 2) {
          const f32x2 s0 = f32x2{xc[2 * qp][rp], xc[2 * qp][rp + 1]} * x_scale;          // object 2qp
          const f32x2 s1 = f32x2{xc[2 * qp + 1][rp], xc[2 * qp + 1][rp + 1]} * x_scale;  // object 2qp+1
          const uint32_t H0 = pack_f16(s0[0], s1[0]), H1 = pack_f16(s0[1], s1[1]);
          // residuals (exact); wide mode scales them by 2^11 before they are rounded to f16
          constexpr float LOW = WIDE ? kLowPieceScale : 1.0f;
          const f32x2 r0 = f32x2{sub_f16_lo(s0[0], H0), sub_f16_lo(s0[1], H1)} * LOW;
          const f32x2 r1 = f32x2{sub_f16_hi(s1[0], H0), sub_f16_hi(s1[1], H1)} * LOW;
          ah[rp][qp] = H0;
          ah[rp + 1][qp] = H1;
          al[rp][qp] = pack_f16(r0[0], r1[0]);
          al[rp + 1][qp] = pack_f16(r0[1], r1[1]);
        }
      // The inputs of chunk c+2 go into the registers just freed, a few requests per MFMA block:
      // requested in one burst right here, by all the workgroup's waves at once, they queue up in the
      // CU's address unit and the waves sit in their load instructions instead of issuing MFMAs.
      constexpr int XB = 2 * NCT >= 4 ? 4 : 2 * NCT;  // blocks that carry input requests
      // 2*NCT blocks (column tile ct = blk >> 1, operand blk & 1: B0 / B1) of 12 MFMAs: three
      // partial products per operand pair, smallest first.  MFMA and VALU instructions do not
      // overlap except for one VALU instruction directly behind an MFMA
      // (tools/experiments/coissue2.hip), so the conversion of the next chunk's gains is
      // woven between the MFMAs of blocks 0 and 1.
      __builtin_amdgcn_sched_barrier(0);  // the splitting above stays above
      constexpr int NBLK = 2 * NCT;
      // (h, l) of a block are read one block ahead into alternating registers; wide mode reads the scaled high piece
      // as its block starts, behind the four MFMAs that do not need it (one register set: the kernel sits at the
      // register limit)
      u32x4 b[2][2], b2;
      auto load_b = [&](int blk, u32x4 (&bb)[2]) {
#pragma unroll
        for (int q = 0; q < 2; q++) bb[q] = bfrag[buf][((blk & 1) * NCT + (blk >> 1)) * 3 + q][lane];
      };
      load_b(0, b[0]);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // (block 0's reads; each block below places the NEXT block's)
#pragma unroll
      for (int blk = 0; blk < NBLK; blk++) {
        u32x4(&bc)[2] = b[blk & 1];
        const int ct = blk >> 1;
        if constexpr (WIDE) b2 = bfrag[buf][((blk & 1) * NCT + (blk >> 1)) * 3 + 2][lane];
        if (blk + 1 < NBLK) load_b(blk + 1, b[(blk + 1) & 1]);
        // (the two small products first, the large one last; wide mode: the one whose operand is read last second)
        f32x4(&tt)[NRT][NCT] = (blk & 1) == 0 ? tot0 : tot1;
        if constexpr (WIDE) {
#pragma unroll
          for (int r = 0; r < NRT; r++) tt[r][ct] = mfma_f16(ah[r], bc[1], tt[r][ct]);
#pragma unroll
          for (int r = 0; r < NRT; r++) tt[r][ct] = mfma_f16(al[r], b2, tt[r][ct]);
        } else {
#pragma unroll
          for (int r = 0; r < NRT; r++) tt[r][ct] = mfma_f16(al[r], bc[0], tt[r][ct]);
#pragma unroll
          for (int r = 0; r < NRT; r++) tt[r][ct] = mfma_f16(ah[r], bc[1], tt[r][ct]);
        }
#pragma unroll
        for (int r = 0; r < NRT; r++) tt[r][ct] = mfma_f16(ah[r], bc[0], tt[r][ct]);
        if (blk < XB) load_x_part(sc, left > 2, xc, blk * (8 / XB), 8 / XB);
        const bool conv0 = blk == NBLK - 2, conv1 = blk == NBLK - 1;
        if (conv0) store_b(D, S, E, buf ^ 1, 0);  // next chunk's gains (after the last chunk:
        if (conv1) store_b(D, S, E, buf ^ 1, 1);  // written, never read)
        // issue order: the LDS reads first, then every MFMA followed by VALU instructions
        if constexpr (WIDE) {
          if (blk + 1 < NBLK) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
          else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        } else {
          if (blk + 1 < NBLK) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        if (conv0 || conv1) {
#pragma unroll
          for (int k = 0; k < 12; k++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
          }
        } else if (blk < XB) {
#pragma unroll
          for (int k = 0; k < 8 / XB; k++) {  // MFMAs, then one request (address arithmetic + load)
            __builtin_amdgcn_sched_group_barrier(0x008, 12 / (8 / XB), 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        }
      }
    };
    // what follows a chunk: the end of its tile (exact-path objects, the check of the totals, the stores), and
    // the pipeline moves on
    auto tile_end = [&](int t) __attribute__((always_inline)) {
      {
        // does any object of this tile need the exact path (a curve point inside the tile, a quiet object)?  K0:
        // k_seg_prep leaves a word per tile; this call clears the words of the call after next (they alternate).
        // (a scalar load: a vector load here would have to wait for every input request in flight)
        typedef const unsigned __attribute__((address_space(4))) *ConstWord;
        const bool any_slow = !slow_cur || ((ConstWord)slow_cur)[t] != 0u;
        if (slow_next && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) slow_next[t] = 0u;
        // objects with curve points inside this workgroup tile (zero rows above)
        // (looking through the descriptors eight groups of 64 at a time — eight independent loads, then their ballots — instead of a
        // dependent trip to memory per group: built in round 6, 1-2 % on the scene with 8 of 1024 objects off the grid, and 1.7 %
        // SLOWER on the headline, whose chunk loop the extra scalar state re-allocated: not kept)
        for (int b0 = 0; any_slow && b0 < nobj; b0 += 64) {
          const int ol = opaque_lane();
          const SegDesc db = (P.desc + (size_t)t * P.M)[min(m_lo + b0 + ol, m_hi - 1)];
          unsigned long long multi = __ballot((db.info & (kSegMulti | kSegQuiet)) && b0 + ol < nobj);
          while (multi) {
            const int j = __builtin_ctzll(multi);
            multi &= multi - 1;
            // (the k field of a descriptor is exact for these objects: the piece ends inside the tile)
            single_object(t, m_lo + b0 + j, x_scale, true, seg_k(__shfl(db.info, j)));
          }
        }
        // an input beyond the f16 range (or not finite) shows as non-finite totals: the wave's tile is redone
        // exactly, unscaled, behind the chunk loop
        bool bad = false;
#pragma unroll
        for (int r = 0; r < NRT; r++)
#pragma unroll
          for (int c = 0; c < NCT; c++)
#pragma unroll
            for (int e = 0; e < 4; e++) bad |= !(__builtin_fabsf(tot0[r][c][e]) < INFINITY) || !(__builtin_fabsf(tot1[r][c][e]) < INFINITY);
        if (__ballot(bad)) {
          if (lane == 0) redo_tiles[w] |= 1ull << ka;
          clear_totals();
        } else {
          flush_tile(t, 1.0f / x_scale, true);
        }
        ka++;
      }
    };
    auto advance = [&]() __attribute__((always_inline)) {
      if (sa.c == nch - 1) tile_end(sa.tile);
      sa = sb;
      sb = sc;
      sc = next_stage(sc);
      left--;
    };
#pragma unroll 1
    while (left > 0) {
      chunk(0, X0);
      advance();
      if (left <= 0) break;
      chunk(1, X1);
      advance();
    }
  }
  // tiles done exactly and unscaled: every tile of unaligned rows or of fewer objects than a chunk, and the tiles
  // whose totals were not finite
  for (int k = 0; k < my_tiles; k++) {
    if (nch > 0 && !((redo_tiles[w] >> (k & 63)) & 1ull)) continue;
    const int t = tile_at(vb0 + k * vstep);
    if (nch == 0 && slow_next && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) slow_next[t] = 0u;
    for (int m = m_lo; m < m_hi; m++) single_object(t, m, 1.0f, false);
    flush_tile(t, 1.0f, false);
  }
  };  // body
  if constexpr (FORM == 2) {
    if (!wide_cur || (*wide_cur & 1u) != 0u) body(std::true_type{});
    else body(std::false_type{});
  } else {
    body(std::integral_constant<bool, FORM == 1>{});
  }
}

}  // namespace earhip
