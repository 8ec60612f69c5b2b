// gain_bf3.h — K1 on the bf16 matrix cores with every fp32 operand split into
// three bf16 pieces ("bf16x3"): the same bus-forming contraction as gain_mfma.h,
//
//     bus[col][s] = sum_m x_m(s) * g_m,col(s),      g linear in s on a curve segment
//
// at 16x the MFMA rate of the f32-in instruction.  The f32-in MFMA runs at the
// vector rate (157 TFLOP/s), which puts the floor of gain_mfma.h at ~0.70 ms for
// the headline scene — above its HBM floor (0.33 ms).  Here
//
//   * the ramp moves to the gain side.  For a workgroup tile starting at sample s0
//         g(s) = B0 + (s - s0) * B1,   B0 = (1-p0)*S + p0*E  (libear's gain AT s0,
//                                      gain_interpolator.hpp:272-274),
//                                      B1 = scale * (E - S)  (slope per sample)
//     so  bus = sum_m x*B0 + (s - s0) * sum_m x*B1 : two plain products with the
//     SAME left operand x, and (s - s0) is applied to the accumulator rows.  x is
//     split once per sample and object; B0/B1 are split once per workgroup tile
//     and object and shared by the workgroup's four waves through LDS;
//   * v = v1 + v2 + v3 with bf16 pieces (8+8+8 significand bits, RNE residuals)
//     is exact to ~2^-25; of the nine partial products the six of order <= 2^-16
//     are kept: x1b1, x1b2, x2b1, x1b3, x2b2, x3b1.  Products of bf16 pairs are
//     exact in the fp32 accumulate of v_mfma_f32_16x16x32_bf16;
//   * every chunk of 32 objects is accumulated from zero, smallest terms first,
//     and then added to the running total on the VALU (RNE): measured on the
//     headline scene this is closer to the float64 result than libear's own
//     sequential fp32 sum (tools/experiments/bf16x3_probe.hip).
//
// Fragment layout of v_mfma_f32_16x16x32_bf16 (8 bf16 = 4 VGPRs per operand):
//   A: lane l holds row l&15, k = 8*(l>>4) .. +7     B: column l&15, same k
//   D: lane l holds column l&15, rows 4*(l>>4) + e, e = 0..3
// k = the 32 objects of a chunk; row i of row tile r = sample 4*i + r of the
// wave's 64-sample tile (a lane's 4 inputs of one object are one 16-byte load and
// the D fragments of the 4 row tiles interleave into float4 stores); column j of
// column tile c = bus column col0 + 16*c + j.
//
// Objects with a curve point inside the workgroup tile, unaligned buffers and
// partial tiles use the exact f32 MFMA on the same accumulators (slow path, same
// arithmetic as gain_mfma.h).
#pragma once

#include <hip/hip_runtime.h>

#include "gain_kernels.h"
#include "gain_mfma.h"

namespace earhip {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// two floats -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// 8 floats -> three bf16x8 fragments with v = h + m + l (to ~2^-25 relative)
__device__ __forceinline__ void split_bf16x3(const float (&v)[8], u32x4 &h, u32x4 &m, u32x4 &l) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float a0 = v[2 * i], a1 = v[2 * i + 1];
    const uint32_t H = pack_bf16(a0, a1);
    const float r0 = a0 - bf16_lo(H), r1 = a1 - bf16_hi(H);  // exact
    const uint32_t M = pack_bf16(r0, r1);
    const float t0 = r0 - bf16_lo(M), t1 = r1 - bf16_hi(M);  // exact
    h[i] = H;
    m[i] = M;
    l[i] = pack_bf16(t0, t1);
  }
}

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4 &a, const u32x4 &b, const f32x4 &c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int kBf3Waves = 4;                  // waves (= 64-sample tiles) per workgroup
constexpr int kBf3Tile = 64 * kBf3Waves;      // samples per workgroup tile (descriptor tile)
constexpr int kBf3Chunk = 32;                 // objects per MFMA (k)

// P.ntiles / P.desc refer to WORKGROUP tiles of kBf3Tile samples.  zero_row: index
// of an all-zero gain row (appended to the point store by CurveSet::commit).
template <int NCT>
__global__ void __launch_bounds__(256, 2) k_gain_mix_bf3(GainMixParams P, int zero_row) {
  constexpr int NRT = 4, TS = 16 * NRT, CH = kBf3Chunk;
  constexpr int NFRAG = 2 * NCT * 3;  // {B0,B1} x column tiles x {h,m,l}
  __shared__ u32x4 bfrag[2][NFRAG + 6][64];  // + 6 never-read fragments: the lanes without a column write there
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kg = lane >> 4;
  const int wgtile = xcd_tile(blockIdx.x, gridDim.x);
  const int nparts = gridDim.y;
  const int part = blockIdx.y;
  const int m_lo = (int)(((int64_t)P.M * part) / nparts);
  const int m_hi = (int)(((int64_t)P.M * (part + 1)) / nparts);
  const int col0 = blockIdx.z * 16 * NCT;
  const int tile_s0 = wgtile * kBf3Tile + w * TS;  // first sample of this wave's tile
  const int tile_len = max(0, min(TS, P.nsamples - tile_s0));
  const int64_t tile_t0 = P.t_call + tile_s0;
  const int64_t tile_t1 = tile_t0 + tile_len;
  const SegDesc *__restrict__ dtile = P.desc + (size_t)wgtile * P.M;
  const float *__restrict__ gain = P.ps.gain;
  const unsigned rowlen = (unsigned)P.ps.row;

  f32x4 tot[NRT][NCT];
#pragma unroll
  for (int r = 0; r < NRT; r++)
#pragma unroll
    for (int c = 0; c < NCT; c++) tot[r][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  // ---- slow path: one object, all its pieces inside this wave's tile, exact f32
  // MFMA with k = {a, b} of ONE object (k slots 2, 3 idle).  Walks the segments like
  // GainInterpolator::process (gain_interpolator.hpp:58-86).
  auto single_object = [&](int m) {
    if (tile_len <= 0) return;
    const int base = P.ps.off[m], n = P.ps.off[m + 1] - base;
    const float *row = P.in + (size_t)m * P.in_stride + tile_s0;
    const bool is_b = kg & 1;
    const bool slot0 = kg < 2;
    int k = upper_bound_time(P.ps.time + base, n, tile_t0);
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
          a[r] = x * coef;
        }
        const int grow = dk.row + ((ramp && is_b && slot0) ? 1 : 0);
        const float *gp = gain + (size_t)grow * rowlen + col0 + li;
#pragma unroll
        for (int c = 0; c < NCT; c++) gv[c] = gp[c * 16];
#pragma unroll
        for (int r = 0; r < NRT; r++)
#pragma unroll
          for (int c = 0; c < NCT; c++)
            tot[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], gv[c], tot[r][c], 0, 0, 0);
        cur = r1;
      }
      if (!(dk.info & kSegMulti)) break;
      k++;
    }
  };

  const int nobj = m_hi - m_lo;
  const int nch = (P.vec_ok && nobj >= CH) ? (nobj + CH - 1) / CH : 0;

  if (nch > 0) {
    // lane-constant part of the input address: byte offset of this lane's float4
    // (lanes past the end of the call re-read the last vector; never stored).  All
    // loads below are (wave-uniform 64-bit base) + (32-bit lane offset).
    const int nvec = (P.nsamples + 3) & ~3;
    const unsigned xs = (unsigned)min(tile_s0 + li * NRT, nvec - 4);
    const unsigned xlane = ((unsigned)(kg * 8) * (unsigned)P.in_stride + xs) * 4u;
    const unsigned bcol_e = (unsigned)(col0 + min(lane, 16 * NCT - 1));  // the lane's gain column
    // fragment this lane fills: B0 pieces at bfr .. bfr+2, B1 pieces NCT*3 further
    const int blane = w * 16 + (lane & 15);
    const int bfr = lane < 16 * NCT ? (lane >> 4) * 3 : NFRAG;
    const int bfr1 = lane < 16 * NCT ? NCT * 3 : 3;
    // first object of chunk c; the last chunk is moved back to end at m_hi (its
    // objects that the previous chunk already covered get the zero row)
    auto chunk_base = [&](int c) { return min(m_lo + c * CH, m_hi - CH); };

    // inputs of chunk c: x[q] = 4 consecutive samples of object chunk_base + 8kg + q
    auto load_x = [&](int c, f32x4 (&x)[8]) {
      const char *bp = reinterpret_cast<const char *>(P.in + (size_t)chunk_base(c) * P.in_stride);
#pragma unroll
      for (int q = 0; q < 8; q++)
        x[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(
            bp + (size_t)q * P.in_stride * sizeof(float) + xlane));
    };
    // What this WAVE converts for chunk c: objects chunk_base + 8w + q.  Lane q < 8
    // fetches and digests the descriptor of object q; the results are broadcast
    // into SGPRs (v_readlane) when the chunk's gains are requested.
    struct LaneDesc {
      int row_s, row_e;
      float p0, scale;
    };
    // the raw descriptor is only fetched here; it is digested (digest_desc) one chunk
    // later, right before its fields are broadcast.  Digesting at once would put a wait
    // for THIS load — the youngest one, so in effect vmcnt(0) — into the chunk that has
    // just requested its inputs, and serialise the whole input prefetch.
    auto load_desc = [&](int c) {
      const int cc = min(c, nch - 1);
      const int m = chunk_base(cc) + w * 8 + (lane & 7);
      return *reinterpret_cast<const int4 *>(dtile + m);
    };
    auto digest_desc = [&](const int4 &d, int c) {
      const int cc = min(c, nch - 1);
      const int m = chunk_base(cc) + w * 8 + (lane & 7);
      const bool valid = m >= m_lo + cc * CH && !(d.w & kSegMulti);
      LaneDesc L;
      L.row_s = valid ? d.x : zero_row;
      L.row_e = L.row_s + ((valid && (d.w & kSegRamp)) ? 1 : 0);
      const float scale = __int_as_float(d.z);
      L.p0 = (float)d.y * scale;  // gain_interpolator.hpp:272 at the tile start
      L.scale = scale;            // constant segments: scale = 0, d0 = 0
      return L;
    };
    struct ChunkDesc {
      float p0[8], scale[8];
    };
    auto load_gains = [&](const int4 &raw, int c, ChunkDesc &D, float (&S)[8], float (&E)[8]) {
      const LaneDesc L = digest_desc(raw, c);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const unsigned rs = (unsigned)__builtin_amdgcn_readlane(L.row_s, q);
        const unsigned re = (unsigned)__builtin_amdgcn_readlane(L.row_e, q);
        D.p0[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.p0), q));
        D.scale[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.scale), q));
        // wave-uniform row pointer (scalar arithmetic) + the lane's column as a 32-bit offset:
        // no vector instructions for the addresses
        const float *rps = gain + (size_t)rs * rowlen, *rpe = gain + (size_t)re * rowlen;
        S[q] = rps[bcol_e];
        E[q] = rpe[bcol_e];
      }
    };
    // B0 (gain at the tile start, part 0) or B1 (slope, part 1) of the wave's 8 objects -> LDS
    auto store_b = [&](const ChunkDesc &D, const float (&S)[8], const float (&E)[8], int buf, int part) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; q++)
        v[q] = part == 0 ? __builtin_fmaf(1.0f - D.p0[q], S[q], D.p0[q] * E[q])
                         : D.scale[q] * (E[q] - S[q]);
      u32x4 h, m, l;
      split_bf16x3(v, h, m, l);
      u32x4 *f = &bfrag[buf][bfr][blane] + (part ? bfr1 * 64 : 0);
      f[0] = h;
      f[64] = m;
      f[128] = l;
    };

    // (s - s0) of the rows of this lane's D fragments: sample 64w + 16kg + 4e + r
    const float wf0 = (float)(w * TS + kg * 16);

    f32x4 xc[8], xn[8];
    int4 L;
    {
      float S[8], E[8];
      ChunkDesc D;
      L = load_desc(0);
      load_gains(L, 0, D, S, E);
      load_x(0, xc);
      store_b(D, S, E, 0, 0);
      store_b(D, S, E, 0, 1);
      L = load_desc(1);
    }
#pragma unroll 1
    for (int c = 0; c < nch; c++) {
      const int buf = c & 1;
      __syncthreads();  // B fragments of chunk c are in bfrag[buf]; bfrag[buf^1] is free
      float S[8], E[8];
      ChunkDesc D;
      load_gains(L, c + 1, D, S, E);  // chunk c+1 (its descriptors were fetched one chunk ago)
      L = load_desc(c + 2);

      // A fragments: row tile r = sample 4*li + r of the 8 objects of this lane
      u32x4 ah[NRT], am[NRT], al[NRT];
      // Split in 2 x 2 blocks: a bf16 pair packs two OBJECTS (q, q+1) of one row tile, while
      // the exact residual subtractions pair two SAMPLES (r, r+1) of one object, which are
      // neighbours in the loaded float4 — packed subtractions without operand moves.
#pragma unroll
      for (int qp = 0; qp < 4; qp++)
#pragma unroll
        for (int rp = 0; rp < NRT; rp += 2) {
          const f32x4 &x0 = xc[2 * qp], &x1 = xc[2 * qp + 1];
          const uint32_t H0 = pack_bf16(x0[rp], x1[rp]), H1 = pack_bf16(x0[rp + 1], x1[rp + 1]);
          const f32x2 r0 = f32x2{x0[rp], x0[rp + 1]} - f32x2{bf16_lo(H0), bf16_lo(H1)};  // object 2qp
          const f32x2 r1 = f32x2{x1[rp], x1[rp + 1]} - f32x2{bf16_hi(H0), bf16_hi(H1)};  // object 2qp+1
          const uint32_t M0 = pack_bf16(r0[0], r1[0]), M1 = pack_bf16(r0[1], r1[1]);
          const f32x2 t0 = r0 - f32x2{bf16_lo(M0), bf16_lo(M1)};
          const f32x2 t1 = r1 - f32x2{bf16_hi(M0), bf16_hi(M1)};
          ah[rp][qp] = H0;
          ah[rp + 1][qp] = H1;
          am[rp][qp] = M0;
          am[rp + 1][qp] = M1;
          al[rp][qp] = pack_bf16(t0[0], t1[0]);
          al[rp + 1][qp] = pack_bf16(t0[1], t1[1]);
        }
      // 2*NCT blocks (column tile ct = blk >> 1, operand blk & 1: B0 / B1) of 24 MFMAs:
      // six partial products per operand pair, smallest first, accumulated from zero.
      // A VALU instruction costs its 4 cycles on top of the MFMAs unless it sits
      // directly behind one (tools/experiments/coissue2.hip: one is free per MFMA), so
      // everything that does not depend on this chunk's MFMAs is woven between them:
      // the fold of the previous block, the conversion of the next chunk's gains and
      // the hand-over of the next chunk's inputs.
      __builtin_amdgcn_sched_barrier(0);  // the splitting above stays above
      constexpr int NBLK = 2 * NCT;
      u32x4 b[2][3];
      f32x4 t[2][NRT];
      const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
      auto load_b = [&](int blk, u32x4 (&bb)[3]) {
#pragma unroll
        for (int q = 0; q < 3; q++) bb[q] = bfrag[buf][((blk & 1) * NCT + (blk >> 1)) * 3 + q][lane];
      };
      auto fold = [&](int blk, const f32x4 (&tt)[NRT]) {
        const int ct = blk >> 1;
#pragma unroll
        for (int r = 0; r < NRT; r++)
#pragma unroll
          for (int e = 0; e < 4; e++) {
            if ((blk & 1) == 0) tot[r][ct][e] += tt[r][e];
            else tot[r][ct][e] = __builtin_fmaf(wf0 + (float)(4 * e + r), tt[r][e], tot[r][ct][e]);
          }
      };
      load_b(0, b[0]);
#pragma unroll
      for (int blk = 0; blk < NBLK; blk++) {
        u32x4(&bc)[3] = b[blk & 1];
        f32x4(&tc)[NRT] = t[blk & 1];
        if (blk + 1 < NBLK) load_b(blk + 1, b[(blk + 1) & 1]);
#pragma unroll
        for (int r = 0; r < NRT; r++) tc[r] = mfma_bf16(al[r], bc[0], zero);
#pragma unroll
        for (int r = 0; r < NRT; r++) tc[r] = mfma_bf16(am[r], bc[1], tc[r]);
#pragma unroll
        for (int r = 0; r < NRT; r++) tc[r] = mfma_bf16(ah[r], bc[2], tc[r]);
#pragma unroll
        for (int r = 0; r < NRT; r++) tc[r] = mfma_bf16(am[r], bc[0], tc[r]);
#pragma unroll
        for (int r = 0; r < NRT; r++) tc[r] = mfma_bf16(ah[r], bc[1], tc[r]);
#pragma unroll
        for (int r = 0; r < NRT; r++) tc[r] = mfma_bf16(ah[r], bc[0], tc[r]);
        if (blk > 0) fold(blk - 1, t[(blk - 1) & 1]);
        // The vector-memory counter is in order: a wait for one load waits for every older
        // one.  So the gain rows (requested at the top of the chunk) are consumed FIRST, in
        // blocks 0 and 1, and only then are the next chunk's inputs requested; they are waited
        // for at the hand-over after the last block.  (With the inputs requested at the top,
        // the first use of a gain row waited for them too and serialised the prefetch.)
        const bool conv0 = blk == 0, conv1 = blk == 1;
        if (conv0) store_b(D, S, E, buf ^ 1, 0);  // next chunk's gains (after the last chunk:
        if (conv1) store_b(D, S, E, buf ^ 1, 1);  // written, never read)
        if (blk == (NBLK > 2 ? 2 : 1)) {
          load_x(min(c + 1, nch - 1), xn);
          __builtin_amdgcn_sched_barrier(0);  // keep the requests here, not at the end of the chunk
        }
        // issue order: the LDS reads first, then every MFMA followed by 1-3 VALU instructions
        if (blk + 1 < NBLK) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        if (conv0 || conv1) {
#pragma unroll
          for (int k = 0; k < 24; k++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
          }
        } else if (blk > 0) {
#pragma unroll
          for (int k = 0; k < 24; k++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          }
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
        }
      }
      fold(NBLK - 1, t[(NBLK - 1) & 1]);
#pragma unroll
      for (int q = 0; q < 8; q++) xc[q] = xn[q];
    }

    // objects with curve points inside this workgroup tile (zero rows above)
    for (int b0 = 0; b0 < nobj; b0 += 64) {
      const SegDesc db = dtile[min(m_lo + b0 + lane, m_hi - 1)];
      unsigned long long multi = __ballot((db.info & kSegMulti) && b0 + lane < nobj);
      while (multi) {
        const int j = __builtin_ctzll(multi);
        multi &= multi - 1;
        single_object(m_lo + b0 + j);
      }
    }
  } else {
    for (int m = m_lo; m < m_hi; m++) single_object(m);  // unaligned rows
  }

  if (tile_len <= 0) return;
  // D fragment of row tile r: rows 4kg + e = samples 16kg + 4e + r: for fixed e the
  // four row tiles are 4 consecutive samples
  float *op = P.out + (size_t)blockIdx.y * P.part_stride + tile_s0;
#pragma unroll
  for (int c = 0; c < NCT; c++) {
    const int col = col0 + c * 16 + li;
    if (col >= P.ncols) continue;
    float *o = op + (size_t)col * P.out_stride;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int s = kg * 16 + e * 4;
      const f32x4 v = {tot[0][c][e], tot[1][c][e], tot[2][c][e], tot[3][c][e]};
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

}  // namespace earhip
