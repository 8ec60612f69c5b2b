// gain_mfma.h — K1 on the matrix cores: the bus-forming contraction
//
//     bus[col][s] = sum_m  a_m(s) * S_m[col] + b_m(s) * E_m[col],
//     a_m(s) = x_m(s) * (1 - p_m(s)),  b_m(s) = x_m(s) * p_m(s)
//
// is a dense [samples x 2M] . [2M x columns] product in fp32 (libear's
// LinearInterpMatrix accumulation, include/ear/dsp/gain_interpolator.hpp:264-277,
// with the ramp folded into the input).  It runs on v_mfma_f32_16x16x4_f32, which
// is exact fp32 (a k-ordered fmaf chain, same rounding as the VALU formulation)
// at the full fp32 rate of the chip, and — unlike the VALU formulation — takes
// the gains as an ordinary per-lane operand:
//
//   A fragment (1 VGPR): lane l holds A[row l&15][k = l>>4]
//   B fragment (1 VGPR): lane l holds B[k = l>>4][column l&15]
//   k = 0..3 of one step = {a, b} of two consecutive objects
//
// Which sample an MFMA row stands for, and which bus column an MFMA column, is
// free to choose.  Row i of row-tile r is sample i*NRT + r and column j of
// column-tile c is bus column j*NCT + c, so that a lane's NRT inputs and its NCT
// gains are CONTIGUOUS in memory: one 16-byte load of x and one 12-byte load of
// gains per step and lane instead of NRT + NCT dword loads (the texture-address
// unit was 54% busy with those), and the D fragments of the NRT row tiles
// interleave back into runs of NRT consecutive samples for the stores.
//
// so the gain rows arrive through plain coalesced vector loads (deep, in-order
// prefetch) instead of wave-uniform scalar loads, whose cache cannot sustain a
// 200 MB stream of misses (measured: the SGPR-operand kernel idles 69% of its
// wave cycles in s_waitcnt, profiles/r01_*).
//
// One wave owns NRT x 16 samples x NCT x 16 columns (NRT*NCT float4 accumulators)
// and walks its share of the objects two at a time; the waves of a workgroup
// split the objects and are combined through LDS in a fixed order.
#pragma once

#include <hip/hip_runtime.h>

#include "gain_kernels.h"

namespace earhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PairRegs {
  int row, d0, info;  // per-lane copy of the SegDesc of "its" object (lanes 0-31 / 32-63)
  float scale;
};

template <int NCT, int NRT>
__global__ void __launch_bounds__(512) k_gain_mix_mfma(GainMixParams P) {
  constexpr int TS = 16 * NRT;      // samples per tile
  constexpr int NC = 16 * NCT;      // columns per wave
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [ngroups][NRT*NCT][64][4]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // waves of a workgroup = column groups x adjacent tiles x object splits.  Waves
  // on adjacent tiles walk the SAME objects, so their gain-row loads (identical
  // addresses, a few hundred cycles apart) are served by the CU's L1 instead of
  // L2: at 64-sample tiles the gain rows are otherwise the larger L2->L1 stream.
  const int g = wave % P.ngroups;
  const int tw = (wave / P.ngroups) % P.tiles_per_wg;
  const int ws = wave / (P.ngroups * P.tiles_per_wg);
  // Workgroup b runs on XCD b % 8 (each XCD has its own L2).  Tiles that are
  // neighbours in time share gain rows, so give every XCD one contiguous run of
  // tiles instead of every 8th tile (speed only; any mapping is correct).
  const int tile_raw = xcd_tile(blockIdx.x, gridDim.x) * P.tiles_per_wg + tw;
  const bool idle = tile_raw >= P.ntiles;  // ragged last workgroup
  const int tile = idle ? P.ntiles - 1 : tile_raw;
  const int nparts = P.wsplit * gridDim.y;
  const int part = blockIdx.y * P.wsplit + ws;
  const int m_lo = (int)(((int64_t)P.M * part) / nparts);
  const int m_hi = (int)(((int64_t)P.M * (part + 1)) / nparts);
  const int col0 = (blockIdx.z * P.ngroups + g) * NC;

  const int li = lane & 15;        // sample (A) / column (B) index inside a 16-tile
  const int kk = lane >> 4;        // k index of this lane: object slot kk>>1, {a,b} = kk&1
  const int slot = kk >> 1;
  const bool is_b = kk & 1;
  const int tile_s0 = tile * TS;   // first sample of the tile inside the call
  const int tile_len = min(TS, P.nsamples - tile_s0);
  const int64_t tile_t0 = P.t_call + tile_s0;
  const int64_t tile_t1 = tile_t0 + tile_len;

  f32x4 acc[NRT][NCT];
#pragma unroll
  for (int r = 0; r < NRT; r++)
#pragma unroll
    for (int c = 0; c < NCT; c++) acc[r][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  auto mma = [&](const float (&a)[NRT], const float (&gv)[NCT]) {
#pragma unroll
    for (int r = 0; r < NRT; r++)
#pragma unroll
      for (int c = 0; c < NCT; c++)
        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], gv[c], acc[r][c], 0, 0, 0);
  };

  // Generic (slow) path: the pieces of ONE object from segment k on, starting at
  // sample `cur` of the tile, one MFMA step per piece with the lanes of the
  // second object slot idle.  Walks the segments like GainInterpolator::process
  // (gain_interpolator.hpp:58-86).  Used for partial tiles, an odd last object
  // and for curves with points inside the tile (after their first piece).
  auto single_object = [&](int m, int k, int cur) {
    const int base = P.ps.off[m], n = P.ps.off[m + 1] - base;
    const float *row = P.in + (size_t)m * P.in_stride + tile_s0;
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
          coef = (slot == 0 && s >= cur && s < r1) ? coef : 0.0f;
          a[r] = x * coef;
        }
        const int grow = dk.row + ((ramp && is_b && slot == 0) ? 1 : 0);
        const float *gp = P.ps.gain + (size_t)grow * P.ps.row + col0 + li * NCT;
#pragma unroll
        for (int c = 0; c < NCT; c++) gv[c] = gp[c];
        mma(a, gv);
        cur = r1;
      }
      if (!(dk.info & kSegMulti)) break;
      k++;
    }
  };
  auto whole_object = [&](int m) {
    const SegDesc d = P.desc[(size_t)tile * P.M + m];
    single_object(m, seg_k(d.info), 0);
  };

  const int npairs = idle ? 0 : (m_hi - m_lo) >> 1;
  const int m_last_obj = m_hi - 1;
  if (idle) {
  } else if (tile_len < TS) {
    for (int m = m_lo; m < m_hi; m++) whole_object(m);  // last, partial tile of a call
  } else if (npairs > 0 && P.vec_ok) {
    // Fast path, full tile, 16-byte aligned rows: two-deep software pipeline over object pairs.  All
    // addresses are (wave-uniform base) + (loop-invariant 32-bit lane offset), so a
    // step costs no address arithmetic on the VALU; descriptors are fetched two
    // steps ahead, inputs and gain rows one step ahead of the MFMAs that use them.
    const unsigned xoff = (unsigned)slot * (unsigned)P.in_stride + (unsigned)(li * NRT);
    const unsigned goff0 = (unsigned)(col0 + li * NCT);
    const float c0 = is_b ? 0.0f : 1.0f;       // constant segment: a = x, b = 0
    const float c1r = is_b ? 1.0f : -1.0f;     // ramp: coef = c0 + c1 * p  (= p or 1 - p)
    auto pair_index = [&](int i) { return m_lo + 2 * min(i, npairs - 1); };
    // Descriptors of this tile are contiguous over objects: lane j of batch b
    // holds the descriptor of object m_lo + 64*b + j (one coalesced 1 KB load per
    // 32 steps, issued a whole batch ahead), and a step picks the two it needs with
    // a cross-lane read.  No per-step descriptor load, and the gain-row address no
    // longer waits on memory.
    const int4 *dtile = reinterpret_cast<const int4 *>(P.desc + (size_t)tile * P.M);
    auto load_batch = [&](int b) { return dtile[min(m_lo + 64 * b + lane, m_last_obj)]; };
    int4 batch0 = load_batch(0), batch1 = load_batch(1);
    // descriptor of pair i out of the batch register that holds it
    auto desc_from = [&](const int4 &reg, int i) {
      const int src = 2 * (min(i, npairs - 1) & 31) + slot;
      int4 d;
      d.x = __shfl(reg.x, src, 64);
      d.y = __shfl(reg.y, src, 64);
      d.z = __shfl(reg.z, src, 64);
      d.w = __shfl(reg.w, src, 64);
      return d;
    };
    auto load_desc = [&](int i) { return desc_from(batch0, i); };  // prologue: pairs 0..3
    auto load_x = [&](int i, float (&x)[NRT]) {
      const f32x4 *xp = reinterpret_cast<const f32x4 *>(P.in + (size_t)pair_index(i) * P.in_stride +
                                                        tile_s0 + xoff);
#pragma unroll
      for (int q = 0; q < NRT / 4; q++) {
        const f32x4 v = __builtin_nontemporal_load(xp + q);  // streamed once: keep L2 for gain rows
#pragma unroll
        for (int e = 0; e < 4; e++) x[q * 4 + e] = v[e];
      }
    };
    auto load_g = [&](const int4 d, float (&gv)[NCT]) {
      // ramp: k even -> start row, k odd -> end row; constant: the one row (b = 0)
      const unsigned row = (unsigned)d.x + (((d.w & kSegRamp) && is_b) ? 1u : 0u);
      const float *gp = P.ps.gain + (row * (unsigned)P.ps.row + goff0);
#pragma unroll
      for (int c = 0; c < NCT; c++) gv[c] = gp[c];
    };
    // A fragments of a pair from its inputs x and descriptor d
    auto make_a = [&](const int4 d, const float (&x)[NRT], float (&a)[NRT]) {
      // coef(s) = c0 + c1 * p(s), p(s) = (float)(d0 + s) * scale (gain_interpolator.hpp:272).
      // The lane's NRT samples are consecutive, so coef advances by c1*scale per
      // sample: one conversion + NRT fused steps instead of NRT conversions (the
      // difference to converting every index is one rounding of p, ~6e-8 relative).
      // An object whose curve has a point inside this tile ("multi") contributes
      // nothing here; it is rendered by the generic path after the loop.
      const bool ramp = d.w & kSegRamp;
      const bool multi = d.w & kSegMulti;
      const float c1 = (ramp && !multi) ? c1r : 0.0f;
      const float c0m = multi ? 0.0f : c0;
      const float scale = __int_as_float(d.z);
      const float cs = c1 * scale;
      const float coef0 = __builtin_fmaf(c1, (float)(d.y + li * NRT) * scale, c0m);
#pragma unroll
      for (int r = 0; r < NRT; r++) a[r] = x[r] * __builtin_fmaf((float)r, cs, coef0);
    };

    // One step = the MFMAs of pair i and, woven between them by the scheduling
    // directives, everything later pairs need: the A fragments of pair i+1 (its
    // inputs were requested 4 steps ago), the refill of that input slot with pair
    // i+5, the gain rows of pair i+2 and the descriptor of pair i+4 (cross-lane
    // reads, no memory).  The loop around it must stay free of other branches: with
    // a rare-path branch or the batch reload inside, the compiler's wait-count
    // insertion fell back to `s_waitcnt vmcnt(0)` at the top of every step, which
    // silently cancelled the whole prefetch pipeline.
    int4 d1 = load_desc(1), d2 = load_desc(2), d3 = load_desc(3);
    auto step = [&](int i, const float (&a_cur)[NRT], float (&a_nxt)[NRT], float (&g_cur)[NCT],
                    float (&x_nxt)[NRT], const int4 &dreg) {
      mma(a_cur, g_cur);
      make_a(d1, x_nxt, a_nxt);
      load_x(i + 5, x_nxt);
      load_g(d2, g_cur);
      const int4 d4 = desc_from(dreg, i + 4);
#pragma unroll
      for (int k = 0; k < NRT * NCT; k++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // 2 VALU
        if (k < NRT / 4 + 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
      }
      d1 = d2;
      d2 = d3;
      d3 = d4;
    };

    // x ring: set j holds the inputs of the pairs p with p % 4 == j
    float aA[NRT], aB[NRT], gA[NCT], gB[NCT], x0[NRT], x1[NRT], x2[NRT], x3[NRT];
    {
      const int4 d0 = load_desc(0);
      load_x(0, x0);
      load_x(1, x1);
      load_x(2, x2);
      load_x(3, x3);
      load_g(d0, gA);
      load_g(d1, gB);
      make_a(d0, x0, aA);
      load_x(4, x0);
    }
    // step i consumes (a, g) of pair i, builds the A fragments of pair i+1 from its
    // ring slot and refills that slot with pair i+5, refills g with pair i+2
    // A batch register must not be touched while its reload is in flight (even a
    // select that reads it would force a wait), so the groups of a batch name their
    // register statically: the first 7 groups of 4 steps read descriptors (pairs
    // i+4..i+7) from the current batch's register, the 8th from the next batch's,
    // whose reload was requested 28 steps earlier.
    auto group = [&](int i, const int4 &dreg) {  // branch-free
      step(i, aA, aB, gA, x1, dreg);
      step(i + 1, aB, aA, gB, x2, dreg);
      step(i + 2, aA, aB, gA, x3, dreg);
      step(i + 3, aB, aA, gB, x0, dreg);
    };
    // batch0 = descriptors of the current batch, batch1 = of the next one
    for (int ib = 0; ib + 4 <= npairs; ib += 32) {
      if (ib > 0) {  // batch1 was requested 32 steps ago
        batch0 = batch1;
        batch1 = load_batch((ib >> 5) + 1);
      }
      const int iend = min(ib + 32, npairs & ~3);
      const int isplit = min(iend, ib + 28);
      int i = ib;
#pragma unroll 1
      for (; i < isplit; i += 4) group(i, batch0);
      if (i < iend) group(i, batch1);
    }
    {
      const int i = npairs & ~3;
      if (i < npairs) {  // 1..3 remaining pairs (descriptor register chosen at run time)
        // batch0 holds the batch of pair i; later pairs of a clamped index stay in it
        const int4 reg = batch0;
        step(i, aA, aB, gA, x1, reg);
        if (i + 1 < npairs) {
          step(i + 1, aB, aA, gB, x2, reg);
          if (i + 2 < npairs) step(i + 2, aA, aB, gA, x3, reg);
        }
      }
    }
    // Objects with curve points inside this tile ("multi", zeroed above): their pieces
    // were laid out by k_piece_list as one compact list per tile.  Two pieces per MFMA
    // step (k = {a, b} of two pieces), shared evenly by all waves that work on this
    // tile, whatever objects they own; each piece masks the samples outside [r0, r1).
    const bool have_list = P.pl.count != nullptr;
    if (have_list) {
      const int n_e = P.pl.count[tile];
      const int e_lo = (int)(((int64_t)n_e * part) / nparts);
      const int e_hi = (int)(((int64_t)n_e * (part + 1)) / nparts);
      const int4 *pd = reinterpret_cast<const int4 *>(P.pl.d + (size_t)tile * P.pl.cap);
      const int *pm = P.pl.m + (size_t)tile * P.pl.cap;
      struct Piece {
        int4 d;
        int m;
      };
      auto load_piece = [&](int e) {
        const int idx = min(e + slot, max(e_hi - 1, 0));
        Piece p;
        p.d = pd[idx];
        p.m = pm[idx];
        if (e + slot >= e_hi) p.d.w = 0;  // r0 = r1 = 0: contributes nothing
        return p;
      };
      auto load_px = [&](const Piece &p, float (&x)[NRT]) {
        const f32x4 *xp = reinterpret_cast<const f32x4 *>(P.in + (size_t)p.m * P.in_stride + tile_s0 + li * NRT);
#pragma unroll
        for (int q = 0; q < NRT / 4; q++) {
          const f32x4 v = xp[q];
#pragma unroll
          for (int e = 0; e < 4; e++) x[q * 4 + e] = v[e];
        }
      };
      auto load_pg = [&](const Piece &p, float (&gv)[NCT]) {
        const unsigned row = (unsigned)p.d.x + (((p.d.w & kSegRamp) && is_b) ? 1u : 0u);
        const float *gp = P.ps.gain + (row * (unsigned)P.ps.row + goff0);
#pragma unroll
        for (int c = 0; c < NCT; c++) gv[c] = gp[c];
      };
      auto piece_a = [&](const Piece &p, const float (&x)[NRT], float (&a)[NRT]) {
        const bool ramp = p.d.w & kSegRamp;
        const int r0 = piece_r0(p.d.w), r1 = seg_r1(p.d.w);
        const float scale = __int_as_float(p.d.z);
#pragma unroll
        for (int r = 0; r < NRT; r++) {
          const int s = li * NRT + r;
          const float pp = (float)(p.d.y + s) * scale;  // gain_interpolator.hpp:272
          float coef = ramp ? (is_b ? pp : 1.0f - pp) : c0;
          coef = (s >= r0 && s < r1) ? coef : 0.0f;
          a[r] = x[r] * coef;
        }
      };
      if (e_lo < e_hi) {
        Piece p0 = load_piece(e_lo), p1 = load_piece(e_lo + 2);
        float px0[NRT], pg0[NCT];
        load_px(p0, px0);
        load_pg(p0, pg0);
        for (int e = e_lo; e < e_hi; e += 2) {
          const Piece p2 = load_piece(e + 4);
          float px1[NRT], pg1[NCT], a[NRT];
          load_px(p1, px1);
          load_pg(p1, pg1);
          piece_a(p0, px0, a);
          mma(a, pg0);
          p0 = p1;
          p1 = p2;
#pragma unroll
          for (int r = 0; r < NRT; r++) px0[r] = px1[r];
#pragma unroll
          for (int c = 0; c < NCT; c++) pg0[c] = pg1[c];
        }
      }
    }
    // multi objects without a piece list (none built, or the tile's list was full):
    // all their pieces through the generic path
    for (int b0 = 0; b0 < 2 * npairs; b0 += 64) {
      const int4 db = dtile[min(m_lo + b0 + lane, m_last_obj)];
      const bool slow = (db.w & kSegMulti) && (!have_list || (db.w & kSegSlow));
      unsigned long long multi = __ballot(slow && b0 + lane < 2 * npairs);
      while (multi) {
        const int j = __builtin_ctzll(multi);
        multi &= multi - 1;
        whole_object(m_lo + b0 + j);
      }
    }
    if ((m_hi - m_lo) & 1) {  // odd object count: the last one alone (unless its pieces are listed)
      const int iw = P.desc[(size_t)tile * P.M + m_hi - 1].info;
      if (!(iw & kSegMulti) || !have_list || (iw & kSegSlow)) whole_object(m_hi - 1);
    }
  } else {
    for (int m = m_lo; m < m_hi; m++) whole_object(m);  // single object or unaligned rows
  }

  // combine the in-workgroup object splits through LDS, highest split first;
  // accumulators keep their fragment layout: [frag][lane] float4
  f32x4 *slab = reinterpret_cast<f32x4 *>(lds) + (size_t)(g * P.tiles_per_wg + tw) * NRT * NCT * 64 + lane;
  for (int r = P.wsplit - 1; r >= 1; r--) {
    if (ws == r) {
#pragma unroll
      for (int f = 0; f < NRT * NCT; f++) {
        f32x4 v = acc[f / NCT][f % NCT];
        if (r != P.wsplit - 1) v += slab[f * 64];
        slab[f * 64] = v;
      }
    }
    __syncthreads();
  }
  if (ws != 0 || idle) return;

  // D fragment of row-tile r: lane holds bus column col0 + li*NCT + c and MFMA rows
  // kk*4 + e (e = 0..3), i.e. samples (kk*4 + e)*NRT + r: for fixed e the NRT row
  // tiles form a run of NRT consecutive samples
  float *op = P.out + (size_t)blockIdx.y * P.part_stride + tile_s0;
#pragma unroll
  for (int c = 0; c < NCT; c++) {
    const int col = col0 + li * NCT + c;
    f32x4 v[NRT];
#pragma unroll
    for (int r = 0; r < NRT; r++) {
      v[r] = acc[r][c];
      if (P.wsplit > 1) v[r] += slab[(r * NCT + c) * 64];
    }
    if (col >= P.ncols) continue;
    float *o = op + (size_t)col * P.out_stride;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int s = (kk * 4 + e) * NRT;
#pragma unroll
      for (int q = 0; q < NRT / 4; q++) {
        const f32x4 w = {v[q * 4 + 0][e], v[q * 4 + 1][e], v[q * 4 + 2][e], v[q * 4 + 3][e]};
        if (P.vec_ok && s + q * 4 + 3 < tile_len) {
          *reinterpret_cast<f32x4 *>(o + s + q * 4) = w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (s + q * 4 + i < tile_len) o[s + q * 4 + i] = w[i];
        }
      }
    }
  }
}

}  // namespace earhip
