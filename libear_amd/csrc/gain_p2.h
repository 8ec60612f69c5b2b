// gain_p2.h — K1 on the f16 matrix cores over per-tile PIECE lists ("f16x2 pieces"): the split-operand
// formulation of gain_h2.h,
//
//     bus[col][s] = sum_k x_m(k)(s) [r0_k <= s < r1_k] * (B0_k,col + (s - s0) * B1_k,col),
//
// with the k dimension of the MFMA running over the PIECES of the workgroup tile instead of its objects: a
// piece is one linear stretch (object, curve segment, sample range) of a gain curve inside the tile —
// what one iteration of libear's segment walk handles (GainInterpolator::process,
// include/ear/dsp/gain_interpolator.hpp:58-86).  An object whose segment covers the whole tile is one
// piece; every curve point inside the tile adds one.  The range of a piece is applied to the INPUT (a
// per-sample scale of 0 or 2^k folded into the prescale the operand split needs anyway); B0 / B1 of a
// piece are its line extended to the tile start s0, so all pieces share the accumulator-side factor
// (s - s0).  Four lists per tile, written by k_piece_list in object order (deterministic):
//
//     constant pieces covering the whole tile | constant pieces with a range |
//     ramp pieces covering the whole tile     | ramp pieces with a range
//
// Constant pieces have B1 = 0, so their chunks issue only the B0 half of the MFMAs (36 instead of 72
// per 32 pieces, 64 samples and 48 columns) and convert half the gains; pieces with a range pay the range
// test on their inputs.  The cost of a call is therefore proportional to its pieces: block-aligned ramps
// cost what they cost in gain_h2.h, static gains half of that, and metadata that ignores the block grid
// (ADM blocks at arbitrary times) pays for the curve points it actually has — there is no alignment rule
// and no cliff.
//
// A ramp piece that starts inside the tile is extended BACK to s0: |B0| <= (1 + 2 |p0|) gmax with
// p0 = (s0 - start) / length.  The gains are prescaled to 2^12 (not 2^14 as in gain_h2.h), which keeps
// |p0| <= 7 inside the f16 range; objects with a steeper piece (a ramp shorter than a seventh of its
// offset into the tile) are listed separately and take the exact per-object path, like objects whose
// pieces do not fit the lists.
#pragma once

#include <hip/hip_runtime.h>

#include "gain_h2.h"

namespace earhip {

struct Piece {
  uint32_t mr;  // object m | r0 << 16 | (r1 - 1) << 24   (range [r0, r1) relative to the tile start)
  int32_t row;  // gain row of the segment's start point (ramp: end point = row + 1)
  float p0;     // ramp: libear's p at the tile start, (float)(s0 - start) * scale (gain_interpolator.hpp:272); else 0
  float scale;  // ramp: 1.0f / (float)(end - start); else 0
};
static_assert(sizeof(Piece) == 16, "Piece is loaded as one dwordx4");

constexpr int kPieceLists = 4;            // 2 * ramp + ranged
constexpr int kPieceCapPerObject = 8;     // capacity of a tile's lists: pieces per object on average (+ padding)
constexpr int kPieceMaxPerObject = 32;    // ranged pieces of ONE object in one tile; beyond: exact path
constexpr float kPieceMaxP0 = 7.0f;       // see above
constexpr int kPieceMaxTile = 256;        // r0, r1 - 1 are 8-bit fields
constexpr int kMaxPieceObjects = 1 << 16;

struct PieceLists {
  Piece *pieces;  // [ntiles][cap()]: the four lists back to back, each padded to a multiple of 32 with null pieces
  int *count;     // [ntiles][8]: CHUNKS (32 pieces) of list 0..3, [4] exact-path objects
  int *ovf;       // [ntiles][M]: objects that take the exact per-object path
  int *cw;        // [ntiles][M]: scratch of k_piece_list (piece counts per object)
  int M;
  __host__ __device__ int cap() const { return kPieceCapPerObject * M + 4 * 32; }
};
// 16-byte units of a buffer holding the descriptors (SegDesc[ntiles][M]) and, behind them, the piece lists
__host__ __device__ inline size_t piece_units(size_t M, size_t ntiles) {
  PieceLists pl;
  pl.M = (int)M;
  return M * ntiles + (size_t)pl.cap() * ntiles + (32 * ntiles + 8 * M * ntiles + 15) / 16 + 1;
}

// K0p: one workgroup per tile, behind k_seg_prep: turns the tile's descriptors (coalesced reads, no
// searching) into its piece lists, threads over objects.  A first pass counts the pieces of every
// object and list, which fixes where each list starts (they lie back to back, each padded to whole
// chunks); a second pass writes them at offsets from ordered scans over the objects: the lists are in
// object order, deterministic.
static __global__ void __launch_bounds__(256)
k_piece_list(PointStore ps, int M, int tile_samples, int64_t t_call, int64_t t_call_end, const SegDesc *desc,
             PieceLists pl) {
  __shared__ unsigned wsum[3][4];
  __shared__ int tot[kPieceLists], lbase[kPieceLists], run[kPieceLists], run_o, all_exact;
  const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t t0 = t_call + (int64_t)tile * tile_samples;
  int64_t t_end = t0 + tile_samples;
  if (t_end > t_call_end) t_end = t_call_end;
  Piece *lists = pl.pieces + (size_t)tile * pl.cap();
  int *ovf = pl.ovf + (size_t)tile * M;
  const SegDesc *dtile = desc + (size_t)tile * M;
  if (tid < kPieceLists) tot[tid] = 0, run[tid] = 0;
  if (tid == 0) run_o = 0;
  __syncthreads();
  Piece null_piece;
  null_piece.mr = 1u << 16;  // object 0, empty range [1, 1)
  null_piece.row = ps.zero_row;
  null_piece.p0 = 0.0f;
  null_piece.scale = 0.0f;

  // the pieces of object m inside the tile, first one described by dk (k_seg_prep), the others found by
  // walking on (GainInterpolator::process, gain_interpolator.hpp:58-86).  out == nullptr: count only.
  // Returns false when the object needs the exact path (a piece too steep to extend back to the tile
  // start, or too many pieces).
  auto walk = [&](int m, SegDesc dk, Piece *const *out, int (&cnt)[kPieceLists]) {
    int pbase = 0, n = 0;
    if (dk.info & kSegMulti) {
      pbase = ps.off[m];
      n = ps.off[m + 1] - pbase;
    }
    int k = seg_k(dk.info), cur = 0;
    bool ok = true;
#pragma unroll
    for (int l = 0; l < kPieceLists; l++) cnt[l] = 0;
    for (;;) {
      const int r1 = seg_r1(dk.info);
      if (r1 > cur) {  // duplicate times make empty segments (steps)
        const bool ramp = dk.info & kSegRamp;
        const bool whole = cur == 0 && r1 == tile_samples;
        Piece a;
        a.mr = (uint32_t)m | ((uint32_t)cur << 16) | ((uint32_t)(r1 - 1) << 24);
        a.row = dk.row;
        a.p0 = ramp ? (float)dk.d0 * dk.scale : 0.0f;  // p(s) = (float)(d0 + s) * scale, :272
        a.scale = ramp ? dk.scale : 0.0f;
        ok = ok && !(__builtin_fabsf(a.p0) > kPieceMaxP0);
        const int l = (ramp ? 2 : 0) + (whole ? 0 : 1);
        if (out) out[l][cnt[l]] = a;
        cnt[l]++;
        cur = r1;
      }
      if (!(dk.info & kSegMulti)) break;
      if (cnt[1] + cnt[3] > kPieceMaxPerObject) {
        ok = false;
        break;
      }
      k++;
      dk = describe_segment(ps, pbase, n, k, t0, t_end);
    }
    return ok;
  };
  // inclusive scan over the 256 threads of three packed counters; totals through `total`
  auto block_scan = [&](unsigned (&v)[3], unsigned (&total)[3]) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(v[j], o, 64);
        if (lane >= o) v[j] += u;
      }
      if (lane == 63) wsum[j][wv] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; j++) {
      unsigned pre = 0, t = 0;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const unsigned x = wsum[j][w];
        if (w < wv) pre += x;
        t += x;
      }
      total[j] = t;
      v[j] += pre;
    }
    __syncthreads();
  };
  // ---- pass 1: the piece counts of every object (kept for pass 2 in the tile's count words) and of every list
  int *cw = pl.cw + (size_t)tile * M;
  {
    int mine[kPieceLists] = {0, 0, 0, 0};
    for (int mb = 0; mb < M; mb += 256) {
      const int m = mb + tid;
      if (m < M) {
        int cnt[kPieceLists];
        // (objects the level probe found far below the call's level — kSegQuiet — take the exact path as well)
        const SegDesc d0 = dtile[m];
        const bool exact = !walk(m, d0, nullptr, cnt) || (d0.info & kSegQuiet);
        // plain counts <= 1, ranged ones <= kPieceMaxPerObject + 1
        cw[m] = exact ? -1 : (cnt[0] | (cnt[2] << 1) | (cnt[1] << 2) | (cnt[3] << 12));
        if (!exact)
#pragma unroll
          for (int l = 0; l < kPieceLists; l++) mine[l] += cnt[l];
      }
    }
#pragma unroll
    for (int l = 0; l < kPieceLists; l++) {
      int v = mine[l];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0 && v) atomicAdd(&tot[l], v);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int at = 0;
    for (int l = 0; l < kPieceLists; l++) {
      lbase[l] = at;
      at += (tot[l] + 31) & ~31;
    }
    // a tile with more pieces than its lists hold (more than kPieceCapPerObject per object on average):
    // only the whole-tile pieces are listed, every object with a curve point inside takes the exact path
    all_exact = at > pl.cap() ? 1 : 0;
    if (all_exact) {
      lbase[0] = 0;
      lbase[1] = lbase[2] = (tot[0] + 31) & ~31;
      lbase[3] = lbase[2] + ((tot[2] + 31) & ~31);
      tot[1] = tot[3] = 0;
    }
  }
  __syncthreads();
  const bool tile_over = all_exact != 0;

  // ---- pass 2: offsets from ordered scans of the counts, then the segment walk again, writing
  for (int mb = 0; mb < M; mb += 256) {
    const int m = mb + tid;
    int cnt[kPieceLists] = {0, 0, 0, 0};
    bool exact = false;
    SegDesc d;
    d.info = 0;
    if (m < M) {
      const int w = cw[m];
      d = dtile[m];
      exact = w < 0;
      if (!exact) {
        cnt[0] = w & 1;
        cnt[2] = (w >> 1) & 1;
        cnt[1] = (w >> 2) & 1023;
        cnt[3] = (w >> 12) & 1023;
        if (tile_over && cnt[1] + cnt[3] > 0) {
          exact = true;
          cnt[1] = cnt[3] = 0;
        }
      }
    }
    unsigned v[3] = {(unsigned)cnt[0] | ((unsigned)cnt[2] << 16), (unsigned)cnt[1] | ((unsigned)cnt[3] << 16), exact ? 1u : 0u};
    const unsigned own[3] = {v[0], v[1], v[2]};
    unsigned last[3];
    block_scan(v, last);
    if (m < M) {
      if (exact) {
        ovf[run_o + (int)(v[2] - own[2])] = m;
      } else {
        Piece *out[kPieceLists];
        out[0] = lists + lbase[0] + run[0] + (int)((v[0] - own[0]) & 0xffffu);
        out[2] = lists + lbase[2] + run[2] + (int)((v[0] - own[0]) >> 16);
        out[1] = lists + lbase[1] + run[1] + (int)((v[1] - own[1]) & 0xffffu);
        out[3] = lists + lbase[3] + run[3] + (int)((v[1] - own[1]) >> 16);
        int c2[kPieceLists];
        walk(m, d, out, c2);
      }
    }
    __syncthreads();
    if (tid == 0) {
      run[0] += (int)(last[0] & 0xffffu);
      run[2] += (int)(last[0] >> 16);
      run[1] += (int)(last[1] & 0xffffu);
      run[3] += (int)(last[1] >> 16);
      run_o += (int)last[2];
    }
    __syncthreads();
  }
  // pad every list to whole chunks with null pieces; publish the chunk counts
#pragma unroll
  for (int l = 0; l < kPieceLists; l++) {
    const int n = tot[l];
    const int padded = (n + 31) & ~31;
    if (n + tid < padded) lists[lbase[l] + n + tid] = null_piece;
    if (tid == 0) pl.count[tile * 8 + l] = padded >> 5;
  }
  if (tid == 0) pl.count[tile * 8 + 4] = run_o;
}

// K1p.  grid = (workgroup tiles, grid-level splits of the chunk schedule, column super-groups),
// block = 64 NW threads; P.ntiles / P.desc refer to WORKGROUP tiles of 64 NW samples.
// x_scale, g_scale: exact powers of two (gain_h2.h; g_scale here puts the largest gain at 2^12).
template <int NCT, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : 2)
k_gain_mix_p2(GainMixParams P, PieceLists pl, float x_scale, float g_scale, const unsigned *level_cur,
              unsigned *level_next) {
  constexpr int NRT = 4, TS = 16 * NRT, CH = kSplitChunk;
  constexpr int NQ = CH / NW;         // pieces whose gains one wave converts per chunk
  constexpr int NFRAG = 2 * NCT * 2;  // {B0,B1} x column tiles x {h,l}
  constexpr int RING = 8;             // chunks of piece words (object | range) staged in LDS for the lanes
  __shared__ u32x4 bfrag[2][NFRAG + 4][64];  // + 4 never-read fragments: the lanes without a column write there
  __shared__ uint32_t ring[RING][CH];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kg = lane >> 4;
  const int wgtile = xcd_tile(blockIdx.x, gridDim.x);
  if (level_cur) {  // input scale of THIS call from the level K0 probed (gain_h2.h)
    const unsigned lv = *level_cur;
    if (lv) {
      const int E = max(-60, min(20, (int)(lv >> 23) - 127));
      x_scale = __uint_as_float((unsigned)(127 + 7 - E) << 23);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *level_next = 0;
  }
  const int nparts = gridDim.y;
  const int part = blockIdx.y;
  const int col0 = blockIdx.z * 16 * NCT;
  const int wave_s0 = w * TS;                          // first sample of this wave inside the workgroup tile
  const int tile_s0 = wgtile * (TS * NW) + wave_s0;    // ... inside the call
  const int tile_len = max(0, min(TS, P.nsamples - tile_s0));
  const int64_t tile_t0 = P.t_call + tile_s0;
  const int64_t tile_t1 = tile_t0 + tile_len;
  const float *__restrict__ gain = P.ps.gain;
  const unsigned rowlen = (unsigned)P.ps.row;

  // running totals in scaled units: bus = (tot0 + (s - s0) tot1) / (x_scale g_scale)
  f32x4 tot0[NRT][NCT], tot1[NRT][NCT];
  auto clear_totals = [&]() {
#pragma unroll
    for (int r = 0; r < NRT; r++)
#pragma unroll
      for (int c = 0; c < NCT; c++) tot0[r][c] = tot1[r][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  };
  clear_totals();

  // ---- exact path: one object, all its pieces inside this wave's 64 samples, f32 MFMA with k = {a, b}
  // of ONE object (k slots 2, 3 idle), accumulated into tot0 in units of 1 / (sx sg)
  auto single_object = [&](int m, float sx, float sg) {
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
          a[r] = (x * coef) * sx;
        }
        const int grow = dk.row + ((ramp && is_b && slot0) ? 1 : 0);
        const float *gp = gain + (size_t)grow * rowlen + col0 + li;
#pragma unroll
        for (int c = 0; c < NCT; c++) gv[c] = gp[c * 16] * sg;
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

  // one piece of a list, exact and unscaled (the wave redoes its share of the schedule this way when an
  // input left the f16 range): p(s) = p0 + s * scale instead of libear's (float)(d0 + s) * scale — equal
  // to within an ulp of p, far inside the tolerance of this (non-strict) kernel
  auto single_piece = [&](const Piece pc) {
    if (tile_len <= 0) return;
    const int r0 = (int)((pc.mr >> 16) & 0xffu), r1 = (int)(pc.mr >> 24) + 1;
    const bool ramp = pc.scale != 0.0f;
    const float *row = P.in + (size_t)(pc.mr & 0xffffu) * P.in_stride + tile_s0;
    const bool is_b = kg & 1;
    const bool slot0 = kg < 2;
    float a[NRT], gv[NCT];
#pragma unroll
    for (int r = 0; r < NRT; r++) {
      const int s = li * NRT + r, sw = wave_s0 + s;
      const float x = row[min(s, tile_len - 1)];
      const float p = __builtin_fmaf((float)sw, pc.scale, pc.p0);
      const float coef = ramp ? (is_b ? p : 1.0f - p) : (is_b ? 0.0f : 1.0f);
      a[r] = (slot0 && sw >= r0 && sw < r1 && s < tile_len) ? x * coef : 0.0f;
    }
    const int grow = pc.row + ((ramp && is_b && slot0) ? 1 : 0);
    const float *gp = gain + (size_t)grow * rowlen + col0 + li;
#pragma unroll
    for (int c = 0; c < NCT; c++) gv[c] = gp[c * 16];
#pragma unroll
    for (int r = 0; r < NRT; r++)
#pragma unroll
      for (int c = 0; c < NCT; c++)
        tot0[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], gv[c], tot0[r][c], 0, 0, 0);
  };

  float inv_x = 1.0f / x_scale, inv_g = 1.0f / g_scale;  // exact: powers of two
  const int *cnt = pl.count + wgtile * 8;

  if (P.vec_ok) {
    // ---- the chunk schedule of this workgroup: the four lists back to back, 32 pieces per chunk
    const Piece *lbase = pl.pieces + (size_t)wgtile * pl.cap();
    const int n0 = cnt[0], n1 = cnt[1], n2 = cnt[2], n3 = cnt[3];
    const int total = n0 + n1 + n2 + n3;
    const int c_lo = (int)(((int64_t)total * part) / nparts), c_hi = (int)(((int64_t)total * (part + 1)) / nparts);
    const int first_ramp = n0 + n1;
    // first piece of chunk c (clamped: requests past the schedule re-read its last chunk)
    auto chunk_ptr = [&](int c) -> const Piece * { return lbase + 32 * min(c, total - 1); };
    auto is_ranged = [&](int c) { return (c >= n0 && c < first_ramp) || c >= first_ramp + n2; };

    if (c_hi > c_lo) {
      const int nvec = (P.nsamples + 3) & ~3;
      // byte offset of this lane's float4 inside an input row (lanes past the end of the call re-read the
      // last vector: never stored)
      const unsigned xlane = (unsigned)min(tile_s0 + li * NRT, nvec - 4) * 4u;
      const unsigned bcol_e = (unsigned)(col0 + min(lane, 16 * NCT - 1));  // the lane's gain column
      // fragment this lane fills: B0 pieces at bfr, bfr+1, B1 pieces NCT*2 further
      const int bfr = lane < 16 * NCT ? (lane >> 4) * 2 : NFRAG;
      const int bfr1 = lane < 16 * NCT ? NCT * 2 : 2;
      const size_t rstride = P.in_stride * sizeof(float);
      const int lane_s = wave_s0 + li * NRT;  // the lane's first sample inside the workgroup tile

      // piece words (object | range) of chunk c for the lanes: wave 0 brings them into the ring,
      // requested five chunks ahead, stored four ahead (visible after the next barrier), read two
      // ahead (input addresses) and at the chunk itself (ranges)
      auto ring_load = [&](int c) -> uint32_t {
        return chunk_ptr(c)[lane & 31].mr;
      };
      // the lane's piece words q0 .. q0 + N - 1 of chunk c (read where they are used: no registers held)
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      auto lane_word2 = [&](int c, int q0) { return *reinterpret_cast<const u32x2 *>(&ring[c & (RING - 1)][kg * 8 + q0]); };
      // inputs q0 .. q0 + n - 1 (n even) of chunk c.  A piece whose range misses this wave's 64 samples
      // contributes nothing here (its input scale is 0 for every lane): its request goes to object 0's
      // row instead — one line the L1 already holds, not another trip to L2 (a quarter of all input
      // requests on metadata that ignores the block grid).
      auto load_x_part = [&](int c, f32x4 (&x)[8], int q0, int n) {
        const char *bp = reinterpret_cast<const char *>(P.in) + xlane;
        const bool ranged = is_ranged(c);  // (wave-uniform)
#pragma unroll
        for (int q = 0; q < 8; q += 2)
          if (q >= q0 && q < q0 + n) {
            const u32x2 mw = lane_word2(c, q);
            uint32_t o0 = mw[0] & 0xffffu, o1 = mw[1] & 0xffffu;
            if (ranged) {
              o0 = (((mw[0] >> 16) & 0xffu) < (unsigned)(wave_s0 + TS) && (mw[0] >> 24) >= (unsigned)wave_s0) ? o0 : 0u;
              o1 = (((mw[1] >> 16) & 0xffu) < (unsigned)(wave_s0 + TS) && (mw[1] >> 24) >= (unsigned)wave_s0) ? o1 : 0u;
            }
            x[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(bp + (size_t)o0 * rstride));
            x[q + 1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(bp + (size_t)o1 * rstride));
          }
      };
      // What this WAVE converts for chunk c: pieces NQ w + q.  Wave-uniform: scalar loads (requested one
      // chunk ahead), scalar row arithmetic; the gain rows come in as (scalar row pointer) + (the lane's column).
      typedef const Piece __attribute__((address_space(4))) *ConstPiece;
      struct PieceGain {
        int32_t row;
        float p0, scale;
      };
      struct ChunkDesc {
        PieceGain d[NQ];
      };
      auto load_desc = [&](int c) {
        ConstPiece dp = (ConstPiece)(chunk_ptr(c) + w * NQ);
        ChunkDesc D;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          D.d[q].row = dp[q].row;
          D.d[q].p0 = dp[q].p0;
          D.d[q].scale = dp[q].scale;
        }
        return D;
      };
      auto load_gains = [&](const ChunkDesc &R, float (&S)[NQ], float (&E)[NQ], auto ramp_tag) {
        constexpr bool RAMP = decltype(ramp_tag)::value;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const float *rps = gain + (size_t)(unsigned)R.d[q].row * rowlen;
          S[q] = rps[bcol_e];
          if (RAMP) E[q] = rps[rowlen + bcol_e];
        }
      };
      // B0 (gain at the tile start, part 0) or B1 (slope, part 1) of the wave's NQ pieces,
      // scaled and split -> LDS (k = NQ w + q of the fragment entry of lane 16 (k / 8) + column)
      auto store_b = [&](const ChunkDesc &R, const float (&S)[NQ], const float (&E)[NQ], int buf, int bpart, auto ramp_tag) {
        constexpr bool RAMP = decltype(ramp_tag)::value;
        uint32_t h[NQ / 2], l[NQ / 2];
#pragma unroll
        for (int i = 0; i < NQ / 2; i++) {
          float v[2];
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const int q = 2 * i + j;
            const float p0 = R.d[q].p0;
            v[j] = (bpart == 0 ? (RAMP ? __builtin_fmaf(1.0f - p0, S[q], p0 * E[q]) : S[q]) : R.d[q].scale * (E[q] - S[q])) *
                   g_scale;
          }
          const uint32_t H = pack_f16(v[0], v[1]);
          h[i] = H;
          l[i] = pack_f16(v[0] - f16_lo(H), v[1] - f16_hi(H));  // residuals: exact in fp32
        }
        u32x4 *f = &bfrag[buf][bfr + (bpart ? bfr1 : 0)][0];
        const int col = lane & 15;
        if constexpr (NQ == 16) {  // two whole entries: k groups 2w, 2w + 1
          f[(2 * w) * 16 + col] = u32x4{h[0], h[1], h[2], h[3]};
          f[(2 * w + 1) * 16 + col] = u32x4{h[4], h[5], h[6], h[7]};
          f[64 + (2 * w) * 16 + col] = u32x4{l[0], l[1], l[2], l[3]};
          f[64 + (2 * w + 1) * 16 + col] = u32x4{l[4], l[5], l[6], l[7]};
        } else if constexpr (NQ == 8) {
          f[w * 16 + col] = u32x4{h[0], h[1], h[2], h[3]};
          f[64 + w * 16 + col] = u32x4{l[0], l[1], l[2], l[3]};
        } else {  // half an entry: words 2 (w & 1), + 1 of k group w / 2
          typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
          u32x2 *g = reinterpret_cast<u32x2 *>(f + (w >> 1) * 16 + col) + (w & 1);
          g[0] = u32x2{h[0], h[1]};
          g[128] = u32x2{l[0], l[1]};
        }
      };

      // ---- prologue: piece words of the first chunks into the ring, gains of the first chunk, inputs of
      // the first two
      uint32_t ring_next = 0;
      if (w == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint32_t v = ring_load(c_lo + j);
          if (lane < 32) ring[(c_lo + j) & (RING - 1)][lane] = v;
        }
        ring_next = ring_load(c_lo + 4);  // (stored by the first chunk)
      }
      __syncthreads();
      f32x4 X0[8], X1[8];
      ChunkDesc L;
      {
        float S[NQ], E[NQ];
        L = load_desc(c_lo);
        if (c_lo >= first_ramp) {
          load_gains(L, S, E, std::true_type{});
          load_x_part(c_lo, X0, 0, 8);
          load_x_part(c_lo + 1, X1, 0, 8);
          store_b(L, S, E, c_lo & 1, 0, std::true_type{});
          store_b(L, S, E, c_lo & 1, 1, std::true_type{});
        } else {
          load_gains(L, S, E, std::false_type{});
          load_x_part(c_lo, X0, 0, 8);
          load_x_part(c_lo + 1, X1, 0, 8);
          store_b(L, S, E, c_lo & 1, 0, std::false_type{});
        }
        L = load_desc(c_lo + 1);
      }

      // chunk c: inputs in xc, B fragments in bfrag[c & 1].  ONE body for all four lists (the list only
      // decides, wave-uniformly, whether the B1 half of the products runs and whether the inputs get a
      // range): six specialised bodies cost more in registers at their joins than they save.
      auto chunk = [&](int c, f32x4 (&xc)[8]) {
        const int buf = c & 1;
        const bool ramp = c >= first_ramp;                          // this chunk's pieces are ramps
        const bool next_ramp = min(c + 1, total - 1) >= first_ramp;  // (past the schedule: its last chunk again)
        __syncthreads();  // B fragments of chunk c are in bfrag[buf]; bfrag[buf^1] is free; ring slots <= c + 3 are visible
        if (w == 0) {
          if (lane < 32) ring[(c + 4) & (RING - 1)][lane] = ring_next;
          ring_next = ring_load(c + 5);
        }
        float S[NQ], E[NQ];
        const ChunkDesc Ln = L;  // chunk c + 1 (fetched one chunk ago)
        if (next_ramp) load_gains(Ln, S, E, std::true_type{});
        else load_gains(Ln, S, E, std::false_type{});
        L = load_desc(c + 2);
        __builtin_amdgcn_sched_barrier(0);  // every gain row is requested before any input

        // A fragments: row tile r = sample 4*li + r of the 8 pieces of this lane.  2 x 2 blocks: an f16
        // pair packs two PIECES (q, q+1) of one row tile, the scaling and the exact residual
        // subtractions pair two SAMPLES (r, r+1) of one piece.  The range of a piece is part of its
        // input scale: x_scale inside [r0, r1), 0 outside.
        u32x4 ah[NRT], al[NRT];
        auto split = [&](auto ranged_tag) {
          constexpr bool RANGED = decltype(ranged_tag)::value;
#pragma unroll
          for (int qp = 0; qp < 4; qp++) {
            f32x2 sc[2][2];  // [piece of the pair][sample pair]
            u32x2 mwp = {0u, 0u};
            if (RANGED) mwp = lane_word2(c, 2 * qp);
#pragma unroll
            for (int j = 0; j < 2; j++) {
              if (RANGED) {
                const uint32_t m = mwp[j];
                const int lo = (int)((m >> 16) & 0xffu) - lane_s;          // r0 relative to the lane's first sample
                const unsigned len = (m >> 24) + 1u - ((m >> 16) & 0xffu);  // r1 - r0
#pragma unroll
                for (int rp = 0; rp < 2; rp++)
                  sc[j][rp] = f32x2{(unsigned)(2 * rp - lo) < len ? x_scale : 0.0f,
                                    (unsigned)(2 * rp + 1 - lo) < len ? x_scale : 0.0f};
              } else {
                sc[j][0] = sc[j][1] = f32x2{x_scale, x_scale};
              }
            }
#pragma unroll
            for (int rp = 0; rp < NRT; rp += 2) {
              const f32x2 s0 = f32x2{xc[2 * qp][rp], xc[2 * qp][rp + 1]} * sc[0][rp >> 1];          // piece 2qp
              const f32x2 s1 = f32x2{xc[2 * qp + 1][rp], xc[2 * qp + 1][rp + 1]} * sc[1][rp >> 1];  // piece 2qp+1
              const uint32_t H0 = pack_f16(s0[0], s1[0]), H1 = pack_f16(s0[1], s1[1]);
              const f32x2 r0 = s0 - f32x2{f16_lo(H0), f16_lo(H1)};  // exact
              const f32x2 r1 = s1 - f32x2{f16_hi(H0), f16_hi(H1)};
              ah[rp][qp] = H0;
              ah[rp + 1][qp] = H1;
              al[rp][qp] = pack_f16(r0[0], r1[0]);
              al[rp + 1][qp] = pack_f16(r0[1], r1[1]);
            }
          }
        };
        if (is_ranged(c)) split(std::true_type{});  // (wave-uniform)
        else split(std::false_type{});
        // The B0 half: NCT blocks of 12 MFMAs (three partial products of one column tile, smallest first).
        // The inputs of chunk c + 2 go into the registers just freed, a few requests per block; the
        // conversion of the next chunk's B0 is woven between the MFMAs of the last block.
        __builtin_amdgcn_sched_barrier(0);  // the splitting above stays above
        constexpr int XB = NCT >= 3 ? 2 : 1;  // blocks that carry input requests (4 or 8 each)
        auto load_b = [&](int fr, u32x4 (&bb)[2]) {
#pragma unroll
          for (int q = 0; q < 2; q++) bb[q] = bfrag[buf][fr + q][lane];
        };
        u32x4 b[2][2];
        load_b(0, b[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) {
          u32x4(&bc)[2] = b[ct & 1];
          if (ct + 1 < NCT) load_b((ct + 1) * 2, b[(ct + 1) & 1]);
#pragma unroll
          for (int r = 0; r < NRT; r++) tot0[r][ct] = mfma_f16(al[r], bc[0], tot0[r][ct]);
#pragma unroll
          for (int r = 0; r < NRT; r++) tot0[r][ct] = mfma_f16(ah[r], bc[1], tot0[r][ct]);
#pragma unroll
          for (int r = 0; r < NRT; r++) tot0[r][ct] = mfma_f16(ah[r], bc[0], tot0[r][ct]);
          if (ct < XB) load_x_part(c + 2, xc, ct * (8 / XB), 8 / XB);
          const bool conv = ct == NCT - 1;
          if (conv) {
            if (next_ramp) store_b(Ln, S, E, buf ^ 1, 0, std::true_type{});
            else store_b(Ln, S, E, buf ^ 1, 0, std::false_type{});
          }
          if (ct + 1 < NCT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          if (ct < XB && !conv) {
#pragma unroll
            for (int k = 0; k < 8 / XB; k++) {  // MFMAs, then one request (address arithmetic + load)
              __builtin_amdgcn_sched_group_barrier(0x008, 12 / (8 / XB), 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
          }
        }
        // The B1 half (ramp pieces only) with the conversion of the next chunk's B1
        if (ramp) {
          load_b(NCT * 2, b[0]);
#pragma unroll
          for (int ct = 0; ct < NCT; ct++) {
            u32x4(&bc)[2] = b[ct & 1];
            if (ct + 1 < NCT) load_b((NCT + ct + 1) * 2, b[(ct + 1) & 1]);
#pragma unroll
            for (int r = 0; r < NRT; r++) tot1[r][ct] = mfma_f16(al[r], bc[0], tot1[r][ct]);
#pragma unroll
            for (int r = 0; r < NRT; r++) tot1[r][ct] = mfma_f16(ah[r], bc[1], tot1[r][ct]);
#pragma unroll
            for (int r = 0; r < NRT; r++) tot1[r][ct] = mfma_f16(ah[r], bc[0], tot1[r][ct]);
            if (ct == NCT - 1) store_b(Ln, S, E, buf ^ 1, 1, std::true_type{});
          }
        } else if (next_ramp) {  // (the one chunk of a tile where the lists change from constant to ramp)
          store_b(Ln, S, E, buf ^ 1, 1, std::true_type{});
        }
      };
      auto run_chunk = [&](int c, f32x4 (&xc)[8]) { chunk(c, xc); };
#pragma unroll 1
      for (int c = c_lo; c < c_hi; c += 2) {
        run_chunk(c, X0);
        if (c + 1 < c_hi) run_chunk(c + 1, X1);
      }
    }

    // objects that take the exact path (pieces too steep or too many for the lists): part 0 only
    if (part == 0) {
      const int *ovf = pl.ovf + (size_t)wgtile * P.M;
      const int novf = cnt[4];
      for (int i = 0; i < novf; i++) single_object(ovf[i], x_scale, g_scale);
    }
    // an input beyond the f16 range (or not finite) shows as non-finite totals: redo the wave's tile
    // exactly, unscaled (every part redoes its share of the objects)
    bool bad = false;
#pragma unroll
    for (int r = 0; r < NRT; r++)
#pragma unroll
      for (int c = 0; c < NCT; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) bad |= !(__builtin_fabsf(tot0[r][c][e]) < INFINITY) || !(__builtin_fabsf(tot1[r][c][e]) < INFINITY);
    if (__ballot(bad)) {
      clear_totals();
      inv_x = inv_g = 1.0f;
      for (int c = c_lo; c < c_hi; c++)
        for (int j = 0; j < CH; j++) single_piece(chunk_ptr(c)[j]);
      if (part == 0) {
        const int *ovf = pl.ovf + (size_t)wgtile * P.M;
        for (int i = 0; i < cnt[4]; i++) single_object(ovf[i], 1.0f, 1.0f);
      }
    }
  } else {
    inv_x = inv_g = 1.0f;
    const int m_lo = (int)(((int64_t)P.M * part) / nparts), m_hi = (int)(((int64_t)P.M * (part + 1)) / nparts);
    for (int m = m_lo; m < m_hi; m++) single_object(m, 1.0f, 1.0f);  // unaligned rows
  }

  if (tile_len <= 0) return;
  // D fragment of row tile r: rows 4kg + e = samples 16kg + 4e + r: for fixed e the
  // four row tiles are 4 consecutive samples.  (s - s0) of the rows: sample 64w + 16kg + 4e + r.
  const float wf0 = (float)(wave_s0 + kg * 16);
  float *op = P.out + (size_t)blockIdx.y * P.part_stride + tile_s0;
#pragma unroll
  for (int c = 0; c < NCT; c++) {
    const int col = col0 + c * 16 + li;
    if (col >= P.ncols) continue;
    float *o = op + (size_t)col * P.out_stride;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int s = kg * 16 + e * 4;
      f32x4 v;
#pragma unroll
      for (int r = 0; r < NRT; r++)
        v[r] = (__builtin_fmaf(wf0 + (float)(4 * e + r), tot1[r][c][e], tot0[r][c][e]) * inv_x) * inv_g;
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
