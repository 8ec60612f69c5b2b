// gain_p2.h — K1 on the f16 matrix cores over per-tile PIECE lists ("f16x2 pieces"), for gain curves whose
// points ignore the tile grid (ADM blocks at arbitrary times).
//
// A gain curve is continuous and piecewise linear through its points (libear: GainInterpolator::process,
// include/ear/dsp/gain_interpolator.hpp:58-86; constant before the first and after the last point, :68-75).
// Inside a tile that starts at sample s0 it is therefore its value at the tile start plus what every ramp
// that overlaps the tile has added so far:
//
//     g(s) = S_base + sum_k clamp(p_k(s), 0, 1) * (E_k - S_k),     p_k(s) = p0_k + (s - s0) * scale_k
//
// with S_base the START row of the segment the tile starts in (its own ramp, if it is one, is the first
// term of the sum: p0 >= 0), p_k libear's ramp position (:272) of segment k with (S_k, E_k) its two gain
// rows, and the clamp holding a ramp at 0 before it starts and at 1 after it ends — flat segments add
// nothing.  Two equal times with different gains (a step) are a ramp of length one: p = s - (r - 1).
// So the contraction over a tile runs over PIECES, the k dimension of the MFMA:
//
//     bus[col][s] = sum_{base pieces}  x_m(s)             * S_row,col
//                 + sum_{delta pieces} x_m(s) clamp(p(s)) * (E - S)_row,col
//
// one base piece per object and one delta piece per (object, ramp overlapping the tile) — no sample ranges,
// no alignment rule, no cliff: a call costs what its curve points cost.  The position factor is not linear
// in s (the clamp), so it goes to the INPUT side: a delta chunk scales its inputs by clamp(p) before the
// operand split; both kinds of chunk are then the same 36 MFMAs per 32 pieces, 64 samples and 48 columns on
// one set of accumulators.  (gain_h2.h keeps the ramp on the gain side and needs a second set; it is the
// kernel for curves ON the tile grid, where it has half the pieces.)  Block-aligned ramps cost a base and a
// delta piece per object here, static gains one, ADM-like metadata (interpolate 5 ms, then hold) 1 + the
// fraction of tiles a ramp touches.
//
// Vector-memory instructions are what this kernel is short of (the CU's address unit is busy two thirds of all
// cycles, profiles/r02_adm_scene_*): per chunk a wave issues its 4-8 input requests, 2-3 sixteen-byte requests for
// the gain rows of the pieces it converts (staged through wave-private LDS to the lanes that need them, instead
// of one 4-byte gather per row) and nothing else — piece words, row indices and ramp positions all come out of
// an LDS ring that wave 0 fills with one request per chunk.  And latency: a chunk of 36 MFMAs is shorter than a
// loaded trip to L2 or to memory, so gain rows are requested two chunks ahead of their conversion (three of their
// use) and inputs two chunks (pair chunks, which carry half the bytes: four chunks) ahead.
//
// ONE list per tile, written by k_piece_build in object order (deterministic), padded to whole chunks of 32 with
// null pieces (object 0, the all-zero gain row), in one of two layouts (k_piece_build; CurveSet::pair_waste picks):
//   packed  every object's base piece followed by its delta pieces.  The pieces of an object are neighbours in k:
//           the repeats of its input request hit in the first-level cache.  All pieces go through the same
//           arithmetic — a base piece is a delta piece with p = 1 (p0 = 1, scale = 0) whose gain operand is S - 0
//           instead of E - S; a tile without any delta piece (static gains) skips the position factors.  For curves
//           that have several ramps of one object inside a tile (always-ramping metadata).
//   paired  the objects without a ramp in the tile in front (SINGLE chunks: 32 base pieces, no position factors),
//           the others behind them as pairs of slots (base piece, delta piece) on ONE input request (PAIR chunks: 16
//           objects, 4 requests per lane, position factors for the odd slots only).  For ADM-like metadata
//           (interpolate for a while, then hold): K1 0.50 -> 0.45-0.46 ms on the ADM scene.
// Either way a call reads its inputs from memory once however many curve points it has.  Objects with more than
// kPieceMaxPerObject ramps in one tile (paired: kPairMaxPerObject) and objects the level probe found far below the
// call's level (kSegQuiet) take the exact per-object path.
//
// Operand scaling and splitting as in gain_h2.h (x by the probed power of two, gains by 2^14 / the curve
// set's largest gain: |E - S| <= 2^15 stays inside the f16 range).
#pragma once

#include <hip/hip_runtime.h>

#include "gain_h2.h"

#ifndef EARHIP_P2_ABL
#define EARHIP_P2_ABL 0  // (timing-only ablations, NOTES round 5: 1 no input requests, 2 no gain-row requests inside the chunk loop)
#endif

namespace earhip {

struct Piece {
  uint32_t m;   // object | kPieceDelta
  int32_t row;  // base: the gain row; delta: the start row S of the ramp (end row E = row + 1)
  float p0;     // delta: libear's p at the tile start, (float)(s0 - start) * scale (gain_interpolator.hpp:272); base: 1
  float scale;  // delta: 1.0f / (float)(end - start); base: 0
};
constexpr uint32_t kPieceDelta = 1u << 31;
static_assert(sizeof(Piece) == 16, "Piece is loaded as one dwordx4");

// Packed chunks add their products to the totals once per chunk (k_gain_mix_p2): always-ramping scene, worst channel
// against the CPU path over 8 seeds 9.9e-7 -> 7.2e-7 (the kernel's own distance from a float64 render 6.8e-7 -> 2.8e-7),
// for 24 more VALU instructions per chunk (K1 0.835 -> 0.886 ms).
constexpr bool kChunkSums = true;
constexpr int kPieceMaxPerObject = 15;    // delta pieces of ONE object in one tile (a curve point every 17 samples at 256-sample
                                          // tiles); beyond: the exact path
constexpr int kPieceCapPerObject = 1 + kPieceMaxPerObject;  // most slots of a tile's list per object (the lists are sized from the
                                                            // curves: CurveSet::piece_cap — a list cannot overflow)
constexpr int kPieceMaxTile = 512;
constexpr int kMaxPieceObjects = 1 << 16;

struct PieceLists {
  Piece *pieces;  // [ntiles][cap()]: the tile's pieces in object order, padded to a multiple of 32 with null pieces
  int *count;     // [ntiles][8]: [0] CHUNKS (32 pieces) of the list, [1] delta pieces in it, [2] the SINGLE chunks among
                  // them (the first ones), [4] exact-path objects
  int *ovf;       // [ntiles][M]: objects that take the exact per-object path
  int M;
  int paired;     // layout of a tile's list (see k_gain_mix_p2): 0 packed, 1 singles + pairs
  int slots;      // slots of a tile's list: what the curves can need (CurveSet::piece_cap), at most kPieceCapPerObject M + 128
  __host__ __device__ int cap() const { return slots; }
  // paired layout: the pairs start here (the singles — at most one per object, padded to a chunk — lie in front)
  __host__ __device__ int pair_off() const { return (M + 63) & ~63; }
};
// paired layout: pairs of ONE object in one tile (2 slots each: with the singles' region they fit cap()); beyond: the
// exact path
constexpr int kPairMaxPerObject = 7;
static_assert(2 * kPairMaxPerObject + 1 <= kPieceCapPerObject, "singles + pairs fit the tile's list");
// 16-byte units of a buffer holding the piece lists of `ntiles` tiles (slots each), their counts and exact-path lists
__host__ __device__ inline size_t piece_units(size_t M, size_t ntiles, size_t slots) {
  return slots * ntiles + (32 * ntiles + 4 * M * ntiles + 15) / 16 + 1;
}

// ---------------------------------------------------------------------------
// K0: the piece lists in ONE pass (round 3; replaces k_seg_prep + k_mark_quiet + k_piece_list of round 2).
// k_seg_prep found every (object, tile)'s segment and counted its ramps with a thread per object and run of 2-4
// tiles, wrote 16-byte descriptors and counts for all of them, and k_piece_list — a 256-thread workgroup per tile —
// read those back, scanned the counts and walked the curve points of every ramping object a second time, one
// dependent load after the other: 0.086 ms per call on ADM-like metadata, 0.23 ms when every object ramps all
// the time (a fifth of the step).  Here a workgroup of 1024 threads owns TPW consecutive tiles and all objects
// (in batches of 1024): a thread searches its object's segment ONCE (at the first tile), walks on through the
// TPW tiles counting ramps, the workgroup scans the counts of all TPW tiles together (one pair of barriers), and
// the thread walks the same points again — cache hits now — writing its pieces at the scanned offsets.  No
// descriptors, no counts in memory, one launch; the lists are exactly the ones k_piece_list wrote.
//
// the ramps of one object inside the tile [t0, t_end) that starts in segment k (k = number of the object's points
// with time <= t0): returns their number, or -1 beyond kPieceMaxPerObject; out != nullptr: writes the delta
// pieces.  k is left at the segment that reaches the end of the tile (where the next tile's search starts).
constexpr int kKeepPieces = 3;
// keep != nullptr: the first kKeepPieces delta pieces are also left there (registers: indexed by constants only)
// (ostride: distance of the pieces written to out — 2 fills the odd slots of a run of pairs)
__device__ __forceinline__ int piece_walk(const PointStore &ps, int base, int n, int &k, int64_t t0, int64_t t_end, int m,
                                          Piece *out, Piece *keep = nullptr, int ostride = 1) {
  const int tile_len = (int)(t_end - t0);
  int cur = 0, nd = 0;
  if (!ps.force_ramp) {
    // The same walk over the packed point records: ONE 16-byte load per segment (its end time, 1 / length and flat
    // bits; its start time is the previous record's) instead of describe_segment's two dependent rounds of loads.
    const int allflat = (1 << ps.nbus) - 1;
    const PointRec *rec = ps.rec + base;
    int64_t prev = k > 0 ? rec[k - 1].time : 0;  // (k == 0: before the first point, never a ramp)
    for (;;) {
      PointRec r;
      r.time = t_end;
      r.scale = 0.0f;
      r.flat = (uint32_t)allflat;
      if (k < n) r = rec[k];
      const bool ramp = k > 0 && k < n && ((int)r.flat & allflat) != allflat;
      const bool multi = k < n && r.time < t_end;
      const int r1 = multi ? (int)(r.time - t0) : tile_len;
      if (ramp) {
        Piece a;
        a.m = (uint32_t)m | kPieceDelta;
        a.row = base + k - 1;
        bool emit = true;
        if (r1 > cur) {
          a.p0 = (float)(int32_t)(t0 - prev) * r.scale;
          a.scale = r.scale;
        } else {
          a.p0 = (float)(1 - cur);
          a.scale = 1.0f;
          emit = cur > 0;
        }
        if (emit) {
          if (out) out[nd * ostride] = a;
          if (keep) {
            if (nd == 0) keep[0] = a;
            if (nd == 1) keep[1] = a;
            if (nd == 2) keep[2] = a;
          }
          nd++;
        }
      }
      if (r1 > cur) cur = r1;
      if (!multi) break;
      if (nd > kPieceMaxPerObject) return -1;
      prev = r.time;
      k++;
    }
    return nd > kPieceMaxPerObject ? -1 : nd;
  }
  for (;;) {
    const SegDesc dk = describe_segment(ps, base, n, k, t0, t_end);
    const int r1 = (dk.info & kSegMulti) ? seg_r1(dk.info) : tile_len;
    if (dk.info & kSegRamp) {
      Piece a;
      a.m = (uint32_t)m | kPieceDelta;
      a.row = dk.row;
      bool emit = true;
      if (r1 > cur) {  // a ramp over [cur, r1) of the tile (and beyond): its own line, clamped
        a.p0 = (float)dk.d0 * dk.scale;  // p(s) = (float)(d0 + s) * scale, gain_interpolator.hpp:272
        a.scale = dk.scale;
      } else {  // two equal times with different gains: a step at cur = a ramp from cur - 1 to cur
        a.p0 = (float)(1 - cur);
        a.scale = 1.0f;
        emit = cur > 0;  // (at the tile start the base piece already has the value after the step)
      }
      if (emit) {
        if (out) out[nd * ostride] = a;
        if (keep) {
          if (nd == 0) keep[0] = a;
          if (nd == 1) keep[1] = a;
          if (nd == 2) keep[2] = a;
        }
        nd++;
      }
    }
    if (r1 > cur) cur = r1;
    if (!(dk.info & kSegMulti)) break;
    if (nd > kPieceMaxPerObject) return -1;
    k++;
  }
  return nd > kPieceMaxPerObject ? -1 : nd;
}

// ---------------------------------------------------------------------------
// Round 6: K0 of the list kernels in TWO kernels.  The one-pass builders below do a search and a walk of the curve per (object,
// tile) pair with the lanes of a wave on 16 different objects: every load instruction of theirs asks for 16-32 different cache
// lines, and ~3 M such line requests are what those kernels' 41-58 us are made of (the CUs' path to L2 passes ~50 requests
// per ns, chip-wide: DESIGN 4) — searches and walks are 70 % of a workgroup's time (tools/build_phases.py).
//   classify (object-major): a WAVE = one object x 64 consecutive tiles.  The lanes search neighbouring points of ONE curve
//     (the header is a scalar load, the window of records and the walk behind it hit the same few lines for the whole wave),
//     and what the placing kernel needs of the pair goes, 16 bytes, into a staging matrix [tile][object] — through LDS, so that
//     a tile's row of 8 objects leaves as one 128-byte line;
//   place (tile-major): the one-pass builder with the search and the walk replaced by ONE coalesced 16-byte load per pair: the
//     scan and the stores are its own.  Pairs with two and more delta pieces (rare where piece lists are used) walk the curve
//     there, from the segment index the classify kernel left.
// The lists are the one-pass builder's, bit for bit (option BUILD_2K = 0 runs that one; tests compare both).
// (Barriers that wait for the waves' LDS operations only — not, like __syncthreads(), for their list stores as well — were tried in
// both builders' batch loops: level.)
struct PairRec {
  int32_t row0;     // the base piece's gain row
  uint32_t info;    // bits 0..7: delta pieces of the object in the tile (kPairExact: it takes the exact path there); bits 8..: the
                    // first delta piece's row - row0
  float p0, scale;  // the first delta piece's; with two and more delta pieces p0 holds the tile's first segment index (its bits)
};
static_assert(sizeof(PairRec) == 16, "a pair of the staging matrix is one dwordx4");
constexpr uint32_t kPairExact = 0xffu;
constexpr int kClassifyObjects = 8;  // objects (= waves) of a classify workgroup: a tile's row of them is one 128-byte line

// -DEARHIP_BUILD_PROF: thread 0 of two workgroups of a list builder (the first and one from the middle of the grid) leaves
// s_memtime stamps at the builder's phase boundaries (earhip_debug_build_prof reads them; tools/build_phases.py)
#ifdef EARHIP_BUILD_PROF
static __device__ unsigned long long g_build_prof[2][32];
#define EARHIP_BUILD_MARK(i)                                                                                 \
  do {                                                                                                       \
    if (prof_slot >= 0 && threadIdx.x == 0 && (i) < 32) g_build_prof[prof_slot][(i)] = __builtin_readcyclecounter(); \
  } while (0)
#define EARHIP_BUILD_PROF_SLOT const int prof_slot = blockIdx.x == 0 ? 0 : blockIdx.x == gridDim.x / 2 ? 1 : -1; int prof_i = 1;
#else
#define EARHIP_BUILD_MARK(i) do {} while (0)
#define EARHIP_BUILD_PROF_SLOT
#endif
// grid = (ceil(ntiles / 64), ceil(M / kClassifyObjects)), block = 64 kClassifyObjects threads: wave w = object blockIdx.y * 8 + w,
// lane = tile blockIdx.x * 64 + lane
static __global__ void __launch_bounds__(64 * kClassifyObjects)
k_piece_classify(PointStore ps, int M, int ntiles, int tile_samples, int64_t t_call, int64_t t_call_end, int paired, PairRec *stage,
                 const unsigned *obj_level, const unsigned *level_cur, const unsigned *gate) {
  if (gate && !(*gate & kGateHingeUnsafe)) return;  // launched behind the hinge kernel's builder, which does this call (k_hinge_gate)
  __shared__ __attribute__((aligned(16))) PairRec sh[64][kClassifyObjects + 1];  // (+ 1: the lanes' rows on different banks)
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = blockIdx.y * kClassifyObjects + w;  // wave-uniform
  const int tile = blockIdx.x * 64 + lane;
  PairRec pr;
  pr.row0 = ps.zero_row, pr.info = 0u, pr.p0 = 1.0f, pr.scale = 0.0f;
  if (m < M && tile < ntiles) {
    const unsigned call_level = level_cur ? *level_cur : 0u;
    const ObjHdr hd = ps.hdr[m];
    const int base = hd.off, n = hd.cnt;
    const int64_t t0 = t_call + (int64_t)tile * tile_samples;
    const int64_t t1 = t0 + tile_samples > t_call_end ? t_call_end : t0 + tile_samples;
    const int kst = ps.force_ramp ? upper_bound_time_window(ps.time + base, n, t0) : upper_bound_rec_window(ps.rec + base, n, hd.first, hd.last, t0);
    int k = kst;
    Piece keep[kKeepPieces];
    int cnt = piece_walk(ps, base, n, k, t0, t1, m, nullptr, keep);
    if (paired && cnt > kPairMaxPerObject) cnt = -1;
    if (obj_level && level_is_quiet(obj_level[m], call_level)) cnt = -1;
    // the row of the segment the tile starts in (k_piece_build's own rule)
    int row;
    if (ps.force_ramp) {
      row = describe_segment(ps, base, n, kst, t0, t1).row;
    } else {
      const bool ramp = kst > 0 && kst < n && ((int)ps.rec[base + kst].flat & ((1 << ps.nbus) - 1)) != (1 << ps.nbus) - 1;
      row = base + (ramp ? kst - 1 : kst == n ? kst - 1 : kst);
    }
    pr.row0 = row;
    if (cnt < 0) {
      pr.info = kPairExact;
    } else if (cnt == 1) {
      pr.info = 1u | (uint32_t)(keep[0].row - row) << 8;
      pr.p0 = keep[0].p0, pr.scale = keep[0].scale;
    } else if (cnt >= 2) {
      pr.info = (uint32_t)cnt;
      pr.p0 = __int_as_float(kst);
    }
  }
  sh[lane][w] = pr;
  __syncthreads();
  // a tile's row of this workgroup's objects: 8 x 16 bytes = one line
  const int to = threadIdx.x / kClassifyObjects, oo = threadIdx.x % kClassifyObjects;
  const int tile_o = blockIdx.x * 64 + to, m_o = blockIdx.y * kClassifyObjects + oo;
  if (tile_o < ntiles && m_o < M) stage[(size_t)tile_o * M + m_o] = sh[to][oo];
}

constexpr int kBuildThreads = 1024;
// grid = ceil(ntiles / TPW) workgroups of 1024 threads; a thread = one (object, tile) pair of a batch of 1024 / TPW
// objects, the TILE index fastest: the TPW lanes of an object read neighbouring points of its curve (the same
// cache lines), like k_seg_prep's lanes did — with the object index fastest every load instruction touched 64
// different lines and the walk was four times slower than the three kernels it replaces.
//
// Two layouts of a tile's list (pl.paired, chosen per call from the curves: CurveSet::pair_waste):
//   packed  every object's base piece followed by its delta pieces, all in one run;
//   paired  the objects WITHOUT a delta piece in this tile in front ("singles": one base piece each, padded to whole
//           chunks), behind them (from pl.pair_off()) the others as PAIRS of slots (even slot: the base piece — or,
//           from the object's second pair on, a piece with the all-zero gain row —, odd slot: a delta piece), padded to
//           whole chunks: the two slots of a pair take the same input, which the kernel then requests ONCE.
// The scan carries three counters in one word: pieces (packed) or singles (paired), pairs, exact-path objects.
template <int TPW>
__global__ void __launch_bounds__(kBuildThreads)
k_piece_build(PointStore ps, int M, int ntiles, int tile_samples, int64_t t_call, int64_t t_call_end, PieceLists pl,
              const unsigned *obj_level, const unsigned *level_cur, const unsigned *gate = nullptr, int level_cap = 0,
              unsigned *wide = nullptr, const PairRec *stage = nullptr) {
  // stage != nullptr: the placing kernel of the two-kernel K0 — what a pair is comes from k_piece_classify's matrix
  if (gate && !(*gate & kGateHingeUnsafe)) return;  // launched behind the hinge kernel's builder, which does this call (k_hinge_gate)
  // the form of k_gain_mix_p2 this call needs (gain_h2.h, k_seg_prep): wide when some object falls more than kPlainBinades
  // below the call's level at some probed instant
  if (wide && obj_level && blockIdx.x == 0) {
    const unsigned call = level_cur ? *level_cur : 0u;
    bool w = false;
    for (int m = threadIdx.x; m < M; m += kBuildThreads) {
      const unsigned lo = obj_level[level_cap + m];
      w |= lo != 0u && call != 0u && (int)(lo >> 23) < (int)(call >> 23) - kPlainBinades;
    }
    if (w) atomicOr(wide, 1u);
  }
  constexpr int OB = kBuildThreads / TPW;  // objects per batch
  constexpr int WPT = OB / 64;             // waves that scan one tile's objects
  static_assert(OB % 64 == 0, "a tile's objects of a batch are whole waves of the scan");
  // (a batch of 128 objects — eight tiles per workgroup, the long calls — has at most 2048 pieces, 896 pairs and 128
  // exact-path objects: the counters fit 32 bits, and the scan's shuffles move half the words)
  constexpr bool kNarrow = OB <= 128;
  typedef typename std::conditional<kNarrow, unsigned, unsigned long long>::type u64;
  constexpr int kShB = kNarrow ? 12 : 20, kShC = kNarrow ? 23 : 44;
  constexpr u64 kOne = 1, kPairUnit = (u64)1 << kShB, kExactUnit = (u64)1 << kShC;
  constexpr u64 kMaskA = ((u64)1 << kShB) - 1, kMaskB = ((u64)1 << (kShC - kShB)) - 1;
  static_assert(kPieceCapPerObject * OB <= (int)kMaskA && kPairMaxPerObject * OB <= (int)kMaskB, "the scan's counters hold a batch");
  __shared__ u64 lcnt[kBuildThreads];   // [tile][object of the batch]: the three counters; then their scan
  __shared__ u64 wtot[kBuildThreads / 64];
  __shared__ int run[TPW][3];           // pieces / singles, pairs and exact-path objects of the batches so far
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = tid % TPW, oi = tid / TPW;   // this thread's tile of the workgroup and object of the batch
  const int tile = blockIdx.x * TPW + j;
  const bool paired = pl.paired != 0;
  const unsigned call_level = level_cur ? *level_cur : 0u;
  if (tid < TPW) run[tid][0] = run[tid][1] = run[tid][2] = 0;
  EARHIP_BUILD_PROF_SLOT
  EARHIP_BUILD_MARK(0);
  const int64_t t0 = t_call + (int64_t)tile * tile_samples;
  const int64_t t1 = t0 + tile_samples > t_call_end ? t_call_end : t0 + tile_samples;
  __syncthreads();
  for (int mb = 0; mb < M; mb += OB) {
    const int m = mb + oi;
    const bool in = m < M && tile < ntiles;
    int base = 0, n = 0, kst = 0, cnt = 0;
    Piece keep[kKeepPieces];
    PairRec pr;
    pr.row0 = 0, pr.info = 0u, pr.p0 = 0.0f, pr.scale = 0.0f;
    if (in && stage) {
      pr = stage[(size_t)tile * M + m];
      const uint32_t c = pr.info & 0xffu;
      cnt = c == kPairExact ? -1 : (int)c;
      keep[0].m = (uint32_t)m | kPieceDelta;
      keep[0].row = pr.row0 + (int32_t)(pr.info >> 8);
      keep[0].p0 = pr.p0, keep[0].scale = pr.scale;
    } else if (in) {
      const ObjHdr hd = ps.hdr[m];  // (offset, count and the curve's end points in one load; the search on the records themselves:
      base = hd.off;                 // the walk below then reads lines the search has just brought in)
      n = hd.cnt;
      kst = ps.force_ramp ? upper_bound_time_window(ps.time + base, n, t0) : upper_bound_rec_window(ps.rec + base, n, hd.first, hd.last, t0);
      int k = kst;
      cnt = piece_walk(ps, base, n, k, t0, t1, m, nullptr, keep);
      if (paired && cnt > kPairMaxPerObject) cnt = -1;
      if (obj_level && level_is_quiet(obj_level[m], call_level)) cnt = -1;
    }
#ifdef EARHIP_BUILD_PROF
    EARHIP_BUILD_MARK(prof_i);  // search + counting walk of this thread done
#endif
    // ---- ordered scan over the batch's objects, per tile: element e = tile * OB + object, a tile = WPT whole waves
    lcnt[j * OB + oi] = !in ? (u64)0 : cnt < 0 ? kExactUnit : !paired ? kOne * (u64)(1 + cnt) : cnt == 0 ? kOne : kPairUnit * (u64)cnt;
    __syncthreads();
    u64 v = lcnt[tid];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const u64 u = __shfl_up(v, o, 64);
      if (lane >= o) v += u;
    }
    if (lane == 63) wtot[wv] = v;
    __syncthreads();
    {
      u64 pre = 0;
      const int w0 = wv - wv % WPT;  // first wave of this tile's segment
#pragma unroll
      for (int w = 0; w < WPT; w++) pre += w0 + w < wv ? wtot[w0 + w] : (u64)0;
      lcnt[tid] = v + pre;  // inclusive over the tile's objects of the batch
    }
    __syncthreads();
#ifdef EARHIP_BUILD_PROF
    EARHIP_BUILD_MARK(prof_i + 1);  // scan done (three barriers)
#endif
    // ---- the pieces, at the scanned offsets
    if (in) {
      const u64 incl = lcnt[j * OB + oi];
      const int iA = (int)(incl & kMaskA), iB = (int)((incl >> kShB) & kMaskB), iC = (int)(incl >> kShC);
      if (cnt < 0) {
        pl.ovf[(size_t)tile * M + run[j][2] + iC - 1] = m;
      } else {
        Piece *tl = pl.pieces + (size_t)tile * pl.cap();
        // delta pieces at hand without another walk: the counting walk's first three, or (two kernels) the staged pair's one
        const int have = stage ? (cnt == 1 ? 1 : 0) : (cnt < kKeepPieces ? cnt : kKeepPieces);
        if (stage && cnt > have) {  // (rare: the walk below starts from what the classify kernel found)
          const ObjHdr hd = ps.hdr[m];
          base = hd.off, n = hd.cnt, kst = __float_as_int(pr.p0);
        }
        int k = kst;
        // the row of the segment the tile starts in: a ramp's start point, else the point itself (the last one
        // beyond the end of the curve): describe_segment's rule
        int row;
        if (stage) {
          row = pr.row0;
        } else if (ps.force_ramp) {
          row = describe_segment(ps, base, n, k, t0, t1).row;
        } else {
          const bool ramp = k > 0 && k < n && ((int)ps.rec[base + k].flat & ((1 << ps.nbus) - 1)) != (1 << ps.nbus) - 1;
          row = base + (ramp ? k - 1 : k == n ? k - 1 : k);
        }
        Piece b;
        b.m = (uint32_t)m;
        b.row = row;
        b.p0 = 1.0f;
        b.scale = 0.0f;
        // the first kKeepPieces delta pieces are still in registers from the counting walk; only an object with more
        // walks its curve points a second time (cache hits now)
        if (!paired) {
          Piece *out = tl + run[j][0] + iA - (1 + cnt);
          out[0] = b;
          if (have >= 1) out[1] = keep[0];
          if (have >= 2) out[2] = keep[1];
          if (have >= 3) out[3] = keep[2];
          if (cnt > have) (void)piece_walk(ps, base, n, k, t0, t1, m, out + 1);
        } else if (cnt == 0) {
          tl[run[j][0] + iA - 1] = b;
        } else {
          Piece *out = tl + pl.pair_off() + 2 * (run[j][1] + iB - cnt);
          out[0] = b;
          if (have >= 1) out[1] = keep[0];
          b.row = ps.zero_row;  // the even slots of the object's further pairs: its input, no gain
          if (have >= 2) out[2] = b, out[3] = keep[1];
          if (have >= 3) out[4] = b, out[5] = keep[2];
          if (cnt > have) {
            (void)piece_walk(ps, base, n, k, t0, t1, m, out + 1, nullptr, 2);
            for (int i = have < 1 ? 1 : have; i < cnt; i++) out[2 * i] = b;
          }
        }
      }
    }
#ifdef EARHIP_BUILD_PROF
    EARHIP_BUILD_MARK(prof_i + 2);  // this thread's pieces stored
#endif
    __syncthreads();  // (everybody has read run[] and lcnt[])
#ifdef EARHIP_BUILD_PROF
    EARHIP_BUILD_MARK(prof_i + 3);
    prof_i += 4;
#endif
    if (tid < TPW) {
      const u64 tot = lcnt[tid * OB + OB - 1];
      run[tid][0] += (int)(tot & kMaskA);
      run[tid][1] += (int)((tot >> kShB) & kMaskB);
      run[tid][2] += (int)(tot >> kShC);
    }
    __syncthreads();
  }
#ifdef EARHIP_BUILD_PROF
  EARHIP_BUILD_MARK(30);
#endif
  // pad the lists to whole chunks with null pieces; publish the counts
  Piece null_piece;
  null_piece.m = 0u;
  null_piece.row = ps.zero_row;
  null_piece.p0 = 1.0f;
  null_piece.scale = 0.0f;
  // (paired: the singles fill an EVEN number of chunks — the kernel takes them in twos)
  for (int i = tid; i < TPW * 64; i += kBuildThreads) {
    const int jj = i >> 6, tl = blockIdx.x * TPW + jj;
    if (tl >= ntiles) continue;
    Piece *lst = pl.pieces + (size_t)tl * pl.cap();
    const int np = run[jj][0], pp = paired ? (np + 63) & ~63 : (np + 31) & ~31;
    if (np + (i & 63) < pp) lst[np + (i & 63)] = null_piece;
    const int np2 = 2 * run[jj][1], pp2 = (np2 + 31) & ~31;  // (packed: no pairs)
    if (np2 + (i & 63) < pp2) lst[pl.pair_off() + np2 + (i & 63)] = null_piece;
    if ((i & 63) == 0) {
      const int listed = M - run[jj][2];  // (packed) every listed object has one base piece: the rest are deltas
      const int deltas = paired ? run[jj][1] : np - listed;
      pl.count[tl * 8 + 0] = (pp + pp2) >> 5;
      pl.count[tl * 8 + 1] = deltas;
      pl.count[tl * 8 + 2] = paired || deltas == 0 ? pp >> 5 : 0;
      pl.count[tl * 8 + 4] = run[jj][2];
    }
  }
}

// K1p.  grid = (workgroups, grid-level splits of the chunk schedule, column super-groups), block = 64 NW threads;
// ntl = WORKGROUP tiles (64 NW samples) of the call.  Workgroup b does tiles b, b + gridDim.x, ... (the host launches as many
// workgroups as are resident at once, a multiple of 8: a workgroup's tiles stay on its XCD) and carries its pipeline of piece
// words and gain rows from one tile's list straight into the next one's: by the time a tile's last chunk is done, the ring
// holds the first chunks of the next list, their gain rows are on their way and the first chunk's B fragments are being
// written — only the inputs of the next tile's first two chunks are requested anew, in front of the stores of the tile
// just finished.  (A workgroup per tile spent ~9 us of a tile's 117 outside its chunk loop on the ADM scene: the dependent
// trips list -> gain rows / inputs -> first MFMA with nothing else on the CU, the stores, the next workgroup's launch.)
// x_scale, g_scale: exact powers of two (gain_h2.h).
// WIDE: the low pieces of the inputs are kept scaled (kLowPieceScale, gain_h2.h) — 16 more multiplies, a third fragment of
// every column tile to write and to read.  Both forms are in the kernel; the probe's word (wide_cur) picks at run time, without
// a probe the wide form runs.
// (the 4-wave forms with three column tiles sit at the register limit: a workgroup per tile, nothing carried)
constexpr bool p2_persistent(int nct, int nw, bool paired) { return !(nw == 4 && nct == 3); }
template <int NCT, int NW, bool PAIRED>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : 2)
k_gain_mix_p2(GainMixParams P, PieceLists pl, float x_scale, const float *__restrict__ gcol, const unsigned *level_cur,
              unsigned *level_next, const unsigned *wide_cur, unsigned *wide_next, const unsigned *gate, int ntl) {
  if (gate && !(*gate & kGateHingeUnsafe)) return;  // launched behind the hinge kernel, which did this call (k_hinge_gate)
  constexpr int NRT = 4, TS = 16 * NRT, CH = kSplitChunk;
  constexpr int NQ = CH / NW;     // pieces whose gains one wave converts per chunk
  constexpr int NFRAG = NCT * 3;  // column tiles x {h, l, h 2^-11}
  constexpr int RING = 8;         // chunks of piece words (the object) staged in LDS for the lanes
  __shared__ u32x4 bfrag[2][NFRAG + 3][64];  // + 3 never-read fragments: the lanes without a column write there
  // the wave's output tile of one column tile on its way from the D fragments to stores of whole rows (gain_h2.h)
  constexpr int OP = TS + 4;
  __shared__ __attribute__((aligned(16))) float otile[NW][16 * OP];
  __shared__ float inv_gcol[16 * NCT];  // 1 / the gain scale of the workgroup's columns: read at the END of a tile, where a
                                        // round trip to memory would stand in the open (3 us of a 130-us tile)
  // ... byte offsets of the objects' input rows (object x row stride: one 64-bit multiply per piece by wave 0 instead of one
  // per request by every wave — quarter-rate each); slot RING: zeros — what requests past the end of a list read
  __shared__ __attribute__((aligned(16))) uint64_t ring[RING + 1][CH];
  __shared__ __attribute__((aligned(16))) u32x4 ringp[RING][CH];       // ... the whole pieces (rows, p0, scale)
  constexpr int NGI = (2 * NQ * 4 * NCT + 63) / 64;                    // float4 gain-row requests per wave and chunk
  __shared__ __attribute__((aligned(16))) f32x4 stage[NW][NGI * 64];   // the wave's gain rows of a chunk: [2 NQ][16 NCT] floats
  // Both forms of the body (WIDE or not) in ONE kernel, the probe's word picking at run time: two kernels launched back to
  // back, one of which looks at the word and returns, cost 5 us per call for the one that returns
  auto body = [&](auto wide_tag) __attribute__((always_inline)) {
  constexpr bool WIDE = decltype(wide_tag)::value;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kg = lane >> 4;
  if (threadIdx.x < 16 * NCT) inv_gcol[threadIdx.x] = 1.0f / gcol[blockIdx.z * 16 * NCT + threadIdx.x];
  if (threadIdx.x < CH) ring[RING][threadIdx.x] = 0ull;
  __syncthreads();
  if (level_cur) {  // input scale of THIS call from the level K0 probed (gain_h2.h)
    const unsigned lv = *level_cur;
    if (lv) {
      const int E = max(-60, min(20, (int)(lv >> 23) - 127));
      x_scale = __uint_as_float((unsigned)(127 + 7 - E) << 23);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
      *level_next = 0;
      // (the grid kernel's mode word of the NEXT call: whichever kernel runs a call clears it, or a "wide" left by an
      // earlier call would keep the grid kernel in its slower form two calls later)
      if (wide_next) *wide_next = 0u;
      record_mode(P, wide_cur != nullptr);
    }
  }
  const int nparts = gridDim.y;
  const int part = blockIdx.y;
  const int col0 = blockIdx.z * 16 * NCT;
  const int wave_s0 = w * TS;  // first sample of this wave inside a workgroup tile
  // the tile the workgroup is on (wave-uniform; set_tile)
  int wgtile = 0, tile_s0 = 0, tile_len = 0;  // ... the wave's first sample inside the call, its samples
  int64_t tile_t0 = 0, tile_t1 = 0;
  auto set_tile = [&](int t) {
    wgtile = t;
    tile_s0 = t * (TS * NW) + wave_s0;
    tile_len = max(0, min(TS, P.nsamples - tile_s0));
    tile_t0 = P.t_call + tile_s0;
    tile_t1 = tile_t0 + tile_len;
  };
  const float *__restrict__ gain = P.ps.gain;
  const unsigned rowlen = (unsigned)P.ps.row;
  // what runs between two tiles (the exact paths, the stores) makes its lane numbers anew, from a value the compiler cannot
  // see through, so that nothing of it is computed ahead of the tile loop and held in registers through the chunk loop
  auto opaque_lane = [&]() __attribute__((always_inline)) {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };

  // running totals in scaled units: bus = tot / (x_scale g_scale)
  f32x4 tot[NRT][NCT];
  auto clear_totals = [&]() {
#pragma unroll
    for (int r = 0; r < NRT; r++)
#pragma unroll
      for (int c = 0; c < NCT; c++) tot[r][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  };

  // ---- exact path: one object, all its pieces inside this wave's 64 samples, f32 MFMA with k = {a, b}
  // of ONE object (k slots 2, 3 idle), accumulated into tot in units of 1 / (sx sg)
  // (sg: the gains are scaled by their column's scale, like the split operands — or not at all)
  auto single_object = [&](int m, float sx, bool sg) {
    if (tile_len <= 0) return;
    const int ol = opaque_lane(), li = ol & 15, kg = ol >> 4;
    float gsc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; c++) gsc[c] = sg ? gcol[col0 + c * 16 + li] : 1.0f;
    // (offset, count and the curve's end points in one load, then a window of eight records around the evenly-spaced guess: two
    // dependent rounds of loads per object instead of the ~12 of a bisection — gain_kernels.h, ObjHdr; search.h)
    const ObjHdr hd = P.ps.hdr[m];
    const int base = hd.off, n = hd.cnt;
    const float *row = P.in + (size_t)m * P.in_stride + tile_s0;
    const bool is_b = kg & 1;
    const bool slot0 = kg < 2;
    int k = P.ps.force_ramp ? upper_bound_time(P.ps.time + base, n, tile_t0) : upper_bound_rec_window(P.ps.rec + base, n, hd.first, hd.last, tile_t0);
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
            tot[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], gv[c], tot[r][c], 0, 0, 0);
        cur = r1;
      }
      if (!(dk.info & kSegMulti)) break;
      k++;
    }
  };

  // one piece of a list, exact and unscaled (the wave redoes its share of the schedule this way when an
  // input left the f16 range).  Base: x S.  Delta: x clamp(p) (E - S) as the two k slots (-x p) S + (x p) E,
  // p(s) = p0 + s * scale instead of libear's (float)(d0 + s) * scale — equal to within an ulp of p, far
  // inside the tolerance of this (non-strict) kernel.
  auto single_piece = [&](const Piece pc) {
    if (tile_len <= 0) return;
    const int ol = opaque_lane(), li = ol & 15, kg = ol >> 4;
    const bool delta = pc.m & kPieceDelta;
    const float *row = P.in + (size_t)(pc.m & ~kPieceDelta) * P.in_stride + tile_s0;
    const bool is_b = kg & 1;
    const bool slot0 = kg < 2;
    float a[NRT], gv[NCT];
#pragma unroll
    for (int r = 0; r < NRT; r++) {
      const int s = li * NRT + r, sw = wave_s0 + s;
      const float x = row[min(s, tile_len - 1)];
      const float p = __builtin_amdgcn_fmed3f(__builtin_fmaf((float)sw, pc.scale, pc.p0), 0.0f, 1.0f);
      const float coef = delta ? (is_b ? p : -p) : (is_b ? 0.0f : 1.0f);
      a[r] = (slot0 && s < tile_len) ? x * coef : 0.0f;
    }
    const int grow = pc.row + ((delta && is_b && slot0) ? 1 : 0);
    const float *gp = gain + (size_t)grow * rowlen + col0 + li;
#pragma unroll
    for (int c = 0; c < NCT; c++) gv[c] = gp[c * 16];
#pragma unroll
    for (int r = 0; r < NRT; r++)
#pragma unroll
      for (int c = 0; c < NCT; c++)
        tot[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], gv[c], tot[r][c], 0, 0, 0);
  };

  // the tile is done: scale and store.  inv_x: the inverse of the input scale the totals carry (exact: a power of two);
  // col_scaled: they are in units of 1 / (the column's gain scale) as well
  auto store_tile = [&](float inv_x, bool col_scaled) {
    if (tile_len <= 0) return;
    const int ol = opaque_lane(), li = ol & 15, kg = ol >> 4;
    // D fragment of row tile r: rows 4kg + e = samples 16kg + 4e + r: for fixed e the
    // four row tiles are 4 consecutive samples.
    float *op = P.out + (size_t)blockIdx.y * P.part_stride + tile_s0;
    const bool whole = P.vec_ok && tile_len == TS;  // (wave-uniform)
    float inv_gc[NCT];  // inverse gain scale of the lane's column in each column tile (the D fragments' layout)
#pragma unroll
    for (int c = 0; c < NCT; c++) inv_gc[c] = col_scaled ? inv_gcol[c * 16 + li] : 1.0f;
#pragma unroll
    for (int c = 0; c < NCT; c++) {
      if (whole) {  // transposed through wave-private LDS: whole 256-byte rows per store instruction (gain_h2.h)
        float *ot = otile[w];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < NRT; r++) v[r] = (tot[r][c][e] * inv_x) * inv_gc[c];
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
        for (int r = 0; r < NRT; r++) v[r] = (tot[r][c][e] * inv_x) * inv_gc[c];
        if (P.vec_ok && s + 3 < tile_len) {
          *reinterpret_cast<f32x4 *>(o + s) = v;
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (s + i < tile_len) o[s + i] = v[i];
        }
      }
    }
  };

  const int gx = gridDim.x;
  if (!P.vec_ok) {  // unaligned rows: every object on the exact path
    const int m_lo = (int)(((int64_t)P.M * part) / nparts), m_hi = (int)(((int64_t)P.M * (part + 1)) / nparts);
    for (int vb = blockIdx.x; vb < ntl; vb += gx) {
      set_tile(xcd_tile(vb, ntl));
      clear_totals();
      for (int m = m_lo; m < m_hi; m++) single_object(m, 1.0f, false);
      store_tile(1.0f, false);
    }
    return;
  }

  // ---- the chunk schedule of a tile: its list, 32 pieces per chunk
  // Three KINDS of chunk (k_piece_build's two layouts):
  //   single  32 base pieces of 32 objects: every p is 1, the position factor is skipped (the first cnt[2] chunks of
  //           a list: all of them when the tile has no delta piece at all, the singles of the paired layout);
  //   pair    16 pairs (base or null-gain piece, delta piece) of 16 objects: 4 input requests per lane instead of 8,
  //           position factors for the odd slots only;
  //   packed  any 32 pieces in a row: a position factor for all of them, 8 requests of which the repeats hit in the
  //           first-level cache.
  struct Sched {
    const Piece *lbase;
    int total, n_single, pair_adj;  // chunks of the list, the single chunks among them, where the pairs' region starts
    int c_lo, c_hi;                 // this part's chunks
    bool has_delta;
  };
  auto tile_sched = [&](int t) {
    // (through the constant address space: scalar loads — K0 wrote the counts, this kernel only reads them)
    typedef const int __attribute__((address_space(4))) *ConstInt;
    ConstInt cn = (ConstInt)(pl.count + (size_t)t * 8);
    Sched s;
    s.lbase = pl.pieces + (size_t)t * pl.cap();
    s.total = cn[0];
    s.n_single = cn[2];
    s.has_delta = cn[1] > 0;
    s.pair_adj = PAIRED ? pl.pair_off() - 32 * s.n_single : 0;  // the pairs' region starts at pl.pair_off()
    // (paired layout: parts start at even chunks — the single chunks come in twos, k_piece_build pads them so)
    const int tunits = PAIRED ? (s.total + 1) / 2 : s.total, tu = PAIRED ? 2 : 1;
    s.c_lo = tu * (int)(((int64_t)tunits * part) / nparts);
    s.c_hi = min(s.total, tu * (int)(((int64_t)tunits * (part + 1)) / nparts));
    return s;
  };
  constexpr int KS = 0, KP = 1, KK = 2;
  constexpr int RD = PAIRED ? 5 : 4;  // ring slots up to chunk c + RD - 1 are visible during chunk c
  const int zero_row = P.ps.zero_row;
  const int nvec = (P.nsamples + 3) & ~3;
  Sched cur, nxt;           // the tile the workgroup is on and its next one
  bool prime_next = false;  // chunks past the end of cur's list are the first ones of nxt's
  int goff = 0;             // chunk c of cur's list is the workgroup's (c + goff)-th: ring slots and fragment buffers go by that
  int live_end = 0;         // input requests for chunks from here on read the zero slot
  // byte offset of this lane's float4 inside an input row (lanes past the end of the call re-read the
  // last vector: never stored)
  auto lane_offset = [&](int t) { return (unsigned)min(t * (TS * NW) + wave_s0 + li * NRT, nvec - 4) * 4u; };
  unsigned xlane = 0;
  const int bfr = lane < 16 * NCT ? (lane >> 4) * 3 : NFRAG;          // fragment triple (h, l, h 2^-11) this lane fills
  const uint64_t rstride = P.in_stride * sizeof(float);
  const float lane_sf = (float)(wave_s0 + li * NRT);  // the lane's first sample inside the workgroup tile

  // first piece of chunk c of cur's list; past its end: of nxt's first chunks (prime_next), else clamped (requests
  // past the schedule re-read its last chunk)
  auto chunk_ptr = [&](int c) -> const Piece * {
    const bool over = prime_next && c >= cur.c_hi;
    const Piece *lb = over ? nxt.lbase : cur.lbase;
    const int tt = over ? nxt.total : cur.total, ns = over ? nxt.n_single : cur.n_single, pa = over ? nxt.pair_adj : cur.pair_adj;
    c = min(over ? nxt.c_lo + (c - cur.c_hi) : c, tt - 1);
    return lb + 32 * c + (c >= ns ? pa : 0);
  };

  // piece words (the object) of chunk c for the lanes: wave 0 brings them into the ring, requested RD + 1
  // chunks ahead, stored RD ahead (visible after the next barrier), read two — pair chunks four — ahead (input
  // addresses, gain rows)
  // (one 16-byte request per chunk: every other per-piece datum the waves need comes out of this ring)
  auto ring_load = [&](int c) -> u32x4 { return *reinterpret_cast<const u32x4 *>(chunk_ptr(c) + (lane & 31)); };
  auto ring_store = [&](int c, u32x4 v) {
    if (lane < 32) {
      ring[(c + goff) & (RING - 1)][lane] = (uint64_t)(v[0] & ~kPieceDelta) * rstride;
      ringp[(c + goff) & (RING - 1)][lane] = v;
    }
  };
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
  auto xslot = [&](int c) { return c < live_end ? (c + goff) & (RING - 1) : RING; };
  auto lane_word2 = [&](int c, int q0) { return *reinterpret_cast<const u64x2 *>(&ring[xslot(c)][kg * 8 + q0]); };
  // inputs q0 .. q0 + n - 1 (n even) of chunk c, requested as a chunk of kind K.  Single and pair chunks ask for
  // every input once: streaming requests.  Packed chunks: ordinary ones — the pieces of one object are neighbours
  // in the list, so a lane asks for the same 16 bytes again in its next request; found in the first-level cache,
  // the repeat costs no second trip to L2 (ADM scene K1 0.545 -> 0.503 ms, always-ramping 0.91 -> 0.84).
  // Pair chunks need the even slots' inputs only — four registers: a chunk's go to the even (PAR = 0) or the odd
  // (PAR = 1) half of x, so that FOUR pair chunks are in flight in the registers that hold two others (half the
  // bytes per chunk: with two in flight the requests outstanding no longer cover the memory latency).
  // tr ("transition", single kind): the odd half takes the even slots of chunk c + 2 instead of the odd slots of
  // chunk c — what the last two single chunks request for the first four pair chunks behind them.
  auto lane_word = [&](int c, int slot) { return ring[xslot(c)][kg * 8 + slot]; };
  auto load_x_part = [&](auto kind_tag, auto par_tag, int c, f32x4 (&x)[8], int q0, int n, bool tr) __attribute__((always_inline)) {
    constexpr int K = decltype(kind_tag)::value, PAR = decltype(par_tag)::value;
    const char *bp = reinterpret_cast<const char *>(P.in) + xlane;
#pragma unroll
    for (int q = 0; q < 8; q += 2)
      if (q >= q0 && q < q0 + n) {
        if constexpr (K == KK) {
          const u64x2 mw = lane_word2(c, q);
          x[q] = *reinterpret_cast<const f32x4 *>(bp + mw[0]);
          x[q + 1] = *reinterpret_cast<const f32x4 *>(bp + mw[1]);
        } else if constexpr (K == KS) {
          const uint64_t m0 = lane_word(c, q), m1 = lane_word(tr ? c + 2 : c, tr ? q : q + 1);
          x[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(bp + m0));
          x[q + 1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(bp + m1));
        } else {
          const uint64_t m0 = lane_word(c, q);
          x[q + PAR] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(bp + m0));
        }
      }
  };
  // (the first two chunks of a list.  Paired kernel: as single chunks do — tr: a part without single chunks starts in the
  // transition form)
  auto load_x_all = [&](int c, f32x4 (&x)[8], bool tr) __attribute__((always_inline)) {
    load_x_part(std::integral_constant<int, PAIRED ? KS : KK>{}, std::integral_constant<int, 0>{}, c, x, 0, 8, tr);
  };
  // (p0, scale) of the lane's piece q of chunk c, out of the ring (read where it is used)
  auto piece_ps = [&](int c, int q) {
    const u32x2 v = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const char *>(&ringp[(c + goff) & (RING - 1)][kg * 8 + q]) + 8);
    return f32x2{__uint_as_float(v[0]), __uint_as_float(v[1])};
  };
  // The gain operand of a piece is row X minus row Y: delta (E, S), base (S, the all-zero row).  What this
  // WAVE converts for chunk c are pieces NQ w + q: their 2 NQ rows (16 NCT floats each) come in as NGI
  // requests of 16 bytes per lane — slot s = lane + 64 i covers floats 4 (s mod 4 NCT) .. + 3 of row
  // s div 4 NCT — instead of one 4-byte gather per row: the vector-memory address unit is the busiest unit
  // of this kernel (two thirds of all cycles) and pays per instruction, not per byte.  The rows go through a
  // wave-private piece of LDS to the lanes that convert them (lane = column).
  constexpr int RS = 4 * NCT;  // 16-byte slots per row
  auto load_gains = [&](int c, f32x4 (&G)[NGI]) {
#pragma unroll
    for (int i = 0; i < NGI; i++) {
      const int sl = min(lane + 64 * i, 2 * NQ * RS - 1);
      const int r = sl / RS, cg = sl - r * RS;
      const u32x2 mr = *reinterpret_cast<const u32x2 *>(&ringp[(c + goff) & (RING - 1)][w * NQ + (r >> 1)]);  // (m, row)
      const bool d = mr[0] & kPieceDelta;
      const unsigned row = (r & 1) ? (d ? mr[1] : (unsigned)zero_row) : mr[1] + (d ? 1u : 0u);
      G[i] = *reinterpret_cast<const f32x4 *>(gain + (size_t)row * rowlen + col0 + 4 * cg);
    }
  };
  const int col_e = min(lane, 16 * NCT - 1);  // the lane's gain column inside the wave's rows
  const float g_scale = gcol[col0 + col_e];   // ... and that column's scale (a power of two)
  auto stage_gains = [&](const f32x4 (&G)[NGI], float (&X)[NQ], float (&Y)[NQ]) {
#pragma unroll
    for (int i = 0; i < NGI; i++) stage[w][lane + 64 * i] = G[i];
    const float *sf = reinterpret_cast<const float *>(&stage[w][0]);
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      X[q] = sf[(2 * q) * (16 * NCT) + col_e];
      Y[q] = sf[(2 * q + 1) * (16 * NCT) + col_e];
    }
  };
  // the B operand of the wave's NQ pieces, scaled and split -> LDS
  // (k = NQ w + q of the fragment entry of lane 16 (k / 8) + column)
  auto store_b = [&](const float (&X)[NQ], const float (&Y)[NQ], int buf) {
    uint32_t h[NQ / 2], l[NQ / 2], hs[NQ / 2];
#pragma unroll
    for (int i = 0; i < NQ / 2; i++) {
      float v[2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int q = 2 * i + j;
        v[j] = (X[q] - Y[q]) * g_scale;
      }
      const uint32_t H = pack_f16(v[0], v[1]);
      h[i] = H;
      l[i] = pack_f16(sub_f16_lo(v[0], H), sub_f16_hi(v[1], H));  // residuals: exact in fp32
      hs[i] = WIDE ? scale_f16x2_down(H) : 0u;                // h 2^-11: partner of the inputs' scaled low piece (gain_h2.h)
    }
    u32x4 *f = &bfrag[buf][bfr][0];
    const int col = lane & 15;
    if constexpr (NQ == 8) {
      f[w * 16 + col] = u32x4{h[0], h[1], h[2], h[3]};
      f[64 + w * 16 + col] = u32x4{l[0], l[1], l[2], l[3]};
      if constexpr (WIDE) f[128 + w * 16 + col] = u32x4{hs[0], hs[1], hs[2], hs[3]};
    } else {  // half an entry: words 2 (w & 1), + 1 of k group w / 2
      u32x2 *g = reinterpret_cast<u32x2 *>(f + (w >> 1) * 16 + col) + (w & 1);
      g[0] = u32x2{h[0], h[1]};
      g[128] = u32x2{l[0], l[1]};
      if constexpr (WIDE) g[256] = u32x2{hs[0], hs[1]};
    }
  };

  u32x4 ring_next = {0u, 0u, 0u, 0u};
  f32x4 X0[8], X1[8];
  // gain rows on their way: requested TWO chunks ahead of their use, converted one chunk ahead (a chunk of 36 MFMAs
  // is shorter than a loaded trip to L2): the chunks that use X0 request into GA and convert GB, the others the
  // other way round
  f32x4 GA[NGI], GB[NGI];

  // ---- prologue of a list nothing was carried into: piece words of the first chunks into the ring, gains of the first
  // chunk, inputs of the first two
  auto prologue = [&](bool first) __attribute__((always_inline)) {
    if (!first) __syncthreads();  // (the list before is done with the ring and the fragments)
    if (w == 0) {
#pragma unroll
      for (int j = 0; j < RD; j++) ring_store(cur.c_lo + j, ring_load(cur.c_lo + j));
      ring_next = ring_load(cur.c_lo + RD);  // (stored by the first chunk)
    }
    __syncthreads();
    f32x4 G[NGI];
    float S[NQ], E[NQ];
    load_gains(cur.c_lo, G);
    load_gains(cur.c_lo + 1, GB);  // (converted by the first chunk)
    load_x_all(cur.c_lo, X0, PAIRED && cur.c_lo >= cur.n_single);
    load_x_all(cur.c_lo + 1, X1, PAIRED && cur.c_lo >= cur.n_single);
    stage_gains(G, S, E);
    store_b(S, E, (cur.c_lo + goff) & 1);
  };

  // chunk c: inputs in xc, B fragments in bfrag[(c + goff) & 1].  KC: its kind — the inputs it requests (chunk c + 2; a pair
  // chunk: c + 4) are requested as that kind's; PAR: the half of xc a pair chunk's inputs are in; tr: see
  // load_x_part; Gld / Gcv: the gain rows it requests (chunk c + 2) and converts (chunk c + 1)
  auto chunk = [&](auto kc_tag, auto par_tag, int c, f32x4 (&xc)[8], f32x4 (&Gld)[NGI], f32x4 (&Gcv)[NGI], bool tr) __attribute__((always_inline)) {
    constexpr int KC = decltype(kc_tag)::value, K2 = KC, PAR = decltype(par_tag)::value;
    const int buf = (c + goff) & 1;
    __syncthreads();  // B fragments of chunk c are in bfrag[buf]; bfrag[buf^1] is free; ring slots <= c + 3 are visible
    if (w == 0) {
      ring_store(c + RD, ring_next);
      ring_next = ring_load(c + RD + 1);
    }
#if !(EARHIP_P2_ABL & 2)
    load_gains(c + 2, Gld);  // the rows of chunk c + 2 (its pieces have been in the ring for two chunks or more)
#endif
    __builtin_amdgcn_sched_barrier(0);  // every gain row is requested before any input

    // A fragments: row tile r = sample 4*li + r of the 8 pieces of this lane.  2 x 2 blocks: an f16
    // pair packs two PIECES (q, q+1) of one row tile, the scaling and the exact residual
    // subtractions pair two SAMPLES (r, r+1) of one piece.  A piece's ramp position is part of its
    // input scale: x_scale clamp(p0 + s scale, 0, 1) (base pieces: p = 1).
    u32x4 ah[NRT], al[NRT];
    auto split = [&](auto ks_tag) __attribute__((always_inline)) {
      constexpr int KSP = decltype(ks_tag)::value;  // how the pieces' inputs become operands: as a single, pair or packed chunk's
#pragma unroll
      for (int qp = 0; qp < 4; qp++) {
#pragma unroll
        for (int rp = 0; rp < NRT; rp += 2) {
          const int XE = KSP == KP ? 2 * qp + PAR : 2 * qp;  // (a pair chunk's inputs: one half of xc)
          f32x2 s0 = f32x2{xc[XE][rp], xc[XE][rp + 1]} * x_scale;  // piece 2qp
          f32x2 s1;                                                        // piece 2qp+1
          if constexpr (KSP == KP) s1 = s0;  // (the same input)
          else s1 = f32x2{xc[2 * qp + 1][rp], xc[2 * qp + 1][rp + 1]} * x_scale;
          if constexpr (KSP != KS) {
            const float f0 = lane_sf + (float)rp, f1 = lane_sf + (float)(rp + 1);
            // (the compiler folds the median-of-three into the FMA's output clamp: one instruction per value)
            if constexpr (KSP == KK) {
              const f32x2 a = piece_ps(c, 2 * qp);
              s0 *= f32x2{__builtin_amdgcn_fmed3f(__builtin_fmaf(f0, a[1], a[0]), 0.0f, 1.0f),
                          __builtin_amdgcn_fmed3f(__builtin_fmaf(f1, a[1], a[0]), 0.0f, 1.0f)};
            }
            const f32x2 b = piece_ps(c, 2 * qp + 1);
            s1 *= f32x2{__builtin_amdgcn_fmed3f(__builtin_fmaf(f0, b[1], b[0]), 0.0f, 1.0f),
                        __builtin_amdgcn_fmed3f(__builtin_fmaf(f1, b[1], b[0]), 0.0f, 1.0f)};
          }
          const uint32_t H0 = pack_f16(s0[0], s1[0]), H1 = pack_f16(s0[1], s1[1]);
          ah[rp][qp] = H0;
          ah[rp + 1][qp] = H1;
          constexpr float LOW = WIDE ? kLowPieceScale : 1.0f;
          const f32x2 r0 = f32x2{sub_f16_lo(s0[0], H0), sub_f16_lo(s0[1], H1)} * LOW;  // exact residuals (wide: scaled, gain_h2.h)
          const f32x2 r1 = f32x2{sub_f16_hi(s1[0], H0), sub_f16_hi(s1[1], H1)} * LOW;
          al[rp][qp] = pack_f16(r0[0], r1[0]);
          al[rp + 1][qp] = pack_f16(r0[1], r1[1]);
        }
      }
    };
    if constexpr (KC == KK) {  // (a packed list without any delta piece: every p is 1 — uniform over the workgroup)
      if (cur.has_delta) split(std::integral_constant<int, KK>{});
      else split(std::integral_constant<int, KS>{});
    } else {
      split(kc_tag);
    }
    __builtin_amdgcn_sched_barrier(0);  // the splitting above stays above
    // NCT blocks of 12 MFMAs (three partial products of one column tile, smallest first).  The inputs
    // of chunk c + 2 go into the registers just freed, a few requests per block; the conversion of the
    // next chunk's B operand is woven between the MFMAs of the last block.
    constexpr int XB = NCT >= 3 ? 2 : 1;  // blocks that carry input requests (4 or 8 each)
    auto load_b = [&](int fr, u32x4 (&bb)[2]) {
#pragma unroll
      for (int q = 0; q < 2; q++) bb[q] = bfrag[buf][fr + q][lane];
    };
    u32x4 b[2][2], b2;  // (wide: the scaled high piece is read as its block starts: gain_h2.h)
    f32x4 tsum[2][NRT];  // (packed chunks: the sums of a block's products, see below)
    load_b(0, b[0]);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {
      u32x4(&bc)[2] = b[ct & 1];
      if constexpr (WIDE) b2 = bfrag[buf][ct * 3 + 2][lane];
      else b2 = bc[0];  // (plain: the low pieces meet the high pieces of the gains as they are)
      if (ct + 1 < NCT) load_b((ct + 1) * 3, b[(ct + 1) & 1]);
      if constexpr (KC == KK && kChunkSums) {
        // Packed lists have the longest chains (three pieces per object when every object ramps all the time: ~300
        // MFMAs on one accumulator, each of which rounds the running total): the chunk's three products are summed
        // among themselves first, the total takes ONE addition per chunk (rounding error of the total ~ 1 / sqrt 3)
        // (a block's sums are added behind the first MFMAs of the NEXT block — the last block's behind the conversion
        // of the next chunk's gains —, where they do not wait for the chain they close)
        f32x4(&t)[NRT] = tsum[ct & 1];
#pragma unroll
        for (int r = 0; r < NRT; r++) t[r] = mfma_f16(ah[r], bc[1], f32x4{0.0f, 0.0f, 0.0f, 0.0f});
        if (ct > 0) {
#pragma unroll
          for (int r = 0; r < NRT; r++) tot[r][ct - 1] += tsum[(ct - 1) & 1][r];
        }
#pragma unroll
        for (int r = 0; r < NRT; r++) t[r] = mfma_f16(al[r], b2, t[r]);
#pragma unroll
        for (int r = 0; r < NRT; r++) t[r] = mfma_f16(ah[r], bc[0], t[r]);
      } else {
#pragma unroll
        for (int r = 0; r < NRT; r++) tot[r][ct] = mfma_f16(ah[r], bc[1], tot[r][ct]);
#pragma unroll
        for (int r = 0; r < NRT; r++) tot[r][ct] = mfma_f16(al[r], b2, tot[r][ct]);
#pragma unroll
        for (int r = 0; r < NRT; r++) tot[r][ct] = mfma_f16(ah[r], bc[0], tot[r][ct]);
      }
#if !(EARHIP_P2_ABL & 1)
      if (ct < XB) load_x_part(kc_tag, par_tag, KC == KP ? c + 4 : c + 2, xc, ct * (8 / XB), 8 / XB, tr);
#endif
      const bool conv = ct == NCT - 1;
      if (conv) {
        float S[NQ], E[NQ];
        stage_gains(Gcv, S, E);  // (the rows of chunk c + 1)
        store_b(S, E, buf ^ 1);
      }
      if constexpr (WIDE) {
        if (ct + 1 < NCT) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      } else {
        if (ct + 1 < NCT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      if (ct < XB && !conv) {
        constexpr int NL = (K2 == KP ? 4 : 8) / XB;  // requests of this block
#pragma unroll
        for (int k = 0; k < NL; k++) {  // MFMAs, then one request (address arithmetic + load)
          __builtin_amdgcn_sched_group_barrier(0x008, 12 / NL, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
    }
    if constexpr (KC == KK && kChunkSums) {
#pragma unroll
      for (int r = 0; r < NRT; r++) tot[r][NCT - 1] += tsum[(NCT - 1) & 1][r];
    }
  };
  // cur's chunks of this part
  auto run_chunks = [&]() __attribute__((always_inline)) {
    const int c_lo = cur.c_lo, c_hi = cur.c_hi;
    constexpr std::integral_constant<int, 0> p0{};
    constexpr std::integral_constant<int, 1> p1{};
    if constexpr (PAIRED) {
      // the single chunks of this part (an even number), the last two of them in the transition form, then its
      // pair chunks, four in flight
      const int c_mid = min(max(cur.n_single, c_lo), c_hi);
      constexpr std::integral_constant<int, KS> ks{};
      constexpr std::integral_constant<int, KP> kp{};
      int c = c_lo;
#pragma unroll 1
      for (; c + 3 < c_mid; c += 2) {
        chunk(ks, p0, c, X0, GA, GB, false);
        chunk(ks, p0, c + 1, X1, GB, GA, false);
      }
      if (c < c_mid) {
        chunk(ks, p0, c, X0, GA, GB, true);
        chunk(ks, p0, c + 1, X1, GB, GA, true);
        c += 2;
      }
#pragma unroll 1
      for (; c < c_hi; c += 4) {
        chunk(kp, p0, c, X0, GA, GB, false);
        if (c + 1 < c_hi) chunk(kp, p0, c + 1, X1, GB, GA, false);
        if (c + 2 < c_hi) chunk(kp, p1, c + 2, X0, GA, GB, false);
        if (c + 3 < c_hi) chunk(kp, p1, c + 3, X1, GB, GA, false);
      }
    } else {
      constexpr std::integral_constant<int, KK> kk{};
#pragma unroll 1
      for (int c = c_lo; c < c_hi; c += 2) {
        chunk(kk, p0, c, X0, GA, GB, false);
        if (c + 1 < c_hi) chunk(kk, p0, c + 1, X1, GB, GA, false);
      }
    }
  };

  bool primed = false, first = true;
  for (int vb = blockIdx.x; vb < ntl; vb += gx) {
    const int t = xcd_tile(vb, ntl);
    set_tile(t);
    if (!primed) {
      cur = tile_sched(t);
      goff = 0;
    }
    const bool has_next = vb + gx < ntl;
    const int tn = xcd_tile(has_next ? vb + gx : vb, ntl);
    nxt = tile_sched(tn);
    // (the next list takes over ring entries up to its chunk RD: it has to have them)
    prime_next = p2_persistent(NCT, NW, PAIRED) && has_next && cur.c_hi > cur.c_lo && nxt.c_hi - nxt.c_lo >= RD + 2;
    live_end = cur.c_hi;
    xlane = lane_offset(t);
    clear_totals();
    if (cur.c_hi > cur.c_lo) {
      if (!primed) prologue(first);
      first = false;
      run_chunks();
      // (carried into the next list: the gain rows of ITS second chunk, requested by this list's last chunk — into GA when
      // that was an odd one of the loop's pairs)
      if (prime_next && ((cur.c_hi - cur.c_lo) & 1)) {
#pragma unroll
        for (int i = 0; i < NGI; i++) GB[i] = GA[i];
      }
    }

    // objects that take the exact path (too many ramps for the lists, quiet ones): part 0 only
    float inv_x = 1.0f / x_scale;  // exact: a power of two
    bool col_scaled = true;        // the totals are in units of 1 / (x_scale x the column's gain scale)
    typedef const int __attribute__((address_space(4))) *ConstInt;
    const int novf = ((ConstInt)(pl.count + (size_t)wgtile * 8))[4];
    const int *ovf = pl.ovf + (size_t)wgtile * P.M;
    if (part == 0)
      for (int i = 0; i < novf; i++) single_object(ovf[i], x_scale, true);
    // an input beyond the f16 range (or not finite) shows as non-finite totals: redo the wave's tile
    // exactly, unscaled (every part redoes its share of the pieces)
    bool bad = false;
#pragma unroll
    for (int r = 0; r < NRT; r++)
#pragma unroll
      for (int c = 0; c < NCT; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) bad |= !(__builtin_fabsf(tot[r][c][e]) < INFINITY);
    if (__ballot(bad)) {
      clear_totals();
      inv_x = 1.0f;
      col_scaled = false;
      for (int c = cur.c_lo; c < cur.c_hi; c++)
        for (int j = 0; j < CH; j++) single_piece(chunk_ptr(c)[j]);
      if (part == 0)
        for (int i = 0; i < novf; i++) single_object(ovf[i], 1.0f, false);
    }
    if (prime_next) {
      // the ring holds nxt's first chunks, the B fragments of its first one are written, the gain rows of its second
      // are on their way.  Its first inputs are requested here, in front of this tile's stores (behind the rare paths
      // above, which want the registers)
      live_end = 0x7fffffff;
      xlane = lane_offset(tn);
      const bool trn = PAIRED && nxt.c_lo >= nxt.n_single;
      load_x_all(cur.c_hi, X0, trn);
      load_x_all(cur.c_hi + 1, X1, trn);
    }
    store_tile(inv_x, col_scaled);
    primed = prime_next;
    if (primed) {
      goff += cur.c_hi - nxt.c_lo;
      cur = nxt;
    }
  }
  };  // body
  if (!wide_cur || (*wide_cur & 1u) != 0u) body(std::true_type{});
  else body(std::false_type{});
}

}  // namespace earhip
