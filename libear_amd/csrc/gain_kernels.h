// gain_kernels.h — K0 (segment prep) and K1 (gain_mix): piecewise-linear gain
// curves applied to M planar input channels and mixed down into loudspeaker
// buses.  Device restatement of libear's LinearInterp{Single,Vector,Matrix}
// (include/ear/dsp/gain_interpolator.hpp:187-299) and of the segment walk of
// GainInterpolator::process (:53-87), re-shaped for CDNA4:
//
//   * lane = 4 consecutive samples (one 16-byte coalesced load per object),
//     wave = one 256-sample tile x one group of <= 24 output columns,
//     workgroup = 8 waves = column groups x in-workgroup object splits;
//   * per object the two gain rows are wave-uniform, so they are fetched with
//     scalar loads and enter the FMAs as SGPR operands; the 96 accumulators
//     (24 columns x 4 samples) never leave VGPRs until the tile is finished;
//   * the ramp is folded into the input once per object and sample:
//         a = x*(1-p), b = x*p          (shared by all columns)
//         acc[col] += a*S[col] + b*E[col]   -> 2 FMAs per (object,column,sample)
//     instead of the reference's 5 un-fused flops;
//   * object splits are combined through LDS in a fixed order (deterministic).
//
// STRICT instantiations keep libear's arithmetic verbatim (un-contracted
// `x * ((1-p)*s + p*e)`, objects accumulated in order by a single wave): they
// are bit-identical to the CPU path and are used for parity pinning and for
// the 1->N policies, which are store-bound anyway.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "search.h"

namespace earhip {

// One curve point as the list builder of the piece-list kernel reads it (gain_p2.h): everything about the segment
// that ENDS at this point in one 16-byte load.
struct PointRec {
  int64_t time;    // the point's time
  float scale;     // 1.0f / (float)(time - previous point's time) (gain_interpolator.hpp:191,221,255); 0 for the first
                   // point of an object and for two equal times
  uint32_t flat;   // bit b: bus b's gain vector equals the previous point's
};
static_assert(sizeof(PointRec) == 16, "PointRec is loaded as one dwordx4");

// Flattened gain curves of all objects (device pointers).
// What a search of an object's points starts from, in one 32-byte load: where its points are and the times of the first and
// the last one (the list builders' threads used to fetch these in two dependent rounds: offset and count, then the two
// times from the curve itself — a trip to HBM each).  Made where curves are committed (CurveSet::commit).
struct ObjHdr {
  int32_t off, cnt;
  int64_t first, last;  // times of points 0 and cnt - 1
  int64_t reserved;
};
static_assert(sizeof(ObjHdr) == 32, "an object's header is two dwordx4 loads");
struct PointStore {
  const ObjHdr *hdr;     // [M] offset, count, first and last time of each object
  const int32_t *off;    // [M] first point of each object (its region of the arena: curves.h)
  const int32_t *cnt;    // [M] its points
  const int64_t *time;   // [P]   point times, sorted per object
  const uint8_t *flat;   // [P]   bit b: bus b's gain vector at point k equals point k-1
  const PointRec *rec;   // [P]   the same three facts per point, packed (time, 1 / segment length, flat bits)
  const float *gain;     // [P][row] gain rows (bus-major columns, zero padded)
  int row;               // floats per row (multiple of 4)
  int bus_cols;          // columns per bus (columns [b*bus_cols, (b+1)*bus_cols) = bus b)
  int nbus;              // 1 or 2; each bus is one libear GainInterpolator
  int zero_row;          // index of an all-zero gain row (the row behind it is all zero as well)
  int npoints;           // points of all objects together
  int rows;              // rows of the gain image (the arena: all regions, live or not)
  int kink_row0;         // hinge kernel: the kink rows (gain_hg.h, k_kink_rows) start this many rows behind `gain`; 0: none
  int force_ramp;        // policy mode: every object is ONE ramp through its 2 points,
                         // extrapolated outside (LinearInterp*::apply_interp as called directly)
};

// What one (object, sample tile) pair has to do; written by K0.  A tile is
// 64 lanes x SPL samples (SPL = 2 or 4).
struct SegDesc {
  int32_t row;    // first gain row to load (ramp: start point; constant: the point)
  int32_t d0;     // ramp: tile_start - curve_start (>= 0, < 2^31)
  float scale;    // ramp: 1.0f / (float)(end - start)
  int32_t info;   // bit0 ramp, bit1 the segment ends inside this tile, bits 2-3 per-bus
                  // "constant" flags, bits 4-12 end of this piece relative to the tile
                  // start (samples, <= 256), bits 13..: segment index k
};
static_assert(sizeof(SegDesc) == 16, "SegDesc is loaded as one dwordx4");

constexpr int kSegRamp = 1;
constexpr int kSegMulti = 2;
constexpr int kSegQuiet = (int)0x80000000u;  // the object's input level is far below the call's (LevelProbe)
constexpr int kMaxPointsPerObject = (1 << 18) - 1;  // the k field of a descriptor: 18 bits below the quiet flag
static_assert((((unsigned)kMaxPointsPerObject << 13) & (unsigned)kSegQuiet) == 0u, "the segment index must not reach the quiet flag");
__host__ __device__ __forceinline__ int seg_r1(int info) { return (info >> 4) & 0x1ff; }
__host__ __device__ __forceinline__ int seg_k(int info) { return (info >> 13) & 0x3ffff; }


// blockIdx.x -> tile such that the workgroups of one XCD (b % 8) cover a
// contiguous range of tiles; identity for the ragged tail
__device__ __forceinline__ int xcd_tile(int b, int n) {
  const int per = n >> 3;  // tiles per XCD
  if (b >= per * 8) return b;
  return (b & 7) * per + (b >> 3);
}

// Describe segment k of an object for the tile [t0, t_end) (absolute times).
__device__ __forceinline__ SegDesc describe_segment(const PointStore &ps, int base,
                                                    int n, int k, int64_t t0,
                                                    int64_t t_end) {
  // (k, t_end are adjusted below in policy mode)
  SegDesc d;
  if (ps.force_ramp) k = 1;
  const int allflat = (1 << ps.nbus) - 1;
  const int fb = (k > 0 && k < n) ? (ps.flat[base + k] & allflat) : allflat;
  // constant before the first / after the last point or between equal points
  // (gain_interpolator.hpp:68-75)
  const bool ramp = ps.force_ramp || fb != allflat;
  const bool multi = !ps.force_ramp && k < n && ps.time[base + k] < t_end;
  // (9 bits: a piece that covers a whole 512-sample tile is stored as 511 — a piece that ends INSIDE a tile,
  // the only kind whose end is ever read, is shorter — so that it cannot spill into the k field)
  const int r1 = min((int)((multi ? ps.time[base + k] : t_end) - t0), 511);
  d.info = (k << 13) | (r1 << 4) | (fb << 2) | (multi ? kSegMulti : 0) | (ramp ? kSegRamp : 0);
  if (ramp) {
    const int64_t start = ps.time[base + k - 1], end = ps.time[base + k];
    d.row = base + k - 1;
    d.d0 = (int32_t)(t0 - start);
    d.scale = 1.0f / (float)(end - start);  // gain_interpolator.hpp:191,221,255
  } else {
    d.row = base + (k == n ? k - 1 : k);  // gain_interpolator.hpp:71-72
    d.d0 = 0;
    d.scale = 0.0f;
  }
  return d;
}

// The same from the packed point records: ONE 16-byte load for the point the segment ends at (time, 1 / length,
// flat bits) beside one for the point it starts at, instead of two dependent rounds of loads over three arrays.
__device__ __forceinline__ SegDesc describe_segment_rec(const PointStore &ps, int base, int n, int k, int64_t t0,
                                                        int64_t t_end) {
  if (ps.force_ramp) return describe_segment(ps, base, n, k, t0, t_end);
  const int allflat = (1 << ps.nbus) - 1;
  PointRec r, q;
  r.time = t_end;
  r.scale = 0.0f;
  r.flat = (uint32_t)allflat;
  q = r;
  if (k < n) r = ps.rec[base + k];
  if (k > 0) q = ps.rec[base + k - 1];
  const int fb = (k > 0 && k < n) ? ((int)r.flat & allflat) : allflat;
  const bool ramp = fb != allflat;
  const bool multi = k < n && r.time < t_end;
  const int r1 = min((int)((multi ? r.time : t_end) - t0), 511);
  SegDesc d;
  d.info = (k << 13) | (r1 << 4) | (fb << 2) | (multi ? kSegMulti : 0) | (ramp ? kSegRamp : 0);
  if (ramp) {
    d.row = base + k - 1;
    d.d0 = (int32_t)(t0 - q.time);
    // (the record holds 1.0f / (float)(end - start); two equal times with different gains: 0 there, inf in
    // describe_segment — such a segment is empty, its scale is never used)
    d.scale = r.scale;
  } else {
    d.row = base + (k == n ? k - 1 : k);
    d.d0 = 0;
    d.scale = 0.0f;
  }
  return d;
}

// K0: one thread per (object, tile); a 256-thread block covers 16 objects x 16
// tiles.  Lanes that are neighbours in `tile` share their object's time array
// (the searches hit the same lines); the block transposes through LDS so that
// the descriptors land TILE-MAJOR, desc[tile][M]: K1 walks the objects of one
// tile and reads them as one contiguous stream.
// Input level (f16x2 gain kernel): with `level` set, every thread of every `level_every`-th tile also
// looks at four input samples of its object and tile; the largest magnitude seen (as float bits) is
// left in *level.  The f16x2 kernel scales the inputs of the call by a power of two chosen from it.
struct LevelProbe {
  const float *in = nullptr;  // nullptr: no probing
  size_t in_stride = 0;
  int nsamples = 0;
  int every = 1;              // probe tiles 0, every, 2 every, ...
  unsigned *level = nullptr;  // zero before the launch
  unsigned *obj_level = nullptr;  // [M], zero before the launch: the same per object (largest magnitude probed)
};
// An object whose probed level lies this many binades below the call's is "quiet": the split-operand
// kernels scale every input of a call by ONE power of two; with the low piece of an input kept as
// (residual x 2^11) (gain_h2.h) both pieces are normal f16 numbers — 2^-22 relative precision — over 21
// binades below the level the prescale aims at (measured: 6.5e-8 relative RMS of the products down to 2^-18,
// 1.4e-7 at 2^-20, 5e-7 at 2^-22; round 2, with the plain residual: 2.5e-7 at 2^-10, 1e-6 at 2^-12).  An
// object quieter than that — more than 120 dB below the loudest — alone on a loudspeaker would be off by
// more than 1e-6 there, so it takes the exact f32 path of the kernel instead.  (Levels are float bits: biased
// exponents compare.)
constexpr int kQuietBinades = 20;
// ... and one more than this many binades below it makes the f16x2 kernel (gain_h2.h) run the call in its wide mode
// (scaled low pieces); up to here the plain residual is as good (2.5e-7 at 2^-10) and 4 % faster
constexpr int kPlainBinades = 8;
__host__ __device__ __forceinline__ bool level_is_quiet(unsigned obj_level, unsigned call_level) {
  return obj_level != 0u && call_level != 0u && (int)(obj_level >> 23) < (int)(call_level >> 23) - kQuietBinades;
}
// The level probe of the split-operand kernels, a small launch ahead of everything that needs it: a WAVE per object
// looks at kProbeInstants float4s spread over the whole call (16 jittered runs of 16 consecutive samples; rounds 2
// and 3 looked at two instants per object from inside the list builders — a decision for 11 s of audio from 8 samples).
// Left behind: obj_level[m] = the object's largest magnitude probed (float bits; plain store: every object is written by
// every call), obj_level[cap + m] = the smallest NON-ZERO one among its instants (how far the object falls below its own
// peaks: fades, tails, pauses in noise; digital silence needs no precision), and *level raised to the call's maximum (zero
// before the launch: the words alternate between calls, gain_h2.h).  Costs ~5 us whatever the length of the call.
constexpr int kProbeInstants = 64, kProbeObjects = 16;  // float4s per object; objects (waves) per workgroup
static __global__ void __launch_bounds__(64 * kProbeObjects)
k_level_probe(const float *in, size_t in_stride, int nsamples, int M, unsigned *level, unsigned *obj_level, int cap, int nruns) {
  __shared__ unsigned wmax[kProbeObjects];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, m = blockIdx.x * kProbeObjects + wv;
  const int nvec = nsamples >> 2, ninst = min(kProbeInstants, nvec);
  unsigned v = 0u;
  if (m < M && lane < ninst) {
    // The instants come in nruns runs of consecutive float4s (a power of two up to 64; 16: runs of 16 samples).  Run g:
    // somewhere in its own stretch of the call, at a place that differs from object to object.
    const unsigned sh = (unsigned)__builtin_ctz((unsigned)nruns), per = 64u >> sh;  // float4s per run
    const unsigned nrun = max((unsigned)ninst / per, 1u), g = min((unsigned)lane / per, nrun - 1u), sub = (unsigned)lane & (per - 1u);
    const unsigned stretch = (unsigned)nvec / nrun;  // float4s of the call per run
    const unsigned h = ((unsigned)m * 2654435761u) ^ (g * 40503u);
    const unsigned room = stretch > per ? stretch - per + 1u : 1u;
    const int at = (int)min(g * stretch + __umulhi(h, room) + sub, (unsigned)nvec - 1u);
    const float4 px = *reinterpret_cast<const float4 *>(in + (size_t)m * in_stride + 4 * (size_t)at);
    v = max(max(__float_as_uint(px.x) & 0x7fffffffu, __float_as_uint(px.y) & 0x7fffffffu),
            max(__float_as_uint(px.z) & 0x7fffffffu, __float_as_uint(px.w) & 0x7fffffffu));
  }
  unsigned hi_v = v, lo_v = v ? v : 0xffffffffu;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    hi_v = max(hi_v, (unsigned)__shfl_xor((int)hi_v, d));
    lo_v = min(lo_v, (unsigned)__shfl_xor((int)lo_v, d));
  }
  if (lane == 0) {
    wmax[wv] = hi_v;
    if (m < M) {
      obj_level[m] = hi_v;
      obj_level[cap + m] = lo_v == 0xffffffffu ? 0u : lo_v;
    }
  }
  __syncthreads();
  // the call's level: one word for everybody behind this kernel.  Same-address traffic is what it costs — a thousand
  // waves that only LOOK at the word (a coherent load each) took 11 of this kernel's 16 us — so a workgroup's 16 objects
  // go there once, and with the mantissa cleared (only the exponent is ever used), so that most find the word there already
  if (threadIdx.x == 0) {
    unsigned he = 0u;
#pragma unroll
    for (int i = 0; i < kProbeObjects; i++) he = max(he, wmax[i]);
    he &= 0x7f800000u;
    if (he != 0 && he > __atomic_load_n(level, __ATOMIC_RELAXED)) atomicMax(level, he);
  }
}

// The hinge kernel (gain_hg.h) keeps 1e-6 for inputs down to kHingeSpreadBinades below the level the prescale aims at (the
// piece-list kernel: 21).  Behind the probe: bit 1 of *gate is raised when some object that does not take the exact path anyway
// FALLS further than that at some probed instant — the hinge kernel and its list builder then return at once and the piece
// lists, launched behind them, do the call (each pair of launches looks at the word: the ones that do not work cost ~3 us).
constexpr int kHingeSpreadBinades = 16;
constexpr unsigned kGateHingeUnsafe = 2u;
// (round 6, `count_quiet`: the objects that are quiet THROUGHOUT the call — their loudest probed instant below the span — are
// counted into bits 8.. of the word as well: one of them is cheaper on the exact path than the robust form is for everybody,
// a tenth of the scene — chunks of a few blocks of non-stationary audio — is not: hinge_span_exceeded)
constexpr int kGateQuietShift = 8;
__host__ __device__ __forceinline__ bool hinge_span_exceeded(unsigned word, int M) {
  const int limit = M / 64 > 4 ? M / 64 : 4;
  return (word & kGateHingeUnsafe) != 0u || (int)(word >> kGateQuietShift) > limit;
}
static __global__ void __launch_bounds__(256)
k_hinge_gate(const unsigned *obj_level, int cap, int M, const unsigned *level_cur, unsigned *gate, bool wide, bool count_quiet) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  bool unsafe = false, quiet = false;
  if (m < M) {
    const unsigned hi = obj_level[m], lo = obj_level[cap + m], call = *level_cur;
    const bool exact_anyway = hi != 0u && call != 0u && (int)(hi >> 23) < (int)(call >> 23) - kHingeSpreadBinades;
    unsafe = !exact_anyway && lo != 0u && call != 0u && (int)(lo >> 23) < (int)(call >> 23) - kHingeSpreadBinades;
    quiet = exact_anyway;
  }
  if (__syncthreads_or(unsafe ? 1 : 0) && threadIdx.x == 0) atomicOr(gate, kGateHingeUnsafe);
  if (count_quiet) {
    const int nq = __syncthreads_count(quiet ? 1 : 0);
    if (nq > 0 && threadIdx.x == 0) atomicAdd(gate, (unsigned)nq << kGateQuietShift);
  }
  // bit 0 of the same word, when `wide` (else the split-operand kernels behind this launch run their wide form only): the form
  // they need — wide when some object falls more than kPlainBinades below the call's level at some probed instant (k_seg_prep)
  if (wide) {
    bool wd = false;
    if (m < M) {
      const unsigned lo = obj_level[cap + m], call = *level_cur;
      wd = lo != 0u && call != 0u && (int)(lo >> 23) < (int)(call >> 23) - kPlainBinades;
    }
    if (__syncthreads_or(wd ? 1 : 0) && threadIdx.x == 0) atomicOr(gate, 1u);
  }
}

// kPrepRun consecutive tiles per thread: 2 for up to 2047 tiles (headline: K0 0.020 -> 0.018 ms), 4 beyond
// (ADM scene, 4096 tiles of 128 samples: K0 + K0s 0.112 -> 0.092 ms)
// What the level probe (a kernel of its own, ahead of this one: complete and visible here) means for the descriptors:
// objects it found "quiet" (level_is_quiet) get kSegQuiet in every tile, which sends them through the split-operand
// kernels' exact path (and their tiles' words in tile_slow raised); *wide is raised when some object FALLS more than
// kPlainBinades below the call's level at some probed instant (zero before the launch; every writer writes 1) — an
// object that is loud at some instants and 60 dB down at others needs the scaled low pieces there.  (Round 3 did this in
// a launch of its own behind this kernel, k_mark_quiet: the probe ran inside this kernel then.)
struct QuietMark {
  const unsigned *obj_level = nullptr;  // [2][cap]: largest / smallest non-zero magnitude probed per object; nullptr: no probe
  int cap = 0;
  const unsigned *level = nullptr;      // the call's level
  unsigned *wide = nullptr;
};
template <int kPrepRun>
static __global__ void __launch_bounds__(256)
k_seg_prep(PointStore ps, int M, int ntiles, int tile_samples, int64_t t_call, int64_t t_call_end,
           SegDesc *desc, QuietMark qm, unsigned *tile_slow = nullptr) {
  // a thread searches the segment of its object at its FIRST tile and walks on from there for the
  // next kPrepRun - 1 (the index only grows): a workgroup covers 16 objects x 16 runs of tiles
  __shared__ SegDesc sh[16 * kPrepRun][17];
  const int ti = threadIdx.x & 15, oi = threadIdx.x >> 4;
  const int tile0 = (blockIdx.x * 16 + ti) * kPrepRun, m = blockIdx.y * 16 + oi;
  bool quiet = false;
  if (qm.obj_level && m < M) {
    const unsigned call = *qm.level, lv = qm.obj_level[m], lo = qm.obj_level[qm.cap + m];
    quiet = level_is_quiet(lv, call);
    if (qm.wide && blockIdx.x == 0 && ti == 0 && lo != 0u && call != 0u && (int)(lo >> 23) < (int)(call >> 23) - kPlainBinades)
      atomicOr(qm.wide, 1u);
  }
  if (tile0 < ntiles && m < M) {
    // (round 6: offset, count and the curve's end points in ONE header load, the search and the walk on the packed records —
    // the lines describe_segment_rec reads next — instead of four dependent trips: header words, end points, a window of
    // times, the records)
    const ObjHdr hd = ps.hdr[m];
    const int base = hd.off, n = hd.cnt;
    int k = 0;
#pragma unroll
    for (int j = 0; j < kPrepRun; j++) {
      const int tile = tile0 + j;
      if (tile >= ntiles) break;
      const int64_t t0 = t_call + (int64_t)tile * tile_samples;
      int64_t t_end = t0 + tile_samples;
      if (t_end > t_call_end) t_end = t_call_end;
      if (j == 0) {
        k = upper_bound_rec_window(ps.rec + base, n, hd.first, hd.last, t0);
      } else {
        while (k < n && ps.rec[base + k].time <= t0) k++;  // = upper_bound_time(.., t0), from the previous tile's
      }
      SegDesc d = describe_segment_rec(ps, base, n, k, t0, t_end);
      if (quiet) d.info |= kSegQuiet;
      sh[ti * kPrepRun + j][oi] = d;
      // (f16x2 gain kernel: a curve point inside the tile — or a quiet object — sends the object through its exact path
      // there; tiles without any such object — all of them on block-aligned metadata — skip the scan for them)
      if (tile_slow && (quiet || (d.info & kSegMulti))) atomicOr(&tile_slow[tile], 1u);
    }
  }
  __syncthreads();
  // tile-major, 16 objects (256 bytes) per tile row
#pragma unroll
  for (int j = 0; j < kPrepRun; j++) {
    const int to = (threadIdx.x >> 4) + 16 * j, oo = threadIdx.x & 15;
    const int tile_o = blockIdx.x * 16 * kPrepRun + to, m_o = blockIdx.y * 16 + oo;
    if (tile_o < ntiles && m_o < M) desc[(size_t)tile_o * M + m_o] = sh[to][oo];
  }
}

// ---------------------------------------------------------------------------
// Slot lists (f32 MFMA kernel).  The contraction  bus[col][s] = sum_k c_k(s) * x_{m(k)}(s) *
// G[row(k)][col]  runs over "slots" k: one (input channel, gain row, coefficient
// function) triple per MFMA k index.  A constant piece of a curve is ONE slot
// (c = 1, its row), a ramp piece two (c = 1 - p on the start row, c = p on the end
// row, gain_interpolator.hpp:272-274), with c(s) = alpha + beta * s for the samples
// [r0, r1) of the tile and 0 elsewhere.  k_slot_list writes, per tile and in object
// order (deterministic), the slots of the pieces that cover the whole tile ("plain",
// no masking needed) and those of the pieces that do not ("masked": curve points
// inside the tile, the partial last tile of a call).
struct Slot {
  uint32_t mr;   // object m | r0 << 16 | (r1 - 1) << 24
  int32_t row;   // gain row
  float alpha, beta;
};
static_assert(sizeof(Slot) == 16, "Slot is loaded as one dwordx4");
constexpr int kMaxSlotObjects = 1 << 16;
constexpr int kPlainSlots = 2;    // per object and tile: at most one whole-tile ramp
constexpr int kMaskedSlots = 6;   // per object and tile on average; beyond: generic path
constexpr int kTileSlots = kPlainSlots + kMaskedSlots;
struct SlotLists {
  Slot *slots;   // [ntiles][kTileSlots*M]: plain slots from 0, masked slots from kPlainSlots*M
  int *count;    // [ntiles][4]: plain, masked, overflowed objects, -
  int *ovf;      // [ntiles][M]: objects whose masked slots did not fit (generic path)
  int M;
};
// 16-byte units of a buffer holding the descriptors (SegDesc[ntiles][M]) and, behind
// them, the slot lists of `ntiles` tiles of M objects
__host__ __device__ inline size_t desc_units(size_t M, size_t ntiles) {
  return (1 + kTileSlots) * M * ntiles + (16 * ntiles + 4 * M * ntiles + 15) / 16 + 1;
}

// K0s: one workgroup per tile, threads over objects, behind k_seg_prep: turns the
// tile's descriptors (coalesced reads, no searching) into its slot lists.  Offsets
// come from ordered scans over the objects, so the lists are deterministic.
static __global__ void __launch_bounds__(256)
k_slot_list(PointStore ps, int M, int tile_samples, int64_t t_call, int64_t t_call_end,
            const SegDesc *desc, SlotLists sl) {
  __shared__ unsigned wsum[4];
  __shared__ int base_p, base_m, base_o, written_m, any_over;
  const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t t0 = t_call + (int64_t)tile * tile_samples;
  int64_t t_end = t0 + tile_samples;
  if (t_end > t_call_end) t_end = t_call_end;
  Slot *plain = sl.slots + (size_t)tile * kTileSlots * M;
  Slot *masked = plain + kPlainSlots * (size_t)M;
  int *ovf = sl.ovf + (size_t)tile * M;
  if (tid == 0) {
    base_p = 0;
    base_m = 0;
    base_o = 0;
    written_m = 0;
  }
  __syncthreads();
  // the pieces of object m inside the tile, first one described by d (k_seg_prep), the
  // others found by walking on (GainInterpolator::process, gain_interpolator.hpp:58-86);
  // op / om == nullptr: count only
  // nx[0..1]: descriptors of the 2nd and 3rd segment, kept in registers between the
  // counting and the writing pass (fill: written by this call; else: read)
  auto walk = [&](int m, SegDesc dk, Slot *op, Slot *om, int &cp, int &cm, SegDesc (&nx)[2], bool fill) {
    int base = 0, n = 0;
    if (dk.info & kSegMulti) {
      base = ps.off[m];
      n = ps.cnt[m];
    }
    int k = seg_k(dk.info), cur = 0, step = 0;
    cp = 0;
    cm = 0;
    for (;;) {
      const int r1 = seg_r1(dk.info);
      if (r1 > cur) {  // duplicate times make empty segments (steps)
        const bool ramp = dk.info & kSegRamp;
        const bool whole = cur == 0 && r1 == tile_samples;
        Slot a;
        a.mr = (uint32_t)m | ((uint32_t)cur << 16) | ((uint32_t)(r1 - 1) << 24);
        a.row = dk.row;
        const float p0 = (float)dk.d0 * dk.scale;  // p(s) = (float)(d0 + s) * scale, :272
        a.alpha = ramp ? 1.0f - p0 : 1.0f;
        a.beta = ramp ? -dk.scale : 0.0f;
        Slot *o = whole ? op : om;
        int &c = whole ? cp : cm;
        if (o) o[c] = a;
        c++;
        if (ramp) {
          Slot b = a;
          b.row = dk.row + 1;
          b.alpha = p0;
          b.beta = dk.scale;
          if (o) o[c] = b;
          c++;
        }
        cur = r1;
      }
      if (!(dk.info & kSegMulti)) break;
      k++;
      if (step < 2 && !fill) {
        dk = nx[step];
      } else {
        dk = describe_segment(ps, base, n, k, t0, t_end);
        if (step < 2) nx[step] = dk;
      }
      step++;
    }
  };
  // inclusive scan over the 256 threads; total returned through `total`
  auto block_scan = [&](unsigned v, unsigned &total) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned u = __shfl_up(v, o, 64);
      if (lane >= o) v += u;
    }
    if (lane == 63) wsum[wv] = v;
    __syncthreads();
    unsigned pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const unsigned x = wsum[w];
      if (w < wv) pre += x;
      tot += x;
    }
    __syncthreads();
    total = tot;
    return v + pre;
  };
  for (int mb = 0; mb < M; mb += 256) {
    const int m = mb + tid;
    SegDesc d, nx[2];
    d.info = 0;
    int cp = 0, cm = 0;
    if (m < M) {
      if (desc) {
        d = desc[(size_t)tile * M + m];
      } else {  // no descriptor pass: find the segment here
        const int base = ps.off[m], n = ps.cnt[m];
        d = describe_segment(ps, base, n, upper_bound_time_window(ps.time + base, n, t0), t0, t_end);
      }
      walk(m, d, nullptr, nullptr, cp, cm, nx, true);
    }
    // cp <= 2, cm <= 2 * 257 per object: 10 + 18 bits of one word per 256 objects
    unsigned last;
    const unsigned incl = block_scan((unsigned)cp | ((unsigned)cm << 10), last);
    const int off_p = base_p + (int)(incl & 1023u) - cp;
    const int off_m = base_m + (int)(incl >> 10) - cm;
    const bool over = cm > 0 && off_m + cm > kMaskedSlots * M;
    if (tid == 0) any_over = 0;
    __syncthreads();
    if (over) any_over = 1;
    __syncthreads();
    unsigned oincl = 0, olast = 0;
    if (any_over) oincl = block_scan(over ? 1u : 0u, olast);  // (uniform branch)
    if (m < M) {
      if (over) {
        ovf[base_o + (int)oincl - 1] = m;
      } else if (cp + cm > 0) {
        int a, b;
        walk(m, d, plain + off_p, masked + off_m, a, b, nx, false);
        if (cm > 0) atomicMax(&written_m, off_m + cm);
      }
    }
    __syncthreads();
    if (tid == 0) {
      base_p += (int)(last & 1023u);
      base_m += (int)(last >> 10);
      base_o += (int)olast;
    }
    __syncthreads();
  }
  if (tid == 0) {
    // (once an object's masked slots do not fit, none of the later ones do)
    sl.count[tile * 4 + 0] = base_p;
    sl.count[tile * 4 + 1] = written_m;
    sl.count[tile * 4 + 2] = base_o;
  }
}

struct GainMixParams {
  const float *in;      // [M][in_stride] planar input, sample 0 = call start
  size_t in_stride;
  float *out;           // [part][col][out_stride]
  size_t out_stride;
  size_t part_stride;   // floats between the slabs of two grid-level splits
  const SegDesc *desc;  // [ntiles][M], tile-major
  PointStore ps;
  int64_t t_call;       // absolute sample time of sample 0
  int nsamples;         // samples in this call
  int ntiles;
  int M;
  int ncols;            // valid output columns (<= row)
  int ngroups;          // column groups per workgroup (waves: group-major)
  int wsplit;           // object splits inside a workgroup
  int tiles_per_wg;     // MFMA kernel: adjacent tiles handled by one workgroup
  int vec_ok;           // in/out rows are 16-byte aligned: vector accesses allowed
  SlotLists sl;         // f32 MFMA kernel: the tile's slot lists (k_slot_list)
  // split-operand kernels: the context's mode word of THIS call (bit 0: wide form, bit 1: the hinge kernel handed the call
  // to the piece lists) and where the kernel that does the call leaves a copy of it for the renderer's queries
  // (earhip_render_wide_form / _hinge_standby): the word itself is the context's and is cleared two calls later
  int tile_runs = 0;    // grid kernel: a workgroup's tiles are a contiguous run (else every gridDim.x-th tile)
  const unsigned *mode_word = nullptr;
  unsigned *record = nullptr;
};
constexpr unsigned kModeRecorded = 4u;  // bit 2 of a recorded mode word: "a split-operand kernel wrote this"
// what the kernel that does a call leaves in P.record (one thread of the grid; `device_form`: the form was picked on the device)
__device__ __forceinline__ void record_mode(const GainMixParams &P, bool device_form) {
  if (P.record) *P.record = (P.mode_word ? *P.mode_word : 0u) | (device_form ? 0u : 1u) | kModeRecorded;
}

// accumulate one segment piece of one object into acc
template <int NOUT, int SPL, bool STRICT>
__device__ __forceinline__ void accumulate_piece(float (&acc)[NOUT][SPL], const float (&xs)[SPL],
                                                 const SegDesc d, const float *__restrict__ rows,
                                                 int rowlen, int lane, int col0, int bus_cols) {
  const bool ramp = d.info & kSegRamp;
  const float *__restrict__ S = rows;
  const float *__restrict__ E = rows + (ramp ? rowlen : 0);
  float p[SPL], q[SPL];
#pragma unroll
  for (int i = 0; i < SPL; i++) {
    p[i] = (float)(d.d0 + lane * SPL + i) * d.scale;  // gain_interpolator.hpp:193,224,272
    q[i] = 1.0f - p[i];
  }
  if (STRICT) {
#pragma unroll
    for (int j = 0; j < NOUT; j++) {
      const float s = S[j], e = E[j];
      // constant segment, or a bus whose two points compare equal: libear's
      // apply_constant with the segment's end point (gain_interpolator.hpp:68-75)
      const bool cflat = !ramp || ((d.info >> (2 + (col0 + j >= bus_cols ? 1 : 0))) & 1);
#pragma unroll
      for (int i = 0; i < SPL; i++) {
        const float g = cflat ? e : q[i] * s + p[i] * e;  // un-contracted (-ffp-contract=off)
        acc[j][i] = acc[j][i] + xs[i] * g;
      }
    }
  } else {
    float a[SPL], b[SPL];
#pragma unroll
    for (int i = 0; i < SPL; i++) {
      a[i] = ramp ? xs[i] * q[i] : xs[i];
      b[i] = xs[i] * p[i];
    }
#pragma unroll
    for (int j = 0; j < NOUT; j++) {
      const float s = S[j];
#pragma unroll
      for (int i = 0; i < SPL; i++) acc[j][i] = __builtin_fmaf(a[i], s, acc[j][i]);
    }
    if (ramp) {
#pragma unroll
      for (int j = 0; j < NOUT; j++) {
        const float e = E[j];
#pragma unroll
        for (int i = 0; i < SPL; i++) acc[j][i] = __builtin_fmaf(b[i], e, acc[j][i]);
      }
    }
  }
}

// K1.  grid = (ntiles, grid-level object splits, column super-groups)
// block = 64 * ngroups * wsplit threads; tile = 64 * SPL samples.
template <int NOUT, int SPL, bool STRICT>
__global__ void __launch_bounds__(512) k_gain_mix(GainMixParams P) {
  constexpr int TS = 64 * SPL;
  typedef float vec_t __attribute__((ext_vector_type(SPL)));
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [ngroups][NOUT][TS]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = wave % P.ngroups;           // column group inside the workgroup
  const int ws = wave / P.ngroups;          // object split inside the workgroup
  const int tile = blockIdx.x;
  const int nparts = P.wsplit * gridDim.y;
  const int part = blockIdx.y * P.wsplit + ws;
  const int m_lo = (int)(((int64_t)P.M * part) / nparts);
  const int m_hi = (int)(((int64_t)P.M * (part + 1)) / nparts);
  const int col0 = (blockIdx.z * P.ngroups + g) * NOUT;

  const int s0 = tile * TS + lane * SPL;  // first sample of this lane
  const int64_t tile_t0 = P.t_call + (int64_t)tile * TS;
  int64_t tile_t1 = tile_t0 + TS;
  if (tile_t1 > P.t_call + P.nsamples) tile_t1 = P.t_call + P.nsamples;
  const bool full = P.vec_ok && s0 + SPL - 1 < P.nsamples;  // whole-vector stores allowed

  float acc[NOUT][SPL];
#pragma unroll
  for (int j = 0; j < NOUT; j++)
#pragma unroll
    for (int i = 0; i < SPL; i++) acc[j][i] = 0.0f;

  // Input tile of object m for this lane, branch-free so the load can be hoisted:
  // lanes past the end of the call re-read the last valid vector (clamped
  // address) and are zeroed by selects.
  const int nvec = (P.nsamples + SPL - 1) / SPL * SPL;
  const int s0c = min(s0, nvec - SPL);
  const int last_s = P.nsamples - 1;

  for (int m = m_lo; m < m_hi; m++) {
    SegDesc d = P.desc[(size_t)tile * P.M + m];
    const float *xrow = P.in + (size_t)m * P.in_stride;
    float x[SPL];
    if (P.vec_ok) {  // rows aligned, stride % 4 == 0: the vector at s0c is in bounds
      const vec_t v = *reinterpret_cast<const vec_t *>(xrow + s0c);
#pragma unroll
      for (int i = 0; i < SPL; i++) x[i] = v[i];
    } else {
#pragma unroll
      for (int i = 0; i < SPL; i++) x[i] = xrow[min(s0 + i, last_s)];
    }
#pragma unroll
    for (int i = 0; i < SPL; i++) x[i] = s0 + i < P.nsamples ? x[i] : 0.0f;

    // Usually the whole tile lies inside one curve segment and the loop body
    // runs once.  If a curve point falls inside the tile, walk the segments
    // like GainInterpolator::process (gain_interpolator.hpp:58-86), masking the
    // lanes outside each piece.
    int64_t cur = tile_t0;
    for (;;) {
      float xm[SPL];
#pragma unroll
      for (int i = 0; i < SPL; i++) xm[i] = x[i];
      int64_t seg_end = tile_t1;
      if (d.info & kSegMulti) {
        seg_end = tile_t0 + seg_r1(d.info);
        const int r0 = (int)(cur - tile_t0), r1 = seg_r1(d.info);
#pragma unroll
        for (int i = 0; i < SPL; i++)
          xm[i] = (lane * SPL + i >= r0 && lane * SPL + i < r1) ? x[i] : 0.0f;
      } else if (cur != tile_t0) {
        const int r0 = (int)(cur - tile_t0);
#pragma unroll
        for (int i = 0; i < SPL; i++) xm[i] = lane * SPL + i >= r0 ? x[i] : 0.0f;
      }
      if (seg_end > cur)  // duplicate times make empty segments (steps)
        accumulate_piece<NOUT, SPL, STRICT>(acc, xm, d,
                                            P.ps.gain + (size_t)d.row * P.ps.row + col0,
                                            P.ps.row, lane, col0, P.ps.bus_cols);
      if (!(d.info & kSegMulti)) break;
      cur = seg_end;
      const int base = P.ps.off[m], n = P.ps.cnt[m];
      d = describe_segment(P.ps, base, n, seg_k(d.info) + 1, tile_t0, tile_t1);
    }
  }

  // combine the in-workgroup object splits through LDS, highest split first
  float *slab = lds + (size_t)g * NOUT * TS + lane * SPL;
  for (int r = P.wsplit - 1; r >= 1; r--) {
    if (ws == r) {
#pragma unroll
      for (int j = 0; j < NOUT; j++) {
        vec_t v;
#pragma unroll
        for (int i = 0; i < SPL; i++) v[i] = acc[j][i];
        if (r != P.wsplit - 1) v += *reinterpret_cast<const vec_t *>(slab + j * TS);
        *reinterpret_cast<vec_t *>(slab + j * TS) = v;
      }
    }
    __syncthreads();
  }
  if (ws != 0) return;

  float *op = P.out + (size_t)blockIdx.y * P.part_stride + s0;
#pragma unroll
  for (int j = 0; j < NOUT; j++) {
    if (col0 + j >= P.ncols) break;
    vec_t v;
#pragma unroll
    for (int i = 0; i < SPL; i++) v[i] = acc[j][i];
    if (P.wsplit > 1) v += *reinterpret_cast<const vec_t *>(slab + j * TS);
    float *o = op + (size_t)(col0 + j) * P.out_stride;
    if (full) {
      *reinterpret_cast<vec_t *>(o) = v;
    } else {
#pragma unroll
      for (int i = 0; i < SPL; i++)
        if (s0 + i < P.nsamples) o[i] = v[i];
    }
  }
}

// sum grid-level partial slabs: out[c][s] = sum_p part[p][c][s]  (when the buses go straight
// to the caller, and ahead of the wave decorrelator kernel in block mode; out may be slab 0)
static __global__ void k_sum_parts(const float *parts, size_t part_stride, int nparts,
                            size_t row_stride, int ncols, int nsamples, float *out,
                            size_t out_stride) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (s >= nsamples || c >= ncols) return;
  const float *q = parts + (size_t)c * row_stride + s;
  float v = q[0];
  int p = 1;
  for (; p + 3 < nparts; p += 4) {  // four loads in flight, summed part after part
    const float a = q[(size_t)p * part_stride], b = q[(size_t)(p + 1) * part_stride];
    const float c2 = q[(size_t)(p + 2) * part_stride], d = q[(size_t)(p + 3) * part_stride];
    v = (((v + a) + b) + c2) + d;
  }
  for (; p < nparts; p++) v += q[(size_t)p * part_stride];
  out[(size_t)c * out_stride + s] = v;
}

}  // namespace earhip
