// gain_hg.h — K1 on the f16 matrix cores for gain curves that RAMP ALL THE TIME with points that ignore the tile grid
// (ADM blocks whose interpolationLength is the block duration: every object is always on its way to its next
// target), the "hinge kernel".
//
// libear evaluates a curve segment by segment (GainInterpolator::process, include/ear/dsp/gain_interpolator.hpp:
// 53-87; a ramp is g(s) = (1 - p) S + p E with p = (float)(s - start) * (1.0f / (float)(end - start)), :264-277).
// A curve through its points is continuous and piecewise linear, so inside a tile of T samples that starts at s0
//
//     g(s) = L(s) + sum_k relu(+-(s - r_k)) * D_k ,      D_k = (slope after r_k) - (slope before r_k),
//
// where L is the LINE of the segment that holds the tile's centre, extended over the whole tile, and the sum runs
// over the curve points r_k inside the tile ("kinks"): a point behind the centre opens FORWARD (relu(s - r)), a point
// in front of it BACKWARD (relu(r - s)) — both with the same D.  The piece-list kernel (gain_p2.h) spends one operand
// split of the inputs per ramp and tile (3.07 per object and tile when a new target arrives every 240 samples on
// 256-sample tiles), which makes it issue-bound (profiles/r03_moving_scene_*: 271 M VALU against 29 M MFMA
// instructions, 0.38 of the HBM roofline).  Here
//
//   * L costs what it costs the grid kernel (gain_h2.h): ONE split of the inputs, two gain operands B0 = L(s0) and
//     B1 = the slope, and (s - s0) applied to the accumulators of the B1 products at the end of the tile;
//   * a kink costs no split at all: its factor F(s) = max(+-(s - r) / 16, 0) is exact in f16 (9 bits), so the
//     operand pieces of x F come out of the pieces (xh, xl) the line's split left in registers by PACKED f16
//     arithmetic: Ah = rn(F xh), the exact residual fma(F, xh, -Ah), plus F xl — five v_pk instructions per
//     register of two operands instead of the ~20 of a position factor and a split from f32;  the gain operand of
//     the kink is 16 D, one MFMA set (the line has two) — a property of the curve POINT, made when the curves are
//     committed (k_kink_rows: a "kink row" per point beside its gain row, already split into f16 pieces);
//   * the objects of a tile are ordered by where their kinks lie, so that whole chunks of 32 have their kinks in the
//     same part of the tile, and a wave (64 samples) skips the kink sets that cannot reach its samples: a forward
//     kink in the last 64 samples only costs the wave of those.
//
// Two forms: 8 waves on 512-sample tiles with up to TWO kinks on either side of the tile's centre (the default), 4
// waves on 256-sample tiles with one.  Bigger tiles ask for fewer rows per sample (25.7 M instead of 29.2 M L2 -> L1
// requests on the always-ramping scene) but a kink reaches further (relu runs to the tile's end: twice the kink MFMAs
// per sample): the two come out even, 0.75-0.77 ms, and the lists of half as many tiles are built faster.
//
// What qualifies: an (object, tile) pair whose curve points inside the tile are at most two (one) on either side of
// the centre, with every ramp that meets such a point at least kHingeMinLen long (so that 16 x slope differences stay
// within the gains' range: the operands are scaled to half the f16 range the other split kernels use).  Everything
// else — more points in one half of a tile, short ramps, steps, quiet objects — takes the exact per-object path;
// curve sets made of such things are the piece-list kernel's (plan_mix decides from the curves).
//
// Lists (k_hinge_build): per tile the objects sorted by (first wave their forward kink reaches, last wave their
// backward kink reaches), those with second kinks behind them, padded to a multiple of 32 at the end only; per slot a
// LinEntry and, in chunks that have any kinks, a HingeEntry (the factors of its up to four kinks); per chunk a flag
// word (per kink set the waves it can reach).
#pragma once

#include <hip/hip_runtime.h>

#include "gain_p2.h"

namespace earhip {

constexpr int kHingeTile = 256;     // samples of the smaller form's tile (what the curve statistics count in)
constexpr int kHingeMinLen = 128;   // ramps that meet a kink are at least this long
constexpr int kMaxHingeCached = 8192;  // (object, tile) pairs a list-building workgroup keeps between its two passes (16 B each, LDS); also the most objects
// The factor of a kink is kept as 16 (s - r) / T, its gain operand as T D / 16: with F in [1 / 256, 1] the product F x of a quiet
// object's input would be a subnormal f16 (x sits 2^-10 below the call's level at -100 dB; the packed-f16 products have no
// wider intermediate), 16 F x stays normal — and its residual exactly representable — down to 17 binades below the level.
// Quieter objects take the exact path (kHingeQuietBinades; the other split kernels: 20).  The price: inputs more than
// 24 dB above the probed level (instead of 48) overflow the operands and send their tile through the exact redo.
constexpr float kHingeFactorScale = 16.0f;
constexpr int kHingeQuietBinades = kHingeSpreadBinades;  // (gain_kernels.h: the same span decides which calls this kernel takes at all)  // (object, tile) pairs a list-building workgroup keeps between its two passes

struct LinEntry {
  uint32_t m;   // object | kLinD2 | which kinks the pair has (kLinF1 ..) | kLinNull
  int32_t row;  // gain row R1 of the point the centre segment starts at (kc - 1, clipped into the object's points); its end
                // (point kc) is row R1 + 1 when kLinD2 is set, else R1 again (no such point: constant)
  float p0;     // libear's p at the tile's centre sample c, (float)(c - start) * scale (gain_interpolator.hpp:272); 0: constant
  float scale;  // 1.0f / (float)(end - start) of the centre segment; 0: constant
};
// the factors of a pair's kinks: f16 pairs (slope of F: +-1 / kHingeFactorScale, offset of F: -+(r - s0) / kHingeFactorScale),
// F(s) = max(s * slope + offset, 0), s counted from the tile start; 0: no such kink (F = 0).  fac[0]: the first kink behind
// the centre (forward), [1]: the first one in front of it (backward), [2], [3]: the second ones (512-sample tiles)
struct HingeEntry {
  uint32_t fac[4];
};
static_assert(sizeof(LinEntry) == 16 && sizeof(HingeEntry) == 16, "list entries are loaded as one dwordx4");
constexpr uint32_t kLinNull = 1u << 31, kLinObjMask = 0xffffu;
constexpr uint32_t kLinD2 = 1u << 16;
constexpr int kLinKinkShift = 17;  // bits 17 .. 20: kinks 0 .. 3 (the order of HingeEntry::fac)

struct HingeLists {
  LinEntry *lin;     // [ntiles][cap]
  HingeEntry *hinge; // [ntiles][cap]
  uint32_t *cflags;  // [ntiles][cap / 32]: byte g: the waves kink set g of the chunk reaches (0: the chunk has no such kinks)
  int *count;        // [ntiles][4]: chunks, exact-path objects
  int *ovf;          // [ntiles][M]: objects that take the exact per-object path
  int M, cap;
};
__host__ __device__ inline int hinge_cap(int M) { return (M + 31) & ~31; }
// 16-byte units of the scratch buffer of a call of `ntiles` tiles
__host__ __device__ inline size_t hinge_units(size_t M, size_t ntiles) {
  const size_t cap = (size_t)hinge_cap((int)M);
  return 2 * cap * ntiles + (4 * (cap / 32) * ntiles + 16 * ntiles + 4 * M * ntiles + 15) / 16 + 4;
}
inline HingeLists hinge_lists(void *scratch, int M, int ntiles) {
  HingeLists hl;
  hl.M = M;
  hl.cap = hinge_cap(M);
  hl.lin = reinterpret_cast<LinEntry *>(scratch);
  hl.hinge = reinterpret_cast<HingeEntry *>(hl.lin + (size_t)hl.cap * ntiles);
  hl.cflags = reinterpret_cast<uint32_t *>(hl.hinge + (size_t)hl.cap * ntiles);
  hl.count = reinterpret_cast<int *>(hl.cflags + (size_t)(hl.cap / 32) * ntiles);
  hl.ovf = hl.count + (size_t)4 * ntiles;
  return hl;
}

// input rows and gain rows are addressed with 32-bit byte offsets from their first row
inline bool hinge_addressable(size_t M, size_t in_stride, size_t nsamples, size_t gain_rows, size_t rowlen) {
  return ((M - 1) * in_stride + nsamples + 4) * sizeof(float) < ((size_t)1 << 32) && gain_rows * rowlen * sizeof(float) < ((size_t)1 << 32);
}
__device__ __forceinline__ uint32_t f16_bits(float v) { return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)v); }

// ---------------------------------------------------------------------------
// K0.  What one (object, tile) pair is for the hinge kernel.  NW = waves of the kernel's workgroup: tiles of 64 NW samples;
// a pair may have NW / 4 kinks on either side of the centre (one on 256-sample tiles, two on 512).
// (sort classes of a pair: 9 (4 waves) or 25 (8 waves) by the waves its first kinks reach, then "has second kinks", then "exact")
template <int NW>
struct HgClasses {
  static constexpr int N = NW == 4 ? 16 : 32, kSecond = N - 2, kExact = N - 1;
};
// the six records around the centre: points kc - 3 .. kc + 2 (missing ones: never looked at)
struct HingeRecs {
  PointRec r[6];
  bool has[6];
};
__device__ __forceinline__ HingeRecs hinge_load(const PointStore &ps, int base, int n, int kc) {
  HingeRecs R;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const int k = kc - 3 + i;
    R.has[i] = k >= 0 && k < n;
    R.r[i] = ps.rec[base + min(max(k, 0), n - 1)];
  }
  return R;
}
struct HingePair {
  int cls;          // sort class of the pair: (first wave its forward kink reaches, last wave its backward kink reaches),
                    // HgClasses::kSecond: it has second kinks, kExact: not a pair for this kernel
  uint32_t kinks;   // bit g: kink g (HingeEntry::fac's order)
  int pos[4];       // their places, samples from the tile start
};
// the waves kink set g of a pair of class `cls` can reach (a bound, exact for the classes below kHgSecond)
template <int NW>
__host__ __device__ inline uint32_t hinge_class_masks(int cls) {
  constexpr int H = NW / 2;
  const uint32_t all = (1u << NW) - 1, upper = all & ~((1u << H) - 1), lower = (1u << H) - 1;
  if (cls == HgClasses<NW>::kSecond) return upper | lower << 8 | upper << 16 | lower << 24;
  const int wf = cls / (H + 1) + H, wb = cls % (H + 1) - 1;  // wf = NW: no forward kink, wb = -1: no backward kink
  const uint32_t mf = wf < NW ? all & ~((1u << wf) - 1) : 0u, mb = wb >= 0 ? (1u << (wb + 1)) - 1 : 0u;
  return mf | mb << 8;
}
// t0, t1: the tile [t0, t1) (t1 clipped to the end of the call)
template <int NW>
__device__ __forceinline__ HingePair hinge_classify(const PointStore &ps, const HingeRecs &R, int64_t t0, int64_t t1) {
  constexpr int T = 64 * NW, H = NW / 2, KMAX = NW / 4;
  const uint32_t allflat = (1u << ps.nbus) - 1;
  // the segment ending at point i (i = 1..5) is flat when the point repeats its predecessor on every bus; the stretches
  // before the first and behind the last point are constant (gain_interpolator.hpp:68-75)
  auto seg_flat = [&](int i) { return !R.has[i] || !R.has[i - 1] || (R.r[i].flat & allflat) == allflat; };
  auto seg_len = [&](int i) { return R.r[i].time - R.r[i - 1].time; };
  HingePair hp;
  hp.kinks = 0;
  bool simple = true;
  // a kink at point i (1..4): at least one of the segments that meet there ramps, and ramps that meet a kink are long
  auto kink_at = [&](int i) {
    const bool bf = seg_flat(i), af = seg_flat(i + 1);
    if (bf && af) return false;  // constant on both sides
    if (!bf && seg_len(i) < kHingeMinLen) simple = false;
    if (!af && seg_len(i + 1) < kHingeMinLen) simple = false;
    return true;
  };
  // points in (t0, centre]: point 2 (the start of the centre segment), then 1; a third one: not for this kernel
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int i = 2 - j;
    if (!(R.has[i] && R.r[i].time > t0)) break;
    if (j >= KMAX) {
      simple = false;
      break;
    }
    const int pos = (int)(R.r[i].time - t0);
    // relu(r - s) > 0 up to s = r - 1
    if (kink_at(i) && pos >= 1) hp.kinks |= 1u << (1 + 2 * j), hp.pos[1 + 2 * j] = pos;
  }
  // points in (centre, t1): point 3 (the end of the centre segment), then 4
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int i = 3 + j;
    if (!(R.has[i] && R.r[i].time < t1)) break;
    if (j >= KMAX) {
      simple = false;
      break;
    }
    const int pos = (int)(R.r[i].time - t0);
    // relu(s - r) > 0 from s = r + 1 on
    if (kink_at(i) && pos + 1 < T) hp.kinks |= 1u << (2 * j), hp.pos[2 * j] = pos;
  }
  if (!simple) {
    hp.cls = HgClasses<NW>::kExact;
  } else if (hp.kinks & 0xcu) {
    hp.cls = HgClasses<NW>::kSecond;
  } else {
    const int wf = (hp.kinks & 1u) ? (hp.pos[0] + 1) / 64 : NW, wb = (hp.kinks & 2u) ? (hp.pos[1] - 1) / 64 : -1;
    hp.cls = (max(wf, H) - H) * (H + 1) + (min(wb, H - 1) + 1);
  }
  return hp;
}
// the two list entries of a pair
__device__ __forceinline__ void hinge_entries(const PointStore &ps, const HingeRecs &R, int base, int n, int kc, int m, int64_t t0,
                                              const HingePair &hp, int half, LinEntry &e, HingeEntry &h) {
  const uint32_t allflat = (1u << ps.nbus) - 1;
  const int k1 = min(max(kc - 1, 0), n - 1);  // the point R1 stands for
  const bool d2 = kc < n && kc > k1;
  e.m = (uint32_t)m | (d2 ? kLinD2 : 0u) | hp.kinks << kLinKinkShift;
  e.row = base + k1;
  e.p0 = 0.0f;
  e.scale = 0.0f;
  if (R.has[3] && R.has[2] && (R.r[3].flat & allflat) != allflat) {  // (constant on every bus: E == S bit for bit, nothing to interpolate)
    e.scale = R.r[3].scale;
    // (the line is anchored at the tile's CENTRE — a sample of its own segment: libear's p there, no extrapolation in the
    // operand; anchored at the tile start its value there ran up to half a tile's slope beyond the gains' range)
    e.p0 = (float)(int32_t)(t0 + half - R.r[2].time) * R.r[3].scale;
  }
  const float slope = 1.0f / kHingeFactorScale;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    h.fac[g] = 0u;
    if (!((hp.kinks >> g) & 1u)) continue;
    if (g & 1) h.fac[g] = f16_bits(-slope) | f16_bits((float)hp.pos[g] * slope) << 16;  // backward: F = (r - s) slope
    else h.fac[g] = f16_bits(slope) | f16_bits(-(float)hp.pos[g] * slope) << 16;          // forward: F = (s - r) slope
  }
}

// ---------------------------------------------------------------------------
// Kink rows.  The gain operand of a kink at curve point k is (slope behind it - slope in front of it) of every column —
// a property of the POINT, not of the tile it is met in — so it is made once per point, when the curves are committed
// (CurveSet::commit, behind the upload), and kept behind the gain rows in the same buffer: row k of the kink image =
// for every column the two f16 pieces (low half h, high half l) of
//     (sA (R[k+1] - R[k]) - sB (R[k] - R[k-1])) x (kink scale of the column),
// sA / sB = 1 / length of the ramp behind / in front of the point (0: constant there, or no such point).  The hinge
// kernel asks for it like for a gain row and only regroups the pieces into its fragments (it used to ask for the rows
// R[k-1], R[k+1] and do this arithmetic and the split per tile and object: a tenth of its time).
// Kink scale: half the column's gain scale x kHingeFactorScale (the factor of a kink is (s - r) / kHingeFactorScale).
__host__ __device__ inline float hinge_kink_scale(float gcol) { return 0.5f * gcol * 16.0f; }
// grid = objects (the listed ones, or all M when `objects` is NULL), block = 256 threads
static __global__ void __launch_bounds__(256)
k_kink_rows(const int32_t *off, const int32_t *cnt, const PointRec *rec, const float *gain, uint32_t *kink, const float *gcol,
            const int32_t *objects, int row, int nbus) {
  const int m = objects ? objects[blockIdx.x] : (int)blockIdx.x;
  const int base = off[m], n = cnt[m];
  const uint32_t allflat = (1u << nbus) - 1;
  for (int i = threadIdx.x; i < n * row; i += 256) {
    const int k = i / row, c = i - k * row;
    // (the ramp that ENDS at point j: rec[j].scale, flat bits of point j)
    const bool rb = k > 0 && (rec[base + k].flat & allflat) != allflat;
    const bool ra = k + 1 < n && (rec[base + k + 1].flat & allflat) != allflat;
    const float sB = rb ? rec[base + k].scale : 0.0f, sA = ra ? rec[base + k + 1].scale : 0.0f;
    const float R1 = gain[(size_t)(base + k) * row + c];
    const float R0 = k > 0 ? gain[(size_t)(base + k - 1) * row + c] : R1;
    const float R2 = k + 1 < n ? gain[(size_t)(base + k + 1) * row + c] : R1;
    const float v = (sA * (R2 - R1) - sB * (R1 - R0)) * hinge_kink_scale(gcol[c]);
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    kink[(size_t)(base + k) * row + c] = (uint32_t)__builtin_bit_cast(uint16_t, h) | (uint32_t)__builtin_bit_cast(uint16_t, l) << 16;
  }
}

struct HingeCached;
// K0 of the hinge lists in two kernels (round 6; gain_p2.h, k_piece_classify has the reasons): the first pass of k_hinge_build —
// a search, six records, the pair's class, kinks and line — OBJECT-major: a wave = one object x 64 consecutive tiles, so the lanes
// of a load instruction ask for neighbouring points of one curve; what the second pass needs of a pair (16 bytes: HingeCached)
// goes through LDS into a staging matrix [tile][object].  k_hinge_build then starts at its counting loop with one coalesced
// load per pair and needs no LDS for what the first pass found (two 1024-thread workgroups per CU instead of one).
// grid = (ceil(ntiles / 64), ceil(M / kClassifyObjects)), block = 64 kClassifyObjects threads.
template <int NW>
__global__ void __launch_bounds__(64 * kClassifyObjects)
k_hinge_classify(PointStore ps, int M, int ntiles, int64_t t_call, int64_t t_call_end, HingeCached *stage, const unsigned *obj_level,
                 const unsigned *level_cur, const unsigned *gate, const unsigned *span);

constexpr int kHingeBuildThreads = 1024;
// grid = ceil(ntiles / TPW) workgroups of 1024 threads; a thread = one (object, tile) pair of a batch of 1024 / TPW
// objects, the tile index fastest (the lanes of an object read neighbouring points: gain_p2.h, k_piece_build).  Two
// passes over the objects: the first finds every pair's centre segment (one search) and class and counts the classes of
// each tile, the second — the classes' places in the list known — ranks the pairs of a class in object order (ballots:
// the lists are deterministic) and writes the entries.  What the first pass found is kept in LDS (dynamic: 16 M TPW
// bytes: segment index and class, the pair's kinks and their places, the line's position and 1 / length — the second pass
// looks at no curve point again).  NW: the waves of the kernel the lists are for (tiles of 64 NW samples).
struct HingeCached {
  uint32_t code;   // kc << 5 | class, bits 23 .. 31: the place of kink 3
  uint32_t kinks;  // bits 0 .. 3: which kinks, then 9 bits each: the places of kinks 0 .. 2
  float p0, scale; // of the LinEntry
};
static_assert(sizeof(HingeCached) == 16, "");
// what the first pass keeps of a pair (k_hinge_build's and k_hinge_classify's: the same expressions)
template <int NW>
__device__ __forceinline__ HingeCached hinge_pair(const PointStore &ps, int m, const ObjHdr &hd, int64_t t0, int64_t t1, const unsigned *obj_level,
                                                  unsigned call_level, int quiet_binades) {
  constexpr int T = 64 * NW, kHgExact = HgClasses<NW>::kExact;
  const int base = hd.off, n = hd.cnt;
  const int kc = upper_bound_rec_window(ps.rec + base, n, hd.first, hd.last, t0 + T / 2);
  const HingeRecs R = hinge_load(ps, base, n, kc);
  const HingePair hp = hinge_classify<NW>(ps, R, t0, t1);
  int cls = hp.cls;
  if (obj_level && obj_level[m] != 0u && call_level != 0u && (int)(obj_level[m] >> 23) < (int)(call_level >> 23) - quiet_binades)
    cls = kHgExact;
  LinEntry e;
  HingeEntry h;
  hinge_entries(ps, R, base, n, kc, m, t0, hp, T / 2, e, h);
  HingeCached hc;
  auto place = [&](int g) { return (hp.kinks >> g) & 1u ? (uint32_t)hp.pos[g] & 511u : 0u; };
  hc.code = (uint32_t)(kc << 5 | cls) | place(3) << 23;  // (classes below 32, kc below 2^18)
  hc.kinks = hp.kinks | place(0) << 4 | place(1) << 13 | place(2) << 22;
  hc.p0 = e.p0, hc.scale = e.scale;
  return hc;
}
template <int NW>
__global__ void __launch_bounds__(64 * kClassifyObjects)
k_hinge_classify(PointStore ps, int M, int ntiles, int64_t t_call, int64_t t_call_end, HingeCached *stage, const unsigned *obj_level,
                 const unsigned *level_cur, const unsigned *gate, const unsigned *span) {
  if (gate && (*gate & kGateHingeUnsafe)) return;  // the piece lists do this call (k_hinge_gate)
  constexpr int T = 64 * NW;
  __shared__ __attribute__((aligned(16))) HingeCached sh[64][kClassifyObjects + 1];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = blockIdx.y * kClassifyObjects + w;  // wave-uniform
  const int tile = blockIdx.x * 64 + lane;
  const int quiet_binades = span && hinge_span_exceeded(*span, M) ? kQuietBinades : kHingeQuietBinades;
  HingeCached hc = {0u, 0u, 0.0f, 0.0f};
  if (m < M && tile < ntiles) {
    const int64_t t0 = t_call + (int64_t)tile * T;
    const int64_t t1 = t0 + T > t_call_end ? t_call_end : t0 + T;
    hc = hinge_pair<NW>(ps, m, ps.hdr[m], t0, t1, obj_level, level_cur ? *level_cur : 0u, quiet_binades);
  }
  sh[lane][w] = hc;
  __syncthreads();
  const int to = threadIdx.x / kClassifyObjects, oo = threadIdx.x % kClassifyObjects;
  const int tile_o = blockIdx.x * 64 + to, m_o = blockIdx.y * kClassifyObjects + oo;
  if (tile_o < ntiles && m_o < M) stage[(size_t)tile_o * M + m_o] = sh[to][oo];
}
template <int TPW, int NW>
__global__ void __launch_bounds__(kHingeBuildThreads)
k_hinge_build(PointStore ps, int M, int ntiles, int64_t t_call, int64_t t_call_end, HingeLists hl, const unsigned *obj_level,
              const unsigned *level_cur, const unsigned *gate, const unsigned *span, const HingeCached *stage = nullptr) {
  // stage != nullptr: the second kernel of the two-kernel K0 — what a pair is comes from k_hinge_classify's matrix (no dynamic LDS)
  if (gate && (*gate & kGateHingeUnsafe)) return;  // the piece lists do this call (k_hinge_gate; gate == NULL: nobody stands by)
  // (span: the kernel's robust form does this call when the word says so — it keeps its precision as far down as the other
  // split-operand kernels, so only objects quieter than THEIR bound take the exact path)
  const int quiet_binades = span && hinge_span_exceeded(*span, M) ? kQuietBinades : kHingeQuietBinades;
  constexpr int OB = kHingeBuildThreads / TPW;  // objects per batch
  constexpr int NWV = kHingeBuildThreads / 64;
  constexpr int T = 64 * NW, NC = HgClasses<NW>::N, kHgExact = HgClasses<NW>::kExact;
  extern __shared__ HingeCached hg_cache[];  // [M][TPW]
  __shared__ int cnt[TPW][NC];              // objects per class
  __shared__ int start[TPW][NC + 2];        // first slot of a class (the listed ones: all but kHgExact), [NC]: listed objects, [NC + 1]: chunks
  __shared__ int run[TPW][NC];              // objects of a class placed by the batches so far
  __shared__ int wcnt[NWV][TPW][NC];        // ... by the waves of this batch
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = tid % TPW, oi = tid / TPW;
  const int tile = blockIdx.x * TPW + j;
  const unsigned call_level = level_cur ? *level_cur : 0u;
  for (int i = tid; i < TPW * NC; i += kHingeBuildThreads) (&cnt[0][0])[i] = 0, (&run[0][0])[i] = 0;
  EARHIP_BUILD_PROF_SLOT
  EARHIP_BUILD_MARK(0);
  const int64_t t0 = t_call + (int64_t)tile * T;
  const int64_t t1 = t0 + T > t_call_end ? t_call_end : t0 + T;
  __syncthreads();
  // ---- pass 1
  for (int mb = 0; mb < M; mb += OB) {
    const int m = mb + oi;
    if (m < M && tile < ntiles) {
      if (stage) {  // (two kernels: the pair's class from the staging matrix; the second pass reads the rest from there)
        atomicAdd(&cnt[j][(int)(stage[(size_t)tile * M + m].code & 31u)], 1);
      } else {
        // (gain_p2.h, k_piece_build: one header load, the search on the records hinge_load reads next)
        const HingeCached hc = hinge_pair<NW>(ps, m, ps.hdr[m], t0, t1, obj_level, call_level, quiet_binades);
        hg_cache[(size_t)m * TPW + j] = hc;
        atomicAdd(&cnt[j][(int)(hc.code & 31u)], 1);
      }
    }
#ifdef EARHIP_BUILD_PROF
    EARHIP_BUILD_MARK(prof_i);  // pass 1, this batch
    prof_i++;
#endif
  }
  __syncthreads();
#ifdef EARHIP_BUILD_PROF
  EARHIP_BUILD_MARK(prof_i);  // everybody's pass 1
  prof_i++;
#endif
  if (tid < TPW) {
    int at = 0;
    for (int b = 0; b < NC; b++)
      if (b != kHgExact) start[tid][b] = at, at += cnt[tid][b];
    start[tid][kHgExact] = 0;
    start[tid][NC] = at;
    start[tid][NC + 1] = (at + 31) >> 5;
  }
  __syncthreads();
  // flags of the chunk a slot lies in: the waves each kink set reaches, over the classes that share the chunk — made ONCE per
  // (tile, chunk) into LDS (round 6: every thread of the second pass used to run this loop over the classes for its own slot —
  // ~640 instructions of a thread's ~800 per batch, half of the whole builder: tools/build_phases.py)
  __shared__ uint32_t cfl[TPW][kMaxHingeCached / 32];
  for (int i = tid; i < TPW * (hl.cap / 32); i += kHingeBuildThreads) {
    const int jj = i / (hl.cap / 32), c = i % (hl.cap / 32);
    uint32_t f = 0;
    for (int b = 0; b < NC; b++)
      if (b != kHgExact && cnt[jj][b] > 0 && start[jj][b] < 32 * c + 32 && start[jj][b] + cnt[jj][b] > 32 * c) f |= hinge_class_masks<NW>(b);
    cfl[jj][c] = f;
  }
  __syncthreads();
  auto chunk_flags = [&](int jj, int c) -> uint32_t { return cfl[jj][c]; };
  // ---- pass 2
  const unsigned long long samej = [&] {  // the lanes of this wave with the same tile
    unsigned long long v = 0;
    for (int l = j; l < 64; l += TPW) v |= 1ull << l;
    return v;
  }();
  const unsigned long long mytile = [&] {  // lane jj < TPW: the lanes of this wave with tile jj
    unsigned long long v = 0;
    for (int l = lane % TPW; l < 64; l += TPW) v |= 1ull << l;
    return v;
  }();
  for (int mb = 0; mb < M; mb += OB) {
    const int m = mb + oi;
    const bool in = m < M && tile < ntiles;
    HingeCached hc = {0u, 0u, 0.0f, 0.0f};
    if (in) hc = stage ? stage[(size_t)tile * M + m] : hg_cache[(size_t)m * TPW + j];
    const int cls = in ? (int)(hc.code & 31u) : -1, kc = (int)((hc.code >> 5) & 0x3ffffu);
    unsigned long long mine = 0;
    for (int b = 0; b < NC; b++) {
      const unsigned long long bal = __ballot(cls == b);
      mine = cls == b ? bal : mine;
      if (lane < TPW) wcnt[wv][lane][b] = __popcll(bal & mytile);
    }
    const int rank = __popcll(mine & samej & ((1ull << lane) - 1));
    __syncthreads();
    if (in) {
      int at = run[j][cls] + rank;
      for (int w2 = 0; w2 < wv; w2++) at += wcnt[w2][j][cls];
      if (cls == kHgExact) {
        hl.ovf[(size_t)tile * M + at] = m;
      } else {
        const int slot = start[j][cls] + at;
        const int base = ps.off[m], n = ps.cnt[m];
        // (the entries from what the first pass kept: hinge_entries' own expressions)
        const int k1 = min(max(kc - 1, 0), n - 1);
        const uint32_t kinks = hc.kinks & 15u;
        LinEntry e;
        e.m = (uint32_t)m | ((kc < n && kc > k1) ? kLinD2 : 0u) | kinks << kLinKinkShift;
        e.row = base + k1;
        e.p0 = hc.p0, e.scale = hc.scale;
        HingeEntry h;
        const float slope = 1.0f / kHingeFactorScale;
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const uint32_t pos = g < 3 ? (hc.kinks >> (4 + 9 * g)) & 511u : hc.code >> 23;
          h.fac[g] = 0u;
          if (!((kinks >> g) & 1u)) continue;
          if (g & 1) h.fac[g] = f16_bits(-slope) | f16_bits((float)pos * slope) << 16;
          else h.fac[g] = f16_bits(slope) | f16_bits(-(float)pos * slope) << 16;
        }
        hl.lin[(size_t)tile * hl.cap + slot] = e;
        if (chunk_flags(j, slot >> 5)) hl.hinge[(size_t)tile * hl.cap + slot] = h;
      }
    }
    __syncthreads();
    for (int i = tid; i < NC * TPW; i += kHingeBuildThreads) {
      const int b = i / TPW, jj = i % TPW;
      int sm = 0;
      for (int w2 = 0; w2 < NWV; w2++) sm += wcnt[w2][jj][b];
      run[jj][b] += sm;
    }
    __syncthreads();
#ifdef EARHIP_BUILD_PROF
    EARHIP_BUILD_MARK(prof_i);  // pass 2, this batch
    prof_i++;
#endif
  }
  // pad the lists to whole chunks with null entries; publish counts and chunk flags
  for (int i = tid; i < TPW * 32; i += kHingeBuildThreads) {
    const int jj = i >> 5, tl = blockIdx.x * TPW + jj;
    if (tl >= ntiles) continue;
    const int listed = start[jj][NC], nch = start[jj][NC + 1], slot = listed + (i & 31);
    if (slot < 32 * nch) {
      LinEntry e;
      e.m = kLinNull;  // (object 0's inputs against the all-zero rows)
      e.row = ps.zero_row;
      e.p0 = e.scale = 0.0f;
      hl.lin[(size_t)tl * hl.cap + slot] = e;
      HingeEntry h;
      h.fac[0] = h.fac[1] = h.fac[2] = h.fac[3] = 0u;
      if (chunk_flags(jj, slot >> 5)) hl.hinge[(size_t)tl * hl.cap + slot] = h;
    }
    if ((i & 31) == 0) {
      hl.count[tl * 4 + 0] = nch;
      hl.count[tl * 4 + 1] = cnt[jj][kHgExact];
    }
  }
  for (int i = tid; i < TPW * (hl.cap / 32); i += kHingeBuildThreads) {
    const int jj = i / (hl.cap / 32), c = i % (hl.cap / 32), tl = blockIdx.x * TPW + jj;
    if (tl < ntiles && c < start[jj][NC + 1]) hl.cflags[(size_t)tl * (hl.cap / 32) + c] = chunk_flags(jj, c);
  }
  EARHIP_BUILD_MARK(31);
}

#ifndef EARHIP_HG_RINGU
#define EARHIP_HG_RINGU 1
#endif
// -DEARHIP_HG_PROF: every wave of two workgroups of the hinge kernel (the grid's first and a middle one, part 0) sums the s_memtime
// cycles of its chunk loop's phases — 0: at the chunk barrier, 1: row requests + operand split (the wait for the chunk's inputs),
// 2: the line's MFMA blocks with the requests and conversions woven in, 3: staging the rows that arrived + ring, 4: kink sets,
// 5: chunks, 6: kink sets run, 7: the whole kernel — (earhip_debug_hg_prof reads them; tools/hg_phases.py)
#ifdef EARHIP_HG_PROF
static __device__ unsigned long long g_hg_prof[2][8][8];
#define EARHIP_HG_MARK(i)                                            \
  do {                                                               \
    const unsigned long long t_now_ = __builtin_readcyclecounter();  \
    hgp_acc[(i)] += t_now_ - hgp_last;                               \
    hgp_last = t_now_;                                               \
  } while (0)
#else
#define EARHIP_HG_MARK(i) do {} while (0)
#endif
typedef _Float16 hg_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hg_h2 as_h2(uint32_t u) { return __builtin_bit_cast(hg_h2, u); }
__device__ __forceinline__ uint32_t h2_bits(hg_h2 h) { return __builtin_bit_cast(uint32_t, h); }

// K1.  grid = (tiles, grid-level splits of the chunk schedule, column super-groups), block = 64 NW threads (NW = 8 or 4):
// wave w = samples [64 w, 64 w + 64) of the workgroup's tile of 64 NW samples.
//
// A chunk of 32 list slots is ONE step between two workgroup barriers: the line (inputs split, 72 MFMAs per 48 columns),
// then — where the chunk has them — its kink sets (first forward, first backward, second forward, second backward: 36
// MFMAs each; a wave skips those that cannot reach its samples).  The vector-memory counter is in order, so everything
// is requested in one place, at the start of a step: first the rows of the chunk after next (per slot the two gain rows
// of its line and the kink rows of its kinks; a kink it does not have: the all-zero row, which stays in L1), then the
// inputs of the next chunk into the registers the split has just freed (one chunk ahead:
// two sets of inputs do not fit the register file beside the 96 accumulators, the operand pieces the kinks need, and
// the rows on their way).  The rows go through a wave-private piece of LDS, a chunk's worth at a time: a step first
// turns the rows staged a step ago into the operand fragments of the NEXT chunk (all three sets; fragments alternate
// between two buffers), then stages the rows that have arrived meanwhile.  The scaled high piece (h 2^-11, gain_h2.h)
// is made from h where it is used.
// WIDE: the low pieces of the inputs scaled by 2^11 (gain_h2.h), their partner made from the gains' high pieces as they are
// read; the plain form saves those multiplies (16 + 4 per block of 12 MFMAs).  Both forms are in the kernel; the probe's word
// (wide_cur: bit 0 of the gate word, k_hinge_gate) picks at run time.
template <int NCT, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 1)
k_gain_mix_hg(GainMixParams P, HingeLists hl, float x_scale, const float *__restrict__ gcol, const unsigned *level_cur,
              unsigned *level_next, const unsigned *wide_cur, unsigned *wide_next, const unsigned *gate, const unsigned *span) {
  if (gate && (*gate & kGateHingeUnsafe)) return;  // the piece lists do this call (k_hinge_gate); they clear the words of the next
  // `span` (round 6; the same word as `gate`, given INSTEAD of it): bit 1 = some object of this call falls further below the call's
  // level than the packed-f16 kink products hold (kHingeSpreadBinades) — the call then runs the ROBUST form of the kinks (their
  // products made in f32, below) instead of being handed to the piece lists

  constexpr int NRT = 4, TS = 16 * NRT, CH = kSplitChunk, T = 64 * NW;
  constexpr int NQ = CH / NW;        // list slots whose gains one wave converts per chunk
  constexpr int KS = NW / 2;         // kink sets: one (4 waves: 256-sample tiles) or two (8 waves: 512) on either side of the centre
  constexpr int NR = 2 + KS;         // rows a slot asks for: its kink rows and the two gain rows of its line
  constexpr int RING = 4;
  constexpr int OP = TS + 4;
  // fragments (1 KB each: 64 lanes x 8 f16), two sets (even / odd chunks) of: the line {B0, B1} x column tiles x {h, l}, the
  // kink sets (column tiles x {h, l} each)
  constexpr int FL = 4 * NCT, FH = 2 * NCT, FSET = FL + KS * FH, NFRAGS = 2 * FSET;
  constexpr size_t kFragBytes = sizeof(u32x4) * NFRAGS * 64, kTileBytes = sizeof(float) * NW * 16 * OP;
  __shared__ __attribute__((aligned(16))) unsigned char fmem[kFragBytes > kTileBytes ? kFragBytes : kTileBytes];  // (the waves' output
                                                                                            // tiles, once the last step is through)
  auto frag = reinterpret_cast<u32x4(*)[64]>(fmem);
  __shared__ float inv_gcol[16 * NCT];
  __shared__ __attribute__((aligned(16))) u32x4 fdummy[1];
  __shared__ __attribute__((aligned(16))) uint32_t ring[RING][CH];   // byte offsets of the slots' input rows (below 4 GB: api_core.hip checks)
  __shared__ __attribute__((aligned(16))) u32x4 ringe[RING][2 * CH];  // the LinEntries [0, 32) and the HingeEntries [32, 64)
  __shared__ __attribute__((aligned(16))) uint32_t ringf[RING][KS][CH];  // ... their factor words, packed for the lanes
  __shared__ uint32_t ringc[RING];                                    // chunk flags
  __shared__ __attribute__((aligned(16))) float stage[NW][NQ * NR * 16 * NCT];  // a wave's rows: [slot][NR rows][16 NCT columns]
  // Both forms of the body (WIDE or not) in ONE kernel, the probe's word picking at run time: two kernels launched back to
  // back, one of which looks at the word and returns, cost 5 us per call for the one that returns
  auto body = [&](auto wide_tag, auto robust_tag) __attribute__((always_inline)) {
  constexpr bool WIDE = decltype(wide_tag)::value;
  constexpr bool ROBUST = decltype(robust_tag)::value;  // (with WIDE only)
  constexpr float LOW = WIDE ? kLowPieceScale : 1.0f;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kg = lane >> 4;
  // workgroups k and k + 32 of an XCD — the two a CU holds when the dispatcher goes round the XCD's 32 CUs — take adjacent
  // tiles: most of the gain rows one of them asks for the other has just fetched or is fetching (L2 -> L1 requests are
  // what this kernel is short of: 2 % of its time, measured; any mapping is correct)
  const int wgtile = [&] {
    const int b = blockIdx.x, n = gridDim.x, per = n >> 3;
    if (NW != 4 || b >= per * 8 || (per & 63)) return xcd_tile(b, n);
    const int k = b >> 3;
    return (b & 7) * per + (((k >> 6) << 6) | ((k & 31) << 1) | ((k >> 5) & 1));
  }();
  // gains are scaled to HALF the range the other split kernels use (T x slope of a ramp of T / 2 samples or more is at
  // most twice the gains' range; differences of two such slopes stay inside the f16 range for gains of one sign, and
  // what does not shows as non-finite totals: exact redo below)
  if (threadIdx.x < 16 * NCT) inv_gcol[threadIdx.x] = 2.0f / gcol[blockIdx.z * 16 * NCT + threadIdx.x];
  __syncthreads();
  if (level_cur) {  // input scale of THIS call from the level K0 probed (gain_h2.h)
    const unsigned lv = *level_cur;
    if (lv) {
      const int E = max(-60, min(20, (int)(lv >> 23) - 127));
      x_scale = __uint_as_float((unsigned)(127 + 7 - E) << 23);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
      *level_next = 0;
      if (wide_next) *wide_next = 0u;
      record_mode(P, wide_cur != nullptr);
    }
  }
  const int nparts = gridDim.y, part = blockIdx.y;
  const int col0 = blockIdx.z * 16 * NCT;
  const int wave_s0 = w * TS;
  const int tile_s0 = wgtile * T + wave_s0;
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
#ifdef EARHIP_HG_PROF
  unsigned long long hgp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long hgp_start = __builtin_readcyclecounter();
  unsigned long long hgp_last = hgp_start;
#endif

  // ---- exact path: one object, all its pieces inside this wave's 64 samples, f32 MFMA with k = {a, b} of ONE
  // object, accumulated into tot0 in units of 1 / (sx sg) (gain_p2.h)
  auto single_object = [&](int m, float sx, bool sg) {
    if (tile_len <= 0) return;
    float gsc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; c++) gsc[c] = sg ? 0.5f * gcol[col0 + c * 16 + li] : 1.0f;
    // (header + a window of records around the guess: two dependent rounds of loads instead of a bisection's ~12 — gain_p2.h)
    const ObjHdr hd = P.ps.hdr[m];
    const int base = hd.off, n = hd.cnt;
    const float *row = P.in + (size_t)m * P.in_stride + tile_s0;
    const bool is_b = kg & 1;
    const bool slot0 = kg < 2;
    int k = upper_bound_rec_window(P.ps.rec + base, n, hd.first, hd.last, tile_t0);
    int cur = 0;
    while (cur < tile_len) {
      const SegDesc dk = describe_segment(P.ps, base, n, k, tile_t0, tile_t1);
      const int r1 = min(seg_r1(dk.info), tile_len);
      if (r1 > cur) {
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

  float inv_x = 1.0f / x_scale;  // exact: a power of two
  bool col_scaled = true;
  const int *cnt = hl.count + wgtile * 4;
  const LinEntry *lbase = hl.lin + (size_t)wgtile * hl.cap;
  const HingeEntry *hbase = hl.hinge + (size_t)wgtile * hl.cap;
  const uint32_t *cfbase = hl.cflags + (size_t)wgtile * (hl.cap / 32);

  if (P.vec_ok) {
    const int total = cnt[0];
    const int c_lo = (int)(((int64_t)total * part) / nparts), c_hi = (int)(((int64_t)total * (part + 1)) / nparts);
    if (c_hi > c_lo) {
      const int nvec = (P.nsamples + 3) & ~3;
      const uint32_t rstride = (uint32_t)(P.in_stride * sizeof(float));
      const float g_scale = 0.5f * gcol[col0 + min(lane, 16 * NCT - 1)];  // the scale of the lane's gain column
      const uint32_t kink_off = (uint32_t)P.ps.kink_row0 * (rowlen * 4u);     // kink rows: bytes behind the gain rows' start
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      typedef float f32x3 __attribute__((ext_vector_type(3)));
      typedef const uint32_t __attribute__((address_space(4))) *ConstWords;  // (K0 wrote them, this kernel only reads: scalar loads)
      // What a step needs to know about its lane.  (An earlier form of this kernel made these anew in every chunk from
      // an opaque copy of the lane number: carried through the chunk loop they were what the register allocator spilled,
      // and a reload behind the input requests waits for the inputs.  With the kink rows precomputed the registers are
      // there: 4 % faster on 256-sample tiles, the same on 512.)
      struct LaneCtx {
        int lane, li, kg;
        unsigned xlane;      // byte offset of the lane's float4 inside an input row
        unsigned glane;      // byte offset of the lane's 12 bytes inside a gain row
        float *sw;           // where the lane's 12 bytes of request 0 go in the wave's staged rows
        const float *sf;     // the lane's column in them
        int fpair;           // the lane's fragment pair (h, l) inside a set of column tiles, or -1: no column
      };
      auto lane_ctx = [&](bool opaque) {
        unsigned lv = threadIdx.x;
        if (opaque) asm volatile("" : "+v"(lv));
        LaneCtx L;
        L.lane = (int)(lv & 63u);
        L.li = L.lane & 15;
        L.kg = L.lane >> 4;
        L.xlane = (unsigned)min(tile_s0 + L.li * NRT, nvec - 4) * 4u;
        L.glane = (unsigned)(col0 + NCT * L.li) * 4u;
        L.sw = reinterpret_cast<float *>(&stage[w][0]) + L.kg * (NR * 16 * NCT) + NCT * L.li;
        L.sf = reinterpret_cast<const float *>(&stage[w][0]) + min(L.lane, 16 * NCT - 1);
        L.fpair = L.lane < 16 * NCT ? 2 * L.kg : -1;
        return L;
      };

      // ---- the ring: wave 0 brings the entries of chunk c + RD into LDS during the line step of chunk c (requested ahead of
      // its MFMAs, stored behind them): lanes 0-31 the LinEntries, 32-63 the HingeEntries
      constexpr int RD = 3;
      u32x4 ring_next = {0u, 0u, 0u, 0u};
      uint32_t ring_next_cf = 0u;
      auto ring_load = [&](const LaneCtx &L, int c) {
        const int cc = min(c, total - 1);
        const int s = 32 * cc + (L.lane & 31);
        ring_next_cf = c < c_hi ? ((ConstWords)cfbase)[cc] : 0u;
        const u32x4 *pa = L.lane < 32 ? reinterpret_cast<const u32x4 *>(lbase + s) : reinterpret_cast<const u32x4 *>(hbase + s);
#if EARHIP_HG_RINGU
        // The HingeEntries are requested whether the chunk has kinks or not (round 5).  Asking only when the chunk's flag word
        // said so made the request depend on a load: wave 0 sat in `s_waitcnt vmcnt(0)` — everything in flight, inputs and
        // gain rows included — behind every chunk's barrier, and seven waves waited for it at the next one.  A chunk without
        // kinks reads 512 bytes nobody looks at (its factor words are only used behind the flag word's test).
        ring_next = *pa;
#else
        const bool want = L.lane < 32 || ring_next_cf != 0u;
        ring_next = want ? *pa : u32x4{0u, 0u, 0u, 0u};
#endif
      };
      auto ring_store = [&](const LaneCtx &L, int c) {
        const int sl = c & (RING - 1);
        ringe[sl][L.lane] = ring_next;
        if (L.lane < 32) {
          ring[sl][L.lane] = (ring_next[0] & kLinObjMask) * rstride;
          if (L.lane == 0) ringc[sl] = ring_next_cf;
        } else {
#pragma unroll
          for (int g = 0; g < KS; g++) ringf[sl][g][L.lane - 32] = ring_next[g];
        }
      };
      // ---- inputs of chunk c: 8 requests of 16 bytes per lane (slots 8 kg .. 8 kg + 7, the lane's 4 samples)
      auto load_x = [&](const LaneCtx &L, int c, f32x4 (&x)[8]) {
        const char *bp = reinterpret_cast<const char *>(P.in);
#pragma unroll
        for (int q = 0; q < 8; q += 4) {
          const u32x4 mw = *reinterpret_cast<const u32x4 *>(&ring[c & (RING - 1)][L.kg * 8 + q]);
#pragma unroll
          for (int i = 0; i < 4; i++) x[q + i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(bp + (mw[i] + L.xlane)));
        }
      };
      // ---- the rows of the slots NQ w .. NQ w + NQ - 1 of chunk c, NR each (j = 0 ..: the kink rows of the backward kinks,
      // second one first; gain rows R1, R2 of the line; the kink rows of the forward kinks).  Request i is row j = i % NR of
      // four slots — lane (kg, li) takes 12 bytes (columns 3 li .. 3 li + 2 of the workgroup's 48) of slot 4 (i / NR) + kg's —
      // so that which row a request is for is known at compile time and a lane reads its slot's entry once per group
      constexpr int NREQ = NQ * NR / 4;
      auto load_gains = [&](const LaneCtx &L, int c, f32x3 (&G)[NREQ]) {
#pragma unroll
        for (int grp = 0; grp < NQ / 4; grp++) {
          const u32x2 mr = *reinterpret_cast<const u32x2 *>(&ringe[c & (RING - 1)][w * NQ + 4 * grp + L.kg]);  // (object | kinks, row R1)
          const unsigned r1 = mr[1] * (rowlen * 4u) + L.glane, r2 = r1 + ((mr[0] >> 16) & 1u) * (rowlen * 4u);
#pragma unroll
          for (int j = 0; j < NR; j++) {
            const int jj = j - KS / 2;  // jj = 0, 1: the gain rows
            unsigned off;
            if (jj == 0) {
              off = r1;
            } else if (jj == 1) {
              off = r2;
            } else {
              // backward kink nb = -jj (1, 2): kink 2 nb - 1, the point nb - 1 in front of R1; forward nf = jj - 1: kink
              // 2 nf - 2, the point nf - 1 behind R2; a kink the pair does not have: the all-zero row of the kink image
              const int kbit = jj < 0 ? kLinKinkShift - 2 * jj - 1 : kLinKinkShift + 2 * jj - 4;
              const unsigned at = jj < 0 ? r1 - (unsigned)(-jj - 1) * (rowlen * 4u) : r2 + (unsigned)(jj - 2) * (rowlen * 4u);
              off = kink_off + (((mr[0] >> kbit) & 1u) ? at : L.glane);
            }
            // (uniform base + 32-bit lane offset: the image of the curves is below 4 GB, api_core.hip checks)
            const float *gp = reinterpret_cast<const float *>(reinterpret_cast<const char *>(gain) + off);
            f32x3 &g = G[grp * NR + j];
            if constexpr (NCT == 3) g = *reinterpret_cast<const f32x3 *>(gp);
            else if constexpr (NCT == 2) g = f32x3{gp[0], gp[1], 0.0f};
            else g = f32x3{gp[0], 0.0f, 0.0f};
          }
        }
      };
      auto stage_gains = [&](const LaneCtx &L, const f32x3 (&G)[NREQ]) {
#pragma unroll
        for (int i = 0; i < NREQ; i++) {
          float *d = L.sw + ((i / NR) * 4 * NR + i % NR) * (16 * NCT);  // staged row NR slot + j
          if constexpr (NCT == 3) *reinterpret_cast<f32x3 *>(d) = G[i];
          else if constexpr (NCT == 2) d[0] = G[i][0], d[1] = G[i][1];
          else d[0] = G[i][0];
        }
      };
      // The conversions in slices of four of the wave's slots (4 sl .. 4 sl + 3: half a fragment entry, k = 8 of its 32), so
      // that they can be woven between the MFMAs.  (Lanes 48 .. 63 have no column: they all write the same 8 bytes nobody
      // reads — no branch, the slices stay inside the MFMA blocks' basic block.)
      constexpr int NSL = NQ / 4;
      auto store_slice = [&](const LaneCtx &L, uint32_t H0, uint32_t H1, uint32_t L0, uint32_t L1, int f, int sl) {
        const int gs = NQ * w + 4 * sl, fe = (gs >> 3) * 16 + L.li, hf = (gs >> 2) & 1;
        u32x2 *ph = L.fpair >= 0 ? reinterpret_cast<u32x2 *>(&frag[f + L.fpair][fe]) + hf : reinterpret_cast<u32x2 *>(&fdummy[0]);
        u32x2 *pl = L.fpair >= 0 ? reinterpret_cast<u32x2 *>(&frag[f + L.fpair + 1][fe]) + hf : reinterpret_cast<u32x2 *>(&fdummy[0]) + 1;
        *ph = u32x2{H0, H1};
        *pl = u32x2{L0, L1};
      };
      auto store_split = [&](const LaneCtx &L, const float (&v)[4], int f, int sl) {  // scaled values: split here
        const uint32_t H0 = pack_f16(v[0], v[1]), H1 = pack_f16(v[2], v[3]);
        const uint32_t L0 = pack_f16(sub_f16_lo(v[0], H0), sub_f16_hi(v[1], H0)), L1 = pack_f16(sub_f16_lo(v[2], H1), sub_f16_hi(v[3], H1));
        store_slice(L, H0, H1, L0, L1, f, sl);
      };
      // the line of chunk c from the staged rows -> its fragment set
      auto convert_lin_slice = [&](const LaneCtx &L, int c, int sl) {
        float b0[4], b1[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int q = 4 * sl + i;
          const float R1 = L.sf[(NR * q + KS / 2) * (16 * NCT)], dC = L.sf[(NR * q + KS / 2 + 1) * (16 * NCT)] - R1;
          const u32x2 ps2 = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const char *>(&ringe[c & (RING - 1)][w * NQ + q]) + 8);
          const float p0 = __uint_as_float(ps2[0]), sc = __uint_as_float(ps2[1]);
          b0[i] = __builtin_fmaf(p0, dC, R1) * g_scale;  // the line at the tile start: S + p (E - S) (gain_interpolator.hpp:272-274)
          b1[i] = (sc * dC) * g_scale;                   // its slope per sample
        }
        const int f0 = (c & 1) * FSET;
        store_split(L, b0, f0, sl);
        store_split(L, b1, f0 + 2 * NCT, sl);
      };
      // the kink sets g0 .. g1 - 1 of chunk c: their rows hold the operands' pieces already (k_kink_rows); regrouped into
      // fragments (an f16 pair = two SLOTS of one column)
      auto convert_kinks_slice = [&](const LaneCtx &L, int c, int sl, int g0, int g1) {
        const uint32_t *su = reinterpret_cast<const uint32_t *>(L.sf);
        const int f0 = (c & 1) * FSET + FL;
#pragma unroll
        for (int g = 0; g < KS; g++) {
          if (g < g0 || g >= g1) continue;
          const int j = (g & 1) ? KS / 2 - 1 - (g >> 1) : KS / 2 + 2 + (g >> 1);  // the staged row of kink g
          uint32_t hw[2], lw[2];
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const int q = 4 * sl + 2 * i;
            const uint32_t w0 = su[(NR * q + j) * (16 * NCT)], w1 = su[(NR * (q + 1) + j) * (16 * NCT)];
            hw[i] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);
            lw[i] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
          }
          store_slice(L, hw[0], hw[1], lw[0], lw[1], f0 + g * FH, sl);
        }
      };

      // ---- prologue: ring slots of the first chunks; fragments of the first chunk; rows of the second one staged
      f32x4 X0[8];
      u32x4 ah[NRT], al[NRT];
      {
        const LaneCtx L = lane_ctx(false);
        if (w == 0) {
#pragma unroll
          for (int jn = 0; jn < RD; jn++) {
            ring_load(L, c_lo + jn);
            ring_store(L, c_lo + jn);
          }
#if EARHIP_HG_RINGU
          ring_load(L, c_lo + RD);  // (stored by the first chunk: every chunk requests the slot the NEXT one stores)
#endif
        }
        __syncthreads();
        f32x3 G[NREQ];
        load_gains(L, c_lo, G);
        stage_gains(L, G);
        load_gains(L, c_lo + 1, G);
        load_x(L, c_lo, X0);
#pragma unroll
        for (int sl = 0; sl < NSL; sl++) {
          convert_lin_slice(L, c_lo, sl);
          convert_kinks_slice(L, c_lo, sl, 0, KS);
        }
        stage_gains(L, G);
      }

      auto load_x_part = [&](const LaneCtx &L, int c, f32x4 (&x)[8], int q0) {  // requests q0, q0 + 1
        const char *bp = reinterpret_cast<const char *>(P.in);
        const u32x2 mw = *reinterpret_cast<const u32x2 *>(&ring[c & (RING - 1)][L.kg * 8 + q0]);
        x[q0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(bp + (mw[0] + L.xlane)));
        x[q0 + 1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(bp + (mw[1] + L.xlane)));
      };

      auto chunk_body = [&](int c, f32x4 (&X)[8]) __attribute__((always_inline)) {
        __syncthreads();  // the fragments of chunk c are in place; ring slots up to c + RD - 1 are visible
        EARHIP_HG_MARK(0);
        const LaneCtx L = lane_ctx(false);
        const uint32_t cf = ringc[c & (RING - 1)];
        f32x3 G[NREQ];
        load_gains(L, c + 2, G);  // (past the schedule: the clamped last chunk's, never used)
#if !EARHIP_HG_RINGU
        if (w == 0) ring_load(L, c + RD);  // (stored behind the line's MFMAs)
#endif
        __builtin_amdgcn_sched_barrier(0);  // every gain row is requested before any input
        // operand split of the inputs (gain_h2.h, wide form): 2 x 2 blocks, an f16 pair packs two SLOTS of one row tile
#pragma unroll
        for (int qp = 0; qp < 4; qp++)
#pragma unroll
          for (int rp = 0; rp < NRT; rp += 2) {
            const f32x2 s0 = f32x2{X[2 * qp][rp], X[2 * qp][rp + 1]} * x_scale;
            const f32x2 s1 = f32x2{X[2 * qp + 1][rp], X[2 * qp + 1][rp + 1]} * x_scale;
            const uint32_t H0 = pack_f16(s0[0], s1[0]), H1 = pack_f16(s0[1], s1[1]);
            const f32x2 r0 = f32x2{sub_f16_lo(s0[0], H0), sub_f16_lo(s0[1], H1)} * LOW;
            const f32x2 r1 = f32x2{sub_f16_hi(s1[0], H0), sub_f16_hi(s1[1], H1)} * LOW;
            ah[rp][qp] = H0;
            ah[rp + 1][qp] = H1;
            al[rp][qp] = pack_f16(r0[0], r1[0]);
            al[rp + 1][qp] = pack_f16(r0[1], r1[1]);
          }
        __builtin_amdgcn_sched_barrier(0);  // the splitting above stays above
        EARHIP_HG_MARK(1);
        // ======== the line: 2 NCT blocks of 12 MFMAs (operand B0 / B1 of a column tile: the three partial products, small
        // ones first).  Between them, in this order: the requests of the next chunk's inputs (into the registers the
        // split has just freed), then the next chunk's operands from the rows staged a step ago — MFMA and VALU
        // instructions of one wave overlap only when they alternate (gain_h2.h).  (h, l) of a block are read a block ahead.
        const int f0 = (c & 1) * FSET;
        constexpr int NBLK = 2 * NCT;
        const int cx = min(c + 1, c_hi - 1);
        u32x4 bh[2];
        auto frag_of = [&](int blk) { return f0 + (blk & 1) * 2 * NCT + 2 * (blk >> 1); };
        bh[0] = frag[frag_of(0)][L.lane];
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        auto block = [&](auto blk_tag) __attribute__((always_inline)) {
          constexpr int blk = decltype(blk_tag)::value;
          constexpr int ct = blk >> 1;
          // the items of this block: it0 .. it1 - 1 of: 4 pairs of input requests, 2 half-slices of the line's conversion,
          // 2 of the kinks'
          constexpr int it0 = NBLK == 6 ? (blk < 2 ? 2 * blk : blk + 2) : NBLK == 4 ? (blk == 0 ? 0 : blk == 1 ? 4 : blk + 4) : 6 * blk;
          constexpr int it1 = NBLK == 6 ? (blk < 2 ? 2 * blk + 2 : blk + 3) : NBLK == 4 ? (blk == 0 ? 4 : blk == 1 ? 6 : blk + 5) : 6 + 2 * blk;
          constexpr int nx = (it1 < 4 ? it1 : 4) - (it0 < 4 ? it0 : 4), nconv = (it1 - it0) - nx;
          f32x4(&tt)[NRT][NCT] = (blk & 1) ? tot1 : tot0;
          // (h of a block is read a block ahead, l as the block starts, behind the four MFMAs that do not need it: registers)
          const u32x4 ch = bh[blk & 1];
          const u32x4 cl = frag[frag_of(blk) + 1][L.lane];
          if constexpr (blk + 1 < NBLK) bh[(blk + 1) & 1] = frag[frag_of(blk + 1)][L.lane];
          u32x4 bs;
#pragma unroll
          for (int i = 0; i < 4; i++) bs[i] = WIDE ? scale_f16x2_down(ch[i]) : ch[i];  // h 2^-11: partner of the inputs' scaled low piece
#pragma unroll
          for (int r = 0; r < NRT; r++) tt[r][ct] = mfma_f16(al[r], bs, tt[r][ct]);
#pragma unroll
          for (int r = 0; r < NRT; r++) tt[r][ct] = mfma_f16(ah[r], cl, tt[r][ct]);
#pragma unroll
          for (int r = 0; r < NRT; r++) tt[r][ct] = mfma_f16(ah[r], ch, tt[r][ct]);
#pragma unroll
          for (int it = it0; it < it1; it++) {
            if (it < 4) {
              load_x_part(L, cx, X, 2 * it);
            } else if (it < 6) {
              if (it - 4 < NSL) convert_lin_slice(L, c + 1, it - 4);
            } else if (NSL == 2) {  // (whether the next chunk has kinks or not: no branch inside the blocks)
              convert_kinks_slice(L, c + 1, it - 6, 0, KS);
            } else {
              convert_kinks_slice(L, c + 1, 0, (KS / 2) * (it - 6), (KS / 2) * (it - 5));
            }
          }
          // issue order: the LDS reads of the next block first, then the MFMAs with the other work between them
          if constexpr (blk + 1 < NBLK) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if constexpr (nconv > 0) {
#pragma unroll
            for (int k = 0; k < 12; k++) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            }
          } else if constexpr (nx > 0) {
#pragma unroll
            for (int k = 0; k < nx; k++) {  // MFMAs, then a pair of requests (address arithmetic + loads)
              __builtin_amdgcn_sched_group_barrier(0x008, 12 / nx, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
              __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            }
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
          }
        };
        block(std::integral_constant<int, 0>{});
        block(std::integral_constant<int, 1>{});
        if constexpr (NBLK > 2) {
          block(std::integral_constant<int, 2>{});
          block(std::integral_constant<int, 3>{});
        }
        if constexpr (NBLK > 4) {
          block(std::integral_constant<int, 4>{});
          block(std::integral_constant<int, 5>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        EARHIP_HG_MARK(2);
        stage_gains(L, G);  // the rows of the chunk after next replace the next chunk's (same wave: in order)
        if (w == 0) ring_store(L, c + RD);
#if EARHIP_HG_RINGU
        // wave 0's list requests for the slot the NEXT chunk stores, here, behind its own store and a whole chunk ahead of their
        // use: requests inside a branch are invisible to the compiler's wait counts — in front of the split they made wave 0
        // wait for gain rows it had only just asked for before it could touch its inputs
        if (w == 0) ring_load(L, c + 1 + RD);
#endif
        // ======== its kinks: forward, then backward
        EARHIP_HG_MARK(3);
#pragma unroll 1
        for (int g = 0; g < KS; g++) {
          if (!((cf >> (8 * g + w)) & 1u)) continue;
#ifdef EARHIP_HG_PROF
          hgp_acc[6]++;
#endif  // (wave-uniform) no such kinks, or none that reach this wave's samples
          // the lane's 8 factor words -> (slope, slope) and (offset, offset) pairs of its slot pairs
          const u32x4 fa = *reinterpret_cast<const u32x4 *>(&ringf[c & (RING - 1)][g][L.kg * 8]);
          const u32x4 fb = *reinterpret_cast<const u32x4 *>(&ringf[c & (RING - 1)][g][L.kg * 8 + 4]);
          uint32_t SC[4], P0[4];
#pragma unroll
          for (int qp = 0; qp < 4; qp++) {
            const uint32_t w0 = qp < 2 ? fa[2 * qp] : fb[2 * qp - 4], w1 = qp < 2 ? fa[2 * qp + 1] : fb[2 * qp - 3];
            SC[qp] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);  // (lo16 of w0, lo16 of w1)
            P0[qp] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);  // (hi16 of w0, hi16 of w1)
          }
          const int fh = f0 + FL + g * FH;
          // the lane's samples as f16 pairs (s, s), s counted from the start of the workgroup tile: the factor of a kink
          // is F = max(s * slope + offset, 0) in packed f16 arithmetic (exact: 8 bits; at most 16: kHingeFactorScale)
          uint32_t SS[NRT];
#pragma unroll
          for (int r = 0; r < NRT; r++) SS[r] = f16_bits((float)(wave_s0 + L.li * NRT + r)) * 0x10001u;
#pragma unroll
          for (int rh = 0; rh < NRT; rh += 2) {  // two row tiles at a time (registers)
            u32x4 Fh[2], Fl[2];
#pragma unroll
            for (int r2 = 0; r2 < 2; r2++)
#pragma unroll
              for (int qp = 0; qp < 4; qp++) {
                const int r = rh + r2;
                const hg_h2 zero = {(_Float16)0.0f, (_Float16)0.0f};
                const hg_h2 k2048 = {(_Float16)LOW, (_Float16)LOW};
                const hg_h2 F = __builtin_elementwise_max(__builtin_elementwise_fma(as_h2(SS[r]), as_h2(SC[qp]), as_h2(P0[qp])), zero);
                const hg_h2 xh = as_h2(ah[r][qp]), xl = as_h2(al[r][qp]);
                if constexpr (ROBUST) {
                  // The robust form: F x in f32 — Q = (F 2^11) xh + F xl = 2^11 F x exactly (F 2^11 <= 65408 is an f16, the products
                  // of f16 pairs are exact in f32), split like an input: Ah = rn16(Q 2^-11), Al = rn16(Q - 2^11 Ah).  An input
                  // 21 binades below the call's level still gets ~17 bits of its product (the packed form above: 2^-25 ABSOLUTE, which
                  // is what keeps it to calls whose objects stay within 16 binades); 7 instructions per value instead of 3.
                  const uint32_t fb = h2_bits(F), flb = h2_bits(F * k2048), xhb = ah[r][qp], xlb = al[r][qp];
                  float q0, q1, t0, t1;
                  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(t0) : "v"(fb), "v"(xlb));
                  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(t1) : "v"(fb), "v"(xlb));
                  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(q0) : "v"(flb), "v"(xhb), "v"(t0));
                  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(q1) : "v"(flb), "v"(xhb), "v"(t1));
                  const uint32_t A = pack_f16(q0 * (1.0f / kLowPieceScale), q1 * (1.0f / kLowPieceScale));
                  float r0, r1;
                  const float mlow = -kLowPieceScale;
                  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(A), "s"(mlow), "v"(q0));
                  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(A), "s"(mlow), "v"(q1));
                  Fh[r2][qp] = A;
                  Fl[r2][qp] = pack_f16(r0, r1);
                } else {
                  const hg_h2 Ah = F * xh;                                         // rn(F xh)
                  const hg_h2 res = __builtin_elementwise_fma(F, xh, -Ah);         // ... its residual: exact
                  const hg_h2 Al = __builtin_elementwise_fma(res, k2048, F * xl);  // low piece, in the scaled units of xl
                  Fh[r2][qp] = h2_bits(Ah);
                  Fl[r2][qp] = h2_bits(Al);
                }
              }
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) {
              const u32x4 bh = frag[fh + 2 * ct][L.lane];
              u32x4 bs;
#pragma unroll
              for (int i = 0; i < 4; i++) bs[i] = WIDE ? scale_f16x2_down(bh[i]) : bh[i];
#pragma unroll
              for (int r2 = 0; r2 < 2; r2++) tot0[rh + r2][ct] = mfma_f16(Fl[r2], bs, tot0[rh + r2][ct]);
              const u32x4 bl = frag[fh + 2 * ct + 1][L.lane];
#pragma unroll
              for (int r2 = 0; r2 < 2; r2++) tot0[rh + r2][ct] = mfma_f16(Fh[r2], bl, tot0[rh + r2][ct]);
#pragma unroll
              for (int r2 = 0; r2 < 2; r2++) tot0[rh + r2][ct] = mfma_f16(Fh[r2], bh, tot0[rh + r2][ct]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
#ifdef EARHIP_HG_PROF
        __builtin_amdgcn_sched_barrier(0);
        EARHIP_HG_MARK(4);
        hgp_acc[5]++;
#endif
      };
#ifdef EARHIP_HG_PROF
      hgp_last = __builtin_readcyclecounter();  // (the prologue is not a phase)
#endif
#pragma unroll 1
      for (int c = c_lo; c < c_hi; c++) chunk_body(c, X0);
    }
    __syncthreads();  // (the fragments' memory becomes the output tiles below)

    // objects that take the exact path: part 0 only
    if (part == 0) {
      const int *ovf = hl.ovf + (size_t)wgtile * P.M;
      for (int i = 0; i < cnt[1]; i++) single_object(ovf[i], x_scale, true);
    }
    // an operand beyond the f16 range (or not finite) shows as non-finite totals: redo the wave's tile exactly, unscaled
    bool bad = false;
#pragma unroll
    for (int r = 0; r < NRT; r++)
#pragma unroll
      for (int c = 0; c < NCT; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) bad |= !(__builtin_fabsf(tot0[r][c][e]) < INFINITY) || !(__builtin_fabsf(tot1[r][c][e]) < INFINITY);
    if (__ballot(bad)) {
      clear_totals();
      inv_x = 1.0f;
      col_scaled = false;
      for (int s = 32 * c_lo; s < 32 * c_hi; s++) {
        const uint32_t mw = lbase[s].m;
        if (!(mw & kLinNull)) single_object((int)(mw & kLinObjMask), 1.0f, false);
      }
      if (part == 0) {
        const int *ovf = hl.ovf + (size_t)wgtile * P.M;
        for (int i = 0; i < cnt[1]; i++) single_object(ovf[i], 1.0f, false);
      }
    }
  } else {
    inv_x = 1.0f;
    col_scaled = false;
    const int m_lo = (int)(((int64_t)P.M * part) / nparts), m_hi = (int)(((int64_t)P.M * (part + 1)) / nparts);
    for (int m = m_lo; m < m_hi; m++) single_object(m, 1.0f, false);  // unaligned rows
  }

#ifdef EARHIP_HG_PROF
  {
    const int slot = blockIdx.x == 0 ? 0 : blockIdx.x == gridDim.x / 2 ? 1 : -1;
    hgp_acc[7] = __builtin_readcyclecounter() - hgp_start;
    if (slot >= 0 && part == 0 && blockIdx.z == 0 && lane == 0 && w < 8)
      for (int i = 0; i < 8; i++) g_hg_prof[slot][w][i] = hgp_acc[i];
  }
#endif
  if (tile_len <= 0) return;
  // D fragment of row tile r: rows 4 kg + e = samples 16 kg + 4 e + r; (s - c) counts from the WORKGROUP tile's centre
  // (where the line is anchored: LinEntry::p0)
  const float wf0 = (float)(wave_s0 + kg * 16 - T / 2);
  float inv_gc[NCT];
#pragma unroll
  for (int c = 0; c < NCT; c++) inv_gc[c] = col_scaled ? inv_gcol[c * 16 + li] : 1.0f;
  float *op = P.out + (size_t)blockIdx.y * P.part_stride + tile_s0;
  const bool whole = P.vec_ok && tile_len == TS;  // (wave-uniform)
  float *ot = reinterpret_cast<float *>(fmem) + w * 16 * OP;
#pragma unroll
  for (int c = 0; c < NCT; c++) {
    if (whole) {  // transposed through wave-private LDS: whole 256-byte rows per store instruction (gain_h2.h)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < NRT; r++)
          v[r] = (__builtin_fmaf(wf0 + (float)(4 * e + r), tot1[r][c][e], tot0[r][c][e]) * inv_x) * inv_gc[c];
        *reinterpret_cast<f32x4 *>(ot + li * OP + kg * 16 + e * 4) = v;
      }
#pragma unroll
      for (int jn = 0; jn < 4; jn++) {
        const int cl = 4 * jn + kg, col = col0 + c * 16 + cl;
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
  };  // body
  // (three forms of the body in one kernel; the words are wave-uniform: scalar branches)
  if (span && hinge_span_exceeded(*span, P.M)) body(std::true_type{}, std::true_type{});
  else if (!wide_cur || (*wide_cur & 1u) != 0u) body(std::true_type{}, std::false_type{});
  else body(std::false_type{}, std::false_type{});
}

}  // namespace earhip
