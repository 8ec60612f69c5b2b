// curves.h — host mirror + device image of the gain curves of M input channels
// (libear's GainInterpolator::interp_points, gain_interpolator.hpp:27-43), and
// the launcher of K0/K1 over them.
#pragma once

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"
#include "gain_kernels.h"
#include "gain_p2.h"
#include "gain_hg.h"
#include "gain_f32g.h"

namespace earhip {

// column tiling of the K*N output columns over waves
struct ColumnPlan {
  // VALU kernel (k_gain_mix): SGPR-operand FMAs
  int nout = 24;     // columns per wave (8, 16 or 24)
  int ngroups = 1;   // column groups per workgroup
  int nz = 1;        // column super-groups (grid.z)
  // MFMA kernel (k_gain_mix_mfma): 16-column tiles
  int nct = 1;       // column tiles per wave (1..3)
  int mgroups = 1;   // column groups per workgroup
  int mnz = 1;       // column super-groups (grid.z)
  int row = 24;      // padded row length, covers both tilings
  static ColumnPlan make(int ncols) {
    ColumnPlan p;
    int groups = (ncols + 23) / 24;
    const int per = (ncols + groups - 1) / groups;
    p.nout = per <= 8 ? 8 : per <= 16 ? 16 : 24;
    groups = (ncols + p.nout - 1) / p.nout;
    p.ngroups = std::min(groups, 8);
    p.nz = (groups + p.ngroups - 1) / p.ngroups;
    int mg = (ncols + 47) / 48;
    p.nct = mg > 1 ? 3 : (ncols + 15) / 16;
    p.mgroups = std::min(mg, 8);
    p.mnz = (mg + p.mgroups - 1) / p.mgroups;
    p.row = std::max(p.nz * p.ngroups * p.nout, p.mnz * p.mgroups * p.nct * 16);
    return p;
  }
};

// k_curve_upload: the changed objects' regions of the curve image, host mirror (pinned, device-visible) -> device, in ONE
// launch: a commit that touches 16 of 1024 objects costs the host one kernel launch, not 64 copy calls, and the device
// what those objects' points weigh (CurveSet::commit).  grid = changed objects, block = 256 threads.
struct CurveImage {
  ObjHdr *hdr;
  int32_t *off, *cnt;
  int64_t *time;
  uint8_t *flat;
  PointRec *rec;
  float *gain;
};
static __global__ void __launch_bounds__(256)
k_curve_upload(CurveImage src, CurveImage dst, const int32_t *changed, int row) {
  const int m = changed[blockIdx.x];
  const int off = src.off[m], n = src.cnt[m];
  if (threadIdx.x == 0) dst.off[m] = off, dst.cnt[m] = n, dst.hdr[m] = src.hdr[m];
  for (int i = threadIdx.x; i < n; i += 256) {
    dst.time[off + i] = src.time[off + i];
    dst.flat[off + i] = src.flat[off + i];
    dst.rec[off + i] = src.rec[off + i];
  }
  const size_t g0 = (size_t)off * row, gn = (size_t)n * row;  // (row is a multiple of 4: float4 copies)
  const float4 *sg = reinterpret_cast<const float4 *>(src.gain + g0);
  float4 *dg = reinterpret_cast<float4 *>(dst.gain + g0);
  for (size_t i = threadIdx.x; i < gn / 4; i += 256) dg[i] = sg[i];
}

// The gain curves of M objects: libear's GainInterpolator::interp_points per object (gain_interpolator.hpp:42-43: a
// public vector the caller changes freely), as a device image the kernels read and a host mirror of it.
//
// Image: an ARENA of curve points — rows 0 and 1 are all zero (the row objects to skip point at, and the row behind it
// that a null ramp reads as its end point), every object owns a region [off, off + cap) of which cnt points are live;
// per point its time, the packed record of the segment that ends there, the per-bus "equals the previous point"
// bits and the gain row.  An object that outgrows its region moves to the end of the arena (regions grow by half;
// the arena is re-packed when more than half of it is abandoned).
//
// commit() costs what CHANGED: the changed objects' points are written to the mirror, their statistics (time grid, ramp
// share, hinge / pair statistics, column maxima) replace their old contributions to the set's, and their regions go
// to the device in one launch of k_curve_upload (more than a quarter of the objects changed, or buffers grown: plain
// copies of the whole arrays).  1024 objects x 64-point windows, 16 objects replaced per block: tools/commit_cost.py.
class CurveSet {
 public:
  // ncols = nbus * bus_cols; each bus is one libear GainInterpolator per object
  CurveSet(int M, int ncols, int nbus = 1, bool force_ramp = false)
      : M_(M), ncols_(ncols), nbus_(nbus), force_ramp_(force_ramp), plan_(ColumnPlan::make(ncols)), obj_(M) {
    // an object without a curve is silent: one all-zero point
    for (int m = 0; m < M; m++) {
      obj_[m].t.assign(1, 0);
      obj_[m].g.assign(plan_.row, 0.0f);
      obj_[m].flat.assign(1, 0);
      obj_[m].dirty = true;
      dirty_list_.push_back(m);
    }
    cmax_.assign(plan_.row, 0.0f);
    obj_cmax_.assign((size_t)M * plan_.row, 0.0f);
    cm_scratch_.assign(plan_.row, 0.0f);
    shrunk_cols_.assign(plan_.row, 0);
    dirty_list_.reserve(M);
    for (auto &p : phases_) p.items.reserve(M);
  }
  ~CurveSet() {
    if (staged_) (void)hipEventDestroy(staged_);
  }
  int M() const { return M_; }
  int ncols() const { return ncols_; }
  const ColumnPlan &plan() const { return plan_; }

  // rows: [npoints][ncols] for this object.  flat_override (optional, [npoints]):
  // caller-decided "point k equals point k-1" bits, used when libear compares a
  // larger point than this object's row (LinearInterpMatrix: the whole matrix).
  void set_object(int m, int npoints, const int64_t *times, const float *rows,
                  const uint8_t *flat_override = nullptr) {
    if (m < 0 || m >= M_) fail_invalid("object index out of range");
    if (npoints < 1) fail_invalid("interp_points must not be empty");
    if (npoints > kMaxPointsPerObject) fail_invalid("too many interpolation points for one object");
    for (int k = 1; k < npoints; k++) {
      if (times[k] < times[k - 1]) fail_invalid("interpolation points are not sorted");
      if (times[k] - times[k - 1] > (int64_t)0x7fffffff)
        fail_invalid("interpolation ramp longer than 2^31-1 samples is not supported");
    }
    Obj &o = obj_[m];
    const size_t row = (size_t)plan_.row;
    o.t.assign(times, times + npoints);
    o.g.assign((size_t)npoints * row, 0.0f);
    for (int k = 0; k < npoints; k++)
      std::memcpy(&o.g[(size_t)k * row], rows + (size_t)k * ncols_, sizeof(float) * ncols_);
    o.flat.assign(npoints, 0);
    const int bc = ncols_ / nbus_;
    for (int k = 1; k < npoints; k++) {
      if (flat_override) {
        o.flat[k] = flat_override[k];
        continue;
      }
      // InterpType::constant_interp: a == b on every component
      // (gain_interpolator.hpp:141-143); float ==, so -0 == +0 and NaN != NaN
      for (int b = 0; b < nbus_; b++) {
        bool same = true;
        for (int c = b * bc; same && c < (b + 1) * bc; c++)
          same = rows[(size_t)(k - 1) * ncols_ + c] == rows[(size_t)k * ncols_ + c];
        if (same) o.flat[k] |= (uint8_t)(1 << b);
      }
    }
    // A point that repeats its predecessor — the same time and bit for bit the same gains — is an empty segment
    // between equal values: GainInterpolator::process never applies it (gain_interpolator.hpp:58-86 skips empty
    // ranges) and the segment behind it starts from the same values either way.  Metadata written block by block
    // has one at every block boundary (the end of a block's ramp, then the start of the next); kept, the end row
    // of one tile and the start row of the next are two rows in memory instead of one fetched once (the panned
    // scene's gain kernel: 0.43 -> 0.40 ms), and the curve takes twice the space.
    {
      int w = 1;
      for (int k = 1; k < npoints; k++) {
        const bool repeat = o.t[k] == o.t[w - 1] && std::memcmp(&o.g[(size_t)k * row], &o.g[(size_t)(w - 1) * row], sizeof(float) * row) == 0;
        if (repeat) continue;
        if (w != k) {
          o.t[w] = o.t[k];
          o.flat[w] = o.flat[k];
          std::memmove(&o.g[(size_t)w * row], &o.g[(size_t)k * row], sizeof(float) * row);
        }
        w++;
      }
      o.t.resize(w);
      o.flat.resize(w);
      o.g.resize((size_t)w * row);
    }
    if (!o.dirty) o.dirty = true, dirty_list_.push_back(m);
  }

  // Bring the device image up to date.  Uploads run on the context's stream: kernels already enqueued still see the old
  // image (stream order), the caller does not wait for the GPU.  Only growing a device buffer, or touching the
  // mirror while its last upload is still in flight, synchronises.
  void commit(earhip_ctx *ctx) {
    if (dirty_list_.empty()) return;
#ifdef EARHIP_COMMIT_TIMING  // (-DEARHIP_COMMIT_TIMING: where a commit's time goes, printed at exit; tools/commit_cost.py)
    static double acc[6] = {0, 0, 0, 0, 0, 0};
    static long ncalls = 0;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tq = now();
    auto lap = [&](int i) { const double t = now(); acc[i] += t - tq; tq = t; };
    struct Print { ~Print() { if (ncalls) fprintf(stderr, "commit timing (us/call over %ld): sync %.2f regions %.2f mirror+stats %.2f finish %.2f upload %.2f tail %.2f\n", ncalls, acc[0]/ncalls, acc[1]/ncalls, acc[2]/ncalls, acc[3]/ncalls, acc[4]/ncalls, acc[5]/ncalls); } };
    static Print printer;
    ncalls++;
#define EARHIP_LAP(i) lap(i)
#else
#define EARHIP_LAP(i)
#endif
    const size_t row = (size_t)plan_.row;
    if (staged_) EARHIP_HIP(hipEventSynchronize(staged_));  // the mirror is free again
    EARHIP_LAP(0);
    // ---- regions: an object that outgrew its region moves to the end of the arena
    bool full_upload = !uploaded_once_;
    for (int m : dirty_list_) {
      Obj &o = obj_[m];
      const int n = (int)o.t.size();
      if (n > o.cap) {
        abandoned_ += (size_t)o.cap;
        o.cap = std::max(16, n + n / 2);
        o.off = (int)arena_used_;
        arena_used_ += (size_t)o.cap;
      }
    }
    if (abandoned_ > arena_used_ / 2 && abandoned_ > 4096) {  // re-pack: every object into a fresh region, in order
      arena_used_ = 2;
      abandoned_ = 0;
      for (int m = 0; m < M_; m++) {
        Obj &o = obj_[m];
        o.cap = std::max(16, (int)o.t.size() + (int)o.t.size() / 2);
        o.off = (int)arena_used_;
        arena_used_ += (size_t)o.cap;
        if (!o.dirty) o.dirty = true, dirty_list_.push_back(m);
      }
      full_upload = true;
    }
    if (arena_used_ >= ((size_t)1 << 31)) fail_invalid("too many interpolation points");
    if (arena_used_ > arena_cap_) {  // grow the buffers (host mirror and device image): everything is copied again
      const size_t cap = arena_used_ + arena_used_ / 2 + 64;
      EARHIP_HIP(hipStreamSynchronize(ctx->stream));  // the old image may still be in use
      PinBuf<int64_t> nt;
      PinBuf<uint8_t> nf;
      PinBuf<PointRec> nr;
      PinBuf<float> ng;
      nt.reserve(cap), nf.reserve(cap), nr.reserve(cap), ng.reserve(cap * row);
      if (arena_cap_) {
        std::memcpy(nt.p, h_time_.p, arena_cap_ * sizeof(int64_t));
        std::memcpy(nf.p, h_flat_.p, arena_cap_);
        std::memcpy(nr.p, h_rec_.p, arena_cap_ * sizeof(PointRec));
        std::memcpy(ng.p, h_gain_.p, arena_cap_ * row * sizeof(float));
      } else {
        std::memset(ng.p, 0, 2 * row * sizeof(float));  // rows 0, 1: all zero
        std::memset(nt.p, 0, 2 * sizeof(int64_t));
        std::memset(nf.p, 0, 2);
        std::memset(nr.p, 0, 2 * sizeof(PointRec));
      }
      std::swap(h_time_.p, nt.p), std::swap(h_time_.n, nt.n);  // (the old buffers are released with nt .. ng)
      std::swap(h_flat_.p, nf.p), std::swap(h_flat_.n, nf.n);
      std::swap(h_rec_.p, nr.p), std::swap(h_rec_.n, nr.n);
      std::swap(h_gain_.p, ng.p), std::swap(h_gain_.n, ng.n);
      d_time_.alloc(cap), d_flat_.alloc(cap), d_rec_.alloc(cap), d_gain_.alloc(cap * row * (kinks_ ? 2 : 1));
      arena_cap_ = cap;
      full_upload = true;
    }
    if (!h_off_.p) {
      h_off_.reserve(M_), h_cnt_.reserve(M_), h_changed_.reserve(M_), h_gcol_.reserve(row), h_hdr_.reserve(M_);
      d_off_.alloc(M_), d_cnt_.alloc(M_), d_gcol_.alloc(row), d_hdr_.alloc(M_);
    }
    EARHIP_LAP(1);
    // ---- the changed objects: mirror, statistics
    const float old_scale = gain_scale();
    bool cmax_shrunk = false;
    for (int m : dirty_list_) {
      Obj &o = obj_[m];
      const int n = (int)o.t.size();
      h_off_.p[m] = o.off;
      h_cnt_.p[m] = n;
      h_hdr_.p[m] = ObjHdr{o.off, n, o.t[0], o.t[n - 1], 0};
      std::memcpy(h_time_.p + o.off, o.t.data(), (size_t)n * sizeof(int64_t));
      std::memcpy(h_flat_.p + o.off, o.flat.data(), (size_t)n);
      std::memcpy(h_gain_.p + (size_t)o.off * row, o.g.data(), (size_t)n * row * sizeof(float));
      for (int k = 0; k < n; k++) {
        PointRec &r = h_rec_.p[o.off + k];
        r.time = o.t[k];
        const int64_t len = k ? o.t[k] - o.t[k - 1] : 0;
        r.scale = len > 0 ? 1.0f / (float)len : 0.0f;  // describe_segment's own expression (gain_kernels.h)
        r.flat = o.flat[k];
      }
      float *cm = obj_cmax_.data() + (size_t)m * row;
      if (o.has_stats) cmax_shrunk = retire(o.st, cm) || cmax_shrunk;
      o.st = stats_of(o, cm);
      admit(o.st, cm);
      o.has_stats = true;
    }
    // A column's largest gain went away with an old curve: that column's maximum is looked up again among all objects, now
    // (O(M) for that column; round 4 put it off for up to 256 commits, during which a column left with gains around
    // -120 dB kept the scale of the loud one and lost the low f16 piece of its gains to subnormals).
    if (cmax_shrunk) {
      for (size_t c = 0; c < row; c++) {
        if (!shrunk_cols_[c]) continue;
        shrunk_cols_[c] = 0;
        float v = 0.0f;
        for (int m = 0; m < M_; m++) {
          const float a = obj_cmax_[(size_t)m * row + c];
          v = a <= v ? v : a;  // (a NaN replaces the maximum)
        }
        cmax_[c] = v;
      }
    }
    EARHIP_LAP(2);
    finish_stats();
    // per-column gain scales (the split-operand kernels scale every output column's gains by its own power of two: a
    // loudspeaker that only ever gets small gains — an object 120 dB down alone on it — keeps both f16 pieces of its
    // gains normal; the inverse is applied to the column's output).  Columns without any gain: the set's scale.
    const float set_scale = gain_scale();
    bool gcol_changed = !uploaded_once_ || set_scale != old_scale;
    for (size_t c = 0; c < row; c++) {
      const float cm = cmax_[c];
      int e;
      std::frexp(cm, &e);
      const float v = (cm >= 1e-30f && cm < 1e30f) ? std::ldexp(1.0f, 14 - e) : (set_scale > 0.0f ? set_scale : 1.0f);
      gcol_changed = gcol_changed || v != h_gcol_.p[c];
      h_gcol_.p[c] = v;
    }
    EARHIP_LAP(3);
    // ---- upload
    if (gcol_changed) EARHIP_HIP(hipMemcpyAsync(d_gcol_.p, h_gcol_.p, row * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (full_upload || dirty_list_.size() > (size_t)M_ / 4) {
      EARHIP_HIP(hipMemcpyAsync(d_off_.p, h_off_.p, M_ * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
      EARHIP_HIP(hipMemcpyAsync(d_cnt_.p, h_cnt_.p, M_ * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
      EARHIP_HIP(hipMemcpyAsync(d_hdr_.p, h_hdr_.p, M_ * sizeof(ObjHdr), hipMemcpyHostToDevice, ctx->stream));
      EARHIP_HIP(hipMemcpyAsync(d_time_.p, h_time_.p, arena_used_ * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
      EARHIP_HIP(hipMemcpyAsync(d_flat_.p, h_flat_.p, arena_used_, hipMemcpyHostToDevice, ctx->stream));
      EARHIP_HIP(hipMemcpyAsync(d_rec_.p, h_rec_.p, arena_used_ * sizeof(PointRec), hipMemcpyHostToDevice, ctx->stream));
      EARHIP_HIP(hipMemcpyAsync(d_gain_.p, h_gain_.p, arena_used_ * row * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    } else {
      for (size_t i = 0; i < dirty_list_.size(); i++) h_changed_.p[i] = dirty_list_[i];
      CurveImage src{h_hdr_.p, h_off_.p, h_cnt_.p, h_time_.p, h_flat_.p, h_rec_.p, h_gain_.p};
      CurveImage dst{d_hdr_.p, d_off_.p, d_cnt_.p, d_time_.p, d_flat_.p, d_rec_.p, d_gain_.p};
      hipLaunchKernelGGL(k_curve_upload, dim3((unsigned)dirty_list_.size()), dim3(256), 0, ctx->stream, src, dst, h_changed_.p,
                         (int)row);
      EARHIP_HIP(hipGetLastError());
    }
    EARHIP_LAP(4);
    // kink rows (hinge kernel; once it has been planned for this set): of the changed objects, or of everybody when
    // everything went up again or a column's scale changed
    if (kinks_) {
      const bool all = full_upload || gcol_changed || dirty_list_.size() > (size_t)M_ / 4;
      if (!all)  // (h_changed_ holds the list: the partial upload above wrote it)
        for (size_t i = 0; i < dirty_list_.size(); i++) h_changed_.p[i] = dirty_list_[i];
      derive_kinks(ctx, all ? nullptr : h_changed_.p, all ? M_ : (int)dirty_list_.size());
    }
    if (!staged_) EARHIP_HIP(hipEventCreateWithFlags(&staged_, hipEventDisableTiming));
    EARHIP_HIP(hipEventRecord(staged_, ctx->stream));
    for (int m : dirty_list_) obj_[m].dirty = false;
    dirty_list_.clear();
    uploaded_once_ = true;
    EARHIP_LAP(5);
  }
#undef EARHIP_LAP

  // The hinge kernel has been planned for this set: from now on it keeps a kink row per point (gain_hg.h) behind the
  // gain rows, in the same buffer (twice its size; sets that never meet that kernel never pay for it).  Called with
  // everything committed; the first call re-allocates and sends the gain rows again.
  void ensure_kinks(earhip_ctx *ctx) {
    if (kinks_) return;
    kinks_ = true;
    const size_t row = (size_t)plan_.row;
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));  // the old image may still be in use
    d_gain_.alloc(arena_cap_ * row * 2);
    EARHIP_HIP(hipMemcpyAsync(d_gain_.p, h_gain_.p, arena_used_ * row * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    derive_kinks(ctx, nullptr, M_);
    if (!staged_) EARHIP_HIP(hipEventCreateWithFlags(&staged_, hipEventDisableTiming));
    EARHIP_HIP(hipEventRecord(staged_, ctx->stream));
  }
  bool has_kinks() const { return kinks_; }
  bool dirty() const { return !dirty_list_.empty(); }  // set_object since the last commit

  // true when no curve point can fall strictly inside a tile of `tile` samples of a
  // call that starts at t_call (all point times lie on tile boundaries)
  bool tiles_aligned(int tile, int64_t t_call) const {
    const int64_t d = t_call >= t_ref_ ? t_call - t_ref_ : t_ref_ - t_call;
    return d % tile == 0 && grid_ % (uint64_t)tile == 0;
  }

  // largest tile of the split-operand kernels (512, 256) with no curve point strictly inside; 0: none
  // A few objects off the grid (at most M / 64; fewer, down to none, when the curves hold most of the time: below) are tolerated: the split-operand kernels send an object
  // with a point inside a tile through their exact slow path for that tile only — 10.7 us per such object and
  // 1024-block call on the headline scene (2.7 % of K1; 8 objects: step 0.546 ms against 0.460), break-even with
  // moving the whole scene to the hinge kernel (0.61) at ~14 objects, to the piece lists (0.67) at ~20
  // (tools/r6_mixed.sh; until round 6 the bound was M / 32, set against the f32 slot kernel's 1.08 ms).
  int aligned_tile(int64_t t_call) const {
    const int grids[2] = {512, 256};
    for (int gi = 0; gi < 2; gi++) {
      const int64_t G = grids[gi];
      // Curves that hold most of the time have the cheaper alternative: the paired piece lists cost the grid kernel's time plus
      // what their ramps' delta pieces add — static gains with a few moving objects: 0.43 ms per step on the lists whatever the
      // number of movers, against 0.46 / 0.47 / 0.49 on the grid kernel with 2 / 4 / 6 of them on its exact path; ADM-style curves
      // (0.8 delta pieces per object and tile): 0.53 on the lists, worth ~4 objects.  So their bound follows the delta pieces per
      // (object, tile) pair of the WHOLE set (tools/r6_mixed2.sh).  (stats index: [0] 256-sample tiles, [1] 512)
      const int si = 1 - gi;
      int off_grid_limit = M_ / 64;
      if (ramp_share() < 0.5) {
        const double deltas = deltas_per_pair(si ? 512 : 256);
        off_grid_limit = (int)(M_ * std::min(1.0 / 64, deltas / 200.0));
      }
      if (tiles_aligned((int)G, t_call)) return (int)G;
      if (grid_off_[gi] <= off_grid_limit && (((t_call - grid_phase_[gi]) % G) + G) % G == 0) return (int)G;
    }
    return 0;
  }

  // ramp pieces (ramp x tile incidences) per (object, tile) pair of the WHOLE set (objects with one point have none); tile = 256 or 512
  double deltas_per_pair(int tile) const {
    const int gi = tile >= 512 ? 1 : 0;
    return tot_.tiles[gi] > 0 ? tot_.incid[gi] / tot_.tiles[gi] * moving_share() : 0.0;
  }
  // share of the objects that have more than one point (a bed of static gains has none)
  double moving_share() const { return M_ > 0 ? (double)tot_.moving / M_ : 0.0; }
  // curve points per sample and object of the whole set
  double point_density_all() const { return point_density() * moving_share(); }
  // fraction of the curves' time in ramps; 0 for static gains
  double ramp_share() const { return tot_.span > 0 ? tot_.ramp / tot_.span : 0.0; }
  // curve points per sample and object (0 for static gains)
  double point_density() const { return tot_.span > 0 ? tot_.npts / tot_.span : 0.0; }
  // The piece-list kernel's paired layout spends two slots on every delta piece (gain_p2.h): an object with k > 1
  // ramps in one tile costs k - 1 slots more than in the packed layout.  Share of such extra slots among the slots
  // of the packed layout, over the curves' whole span, on grids of 256- and of 512-sample tiles from time 0 (a
  // call's grid starts at its own first sample: same statistics).
  double pair_waste(int tile) const {
    const int gi = tile >= 512 ? 1 : 0;
    return (tot_.incid[gi] - tot_.touched[gi]) / std::max(1.0, tot_.tiles[gi] + tot_.incid[gi]);
  }
  // What the hinge kernel (gain_hg.h) would have to send through its exact path: (object, tile) pairs with a ramp
  // shorter than kHingeMinLen (a step: a ramp of no length) or with curve points closer together than that — as a share
  // of all pairs on 256-sample tiles (a short ramp spoils the tiles it touches, two close points on a constant stretch
  // the tile they share about every other time); > 1: not a curve set for it.
  // (in_stride, nsamples: the input rows of the call — the kernel addresses them, and the gain rows, with 32-bit offsets)
  double hinge_exact_share(size_t in_stride, size_t nsamples) const {
    if (force_ramp_ || !hinge_addressable((size_t)M_, in_stride, nsamples, 2 * arena_cap_, (size_t)plan_.row)) return 2.0;
    return tot_.span > 0 ? std::min(1.0, tot_.bad / std::max(1.0, tot_.span / kHingeTile)) : 0.0;
  }

  // Slots of ONE tile's piece list (gain_p2.h) that hold it whatever the tile: every object's base piece and as many delta
  // pieces as its curve can have ramps in one window of `tile` samples (objects beyond the per-object limits take the exact
  // path and have none), in either layout, padding included.  (Round 3 sized the lists for 15 ramps per object and tile:
  // 537 MB at the headline's size whatever the curves.)
  int piece_cap(int tile, bool paired) const {
    const int gi = tile >= 512 ? 1 : 0;
    const long slots = paired ? (long)((M_ + 63) & ~63) + 2 * tot_.sum_r7[gi] : (long)M_ + tot_.sum_r15[gi];
    return (int)std::min<long>(slots + 128, (long)kPieceCapPerObject * M_ + 128);
  }

  // 2^k with 2^k * |gain| <= 2^14 for every gain of the set (slopes and differences of two gains stay
  // below the f16 limit 65504); 0 when the gains are not finite or beyond what a scale can fix
  float gain_scale() const {
    if (!(gain_max_ < 1e30f)) return 0.0f;
    if (gain_max_ < 1e-18f) return 1.0f;  // (nothing audible to scale)
    int e;
    std::frexp(gain_max_, &e);  // gain_max_ = f * 2^e, 0.5 <= f < 1
    return std::ldexp(1.0f, 14 - e);
  }

  // [row] per-column gain scales of the split-operand kernels (powers of two; device memory)
  const float *column_scales() const { return d_gcol_.p; }

  PointStore device() const {
    PointStore ps;
    ps.hdr = d_hdr_.p;
    ps.off = d_off_.p;
    ps.cnt = d_cnt_.p;
    ps.time = d_time_.p;
    ps.flat = d_flat_.p;
    ps.rec = d_rec_.p;
    ps.gain = d_gain_.p;
    ps.row = plan_.row;
    ps.bus_cols = ncols_ / nbus_;
    ps.nbus = nbus_;
    ps.zero_row = 0;
    ps.npoints = (int)tot_.points;
    ps.rows = (int)arena_used_;
    ps.kink_row0 = kinks_ ? (int)arena_cap_ : 0;
    ps.force_ramp = force_ramp_ ? 1 : 0;
    return ps;
  }

 private:
  void derive_kinks(earhip_ctx *ctx, const int32_t *objects, int count) {
    const size_t row = (size_t)plan_.row;
    uint32_t *kink = reinterpret_cast<uint32_t *>(d_gain_.p + arena_cap_ * row);
    // (rows 0 and 1 of the kink image: all zero, like the gain rows they stand behind)
    EARHIP_HIP(hipMemsetAsync(kink, 0, 2 * row * sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(k_kink_rows, dim3((unsigned)count), dim3(256), 0, ctx->stream, d_off_.p, d_cnt_.p, d_rec_.p, d_gain_.p, kink,
                       d_gcol_.p, objects, (int)row, nbus_);
    EARHIP_HIP(hipGetLastError());
  }
  // what ONE object's curve contributes to the statistics of the set
  struct ObjStats {
    uint64_t gcd = 0;        // of the differences of its point times
    int64_t first = 0;       // its first point's time
    int64_t phase[2] = {-1, -1};  // modulo 512 / 256 when all its points share one; -1: they do not
    double span = 0, ramp = 0, npts = 0, bad = 0, points = 0;
    double incid[2] = {0, 0}, touched[2] = {0, 0}, tiles[2] = {0, 0};
    int maxr[2] = {0, 0};  // the most ramps of this curve that any window of 256 / 512 samples overlaps
    // (its column maxima: obj_cmax_[m][row] — commit() runs inside process calls and must not allocate)
  };
  struct Obj {
    std::vector<int64_t> t;
    std::vector<float> g;
    std::vector<uint8_t> flat;
    int off = 0, cap = 0;  // region of the arena
    bool dirty = false, has_stats = false;
    ObjStats st;
  };

  // cm: [row], the object's column maxima (largest |gain| per column; NaN / inf count as huge)
  ObjStats stats_of(const Obj &o, float *cm) const {
    ObjStats s;
    const auto &t = o.t;
    const size_t row = (size_t)plan_.row, n = t.size();
    const int allflat = (1 << nbus_) - 1;
    s.first = t[0];
    s.points = (double)n;
    for (size_t k = 1; k < n && s.gcd != 1; k++) {
      uint64_t a = (uint64_t)(t[k] - t[k - 1]), b = s.gcd;
      while (a) {
        const uint64_t r = b % a;
        b = a;
        a = r;
      }
      s.gcd = b;
    }
    const int64_t grids[2] = {512, 256};
    for (int gi = 0; gi < 2; gi++) {
      const int64_t G = grids[gi];
      const int64_t p0 = ((t[0] % G) + G) % G;
      bool same = s.gcd % (uint64_t)G == 0;  // (all differences multiples of G: one phase)
      s.phase[gi] = same ? p0 : -1;
    }
    for (size_t k = 1; k < n; k++) {
      const int64_t len = t[k] - t[k - 1];
      const bool is_ramp = (o.flat[k] & allflat) != allflat;
      s.span += (double)len;
      if (is_ramp) s.ramp += (double)len;
      if (len < kHingeMinLen) s.bad += is_ramp ? 1.0 + (double)len / kHingeTile : 0.5;
    }
    if (n > 1) s.npts = (double)(n - 1);
    for (int gi = 0; gi < 2; gi++) {
      const int64_t T = gi ? 512 : 256;
      auto tile_of = [T](int64_t a) { return a >= 0 ? a / T : -((-a + T - 1) / T); };
      if (n > 1) s.tiles[gi] = (double)(t.back() - t.front()) / (double)T;
      int64_t last_tile = INT64_MIN;
      for (size_t k = 1; k < n; k++) {
        if ((o.flat[k] & allflat) == allflat) continue;
        // a step (two equal times) is a ramp of length one ending at the time
        const int64_t a = t[k] > t[k - 1] ? t[k - 1] : t[k] - 1, b = t[k];  // the ramp covers samples [a, b)
        const int64_t ta = tile_of(a), tb = tile_of(b - 1);
        s.incid[gi] += (double)(tb - ta + 1);
        s.touched[gi] += (double)(tb - ta + 1) - (ta == last_tile ? 1.0 : 0.0);
        last_tile = tb;
      }
    }
    // the most ramps any window of T samples overlaps (a bound for the delta pieces of this object in any tile of a
    // piece list: the lists are sized from the sum of these, piece_cap()); ramp k covers samples [a_k, b_k), a step is
    // a ramp of length one ending at its time — two pointers over the ramps in time order
    for (int gi = 0; gi < 2; gi++) {
      const int64_t T = gi ? 512 : 256;
      auto is_r = [&](size_t k) { return (o.flat[k] & allflat) != allflat; };
      auto ra = [&](size_t k) { return t[k] > t[k - 1] ? t[k - 1] : t[k] - 1; };
      int best = 0, inside = 0;
      size_t j = 1;
      for (size_t i = 1; i < n; i++) {  // windows that start on the last sample of ramp i: [b_i - 1, b_i - 1 + T)
        if (!is_r(i)) continue;
        if (j < i) j = i, inside = 0;
        while (j < n && (!is_r(j) || ra(j) < t[i] - 1 + T)) {
          if (is_r(j)) inside++;
          j++;
        }
        best = std::max(best, inside);
        inside--;  // ramp i leaves the window before the next one's
      }
      s.maxr[gi] = force_ramp_ ? 1 : best;
    }
    for (size_t c = 0; c < row; c++) cm[c] = 0.0f;
    for (size_t k = 0; k < n; k++) {
      const float *g = o.g.data() + k * row;
      for (size_t c = 0; c < row; c++) {
        const float a = std::fabs(g[c]);
        cm[c] = a <= cm[c] ? cm[c] : a;  // (a NaN replaces the maximum)
      }
    }
    return s;
  }
  // take an object's old contribution out of the totals; true when one of its column maxima was the set's
  bool retire(const ObjStats &s, const float *cm) {
    tot_.span -= s.span, tot_.ramp -= s.ramp, tot_.npts -= s.npts, tot_.bad -= s.bad, tot_.points -= s.points;
    tot_.moving -= s.npts > 0 ? 1 : 0;
    for (int gi = 0; gi < 2; gi++) {
      tot_.incid[gi] -= s.incid[gi], tot_.touched[gi] -= s.touched[gi], tot_.tiles[gi] -= s.tiles[gi];
      tot_.sum_r15[gi] -= std::min(s.maxr[gi], kPieceMaxPerObject), tot_.sum_r7[gi] -= std::min(s.maxr[gi], kPairMaxPerObject);
      if (s.phase[gi] >= 0) phases_[gi].add(s.phase[gi], -1);
    }
    grid_stale_ = true;  // (a gcd cannot be un-done: recomputed over the objects' own, finish_stats)
    bool shrunk = false;
    for (size_t c = 0; c < cmax_.size(); c++) {
      // (a NaN maximum — cm != cm — goes away with its object as well)
      const bool was_max = cm[c] != cm[c] || (cm[c] > 0.0f && !(cm[c] < cmax_[c]));
      if (was_max) shrunk_cols_[c] = 1, shrunk = true;
    }
    return shrunk;
  }
  void admit(const ObjStats &s, const float *cm) {
    tot_.span += s.span, tot_.ramp += s.ramp, tot_.npts += s.npts, tot_.bad += s.bad, tot_.points += s.points;
    tot_.moving += s.npts > 0 ? 1 : 0;
    for (int gi = 0; gi < 2; gi++) {
      tot_.incid[gi] += s.incid[gi], tot_.touched[gi] += s.touched[gi], tot_.tiles[gi] += s.tiles[gi];
      tot_.sum_r15[gi] += std::min(s.maxr[gi], kPieceMaxPerObject), tot_.sum_r7[gi] += std::min(s.maxr[gi], kPairMaxPerObject);
      if (s.phase[gi] >= 0) phases_[gi].add(s.phase[gi], +1);
    }
    grid_stale_ = true;
    for (size_t c = 0; c < cmax_.size(); c++) cmax_[c] = cm[c] <= cmax_[c] ? cmax_[c] : cm[c];
  }
  void finish_stats() {
    // common grid of all point times: every time is t_ref_ + k * grid_ (grid_ = 0: all equal) — from the objects' own
    // gcds and first points: O(M) small steps, only when some object changed
    if (grid_stale_) {
      t_ref_ = obj_[0].st.first;
      uint64_t g = 0;
      auto fold = [&g](uint64_t a) {
        uint64_t b = g;
        while (a) {
          const uint64_t r = b % a;
          b = a;
          a = r;
        }
        g = b;
      };
      for (int m = 0; m < M_ && g != 1; m++) {
        const ObjStats &s = obj_[m].st;
        fold(s.gcd);
        fold((uint64_t)(s.first >= t_ref_ ? s.first - t_ref_ : t_ref_ - s.first));
      }
      grid_ = g;
      grid_stale_ = false;
    }
    // per grid size of the split-operand kernels, tolerant of a few objects: an object is ON the grid G when all its
    // points are congruent modulo G to the phase most objects share (objects off the grid only cost their own slow
    // path in the tiles where their points fall)
    for (int gi = 0; gi < 2; gi++) {
      int64_t best = 0;
      int best_n = 0;
      for (const auto &kv : phases_[gi].items)
        if (kv.second > best_n) best_n = kv.second, best = kv.first;
      grid_phase_[gi] = best;
      grid_off_[gi] = M_ - best_n;
    }
    float gmax = 0.0f;
    for (float v : cmax_) gmax = v <= gmax ? gmax : v;
    gain_max_ = gmax;
  }

  int M_, ncols_, nbus_;
  bool force_ramp_;
  ColumnPlan plan_;
  std::vector<Obj> obj_;
  std::vector<int> dirty_list_;
  bool uploaded_once_ = false;
  size_t arena_used_ = 2, arena_cap_ = 0, abandoned_ = 0;  // in points (rows 0, 1: the all-zero rows)
  // statistics of the set
  struct Totals {
    double span = 0, ramp = 0, npts = 0, bad = 0, points = 0;
    long moving = 0;  // objects with more than one point
    double incid[2] = {0, 0}, touched[2] = {0, 0}, tiles[2] = {0, 0};
    long sum_r15[2] = {0, 0}, sum_r7[2] = {0, 0};  // sums of the objects' maxr, clipped to what a list takes per object (packed / paired)
  } tot_;
  // [512, 256]: how many objects have all their points on that phase (at most M distinct ones: room for them from the
  // start — commit() must not allocate)
  struct PhaseCounts {
    std::vector<std::pair<int64_t, int>> items;
    void add(int64_t phase, int d) {
      for (size_t i = 0; i < items.size(); i++)
        if (items[i].first == phase) {
          items[i].second += d;
          if (items[i].second <= 0) items[i] = items.back(), items.pop_back();
          return;
        }
      if (d > 0) items.emplace_back(phase, d);
    }
  } phases_[2];
  std::vector<float> obj_cmax_;  // [M][row] the objects' column maxima
  std::vector<float> cm_scratch_;  // [row]
  std::vector<uint8_t> shrunk_cols_;  // [row] columns whose maximum went away with a retired curve (commit)
  bool grid_stale_ = true;
  int64_t t_ref_ = 0;
  uint64_t grid_ = 0;
  int64_t grid_phase_[2] = {0, 0};  // [512, 256]: the phase most objects' points share
  int grid_off_[2] = {0, 0};        // objects whose points are not all on that phase
  std::vector<float> cmax_;         // [row] largest |gain| per column over the set
  float gain_max_ = 0;
  DevBuf<int32_t> d_off_, d_cnt_;
  DevBuf<ObjHdr> d_hdr_;
  PinBuf<ObjHdr> h_hdr_;
  DevBuf<int64_t> d_time_;
  DevBuf<uint8_t> d_flat_;
  DevBuf<PointRec> d_rec_;
  DevBuf<float> d_gcol_, d_gain_;
  bool kinks_ = false;  // d_gain_ is twice as long: the kink rows behind the gain rows (ensure_kinks)
  PinBuf<float> h_gcol_, h_gain_;
  PinBuf<int32_t> h_off_, h_cnt_, h_changed_;
  PinBuf<int64_t> h_time_;
  PinBuf<uint8_t> h_flat_;
  PinBuf<PointRec> h_rec_;
  hipEvent_t staged_ = nullptr;
};

// How K1 is spread over the chip for one call.
struct MixLaunch {
  bool mfma;                   // matrix-core kernel (default) or VALU kernel (strict mode)
  bool split = false;          // matrix-core kernel on f16x2 split operands (gain_h2.h); tile = 256 samples
  bool wide = false;           // with split: 8 waves on 512-sample tiles
  bool pieces = false;         // matrix-core kernel on f16x2 split operands over per-tile piece lists (gain_p2.h)
  int pw = 2;                  // with pieces: waves per workgroup (2 or 4); tile = 64 pw samples
  bool paired = false;         // with pieces: the lists' paired layout (gain_p2.h)
  bool f32grid = false;        // exact-f32 matrix-core kernel for curves on the 512-sample tile grid (gain_f32g.h)
  bool hinge = false;          // matrix-core kernel on f16x2 split operands with the curve points inside a tile as hinges (gain_hg.h)
  int hinge_tile = 512;        // with hinge: 512 (8 waves, up to two kinks on either side of a tile's centre) or 256 (4 waves, one:
                               // EARHIP_HG_TILE=256; 4 % slower per step, 8.1e-7 instead of 8.6e-7 from the CPU path on the
                               // always-ramping scene at 1024 objects)
  int spl;                     // VALU: samples per lane (2 or 4); tile = 64 * spl samples
  int nrt;                     // MFMA: 16-sample row tiles per wave; tile = 16 * nrt samples
  int ntiles, wsplit, gsplit;  // tiles, in-workgroup object splits, grid-level splits
  int tpw = 1;                 // MFMA: adjacent tiles per workgroup
  int tile() const { return f32grid ? kF32GridTile : hinge ? hinge_tile : pieces ? 64 * pw : split ? (wide ? 512 : 256) : mfma ? 16 * nrt : 64 * spl; }
};

inline MixLaunch plan_mix(const earhip_ctx *ctx, const ColumnPlan &cp, int M, int nsamples,
                          bool strict, int max_gsplit, int aligned_tile = 0, double ramp_share = 1.0,
                          float gain_scale = 0.0f, double point_density = 0.0, double pair_waste256 = 1.0,
                          double pair_waste512 = 1.0, double hinge_exact_share = 2.0, bool grid512_strict = false,
                          double deltas256 = 0.0) {
  const bool aligned = aligned_tile >= 256;  // no curve point inside 256- (512-) sample tiles of the call
  MixLaunch L;
  L.mfma = !strict && ctx->use_mfma;
  // Split-operand kernel (tile = 256 or 512 samples, no curve point inside a tile; 4 forces it where the
  // gains allow): f16x2 when the gains can be scaled into f16 range — its cost does not depend on the
  // curves and is below the f32 slot kernel's even for static gains.  Curve sets with non-finite gains
  // stay on the f32 slot kernel.
  L.split = L.mfma && M >= 32 && gain_scale > 0.0f && ctx->use_mfma != 5 && ctx->use_mfma != 6 &&
            (ctx->use_mfma == 4 || (ctx->use_mfma == 3 && aligned));
  // long calls on a 512 grid: 8-wave workgroups on 512-sample tiles (two rounds of workgroups or more:
  // 512 blocks of 512: K1 0.215 vs 0.236 ms; one round, 256 blocks: 0.127 vs 0.120)
  // (EARHIP_H2_TILE=256|512 forces one of them where the curves allow it: tests, tuning)
  const int forced = ctx->get(OPT_H2_TILE);
  // Piece-list kernel: everything else the f16x2 operands can represent — metadata that ignores the tile
  // grid costs its curve points, not a different kernel (5 forces it for aligned curves as well).
  L.pieces = L.mfma && !L.split && M >= 32 && M <= kMaxPieceObjects && gain_scale > 0.0f &&
             (ctx->use_mfma == 3 || ctx->use_mfma == 5 || ctx->use_mfma == 6);
  if ((ctx->use_mfma == 5 || ctx->use_mfma == 6) && L.pieces) L.split = false;
  if (L.pieces) {
    const int ptile = ctx->get(OPT_P2_TILE);
    // Layout of the lists: paired (an object's base and delta piece share one input request, and the objects without a
    // ramp in a tile skip the position factors) unless objects often have several ramps inside one tile — every ramp
    // beyond the first costs a slot more than in the packed layout (always-ramping curves: a third more chunks).
    // Tile: the paired layout on 512 samples (8 waves) where the curves allow it — as many chunks per sample as on
    // 256, but half the lists to build and half the gain rows to fetch (ADM scene: K0 0.072 -> 0.050 ms, moving point
    // sources on ADM metadata K1 0.463 -> 0.440 as well); the packed layout on 256 (a ramp's delta piece lives on
    // over the whole tile: always-ramping curves K1 0.82 vs 0.97 ms on 512).
    // EARHIP_P2_PAIRS=0|1 and EARHIP_P2_TILE=256|512 force one of them (tests, tuning).
    const double kPairWaste = 0.06;
    // Curves that ramp (nearly) all the time never get it, whatever their waste: with every object ramping in every tile the pair
    // chunks put ~3 M products straight onto the running totals (1.0e-6 from the CPU path at 1024 objects, measured), where the
    // packed layout's chunks sum among themselves first (7.2e-7).  The bound was one half until round 6; ramp-then-hold curves
    // between it and 0.7 (tools/r6_pairs.sh, period 960: ramp 500 / 600 — 8.65e-7 / 8.67e-7 worst channel against the ADM scene's
    // 8.45e-7 at a share of 0.25 and 8.74e-7 at 0.83) run 10 % faster paired than packed (0.567 / 0.578 ms per step against
    // 0.633 / 0.659) and 15 % faster than on the hinge kernel (0.675 / 0.682).
    // (fitted on three column tiles — 24 channels x 2 buses; with one or two the hinge kernel is the cheaper one in that band —
    // 5 channels: 0.457 against 0.490 ms per step — so the old bound stays there: tools/r6_layouts.sh)
    // (and up to 1024 objects, where eight seeds hold 9.1e-7: the paired layout's distance from the CPU path grows with the number
    // of products a running total takes — 2048 objects at a share of 0.62: 1.18e-6 paired, beyond any BASELINE configuration)
    const double kMostlyRamping = cp.nct == 3 && M <= 1024 ? 0.7 : 0.5;
    const bool paired_by_rule = pair_waste256 < kPairWaste && ramp_share < kMostlyRamping;
    L.paired = paired_by_rule;
    if (ctx->has(OPT_P2_PAIRS)) L.paired = ctx->get(OPT_P2_PAIRS) != 0;
    // (short calls — block mode — keep the 256-sample tiles: twice the workgroups)
    const bool long_call = nsamples / 512 >= 2 * ctx->num_cus;
    L.pw = ptile == 512 ? 8 : ptile == 256 ? 4 : (L.paired && long_call && pair_waste512 < kPairWaste) ? 8 : 4;
    // Hinge kernel (gain_hg.h): curves that ramp most of the time in ramps of half a tile or more — every object always on
    // its way to its next target, the points at arbitrary times.  There the piece lists cost an operand split per ramp
    // and tile; the hinge kernel one per object and tile.  Every (object, tile) pair it cannot take (short ramps, steps,
    // points close together) goes through its exact path at ~12 times the cost: such pairs have to be rare.  Curves that
    // hold most of the time (ramp for a while, then constant) stay on the paired piece lists, which skip the position
    // factors of the objects at rest.  6 forces it, EARHIP_HINGE=0 / 1 overrides the choice.
    const double kHingeExact = 0.005, kHingeRamps = 0.5;
    // (its list builder keeps 16 bytes per object and tile in LDS: as many objects as this device's limit holds)
    const int hinge_max = (int)std::min((size_t)kMaxHingeCached, ctx->hinge_build_lds / 16);
    // (curves the paired lists take are theirs: faster there than here in every measured case)
    L.hinge = M <= hinge_max && hinge_exact_share <= kHingeExact && ramp_share >= kHingeRamps && !(paired_by_rule && L.paired);
    if (ctx->has(OPT_HINGE)) L.hinge = ctx->get(OPT_HINGE) != 0 && M <= hinge_max && hinge_exact_share <= 1.0;
    if (ctx->use_mfma == 6) L.hinge = M <= hinge_max && hinge_exact_share <= 1.0;
    if (ctx->use_mfma == 5) L.hinge = false;  // (5 forces the piece lists)
    // Curves that ramp most of the time AND hold in between (ADM blocks whose interpolationLength is 50-80 % of their duration):
    // both kernels take them, and which is faster follows the curves — the hinge kernel pays per curve point, the packed lists per
    // ramp piece.  Fitted on 1024 objects x 24 channels over update periods of 240-1920 samples and ramp shares of 0.5-1
    // (tools/r6_period.sh, r6_rampshare.sh; ms per 1024-block call, K0 + K1): hinge 0.56 + 0.077 x (points per object and 512-sample
    // tile), lists 0.15 + 0.238 x (1 + ramp pieces per object and 256-sample tile).  Always-ramping curves stay on the hinge kernel
    // at every period (0.72 vs 0.88 at 5 ms, 0.58 vs 0.66 at 40 ms); ramp 500 of 960: lists 0.59 vs 0.63 (measured 0.636 / 0.675 per
    // step), ramp 260 of 480: 0.65 vs 0.72 (0.710 / 0.766).  Both counts are per pair of the WHOLE set: a bed of static gains with
    // always-ramping objects among it (one in 4 / 8 / 16: lists 0.579 / 0.532 / 0.514 ms per step, hinge 0.631 / 0.612 / 0.603;
    // one in 2: 0.694 / 0.679).  Within 3 %: the hinge kernel.  Only where nothing forces a kernel.
    // (three column tiles only: with one or two a kink set is one or two thirds of the MFMAs and the hinge kernel wins these
    // cases too — a bed with one fast mover in 8 on 5 / 10 channels: 0.414 / 0.443 ms per step against the lists' 0.489 / 0.505)
    if (L.hinge && !ctx->has(OPT_HINGE) && ctx->use_mfma == 3 && cp.nct == 3 && deltas256 > 0.0 && point_density > 0.0) {
      const double hinge_est = 0.56 + 0.077 * (point_density * 512.0), lists_est = 0.151 + 0.238 * (1.0 + deltas256);
      if (lists_est < 0.97 * hinge_est) L.hinge = false;
    }
    if (L.hinge) {
      if (ctx->has(OPT_HG_TILE)) L.hinge_tile = ctx->get(OPT_HG_TILE) == 256 ? 256 : 512;  // tuning knob
      // (the piece lists stand by on 256-sample tiles of their own, packed: k_hinge_gate)
      L.pieces = false, L.paired = false, L.pw = 4;
    }
  }
  // (rounds 3-5 kept one column tile — up to 16 output columns, BASELINE config 2 — on the 4-wave kernel: its 8-wave form was held to
  // 128 registers for two workgroups per CU and spilled inside the chunk loop: 0.296 ms against 0.261)
  // (round 6: the single-column-tile form too — its 8-wave kernel no longer spills (154 registers, one workgroup per CU): config 2
  // 0.2257 ms against 0.2283 / 0.2487 (the two buffer modes) over six fresh processes each, and never in the slow mode)
  L.wide = L.split && aligned_tile >= 512 && forced != 256 && (forced == 512 || nsamples / 512 >= 2 * ctx->num_cus);
  // Exact f32 asked for (MFMA = 1) and EVERY curve point on the 512-sample tile grid of a call of whole tiles: the line form
  // of the ramp on the f32 matrix pipe, no per-sample arithmetic in front of it (gain_f32g.h).  Finite gains only (a
  // constant segment's row is used as it is, but a ramp through a non-finite gain is libear's own arithmetic's business).
  L.f32grid = L.mfma && !L.split && !L.pieces && !L.hinge && ctx->use_mfma == 1 && grid512_strict && gain_scale > 0.0f &&
              nsamples % kF32GridTile == 0 && M >= 16;
  // the slot lists of the f32 MFMA kernel address objects with 16 bits
  if (L.mfma && !L.split && !L.pieces && !L.hinge && !L.f32grid && M > kMaxSlotObjects) L.mfma = false;
  L.spl = ctx->spl;
  L.nrt = ctx->nrt;
  L.ntiles = (nsamples + L.tile() - 1) / L.tile();
  if (L.split || L.pieces || L.hinge || L.f32grid) {
    // one workgroup = 4 adjacent 64-sample tiles x all objects of its grid-level
    // split; few tiles (block mode): split the objects across workgroups
    L.wsplit = 1;
    L.tpw = 1;
    int g = 1;
    const int nz = cp.mnz * cp.mgroups;
    while (g < max_gsplit && L.ntiles * nz * g < 2 * ctx->num_cus && M / (g * 2) >= 64) g *= 2;
    L.gsplit = g;
    return L;
  }
  if (strict) {
    L.wsplit = 1;
    L.gsplit = 1;
    return L;
  }
  // 8 waves per workgroup: what the column groups leave goes to object splits
  const int groups = L.mfma ? cp.mgroups : cp.ngroups;
  if (L.mfma) L.tpw = std::max(1, std::min(std::min(ctx->tiles_per_wg, L.ntiles), ctx->max_waves / groups));
  // The exact f32 kernel adds every term to its running total in its own rounding step (an MFMA of k = 4 is four fused
  // multiply-adds in a row): with many objects the waves of a workgroup split the slot list rather than taking adjacent
  // tiles, so that no wave's chain is longer than ~256 terms (1024 ramping objects, one wave: 8.5e-7 from a float64 render
  // and 1.06e-6 from the CPU path; eight waves: 4.6e-7 and 7.9e-7, for 3 % of that kernel's time)
  if (L.mfma && !ctx->tiles_per_wg_forced) L.tpw = std::max(1, std::min(L.tpw, (ctx->max_waves / groups) / std::max(1, (M + 127) / 128)));
  L.wsplit = std::max(1, std::min(std::max(1, ctx->max_waves / (groups * L.tpw)), std::max(1, M / 8)));
  // few tiles (block mode): split the objects across workgroups as well until
  // the grid covers the chip about twice over
  const int per_wg = std::max(1, M / L.wsplit);
  int g = 1;
  const int want = 2 * ctx->num_cus;
  while (g < max_gsplit && (L.ntiles / L.tpw) * (L.mfma ? cp.mnz : cp.nz) * g < want && per_wg / (g * 2) >= 8) g *= 2;
  L.gsplit = g;
  return L;
}

// Samples per bus column that hold [gsplit][pad4(nsamples)] for every plan plan_mix can make for calls
// of up to max_samples samples: it doubles gsplit only while gsplit * (ntiles / tpw) < 2 * num_cus, so
// gsplit * ntiles < (4 * num_cus + gsplit) * tpw, with tiles of at most 512 samples (split-operand
// kernels, tpw = 1), 256 (VALU) or 16 * nrt (f32 MFMA kernel, tpw adjacent tiles per workgroup).
inline size_t bus_samples_bound(const earhip_ctx *ctx, size_t max_samples, int max_gsplit) {
  const size_t pad = (max_samples + 3) & ~(size_t)3;
  const size_t tpw_max = (size_t)std::max(1, std::min(ctx->tiles_per_wg, ctx->max_waves));
  const size_t split_tiles = (size_t)4 * ctx->num_cus + max_gsplit;
  const size_t split = std::max(split_tiles * 512, split_tiles * tpw_max * 16 * (size_t)ctx->nrt) + 4 * (size_t)max_gsplit;
  return std::max(pad, std::min(pad * (size_t)max_gsplit, split));
}

size_t mix_lds_bytes(const ColumnPlan &cp, const MixLaunch &ml);
void reserve_call_words(earhip_ctx *ctx, int M, size_t max_samples);  // api_core.hip

// 16-byte units of the scratch buffer launch_gain_mix needs for a plan: descriptors (grid kernel, VALU kernel), + slot lists
// (f32 kernel), piece lists sized from the curves, hinge lists (+ the piece lists standing by for them)
inline size_t scratch_units(const CurveSet &cs, const MixLaunch &ml, int M) {
  const size_t nt = (size_t)ml.ntiles;
  // (the piece lists that stand by behind the hinge kernel: tiles of 64 pw samples, at most tile / (64 pw) times as many)
  // (+ M x tiles units behind the lists: the staging matrix of the two-kernel builders, 16 bytes per (object, tile) pair)
  if (ml.hinge) {
    const size_t pnt = nt * (size_t)std::max(1, ml.tile() / (64 * ml.pw));
    return std::max(hinge_units((size_t)M, nt) + (size_t)M * nt,
                    piece_units((size_t)M, pnt, (size_t)cs.piece_cap(64 * ml.pw, ml.paired)) + (size_t)M * pnt);
  }
  if (ml.pieces) return piece_units((size_t)M, nt, (size_t)cs.piece_cap(ml.tile(), ml.paired)) + (size_t)M * nt;
  if (ml.split || ml.f32grid || !ml.mfma) return (size_t)M * nt + 1;
  return desc_units((size_t)M, nt);
}

// Enqueue K0 + K1.  out: [gsplit][ncols][out_stride] (part_stride floats apart)
void launch_gain_mix(earhip_ctx *ctx, const CurveSet &cs, const MixLaunch &ml, bool strict,
                     int64_t t_call, int nsamples, const float *in_dev, size_t in_stride,
                     float *out_dev, size_t out_stride, size_t part_stride, SegDesc *desc,
                     hipEvent_t *ev /* optional [4]: prep begin/end, mix begin/end */);

}  // namespace earhip
