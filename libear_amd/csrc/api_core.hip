// api_core.hip — context, error reporting, the interpolation-policy entry
// points (A) and the whole-curve GainInterpolator (A') of include/earhip.h.
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <string>
#include <memory>
#include <mutex>

#include "common.h"
#include "curves.h"
#include "gain_kernels.h"
#include "gain_mfma.h"
#include "gain_h2.h"
#include "gain_h2_t1.h"
#include "gain_p2.h"
#include "gain_hg.h"
#include "gain_f32g.h"

namespace earhip {

static_assert(kSplitTile == 256, "MixLaunch::tile() and tiles_aligned(256, ...) assume the 4-wave tile");

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }

size_t mix_lds_bytes(const ColumnPlan &cp, const MixLaunch &ml) {
  if (ml.wsplit <= 1 || ml.split || ml.pieces || ml.hinge) return 0;
  if (ml.mfma) return (size_t)cp.mgroups * ml.tpw * cp.nct * 16 * ml.tile() * sizeof(float);
  return (size_t)cp.ngroups * cp.nout * ml.tile() * sizeof(float);
}

template <int NCT, int NRT>
static void launch_mfma_t(const GainMixParams &P, dim3 grid, dim3 block, size_t lds,
                          hipStream_t s) {
  hipLaunchKernelGGL((k_gain_mix_mfma<NCT, NRT>), grid, block, lds, s, P);
}

template <int NOUT, int SPL, bool STRICT>
static void launch_mix_t(const GainMixParams &P, dim3 grid, dim3 block, size_t lds,
                         hipStream_t s) {
  hipLaunchKernelGGL((k_gain_mix<NOUT, SPL, STRICT>), grid, block, lds, s, P);
}

void launch_gain_mix(earhip_ctx *ctx, const CurveSet &cs, const MixLaunch &ml, bool strict,
                     int64_t t_call, int nsamples, const float *in_dev, size_t in_stride,
                     float *out_dev, size_t out_stride, size_t part_stride, SegDesc *desc,
                     hipEvent_t *ev) {
  const ColumnPlan &cp = cs.plan();
  const PointStore ps = cs.device();
  const int M = cs.M();
  const bool slots = ml.mfma && !ml.split && !ml.pieces && !ml.hinge && !ml.f32grid;
  if (slots && M > kMaxSlotObjects) fail_internal("slot lists address objects with 16 bits");
  if (ev) EARHIP_HIP(hipEventRecord(ev[0], ctx->stream));
  // K0: segment descriptors; for the f32 MFMA kernel K0s then turns them into the
  // tiles' slot lists, which live behind the descriptors in the same buffer (desc_units())
  // short curves (static or nearly static gains, <= 8 points per object on average): K0s finds
  // the segments itself and the descriptor pass is skipped; with long curves the 16-lanes-per-
  // object search of k_seg_prep is the faster one (measured both ways)
  const bool fused_prep = slots && (size_t)ps.npoints <= (size_t)8 * M;
  // split-operand kernels: K0 starts with the level probe of the call's inputs (k_level_probe: 64 instants per object; rows
  // must allow 16-byte loads).  The two level words alternate between calls: this call's probe raises word `li` (zero since
  // the last such call cleared it), its K1 reads it and clears the other.
  LevelProbe probe;
  unsigned *level_cur = nullptr, *level_next = nullptr;
  unsigned *wide_cur = nullptr, *wide_next = nullptr;  // f16x2 kernel: "run this call in wide mode" (gain_h2.h)
  unsigned *gate = nullptr;                            // hinge kernel: "this call is the piece lists'" (k_hinge_gate)
  ctx->last_gate_idx = -1;
  ctx->last_wide_idx = -1;
  unsigned *const record = ctx->record_slot;  // (the renderer's slot for this call's mode word; consumed here)
  ctx->record_slot = nullptr;
  // hinge kernel: what a call whose levels spread beyond the packed-f16 kink products' span gets — the kernel's robust form
  // (default) or, option HG_ROBUST = 0, the piece lists standing by behind it (rounds 4-5)
  const bool hg_robust = ml.hinge && ctx->get(OPT_HG_ROBUST, 1) != 0;
  if ((ml.split || ml.pieces || ml.hinge) && ctx->x_scale_auto && in_stride % 4 == 0 && ((uintptr_t)in_dev & 15) == 0) {
    // (the words, the per-object levels and the per-tile words below are made where a renderer is created —
    // reserve_call_words —; a gain stage that comes here first, or a call larger than any renderer of the context announced,
    // grows them here behind a synchronisation: counted in ctx->lazy_allocs, which a renderer reports with its regrows)
    if (!ctx->level.p) ctx->level.alloc_zero(4, ctx->stream), ctx->lazy_allocs++;  // [0..1] level words, [2..3] wide-mode words
    level_cur = ctx->level.p + ctx->level_idx;
    level_next = ctx->level.p + (ctx->level_idx ^ 1);
    // a short call (less than two rounds of workgroups: block mode) runs the wide form only: its latency does not
    // depend on the form, and the second launch — the form that returns at once — is 4-5 us of it
    // (every call clears the word the NEXT call decides with, whether it decides itself or not)
    if ((size_t)ml.ntiles * ml.gsplit >= 2 * (size_t)ctx->num_cus) wide_cur = ctx->level.p + 2 + ctx->level_idx;
    wide_next = ctx->level.p + 2 + (ctx->level_idx ^ 1);
    gate = ctx->level.p + 2 + ctx->level_idx;  // (the same word: bit 0 the grid kernel's wide mode, bit 1 "not the hinge kernel")
    ctx->last_gate_idx = ml.hinge ? ctx->level_idx : -1;
    ctx->last_hinge_robust = hg_robust;
    ctx->last_wide_idx = wide_cur ? ctx->level_idx : -1;
    ctx->level_idx ^= 1;
    // per-object levels (k_level_probe, gain_kernels.h): grown with the largest M this context has seen — contexts are
    // shared by gain stages of different sizes
    if (ctx->obj_level_cap < M) {
      EARHIP_HIP(hipStreamSynchronize(ctx->stream));
      ctx->obj_level_cap = M + M / 2 + 64;
      ctx->obj_level.alloc_zero(2 * (size_t)ctx->obj_level_cap, ctx->stream);
      ctx->lazy_allocs++;
    }
    probe.obj_level = ctx->obj_level.p;
    probe.level = level_cur;
    const int probe_runs = [&] {  // tuning knob: runs of consecutive samples per object (1 .. 64)
      const int v = ctx->get(OPT_PROBE_RUNS);
      return v >= 1 && v <= 64 && (v & (v - 1)) == 0 ? v : 16;
    }();
    hipLaunchKernelGGL(k_level_probe, dim3((M + kProbeObjects - 1) / kProbeObjects), dim3(64 * kProbeObjects), 0, ctx->stream, in_dev,
                       in_stride, nsamples, M, level_cur, ctx->obj_level.p, ctx->obj_level_cap, probe_runs);
  }
  // (piece lists: K0 also counts every object's ramps per tile, into the list builder's count words)
  PieceLists pl;
  pl.pieces = reinterpret_cast<Piece *>(desc);  // (the list builders need no descriptors)
  pl.M = M;
  pl.paired = ml.paired ? 1 : 0;
  // (the piece lists that stand by behind the hinge kernel have their own tiles: 64 pw samples, whatever the hinge kernel's)
  const int ptile = ml.hinge ? 64 * ml.pw : ml.tile();
  const int pnt = ml.hinge ? (nsamples + ptile - 1) / ptile : ml.ntiles;
  pl.slots = cs.piece_cap(ptile, ml.paired);
  pl.count = reinterpret_cast<int *>(pl.pieces + (size_t)pl.cap() * pnt);
  pl.ovf = pl.count + (size_t)8 * pnt;
  // f16x2 gain kernel: a word per tile, "some object needs the exact path here" (see gain_h2.h)
  unsigned *slow_cur = nullptr, *slow_next = nullptr;
  if (ml.split) {
    if (ctx->tile_slow_cap < (size_t)ml.ntiles) {
      EARHIP_HIP(hipStreamSynchronize(ctx->stream));
      ctx->tile_slow_cap = std::max<size_t>(2 * (size_t)ml.ntiles, 1024);
      ctx->tile_slow.alloc_zero(2 * ctx->tile_slow_cap, ctx->stream);
      ctx->lazy_allocs++;
    }
    slow_cur = ctx->tile_slow.p + (size_t)ctx->tile_slow_idx * ctx->tile_slow_cap;
    slow_next = ctx->tile_slow.p + (size_t)(ctx->tile_slow_idx ^ 1) * ctx->tile_slow_cap;
    ctx->tile_slow_idx ^= 1;
  }
  // hinge kernel: its own list builder (k_hinge_build, gain_hg.h) behind the same probe launch
  HingeLists hl = hinge_lists(desc, M, ml.ntiles);
  if (ml.hinge) {
    if (M > kMaxHingeCached || (ml.tile() != 256 && ml.tile() != 512)) fail_internal("hinge lists: object count or tile out of range");
    // (the kernel addresses input rows and gain rows with 32-bit byte offsets; plan_mix only picks it within these limits)
    if (!ps.kink_row0) fail_internal("hinge kernel: the curve set has no kink rows");
    if (!hinge_addressable(M, in_stride, nsamples, (size_t)ps.kink_row0 + ps.rows, ps.row)) fail_internal("hinge kernel: buffers beyond its 32-bit offsets");
    unsigned *obj_lv = probe.obj_level;
    // with a probe: does the hinge kernel's span of levels cover this call?  (decided on the device: k_hinge_gate)
    if (gate) hipLaunchKernelGGL(k_hinge_gate, dim3((M + 255) / 256), dim3(256), 0, ctx->stream, obj_lv, ctx->obj_level_cap, M, level_cur, gate, wide_cur != nullptr, hg_robust);
    int tpw = 1;
    const size_t cached_max = std::min((size_t)kMaxHingeCached, ctx->hinge_build_lds / sizeof(HingeCached));  // (pairs a workgroup may keep)
    if ((size_t)M > cached_max) fail_internal("hinge lists: more objects than the builder's LDS holds on this device");
    while (tpw < 8 && ml.ntiles / (2 * tpw) >= ctx->num_cus && (size_t)M * (2 * tpw) <= cached_max) tpw *= 2;
    if (ctx->has(OPT_HBUILD_TPW)) {  // tuning knob
      const int v = ctx->get(OPT_HBUILD_TPW);
      if ((v == 1 || v == 2 || v == 4 || v == 8) && (size_t)M * v <= cached_max) tpw = v;
    }
    // two kernels (option BUILD_2K = 1; NOT the default here): k_hinge_classify object-major into the staging matrix behind the
    // lists, then the builder from its counting loop on, without the LDS its first pass kept its results in.  Measured: classify
    // 21 us + the builder's second half 42 us against 58 us in one kernel — that half (ranking by ballots over 32 classes between
    // three barriers per batch, two scattered 16-byte stores per pair) is what the builder is made of, and it does not get faster
    // at two workgroups per CU.  The lists are the same bit for bit (tests run both).
    const HingeCached *stage = nullptr;
    if (ctx->get(OPT_BUILD_2K, 0) != 0) {
      HingeCached *st = reinterpret_cast<HingeCached *>(desc) + hinge_units((size_t)M, (size_t)ml.ntiles);
      stage = st;
      const dim3 cgrid((ml.ntiles + 63) / 64, (M + kClassifyObjects - 1) / kClassifyObjects);
      if (ml.tile() == 256)
        hipLaunchKernelGGL((k_hinge_classify<4>), cgrid, dim3(64 * kClassifyObjects), 0, ctx->stream, ps, M, ml.ntiles, t_call, t_call + nsamples, st,
                           obj_lv, level_cur, hg_robust ? nullptr : gate, hg_robust ? gate : nullptr);
      else
        hipLaunchKernelGGL((k_hinge_classify<8>), cgrid, dim3(64 * kClassifyObjects), 0, ctx->stream, ps, M, ml.ntiles, t_call, t_call + nsamples, st,
                           obj_lv, level_cur, hg_robust ? nullptr : gate, hg_robust ? gate : nullptr);
      if (!ctx->has(OPT_HBUILD_TPW)) {
        tpw = 1;
        while (tpw < 8 && ml.ntiles / (2 * tpw) >= 2 * ctx->num_cus) tpw *= 2;
      }
    }
    const dim3 bgrid((ml.ntiles + tpw - 1) / tpw);
    const size_t lds = stage ? 0 : sizeof(HingeCached) * (size_t)M * tpw;
    // (more than 64 KB of dynamic LDS has to be asked for, per device and instantiation: earhip_ctx_create does, hinge_build_allow_lds)
#define EARHIP_HBUILD_ONE(T_, NW_)                                                                                    \
  hipLaunchKernelGGL((k_hinge_build<T_, NW_>), bgrid, dim3(kHingeBuildThreads), lds, ctx->stream, ps, M, ml.ntiles, t_call, \
                     t_call + nsamples, hl, obj_lv, level_cur, hg_robust ? nullptr : gate, hg_robust ? gate : nullptr, stage);
#define EARHIP_HBUILD_CASE(T_)                                                                                        \
  if (tpw == T_) {                                                                                                    \
    if (ml.tile() == 256) EARHIP_HBUILD_ONE(T_, 4)                                                                    \
    else EARHIP_HBUILD_ONE(T_, 8)                                                                                     \
  }
    EARHIP_HBUILD_CASE(1) EARHIP_HBUILD_CASE(2) EARHIP_HBUILD_CASE(4) EARHIP_HBUILD_CASE(8)
#undef EARHIP_HBUILD_CASE
#undef EARHIP_HBUILD_ONE
  }
  // piece-list kernel: ONE pass builds the lists (k_piece_build, gain_p2.h) behind a small probe launch
  const bool one_pass = ml.pieces || ml.hinge;
  if (ml.pieces || (ml.hinge && gate && !hg_robust)) {  // (behind the hinge kernel's builder: the lists of the call it may not take)
    if (M > kMaxPieceObjects || ptile > kPieceMaxTile) fail_internal("piece lists: object index or tile out of range");
    unsigned *obj_lv = probe.obj_level;
    // tiles per workgroup: as many as leave one workgroup per CU (eight tiles = eight lanes per object reading
    // neighbouring points beat four tiles and two workgroups per CU: 0.084 vs 0.096 ms on the ADM scene)
    int tpw = 1;
    while (tpw < 8 && pnt / (2 * tpw) >= ctx->num_cus) tpw *= 2;
    if (ctx->has(OPT_BUILD_TPW)) {  // tuning knob
      const int v = ctx->get(OPT_BUILD_TPW);
      if (v == 1 || v == 2 || v == 4 || v == 8) tpw = v;
    }
    const dim3 bgrid((pnt + tpw - 1) / tpw);
    // two kernels (default; option BUILD_2K = 0: the one-pass builder): classify object-major into the staging matrix behind the
    // lists (scratch_units holds room for it), then place tile-major
    PairRec *stage = nullptr;
    // (default: where the lists are PAIRED — curves that hold most of the time: at most one ramp per object and tile nearly
    // everywhere, which the staged pair carries; always-ramping curves on packed lists have several per pair, the placing kernel
    // would walk most pairs itself: 0.097 against 0.076 ms, measured)
    if (ctx->has(OPT_BUILD_2K) ? ctx->get(OPT_BUILD_2K) != 0 : ml.paired) {
      stage = reinterpret_cast<PairRec *>(desc) + piece_units((size_t)M, (size_t)pnt, (size_t)pl.slots);
      hipLaunchKernelGGL(k_piece_classify, dim3((pnt + 63) / 64, (M + kClassifyObjects - 1) / kClassifyObjects), dim3(64 * kClassifyObjects), 0,
                         ctx->stream, ps, M, pnt, ptile, t_call, t_call + nsamples, pl.paired, stage, obj_lv, level_cur,
                         ml.hinge ? gate : nullptr);
    }
#define EARHIP_BUILD_CASE(T_)                                                                                          \
  if (tpw == T_)                                                                                                       \
    hipLaunchKernelGGL(k_piece_build<T_>, bgrid, dim3(kBuildThreads), 0, ctx->stream, ps, M, pnt, ptile, t_call, \
                       t_call + nsamples, pl, obj_lv, level_cur, ml.hinge ? gate : nullptr, ctx->obj_level_cap, wide_cur, stage);
    EARHIP_BUILD_CASE(1) EARHIP_BUILD_CASE(2) EARHIP_BUILD_CASE(4) EARHIP_BUILD_CASE(8)
#undef EARHIP_BUILD_CASE
  }
  if (!fused_prep && !one_pass) {
    const dim3 ogrid((M + 15) / 16);
    QuietMark qm;  // (what the probe found goes into the descriptors as they are made)
    if (probe.obj_level) {
      qm.obj_level = probe.obj_level, qm.cap = ctx->obj_level_cap, qm.level = level_cur;
      qm.wide = ml.split ? wide_cur : nullptr;
    }
    if (ml.ntiles >= 2048)
      hipLaunchKernelGGL(k_seg_prep<4>, dim3((ml.ntiles + 63) / 64, ogrid.x), dim3(256), 0, ctx->stream, ps, M,
                         ml.ntiles, ml.tile(), t_call, t_call + nsamples, desc, qm, slow_cur);
    else
      hipLaunchKernelGGL(k_seg_prep<2>, dim3((ml.ntiles + 31) / 32, ogrid.x), dim3(256), 0, ctx->stream, ps, M,
                         ml.ntiles, ml.tile(), t_call, t_call + nsamples, desc, qm, slow_cur);
  }
  SlotLists sl;
  sl.slots = reinterpret_cast<Slot *>(desc + (size_t)M * ml.ntiles);
  sl.count = reinterpret_cast<int *>(sl.slots + (size_t)kTileSlots * M * ml.ntiles);
  sl.ovf = sl.count + (size_t)4 * ml.ntiles;
  sl.M = M;
  if (slots)
    hipLaunchKernelGGL(k_slot_list, dim3(ml.ntiles), dim3(256), 0, ctx->stream, ps, M, ml.tile(),
                       t_call, t_call + nsamples, fused_prep ? nullptr : desc, sl);
  if (ev) EARHIP_HIP(hipEventRecord(ev[1], ctx->stream));
  GainMixParams P;
  P.mode_word = gate;
  P.record = gate ? record : nullptr;
  P.sl = sl;
  P.in = in_dev;
  P.in_stride = in_stride;
  P.out = out_dev;
  P.out_stride = out_stride;
  P.part_stride = part_stride;
  P.desc = desc;
  P.ps = ps;
  P.t_call = t_call;
  P.nsamples = nsamples;
  P.ntiles = ml.ntiles;
  P.M = M;
  P.ncols = cs.ncols();
  P.ngroups = ml.mfma ? cp.mgroups : cp.ngroups;
  P.wsplit = ml.wsplit;
  P.tiles_per_wg = ml.mfma ? ml.tpw : 1;
  P.vec_ok = (in_stride % 4 == 0 && out_stride % 4 == 0 && part_stride % 4 == 0 &&
              ((uintptr_t)in_dev & 15) == 0 && ((uintptr_t)out_dev & 15) == 0)
                 ? 1
                 : 0;
  const dim3 grid(ml.mfma ? (ml.ntiles + ml.tpw - 1) / ml.tpw : ml.ntiles, ml.gsplit,
                  ml.mfma ? cp.mnz : cp.nz);
  const dim3 block(64 * P.ngroups * P.tiles_per_wg * ml.wsplit);
  const size_t lds = mix_lds_bytes(cp, ml);
  if (ev) EARHIP_HIP(hipEventRecord(ev[2], ctx->stream));
  bool launched = false;
  if (ml.hinge) {
    const dim3 bgrid(ml.ntiles, ml.gsplit, cp.mnz * cp.mgroups);
    const float xs = std::ldexp(1.0f, ctx->x_scale_log2);
    const float *gs = cs.column_scales();
#define EARHIP_HG_LAUNCH(NCT_, NW_)                                                                                  \
  hipLaunchKernelGGL((k_gain_mix_hg<NCT_, NW_>), bgrid, dim3(64 * NW_), 0, ctx->stream, P, hl, xs, gs, level_cur,     \
                     level_next, wide_cur, wide_next, hg_robust ? nullptr : gate, hg_robust ? gate : nullptr);
#define EARHIP_HG_CASE(NCT_)                                                                                          \
  if (cp.nct == NCT_) {                                                                                               \
    if (ml.tile() == 256) EARHIP_HG_LAUNCH(NCT_, 4)                                                                   \
    else EARHIP_HG_LAUNCH(NCT_, 8)                                                                                    \
  }
    EARHIP_HG_CASE(1) EARHIP_HG_CASE(2) EARHIP_HG_CASE(3)
#undef EARHIP_HG_CASE
#undef EARHIP_HG_LAUNCH
    launched = true;
  }
  if (ml.pieces || (ml.hinge && gate && !hg_robust)) {
    // as many workgroups as are resident at once (a multiple of 8: a workgroup's tiles stay on its XCD), each carrying its
    // pipeline from one tile's list into the next (gain_p2.h); option P2_WGS: that number (0: a workgroup per tile)
    int wgs = std::max(8, (ctx->num_cus * (ml.pw == 4 ? 2 : 1) / std::max(1, ml.gsplit * cp.mnz * cp.mgroups)) & ~7);
    if (ctx->has(OPT_P2_WGS)) wgs = ctx->get(OPT_P2_WGS) > 0 ? ctx->get(OPT_P2_WGS) : pnt;
    if (!p2_persistent(cp.nct, ml.pw, ml.paired)) wgs = pnt;  // (a tile per workgroup)
    const dim3 bgrid(std::min(pnt, wgs), ml.gsplit, cp.mnz * cp.mgroups);
    const float xs = std::ldexp(1.0f, ctx->x_scale_log2);
    const float *gs = cs.column_scales();
    // (both forms of the kernel body in one kernel: the device-side mode word picks; without a probe the wide form)
#define EARHIP_P2_LAUNCH(NCT_, NW_, PR_)                                                                             \
  hipLaunchKernelGGL((k_gain_mix_p2<NCT_, NW_, PR_>), bgrid, dim3(64 * NW_), 0, ctx->stream, P, pl, xs, gs,         \
                     level_cur, level_next, wide_cur, wide_next, ml.hinge ? gate : nullptr, pnt);
#define EARHIP_P2_CASE(NCT_, PR_)                                                                                   \
  if (cp.nct == NCT_ && ml.paired == PR_) {                                                                         \
    if (ml.pw == 4) EARHIP_P2_LAUNCH(NCT_, 4, PR_)                                                                  \
    else EARHIP_P2_LAUNCH(NCT_, 8, PR_)                                                                             \
  }
    EARHIP_P2_CASE(1, false) EARHIP_P2_CASE(2, false) EARHIP_P2_CASE(3, false)
    EARHIP_P2_CASE(1, true) EARHIP_P2_CASE(2, true) EARHIP_P2_CASE(3, true)
#undef EARHIP_P2_CASE
#undef EARHIP_P2_LAUNCH
    launched = true;
  }
  if (ml.f32grid) {  // exact f32 on the tile grid (gain_f32g.h): a workgroup per 512-sample tile, descriptors from k_seg_prep
    if (nsamples % kF32GridTile != 0) fail_internal("f32 grid kernel: the call is not whole tiles");
    const dim3 bgrid(ml.ntiles, ml.gsplit, cp.mnz * cp.mgroups);
#define EARHIP_F32G_CASE(NCT_)                                                                       \
  if (cp.nct == NCT_) hipLaunchKernelGGL((k_gain_mix_f32g<NCT_>), bgrid, dim3(512), 0, ctx->stream, P, ps.zero_row);
    EARHIP_F32G_CASE(1) EARHIP_F32G_CASE(2) EARHIP_F32G_CASE(3)
#undef EARHIP_F32G_CASE
    launched = true;
  }
  if (ml.split) {
    // as many workgroups as are resident at once (a multiple of 8: a workgroup's tiles stay on its XCD), each working
    // through its share of the tiles in one software pipeline (gain_h2.h)
    const int per_cu = ml.tile() == 512 ? 1 : (cp.nct == 1 ? 3 : 2);  // (the 8-wave forms: 154-243 registers, one workgroup per CU)
    int wgs = std::max(8, (ctx->num_cus * per_cu / std::max(1, ml.gsplit * cp.mnz * cp.mgroups)) & ~7);
    if (ctx->has(OPT_H2_WGS)) wgs = std::max(1, ctx->get(OPT_H2_WGS));  // tuning knob
    wgs = std::max(wgs, ((ml.ntiles + 63) / 64 + 7) & ~7);  // (at most 64 tiles per workgroup: its redo mask)
    // a workgroup's tiles as one contiguous run (option H2_RUNS: 1 on, 0 off; default: the single-column-tile forms — config 2)
    P.tile_runs = ctx->has(OPT_H2_RUNS) ? (ctx->get(OPT_H2_RUNS) != 0) : 0;
    if (!h2_persistent(cp.nct, ml.tile() == 512 ? 8 : 4)) wgs = ml.ntiles;  // (a tile per workgroup)
    const dim3 bgrid(std::min(ml.ntiles, wgs), ml.gsplit, cp.mnz * cp.mgroups);
    const float xs = std::ldexp(1.0f, ctx->x_scale_log2);
    const float *gs = cs.column_scales();
    // two forms of the kernel: the 8-wave kernel carries both and branches on the device-side mode word (round 6); the 4-wave
    // kernels are launched as a pair, back to back: the form the word names runs, the other returns at once
    // (gain_h2.h; the piece-list and hinge kernels carry both forms in one kernel — for this one the merged kernel measured
    // slower: config 2 0.224 -> 0.259 ms, its 4-wave form loses a wave per SIMD to the larger of the two register counts);
    // without a probe only the wide form
    const bool two_launches = ctx->get(OPT_H2_PAIR, 0) != 0;  // (tuning knob: the 8-wave kernel as a pair of launches, as until round 6)
#define EARHIP_H2_LAUNCH(NCT_, NW_, WIDE_)                                                                          \
  hipLaunchKernelGGL((k_gain_mix_h2<NCT_, NW_, WIDE_>), bgrid, dim3(64 * NW_), 0, ctx->stream, P, ps.zero_row, xs,  \
                     gs, level_cur, level_next, slow_cur, slow_next, wide_cur, wide_next);
#define EARHIP_H2T1_LAUNCH(NCT_, NW_, WIDE_)                                                                        \
  hipLaunchKernelGGL((k_gain_mix_h2_t1<NCT_, NW_, WIDE_>), bgrid, dim3(64 * NW_), 0, ctx->stream, P, ps.zero_row, xs, \
                     gs, level_cur, level_next, slow_cur, slow_next, wide_cur, wide_next);
#define EARHIP_H2_CASE(NCT_)                                                                                        \
  if (cp.nct == NCT_) {                                                                                             \
    if (ml.tile() == 512) {                                                                                         \
      if (wide_cur && !two_launches) EARHIP_H2_LAUNCH(NCT_, 8, 2)                                                   \
      else {                                                                                                        \
        if (wide_cur) EARHIP_H2_LAUNCH(NCT_, 8, 0)                                                                  \
        EARHIP_H2_LAUNCH(NCT_, 8, 1)                                                                                \
      }                                                                                                             \
    } else if (NCT_ == 1) {                                                                                         \
      if (wide_cur) EARHIP_H2_LAUNCH(1, 4, 0)                                                                       \
      EARHIP_H2_LAUNCH(1, 4, 1)                                                                                     \
    } else {                                                                                                        \
      if (wide_cur) EARHIP_H2T1_LAUNCH(NCT_, 4, false)                                                              \
      EARHIP_H2T1_LAUNCH(NCT_, 4, true)                                                                             \
    }                                                                                                               \
  }
    EARHIP_H2_CASE(1) EARHIP_H2_CASE(2) EARHIP_H2_CASE(3)
#undef EARHIP_H2_CASE
#undef EARHIP_H2_LAUNCH
#undef EARHIP_H2T1_LAUNCH
    launched = true;
  }
#define EARHIP_MIX_CASE(NOUT_, SPL_, STRICT_)                                   \
  if (cp.nout == NOUT_ && ml.spl == SPL_ && strict == STRICT_) {                 \
    launch_mix_t<NOUT_, SPL_, STRICT_>(P, grid, block, lds, ctx->stream);        \
    launched = true;                                                             \
  }
#define EARHIP_MFMA_CASE(NCT_, NRT_)                                            \
  if (slots && cp.nct == NCT_ && ml.nrt == NRT_) {                                          \
    launch_mfma_t<NCT_, NRT_>(P, grid, block, lds, ctx->stream);                 \
    launched = true;                                                             \
  }
  EARHIP_MFMA_CASE(1, 8) EARHIP_MFMA_CASE(2, 8) EARHIP_MFMA_CASE(3, 8)
  EARHIP_MFMA_CASE(1, 4) EARHIP_MFMA_CASE(2, 4) EARHIP_MFMA_CASE(3, 4)
#undef EARHIP_MFMA_CASE
  if (!ml.mfma && !launched) {
  EARHIP_MIX_CASE(8, 4, false) EARHIP_MIX_CASE(8, 4, true)
  EARHIP_MIX_CASE(16, 4, false) EARHIP_MIX_CASE(16, 4, true)
  EARHIP_MIX_CASE(24, 4, false) EARHIP_MIX_CASE(24, 4, true)
  EARHIP_MIX_CASE(8, 2, false) EARHIP_MIX_CASE(8, 2, true)
  EARHIP_MIX_CASE(16, 2, false) EARHIP_MIX_CASE(16, 2, true)
  EARHIP_MIX_CASE(24, 2, false) EARHIP_MIX_CASE(24, 2, true)
  }
#undef EARHIP_MIX_CASE
  if (!launched) fail_internal("no gain_mix instantiation for this column plan");
  if (ev) EARHIP_HIP(hipEventRecord(ev[3], ctx->stream));
  EARHIP_HIP(hipGetLastError());
}

// What the split-operand kernels of a call of up to `max_samples` samples on M objects keep PER CONTEXT — the level and mode
// words, the per-object levels of the probe, the per-tile "exact path" words of the grid kernel (tiles of 256 samples at the
// smallest) — made where a renderer is created, so that no process call allocates or synchronises for them.
void reserve_call_words(earhip_ctx *ctx, int M, size_t max_samples) {
  if (!ctx->level.p) ctx->level.alloc_zero(4, ctx->stream);
  const size_t ntiles = (max_samples + 255) / 256;
  const bool grow_obj = ctx->obj_level_cap < M, grow_tiles = ctx->tile_slow_cap < ntiles;
  if (grow_obj || grow_tiles) EARHIP_HIP(hipStreamSynchronize(ctx->stream));  // (calls of other stages may be using the old ones)
  if (grow_obj) {
    ctx->obj_level_cap = M + M / 2 + 64;
    ctx->obj_level.alloc_zero(2 * (size_t)ctx->obj_level_cap, ctx->stream);
  }
  if (grow_tiles) {
    ctx->tile_slow_cap = std::max<size_t>(ntiles + ntiles / 4, 1024);
    ctx->tile_slow.alloc_zero(2 * ctx->tile_slow_cap, ctx->stream);
    ctx->tile_slow_idx = 0;
  }
}

// k_hinge_build keeps up to 128 KB of dynamic LDS per workgroup: the limit is an attribute of the function ON A DEVICE, so
// every context raises it for its own device when it is created (a process-wide "done" flag left the other GPUs of a
// multi-GPU process at the default 64 KB: launches with more failed there).
// Returns the dynamic LDS a launch of the builder may ask for on this device: `want` (128 KB) where the device has it, else
// what it allows (a device or runtime that refuses the attribute: the 64 KB every kernel has) — the launch picks its tiles
// per workgroup within that (a limit, never a reason for context creation to fail).
size_t hinge_build_allow_lds(const hipDeviceProp_t &prop) {
  const size_t want = 128 * 1024, dflt = 64 * 1024;
  const size_t device_max = std::max((size_t)prop.sharedMemPerBlock, (size_t)prop.maxSharedMemoryPerMultiProcessor);
  size_t ask = std::min(want, device_max > 4096 ? device_max - 4096 : dflt);  // (the kernel's static arrays: ~3 KB)
  if (ask <= dflt) return dflt;
  bool ok = true;
#define EARHIP_HBUILD_ATTR(T_, NW_)                                                                       \
  ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hinge_build<T_, NW_>),                 \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)ask) == hipSuccess;
  EARHIP_HBUILD_ATTR(1, 4) EARHIP_HBUILD_ATTR(2, 4) EARHIP_HBUILD_ATTR(4, 4) EARHIP_HBUILD_ATTR(8, 4)
  EARHIP_HBUILD_ATTR(1, 8) EARHIP_HBUILD_ATTR(2, 8) EARHIP_HBUILD_ATTR(4, 8) EARHIP_HBUILD_ATTR(8, 8)
#undef EARHIP_HBUILD_ATTR
  if (!ok) (void)hipGetLastError();  // (refused: the default limit stands)
  return ok ? ask : dflt;
}

// Device-resident M -> N gain stage with host staging; shared by the policy
// entry points and by earhip_gain_interp.
struct GainStage {
  earhip_ctx *ctx;
  int n_in, n_out;
  CurveSet curves;
  DevBuf<SegDesc> desc;
  DevBuf<float> parts;  // grid-level partial slabs
  std::vector<float> rows;  // scratch of the policy entry points (two gain rows)
  bool force_ramp;
  GainStage(earhip_ctx *c, int ni, int no, bool ramp = false)
      : ctx(c), n_in(ni), n_out(no), curves(ni, no, 1, ramp), force_ramp(ramp) {}

  // in_dev [n_in][in_stride] -> out_dev [n_out][out_stride]
  void run_device(int64_t t_call, int nsamples, const float *in_dev, size_t in_stride,
                  float *out_dev, size_t out_stride) {
    curves.commit(ctx);
    // 1 -> N policies have no accumulation: always use libear's exact arithmetic
    const bool strict = ctx->strict || n_in == 1;
    MixLaunch ml = plan_mix(ctx, curves.plan(), n_in, nsamples, strict, 32,
                            curves.aligned_tile(t_call), curves.ramp_share(), curves.gain_scale(), curves.point_density_all(),
                            curves.pair_waste(256), curves.pair_waste(512), curves.hinge_exact_share(in_stride, (size_t)nsamples), false,
                            curves.deltas_per_pair(256));
    if (ml.hinge) curves.ensure_kinks(ctx);
    desc.reserve(scratch_units(curves, ml, n_in));
    if (ml.gsplit == 1) {
      launch_gain_mix(ctx, curves, ml, strict, t_call, nsamples, in_dev, in_stride, out_dev,
                      out_stride, 0, desc.p, nullptr);
    } else {
      const size_t row_stride = ((size_t)nsamples + 3) & ~(size_t)3;
      const size_t part_stride = row_stride * n_out;
      parts.reserve(part_stride * ml.gsplit);
      launch_gain_mix(ctx, curves, ml, strict, t_call, nsamples, in_dev, in_stride, parts.p,
                      row_stride, part_stride, desc.p, nullptr);
      hipLaunchKernelGGL(k_sum_parts, dim3((nsamples + 255) / 256, n_out), dim3(256), 0,
                         ctx->stream, parts.p, part_stride, ml.gsplit, row_stride, n_out,
                         nsamples, out_dev, out_stride);
      EARHIP_HIP(hipGetLastError());
    }
  }

  // host planar pointers, samples [r0, r1) of each channel
  void run_host(int64_t t_call, const float *const *in, float *const *out, int64_t r0,
                int64_t r1) {
    const int n = (int)(r1 - r0);
    if (n <= 0) return;
    const size_t stride = ((size_t)n + 3) & ~(size_t)3;
    ctx->pin_in.reserve(stride * n_in);
    ctx->pin_out.reserve(stride * n_out);
    ctx->dev_in.reserve(stride * n_in);
    ctx->dev_out.reserve(stride * n_out);
    for (int c = 0; c < n_in; c++)
      std::memcpy(ctx->pin_in.p + c * stride, in[c] + r0, sizeof(float) * n);
    EARHIP_HIP(hipMemcpyAsync(ctx->dev_in.p, ctx->pin_in.p, sizeof(float) * stride * n_in,
                              hipMemcpyHostToDevice, ctx->stream));
    run_device(t_call, n, ctx->dev_in.p, stride, ctx->dev_out.p, stride);
    EARHIP_HIP(hipMemcpyAsync(ctx->pin_out.p, ctx->dev_out.p, sizeof(float) * stride * n_out,
                              hipMemcpyDeviceToHost, ctx->stream));
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < n_out; c++)
      std::memcpy(out[c] + r0, ctx->pin_out.p + c * stride, sizeof(float) * n);
  }
};

}  // namespace earhip

using namespace earhip;

// keys of earhip_ctx_set_option (and, prefixed with EARHIP_, the environment variables read at earhip_ctx_create)
static const char *const kOptNames[OPT_COUNT] = {
    "SPL", "MFMA", "XSCALE", "WAVES", "TPW", "NRT", "H2_TILE", "H2_WGS", "H2_RUNS", "P2_TILE", "P2_PAIRS", "P2_WGS", "HINGE", "HG_TILE", "HBUILD_TPW",
    "BUILD_TPW", "K2_WG", "K2_OWN_BLOCK", "RUN", "GSPLIT", "PROBE_RUNS", "BLOCK_GROUPS", "DEBUG_TIMING", "TAILCUT", "HOST_CHUNK_MB", "HOST_THREADS", "HG_ROBUST", "BUILD_2K", "HOST_BIND", "HOST_NT", "HOST_FIRST", "H2_PAIR"};

// an option's value: a decimal integer (optional sign, surrounding blanks), nothing else — "abc" or "1x" used to read as 0 / 1
static int parse_option_value(const std::string &key, const char *text) {
  char *end = nullptr;
  errno = 0;
  const long v = std::strtol(text, &end, 10);
  while (end && *end && std::isspace((unsigned char)*end)) end++;
  if (end == text || (end && *end) || errno == ERANGE || v < -(1L << 30) || v > (1L << 30))
    fail_invalid("option '" + key + "': '" + std::string(text) + "' is not an integer");
  return (int)v;
}

void earhip_ctx::apply_options() {
  spl = 4, use_mfma = 3, x_scale_log2 = 14, x_scale_auto = true, max_waves = 4, tiles_per_wg = 4, tiles_per_wg_forced = false, nrt = 8;
  if (has(OPT_SPL) && (get(OPT_SPL) == 2 || get(OPT_SPL) == 4)) spl = get(OPT_SPL);  // samples per lane in gain_mix
  if (has(OPT_MFMA)) use_mfma = get(OPT_MFMA);
  if (has(OPT_XSCALE) && get(OPT_XSCALE) >= -64 && get(OPT_XSCALE) <= 64)  // f16x2 kernel: log2 of the input prescale
    x_scale_log2 = get(OPT_XSCALE), x_scale_auto = false;
  if (has(OPT_WAVES) && get(OPT_WAVES) >= 1 && get(OPT_WAVES) <= 8) max_waves = get(OPT_WAVES);
  if (has(OPT_TPW) && get(OPT_TPW) >= 1 && get(OPT_TPW) <= 8) tiles_per_wg = get(OPT_TPW), tiles_per_wg_forced = true;
  if (has(OPT_NRT) && (get(OPT_NRT) == 4 || get(OPT_NRT) == 8)) nrt = get(OPT_NRT);
}

struct earhip_gain_interp {
  std::unique_ptr<GainStage> stage;
};

// policy scratch: one GainStage per (n_in, n_out) shape seen on this context
struct PolicyCache {
  std::vector<std::unique_ptr<GainStage>> stages;
  GainStage *get(earhip_ctx *ctx, int n_in, int n_out, bool ramp) {
    for (auto &s : stages)
      if (s->n_in == n_in && s->n_out == n_out && s->force_ramp == ramp) return s.get();
    stages.emplace_back(new GainStage(ctx, n_in, n_out, ramp));
    return stages.back().get();
  }
};
static std::vector<std::pair<earhip_ctx *, std::unique_ptr<PolicyCache>>> g_policy;
static std::mutex g_policy_mutex;  // contexts are single-owner, the table of them is shared by all threads
static PolicyCache *policy_cache(earhip_ctx *ctx) {
  std::lock_guard<std::mutex> lock(g_policy_mutex);
  for (auto &p : g_policy)
    if (p.first == ctx) return p.second.get();
  g_policy.emplace_back(ctx, std::unique_ptr<PolicyCache>(new PolicyCache));
  return g_policy.back().second.get();
}

static __global__ void k_clock_probe(unsigned long long *out) {
  out[0] = clock64();       // shader clock (s_memtime)
  out[1] = wall_clock64();  // constant-rate counter
}

// Read-bandwidth probes (earhip_debug_read_bandwidth): what this chip delivers to a kernel that does
// nothing but read, (a) one linear stream, (b) the gain stage's pattern — every workgroup reads a 1 KB
// piece of each of `rows` rows that lie `stride` floats apart, 32 rows in flight per wave.
static __global__ void __launch_bounds__(256) k_read_linear(const f32x4 *in, float *out, size_t n4) {
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t step = (size_t)gridDim.x * 256;
  for (; i + 3 * step < n4; i += 4 * step) {
    const f32x4 a = __builtin_nontemporal_load(in + i), b = __builtin_nontemporal_load(in + i + step);
    const f32x4 c = __builtin_nontemporal_load(in + i + 2 * step), d = __builtin_nontemporal_load(in + i + 3 * step);
    acc += (a + b) + (c + d);
  }
  for (; i < n4; i += step) acc += __builtin_nontemporal_load(in + i);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.0f;  // (keeps the loads alive)
}
static __global__ void __launch_bounds__(256) k_read_rows(const float *in, float *out, int rows, size_t stride) {
  constexpr int DEPTH = 4;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 15, kg = lane >> 4;
  const float *base = in + (size_t)blockIdx.x * 256 + w * 64 + li * 4;
  auto row = [&](int r) { return base + (size_t)min(r, rows - 1) * stride; };
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 ring[DEPTH][8];
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
#pragma unroll
    for (int q = 0; q < 8; q++)
      ring[d][q] = __builtin_nontemporal_load((const f32x4 *)row(d * 32 + kg * 8 + q));
  for (int r0 = 0; r0 < rows; r0 += 32 * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
#pragma unroll
      for (int q = 0; q < 8; q++) acc += ring[d][q];
      const int rn = r0 + 32 * DEPTH + d * 32;
      if (rn < rows)
#pragma unroll
        for (int q = 0; q < 8; q++)
          ring[d][q] = __builtin_nontemporal_load((const f32x4 *)row(rn + kg * 8 + q));
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.0f;
}

extern "C" {

// Measurement aid: time `reps` launches of the two read kernels above over in_dev [rows][stride]
// (nsamples valid floats per row; 16-byte aligned, stride and nsamples multiples of 256) with HIP
// events on the context's stream.  ms[0]: average of the linear read of rows * stride floats,
// ms[1]: average of the row-pattern read of rows * nsamples floats.
int earhip_debug_read_bandwidth(earhip_ctx *ctx, const float *in_dev, size_t rows, size_t stride,
                                size_t nsamples, int reps, double ms[2]) {
  return guarded([&] {
    require(ctx != nullptr && in_dev != nullptr && ms != nullptr, "NULL argument");
    require(rows >= 1 && rows < ((size_t)1 << 24) && reps >= 1 && reps <= 1000, "rows / reps out of range");
    require(stride % 256 == 0 && nsamples % 256 == 0 && nsamples >= 256 && nsamples <= stride &&
                ((uintptr_t)in_dev & 15) == 0,
            "buffer must be 16-byte aligned with stride and nsamples multiples of 256");
    ctx->use();
    if (!ctx->level.p) ctx->level.alloc_zero(4, ctx->stream);
    float *sink = reinterpret_cast<float *>(ctx->level.p);  // never written (the kernels' condition is never true)
    hipEvent_t e[4];
    for (auto &x : e) EARHIP_HIP(hipEventCreate(&x));
    const size_t n4 = rows * stride / 4;
    const unsigned lin_blocks = (unsigned)std::min<size_t>((n4 + 255) / 256, (size_t)ctx->num_cus * 32);
    auto linear = [&] {
      hipLaunchKernelGGL(k_read_linear, dim3(lin_blocks), dim3(256), 0, ctx->stream,
                         reinterpret_cast<const f32x4 *>(in_dev), sink, n4);
    };
    auto pattern = [&] {
      hipLaunchKernelGGL(k_read_rows, dim3((unsigned)(nsamples / 256)), dim3(256), 0, ctx->stream, in_dev, sink,
                         (int)rows, stride);
    };
    linear();  // (warm-up launch of each kernel outside its interval)
    EARHIP_HIP(hipEventRecord(e[0], ctx->stream));
    for (int i = 0; i < reps; i++) linear();
    EARHIP_HIP(hipEventRecord(e[1], ctx->stream));
    pattern();
    EARHIP_HIP(hipEventRecord(e[2], ctx->stream));
    for (int i = 0; i < reps; i++) pattern();
    EARHIP_HIP(hipEventRecord(e[3], ctx->stream));
    EARHIP_HIP(hipGetLastError());
    EARHIP_HIP(hipEventSynchronize(e[3]));
    float t = 0.0f;
    EARHIP_HIP(hipEventElapsedTime(&t, e[0], e[1]));
    ms[0] = (double)t / reps;
    EARHIP_HIP(hipEventElapsedTime(&t, e[2], e[3]));
    ms[1] = (double)t / reps;
    for (auto &x : e) (void)hipEventDestroy(x);
  });
}

int earhip_debug_copy_bandwidth(earhip_ctx *ctx, void *host, size_t bytes, int reps, double ms[2]) {
  return guarded([&] {
    require(ctx != nullptr && host != nullptr && ms != nullptr, "NULL argument");
    require(bytes >= 4 && reps >= 1 && reps <= 1000, "bytes / reps out of range");
    ctx->use();
    DevBuf<char> dev;
    dev.alloc(bytes);
    hipEvent_t e[2];
    for (auto &x : e) EARHIP_HIP(hipEventCreate(&x));
    for (int dir = 0; dir < 2; dir++) {
      auto copy = [&] {
        if (dir == 0) EARHIP_HIP(hipMemcpyAsync(dev.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
        else EARHIP_HIP(hipMemcpyAsync(host, dev.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
      };
      copy();  // (warm-up outside the interval)
      EARHIP_HIP(hipEventRecord(e[0], ctx->stream));
      for (int i = 0; i < reps; i++) copy();
      EARHIP_HIP(hipEventRecord(e[1], ctx->stream));
      EARHIP_HIP(hipEventSynchronize(e[1]));
      float t = 0.0f;
      EARHIP_HIP(hipEventElapsedTime(&t, e[0], e[1]));
      ms[dir] = (double)t / reps;
    }
    for (auto &x : e) (void)hipEventDestroy(x);
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
  });
}

/* diagnostic builds (-DEARHIP_HG_PROF) only: the phase sums every wave of two workgroups of the LAST hinge-kernel launch left
 * (gain_hg.h), out[2][8][8]; an ordinary build reports "not built in" */
int earhip_debug_hg_prof(earhip_ctx *ctx, unsigned long long *out128) {
  return guarded([&] {
    require(ctx != nullptr && out128 != nullptr, "NULL argument");
#ifdef EARHIP_HG_PROF
    ctx->use();
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    EARHIP_HIP(hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_hg_prof), sizeof(unsigned long long) * 128));
#else
    fail_invalid("this build has no hinge-kernel phase sums (-DEARHIP_HG_PROF)");
#endif
  });
}

/* diagnostic builds (-DEARHIP_BUILD_PROF) only: the s_memtime stamps thread 0 of two workgroups of the LAST list-builder launch
 * (k_piece_build or k_hinge_build) left at its phase boundaries, out[2][32]; an ordinary build reports "not built in" */
int earhip_debug_build_prof(earhip_ctx *ctx, unsigned long long *out64) {
  return guarded([&] {
    require(ctx != nullptr && out64 != nullptr, "NULL argument");
#ifdef EARHIP_BUILD_PROF
    ctx->use();
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    EARHIP_HIP(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_build_prof), sizeof(unsigned long long) * 64));
#else
    fail_invalid("this build has no list-builder phase stamps (-DEARHIP_BUILD_PROF)");
#endif
  });
}

// Tuning aid: enqueue a probe that samples the shader-cycle and the constant-rate
// counters into device memory `out_dev[2]` (two probes bracket a region to get
// the average shader clock).  wall_clock64 rate: hipDeviceAttributeWallClockRate.
int earhip_debug_clock_probe(earhip_ctx *ctx, void *out_dev) {
  return guarded([&] {
    require(ctx != nullptr && out_dev != nullptr, "NULL argument");
    ctx->use();
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(1), 0, ctx->stream,
                       (unsigned long long *)out_dev);
    EARHIP_HIP(hipGetLastError());
  });
}

int earhip_version(void) { return EARHIP_VERSION; }
const char *earhip_last_error(void) { return g_last_error.c_str(); }

int earhip_device_count(int *count) {
  return guarded([&] {
    require(count != nullptr, "count must not be NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    *count = n;
  });
}

int earhip_ctx_create(int device, void *hip_stream, earhip_ctx **out) {
  return guarded([&] {
    require(out != nullptr, "out must not be NULL");
    int n = 0;
    EARHIP_HIP(hipGetDeviceCount(&n));
    if (n <= 0)
      throw Error{EARHIP_DEVICE_ERROR,
                  "internal error: no HIP device available (libearhip has no CPU fallback)"};
    require(device >= 0 && device < n, "device index out of range");
    // (an owned stream goes with the context whichever way this function is left)
    struct CtxDelete {
      void operator()(earhip_ctx *c) const {
        if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
      }
    };
    std::unique_ptr<earhip_ctx, CtxDelete> c(new earhip_ctx);
    c->device = device;
    EARHIP_HIP(hipSetDevice(device));
    // the options: once, from the environment (EARHIP_<KEY>); afterwards only earhip_ctx_set_option changes them
    for (int o = 0; o < OPT_COUNT; o++) {
      const std::string name = std::string("EARHIP_") + kOptNames[o];
      if (const char *e = getenv(name.c_str())) c->opt[o].set = true, c->opt[o].v = parse_option_value(name, e);
    }
    if (hip_stream) {
      c->stream = (hipStream_t)hip_stream;
    } else {
      EARHIP_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
      c->own_stream = true;
    }
    hipDeviceProp_t prop;
    EARHIP_HIP(hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->hinge_build_lds = hinge_build_allow_lds(prop);  // (this device's limit)
    c->apply_options();
    *out = c.release();
  });
}

int earhip_ctx_set_option(earhip_ctx *ctx, const char *key, const char *value) {
  return guarded([&] {
    require(ctx != nullptr && key != nullptr, "NULL argument");
    std::string k(key);
    for (auto &ch : k) ch = (char)std::toupper((unsigned char)ch);
    if (k.compare(0, 7, "EARHIP_") == 0) k = k.substr(7);
    for (int o = 0; o < OPT_COUNT; o++)
      if (k == kOptNames[o]) {
        const bool set = value != nullptr && value[0] != 0;
        const int v = set ? parse_option_value(k, value) : 0;  // (refused before anything changes)
        ctx->opt[o].set = set;
        ctx->opt[o].v = v;
        ctx->apply_options();
        return;
      }
    fail_invalid("unknown option '" + std::string(key) + "'");
  });
}

int earhip_ctx_get_option(const earhip_ctx *ctx, const char *key, int *is_set, int *value) {
  return guarded([&] {
    require(ctx != nullptr && key != nullptr && is_set != nullptr && value != nullptr, "NULL argument");
    std::string k(key);
    for (auto &ch : k) ch = (char)std::toupper((unsigned char)ch);
    if (k.compare(0, 7, "EARHIP_") == 0) k = k.substr(7);
    for (int o = 0; o < OPT_COUNT; o++)
      if (k == kOptNames[o]) {
        *is_set = ctx->opt[o].set ? 1 : 0;
        *value = ctx->opt[o].v;
        return;
      }
    fail_invalid("unknown option '" + std::string(key) + "'");
  });
}

int earhip_ctx_destroy(earhip_ctx *ctx) {
  return guarded([&] {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    {
      std::lock_guard<std::mutex> lock(g_policy_mutex);
      for (size_t i = 0; i < g_policy.size(); i++)
        if (g_policy[i].first == ctx) {
          g_policy.erase(g_policy.begin() + i);
          break;
        }
    }
    for (earhip_vbs *v : ctx->pinned_adapters) vbs_orphan(v);
    ctx->pinned_adapters.clear();
    for (const auto &r : ctx->host_ranges) {
      if (r.owned) (void)hipHostFree(const_cast<char *>(r.base));
      else (void)hipHostUnregister(const_cast<char *>(r.base));
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
  });
}

// ---- host memory the device reaches directly (include/earhip.h, group A) ------------------------------
int earhip_host_alloc(earhip_ctx *ctx, size_t bytes, void **out) {
  return guarded([&] {
    require(ctx != nullptr && out != nullptr && bytes > 0, "NULL argument or zero size");
    ctx->use();
    void *p = nullptr;
    EARHIP_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    ctx->host_ranges.push_back({static_cast<const char *>(p), bytes, true});
    *out = p;
  });
}

int earhip_host_register(earhip_ctx *ctx, void *ptr, size_t bytes) {
  return guarded([&] {
    require(ctx != nullptr && ptr != nullptr && bytes > 0, "NULL argument or zero size");
    ctx->use();
    EARHIP_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    // the kernels use the host address itself: true on this platform (unified addressing), checked once here
    void *dptr = nullptr;
    if (hipHostGetDevicePointer(&dptr, ptr, 0) != hipSuccess || dptr != ptr) {
      (void)hipHostUnregister(ptr);
      fail_internal("registered host memory is not mapped at its own address on this platform");
    }
    ctx->host_ranges.push_back({static_cast<const char *>(ptr), bytes, false});
  });
}

int earhip_host_release(earhip_ctx *ctx, void *ptr) {
  return guarded([&] {
    require(ctx != nullptr, "ctx must not be NULL");
    if (!ptr) return;
    ctx->use();
    for (size_t i = 0; i < ctx->host_ranges.size(); i++)
      if (ctx->host_ranges[i].base == static_cast<const char *>(ptr)) {
        (void)hipStreamSynchronize(ctx->stream);  // a queued copy or kernel may still use the range
        const bool owned = ctx->host_ranges[i].owned;
        ctx->host_ranges.erase(ctx->host_ranges.begin() + i);
        if (owned) EARHIP_HIP(hipHostFree(ptr));
        else EARHIP_HIP(hipHostUnregister(ptr));
        return;
      }
    fail_invalid("pointer is not the start of a range of earhip_host_alloc / earhip_host_register");
  });
}

int earhip_ctx_synchronize(earhip_ctx *ctx) {
  return guarded([&] {
    require(ctx != nullptr, "ctx must not be NULL");
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
  });
}

int earhip_ctx_set_strict(earhip_ctx *ctx, int strict) {
  return guarded([&] {
    require(ctx != nullptr, "ctx must not be NULL");
    ctx->strict = strict != 0;
  });
}

static void check_policy_args(earhip_ctx *ctx, int n_in, int n_out, const float *const *in,
                              float *const *out, int64_t r0, int64_t r1) {
  require(ctx != nullptr, "ctx must not be NULL");
  require(n_in >= 1 && n_out >= 1, "n_in and n_out must be >= 1");
  require(in != nullptr && out != nullptr, "in and out must not be NULL");
  require(r0 >= 0 && r1 >= r0, "invalid sample range");
  require(r1 - r0 < (int64_t)1 << 30, "sample range too long");
}

int earhip_interp_apply_interp(earhip_ctx *ctx, int n_in, int n_out, const float *const *in,
                               float *const *out, int64_t range_start, int64_t range_end,
                               int64_t block_start, int64_t start, int64_t end,
                               const float *start_point, const float *end_point) {
  return guarded([&] {
    check_policy_args(ctx, n_in, n_out, in, out, range_start, range_end);
    require(start_point && end_point, "points must not be NULL");
    require(end > start, "interpolation curve must have end > start");
    const int64_t first = block_start + range_start - start;
    require(first > -((int64_t)1 << 30) && first + (range_end - range_start) < ((int64_t)1 << 30),
            "sample range too far from the interpolation curve");
    ctx->use();
    GainStage *st = policy_cache(ctx)->get(ctx, n_in, n_out, true);
    // a two-point curve per input channel, evaluated as ONE ramp (also outside
    // [start, end) and between equal points, like a direct call of apply_interp,
    // gain_interpolator.hpp:147-169)
    const int64_t t[2] = {start, end};
    std::vector<float> &rows = st->rows;
    rows.resize(2 * (size_t)n_out);
    for (int m = 0; m < n_in; m++) {
      std::memcpy(rows.data(), start_point + (size_t)m * n_out, sizeof(float) * n_out);
      std::memcpy(rows.data() + n_out, end_point + (size_t)m * n_out, sizeof(float) * n_out);
      st->curves.set_object(m, 2, t, rows.data());
    }
    st->curves.commit(ctx);
    st->run_host(block_start + range_start, in, out, range_start, range_end);
  });
}

int earhip_interp_apply_constant(earhip_ctx *ctx, int n_in, int n_out, const float *const *in,
                                 float *const *out, int64_t range_start, int64_t range_end,
                                 const float *point) {
  return guarded([&] {
    check_policy_args(ctx, n_in, n_out, in, out, range_start, range_end);
    require(point != nullptr, "point must not be NULL");
    ctx->use();
    GainStage *st = policy_cache(ctx)->get(ctx, n_in, n_out, false);
    const int64_t t[1] = {0};
    for (int m = 0; m < n_in; m++) st->curves.set_object(m, 1, t, point + (size_t)m * n_out);
    st->curves.commit(ctx);
    st->run_host(range_start, in, out, range_start, range_end);
  });
}

int earhip_gain_interp_create(earhip_ctx *ctx, int n_in, int n_out, earhip_gain_interp **out) {
  return guarded([&] {
    require(ctx != nullptr && out != nullptr, "ctx/out must not be NULL");
    require(n_in >= 1 && n_out >= 1, "n_in and n_out must be >= 1");
    ctx->use();
    std::unique_ptr<earhip_gain_interp> g(new earhip_gain_interp);
    g->stage.reset(new GainStage(ctx, n_in, n_out));
    *out = g.release();
  });
}

int earhip_gain_interp_destroy(earhip_gain_interp *gi) {
  return guarded([&] {
    if (!gi) return;
    (void)hipStreamSynchronize(gi->stage->ctx->stream);
    delete gi;
  });
}

int earhip_gain_interp_set_points(earhip_gain_interp *gi, int npoints, const int64_t *times,
                                  const float *values) {
  return guarded([&] {
    require(gi != nullptr, "gi must not be NULL");
    require(npoints >= 1, "interp_points must not be empty");
    require(times != nullptr && values != nullptr, "times/values must not be NULL");
    GainStage *st = gi->stage.get();
    st->ctx->use();
    // libear compares whole points (all inputs x outputs) to choose the constant
    // path (gain_interpolator.hpp:68-70,141-143)
    const size_t psz = (size_t)st->n_in * st->n_out;
    std::vector<uint8_t> flat(npoints, 0);
    for (int k = 1; k < npoints; k++) {
      bool same = true;
      for (size_t i = 0; same && i < psz; i++)
        same = values[(size_t)(k - 1) * psz + i] == values[(size_t)k * psz + i];
      flat[k] = same ? 1 : 0;
    }
    // values [npoints][n_in][n_out] -> per input channel [npoints][n_out]
    std::vector<float> rows((size_t)npoints * st->n_out);
    for (int m = 0; m < st->n_in; m++) {
      for (int k = 0; k < npoints; k++)
        std::memcpy(&rows[(size_t)k * st->n_out],
                    values + ((size_t)k * st->n_in + m) * st->n_out, sizeof(float) * st->n_out);
      st->curves.set_object(m, npoints, times, rows.data(), flat.data());
    }
    st->curves.commit(st->ctx);
  });
}

int earhip_gain_interp_process(earhip_gain_interp *gi, int64_t block_start, size_t nsamples,
                               const float *const *in, float *const *out) {
  return guarded([&] {
    require(gi != nullptr, "gi must not be NULL");
    require(in != nullptr && out != nullptr, "in and out must not be NULL");
    require(nsamples < ((size_t)1 << 30), "nsamples too large");
    gi->stage->ctx->use();
    gi->stage->run_host(block_start, in, out, 0, (int64_t)nsamples);
  });
}

int earhip_gain_interp_process_device(earhip_gain_interp *gi, int64_t block_start,
                                      size_t nsamples, const float *in_dev, size_t in_stride,
                                      float *out_dev, size_t out_stride) {
  return guarded([&] {
    require(gi != nullptr, "gi must not be NULL");
    require(in_dev != nullptr && out_dev != nullptr, "device pointers must not be NULL");
    require(nsamples < ((size_t)1 << 30), "nsamples too large");
    require(in_stride % 4 == 0 && out_stride % 4 == 0 && ((uintptr_t)in_dev & 15) == 0 &&
                ((uintptr_t)out_dev & 15) == 0,
            "device buffers must be 16-byte aligned with strides that are multiples of 4");
    if (nsamples == 0) return;
    gi->stage->ctx->use();
    gi->stage->run_device(block_start, (int)nsamples, in_dev, in_stride, out_dev, out_stride);
  });
}

}  // extern "C"
