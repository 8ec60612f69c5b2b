// gain_mfma.h — K1 on the fp32 matrix cores: the bus-forming contraction
//
//     bus[col][s] = sum_k  c_k(s) * x_{m(k)}(s) * G[row(k)][col]
//
// over the "slots" k of a tile (gain_kernels.h: a constant piece of a curve is one
// slot, a ramp piece two — start row with c = 1 - p, end row with c = p — libear's
// LinearInterpMatrix accumulation, include/ear/dsp/gain_interpolator.hpp:264-277,
// with the ramp folded into the input).  It is a dense [samples x slots] . [slots x
// columns] product in fp32 and runs on v_mfma_f32_16x16x4_f32, which is exact fp32 (a
// k-ordered fmaf chain, same rounding as the VALU formulation) at the full fp32 rate
// of the chip, and — unlike the VALU formulation — takes the gains as an ordinary
// per-lane operand:
//
//   A fragment (1 VGPR): lane l holds A[row l&15][k = l>>4]
//   B fragment (1 VGPR): lane l holds B[k = l>>4][column l&15]
//   k = 0..3 of one step = four consecutive slots of the tile's list
//
// Which sample an MFMA row stands for, and which bus column an MFMA column, is
// free to choose.  Row i of row-tile r is sample i*NRT + r and column j of
// column-tile c is bus column j*NCT + c, so that a lane's NRT inputs and its NCT
// gains are CONTIGUOUS in memory: one 16-byte load of x and one 12-byte load of
// gains per step and lane instead of NRT + NCT dword loads (the texture-address
// unit was 54% busy with those), and the D fragments of the NRT row tiles
// interleave back into runs of NRT samples for the stores.
//
// The gain rows arrive through plain coalesced vector loads (deep, in-order
// prefetch) instead of wave-uniform scalar loads, whose cache cannot sustain a
// 200 MB stream of misses (measured: the SGPR-operand kernel idles 69% of its
// wave cycles in s_waitcnt, profiles/r01_*).
//
// One wave owns NRT x 16 samples x NCT x 16 columns (NRT*NCT float4 accumulators)
// and walks its share of the tile's slots four at a time; the waves of a workgroup
// split the slot list and are combined through LDS in a fixed order.  Static gains
// cost one slot per object, ramps two, and an object whose curve has points inside
// the tile as many as its pieces need (masked to their sample ranges): metadata that
// ignores the tile grid costs MFMA work in proportion, not a different code path.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "gain_kernels.h"

namespace earhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCT, int NRT>
__global__ void __launch_bounds__(512) k_gain_mix_mfma(GainMixParams P) {
  constexpr int TS = 16 * NRT;      // samples per tile
  constexpr int NC = 16 * NCT;      // columns per wave
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [ngroups][NRT*NCT][64][4]
  // per wave: two batches of 64 slots (16 steps), see run()
  __shared__ int4 slot_stage[8][2][64];
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

  // Generic (slow) path: all pieces of ONE object, one MFMA step per piece with the
  // lanes of the second object slot idle.  Walks the segments like
  // GainInterpolator::process (gain_interpolator.hpp:58-86).  Used for unaligned
  // buffers and for objects whose slots did not fit the tile's list.
  auto single_object = [&](int m) {
    const int base = P.ps.off[m], n = P.ps.cnt[m];
    const float *row = P.in + (size_t)m * P.in_stride + tile_s0;
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

  if (idle) {
  } else if (!P.vec_ok) {
    for (int m = m_lo; m < m_hi; m++) single_object(m);  // unaligned rows: objects split over the waves
  } else {
    const Slot *plain = P.sl.slots + (size_t)tile * kTileSlots * P.M;
    const int *cnt = P.sl.count + tile * 4;
    // lane-constant input offsets: the NRT samples of this lane, as float4 loads;
    // lanes past the end of the call re-read the last vector (their slots are masked)
    const int nvec = (P.nsamples + 3) & ~3;
    int xo[NRT / 4];
#pragma unroll
    for (int q = 0; q < NRT / 4; q++) xo[q] = min(tile_s0 + li * NRT + 4 * q, nvec - 4);
    const unsigned goff0 = (unsigned)(col0 + li * NCT);
    const float lif = (float)(li * NRT);

    // Software pipeline over the steps [lo, hi) of one slot list: slots 8 steps
    // ahead, inputs 4, gain rows 2, next step's A fragments woven between the MFMAs.
    // Steps are issued in groups of 8 with fixed ring positions (no branches and no
    // register moves inside a group); steps past `hi` and slots past `n` get zero
    // coefficients.
    auto run = [&](const Slot *base, int n, auto masked_tag) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      const int nsteps = (n + 3) >> 2;
      const int lo = (int)(((int64_t)nsteps * part) / nparts);
      const int hi = (int)(((int64_t)nsteps * (part + 1)) / nparts);
      if (lo >= hi) return;
      const int4 *sp = reinterpret_cast<const int4 *>(base);
      // Slots reach the lanes through LDS: one coalesced 1 KB load brings the 64 slots of
      // 16 steps into a wave-private buffer a whole batch ahead, and each step reads its
      // four slots from there (LDS has its own counter).  Loading a slot per step and lane
      // from global memory put a dependent load (slot -> input address) into the in-order
      // vector-memory stream, and every wait for it also waited for the inputs in flight.
      int4(&stage)[2][64] = slot_stage[threadIdx.x >> 6];
      auto batch_load = [&](int b) { return sp[min(4 * (lo + 16 * b) + lane, n - 1)]; };
      auto read_slot = [&](int step) {
        const int r = step - lo;
        return stage[(r >> 4) & 1][((r & 15) << 2) + kk];
      };
      auto load_x = [&](const int4 &sv, float (&x)[NRT]) {
        const float *xr = P.in + (size_t)(sv.x & 0xffff) * P.in_stride;
#pragma unroll
        for (int q = 0; q < NRT / 4; q++) {
          const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xr + xo[q]));
#pragma unroll
          for (int e = 0; e < 4; e++) x[q * 4 + e] = v[e];
        }
      };
      auto load_g = [&](const int4 &sv, float (&gv)[NCT]) {
        const float *gp = P.ps.gain + ((unsigned)sv.y * (unsigned)P.ps.row + goff0);
#pragma unroll
        for (int c = 0; c < NCT; c++) gv[c] = gp[c];
      };
      auto make_a = [&](const int4 &sv, int step, const float (&x)[NRT], float (&a)[NRT]) {
        const bool live = 4 * step + kk < n && step < hi;
        const float alpha = live ? __int_as_float(sv.z) : 0.0f, beta = live ? __int_as_float(sv.w) : 0.0f;
        const float c0 = __builtin_fmaf(beta, lif, alpha);  // coefficient of the lane's first sample
        const int r0 = (sv.x >> 16) & 0xff;
        const int rel = li * NRT - r0;                          // s - r0 of the lane's first sample
        const unsigned len = ((unsigned)sv.x >> 24) + 1u - (unsigned)r0;  // r1 - r0
#pragma unroll
        for (int r = 0; r < NRT; r++) {
          float v = x[r] * __builtin_fmaf((float)r, beta, c0);
          if (MASKED) v = (unsigned)(rel + r) < len ? v : 0.0f;  // r0 <= s < r1
          a[r] = v;
        }
      };
      int4 S[8];
      float X[4][NRT], A[2][NRT], G[2][NCT];
      stage[0][lane] = batch_load(0);
      int4 gnext = batch_load(1);
#pragma unroll
      for (int j = 0; j < 8; j++) S[j] = read_slot(lo + j);
#pragma unroll
      for (int j = 0; j < 4; j++) load_x(S[j], X[j]);
      load_g(S[0], G[0]);
      load_g(S[1], G[1]);
      make_a(S[0], lo, X[0], A[0]);
      load_x(S[4], X[0]);
#pragma unroll 1
      for (int i = lo; i < hi; i += 16) {
        // batch (i - lo)/16 + 1 becomes readable (the steps below look 8 steps ahead), the
        // one after it is requested
        stage[(((i - lo) >> 4) + 1) & 1][lane] = gnext;
        gnext = batch_load(((i - lo) >> 4) + 2);
#pragma unroll
        for (int j = 0; j < 16; j++) {
          // step i + j: slot ring position j & 7, inputs (j+1)&3 next, gains j&1
          mma(A[j & 1], G[j & 1]);
          make_a(S[(j + 1) & 7], i + j + 1, X[(j + 1) & 3], A[(j + 1) & 1]);
          // requests in the order they are consumed (the vector-memory counter is in order:
          // waiting for a gain row also waits for everything requested before it)
          load_g(S[(j + 2) & 7], G[j & 1]);
          load_x(S[(j + 5) & 7], X[(j + 1) & 3]);
          S[j & 7] = read_slot(i + j + 8);
#pragma unroll
          for (int k = 0; k < NRT * NCT; k++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, MASKED ? 4 : 2, 0);  // VALU
            if (k < NRT / 4 + 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
          }
          __builtin_amdgcn_sched_barrier(0);  // nothing moves across a step
        }
      }
    };
    run(plain, cnt[0], std::false_type{});
    run(plain + kPlainSlots * (size_t)P.M, cnt[1], std::true_type{});
    // objects whose masked slots did not fit the list: generic path, one wave per
    // tile and column group
    if (part == 0) {
      const int *ovf = P.sl.ovf + (size_t)tile * P.M;
      for (int i = 0; i < cnt[2]; i++) single_object(ovf[i]);
    }
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
