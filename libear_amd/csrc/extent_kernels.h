// extent_kernels.h — libear's polar extent panner (src/object_based/polar_extent.cpp:12-302, core:
// src/object_based/polar_extent_scalar.cpp:25-108 / polar_extent_simd.hpp:28-134) as a batch kernel: ONE WAVE
// per (object, metadata block).  Included by api_panner.hip behind the point source panner's device code.
//
// libear spreads an object with width / height / depth over 1652 points on the sphere (37 rows, 5 degrees
// apart), each with the point source panner's gains precomputed: a weight per point (1 inside a "stadium"
// of width x height around the object's direction, fading to 0 over 10 degrees), the weighted sum of the
// points' gain vectors, normalised, blended by power with the point-source gains for extents under 10
// degrees, and for depth != 0 the rms of a near and a far rendering.  The weighting and the sum are the
// reference's one SIMD kernel (float, xsimd batches of 4-16 points); everything before it is scalar double
// set-up per call.
//
// Here the work is split between two kernels.  k_pan_objects (a thread per position, api_panner.hip) does what
// is scalar in libear — extentMod, the share of the spread part, the basis at the object's direction, the
// stadium's parameters, the point source panner's gains for extents under 10 degrees — and leaves a JOB record
// per position that needs spreading.  k_pan_objects_extent takes ONE WAVE per job:
//   * the 1652 points are stored in 26 spatially compact chunks of 64 (bands of 8 rows cut along the azimuth)
//     with, per chunk, its mean direction, its angular radius and the float sum of its points' gain vectors;
//   * lane c classifies chunk c against the extent with one dot product: entirely outside (further from the
//     object than half the larger dimension + fade: every weight is 0), entirely inside (closer than half the
//     smaller dimension: every weight is 1 — the chunk's precomputed sum is added, like libear's all-ones
//     batches, polar_extent_simd.hpp:108-121), or on the boundary;
//   * only boundary chunks are evaluated: a lane per point (positions and gains read as whole 256-byte rows),
//     the per-lane partial sums meet in a butterfly of lane exchanges;
//   * from there lane s carries loudspeaker s: normalisation, the blend with the point-source gains, depth's
//     rms of two renderings, gain, LFE mask, the direct / diffuse split.
// Arithmetic types follow the reference: float weights and sums (extent_float_t), double around them.  The
// order of the float additions differs from the scalar core's (as it does between the reference's scalar and
// SIMD cores, which its tests hold to 1e-5: tests/extent_tests.cpp:140-169).
#pragma once

namespace earhip {

constexpr int kExtentMaxChunks = 32;

struct ExtentTable {
  int n_points;  // 1652
  int n_padded;  // multiple of 64; padded points carry zero gains
  int n_chunks;  // n_padded / 64 (26)
  int n_pv;      // panner outputs: loudspeakers without LFE, or 2 for 0+2+0
  const float *xs, *ys, *zs;  // [n_padded], chunk after chunk
  const float *gains;         // [n_pv][n_padded]
  const float *chunk_dir;     // [3][kExtentMaxChunks]: unit mean direction of a chunk's points
  const float *chunk_rad;     // [kExtentMaxChunks]: largest angle between it and a point of the chunk (radians)
  const float *chunk_sum;     // [kExtentMaxChunks][kMaxPanOut]: float sum of the chunk's gain vectors
};

constexpr int kExtentRows = 37;          // polar_extent.cpp:14
constexpr double kExtentFade = 10.0;     // :13
constexpr int kExtentWaves = 4;          // positions per workgroup

// eigen_helpers.hpp:26-48 for 2, 3 and 4 points
__host__ __device__ inline double ext_interp(double x, const double *xp, const double *yp, int n) {
  if (x <= xp[0]) return yp[0];
  for (int i = 0; i + 1 < n; i++)
    if (xp[i + 1] > x) return yp[i] + (yp[i + 1] - yp[i]) / (xp[i + 1] - xp[i]) * (x - xp[i]);
  return yp[n - 1];
}
__host__ __device__ inline double ext_degrees(double r) { return r * 180.0 / 3.14159265358979323846264338327950288; }
__host__ __device__ inline double ext_radians(double d) { return d * 3.14159265358979323846264338327950288 / 180.0; }

// polar_extent.cpp:62-70
__host__ __device__ inline double extent_mod(double extent, double distance) {
  const double x2[2] = {0.0, 360.0}, y2[2] = {0.2, 1.0};
  const double size = ext_interp(extent, x2, y2, 2);
  const double extent1 = 4.0 * ext_degrees(atan2(size, 1.0));
  const double x3[3] = {0.0, extent1, 360.0}, y3[3] = {0.0, extent, 360.0};
  return ext_interp(4.0 * ext_degrees(atan2(size, distance)), x3, y3, 3);
}

// PolarExtent::handle's renderings (:290-302): one, or with depth a near and a far one; per rendering the
// distance-modified width and height and the share of the spread part (:263-266)
struct ExtentPasses {
  int n_pass;
  double w[2], h[2], spread[2];
};
__device__ inline void extent_passes(Vec3 p, double w0, double h0, double dp, ExtentPasses &X) {
  const double distance = sqrt(p.x * p.x + p.y * p.y + p.z * p.z);
  X.n_pass = dp != 0.0 ? 2 : 1;
  for (int k = 0; k < 2; k++) {
    double d = distance;
    if (X.n_pass == 2) d = fmax(k == 0 ? distance - dp / 2.0 : distance + dp / 2.0, 0.0);
    X.w[k] = extent_mod(w0, d);
    X.h[k] = extent_mod(h0, d);
    const double x2[2] = {0.0, kExtentFade}, y2[2] = {0.0, 1.0};
    X.spread[k] = ext_interp(fmax(X.w[k], X.h[k]), x2, y2, 2);
  }
}

// what the core needs per call (PolarExtentCoreContext, polar_extent_core.hpp:30-42)
struct ExtentWeighting {
  bool is_circular;
  float basis[9], circle_test[2], right_circle_centre[2];
  float cos_start, cos_end, sin_start, sin_end, m, c;
  // every point further than r_out from the object's direction weighs 0, every point closer than r_in weighs 1
  // (radians; r_out = half the larger dimension after its modification + fade, r_in = half the smaller one)
  float r_in, r_out;
};

// what k_pan_objects leaves for k_pan_objects_extent
struct ExtentJob {
  int pos;                 // index of the position in the call's arrays
  int n_pass;              // renderings: 1, or 2 with depth
  double spread[2];        // share of the spread part per rendering (:263-266)
  ExtentWeighting W[2];
  float dir[3];            // the object's direction (unit; +y for the origin, :52-60)
  double psq[kMaxPanOut];  // the point source panner's gains squared (zero when no rendering needs them)
};

// setup_weighting_function + setup_angle_to_weight (polar_extent.cpp:186-255); calcBasis (:80-91)
__device__ inline void extent_setup(Vec3 position, double width, double height, ExtentWeighting &W) {
  const double pi = 3.14159265358979323846264338327950288;
  width = ext_radians(width) / 2;
  height = ext_radians(height) / 2;
  const double n = sqrt(position.x * position.x + position.y * position.y + position.z * position.z);
  const Vec3 u = n < 1e-10 ? Vec3{0.0, 1.0, 0.0} : Vec3{position.x / n, position.y / n, position.z / n};
  double az = -ext_degrees(atan2(u.x, u.y));
  const double el = ext_degrees(atan2(u.z, hypot(u.x, u.y)));
  if (fabs(el) > 90.0 - 1e-5) az = 0.0;  // near the poles the azimuth is indeterminate
  // rows along x, y, z of a frame whose +y points at the object (geom.hpp:91-98)
  Vec3 r0 = polar_to_cart(az - 90.0, 0.0, 1.0), r1 = polar_to_cart(az, el, 1.0), r2 = polar_to_cart(az, el + 90.0, 1.0);
  if (height > width) {  // wider than high from here on: the frame rotated about the object's direction
    const double t = height;
    height = width;
    width = t;
    const Vec3 old0 = r0;
    r0 = r2;
    r2 = {-old0.x, -old0.y, -old0.z};
  }
  W.basis[0] = (float)r0.x, W.basis[1] = (float)r0.y, W.basis[2] = (float)r0.z;
  W.basis[3] = (float)r1.x, W.basis[4] = (float)r1.y, W.basis[5] = (float)r1.z;
  W.basis[6] = (float)r2.x, W.basis[7] = (float)r2.y, W.basis[8] = (float)r2.z;
  // the ends of a wide extent meet at the back (:236-242)
  const double width_full = pi + height;
  const double x3[3] = {0.0, pi / 2.0, pi}, y3[3] = {0.0, pi / 2.0, width_full};
  const double width_mod = ext_interp(width, x3, y3, 3);
  const double x4[4] = {0.0, pi / 4.0, pi / 2.0, pi}, y4[4] = {width_mod, width_mod, width, width};
  width = ext_interp(height, x4, y4, 4);
  W.is_circular = (width - height) < 1e-6;
  const double circle_pos = width - height;
  W.right_circle_centre[0] = (float)sin(circle_pos);
  W.right_circle_centre[1] = (float)cos(circle_pos);
  W.circle_test[0] = (float)-cos(circle_pos);
  W.circle_test[1] = (float)sin(circle_pos);
  const double start_angle = height, end_angle = height + ext_radians(kExtentFade);
  W.cos_start = (float)(start_angle < pi ? cos(start_angle) : -1.0);
  W.cos_end = (float)(end_angle < pi ? cos(end_angle) : -(1.0 + 1e-6));
  W.sin_start = (float)(start_angle < pi / 2 ? sin(start_angle) : 1.0);
  W.sin_end = (float)(end_angle < pi / 2 ? sin(end_angle) : 1.0 + 1e-6);
  W.m = (float)(1.0 / (start_angle - end_angle));
  W.c = (float)(-W.m * end_angle);
  W.r_in = (float)start_angle;
  W.r_out = (float)(width + ext_radians(kExtentFade));
}

// polar_extent_scalar.cpp:34-76 (float)
__device__ inline float extent_weight(const ExtentWeighting &W, float x, float y, float z) {
  auto from_cos = [&](float ca) { return ca >= W.cos_start ? 1.0f : ca <= W.cos_end ? 0.0f : W.m * acosf(ca) + W.c; };
  auto from_sin = [&](float sa) { return sa <= W.sin_start ? 1.0f : sa >= W.sin_end ? 0.0f : W.m * asinf(sa) + W.c; };
  const float ty = x * W.basis[3] + y * W.basis[4] + z * W.basis[5];
  if (W.is_circular) return from_cos(ty);
  const float tx = x * W.basis[0] + y * W.basis[1] + z * W.basis[2];
  const float tz = x * W.basis[6] + y * W.basis[7] + z * W.basis[8];
  const float rx = fabsf(tx);
  if (rx * W.circle_test[0] + ty * W.circle_test[1] >= 0.0f) return from_sin(fabsf(tz));
  return from_cos(rx * W.right_circle_centre[0] + ty * W.right_circle_centre[1]);
}

// direct / diffuse [npos][n_full] float for the jobs k_pan_objects left.  Wave w of the launch takes jobs[w];
// waves past *job_count leave at once.
static __global__ void __launch_bounds__(64 * kExtentWaves)
k_pan_objects_extent(PanParams P, ExtentTable E, const ExtentJob *jobs, const unsigned *job_count, const double *gain,
                     const double *diffuse, float *direct, float *diff) {
  const int lane = threadIdx.x & 63;
  const unsigned wi = __builtin_amdgcn_readfirstlane(blockIdx.x * kExtentWaves + (threadIdx.x >> 6));
  if (wi >= *job_count) return;  // (whole waves)
  const ExtentJob &J = jobs[wi];
  const size_t i = (size_t)J.pos;
  const int S = E.n_pv;
  const int n_pass = J.n_pass;
  // lane c looks at chunk c: the angle between its mean direction and the object's
  float chunk_angle = 0.0f, chunk_rad = 0.0f;
  if (lane < E.n_chunks) {
    const float cd = E.chunk_dir[lane] * J.dir[0] + E.chunk_dir[kExtentMaxChunks + lane] * J.dir[1] +
                     E.chunk_dir[2 * kExtentMaxChunks + lane] * J.dir[2];
    chunk_angle = acosf(fminf(fmaxf(cd, -1.0f), 1.0f));
    chunk_rad = E.chunk_rad[lane];
  }
  // From here on lane s (< S) carries the panner's output s.
  const double psq = lane < S ? J.psq[lane] : 0.0;
  double fin = 0.0;
  for (int k = 0; k < n_pass; k++) {  // calc_pv_spread (:257-288)
    const double amount_spread = J.spread[k], amount_point = 1.0 - amount_spread;
    double out = amount_point > 1e-10 ? amount_point * psq : 0.0;
    if (amount_spread > 1e-10) {
      const ExtentWeighting &W = J.W[k];
      const float margin = 1e-3f;  // (radians: float rounding of the angles and of the weights' own thresholds)
      const bool is_chunk = lane < E.n_chunks;
      const bool outside = is_chunk && chunk_angle - chunk_rad > W.r_out + margin;
      const bool inside = is_chunk && chunk_angle + chunk_rad + margin < W.r_in;
      unsigned long long m_inside = __ballot(inside);
      unsigned long long m_edge = __ballot(is_chunk && !outside && !inside);
      // chunks entirely inside: their precomputed sums (lane s adds loudspeaker s)
      float total = 0.0f;
      while (m_inside) {
        const int c = __ffsll((long long)m_inside) - 1;
        m_inside &= m_inside - 1;
        if (lane < S) total += E.chunk_sum[c * kMaxPanOut + lane];
      }
      // chunks on the boundary: a lane per point
      float acc[kMaxPanOut];
#pragma unroll
      for (int s = 0; s < kMaxPanOut; s++) acc[s] = 0.0f;
      const bool any_edge = m_edge != 0;
      while (m_edge) {
        const int c = __ffsll((long long)m_edge) - 1;
        m_edge &= m_edge - 1;
        const int q = c * 64 + lane;
        const float w = extent_weight(W, E.xs[q], E.ys[q], E.zs[q]);
        if (__ballot(w != 0.0f) == 0) continue;
        const float *g = E.gains + q;
#pragma unroll
        for (int s = 0; s < kMaxPanOut; s++)
          if (s < S) acc[s] += w * g[(size_t)s * E.n_padded];
      }
      if (any_edge) {
#pragma unroll
        for (int s = 0; s < kMaxPanOut; s++)
          if (s < S) {
            float v = acc[s];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if (lane == s) total += v;
          }
      }
      float n2 = total * total;  // (lanes >= S hold 0)
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) n2 += __shfl_xor(n2, off);
      const float scale = (float)(1.0 / (double)sqrtf(n2));  // (:281: float norm, double reciprocal, float scaling)
      const float r = total * scale;
      out += amount_spread * (double)(r * r);
    }
    const double v = sqrt(out);
    fin = n_pass == 2 ? fin + v * v : v;
  }
  if (n_pass == 2) fin = sqrt(fin / 2.0);  // (:297-299)
  // lane c writes channel c of the full layout: the panner output that feeds it (none: an LFE channel), gain,
  // sqrt(1 - diffuse) / sqrt(diffuse) (gain_calculator_objects.cpp:47-56)
  const int src = lane < P.n_full ? P.real_of_full[lane] : -1;
  const double got = __shfl(fin, src < 0 ? 0 : src);
  if (lane < P.n_full) {
    const double g = gain ? gain[i] : 1.0, df = diffuse ? diffuse[i] : 0.0;
    const double v = src < 0 ? 0.0 : got * g;
    direct[i * P.n_full + lane] = (float)(v * sqrt(1.0 - df));
    diff[i * P.n_full + lane] = (float)(v * sqrt(df));
  }
}

}  // namespace earhip
