// api_panner.hip — (I) the gain-vector producer for Objects content of include/earhip.h: libear's
// GainCalculatorObjects (src/object_based/gain_calculator_objects.cpp:24-57) for point sources — the polar
// point-source panner of src/common/point_source_panner.cpp (triplets, quads, virtual n-gons, the extra
// height loudspeakers and their downmix, the 0+2+0 stereo downmix), the LFE mask and the
// sqrt(1 - diffuse) / sqrt(diffuse) split — as a BATCH kernel: one thread per (object, metadata block).
//
// At 10^4 x real time a scene of 1024 objects with an ADM block every 20 ms needs 5 * 10^8 gain vectors
// per wall-clock second; the reference computes them one at a time on the host (virtual dispatch over
// region objects, Eigen temporaries).  Here the layout's regions are flattened once, on the host and in
// double precision, into a table of plain records (set-up: cold), and the per-position work — a walk over
// at most ~45 regions with 3x3 products, one or two square roots — runs on the device in double
// precision.  The arithmetic of a region test follows the reference's operation for operation
// (point_source_panner.cpp:43-51, :86-101, :118-136, :157-190), so results agree to the last bits of a
// double and are identical after the cast to float in all but rounding-boundary cases.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "layout_table.h"
#include "tdesign_5200.h"

namespace earhip {

constexpr int kMaxNgon = 16;   // real loudspeakers around one virtual loudspeaker
constexpr int kMaxPanOut = 32; // loudspeakers of a layout (without LFE)

struct PanRegion {
  int kind;                // 0 triplet, 1 quad, 2 virtual n-gon
  int n;                   // vertices
  int out[kMaxNgon];       // REAL channel (without LFE) each vertex's gain is mixed into (extra height
                           // loudspeakers are mixed into the loudspeaker below / above them)
  // triplet: basis[0]; n-gon: basis[i] of the triplet (tri[i][0], tri[i][1], virtual centre)
  double basis[kMaxNgon][9];
  int tri[kMaxNgon][2];
  double centre_downmix[kMaxNgon];
  // quad
  int order[4];
  double poly_x[9], poly_y[9], pos[4][3];
};

struct PanTable {
  int n_regions;
  int n_real;  // loudspeakers without LFE
  const PanRegion *regions;
};

struct Vec3 {
  double x, y, z;
};
__host__ __device__ inline Vec3 vsub(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ inline Vec3 vadd(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__host__ __device__ inline double vdot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline Vec3 vcross(Vec3 a, Vec3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// libear's polar convention (src/common/geom.cpp:82-87): azimuth anticlockwise from the front, degrees
__host__ __device__ inline Vec3 polar_to_cart(double az, double el, double dist) {
  const double pi = 3.14159265358979323846264338327950288;
  const double a = -az * pi / 180.0, e = el * pi / 180.0;
  return {sin(a) * cos(e) * dist, cos(a) * cos(e) * dist, sin(e) * dist};
}

// position^T * basis of a triplet; true when inside (all coordinates >= -1e-11), pv normalised and
// clamped to [0, 1] (point_source_panner.cpp:43-51)
__device__ inline bool triplet_gains(const double *b, Vec3 p, double (&pv)[3]) {
#pragma unroll
  for (int j = 0; j < 3; j++) pv[j] = p.x * b[j] + p.y * b[3 + j] + p.z * b[6 + j];
  const double eps = -1e-11;
  if (!(pv[0] >= eps && pv[1] >= eps && pv[2] >= eps)) return false;
  const double n = sqrt(pv[0] * pv[0] + pv[1] * pv[1] + pv[2] * pv[2]);
#pragma unroll
  for (int j = 0; j < 3; j++) pv[j] = fmin(fmax(pv[j] / n, 0.0), 1.0);
  return true;
}

// first real root in (-1e-10, 1 + 1e-10) of poly . position, clamped (point_source_panner.cpp:157-190)
__device__ inline bool quad_pan(const double *poly, Vec3 p, double &out) {
  const double a = poly[0] * p.x + poly[1] * p.y + poly[2] * p.z;
  const double b = poly[3] * p.x + poly[4] * p.y + poly[5] * p.z;
  const double c = poly[6] * p.x + poly[7] * p.y + poly[8] * p.z;
  const double eps = 1e-10;
  double roots[2];
  int nr = 0;
  if (fabs(c) < eps) {
    roots[nr++] = 0.0;
  } else if (fabs(a) < eps) {
    roots[nr++] = -c / b;
  } else {
    const double det = b * b - 4.0 * a * c;
    if (det > eps) {
      roots[nr++] = (-b + sqrt(det)) / (2.0 * a);
      roots[nr++] = (-b - sqrt(det)) / (2.0 * a);
    } else if (det > -eps) {
      roots[nr++] = -b / (2.0 * a);
    }
  }
  for (int i = 0; i < nr; i++)
    if (-eps < roots[i] && roots[i] < 1.0 + eps) {
      out = fmin(fmax(roots[i], 0.0), 1.0);
      return true;
    }
  return false;
}

// One region's answer for a direction: false when the region does not take it, else its gains ADDED into
// real[] (extra loudspeakers already mixed into the real ones).  Triplet :43-51, VirtualNgon :86-101,
// QuadRegion :118-136, :157-190.
__device__ inline bool region_try(const PanRegion &R, Vec3 p, double (&real)[kMaxPanOut]) {
  if (R.kind == 0) {
    double pv[3];
    if (!triplet_gains(R.basis[0], p, pv)) return false;
    for (int k = 0; k < 3; k++) real[R.out[k]] += pv[k];
    return true;
  }
  if (R.kind == 1) {
    double x, y;
    if (!(quad_pan(R.poly_x, p, x) && quad_pan(R.poly_y, p, y))) return false;
    double pvs[4];
    pvs[R.order[0]] = (1 - x) * (1 - y);
    pvs[R.order[1]] = x * (1 - y);
    pvs[R.order[2]] = x * y;
    pvs[R.order[3]] = (1 - x) * y;
    Vec3 vel = {0.0, 0.0, 0.0};
    for (int k = 0; k < 4; k++) {
      vel.x += pvs[k] * R.pos[k][0];
      vel.y += pvs[k] * R.pos[k][1];
      vel.z += pvs[k] * R.pos[k][2];
    }
    if (!(vdot(vel, p) > 0)) return false;
    const double n = sqrt(pvs[0] * pvs[0] + pvs[1] * pvs[1] + pvs[2] * pvs[2] + pvs[3] * pvs[3]);
    for (int k = 0; k < 4; k++) real[R.out[k]] += pvs[k] / n;
    return true;
  }
  for (int t = 0; t < R.n; t++) {
    double pv[3];
    if (triplet_gains(R.basis[t], p, pv)) {
      // the virtual centre's gain is distributed to the real loudspeakers, then the n-gon's
      // gains are normalised (:86-101)
      double g[kMaxNgon];
      double n2 = 0.0;
      for (int k = 0; k < R.n; k++) {
        double v = 0.0;
        if (k == R.tri[t][0]) v = pv[0];
        if (k == R.tri[t][1]) v = pv[1];
        g[k] = v + R.centre_downmix[k] * pv[2];
        n2 += g[k] * g[k];
      }
      const double n = sqrt(n2);
      for (int k = 0; k < R.n; k++) real[R.out[k]] += g[k] / n;
      return true;
    }
  }
  return false;
}

// the mix-down of the extra loudspeakers leaves a vector that is normalised again
// (PointSourcePannerDownmix::handle, :239-250)
__device__ inline void pan_normalise(const PanTable &T, double (&real)[kMaxPanOut]) {
  double n2 = 0.0;
  for (int c = 0; c < T.n_real; c++) n2 += real[c] * real[c];
  const double n = sqrt(n2);
  for (int c = 0; c < T.n_real; c++) real[c] /= n;
}

// Gains of the real loudspeakers (without LFE) for one direction: the first region that takes the
// position (PolarPointSourcePanner::handle, :208-217).  false: no region took it.
__device__ inline bool pan_full(const PanTable &T, Vec3 p, double (&real)[kMaxPanOut]) {
  for (int c = 0; c < T.n_real; c++) real[c] = 0.0;
  bool found = false;
  for (int ri = 0; ri < T.n_regions && !found; ri++) found = region_try(T.regions[ri], p, real);
  if (!found) return false;
  pan_normalise(T, real);
  return true;
}

struct PanParams {
  PanTable table;      // the layout's regions; stereo: those of 0+5+0
  int stereo;          // 0+2+0: pan with 0+5+0, downmix (point_source_panner.cpp:374-398)
  int n_full;          // loudspeakers of the layout incl. LFE
  int full_index[kMaxPanOut];  // real loudspeaker (without LFE) -> index in the full layout
  int stereo_index[2];         // 0+2+0: M+030, M-030
  int real_of_full[kMaxPanOut]; // channel of the full layout -> the panner output that feeds it, -1: LFE
};

// real[] -> the panner's output channels: the layout's loudspeakers without LFE, or the two of 0+2+0
// (StereoPannerDownmix, point_source_panner.cpp:374-398)
__device__ inline int pan_outputs(const PanParams &P, const double (&real)[kMaxPanOut], double (&pv)[kMaxPanOut]) {
  if (P.stereo) {
    const double s3 = sqrt(3.0) / 3.0, s5 = sqrt(0.5);
    const double l = real[0] + s3 * real[2] + s5 * real[3];
    const double r = real[1] + s3 * real[2] + s5 * real[4];
    const double n = sqrt(l * l + r * r);
    const double front = fmax(real[0], fmax(real[1], real[2])), back = fmax(real[3], real[4]);
    const double lev = pow(0.5, 0.5 * back / (front + back));  // 0 dB at the front to -3 dB at the back
    pv[0] = l / n * lev;
    pv[1] = r / n * lev;
    return 2;
  }
  for (int c = 0; c < P.table.n_real; c++) pv[c] = real[c];
  return P.table.n_real;
}

}  // namespace earhip
#include "extent_kernels.h"
namespace earhip {

// One thread per position: direct / diffuse [npos][n_full] float; *missed counts positions no region took.
// Positions that libear spreads over the sphere — an extent, or a distance under 1, which widens even a
// zero extent (polar_extent.cpp:62-70) — are prepared here (everything that is scalar double work in libear)
// and left as a job for k_pan_objects_extent (a wave each).  jobs == NULL: every distance is 1 and there is
// no extent.
static __global__ void __launch_bounds__(128)
k_pan_objects(PanParams P, size_t npos, const double *az, const double *el, const double *dist, const double *width,
              const double *height, const double *depth, const double *gain, const double *diffuse, float *direct,
              float *diff, unsigned *missed, ExtentJob *jobs, unsigned *job_count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npos) return;
  const Vec3 p = polar_to_cart(az[i], el[i], dist ? dist[i] : 1.0);
  ExtentPasses X;
  X.n_pass = 1;
  X.spread[0] = X.spread[1] = 0.0;
  if (jobs) extent_passes(p, width ? width[i] : 0.0, height ? height[i] : 0.0, depth ? depth[i] : 0.0, X);
  bool spread_any = false, need_point = false;
  for (int k = 0; k < X.n_pass; k++) {
    spread_any = spread_any || X.spread[k] > 1e-10;
    need_point = need_point || (1.0 - X.spread[k]) > 1e-10;
  }
  double real[kMaxPanOut];
  double pv[kMaxPanOut];
  int n_pv = 0;
  float *d = direct + i * P.n_full, *f = diff + i * P.n_full;
  if (need_point) {
    if (!pan_full(P.table, p, real)) {
      for (int c = 0; c < P.n_full; c++) d[c] = f[c] = 0.0f;
      atomicAdd(missed, 1u);
      return;
    }
    n_pv = pan_outputs(P, real, pv);
  }
  if (spread_any) {
    ExtentJob &J = jobs[atomicAdd(job_count, 1u)];
    J.pos = (int)i;
    J.n_pass = X.n_pass;
    for (int k = 0; k < 2; k++) {
      J.spread[k] = X.spread[k];
      // (:270-272: at least half the fade width in each dimension)
      if (k < X.n_pass && X.spread[k] > 1e-10)
        extent_setup(p, fmax(X.w[k], kExtentFade / 2.0), fmax(X.h[k], kExtentFade / 2.0), J.W[k]);
    }
    const double n = sqrt(p.x * p.x + p.y * p.y + p.z * p.z);
    J.dir[0] = n < 1e-10 ? 0.0f : (float)(p.x / n);
    J.dir[1] = n < 1e-10 ? 1.0f : (float)(p.y / n);
    J.dir[2] = n < 1e-10 ? 0.0f : (float)(p.z / n);
    for (int c = 0; c < kMaxPanOut; c++) J.psq[c] = c < n_pv ? pv[c] * pv[c] : 0.0;
    return;
  }
  for (int c = 0; c < P.n_full; c++) d[c] = f[c] = 0.0f;
  const double g = gain ? gain[i] : 1.0, df = diffuse ? diffuse[i] : 0.0;
  const double sd = sqrt(1.0 - df), sf = sqrt(df);
  for (int c = 0; c < n_pv; c++) {
    // PolarExtent without spread: sqrt(amount_point pv^2), with depth the rms of two of them
    // (src/object_based/polar_extent.cpp:257-302); gain; split (gain_calculator_objects.cpp:47-56)
    double v = sqrt((1.0 - X.spread[0]) * (pv[c] * pv[c]));
    if (X.n_pass == 2) {
      const double v2 = sqrt((1.0 - X.spread[1]) * (pv[c] * pv[c]));
      v = sqrt((v * v + v2 * v2) / 2.0);
    }
    v *= g;
    const int o = P.stereo ? P.stereo_index[c] : P.full_index[c];
    d[o] = (float)(v * sd);
    f[o] = (float)(v * sf);
  }
}

// HOA decoder design and the extent panner's point grid: panning values of unit vectors in double,
// [npos][n_pv] (2 for 0+2+0)
static __global__ void __launch_bounds__(128)
k_pan_points(PanParams P, size_t npos, const double *xyz, double *pv_out, unsigned *missed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npos) return;
  const Vec3 p = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  double real[kMaxPanOut], pv[kMaxPanOut];
  const int n_pv = P.stereo ? 2 : P.table.n_real;
  if (!pan_full(P.table, p, real)) {
    atomicAdd(missed, 1u);
    for (int c = 0; c < n_pv; c++) pv_out[i * n_pv + c] = 0.0;
    return;
  }
  pan_outputs(P, real, pv);
  for (int c = 0; c < n_pv; c++) pv_out[i * n_pv + c] = pv[c];
}

// ------------------------------------------------------------------------------------------------
// Set-up (host, double): flatten the regions of a layout
// ------------------------------------------------------------------------------------------------
namespace {

struct SetupChannel {
  std::string name;
  double az, el;          // real position (Channel::polarPosition)
  double az_nom, el_nom;  // nominal position (the BS.2051 table)
};

void invert3(const Vec3 (&rows)[3], double (&inv)[9]) {
  const double m[3][3] = {{rows[0].x, rows[0].y, rows[0].z}, {rows[1].x, rows[1].y, rows[1].z}, {rows[2].x, rows[2].y, rows[2].z}};
  double cof[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      cof[i][j] = m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
    }
  const double inv_det = 1.0 / (m[0][0] * cof[0][0] + m[0][1] * cof[0][1] + m[0][2] * cof[0][2]);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) inv[i * 3 + j] = cof[j][i] * inv_det;  // adjugate / determinant
}

// order of the vertices around their centre (src/common/geom.cpp:37-70)
std::vector<int> vertex_order(const std::vector<Vec3> &v) {
  Vec3 c = {0, 0, 0};
  for (auto &p : v) c = vadd(c, p);
  c = {c.x / v.size(), c.y / v.size(), c.z / v.size()};
  const Vec3 a = vsub(v[0], c);
  Vec3 b = {0, 0, 0};
  double best = 1e300;
  for (size_t i = 1; i < v.size(); i++) {
    const Vec3 r = vsub(v[i], c);
    const double col = std::fabs(vdot(r, a));
    if (col < best) best = col, b = r;
  }
  std::vector<double> ang(v.size());
  for (size_t i = 0; i < v.size(); i++) {
    const Vec3 r = vsub(v[i], c);
    ang[i] = std::atan2(vdot(r, a), vdot(r, b));
  }
  std::vector<int> idx(v.size());
  for (size_t i = 0; i < v.size(); i++) idx[i] = (int)i;
  std::sort(idx.begin(), idx.end(), [&](int x, int y) { return ang[x] < ang[y]; });
  return idx;
}

void poly_basis(const Vec3 (&q)[4], double (&rows)[9]) {  // point_source_panner.cpp:138-155
  const Vec3 a = q[0], b = q[1], c = q[2], d = q[3];
  const Vec3 r0 = vcross(vsub(b, a), vsub(c, d));
  const Vec3 r1 = vadd(vcross(a, vsub(c, d)), vcross(vsub(b, a), d));
  const Vec3 r2 = vcross(a, d);
  const Vec3 r[3] = {r0, r1, r2};
  for (int i = 0; i < 3; i++) rows[3 * i] = r[i].x, rows[3 * i + 1] = r[i].y, rows[3 * i + 2] = r[i].z;
}

const LayoutEntry &layout_entry(const char *name) {
  require(name != nullptr, "layout name must not be NULL");
  for (int i = 0; i < kNumLayouts; i++)
    if (std::strcmp(kLayouts[i].name, name) == 0) return kLayouts[i];
  throw Error{EARHIP_UNKNOWN_LAYOUT, std::string("unknown layout: ") + name};
}

// the regions of a layout with precomputed hull facets (configureFullPolarPanner, :478-561).
// real_az / real_el (optional, one per channel of the full layout): the loudspeakers' real positions; the
// layer logic and the facets follow the nominal ones, the geometry the real ones, as in libear
std::vector<PanRegion> build_regions(const LayoutEntry &L, int &n_real, const double *real_az = nullptr,
                                     const double *real_el = nullptr) {
  require(L.facets != nullptr, "layout has no facet table");
  std::vector<SetupChannel> real;
  for (int c = 0; c < L.n; c++)
    if (!L.channels[c].is_lfe)
      real.push_back({L.channels[c].name, real_az ? real_az[c] : L.channels[c].azimuth,
                      real_el ? real_el[c] : L.channels[c].elevation, L.channels[c].azimuth, L.channels[c].elevation});
  n_real = (int)real.size();
  require(n_real <= kMaxPanOut, "too many loudspeakers");
  // checkScreenSpeakers (:558-577)
  for (auto &ch : real)
    if (ch.name == "M+SC" || ch.name == "M-SC") {
      const double abs_az = std::fabs(ch.az);
      if (!((5.0 <= abs_az && abs_az < 25.0) || (35.0 <= abs_az && abs_az < 60.0)))
        fail_invalid("M+SC or M-SC has azimuth not in the allowed ranges of 5 to 25 and 35 to 60 degrees");
      if (25.0 < abs_az)
        throw Error{EARHIP_NOT_IMPLEMENTED, "M+SC and M-SC with azimuths wider than 25 degrees are not currently supported"};
    }
  // extra loudspeakers above / below the mid-layer ones where a layer has none in that direction,
  // each mixed into its mid-layer loudspeaker (extraPosVerticalNominal, :256-349)
  std::vector<Vec3> verts;
  std::vector<int> mix_to;  // augmented vertex -> real loudspeaker
  for (int c = 0; c < n_real; c++) verts.push_back(polar_to_cart(real[c].az, real[c].el, 1.0)), mix_to.push_back(c);
  const double layers[2][3] = {{-30.0, -70.0, -10.0}, {30.0, 10.0, 70.0}};
  for (auto &layer : layers) {
    double az_range = -1.0, el_sum = 0.0;
    int in_layer = 0;
    for (auto &ch : real)
      if (layer[1] <= ch.el_nom && ch.el_nom <= layer[2])
        az_range = std::max(az_range, std::fabs(ch.az_nom)), el_sum += ch.el, in_layer++;
    const double az_limit = in_layer ? az_range + 40.0 : 0.0;
    const double layer_el = in_layer ? el_sum / in_layer : layer[0];
    for (int c = 0; c < n_real; c++)
      if (-10 <= real[c].el_nom && real[c].el_nom <= 10 && std::fabs(real[c].az) >= az_limit - 1e-5) {
        verts.push_back(polar_to_cart(real[c].az, layer_el, 1.0));
        mix_to.push_back(c);
      }
  }
  // virtual loudspeakers below and (unless the layout has one overhead) above (:442-455)
  std::vector<int> virt;
  bool overhead = false;
  for (auto &ch : real) overhead = overhead || ch.name == "T+000" || ch.name == "UH+180";
  virt.push_back((int)verts.size());
  verts.push_back({0.0, 0.0, -1.0});
  if (!overhead) {
    virt.push_back((int)verts.size());
    verts.push_back({0.0, 0.0, 1.0});
  }
  auto is_virtual = [&](int v) { return std::find(virt.begin(), virt.end(), v) != virt.end(); };
  std::vector<std::vector<int>> facets;
  for (const int *f = L.facets; *f; f += 1 + *f) facets.emplace_back(f + 1, f + 1 + *f);
  std::vector<PanRegion> regions;
  // n-gons around the virtual loudspeakers first (:497-527)
  for (int vv : virt) {
    std::vector<int> ring;  // ascending, like the reference's std::set
    for (auto &f : facets)
      if (std::find(f.begin(), f.end(), vv) != f.end())
        for (int v : f)
          if (v != vv && std::find(ring.begin(), ring.end(), v) == ring.end()) ring.push_back(v);
    std::sort(ring.begin(), ring.end());
    require((int)ring.size() <= kMaxNgon && ring.size() >= 3, "unsupported n-gon size");
    for (int v : ring) require(!is_virtual(v), "invalid triangulation");
    PanRegion R;
    std::memset(&R, 0, sizeof(R));
    R.kind = 2;
    R.n = (int)ring.size();
    std::vector<Vec3> pos;
    for (int k = 0; k < R.n; k++) {
      R.out[k] = mix_to[ring[k]];
      R.centre_downmix[k] = 1.0 / std::sqrt((double)R.n);
      pos.push_back(verts[ring[k]]);
    }
    const std::vector<int> order = vertex_order(pos);
    for (int t = 0; t < R.n; t++) {
      const int a = order[t], b = order[(t + 1) % R.n];
      R.tri[t][0] = a;
      R.tri[t][1] = b;
      const Vec3 rows[3] = {pos[a], pos[b], verts[vv]};
      invert3(rows, R.basis[t]);
    }
    regions.push_back(R);
  }
  // every other facet: a triplet or a quad (:528-559)
  for (auto &f : facets) {
    bool touches = false;
    for (int v : f) touches = touches || is_virtual(v);
    if (touches) continue;
    PanRegion R;
    std::memset(&R, 0, sizeof(R));
    R.n = (int)f.size();
    for (int k = 0; k < R.n; k++) R.out[k] = mix_to[f[k]];
    if (f.size() == 3) {
      R.kind = 0;
      const Vec3 rows[3] = {verts[f[0]], verts[f[1]], verts[f[2]]};
      invert3(rows, R.basis[0]);
    } else if (f.size() == 4) {
      R.kind = 1;
      std::vector<Vec3> pos;
      for (int k = 0; k < 4; k++) {
        pos.push_back(verts[f[k]]);
        R.pos[k][0] = verts[f[k]].x, R.pos[k][1] = verts[f[k]].y, R.pos[k][2] = verts[f[k]].z;
      }
      const std::vector<int> order = vertex_order(pos);
      Vec3 q[4], qs[4];
      for (int k = 0; k < 4; k++) R.order[k] = order[k], q[k] = pos[order[k]];
      for (int k = 0; k < 4; k++) qs[k] = q[(k + 1) % 4];
      poly_basis(q, R.poly_x);
      poly_basis(qs, R.poly_y);
    } else {
      fail_internal("facets with more than 4 vertices are not supported");
    }
    regions.push_back(R);
  }
  return regions;
}

// ---- spherical harmonics (src/hoa/hoa.hpp:16-112; Boost.Math's Legendre functions and factorials from
// their definitions)
double factorial_d(int n) {
  double f = 1.0;
  for (int i = 2; i <= n; i++) f *= i;
  return f;
}
// associated Legendre function P_n^m(x) without the Condon-Shortley phase, upward recurrence in n
double legendre_nm(int n, int m, double x) {
  const double root = std::sqrt((1.0 - x) * (1.0 + x));
  double p0 = 1.0;
  for (int i = 1; i <= m; i++) p0 *= (2.0 * i - 1.0) * root;  // P_m^m
  if (n == m) return p0;
  double p1 = x * (2.0 * m + 1.0) * p0;                         // P_(m+1)^m
  for (int l = m + 2; l <= n; l++) {
    const double p2 = (x * (2.0 * l - 1.0) * p1 - (l + m - 1.0) * p0) / (l - m);
    p0 = p1;
    p1 = p2;
  }
  return p1;
}
enum HoaNorm { kN3D, kSN3D, kFuMa };
double hoa_norm(HoaNorm t, int n, int am) {
  const double sn3d = std::sqrt(factorial_d(n - am) / factorial_d(n + am));
  if (t == kSN3D) return sn3d;
  if (t == kN3D) return std::sqrt(2.0 * n + 1.0) * sn3d;
  // FuMa: defined up to order 3 (hoa.hpp:56-72)
  static const double f[4][4] = {{0.70710678118654752440, 0, 0, 0},
                                 {1.0, 1.0, 0, 0},
                                 {1.0, 1.15470053837925152902, 1.15470053837925152902, 0},
                                 {1.0, 1.18585412256314232322, 1.34164078649987381784, 1.26491106406735172760}};
  if (n > 3) fail_invalid("FuMa normalisation is defined up to order 3");
  return f[n][am] * sn3d;
}

}  // namespace
}  // namespace earhip

using namespace earhip;

struct earhip_panner {
  earhip_ctx *ctx;
  PanParams P;
  DevBuf<PanRegion> regions;
  DevBuf<unsigned> missed;
  // the extent panner's point grid: positions [3][n_padded], gains [n_pv][n_padded], per-chunk direction,
  // radius and gain sum (float)
  ExtentTable E;
  DevBuf<float> ext_pos, ext_gains, ext_chunks;
  // what the thread-per-position kernel leaves for the wave-per-position one (grown at first use; a call is
  // cut into launches of at most kExtentBatch positions, so at most that many records)
  DevBuf<ExtentJob> jobs;
  DevBuf<unsigned> job_count;
  // staging of the host-pointer entry point (grown at first use)
  DevBuf<double> d_in;
  DevBuf<float> d_out;
  PinBuf<double> p_in;
  PinBuf<float> p_out;
  PinBuf<unsigned> p_missed;
};

// PolarExtent's constructor (src/object_based/polar_extent.cpp:16-50, :96-176): the 37 rows of points and the
// point source panner's gains at each of them (device, one thread per point, double), kept as float
static void build_extent_table(earhip_panner &pn) {
  earhip_ctx *ctx = pn.ctx;
  const double pi = 3.14159265358979323846264338327950288;
  struct GridPoint {
    Vec3 v;
    int band;
    double az;
    int row;
  };
  std::vector<GridPoint> grid;
  for (int r = 0; r < kExtentRows; r++) {
    const double el = -90.0 + r * (180.0 / (kExtentRows - 1));
    const double radius = std::cos(el * pi / 180.0);
    int n_points = (int)std::round(((2 * pi * radius) / (2 * pi)) * 2 * (kExtentRows - 1));
    if (n_points == 0) n_points = 1;
    for (int i = 0; i < n_points; i++) {
      const double az = i * (360.0 / n_points);
      grid.push_back({polar_to_cart(az, el, 1.0), r / 8, az, r});
    }
  }
  // The device kernel classifies CHUNKS of 64 points against an extent, so a chunk should be a compact patch:
  // bands of 8 rows (40 degrees), each walked along the azimuth.  (The order of the points only changes the
  // order of the float additions; libear's is row by row.)
  std::stable_sort(grid.begin(), grid.end(), [](const GridPoint &a, const GridPoint &b) {
    if (a.band != b.band) return a.band < b.band;
    if (a.az != b.az) return a.az < b.az;
    return a.row < b.row;
  });
  const size_t np = grid.size(), padded = (np + 63) / 64 * 64, n_chunks = padded / 64;
  require(n_chunks <= (size_t)kExtentMaxChunks, "extent grid too large");
  std::vector<double> xyz(3 * np);
  for (size_t q = 0; q < np; q++) xyz[3 * q] = grid[q].v.x, xyz[3 * q + 1] = grid[q].v.y, xyz[3 * q + 2] = grid[q].v.z;
  const size_t S = pn.P.stereo ? 2 : (size_t)pn.P.table.n_real;
  DevBuf<double> d_xyz, d_pv;
  d_xyz.alloc(3 * np);
  d_pv.alloc(np * S);
  EARHIP_HIP(hipMemcpyAsync(d_xyz.p, xyz.data(), sizeof(double) * 3 * np, hipMemcpyHostToDevice, ctx->stream));
  EARHIP_HIP(hipMemsetAsync(pn.missed.p, 0, sizeof(unsigned), ctx->stream));
  hipLaunchKernelGGL(k_pan_points, dim3((unsigned)((np + 127) / 128)), dim3(128), 0, ctx->stream, pn.P, np, d_xyz.p, d_pv.p,
                     pn.missed.p);
  EARHIP_HIP(hipGetLastError());
  std::vector<double> G(np * S);
  unsigned missed = 0;
  EARHIP_HIP(hipMemcpyAsync(G.data(), d_pv.p, sizeof(double) * np * S, hipMemcpyDeviceToHost, ctx->stream));
  EARHIP_HIP(hipMemcpyAsync(&missed, pn.missed.p, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  EARHIP_HIP(hipStreamSynchronize(ctx->stream));
  if (missed) fail_internal("point source panner: a point of the extent grid was not handled by any region");
  std::vector<float> pos(3 * padded), gains(S * padded, 0.0f);
  for (size_t q = 0; q < padded; q++) {
    const size_t src = std::min(q, np - 1);  // (padding repeats the last point; its gains stay zero)
    for (int k = 0; k < 3; k++) pos[k * padded + q] = (float)xyz[3 * src + k];
  }
  for (size_t q = 0; q < np; q++)
    for (size_t c = 0; c < S; c++) gains[c * padded + q] = (float)G[q * S + c];
  // per chunk: mean direction, angular radius around it, float sum of the gain vectors
  const size_t MC = (size_t)kExtentMaxChunks;
  std::vector<float> chunks(3 * MC + MC + MC * kMaxPanOut, 0.0f);
  float *dir = chunks.data(), *rad = dir + 3 * MC, *sum = rad + MC;
  for (size_t c = 0; c < n_chunks; c++) {
    const size_t lo = 64 * c, hi = std::min(lo + 64, np);
    Vec3 m = {0.0, 0.0, 0.0};
    for (size_t q = lo; q < hi; q++) m = vadd(m, grid[q].v);
    double n = std::sqrt(vdot(m, m));
    if (n < 1e-6) m = grid[lo].v, n = 1.0;
    // (the kernel works with the float direction: measure the radius around that)
    const float fx = (float)(m.x / n), fy = (float)(m.y / n), fz = (float)(m.z / n);
    const double fn = std::sqrt((double)fx * fx + (double)fy * fy + (double)fz * fz);
    double worst = 0.0;
    for (size_t q = lo; q < hi; q++) {
      const double cs = (grid[q].v.x * fx + grid[q].v.y * fy + grid[q].v.z * fz) / fn;
      worst = std::max(worst, std::acos(std::min(1.0, std::max(-1.0, cs))));
    }
    dir[c] = fx, dir[MC + c] = fy, dir[2 * MC + c] = fz;
    rad[c] = (float)(worst + 1e-4);
    for (size_t q = lo; q < hi; q++)
      for (size_t k = 0; k < S; k++) sum[c * kMaxPanOut + k] += gains[k * padded + q];
  }
  pn.ext_pos.alloc(3 * padded);
  pn.ext_gains.alloc(S * padded);
  pn.ext_chunks.alloc(chunks.size());
  EARHIP_HIP(hipMemcpy(pn.ext_pos.p, pos.data(), sizeof(float) * pos.size(), hipMemcpyHostToDevice));
  EARHIP_HIP(hipMemcpy(pn.ext_gains.p, gains.data(), sizeof(float) * gains.size(), hipMemcpyHostToDevice));
  EARHIP_HIP(hipMemcpy(pn.ext_chunks.p, chunks.data(), sizeof(float) * chunks.size(), hipMemcpyHostToDevice));
  pn.E.n_points = (int)np;
  pn.E.n_padded = (int)padded;
  pn.E.n_chunks = (int)n_chunks;
  pn.E.n_pv = (int)S;
  pn.E.xs = pn.ext_pos.p;
  pn.E.ys = pn.ext_pos.p + padded;
  pn.E.zs = pn.ext_pos.p + 2 * padded;
  pn.E.gains = pn.ext_gains.p;
  pn.E.chunk_dir = pn.ext_chunks.p;
  pn.E.chunk_rad = pn.ext_chunks.p + 3 * MC;
  pn.E.chunk_sum = pn.ext_chunks.p + 4 * MC;
}

// with_extent = false: the point source panner alone (the HOA design only pans points: no extent grid to pan,
// upload and wait for on every decode matrix)
static int panner_create(earhip_ctx *ctx, const char *layout, int n_channels, const double *azimuth,
                         const double *elevation, earhip_panner **out, bool with_extent) {
  return guarded([&] {
    require(ctx != nullptr && out != nullptr, "NULL argument");
    const LayoutEntry &L = layout_entry(layout);
    require((azimuth == nullptr) == (elevation == nullptr), "azimuth and elevation come together");
    if (azimuth) {
      require(n_channels == L.n, "one position per channel of the layout (LFE channels included)");
      for (int c = 0; c < L.n; c++)
        require(std::isfinite(azimuth[c]) && std::isfinite(elevation[c]), "loudspeaker positions must be finite");
    }
    ctx->use();
    std::unique_ptr<earhip_panner> p(new earhip_panner);
    p->ctx = ctx;
    std::memset(&p->P, 0, sizeof(p->P));
    p->P.n_full = L.n;
    int n_real = 0;
    std::vector<PanRegion> regions;
    if (std::strcmp(L.name, "0+2+0") == 0) {  // configureStereoPolarPanner (:406-429): pans on the nominal 0+5+0
      regions = build_regions(layout_entry("0+5+0"), n_real);
      p->P.stereo = 1;
      for (int c = 0; c < L.n; c++) {
        if (std::strcmp(L.channels[c].name, "M+030") == 0) p->P.stereo_index[0] = c;
        if (std::strcmp(L.channels[c].name, "M-030") == 0) p->P.stereo_index[1] = c;
      }
    } else {
      regions = build_regions(L, n_real, azimuth, elevation);
      int k = 0;
      for (int c = 0; c < L.n; c++)
        if (!L.channels[c].is_lfe) p->P.full_index[k++] = c;
    }
    require(L.n <= kMaxPanOut, "too many channels");
    for (int c = 0; c < kMaxPanOut; c++) p->P.real_of_full[c] = -1;
    const int n_pv = p->P.stereo ? 2 : n_real;
    for (int k = 0; k < n_pv; k++) p->P.real_of_full[p->P.stereo ? p->P.stereo_index[k] : p->P.full_index[k]] = k;
    p->regions.alloc(regions.size());
    EARHIP_HIP(hipMemcpy(p->regions.p, regions.data(), sizeof(PanRegion) * regions.size(), hipMemcpyHostToDevice));
    p->P.table.n_regions = (int)regions.size();
    p->P.table.n_real = n_real;
    p->P.table.regions = p->regions.p;
    p->missed.alloc_zero(1, ctx->stream);
    p->p_missed.reserve(1);
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    if (with_extent) build_extent_table(*p);
    *out = p.release();
  });
}

extern "C" {

int earhip_panner_create(earhip_ctx *ctx, const char *layout, earhip_panner **out) {
  return panner_create(ctx, layout, 0, nullptr, nullptr, out, true);
}

int earhip_panner_create_positions(earhip_ctx *ctx, const char *layout, int n_channels, const double *azimuth,
                                   const double *elevation, earhip_panner **out) {
  return panner_create(ctx, layout, n_channels, azimuth, elevation, out, true);
}

int earhip_panner_destroy(earhip_panner *p) {
  return guarded([&] {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    delete p;
  });
}

int earhip_panner_num_channels(const earhip_panner *p, int *n_channels) {
  return guarded([&] {
    require(p != nullptr && n_channels != nullptr, "NULL argument");
    *n_channels = p->P.n_full;
  });
}

// GainCalculatorHOAImpl (src/hoa/gain_calculator_hoa.cpp:8-72): AllRAD — the point source panner sampled
// at the 5200 directions of a spherical t-design (device, one thread per direction), times the N3D
// spherical harmonics at those directions (host, double: a cold 24 x 5200 x n_coef product), power
// normalised and converted to the requested normalisation.
int earhip_hoa_decode_matrix(earhip_ctx *ctx, const char *layout, int n_coef, const int *orders, const int *degrees,
                             const char *normalization, float *out) {
  return earhip_hoa_decode_matrix_positions(ctx, layout, 0, nullptr, nullptr, n_coef, orders, degrees, normalization, out);
}

int earhip_hoa_decode_matrix_positions(earhip_ctx *ctx, const char *layout, int n_channels, const double *azimuth,
                                       const double *elevation, int n_coef, const int *orders, const int *degrees,
                                       const char *normalization, float *out) {
  return guarded([&] {
    require(ctx != nullptr && out != nullptr, "NULL argument");
    require(n_coef >= 1 && orders != nullptr && degrees != nullptr, "orders and degrees must be the same size");
    require(n_coef <= 4096, "too many coefficients");
    for (int i = 0; i < n_coef; i++) {
      require(orders[i] >= 0, "orders must not be negative");
      require(std::abs(degrees[i]) <= orders[i], "magnitude of degree must not be greater than order");
      require(orders[i] <= 64, "order too high");
    }
    require(normalization != nullptr, "normalization must not be NULL");
    HoaNorm norm;
    if (std::strcmp(normalization, "N3D") == 0) norm = kN3D;
    else if (std::strcmp(normalization, "SN3D") == 0) norm = kSN3D;
    else if (std::strcmp(normalization, "FuMa") == 0) norm = kFuMa;
    else throw Error{EARHIP_ADM_ERROR, std::string("ADM error: unknown normalization type: '") + normalization + "'"};
    earhip_panner *pn = nullptr;
    const int st = panner_create(ctx, layout, n_channels, azimuth, elevation, &pn, false);
    if (st != EARHIP_OK) throw Error{st, earhip_last_error()};
    std::unique_ptr<earhip_panner, int (*)(earhip_panner *)> guard(pn, earhip_panner_destroy);
    const size_t P = (size_t)kTDesignPoints, C = (size_t)n_coef;
    const size_t S = pn->P.stereo ? 2 : (size_t)pn->P.table.n_real;
    // the design's directions (src/hoa/hoa.cpp:4-14)
    std::vector<double> xyz(3 * P);
    for (size_t i = 0; i < P; i++) {
      const double phi = kTDesign[i][0], theta = kTDesign[i][1];
      xyz[3 * i] = std::sin(theta) * std::cos(phi);
      xyz[3 * i + 1] = std::sin(theta) * std::sin(phi);
      xyz[3 * i + 2] = std::cos(theta);
    }
    // G_virt [P][S] on the device
    DevBuf<double> d_xyz, d_pv;
    d_xyz.alloc(3 * P);
    d_pv.alloc(P * S);
    EARHIP_HIP(hipMemcpyAsync(d_xyz.p, xyz.data(), sizeof(double) * 3 * P, hipMemcpyHostToDevice, ctx->stream));
    EARHIP_HIP(hipMemsetAsync(pn->missed.p, 0, sizeof(unsigned), ctx->stream));
    hipLaunchKernelGGL(k_pan_points, dim3((unsigned)((P + 127) / 128)), dim3(128), 0, ctx->stream, pn->P, P, d_xyz.p, d_pv.p,
                       pn->missed.p);
    EARHIP_HIP(hipGetLastError());
    std::vector<double> G(P * S);
    unsigned missed = 0;
    EARHIP_HIP(hipMemcpyAsync(G.data(), d_pv.p, sizeof(double) * P * S, hipMemcpyDeviceToHost, ctx->stream));
    EARHIP_HIP(hipMemcpyAsync(&missed, pn->missed.p, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    if (missed) fail_internal("point source panner: a design direction was not handled by any region");
    // Y_virt [C][P], N3D (hoa.hpp:115-133)
    std::vector<double> Y(C * P);
    for (size_t i = 0; i < P; i++) {
      const double az = -std::atan2(xyz[3 * i], xyz[3 * i + 1]);
      const double el = std::atan2(xyz[3 * i + 2], std::hypot(xyz[3 * i], xyz[3 * i + 1]));
      for (size_t c = 0; c < C; c++) {
        const int n = orders[c], m = degrees[c], am = std::abs(m);
        const double trig = m > 0 ? std::sqrt(2.0) * std::cos(m * az) : m < 0 ? -std::sqrt(2.0) * std::sin(m * az) : 1.0;
        Y[c * P + i] = hoa_norm(kN3D, n, am) * legendre_nm(n, am, std::sin(el)) * trig;
      }
    }
    // D = G_virt (Y_virt^T / P), then |D Y_virt|_F = sqrt(P), then N3D -> norm (gain_calculator_hoa.cpp:56-63)
    std::vector<double> D(S * C, 0.0);
    for (size_t sp = 0; sp < S; sp++)
      for (size_t c = 0; c < C; c++) {
        double acc = 0.0;
        for (size_t i = 0; i < P; i++) acc += G[i * S + sp] * (Y[c * P + i] / (double)P);
        D[sp * C + c] = acc;
      }
    double fro = 0.0;
    for (size_t sp = 0; sp < S; sp++)
      for (size_t i = 0; i < P; i++) {
        double v = 0.0;
        for (size_t c = 0; c < C; c++) v += D[sp * C + c] * Y[c * P + i];
        fro += v * v;
      }
    const double k = std::sqrt((double)P) / std::sqrt(fro);
    const int n_full = pn->P.n_full;
    for (size_t i = 0; i < (size_t)n_full * C; i++) out[i] = 0.0f;
    for (size_t sp = 0; sp < S; sp++) {
      const int row = pn->P.stereo ? pn->P.stereo_index[sp] : pn->P.full_index[sp];
      for (size_t c = 0; c < C; c++) {
        const int am = std::abs(degrees[c]);
        const double conv = hoa_norm(kN3D, orders[c], am) / hoa_norm(norm, orders[c], am);
        out[(size_t)row * C + c] = (float)(D[sp * C + c] * k * conv);
      }
    }
  });
}

constexpr size_t kExtentBatch = (size_t)1 << 18;  // positions per pair of launches (bounds the job buffer: 115 MB)

static void launch_pan(earhip_panner *p, size_t npos, const double *az, const double *el, const double *dist,
                       const double *width, const double *height, const double *depth, const double *gain,
                       const double *diffuse, float *direct, float *diffuse_out) {
  hipStream_t s = p->ctx->stream;
  // a distance of 1 and no extent: nothing is spread (extentMod(0, 1) = 0)
  const bool may_spread = dist || width || height || depth;
  if (!may_spread) {
    hipLaunchKernelGGL(k_pan_objects, dim3((unsigned)((npos + 127) / 128)), dim3(128), 0, s, p->P, npos, az, el, dist, width,
                       height, depth, gain, diffuse, direct, diffuse_out, p->missed.p, (ExtentJob *)nullptr,
                       (unsigned *)nullptr);
    EARHIP_HIP(hipGetLastError());
    return;
  }
  p->jobs.reserve(std::min(npos, kExtentBatch));
  p->job_count.reserve(1);
  auto at = [](const double *q, size_t o) { return q ? q + o : q; };
  for (size_t o = 0; o < npos; o += kExtentBatch) {
    const size_t n = std::min(kExtentBatch, npos - o);
    const size_t N = (size_t)p->P.n_full;
    EARHIP_HIP(hipMemsetAsync(p->job_count.p, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_pan_objects, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, p->P, n, az + o, el + o, at(dist, o),
                       at(width, o), at(height, o), at(depth, o), at(gain, o), at(diffuse, o), direct + o * N,
                       diffuse_out + o * N, p->missed.p, p->jobs.p, p->job_count.p);
    EARHIP_HIP(hipGetLastError());
    // (launched for the worst case — every position a job; surplus waves leave at once)
    hipLaunchKernelGGL(k_pan_objects_extent, dim3((unsigned)((n + kExtentWaves - 1) / kExtentWaves)), dim3(64 * kExtentWaves),
                       0, s, p->P, p->E, p->jobs.p, p->job_count.p, at(gain, o), at(diffuse, o), direct + o * N,
                       diffuse_out + o * N);
    EARHIP_HIP(hipGetLastError());
  }
}

int earhip_panner_calculate_extent_device(earhip_panner *p, size_t npos, const double *azimuth, const double *elevation,
                                          const double *distance, const double *width, const double *height,
                                          const double *depth, const double *gain, const double *diffuse, float *direct,
                                          float *diffuse_out) {
  return guarded([&] {
    require(p != nullptr, "panner must not be NULL");
    require(azimuth && elevation && direct && diffuse_out, "azimuth, elevation and the outputs must not be NULL");
    require(npos < ((size_t)1 << 28), "too many positions");
    p->ctx->use();
    // the counter of positions no region takes means "this call" (earhip_panner_missed reads it) — an empty call too
    EARHIP_HIP(hipMemsetAsync(p->missed.p, 0, sizeof(unsigned), p->ctx->stream));
    if (npos == 0) return;
    launch_pan(p, npos, azimuth, elevation, distance, width, height, depth, gain, diffuse, direct, diffuse_out);
  });
}

int earhip_panner_missed(earhip_panner *p, unsigned *count) {
  return guarded([&] {
    require(p != nullptr && count != nullptr, "NULL argument");
    earhip_ctx *ctx = p->ctx;
    ctx->use();
    EARHIP_HIP(hipMemcpyAsync(p->p_missed.p, p->missed.p, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    *count = *p->p_missed.p;
  });
}

int earhip_panner_calculate_device(earhip_panner *p, size_t npos, const double *azimuth, const double *elevation,
                                   const double *distance, const double *gain, const double *diffuse, float *direct,
                                   float *diffuse_out) {
  return earhip_panner_calculate_extent_device(p, npos, azimuth, elevation, distance, nullptr, nullptr, nullptr, gain, diffuse,
                                               direct, diffuse_out);
}

int earhip_panner_calculate_extent(earhip_panner *p, size_t npos, const double *azimuth, const double *elevation,
                                   const double *distance, const double *width, const double *height, const double *depth,
                                   const double *gain, const double *diffuse, float *direct, float *diffuse_out) {
  return guarded([&] {
    require(p != nullptr, "panner must not be NULL");
    require(azimuth && elevation && direct && diffuse_out, "azimuth, elevation and the outputs must not be NULL");
    require(npos < ((size_t)1 << 28), "too many positions");
    if (npos == 0) return;
    earhip_ctx *ctx = p->ctx;
    ctx->use();
    const size_t N = (size_t)p->P.n_full;
    constexpr int kIn = 8;
    p->p_in.reserve(kIn * npos);
    p->d_in.reserve(kIn * npos);
    p->p_out.reserve(2 * npos * N);
    p->d_out.reserve(2 * npos * N);
    const double *src[kIn] = {azimuth, elevation, distance, width, height, depth, gain, diffuse};
    const double *dev[kIn];
    for (int k = 0; k < kIn; k++) {
      if (src[k]) std::memcpy(p->p_in.p + k * npos, src[k], sizeof(double) * npos);
      dev[k] = src[k] ? p->d_in.p + k * npos : nullptr;
    }
    EARHIP_HIP(hipMemcpyAsync(p->d_in.p, p->p_in.p, sizeof(double) * kIn * npos, hipMemcpyHostToDevice, ctx->stream));
    EARHIP_HIP(hipMemsetAsync(p->missed.p, 0, sizeof(unsigned), ctx->stream));
    launch_pan(p, npos, dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], dev[6], dev[7], p->d_out.p, p->d_out.p + npos * N);
    EARHIP_HIP(hipMemcpyAsync(p->p_out.p, p->d_out.p, sizeof(float) * 2 * npos * N, hipMemcpyDeviceToHost, ctx->stream));
    EARHIP_HIP(hipMemcpyAsync(p->p_missed.p, p->missed.p, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(direct, p->p_out.p, sizeof(float) * npos * N);
    std::memcpy(diffuse_out, p->p_out.p + npos * N, sizeof(float) * npos * N);
    // (the reference dereferences an empty optional when no region takes a position: undefined there,
    // an error here)
    if (*p->p_missed.p != 0) fail_internal("point source panner: a position was not handled by any region");
  });
}

int earhip_panner_calculate(earhip_panner *p, size_t npos, const double *azimuth, const double *elevation,
                            const double *distance, const double *gain, const double *diffuse, float *direct,
                            float *diffuse_out) {
  return earhip_panner_calculate_extent(p, npos, azimuth, elevation, distance, nullptr, nullptr, nullptr, gain, diffuse, direct,
                                        diffuse_out);
}

}  // extern "C"
