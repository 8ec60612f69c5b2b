// extent_oracle.hpp — TEST INFRASTRUCTURE (the checker, never the product): a plain C++14 restatement, without
// Eigen / Boost / xsimd, of libear's polar extent panner — the part of GainCalculatorObjects that handles
// width / height / depth:
//
//   PolarExtent (set-up, handle, calc_pv_spread,     src/object_based/polar_extent.cpp:12-302
//     setup_weighting_function, setup_angle_to_weight,
//     extentMod, calcBasis, the 37-row point grid)
//   PolarExtentCoreScalar::run and its weights        src/object_based/polar_extent_scalar.cpp:25-108
//     (extent_float_t = float: polar_extent_core.hpp:9)
//   localCoordinateSystem, azimuth, elevation         src/common/geom.hpp:91-98, src/common/geom.cpp:71-78
//   interp                                            src/common/helpers/eigen_helpers.hpp:26-48
//
// and, as a SECOND, independent statement used to pin the first, the implementation libear's own tests keep
// beside the library ("reference::PolarExtentPanner", double precision, azimuth / elevation form of the
// weighting function): tests/reference/extent.cpp:21-249.  libear's test `same_as_reference`
// (tests/extent_tests.cpp:140-169) holds the two to 1e-5 (Eigen isApprox) on 1000 random cases; tests/
// test_oracle_extent.py does the same with these restatements, and restates test_basis, test_weight_func and
// test_pv (tests/extent_tests.cpp:16-138).
#pragma once

#include <cmath>
#include <memory>
#include <vector>

#include "panner_oracle.hpp"

namespace extent_oracle {

using panner_oracle::cart;
using panner_oracle::kPi;
using panner_oracle::Opt;
using panner_oracle::PointSourcePanner;
using panner_oracle::radians;
using panner_oracle::V3;
using panner_oracle::Vec;

inline double degrees(double r) { return r * 180.0 / kPi; }

// eigen_helpers.hpp:26-48: piecewise-linear, clamped at both ends
inline double interp(double x, std::initializer_list<double> xp_, std::initializer_list<double> yp_) {
  const std::vector<double> xp(xp_), yp(yp_);
  if (x <= xp[0]) return yp[0];
  for (size_t i = 0; i + 1 < xp.size(); i++)
    if (xp[i + 1] > x) return yp[i] + (yp[i + 1] - yp[i]) / (xp[i + 1] - xp[i]) * (x - xp[i]);
  return yp.back();
}

// geom.cpp:71-78
inline double azimuth_of(V3 p) { return -degrees(std::atan2(p.x, p.y)); }
inline double elevation_of(V3 p) { return degrees(std::atan2(p.z, std::hypot(p.x, p.y))); }

struct Mat3 {
  double m[3][3];  // rows
};
// geom.hpp:91-98: rows pointing along x, y, z of a frame whose +y points at cart(az, el, 1)
inline Mat3 local_coordinate_system(double az, double el) {
  const V3 r[3] = {cart(az - 90.0, 0.0, 1.0), cart(az, el, 1.0), cart(az, el + 90.0, 1.0)};
  Mat3 out;
  for (int i = 0; i < 3; i++) out.m[i][0] = r[i].x, out.m[i][1] = r[i].y, out.m[i][2] = r[i].z;
  return out;
}
// polar_extent.cpp:52-60
inline V3 safe_norm_position(V3 p) {
  const double n = std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z);
  if (n < 1e-10) return {0.0, 1.0, 0.0};
  return {p.x / n, p.y / n, p.z / n};
}
// polar_extent.cpp:80-91 (and tests/reference/extent.cpp:42-52)
inline Mat3 calc_basis(V3 position) {
  position = safe_norm_position(position);
  double az = azimuth_of(position);
  const double el = elevation_of(position);
  if (std::fabs(el) > 90.0 - 1e-5) az = 0.0;  // near the poles the azimuth is indeterminate
  return local_coordinate_system(az, el);
}
// polar_extent.cpp:62-70
inline double extent_mod(double extent, double distance) {
  const double min_size = 0.2;
  const double size = interp(extent, {0.0, 360.0}, {min_size, 1.0});
  const double extent1 = 4.0 * degrees(std::atan2(size, 1.0));
  return interp(4.0 * degrees(std::atan2(size, distance)), {0.0, extent1, 360.0}, {0.0, extent, 360.0});
}

// polar_extent.cpp:16-41 (same in tests/reference/extent.cpp:171-195): rows of points every 5 degrees of
// elevation, as many per row as keep the spacing along the row about the same
inline std::vector<V3> panning_positions_even(int n_rows) {
  std::vector<V3> out;
  for (int r = 0; r < n_rows; r++) {
    const double el = -90.0 + r * (180.0 / (n_rows - 1));
    const double radius = std::cos(radians(el));
    const double perimeter = 2 * kPi * radius, perimeter_centre = 2 * kPi;
    int n_points = (int)std::round((perimeter / perimeter_centre) * 2 * (n_rows - 1));
    if (n_points == 0) n_points = 1;
    for (int i = 0; i < n_points; i++) out.push_back(cart(i * (360.0 / n_points), el, 1.0));
  }
  return out;
}

constexpr double kFadeWidth = 10.0;  // polar_extent.cpp:13
constexpr int kRows = 37;            // :14

// ---- the library's implementation: PolarExtent + the scalar core ----------------------------------
struct PolarExtent {
  typedef float ef;  // extent_float_t (polar_extent_core.hpp:9)
  std::shared_ptr<PointSourcePanner> psp;
  size_t num_points = 0, num_speakers = 0;
  std::vector<ef> xs, ys, zs;
  std::vector<ef> point_gains;  // [point][speaker] ("summed_panning_results" at a batch size of one, :157-159)
  // PolarExtentCoreContext (polar_extent_core.hpp:12-45), the per-call part
  mutable bool is_circular = false;
  mutable ef basis[9], circle_test[2], right_circle_centre[2];
  mutable ef cos_start, cos_end, sin_start, sin_end, m, c;
  mutable std::vector<ef> results;

  explicit PolarExtent(std::shared_ptr<PointSourcePanner> p) : psp(std::move(p)) {  // :96-176
    const std::vector<V3> pos = panning_positions_even(kRows);
    num_points = pos.size();
    num_speakers = (size_t)psp->n_out();
    xs.resize(num_points), ys.resize(num_points), zs.resize(num_points);
    point_gains.assign(num_points * num_speakers, 0.0f);
    results.assign(num_speakers, 0.0f);
    for (size_t i = 0; i < num_points; i++) {
      xs[i] = (ef)pos[i].x, ys[i] = (ef)pos[i].y, zs[i] = (ef)pos[i].z;
      const Opt pv = psp->handle(pos[i]);
      if (!pv.ok) throw std::runtime_error("extent grid point not handled by the point source panner");
      for (size_t s = 0; s < num_speakers; s++) point_gains[i * num_speakers + s] += (ef)pv.v[s];
    }
  }

  // :186-211
  void setup_angle_to_weight(double start_angle, double end_angle) const {
    cos_start = (ef)(start_angle < kPi ? std::cos(start_angle) : -1.0);
    cos_end = (ef)(end_angle < kPi ? std::cos(end_angle) : -(1.0 + 1e-6));
    sin_start = (ef)(start_angle < kPi / 2 ? std::sin(start_angle) : 1.0);
    sin_end = (ef)(end_angle < kPi / 2 ? std::sin(end_angle) : 1.0 + 1e-6);
    m = (ef)(1.0 / (start_angle - end_angle));
    c = (ef)(-m * end_angle);  // (float m, double angle: a double product)
  }
  // :213-255
  void setup_weighting_function(V3 position, double width, double height) const {
    width = radians(width) / 2;
    height = radians(height) / 2;
    Mat3 b = calc_basis(position);
    if (height > width) {  // always wider than high from here on: rotate the frame
      std::swap(height, width);
      Mat3 f;
      for (int j = 0; j < 3; j++) f.m[0][j] = b.m[2][j], f.m[1][j] = b.m[1][j], f.m[2][j] = -b.m[0][j];
      b = f;
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) basis[i * 3 + j] = (ef)b.m[i][j];
    const double width_full = kPi + height;  // make the ends meet at the back
    const double width_mod = interp(width, {0.0, kPi / 2.0, kPi}, {0.0, kPi / 2.0, width_full});
    width = interp(height, {0.0, kPi / 4.0, kPi / 2.0, kPi}, {width_mod, width_mod, width, width});
    is_circular = (width - height) < 1e-6;
    const double circle_pos = width - height;
    right_circle_centre[0] = (ef)std::sin(circle_pos);
    right_circle_centre[1] = (ef)std::cos(circle_pos);
    circle_test[0] = (ef)-std::cos(circle_pos);
    circle_test[1] = (ef)std::sin(circle_pos);
    setup_angle_to_weight(height, height + radians(kFadeWidth));
  }

  // polar_extent_scalar.cpp:34-76, float arithmetic throughout
  ef weight_from_cos(ef cos_angle) const {
    if (cos_angle >= cos_start) return 1.0f;
    if (cos_angle <= cos_end) return 0.0f;
    return m * std::acos(cos_angle) + c;
  }
  ef weight_from_sin(ef sin_angle) const {
    if (sin_angle <= sin_start) return 1.0f;
    if (sin_angle >= sin_end) return 0.0f;
    return m * std::asin(sin_angle) + c;
  }
  static ef dot3(ef x, ef y, ef z, const ef *v) { return x * v[0] + y * v[1] + z * v[2]; }
  ef weight(ef x, ef y, ef z) const {
    if (is_circular) return weight_from_cos(dot3(x, y, z, basis + 3));
    const ef tx = dot3(x, y, z, basis), ty = dot3(x, y, z, basis + 3), tz = dot3(x, y, z, basis + 6);
    const ef rx = std::fabs(tx);
    if (rx * circle_test[0] + ty * circle_test[1] >= 0.0f) return weight_from_sin(std::fabs(tz));
    return weight_from_cos(rx * right_circle_centre[0] + ty * right_circle_centre[1]);
  }
  // :78-108: weighted sum of the points' gains, in point order
  void run_core() const {
    for (auto &r : results) r = 0.0f;
    for (size_t i = 0; i < num_points; i++) {
      const ef w = weight(xs[i], ys[i], zs[i]);
      const ef *g = &point_gains[i * num_speakers];
      if (w == 1.0f) {
        for (size_t s = 0; s < num_speakers; s++) results[s] += g[s];
      } else if (w != 0.0f) {
        for (size_t s = 0; s < num_speakers; s++) results[s] += w * g[s];
      }
    }
  }

  // polar_extent.cpp:257-288; false: the point source panner took no region for `position`
  bool calc_pv_spread(V3 position, double width, double height, Vec &out) const {
    const double amount_spread = interp(std::max(width, height), {0.0, kFadeWidth}, {0.0, 1.0});
    const double amount_point = 1.0 - amount_spread;
    out.assign(num_speakers, 0.0);
    if (amount_point > 1e-10) {
      const Opt pv = psp->handle(position);
      if (!pv.ok) return false;
      for (size_t s = 0; s < num_speakers; s++) out[s] += amount_point * (pv.v[s] * pv.v[s]);
    }
    if (amount_spread > 1e-10) {
      width = std::max(width, kFadeWidth / 2.0);
      height = std::max(height, kFadeWidth / 2.0);
      setup_weighting_function(position, width, height);
      run_core();
      ef n2 = 0.0f;
      for (ef r : results) n2 += r * r;
      const ef scale = (ef)(1.0 / std::sqrt(n2));  // (float norm, double reciprocal, float scaling: :281)
      for (size_t s = 0; s < num_speakers; s++) {
        const ef r = results[s] * scale;
        out[s] += amount_spread * (double)(r * r);
      }
    }
    for (double &v : out) v = std::sqrt(v);
    return true;
  }
  // :290-302
  bool handle(V3 position, double width, double height, double depth, Vec &out) const {
    const double distance = std::sqrt(position.x * position.x + position.y * position.y + position.z * position.z);
    if (depth != 0.0) {
      double dmin = distance - depth / 2.0, dmax = distance + depth / 2.0;
      dmin = dmin < 0 ? 0.0 : dmin;
      dmax = dmax < 0 ? 0.0 : dmax;
      Vec a, b;
      if (!calc_pv_spread(position, extent_mod(width, dmin), extent_mod(height, dmin), a)) return false;
      if (!calc_pv_spread(position, extent_mod(width, dmax), extent_mod(height, dmax), b)) return false;
      out.resize(num_speakers);
      for (size_t s = 0; s < num_speakers; s++) out[s] = std::sqrt((a[s] * a[s] + b[s] * b[s]) / 2.0);
      return true;
    }
    return calc_pv_spread(position, extent_mod(width, distance), extent_mod(height, distance), out);
  }
};

// ---- the implementation libear's tests keep beside the library (tests/reference/extent.cpp) -------
namespace test_reference {

// :54-61, :63-74
inline V3 cart_on_basis(const Mat3 &b, double az, double el) {
  const double r[3] = {std::sin(az) * std::cos(el), std::cos(az) * std::cos(el), std::sin(el)};
  return {r[0] * b.m[0][0] + r[1] * b.m[1][0] + r[2] * b.m[2][0], r[0] * b.m[0][1] + r[1] * b.m[1][1] + r[2] * b.m[2][1],
          r[0] * b.m[0][2] + r[1] * b.m[1][2] + r[2] * b.m[2][2]};
}
inline void azimuth_elevation_on_basis(const Mat3 &b, V3 p, double &az, double &el) {
  double comp[3];
  for (int i = 0; i < 3; i++) {
    comp[i] = p.x * b.m[i][0] + p.y * b.m[i][1] + p.z * b.m[i][2];
    comp[i] = std::max(std::min(comp[i], 1.0), -1.0);
  }
  az = std::atan2(comp[0], comp[1]);
  el = std::asin(comp[2]);
}

// :76-143
struct WeightingFunction {
  double width, height, circle_radius, circle_pos;
  Mat3 basis;
  V3 circle[2];
  WeightingFunction(V3 position, double width_deg, double height_deg) {
    width = radians(width_deg) / 2;
    height = radians(height_deg) / 2;
    const Mat3 b = calc_basis(position);
    circle_radius = std::min(width, height);
    if (height > width) {
      std::swap(height, width);
      for (int j = 0; j < 3; j++) basis.m[0][j] = b.m[2][j], basis.m[1][j] = b.m[1][j], basis.m[2][j] = b.m[0][j];
    } else {
      basis = b;
    }
    const double width_full = kPi + height;
    const double width_mod = interp(width, {0.0, kPi / 2.0, kPi}, {0.0, kPi / 2.0, width_full});
    width = interp(height, {0.0, kPi / 4.0, kPi / 2.0, kPi}, {width_mod, width_mod, width, width});
    circle_pos = width - circle_radius;
    circle[0] = cart_on_basis(basis, -circle_pos, 0.0);
    circle[1] = cart_on_basis(basis, circle_pos, 0.0);
  }
  double operator()(V3 p) const {
    double az, el;
    azimuth_elevation_on_basis(basis, p, az, el);
    double distance;
    if (std::fabs(az) <= circle_pos) {
      distance = std::fabs(el) - circle_radius;
    } else {
      const V3 cc = circle[az < 0 ? 0 : 1];
      const double angle = p.x * cc.x + p.y * cc.y + p.z * cc.z;
      distance = std::acos(std::max(std::min(angle, 1.0), -1.0)) - circle_radius;
    }
    return interp(distance, {0.0, radians(kFadeWidth)}, {1.0, 0.0});
  }
};

// :145-249 (SpreadingPanner + PolarExtentPanner), double precision
struct PolarExtentPanner {
  std::shared_ptr<PointSourcePanner> psp;
  std::vector<V3> positions;
  std::vector<Vec> position_gains;
  explicit PolarExtentPanner(std::shared_ptr<PointSourcePanner> p) : psp(std::move(p)) {
    positions = panning_positions_even(kRows);
    for (const V3 &q : positions) position_gains.push_back(psp->handle(q).v);
  }
  Vec calc_pv_spread(V3 position, double width, double height) const {
    const double amount_spread = interp(std::max(width, height), {0.0, kFadeWidth}, {0.0, 1.0});
    const double amount_point = 1.0 - amount_spread;
    const size_t S = (size_t)psp->n_out();
    Vec pv(S, 0.0);
    if (amount_point > 1e-10) {
      const Vec g = psp->handle(position).v;
      for (size_t s = 0; s < S; s++) pv[s] += amount_point * g[s] * g[s];
    }
    if (amount_spread > 1e-10) {
      width = std::max(width, kFadeWidth / 2.0);
      height = std::max(height, kFadeWidth / 2.0);
      const WeightingFunction wf(position, width, height);
      Vec total(S, 0.0);
      for (size_t i = 0; i < positions.size(); i++) {
        const double w = wf(positions[i]);
        for (size_t s = 0; s < S; s++) total[s] += w * position_gains[i][s];
      }
      const double n = panner_oracle::norm(total);
      for (size_t s = 0; s < S; s++) pv[s] += amount_spread * (total[s] / n) * (total[s] / n);
    }
    for (double &v : pv) v = std::sqrt(v);
    return pv;
  }
  Vec handle(V3 position, double width, double height, double depth) const {
    const double distance = std::sqrt(position.x * position.x + position.y * position.y + position.z * position.z);
    if (depth != 0.0) {
      double dmin = distance - depth / 2.0, dmax = distance + depth / 2.0;
      dmin = dmin < 0 ? 0.0 : dmin;
      dmax = dmax < 0 ? 0.0 : dmax;
      const Vec a = calc_pv_spread(position, extent_mod(width, dmin), extent_mod(height, dmin));
      const Vec b = calc_pv_spread(position, extent_mod(width, dmax), extent_mod(height, dmax));
      Vec out(a.size());
      for (size_t s = 0; s < a.size(); s++) out[s] = std::sqrt((a[s] * a[s] + b[s] * b[s]) / 2.0);
      return out;
    }
    return calc_pv_spread(position, extent_mod(width, distance), extent_mod(height, distance));
  }
};

}  // namespace test_reference

// GainCalculatorObjectsImpl::calculate (src/object_based/gain_calculator_objects.cpp:33-57)
struct GainCalculatorObjects {
  panner_oracle::PannerSetup base;
  PolarExtent extent;
  explicit GainCalculatorObjects(const std::string &layout, const double *real_az = nullptr, const double *real_el = nullptr)
      : base(layout, real_az, real_el), extent(base.psp) {}
  bool calculate(double az, double el, double dist, double width, double height, double depth, double gain, double diffuse,
                 float *direct, float *diff) const {
    Vec pv;
    if (!extent.handle(cart(az, el, dist), width, height, depth, pv)) return false;
    size_t j = 0;
    for (size_t ch = 0; ch < base.is_lfe.size(); ch++) {
      double v = 0.0;
      if (!base.is_lfe[ch]) v = pv[j++] * gain;
      direct[ch] = (float)(v * std::sqrt(1.0 - diffuse));
      diff[ch] = (float)(v * std::sqrt(diffuse));
    }
    return true;
  }
};

}  // namespace extent_oracle
