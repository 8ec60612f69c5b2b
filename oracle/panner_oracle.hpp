// panner_oracle.hpp — TEST INFRASTRUCTURE (the checker, never the product): a plain C++14 restatement, in
// double precision and without Eigen/Boost, of libear's gain-vector producer for Objects content:
//
//   GainCalculatorObjectsImpl's constructor     src/object_based/gain_calculator_objects.cpp:24-31
//     (calculate() and PolarExtent: oracle/extent_oracle.hpp)
//   configurePolarPanner & friends              src/common/point_source_panner.cpp:14-600
//   cart, ngonVertexOrder                       src/common/geom.cpp:37-92
//
// following the reference class by class (RegionHandler -> Triplet / VirtualNgon / QuadRegion /
// StereoPannerDownmix, PolarPointSourcePanner, PointSourcePannerDownmix) so that each function can be read
// against the lines it cites.  Layout data: oracle/bs2051_data.h (generated from the reference's tables).
// Pinned by the reference's own tests (tests/point_source_panner_tests.cpp, tests/
// gain_calculator_objects_tests.cpp), restated in tests/test_oracle_panner.py.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "bs2051_data.h"

namespace panner_oracle {

using Vec = std::vector<double>;
struct V3 {
  double x, y, z;
};
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(const Vec &v) {
  double s = 0;
  for (double x : v) s += x * x;
  return std::sqrt(s);
}

const double kPi = 3.14159265358979323846264338327950288;
inline double radians(double d) { return d * kPi / 180.0; }

// geom.cpp:82-87
inline V3 cart(double az, double el, double dist) {
  return {std::sin(radians(-az)) * std::cos(radians(el)) * dist, std::cos(radians(-az)) * std::cos(radians(el)) * dist,
          std::sin(radians(el)) * dist};
}

struct Chan {
  std::string name;
  double az, el;          // real position
  double az_nom, el_nom;  // nominal position
};

// Layout::withoutLfe of getLayout(name) (src/bs2051.cpp:13-22, include/ear/layout.hpp)
// real_az / real_el (optional, one value per channel of the FULL layout): the loudspeakers' real positions
// (Channel::polarPosition, include/ear/layout.hpp:32-40); without them real = nominal, as getLayout returns
inline std::vector<Chan> layout_without_lfe(const std::string &name, std::vector<bool> *is_lfe_full = nullptr,
                                            const double *real_az = nullptr, const double *real_el = nullptr) {
  using namespace ear_oracle_data;
  for (int i = 0; i < kNumLayouts; i++)
    if (name == kLayouts[i].name) {
      std::vector<Chan> out;
      for (int c = 0; c < kLayouts[i].n; c++) {
        const LayoutChannel &ch = kLayouts[i].channels[c];
        if (is_lfe_full) is_lfe_full->push_back(ch.is_lfe);
        if (!ch.is_lfe)
          out.push_back({ch.name, real_az ? real_az[c] : ch.azimuth, real_el ? real_el[c] : ch.elevation, ch.azimuth, ch.elevation});
      }
      return out;
    }
  throw std::invalid_argument("unknown layout " + name);
}
struct not_implemented : std::runtime_error {  // ear::not_implemented (include/ear/exceptions.hpp)
  using std::runtime_error::runtime_error;
};
// point_source_panner.cpp:558-577
inline void check_screen_speakers(const std::vector<Chan> &layout) {
  for (auto &c : layout)
    if (c.name == "M+SC" || c.name == "M-SC") {
      const double abs_az = std::fabs(c.az);
      if (!((5.0 <= abs_az && abs_az < 25.0) || (35.0 <= abs_az && abs_az < 60.0)))
        throw std::invalid_argument("M+SC or M-SC has azimuth not in the allowed ranges of 5 to 25 and 35 to 60 degrees");
      if (25.0 < abs_az) throw not_implemented("M+SC and M-SC with azimuths wider than 25 degrees are not currently supported");
    }
}
inline const int *layout_facets(const std::string &name) {
  using namespace ear_oracle_data;
  for (int i = 0; i < kNumLayouts; i++)
    if (name == kLayouts[i].name) return kLayouts[i].facets;
  return nullptr;
}

// geom.cpp:37-70
inline std::vector<int> ngon_vertex_order(const std::vector<V3> &vertices) {
  V3 centre{0, 0, 0};
  for (auto &v : vertices) centre = centre + v;
  centre = centre * (1.0 / (double)vertices.size());
  V3 a = vertices[0] - centre, b{0, 0, 0};
  double mn = std::numeric_limits<double>::max();
  for (size_t i = 1; i < vertices.size(); i++) {
    V3 vertex = vertices[i] - centre;
    double angle = std::fabs(dot(vertex, a));
    if (angle < mn) {
      mn = angle;
      b = vertex;
    }
  }
  Vec ang(vertices.size());
  for (size_t i = 0; i < vertices.size(); i++) {
    V3 rel = vertices[i] - centre;
    ang[i] = std::atan2(dot(rel, a), dot(rel, b));
  }
  std::vector<int> idx(vertices.size());
  std::iota(idx.begin(), idx.end(), 0);
  std::sort(idx.begin(), idx.end(), [&](int i1, int i2) { return ang[i1] < ang[i2]; });
  return idx;
}

struct Opt {  // boost::optional<Eigen::VectorXd>
  bool ok = false;
  Vec v;
};

// point_source_panner.hpp:15-30, .cpp:16-34
struct RegionHandler {
  std::vector<int> channels;
  std::vector<V3> positions;
  RegionHandler(std::vector<int> ch, std::vector<V3> pos) : channels(std::move(ch)), positions(std::move(pos)) {}
  virtual ~RegionHandler() = default;
  virtual Opt handle(V3 position) const = 0;
  Opt handle_remap(V3 position, int n_channels) const {
    Opt pv = handle(position);
    if (!pv.ok) return pv;
    Opt out;
    out.ok = true;
    out.v.assign(n_channels, 0.0);
    for (size_t i = 0; i < pv.v.size(); i++) out.v[channels[i]] = pv.v[i];
    return out;
  }
};

// point_source_panner.cpp:36-51
struct Triplet : RegionHandler {
  double basis[3][3];  // inverse of the matrix whose ROWS are the loudspeaker positions
  Triplet(std::vector<int> ch, std::vector<V3> pos) : RegionHandler(std::move(ch), std::move(pos)) {
    const double m[3][3] = {{positions[0].x, positions[0].y, positions[0].z},
                            {positions[1].x, positions[1].y, positions[1].z},
                            {positions[2].x, positions[2].y, positions[2].z}};
    const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                       m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    const double id = 1.0 / det;
    basis[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) * id;
    basis[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
    basis[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
    basis[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * id;
    basis[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id;
    basis[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
    basis[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * id;
    basis[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
    basis[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
  }
  Opt handle(V3 p) const override {
    Opt out;
    Vec pv(3);
    for (int j = 0; j < 3; j++) pv[j] = p.x * basis[0][j] + p.y * basis[1][j] + p.z * basis[2][j];  // position^T * basis
    const double epsilon = -1e-11;
    if (pv[0] >= epsilon && pv[1] >= epsilon && pv[2] >= epsilon) {
      const double n = norm(pv);
      for (auto &x : pv) x = std::min(std::max(x / n, 0.0), 1.0);
      out.ok = true;
      out.v = pv;
    }
    return out;
  }
};

// point_source_panner.cpp:53-101
struct VirtualNgon : RegionHandler {
  V3 centre;
  Vec centre_downmix;
  std::vector<std::unique_ptr<Triplet>> regions;
  VirtualNgon(std::vector<int> ch, std::vector<V3> pos, V3 centre_position, Vec downmix)
      : RegionHandler(std::move(ch), std::move(pos)), centre(centre_position), centre_downmix(std::move(downmix)) {
    const int n = (int)channels.size();
    const std::vector<int> order = ngon_vertex_order(positions);
    for (int i = 0; i < n; i++) {
      const int j = (i + 1) % n;
      regions.emplace_back(new Triplet({order[i], order[j], n}, {positions[order[i]], positions[order[j]], centre}));
    }
  }
  Opt handle(V3 position) const override {
    for (auto &region : regions) {
      Opt pv = region->handle_remap(position, (int)centre_downmix.size() + 1);
      if (pv.ok) {
        Opt out;
        out.ok = true;
        const size_t n = centre_downmix.size();
        out.v.resize(n);
        for (size_t i = 0; i < n; i++) out.v[i] = pv.v[i] + centre_downmix[i] * pv.v[n];
        const double nn = norm(out.v);
        for (auto &x : out.v) x /= nn;
        return out;
      }
    }
    return Opt();
  }
};

// point_source_panner.cpp:157-174
inline Vec real_quadratic_roots(double a, double b, double c) {
  const double eps = 1e-10;
  if (std::fabs(c) < eps) return {0.0};
  if (std::fabs(a) < eps) return {-c / b};
  const double det = b * b - 4.0 * a * c;
  if (det > eps) return {(-b + std::sqrt(det)) / (2.0 * a), (-b - std::sqrt(det)) / (2.0 * a)};
  if (det > -eps) return {-b / (2.0 * a)};
  return {};
}

// point_source_panner.cpp:103-190
struct QuadRegion : RegionHandler {
  std::vector<int> order;
  V3 basis_x[3], basis_y[3];  // rows of the poly bases
  static void calc_poly_basis(const V3 (&p)[4], V3 (&rows)[3]) {
    const V3 a = p[0], b = p[1], c = p[2], d = p[3];
    rows[0] = cross(b - a, c - d);
    rows[1] = cross(a, c - d) + cross(b - a, d);
    rows[2] = cross(a, d);
  }
  QuadRegion(std::vector<int> ch, std::vector<V3> pos) : RegionHandler(std::move(ch), std::move(pos)) {
    order = ngon_vertex_order(positions);
    V3 re[4], sh[4];
    for (int i = 0; i < 4; i++) re[i] = positions[order[i]];
    for (int i = 0; i < 4; i++) sh[i] = re[(i + 1) % 4];
    calc_poly_basis(re, basis_x);
    calc_poly_basis(sh, basis_y);
  }
  static bool pan(V3 position, const V3 (&basis)[3], double &out) {
    const double epsilon = 1e-10;
    for (double root : real_quadratic_roots(dot(basis[0], position), dot(basis[1], position), dot(basis[2], position)))
      if (-epsilon < root && root < 1.0 + epsilon) {
        out = std::min(std::max(root, 0.0), 1.0);
        return true;
      }
    return false;
  }
  Opt handle(V3 position) const override {
    double x, y;
    if (!pan(position, basis_x, x) || !pan(position, basis_y, y)) return Opt();
    Vec pvs(4, 0.0);
    pvs[order[0]] = (1 - x) * (1 - y);
    pvs[order[1]] = x * (1 - y);
    pvs[order[2]] = x * y;
    pvs[order[3]] = (1 - x) * y;
    V3 vel{0, 0, 0};
    for (int i = 0; i < 4; i++) vel = vel + positions[i] * pvs[i];
    if (dot(vel, position) <= 0) return Opt();
    const double n = norm(pvs);
    for (auto &v : pvs) v /= n;
    Opt out;
    out.ok = true;
    out.v = pvs;
    return out;
  }
};

// point_source_panner.hpp:113-120
struct PointSourcePanner {
  virtual ~PointSourcePanner() = default;
  virtual Opt handle(V3 position) const = 0;
  virtual int n_out() const = 0;
};

// point_source_panner.cpp:192-231
struct PolarPointSourcePanner : PointSourcePanner {
  std::vector<std::unique_ptr<RegionHandler>> regions;
  int n;
  explicit PolarPointSourcePanner(std::vector<std::unique_ptr<RegionHandler>> r) : regions(std::move(r)) {
    int mx = 0;
    for (auto &reg : regions)
      for (int c : reg->channels) mx = std::max(mx, c);
    n = mx + 1;
  }
  Opt handle(V3 position) const override {
    for (auto &region : regions) {
      Opt pv = region->handle_remap(position, n);
      if (pv.ok) return pv;
    }
    return Opt();
  }
  int n_out() const override { return n; }
};

// point_source_panner.cpp:235-254: downmix is (inputs x outputs)
struct PointSourcePannerDownmix : PointSourcePanner {
  std::shared_ptr<PointSourcePanner> psp;
  std::vector<Vec> downmix;
  PointSourcePannerDownmix(std::shared_ptr<PointSourcePanner> p, std::vector<Vec> d) : psp(std::move(p)), downmix(std::move(d)) {}
  Opt handle(V3 position) const override {
    Opt pv = psp->handle(position);
    if (!pv.ok) return pv;
    Vec out(downmix[0].size(), 0.0);
    for (size_t i = 0; i < downmix.size(); i++)
      for (size_t j = 0; j < out.size(); j++) out[j] += downmix[i][j] * pv.v[i];
    const double n = norm(out);
    for (auto &v : out) v /= n;
    pv.v = out;
    return pv;
  }
  int n_out() const override { return (int)downmix[0].size(); }
};

struct ExtraChan {
  double az, el_real, az_nom, el_nom;
};
// point_source_panner.cpp:256-349
inline void extra_pos_vertical_nominal(const std::vector<Chan> &layout, std::vector<ExtraChan> &extra, std::vector<Vec> &downmix) {
  const size_t n = layout.size();
  downmix.assign(n, Vec(n, 0.0));
  for (size_t i = 0; i < n; i++) downmix[i][i] = 1.0;
  std::vector<Chan> mid;
  for (auto &c : layout)
    if (-10 <= c.el_nom && c.el_nom <= 10) mid.push_back(c);
  const double layers[2][3] = {{-30.0, -70.0, -10.0}, {30.0, 10.0, 70.0}};
  for (auto &layer : layers) {
    const double nominal = layer[0], lower = layer[1], upper = layer[2];
    std::vector<Chan> cur;
    for (auto &c : layout)
      if (lower <= c.el_nom && c.el_nom <= upper) cur.push_back(c);
    double az_limit = 0.0, real_el = 0.0;
    if (!cur.empty()) {
      double range = std::numeric_limits<double>::min();
      for (auto &c : cur) range = std::max(range, std::fabs(c.az_nom));
      az_limit = range + 40.0;
      for (auto &c : cur) real_el += c.el;
      real_el /= (double)cur.size();
    } else {
      real_el = nominal;
    }
    const double epsilon = 1e-5;
    for (auto &m : mid)
      if (std::fabs(m.az) >= az_limit - epsilon) {
        extra.push_back({m.az, real_el, m.az_nom, nominal});
        Vec row(n, 0.0);
        for (size_t i = 0; i < n; i++)
          if (layout[i].name == m.name) {
            row[i] = 1.0;
            break;
          }
        downmix.push_back(row);
      }
  }
}

// point_source_panner.cpp:351-365
inline std::set<int> adjacent_verts(const std::vector<std::vector<int>> &facets, int vert) {
  std::set<int> ret;
  for (auto &f : facets)
    if (std::find(f.begin(), f.end(), vert) != f.end()) ret.insert(f.begin(), f.end());
  ret.erase(vert);
  return ret;
}

// point_source_panner.cpp:431-476 (getAugmentedLayout) and :478-561 (configureFullPolarPanner)
inline std::shared_ptr<PointSourcePanner> configure_full_polar_panner(const std::string &name, const std::vector<Chan> &layout) {
  std::vector<ExtraChan> extra;
  std::vector<Vec> downmix;
  extra_pos_vertical_nominal(layout, extra, downmix);
  std::vector<V3> real;
  for (auto &c : layout) real.push_back(cart(c.az, c.el, 1.0));
  for (auto &e : extra) real.push_back(cart(e.az, e.el_real, 1.0));
  std::vector<V3> virt;
  virt.push_back({0.0, 0.0, -1.0});
  bool has_top = false;
  for (auto &c : layout) has_top = has_top || c.name == "T+000" || c.name == "UH+180";
  if (!has_top) virt.push_back({0.0, 0.0, 1.0});
  std::set<int> virtual_verts;
  for (auto &v : virt) {
    virtual_verts.insert((int)real.size());
    real.push_back(v);
  }
  const int *ft = layout_facets(name);
  if (!ft) throw std::invalid_argument("no facets for layout " + name);
  std::vector<std::vector<int>> facets;
  while (*ft) {
    const int k = *ft++;
    facets.emplace_back(ft, ft + k);  // (ascending: the reference's Facet is a std::set<int>)
    ft += k;
  }
  std::vector<std::unique_ptr<RegionHandler>> regions;
  for (int vv : virtual_verts) {
    const std::set<int> rv = adjacent_verts(facets, vv);
    for (int r : rv)
      if (virtual_verts.count(r)) throw std::runtime_error("invalid triangulation");
    std::vector<int> ch(rv.begin(), rv.end());
    std::vector<V3> pos;
    for (int r : ch) pos.push_back(real[r]);
    Vec cd(ch.size(), 1.0 / std::sqrt((double)ch.size()));
    regions.emplace_back(new VirtualNgon(ch, pos, real[vv], cd));
  }
  for (auto &f : facets) {
    bool touches = false;
    for (int v : f) touches = touches || virtual_verts.count(v);
    if (touches) continue;
    std::vector<V3> pos;
    for (int v : f) pos.push_back(real[v]);
    if (f.size() == 3) regions.emplace_back(new Triplet(f, pos));
    else if (f.size() == 4) regions.emplace_back(new QuadRegion(f, pos));
    else throw std::runtime_error("facets with more than 4 vertices are not supported");
  }
  return std::make_shared<PointSourcePannerDownmix>(std::make_shared<PolarPointSourcePanner>(std::move(regions)), downmix);
}

// point_source_panner.cpp:367-398
struct StereoPannerDownmix : RegionHandler {
  std::shared_ptr<PointSourcePanner> psp;
  StereoPannerDownmix(std::vector<int> ch, std::vector<V3> pos) : RegionHandler(std::move(ch), std::move(pos)) {
    psp = configure_full_polar_panner("0+5+0", layout_without_lfe("0+5+0"));
  }
  Opt handle(V3 position) const override {
    const double dm[2][5] = {{1.0, 0.0, std::sqrt(3.0) / 3.0, std::sqrt(0.5), 0.0}, {0.0, 1.0, std::sqrt(3.0) / 3.0, 0.0, std::sqrt(0.5)}};
    Opt pv = psp->handle(position);
    if (!pv.ok) return pv;
    Vec out(2, 0.0);
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 5; j++) out[i] += dm[i][j] * pv.v[j];
    const double n = norm(out);
    for (auto &v : out) v /= n;
    const double front = std::max(pv.v[0], std::max(pv.v[1], pv.v[2]));
    const double back = std::max(pv.v[3], pv.v[4]);
    const double s = std::pow(0.5, 0.5 * back / (front + back));
    for (auto &v : out) v *= s;
    Opt o;
    o.ok = true;
    o.v = out;
    return o;
  }
};

// point_source_panner.cpp:406-429, :586-600
inline std::shared_ptr<PointSourcePanner> configure_polar_panner(const std::string &name, const double *real_az = nullptr,
                                                                 const double *real_el = nullptr) {
  const std::vector<Chan> layout = layout_without_lfe(name, nullptr, real_az, real_el);
  check_screen_speakers(layout);
  if (name == "0+2+0") {
    int li = 0, ri = 0;
    for (size_t i = 0; i < layout.size(); i++) {
      if (layout[i].name == "M+030") li = (int)i;
      if (layout[i].name == "M-030") ri = (int)i;
    }
    std::vector<std::unique_ptr<RegionHandler>> regions;
    regions.emplace_back(new StereoPannerDownmix({li, ri}, {cart(layout[li].az, layout[li].el, 1.0), cart(layout[ri].az, layout[ri].el, 1.0)}));
    return std::make_shared<PolarPointSourcePanner>(std::move(regions));
  }
  return configure_full_polar_panner(name, layout);
}

// What GainCalculatorObjectsImpl's constructor builds (gain_calculator_objects.cpp:24-31): the polar point
// source panner of the layout without LFE and the LFE mask.  calculate() itself goes through PolarExtent
// even for point sources (a distance under 1 widens a zero extent): oracle/extent_oracle.hpp.
struct PannerSetup {
  std::shared_ptr<PointSourcePanner> psp;
  std::vector<bool> is_lfe;
  explicit PannerSetup(const std::string &layout, const double *real_az = nullptr, const double *real_el = nullptr) {
    layout_without_lfe(layout, &is_lfe);
    psp = configure_polar_panner(layout, real_az, real_el);
  }
  int n_out() const { return (int)is_lfe.size(); }
};

}  // namespace panner_oracle

// ------------------------------------------------------------------------------------------------
// HOA decode matrix (AllRAD): src/hoa/hoa.hpp:16-182, src/hoa/hoa.cpp:4-14,
// src/hoa/gain_calculator_hoa.cpp:8-72.  Associated Legendre functions and factorials (Boost.Math in the
// reference) restated from their definitions.
// ------------------------------------------------------------------------------------------------
#include "tdesign_5200.h"

namespace hoa_oracle {

using panner_oracle::V3;
using panner_oracle::Vec;

inline double factorial(int n) {
  double f = 1.0;
  for (int i = 2; i <= n; i++) f *= i;
  return f;
}
// P_n^m(x) WITHOUT the Condon-Shortley phase (hoa.hpp:18-22: (-1)^m * boost::math::legendre_p, which has it):
// (1 - x^2)^(m/2) d^m/dx^m P_n(x), by the standard upward recurrence in n
inline double alegendre(int n, int m, double x) {
  double pmm = 1.0;
  const double somx2 = std::sqrt((1.0 - x) * (1.0 + x));
  for (int i = 1; i <= m; i++) pmm *= (2.0 * i - 1.0) * somx2;
  if (n == m) return pmm;
  double pmmp1 = x * (2.0 * m + 1.0) * pmm;
  if (n == m + 1) return pmmp1;
  double pll = 0.0;
  for (int ll = m + 2; ll <= n; ll++) {
    pll = (x * (2.0 * ll - 1.0) * pmmp1 - (ll + m - 1.0) * pmm) / (ll - m);
    pmm = pmmp1;
    pmmp1 = pll;
  }
  return pll;
}
// hoa.hpp:44-76
inline double norm_N3D(int n, int am) { return std::sqrt((2.0 * n + 1.0) * factorial(n - am) / factorial(n + am)); }
inline double norm_SN3D(int n, int am) { return std::sqrt(factorial(n - am) / factorial(n + am)); }
inline double norm_FuMa(int n, int am) {
  const double f[4][4] = {{1.0 / std::sqrt(2.0), 0, 0, 0},
                          {1.0, 1.0, 0, 0},
                          {1.0, 2.0 / std::sqrt(3.0), 2.0 / std::sqrt(3.0), 0},
                          {1.0, std::sqrt(45.0 / 32.0), 3.0 / std::sqrt(5.0), std::sqrt(8.0 / 5.0)}};
  if (n > 3) throw std::out_of_range("FuMa is defined up to order 3");
  return f[n][am] * norm_SN3D(n, am);
}
typedef double (*norm_f)(int, int);
inline norm_f get_norm(const std::string &name) {
  if (name == "N3D") return norm_N3D;
  if (name == "SN3D") return norm_SN3D;
  if (name == "FuMa") return norm_FuMa;
  throw std::invalid_argument("ADM error: unknown normalization type: '" + name + "'");
}
// hoa.hpp:99-112
inline double sph_harm(int n, int m, double az, double el, norm_f norm) {
  double scale = 1.0;
  if (m > 0) scale = std::sqrt(2.0) * std::cos(m * az);
  else if (m < 0) scale = -std::sqrt(2.0) * std::sin(m * az);
  return norm(n, std::abs(m)) * alegendre(n, std::abs(m), std::sin(el)) * scale;
}
// hoa.cpp:4-14
inline std::vector<V3> load_points() {
  std::vector<V3> p(ear_oracle_data::kTDesignPoints);
  for (int i = 0; i < ear_oracle_data::kTDesignPoints; i++) {
    const double phi = ear_oracle_data::kTDesign[i][0], theta = ear_oracle_data::kTDesign[i][1];
    p[i] = {std::sin(theta) * std::cos(phi), std::sin(theta) * std::sin(phi), std::cos(theta)};
  }
  return p;
}
// gain_calculator_hoa.cpp:25-71; out: D_full [n_channels][n_coef] row-major (LFE rows zero)
inline void decode_matrix(const std::string &layout, const std::vector<int> &orders, const std::vector<int> &degrees,
                          const std::string &normalization, std::vector<double> &out, int &n_channels,
                          const double *real_az = nullptr, const double *real_el = nullptr) {
  if (orders.size() != degrees.size()) throw std::invalid_argument("orders and degrees must be the same size");
  for (size_t i = 0; i < orders.size(); i++) {
    if (orders[i] < 0) throw std::invalid_argument("orders must not be negative");
    if (std::abs(degrees[i]) > orders[i]) throw std::invalid_argument("magnitude of degree must not be greater than order");
  }
  const norm_f norm = get_norm(normalization);
  std::vector<bool> is_lfe;
  panner_oracle::layout_without_lfe(layout, &is_lfe);
  const auto psp = panner_oracle::configure_polar_panner(layout, real_az, real_el);
  const std::vector<V3> points = load_points();
  const size_t P = points.size(), C = orders.size(), S = (size_t)psp->n_out();
  // Y_virt [C][P] (N3D), G_virt [S][P]
  std::vector<double> Y(C * P), G(S * P);
  for (size_t pi = 0; pi < P; pi++) {
    const double az = -std::atan2(points[pi].x, points[pi].y);
    const double el = std::atan2(points[pi].z, std::hypot(points[pi].x, points[pi].y));
    for (size_t c = 0; c < C; c++) Y[c * P + pi] = sph_harm(orders[c], degrees[c], az, el, norm_N3D);
    const panner_oracle::Opt pv = psp->handle(points[pi]);
    if (!pv.ok) throw std::runtime_error("point not handled by the panner");
    for (size_t s = 0; s < S; s++) G[s * P + pi] = pv.v[s];
  }
  // D = G_virt * (Y_virt^T / P)
  std::vector<double> D(S * C, 0.0);
  for (size_t s = 0; s < S; s++)
    for (size_t c = 0; c < C; c++) {
      double acc = 0.0;
      for (size_t pi = 0; pi < P; pi++) acc += G[s * P + pi] * (Y[c * P + pi] / (double)P);
      D[s * C + c] = acc;
    }
  // normalize_decode_matrix: D *= sqrt(P) / |D Y_virt|_F   (hoa.hpp:140-143)
  double fro = 0.0;
  for (size_t s = 0; s < S; s++)
    for (size_t pi = 0; pi < P; pi++) {
      double v = 0.0;
      for (size_t c = 0; c < C; c++) v += D[s * C + c] * Y[c * P + pi];
      fro += v * v;
    }
  const double k = std::sqrt((double)P) / std::sqrt(fro);
  for (auto &v : D) v *= k;
  // D *= diag(norm_N3D / norm)   (normalisation_conversion(n, m, norm_N3D, norm), hoa.hpp:147-160)
  for (size_t c = 0; c < C; c++) {
    const double conv = norm_N3D(orders[c], std::abs(degrees[c])) / norm(orders[c], std::abs(degrees[c]));
    for (size_t s = 0; s < S; s++) D[s * C + c] *= conv;
  }
  n_channels = (int)is_lfe.size();
  out.assign((size_t)n_channels * C, 0.0);
  size_t s = 0;
  for (size_t ch = 0; ch < is_lfe.size(); ch++) {
    if (is_lfe[ch]) continue;
    for (size_t c = 0; c < C; c++) out[ch * C + c] = D[s * C + c];
    s++;
  }
}

}  // namespace hoa_oracle
