// oracle/oracle_capi.cpp — plain-C entry points over ear_oracle.hpp so that the
// pytest suite (ctypes) and bench.py's cpu_baseline leg can drive the CPU
// restatement.  TEST INFRASTRUCTURE ONLY; never linked into libearhip.so.
#include <cstring>
#include <sstream>

#include "ear_oracle.hpp"

using namespace ear_oracle;

namespace {
thread_local std::string g_err;
template <typename F>
int guarded(F &&f) {
  try {
    f();
    return 0;
  } catch (const invalid_argument &e) {
    g_err = e.what();
    return 1;
  } catch (const internal_error &e) {
    g_err = e.what();
    return 2;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 3;
  }
}
std::vector<const float *> planar_c(const float *base, size_t nch, size_t stride,
                                    size_t ofs) {
  std::vector<const float *> p(nch);
  for (size_t c = 0; c < nch; c++) p[c] = base + c * stride + ofs;
  return p;
}
std::vector<float *> planar(float *base, size_t nch, size_t stride, size_t ofs) {
  std::vector<float *> p(nch);
  for (size_t c = 0; c < nch; c++) p[c] = base + c * stride + ofs;
  return p;
}
}  // namespace

extern "C" {

const char *oracle_last_error() { return g_err.c_str(); }

// ---- GainInterpolator ------------------------------------------------------
// kind: 0 single (1->1), 1 vector (1->n_out), 2 matrix (n_in->n_out).
// values: [npoints][n_in][n_out].  in: [n_in][total], out: [n_out][total],
// processed as consecutive calls of call_sizes[0..ncalls) samples starting at
// sample index t0 (so block_start of call k = t0 + sum(call_sizes[:k])).
int oracle_gain_interp(int kind, int n_in, int n_out, int npoints,
                       const int64_t *times, const float *values, int64_t t0,
                       const size_t *call_sizes, int ncalls, const float *in,
                       float *out) {
  return guarded([&] {
    size_t total = 0;
    for (int k = 0; k < ncalls; k++) total += call_sizes[k];
    auto run = [&](auto &gi) {
      size_t ofs = 0;
      for (int k = 0; k < ncalls; k++) {
        auto ip = planar_c(in, n_in, total, ofs);
        auto op = planar(out, n_out, total, ofs);
        gi.process((SampleIndex)(t0 + (int64_t)ofs), call_sizes[k], ip.data(),
                   op.data());
        ofs += call_sizes[k];
      }
    };
    const size_t pstride = (size_t)n_in * n_out;
    if (kind == 0) {
      GainInterpolator<LinearInterpSingle> gi;
      for (int p = 0; p < npoints; p++)
        gi.interp_points.emplace_back((SampleIndex)times[p], values[p * pstride]);
      run(gi);
    } else if (kind == 1) {
      GainInterpolator<LinearInterpVector> gi;
      for (int p = 0; p < npoints; p++)
        gi.interp_points.emplace_back(
            (SampleIndex)times[p],
            std::vector<float>(values + p * pstride, values + (p + 1) * pstride));
      run(gi);
    } else {
      GainInterpolator<LinearInterpMatrix> gi;
      for (int p = 0; p < npoints; p++) {
        std::vector<std::vector<float>> mat(n_in);
        for (int m = 0; m < n_in; m++)
          mat[m].assign(values + p * pstride + (size_t)m * n_out,
                        values + p * pstride + (size_t)(m + 1) * n_out);
        gi.interp_points.emplace_back((SampleIndex)times[p], std::move(mat));
      }
      run(gi);
    }
  });
}

// ---- FFT -------------------------------------------------------------------
int oracle_cfft_f32(size_t n, int inverse, const float *in, float *out) {
  return guarded([&] {
    KissLikeFFT<float> f(n, inverse != 0);
    f.transform(reinterpret_cast<const std::complex<float> *>(in),
                reinterpret_cast<std::complex<float> *>(out));
  });
}
int oracle_cfft_f64(size_t n, int inverse, const double *in, double *out) {
  return guarded([&] {
    KissLikeFFT<double> f(n, inverse != 0);
    f.transform(reinterpret_cast<const std::complex<double> *>(in),
                reinterpret_cast<std::complex<double> *>(out));
  });
}
// out: n_fft/2+1 complex
int oracle_rfft_forward(size_t n_fft, const float *in, float *out) {
  return guarded([&] {
    RealFFT f(n_fft);
    f.forward(in, reinterpret_cast<std::complex<float> *>(out));
  });
}
int oracle_rfft_reverse(size_t n_fft, const float *in, float *out) {
  return guarded([&] {
    RealFFT f(n_fft);
    f.reverse(reinterpret_cast<const std::complex<float> *>(in), out);
  });
}

// ---- BlockConvolver ----------------------------------------------------------
struct OCtx { std::shared_ptr<block_convolver::Context> p; };
struct OFilter { std::shared_ptr<const block_convolver::Filter> p; };
struct OConv { std::unique_ptr<block_convolver::BlockConvolver> p; };

OCtx *oracle_conv_ctx_create(size_t block_size) {
  return new OCtx{std::make_shared<block_convolver::Context>(block_size)};
}
void oracle_conv_ctx_destroy(OCtx *c) { delete c; }
OFilter *oracle_conv_filter_create(OCtx *c, size_t n, const float *taps) {
  return new OFilter{std::make_shared<block_convolver::Filter>(c->p, n, taps)};
}
size_t oracle_conv_filter_num_blocks(OFilter *f) { return f->p->num_blocks(); }
// spectrum of partition `block`, fd_size complex values
void oracle_conv_filter_spectrum(OFilter *f, size_t block, float *out) {
  auto &b = f->p->blocks[block];
  std::memcpy(out, b.data(), b.size() * sizeof(std::complex<float>));
}
void oracle_conv_filter_destroy(OFilter *f) { delete f; }
// filter may be null; num_blocks 0 = take from filter
int oracle_conv_create(OCtx *c, OFilter *f, size_t num_blocks, OConv **out) {
  return guarded([&] {
    if (f)
      *out = new OConv{std::unique_ptr<block_convolver::BlockConvolver>(
          new block_convolver::BlockConvolver(c->p, f->p, num_blocks))};
    else
      *out = new OConv{std::unique_ptr<block_convolver::BlockConvolver>(
          new block_convolver::BlockConvolver(c->p, num_blocks))};
  });
}
void oracle_conv_destroy(OConv *c) { delete c; }
int oracle_conv_set_filter(OConv *c, OFilter *f) {
  return guarded([&] { c->p->set_filter(f ? f->p : nullptr); });
}
int oracle_conv_crossfade_filter(OConv *c, OFilter *f) {
  return guarded([&] { c->p->crossfade_filter(f ? f->p : nullptr); });
}
int oracle_conv_process(OConv *c, const float *in, float *out) {
  return guarded([&] { c->p->process(in, out); });
}

// ---- DelayBuffer -----------------------------------------------------------
DelayBuffer *oracle_delay_create(size_t nch, size_t delay) {
  return new DelayBuffer(nch, delay);
}
void oracle_delay_destroy(DelayBuffer *d) { delete d; }
// in/out: [nch][stride], processes nsamples from offset ofs
int oracle_delay_process(DelayBuffer *d, size_t nch, size_t nsamples,
                         const float *in, float *out, size_t stride, size_t ofs) {
  return guarded([&] {
    auto ip = planar_c(in, nch, stride, ofs);
    auto op = planar(out, nch, stride, ofs);
    d->process(nsamples, ip.data(), op.data());
  });
}

// ---- VariableBlockSizeAdapter ------------------------------------------------
typedef void (*oracle_process_cb)(const float *const *in, float *const *out,
                                  void *user);
struct OVbs { std::unique_ptr<VariableBlockSizeAdapter> p; };
OVbs *oracle_vbs_create(size_t block_size, size_t n_in, size_t n_out,
                        oracle_process_cb cb, void *user) {
  return new OVbs{std::unique_ptr<VariableBlockSizeAdapter>(
      new VariableBlockSizeAdapter(
          block_size, n_in, n_out,
          [cb, user](const float *const *i, float *const *o) { cb(i, o, user); }))};
}
void oracle_vbs_destroy(OVbs *v) { delete v; }
int oracle_vbs_get_delay(OVbs *v) { return v->p->get_delay(); }
int oracle_vbs_process(OVbs *v, size_t n_in, size_t n_out, size_t nsamples,
                       const float *in, float *out, size_t stride, size_t ofs) {
  return guarded([&] {
    auto ip = planar_c(in, n_in, stride, ofs);
    auto op = planar(out, n_out, stride, ofs);
    v->p->process(nsamples, ip.data(), op.data());
  });
}

// ---- decorrelator design -----------------------------------------------------
int oracle_design_decorrelator_basic(int id, int size, double *out) {
  return guarded([&] {
    auto h = design_decorrelator_basic(id, size);
    std::copy(h.begin(), h.end(), out);
  });
}
// names: '\n'-separated channel names; out: [nch][512] float
int oracle_design_decorrelators(const char *names, float *out) {
  return guarded([&] {
    std::vector<std::string> v;
    std::stringstream ss(names);
    for (std::string s; std::getline(ss, s, '\n');) v.push_back(s);
    auto f = design_decorrelators(v);
    for (size_t c = 0; c < f.size(); c++)
      std::copy(f[c].begin(), f[c].end(), out + c * kDecorrelatorSize);
  });
}
int oracle_decorrelator_compensation_delay() {
  return decorrelator_compensation_delay();
}

// ---- composed Objects render -------------------------------------------------
ObjectsRenderer *oracle_render_create(size_t n_obj, size_t n_out, size_t block,
                                      const float *filters, size_t ntaps,
                                      size_t delay) {
  std::vector<std::vector<float>> f(n_out);
  for (size_t c = 0; c < n_out; c++)
    f[c].assign(filters + c * ntaps, filters + (c + 1) * ntaps);
  return new ObjectsRenderer(n_obj, n_out, block, f, delay);
}
void oracle_render_destroy(ObjectsRenderer *r) { delete r; }
// bus: 0 direct, 1 diffuse; gains: [npoints][n_out]
int oracle_render_set_points(ObjectsRenderer *r, size_t obj, int bus, int npoints,
                             const int64_t *times, const float *gains,
                             size_t n_out) {
  return guarded([&] {
    auto &gi = bus == 0 ? r->direct.at(obj) : r->diffuse.at(obj);
    gi.interp_points.clear();
    for (int p = 0; p < npoints; p++)
      gi.interp_points.emplace_back(
          (SampleIndex)times[p],
          std::vector<float>(gains + (size_t)p * n_out, gains + (size_t)(p + 1) * n_out));
  });
}
// in: [n_obj][nblocks*block], out: [n_out][nblocks*block]
int oracle_render_process(ObjectsRenderer *r, size_t n_obj, size_t n_out,
                          size_t block, size_t nblocks, const float *in,
                          float *out) {
  return guarded([&] {
    const size_t stride = nblocks * block;
    for (size_t t = 0; t < nblocks; t++) {
      auto ip = planar_c(in, n_obj, stride, t * block);
      auto op = planar(out, n_out, stride, t * block);
      r->process(ip.data(), op.data());
    }
  });
}

}  // extern "C"

// ---- Objects gain producer (panner_oracle.hpp) -------------------------------------------------
#include "panner_oracle.hpp"
#include "extent_oracle.hpp"

namespace {
typedef extent_oracle::GainCalculatorObjects OracleObjects;  // point source panner + polar extent panner
const panner_oracle::PannerSetup &setup_of(void *p) { return static_cast<OracleObjects *>(p)->base; }
}  // namespace

extern "C" {

void *oracle_panner_create(const char *layout) {
  void *out = nullptr;
  guarded([&] { out = new OracleObjects(layout); });
  return out;
}
// real loudspeaker positions, one (azimuth, elevation) per channel of the full layout; *status: 0 ok,
// 1 invalid_argument, 4 not_implemented, 3 other (message: oracle_last_error)
void *oracle_panner_create_positions(const char *layout, const double *az, const double *el, int *status) {
  void *out = nullptr;
  try {
    out = new OracleObjects(layout, az, el);
    *status = 0;
  } catch (const panner_oracle::not_implemented &e) {
    g_err = e.what();
    *status = 4;
  } catch (const std::invalid_argument &e) {
    g_err = e.what();
    *status = 1;
  } catch (const std::exception &e) {
    g_err = e.what();
    *status = 3;
  }
  return out;
}
void oracle_panner_destroy(void *p) { delete static_cast<OracleObjects *>(p); }
int oracle_panner_n_out(void *p) { return setup_of(p).n_out(); }
// n positions: az, el, dist, gain, diffuse [n] -> direct, diffuse [n][n_out]; returns the number of
// positions no region handled (their rows are left untouched)
int oracle_panner_calculate(void *p, size_t n, const double *az, const double *el, const double *dist,
                            const double *gain, const double *diffuse, float *direct, float *diff) {
  auto *g = static_cast<OracleObjects *>(p);
  const size_t no = (size_t)g->base.n_out();
  int missed = 0;
  for (size_t i = 0; i < n; i++)
    if (!g->calculate(az[i], el[i], dist[i], 0.0, 0.0, 0.0, gain[i], diffuse[i], direct + i * no, diff + i * no)) missed++;
  return missed;
}
// the inner point source panner (layout without LFE): pv [n][n_out_without_lfe] in double
int oracle_psp_n_out(void *p) { return setup_of(p).psp->n_out(); }
int oracle_psp_handle(void *p, size_t n, const double *xyz, double *pv) {
  const panner_oracle::PannerSetup *g = &setup_of(p);
  const size_t no = (size_t)g->psp->n_out();
  int missed = 0;
  for (size_t i = 0; i < n; i++) {
    const panner_oracle::Opt o = g->psp->handle({xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
    if (!o.ok) {
      missed++;
      continue;
    }
    for (size_t c = 0; c < no; c++) pv[i * no + c] = o.v[c];
  }
  return missed;
}
// extraPosVerticalNominal of a layout without LFE: returns the number of extra channels; az / el [<= 32];
// downmix_index [<= 32]: the real channel each extra one is mixed into
int oracle_extra_pos_vertical_nominal(const char *layout, double *az, double *el, int *downmix_index) {
  int n = -1;
  guarded([&] {
    const auto chans = panner_oracle::layout_without_lfe(layout);
    std::vector<panner_oracle::ExtraChan> extra;
    std::vector<panner_oracle::Vec> dm;
    panner_oracle::extra_pos_vertical_nominal(chans, extra, dm);
    for (size_t i = 0; i < extra.size(); i++) {
      az[i] = extra[i].az;
      el[i] = extra[i].el_real;
      downmix_index[i] = -1;
      for (size_t c = 0; c < chans.size(); c++)
        if (dm[chans.size() + i][c] == 1.0) downmix_index[i] = (int)c;
    }
    n = (int)extra.size();
  });
  return n;
}
// single regions for the reference's region-level tests: kind 0 Triplet, 1 QuadRegion, 2 VirtualNgon
// (positions [n][3]; VirtualNgon: centre [3], downmix [n]); pv [n]; returns 1 when handled
int oracle_region_handle(int kind, int n, const double *positions, const double *centre, const double *downmix,
                         const double *xyz, double *pv) {
  int ok = 0;
  guarded([&] {
    std::vector<int> ch(n);
    std::vector<panner_oracle::V3> pos(n);
    for (int i = 0; i < n; i++) ch[i] = i, pos[i] = {positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]};
    std::unique_ptr<panner_oracle::RegionHandler> r;
    if (kind == 0) r.reset(new panner_oracle::Triplet(ch, pos));
    else if (kind == 1) r.reset(new panner_oracle::QuadRegion(ch, pos));
    else r.reset(new panner_oracle::VirtualNgon(ch, pos, {centre[0], centre[1], centre[2]}, panner_oracle::Vec(downmix, downmix + n)));
    const panner_oracle::Opt o = r->handle({xyz[0], xyz[1], xyz[2]});
    if (o.ok) {
      ok = 1;
      for (int i = 0; i < n; i++) pv[i] = o.v[i];
    }
  });
  return ok;
}
int oracle_stereo_downmix_handle(const double *xyz, double *pv2) {
  int ok = 0;
  guarded([&] {
    panner_oracle::StereoPannerDownmix s({0, 1}, {panner_oracle::cart(30, 0, 1), panner_oracle::cart(-30, 0, 1)});
    const panner_oracle::Opt o = s.handle({xyz[0], xyz[1], xyz[2]});
    if (o.ok) ok = 1, pv2[0] = o.v[0], pv2[1] = o.v[1];
  });
  return ok;
}

}  // extern "C"

// ---- HOA decode matrix (hoa_oracle) -----------------------------------------------------------------
extern "C" {
// out [n_channels][n_coef] double; returns 0, or 1 (invalid argument) / 3 (other) with oracle_last_error()
int oracle_hoa_decode_matrix_positions(const char *layout, const double *az, const double *el, int n_coef, const int *orders,
                                       const int *degrees, const char *norm, double *out, int *n_channels);
int oracle_hoa_decode_matrix(const char *layout, int n_coef, const int *orders, const int *degrees, const char *norm,
                             double *out, int *n_channels) {
  return oracle_hoa_decode_matrix_positions(layout, nullptr, nullptr, n_coef, orders, degrees, norm, out, n_channels);
}
// az / el: real loudspeaker positions per channel of the full layout, or NULL (nominal)
int oracle_hoa_decode_matrix_positions(const char *layout, const double *az, const double *el, int n_coef, const int *orders,
                                       const int *degrees, const char *norm, double *out, int *n_channels) {
  try {
    std::vector<double> D;
    int nc = 0;
    hoa_oracle::decode_matrix(layout, std::vector<int>(orders, orders + n_coef), std::vector<int>(degrees, degrees + n_coef),
                              norm, D, nc, az, el);
    std::memcpy(out, D.data(), sizeof(double) * D.size());
    *n_channels = nc;
    return 0;
  } catch (const std::invalid_argument &e) {
    g_err = e.what();
    return 1;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 3;
  }
}
double oracle_sph_harm(int n, int m, double az, double el, const char *norm) {
  return hoa_oracle::sph_harm(n, m, az, el, hoa_oracle::get_norm(norm));
}
void oracle_tdesign_points(double *xyz) {
  const auto p = hoa_oracle::load_points();
  for (size_t i = 0; i < p.size(); i++) xyz[3 * i] = p[i].x, xyz[3 * i + 1] = p[i].y, xyz[3 * i + 2] = p[i].z;
}
}  // extern "C"

// ---- polar extent panner (extent_oracle.hpp) ----------------------------------------------------------
extern "C" {

// (the handles are oracle_panner_create's)
int oracle_extent_num_points(void *p) { return (int)static_cast<OracleObjects *>(p)->extent.num_points; }
// the grid: xyz [num_points][3] (double, before the cast to float)
void oracle_extent_grid(double *xyz) {
  const auto pos = extent_oracle::panning_positions_even(extent_oracle::kRows);
  for (size_t i = 0; i < pos.size(); i++) xyz[3 * i] = pos[i].x, xyz[3 * i + 1] = pos[i].y, xyz[3 * i + 2] = pos[i].z;
}
// PolarExtent::handle (library form, float core) and the tests' reference form (double): xyz [n][3],
// width / height / depth [n] -> pv [n][n_out_without_lfe] double; returns the number of unhandled positions
int oracle_extent_handle(void *p, int which, size_t n, const double *xyz, const double *width, const double *height,
                         const double *depth, double *pv) {
  auto *g = static_cast<OracleObjects *>(p);
  const size_t S = g->extent.num_speakers;
  int missed = 0;
  guarded([&] {
    std::unique_ptr<extent_oracle::test_reference::PolarExtentPanner> ref;
    if (which == 1) ref.reset(new extent_oracle::test_reference::PolarExtentPanner(g->base.psp));
    for (size_t i = 0; i < n; i++) {
      const panner_oracle::V3 pos = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
      panner_oracle::Vec out;
      if (which == 1) {
        out = ref->handle(pos, width[i], height[i], depth[i]);
      } else if (!g->extent.handle(pos, width[i], height[i], depth[i], out)) {
        missed++;
        continue;
      }
      for (size_t s = 0; s < S; s++) pv[i * S + s] = out[s];
    }
  });
  return missed;
}
void oracle_extent_calc_basis(const double *xyz, double *m9) {
  const extent_oracle::Mat3 b = extent_oracle::calc_basis({xyz[0], xyz[1], xyz[2]});
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) m9[3 * i + j] = b.m[i][j];
}
double oracle_extent_mod(double extent, double distance) { return extent_oracle::extent_mod(extent, distance); }
// weight of `point` for an extent of width x height (degrees) centred on `centre`: which 0 = the library's
// float core, 1 = the tests' reference weighting function
double oracle_extent_weight(void *p, int which, const double *centre, double width, double height, const double *point) {
  const panner_oracle::V3 c = {centre[0], centre[1], centre[2]}, q = {point[0], point[1], point[2]};
  if (which == 1) return extent_oracle::test_reference::WeightingFunction(c, width, height)(q);
  auto *g = static_cast<OracleObjects *>(p);
  g->extent.setup_weighting_function(c, width, height);
  return (double)g->extent.weight((float)q.x, (float)q.y, (float)q.z);
}
// GainCalculatorObjects::calculate with extent: returns the number of unhandled positions
int oracle_extent_calculate(void *p, size_t n, const double *az, const double *el, const double *dist, const double *width,
                            const double *height, const double *depth, const double *gain, const double *diffuse,
                            float *direct, float *diff) {
  auto *g = static_cast<OracleObjects *>(p);
  const size_t no = g->base.is_lfe.size();
  int missed = 0;
  for (size_t i = 0; i < n; i++)
    if (!g->calculate(az[i], el[i], dist[i], width[i], height[i], depth[i], gain[i], diffuse[i], direct + i * no, diff + i * no))
      missed++;
  return missed;
}

}  // extern "C"
