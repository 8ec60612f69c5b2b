// Drop-in test: the C++14 mirror classes (libear_amd/host/ear/...) driven the way
// libear's own Catch2 tests drive the originals (reference tests/
// gain_interpolator_tests.cpp, block_convolver_tests.cpp, delay_buffer_tests.cpp,
// variable_block_size_tests.cpp), without Eigen/Catch2.  Expected values come
// from closed forms and brute-force time-domain convolution computed here in
// double precision, so this program needs nothing but libearhip.so and a GPU.
// Build (one line): g++ -std=c++14 -Iinclude -Ilibear_amd/host tests/cpp/test_dropin.cpp
//            -Llibear_amd/lib -learhip -Wl,-rpath,$PWD/libear_amd/lib -o test_dropin
#include <dlfcn.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <random>
#include <vector>

// (the include lines of a libear application: the reference's tests use exactly these paths)
#include "ear/bs2051.hpp"
#include "ear/common_types.hpp"
#include "ear/decorrelate.hpp"
#include "ear/dsp/dsp.hpp"
#include "ear/ear.hpp"
#include "ear/gain_calculators.hpp"
#include "ear/screen.hpp"
#include "ear/warnings.hpp"

using namespace ear;
using namespace ear::dsp;

// ---- heap watch: while armed, every operator new issued from this program (the mirror headers are
// header-only, so their allocations are here) or from libearhip.so is counted; the HIP runtime's own
// allocations are not ours to judge.  libear enforces the same for its process() calls with
// EIGEN_RUNTIME_NO_MALLOC (reference tests/gain_interpolator_tests.cpp:1,89,96).
static std::atomic<bool> g_heap_armed{false};
static std::atomic<long> g_heap_count{0};
static void note_allocation(void *caller) {
  if (!g_heap_armed.load(std::memory_order_relaxed)) return;
  g_heap_armed.store(false);  // (dladdr may allocate)
  Dl_info info;
  if (dladdr(caller, &info) && info.dli_fname &&
      (std::strstr(info.dli_fname, "libearhip") || std::strstr(info.dli_fname, "test_dropin")))
    g_heap_count++;
  g_heap_armed.store(true);
}
void *operator new(std::size_t n) {
  note_allocation(__builtin_return_address(0));
  void *p = std::malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void *operator new[](std::size_t n) {
  note_allocation(__builtin_return_address(0));
  void *p = std::malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void operator delete(void *p) noexcept { std::free(p); }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }

static int g_failed = 0, g_checks = 0;
#define CHECK(cond)                                                         \
  do {                                                                      \
    g_checks++;                                                             \
    if (!(cond)) {                                                          \
      g_failed++;                                                           \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);         \
    }                                                                       \
  } while (0)

using Vec = std::vector<float>;

static Vec random_vec(size_t n, unsigned seed) {
  std::mt19937 g(seed);
  Vec v(n);
  for (auto &x : v) x = (float)((double)g() / 4294967296.0 * 2.0 - 1.0);
  return v;
}
static bool is_approx(const Vec &a, const std::vector<double> &b, double prec = 1e-5) {
  double d = 0, na = 0, nb = 0;
  for (size_t i = 0; i < a.size(); i++) {
    d += (a[i] - b[i]) * (a[i] - b[i]);
    na += (double)a[i] * a[i];
    nb += b[i] * b[i];
  }
  return std::sqrt(d) <= prec * std::sqrt(std::min(na, nb));
}
static double rel_rms(const Vec &a, const Vec &b) {
  double d = 0, n = 0;
  for (size_t i = 0; i < a.size(); i++) {
    d += ((double)a[i] - b[i]) * ((double)a[i] - b[i]);
    n += (double)b[i] * b[i];
  }
  return std::sqrt(d / std::max(n, 1e-300));
}

// ---- GainInterpolator (reference tests/gain_interpolator_tests.cpp:83-183) --------------------
struct Seg { long a, b; bool ramp; long start, end; float s, e; };

static void run_single_case(const char *name, const std::vector<std::pair<long, float>> &pts,
                            long len, const std::vector<Seg> &segs, const std::vector<long> &sizes) {
  const Vec x = random_vec(len, 3);
  std::vector<double> want(len, 0.0);
  for (auto &sg : segs)
    for (long i = sg.a; i < sg.b; i++) {
      double g = sg.s;
      if (sg.ramp) {
        const double p = (double)(i - sg.start) / (double)(sg.end - sg.start);
        g = (double)sg.e * p + (1.0 - p) * (double)sg.s;
      }
      want[i] = g * x[i];
    }
  GainInterpolator<LinearInterpSingle> interp;
  for (auto &p : pts) interp.interp_points.emplace_back(p.first, p.second);
  for (long bs : sizes) {
    Vec out(len, 0.0f);
    for (long ofs = 0; ofs < len; ofs += bs) {
      const long n = std::min(bs, len - ofs);
      const float *ip = x.data() + ofs;
      float *op = out.data() + ofs;
      interp.process(ofs, (size_t)n, &ip, &op);
    }
    const bool ok = is_approx(out, want);
    if (!ok) std::printf("  case %s block size %ld\n", name, bs);
    CHECK(ok);
  }
}

static void test_gain_interpolator() {
  run_single_case("basic", {{100, 0.2f}, {200, 0.8f}, {300, 0.8f}, {400, 0.3f}}, 500,
                  {{0, 100, false, 0, 0, 0.2f, 0}, {100, 200, true, 100, 200, 0.2f, 0.8f},
                   {200, 300, false, 0, 0, 0.8f, 0}, {300, 400, true, 300, 400, 0.8f, 0.3f},
                   {400, 500, false, 0, 0, 0.3f, 0}},
                  {50, 75, 100, 500});
  run_single_case("step", {{100, 0.2f}, {200, 0.2f}, {200, 0.8f}, {300, 0.8f}}, 400,
                  {{0, 200, false, 0, 0, 0.2f, 0}, {200, 400, false, 0, 0, 0.8f, 0}}, {50, 75, 100, 400});
  run_single_case("only_step", {{100, 0.2f}, {100, 0.8f}}, 200,
                  {{0, 100, false, 0, 0, 0.2f, 0}, {100, 200, false, 0, 0, 0.8f, 0}}, {50, 75, 100, 200});
  run_single_case("one_point", {{100, 0.2f}}, 200, {{0, 200, false, 0, 0, 0.2f, 0}}, {50, 75, 100, 200});

  // vector (:187-219) and matrix (:221-257) against sums of Single interpolators
  const std::vector<std::vector<float>> a{{0.0f, 0.3f}, {0.5f, 0.0f}, {0.4f, 1.0f}};
  const std::vector<std::vector<float>> b{{0.6f, 0.0f}, {0.0f, 0.7f}, {1.0f, 0.2f}};
  const long len = 300;
  std::vector<Vec> x{random_vec(len, 5), random_vec(len, 6), random_vec(len, 7)};
  auto single = [&](int in, float s, float e) {
    GainInterpolator<LinearInterpSingle> gi;
    gi.interp_points.emplace_back(100, s);
    gi.interp_points.emplace_back(200, e);
    Vec out(len);
    const float *ip = x[in].data();
    float *op = out.data();
    gi.process(0, len, &ip, &op);
    return out;
  };
  {
    GainInterpolator<LinearInterpVector> gi;
    gi.interp_points.emplace_back(100, a[2]);
    gi.interp_points.emplace_back(200, b[2]);
    Vec o0(len), o1(len);
    const float *ip = x[2].data();
    float *op[2] = {o0.data(), o1.data()};
    gi.process(0, len, &ip, op);
    CHECK(o0 == single(2, a[2][0], b[2][0]));  // 1 -> N is bit-exact
    CHECK(o1 == single(2, a[2][1], b[2][1]));
  }
  for (int strict = 0; strict < 2; strict++) {
    hip::default_context().set_strict(strict != 0);
    GainInterpolator<LinearInterpMatrix> gi;
    gi.interp_points.emplace_back(100, a);
    gi.interp_points.emplace_back(200, b);
    Vec o0(len), o1(len);
    const float *ip[3] = {x[0].data(), x[1].data(), x[2].data()};
    float *op[2] = {o0.data(), o1.data()};
    gi.process(0, len, ip, op);
    Vec w0(len, 0.0f), w1(len, 0.0f);
    for (int in = 0; in < 3; in++) {
      const Vec s0 = single(in, a[in][0], b[in][0]), s1 = single(in, a[in][1], b[in][1]);
      for (long i = 0; i < len; i++) {
        w0[i] += s0[i];
        w1[i] += s1[i];
      }
    }
    if (strict) {
      CHECK(o0 == w0);
      CHECK(o1 == w1);
    } else {
      CHECK(rel_rms(o0, w0) <= 1e-6);
      CHECK(rel_rms(o1, w1) <= 1e-6);
    }
  }
  hip::default_context().set_strict(false);
  {
    GainInterpolator<LinearInterpSingle> gi;  // empty curve: defined as invalid_argument
    Vec xx(10), oo(10);
    const float *ip = xx.data();
    float *op = oo.data();
    bool threw = false;
    try {
      gi.process(0, 10, &ip, &op);
    } catch (const ear::invalid_argument &) {
      threw = true;
    }
    CHECK(threw);
  }
}

// ---- BlockConvolver (reference tests/block_convolver_tests.cpp) ----------------------------
static Vec sparse_random(size_t len, size_t nonzero, unsigned seed) {
  std::mt19937 g(1000 + seed);
  Vec v(len, 0.0f);
  for (size_t i = 0; i < nonzero; i++) {
    const size_t at = g() % len;
    v[at] = (float)((double)g() / 4294967296.0);
  }
  return v;
}

struct ConvCase {
  size_t B, nblocks;
  std::vector<Vec> irs;
  int initial;
  std::vector<int> ir_for_block;
  bool null_for_zeros;
  Vec input;
  // every Filter handed to the convolver is a temporary that dies right after the call, as libear
  // allows (its queue holds shared_ptrs, src/dsp/block_convolver_impl.hpp:154-167)
  bool temporaries = false;
};

static void run_conv_case(const char *name, ConvCase c) {
  using namespace ear::dsp::block_convolver;
  const size_t len = c.B * c.nblocks;
  // brute force: per IR, fade the input per block, convolve, mix (:83-116)
  std::vector<double> want(len, 0.0);
  for (size_t i = 0; i < c.irs.size(); i++) {
    std::vector<double> xin(len, 0.0);
    for (size_t blk = 0; blk < c.nblocks; blk++) {
      const bool last = (int)i == (blk == 0 ? c.initial : c.ir_for_block[blk - 1]);
      const bool cur = (int)i == c.ir_for_block[blk];
      for (size_t j = 0; j < c.B; j++) {
        const float v = c.input[blk * c.B + j];
        const float av = (float)j * (1.0f / (float)c.B);
        if (last && cur) xin[blk * c.B + j] = v;
        else if (cur) xin[blk * c.B + j] = av * v;
        else if (last) xin[blk * c.B + j] = (1.0f - av) * v;
      }
    }
    for (size_t n = 0; n < len; n++) {
      double acc = 0;
      for (size_t j = 0; j < c.irs[i].size() && j <= n; j++) acc += xin[n - j] * c.irs[i][j];
      want[n] += acc;
    }
  }
  Context ctx(c.B, get_fft_kiss<float>());  // (libear's accessor: the device transform behind it)
  std::vector<Filter> filters;
  size_t max_blocks = 0;
  for (auto &ir : c.irs) {
    filters.emplace_back(ctx, ir.size(), ir.data());
    max_blocks = std::max(max_blocks, filters.back().num_blocks());
  }
  BlockConvolver conv(ctx, max_blocks);
  if (c.initial >= 0) {
    if (c.temporaries) conv.set_filter(Filter(ctx, c.irs[c.initial].size(), c.irs[c.initial].data()));
    else conv.set_filter(filters[c.initial]);
  }
  if (c.temporaries) filters.clear();
  Vec out(len, 0.0f);
  for (size_t blk = 0; blk < c.nblocks; blk++) {
    const int cur = c.ir_for_block[blk], last = blk == 0 ? c.initial : c.ir_for_block[blk - 1];
    if (cur != last) {
      if (cur < 0) conv.fade_down();
      else if (c.temporaries) conv.crossfade_filter(Filter(ctx, c.irs[cur].size(), c.irs[cur].data()));
      else conv.crossfade_filter(filters[cur]);
    }
    bool zero = true;
    for (size_t j = 0; j < c.B; j++) zero = zero && c.input[blk * c.B + j] == 0.0f;
    conv.process(c.null_for_zeros && zero ? nullptr : &c.input[blk * c.B], &out[blk * c.B]);
  }
  double worst = 0;
  for (size_t i = 0; i < len; i++) worst = std::max(worst, std::fabs(out[i] - want[i]));
  if (!(worst < 1e-6)) std::printf("  conv case %s: max error %g\n", name, worst);
  CHECK(worst < 1e-6);  // reference max_error, block_convolver_tests.cpp:77
}

static void test_block_convolver() {
  using namespace ear::dsp::block_convolver;
  {
    Context ctx(512, get_fft_hip());
    const Vec coeff = sparse_random(2000, 100, 0);
    CHECK(Filter(ctx, 1, coeff.data()).num_blocks() == 1);
    CHECK(Filter(ctx, 511, coeff.data()).num_blocks() == 1);
    CHECK(Filter(ctx, 512, coeff.data()).num_blocks() == 1);
    CHECK(Filter(ctx, 513, coeff.data()).num_blocks() == 2);
  }
  run_conv_case("single_block", {512, 1, {sparse_random(100, 10, 1)}, 0, {0}, false, sparse_random(512, 200, 0)});
  run_conv_case("two_blocks", {512, 2, {sparse_random(1536, 20, 1)}, 0, {0, 0}, false, sparse_random(1024, 300, 0)});
  run_conv_case("fade_once", {512, 3, {sparse_random(100, 10, 1), sparse_random(512, 10, 2)}, 0, {0, 1, 1}, false,
                              sparse_random(1536, 300, 0)});
  run_conv_case("fade_to_silence", {512, 3, {sparse_random(512, 10, 1)}, 0, {0, -1, -1}, false,
                                    sparse_random(1536, 300, 0)});
  run_conv_case("fade_from_silence", {512, 3, {sparse_random(512, 10, 1)}, -1, {-1, 0, 0}, false,
                                      sparse_random(1536, 300, 0)});
  {
    Vec in = sparse_random(512 * 5, 300, 0);
    std::fill(in.begin() + 512, in.begin() + 512 * 4, 0.0f);
    run_conv_case("zero_input_blocks", {512, 5, {sparse_random(1024, 20, 1)}, 0, {0, 0, 0, 0, 0}, false, in});
    run_conv_case("null_input_blocks", {512, 5, {sparse_random(1024, 20, 1)}, 0, {0, 0, 0, 0, 0}, true, in});
  }
  run_conv_case("lots_of_filters",
                {512, 9, {sparse_random(1024, 20, 1), sparse_random(1536, 20, 2), sparse_random(512, 20, 3),
                          sparse_random(2048, 20, 4)},
                 0, {0, 1, 2, 3, 3, 2, 2, 1, 0}, false, sparse_random(512 * 9, 500, 0)});
  {
    // a filter change on EVERY block with 6 partitions, each Filter a temporary: the old filters are
    // still queued (fading out over 6 blocks) long after their handles are gone
    ConvCase c{256, 14, {}, 0, {}, false, sparse_random(256 * 14, 400, 0)};
    for (int i = 0; i < 7; i++) c.irs.push_back(sparse_random(256 * 6 - 17 * i, 25, 10 + i));
    for (int b = 0; b < 14; b++) c.ir_for_block.push_back((b + 1) % 7);
    c.temporaries = true;
    run_conv_case("temporary_filters_every_block", c);
    c.ir_for_block[5] = -1;  // ... one of them a fade to silence and back
    run_conv_case("temporary_filters_with_silence", c);
  }
  {
    Context c512(512, get_fft_hip()), c256(256, get_fft_hip());
    BlockConvolver conv(c512, 1);
    const Vec ones(600, 1.0f);
    bool threw = false;
    try {
      conv.set_filter(Filter(c256, 10, ones.data()));
    } catch (const ear::invalid_argument &) {
      threw = true;
    }
    CHECK(threw);
    threw = false;
    try {
      conv.crossfade_filter(Filter(c512, 513, ones.data()));
    } catch (const ear::invalid_argument &e) {
      threw = std::string(e.what()).find("too many blocks") != std::string::npos;
    }
    CHECK(threw);
  }
}

// ---- DelayBuffer / VariableBlockSizeAdapter ---------------------------------------------------
static void test_delay_and_adapter() {
  {
    const int delay = 128, nch = 5, sizes[3] = {64, 128, 256}, total = 448;
    DelayBuffer db(nch, delay);
    CHECK(db.get_delay() == delay);
    std::vector<Vec> in(nch), out(nch, Vec(total, 0.0f));
    for (int c = 0; c < nch; c++) in[c] = random_vec(total, 10 + c);
    int ofs = 0;
    for (int n : sizes) {
      std::vector<const float *> ip(nch);
      std::vector<float *> op(nch);
      for (int c = 0; c < nch; c++) {
        ip[c] = in[c].data() + ofs;
        op[c] = out[c].data() + ofs;
      }
      db.process(n, ip.data(), op.data());
      ofs += n;
    }
    bool ok = true;
    for (int c = 0; c < nch; c++)
      for (int i = 0; i < total; i++) ok = ok && out[c][i] == (i < delay ? 0.0f : in[c][i - delay]);
    CHECK(ok);
  }
  {
    const size_t B = 512, nin = 2, nout = 4;
    auto do_process = [&](size_t n, const float *const *in, float *const *out) {
      for (size_t i = 0; i < n; i++) {
        out[0][i] = in[0][i] * 2.0f;
        out[1][i] = in[1][i] * 3.0f;
        out[2][i] = in[0][i] * 4.0f;
        out[3][i] = in[1][i] * 5.0f;
      }
    };
    VariableBlockSizeAdapter adapter(B, nin, nout,
                                     [&](const float *const *i, float *const *o) { do_process(B, i, o); });
    CHECK(adapter.get_delay() == (int)B);
    const size_t sizes[5] = {0, 512, 1024, 300, 500}, total = 2336;
    std::vector<Vec> in{random_vec(total, 20), random_vec(total, 21)};
    std::vector<Vec> want(nout, Vec(total, 0.0f)), out(nout, Vec(total, -1.0f));
    {
      const float *ip[2] = {in[0].data(), in[1].data()};
      float *op[4] = {want[0].data() + B, want[1].data() + B, want[2].data() + B, want[3].data() + B};
      do_process(total - B, ip, op);
    }
    size_t ofs = 0;
    for (size_t n : sizes) {
      const float *ip[2] = {in[0].data() + ofs, in[1].data() + ofs};
      float *op[4] = {out[0].data() + ofs, out[1].data() + ofs, out[2].data() + ofs, out[3].data() + ofs};
      adapter.process(n, ip, op);
      ofs += n;
    }
    CHECK(out == want);
  }
}

// ---- the fused ObjectsRenderer == the composition of the drop-in components --------------------
static void test_objects_renderer() {
  const std::vector<std::string> names{"M+030", "M-030", "M+000", "LFE1", "M+110", "M-110"};
  const size_t M = 9, N = names.size(), B = 512, T = 3, total = B * T;
  const auto dec = designDecorrelators(names);
  const int delay = decorrelatorCompensationDelay();
  CHECK(delay == 255 && dec.size() == N && dec[0].size() == 512);
  {  // libear's own signatures (include/ear/decorrelate.hpp:16-27): a Layout in, float by default, double on request
    const Layout l050 = getLayout("0+5+0");
    CHECK(designDecorrelators(l050) == dec);
    CHECK(designDecorrelator(l050, 4) == dec[4]);
    const std::vector<double> d64 = designDecorrelator<double>(l050, 0);
    bool same = d64.size() == 512;
    for (size_t i = 0; same && i < 512; i++) same = (float)d64[i] == dec[0][i];
    CHECK(same);
    // M+030 of 4+5+0 without LFE has one name before it (tests/decorrelate_tests.cpp:35-44)
    const Layout l450 = getLayout("4+5+0").withoutLfe();
    const std::vector<double> basic1 = designDecorrelatorBasic(1, 512);
    CHECK(designDecorrelator<double>(l450, (size_t)l450.indexForName("M+030")) == basic1);
  }
  std::mt19937 g(99);
  auto rnd = [&] { return (float)((double)g() / 4294967296.0); };
  std::vector<Vec> in(M);
  for (size_t m = 0; m < M; m++) in[m] = random_vec(total, 40 + (unsigned)m);
  std::vector<int64_t> times;
  for (size_t t = 0; t <= T; t++) times.push_back((int64_t)(t * B));
  std::vector<std::vector<std::vector<float>>> dgain(M), fgain(M);
  for (size_t m = 0; m < M; m++)
    for (size_t t = 0; t <= T; t++) {
      Vec d(N), f(N);
      for (size_t c = 0; c < N; c++) {
        d[c] = rnd();
        f[c] = rnd();
      }
      dgain[m].push_back(d);
      fgain[m].push_back(f);
    }
  // fused
  std::vector<Vec> out(N, Vec(total));
  {
    ObjectsRenderer r(M, N, B, dec, delay, T);
    for (size_t m = 0; m < M; m++) r.set_object_points(m, times, dgain[m], fgain[m]);
    std::vector<const float *> ip(M);
    std::vector<float *> op(N);
    for (size_t m = 0; m < M; m++) ip[m] = in[m].data();
    for (size_t c = 0; c < N; c++) op[c] = out[c].data();
    r.process(T, ip.data(), op.data());
  }
  // composition of the drop-in components, block by block (docs/dsp.rst:40-71)
  using namespace ear::dsp::block_convolver;
  std::vector<Vec> want(N, Vec(total));
  {
    Context ctx(B, get_fft_hip());
    std::vector<std::unique_ptr<BlockConvolver>> convs;
    for (size_t c = 0; c < N; c++)
      convs.emplace_back(new BlockConvolver(ctx, Filter(ctx, dec[c].size(), dec[c].data())));
    DelayBuffer db(N, delay);
    std::vector<GainInterpolator<LinearInterpVector>> gd(M), gf(M);
    for (size_t m = 0; m < M; m++)
      for (size_t t = 0; t <= T; t++) {
        gd[m].interp_points.emplace_back((long)times[t], dgain[m][t]);
        gf[m].interp_points.emplace_back((long)times[t], fgain[m][t]);
      }
    std::vector<Vec> dbus(N, Vec(B)), fbus(N, Vec(B)), tmp(N, Vec(B)), decd(N, Vec(B)), deld(N, Vec(B));
    for (size_t t = 0; t < T; t++) {
      for (size_t c = 0; c < N; c++) {
        std::fill(dbus[c].begin(), dbus[c].end(), 0.0f);
        std::fill(fbus[c].begin(), fbus[c].end(), 0.0f);
      }
      std::vector<float *> tp(N);
      for (size_t c = 0; c < N; c++) tp[c] = tmp[c].data();
      for (size_t m = 0; m < M; m++) {
        const float *ip = in[m].data() + t * B;
        gd[m].process((long)(t * B), B, &ip, tp.data());
        for (size_t c = 0; c < N; c++)
          for (size_t i = 0; i < B; i++) dbus[c][i] += tmp[c][i];
        gf[m].process((long)(t * B), B, &ip, tp.data());
        for (size_t c = 0; c < N; c++)
          for (size_t i = 0; i < B; i++) fbus[c][i] += tmp[c][i];
      }
      std::vector<const float *> dp(N);
      std::vector<float *> lp(N);
      for (size_t c = 0; c < N; c++) {
        convs[c]->process(fbus[c].data(), decd[c].data());
        dp[c] = dbus[c].data();
        lp[c] = deld[c].data();
      }
      db.process(B, dp.data(), lp.data());
      for (size_t c = 0; c < N; c++)
        for (size_t i = 0; i < B; i++) want[c][t * B + i] = decd[c][i] + deld[c][i];
    }
  }
  double d = 0, n = 0;
  for (size_t c = 0; c < N; c++)
    for (size_t i = 0; i < total; i++) {
      d += ((double)out[c][i] - want[c][i]) * ((double)out[c][i] - want[c][i]);
      n += (double)want[c][i] * want[c][i];
    }
  const double err = std::sqrt(d / n);
  std::printf("  fused renderer vs composed drop-in components: rel RMS %.3g\n", err);
  CHECK(err <= 1e-6);
}

// ---- PtrAdapter (reference include/ear/dsp/ptr_adapter.hpp:10-40; used by every reference DSP test) ----
// a minimal column-major matrix with the two members set_eigen needs (cols(), col(c).data())
struct ColMajor {
  size_t rows_, cols_;
  Vec data_;
  ColMajor(size_t r, size_t c) : rows_(r), cols_(c), data_(r * c, 0.0f) {}
  struct Col {
    float *p;
    float *data() const { return p; }
  };
  struct ConstCol {
    const float *p;
    const float *data() const { return p; }
  };
  size_t cols() const { return cols_; }
  Col col(size_t c) { return Col{data_.data() + c * rows_}; }
  ConstCol col(size_t c) const { return ConstCol{data_.data() + c * rows_}; }
  float &at(size_t r, size_t c) { return data_[c * rows_ + r]; }
};

static void test_ptr_adapter() {
  const size_t n = 300, nch = 3, delay = 40;
  ColMajor in(n, nch), out(n, nch);
  for (size_t c = 0; c < nch; c++)
    for (size_t i = 0; i < n; i++) in.at(i, c) = (float)(c * 1000 + i);
  PtrAdapterConst in_p(nch);
  PtrAdapter out_p(nch);
  CHECK(in_p.size() == nch);
  // the reference's idiom (tests/delay_buffer_tests.cpp:17-24): re-point the adapters at an offset
  // for every call, process in pieces
  DelayBuffer db(nch, delay);
  size_t ofs = 0;
  for (size_t len : {100u, 7u, 193u}) {
    const ColMajor &cin = in;
    in_p.set_eigen(cin, ofs);
    out_p.set_eigen(out, ofs);
    CHECK(in_p.ptrs()[1] == in.data_.data() + n + ofs);
    db.process(len, in_p.ptrs(), out_p.ptrs());
    ofs += len;
  }
  bool ok = true;
  for (size_t c = 0; c < nch; c++)
    for (size_t i = 0; i < n; i++) ok = ok && out.at(i, c) == (i < delay ? 0.0f : in.at(i - delay, c));
  CHECK(ok);
  // wrong channel count: ear_assert -> internal_error (ptr_adapter.hpp:19-20)
  bool threw = false;
  try {
    ColMajor two(n, 2);
    out_p.set_eigen(two);
  } catch (const ear::internal_error &) {
    threw = true;
  }
  CHECK(threw);
  // the Eigen-free spelling gives the same pointers
  PtrAdapter planar(nch);
  planar.set_planar(out.data_.data(), n, 5);
  out_p.set_eigen(out, 5);
  for (size_t c = 0; c < nch; c++) CHECK(planar.ptrs()[c] == out_p.ptrs()[c]);
  // GainInterpolator driven through adapters (reference tests/gain_interpolator_tests.cpp:42-52)
  GainInterpolator<LinearInterpVector> interp;
  interp.interp_points.emplace_back(0, std::vector<float>{1.0f, 0.0f, 0.5f});
  interp.interp_points.emplace_back(100, std::vector<float>{0.0f, 1.0f, 0.5f});
  ColMajor x(200, 1), y(200, nch);
  for (size_t i = 0; i < 200; i++) x.at(i, 0) = 1.0f;
  PtrAdapterConst xp(1);
  PtrAdapter yp(nch);
  const ColMajor &cx = x;
  xp.set_eigen(cx);
  yp.set_eigen(y);
  interp.process(0, 200, xp.ptrs(), yp.ptrs());
  ok = true;
  for (size_t i = 0; i < 200; i++) {
    const float p = i < 100 ? (float)i * (1.0f / 100.0f) : 1.0f;
    ok = ok && y.at(i, 0) == (i < 100 ? (1.0f - p) * 1.0f + p * 0.0f : 0.0f) && y.at(i, 2) == 0.5f;
  }
  CHECK(ok);
}

static void test_layout_names() {
  // "decorrelators for 4+5+0 without LFE": M+030 gets filter id 1 (reference
  // tests/decorrelate_tests.cpp:35-44), the caller never spells a channel name
  const auto f = designDecorrelators("4+5+0", true);
  CHECK(f.size() == 9);
  const auto basic = designDecorrelatorBasic(1, 512);
  bool ok = f.size() == 9 && f[0].size() == 512;
  for (size_t i = 0; ok && i < 512; i++) ok = f[0][i] == (float)basic[i];
  CHECK(ok);
  CHECK(designDecorrelators("9+10+3").size() == 24);
  bool threw = false;
  try {
    designDecorrelators("1+2+3");
  } catch (const ear::unknown_layout &e) {  // (tests/bs2051_tests.cpp:25; libear's message)
    threw = std::string(e.what()) == "unknown layout: 1+2+3";
  }
  CHECK(threw);
  threw = false;
  try {
    getLayout("wat");
  } catch (const std::invalid_argument &) {  // (unknown_layout is a std::invalid_argument, as in libear)
    threw = true;
  }
  CHECK(threw);
}

// ---- GainCalculatorObjects (reference tests/gain_calculator_objects_tests.cpp:74-160) ----------------
static void test_gain_calculator_objects() {
  const Layout layout = getLayout("4+7+0").withoutLfe();
  CHECK(layout.channels().size() == 11);
  GainCalculatorObjects calc(layout);
  auto only = [&](const std::vector<float> &g, const char *name, double value) {
    bool ok = g.size() == layout.channels().size();
    for (size_t i = 0; ok && i < g.size(); i++)
      ok = std::fabs(g[i] - (layout.channels()[i].name() == name ? value : 0.0)) < 1e-6;
    return ok;
  };
  std::vector<float> direct(11), diffuse(11);  // (as libear's tests size them, tests/gain_calculator_objects_tests.cpp:80-81)
  ObjectsTypeMetadata otm;
  otm.position = PolarPosition(0.0, 0.0, 1.0);
  {  // an output vector of the wrong size is refused, not resized (include/ear/helpers/output_gains.hpp:40-43)
    std::vector<float> small(10), none;
    bool threw = false;
    try {
      calc.calculate(otm, small, diffuse);
    } catch (const ear::invalid_argument &e) {
      threw = std::string(e.what()).find("incorrect size for output vector") != std::string::npos;
    }
    CHECK(threw);
    threw = false;
    try {
      calc.calculate(otm, direct, none);
    } catch (const ear::invalid_argument &) {
      threw = true;
    }
    CHECK(threw);
    // a channel without explicit ranges stands where it really stands (src/layout.cpp:24-33)
    Channel moved("X", PolarPosition(12.0, 3.0, 1.0), PolarPosition(10.0, 0.0, 1.0));
    CHECK(moved.azimuthRange() == std::make_pair(12.0, 12.0) && moved.elevationRange() == std::make_pair(3.0, 3.0));
  }
  calc.calculate(otm, direct, diffuse);
  CHECK(only(direct, "M+000", 1.0) && only(diffuse, "", 0.0));
  otm.position = PolarPosition(30.0, 0.0, 1.0);
  calc.calculate(otm, direct, diffuse);
  CHECK(only(direct, "M+030", 1.0));
  otm.position = PolarPosition(45.0, 30.0, 1.0);
  calc.calculate(otm, direct, diffuse);
  CHECK(only(direct, "U+045", 1.0));
  otm.position = PolarPosition(0.0, 0.0, 1.0);
  otm.diffuse = 0.5;
  calc.calculate(otm, direct, diffuse);
  CHECK(only(direct, "M+000", std::sqrt(0.5)) && only(diffuse, "M+000", std::sqrt(0.5)));
  otm.diffuse = 1.0;
  calc.calculate(otm, direct, diffuse);
  CHECK(only(direct, "", 0.0) && only(diffuse, "M+000", 1.0));
  otm.diffuse = 0.0;
  otm.gain = 0.5;
  calc.calculate(otm, direct, diffuse);
  CHECK(only(direct, "M+000", 0.5));
  // "not implemented" (:134-160)
  auto refuses = [&](ObjectsTypeMetadata m) {
    try {
      calc.calculate(m, direct, diffuse);
    } catch (const ear::not_implemented &) {
      return true;
    }
    return false;
  };
  ObjectsTypeMetadata base;
  ObjectsTypeMetadata m = base;
  m.cartesian = true;
  CHECK(refuses(m));
  m = base;
  m.position = CartesianPosition(0.0, 1.0, 0.0);
  CHECK(refuses(m));
  m = base;
  m.objectDivergence = PolarObjectDivergence(0.5);  // (the reference test's own lines, :142-149)
  CHECK(refuses(m));
  m = base;
  m.objectDivergence = CartesianObjectDivergence(0.5);
  CHECK(refuses(m));
  m = base;
  m.channelLock.flag = true;
  CHECK(refuses(m));
  m = base;
  m.zoneExclusion.zones.push_back(PolarExclusionZone{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, ""});
  CHECK(refuses(m));
  m = base;
  m.screenRef = true;
  CHECK(refuses(m));
  CHECK(!refuses(base));
  CHECK(!base.referenceScreen.isCartesian && base.referenceScreen.polar.widthAzimuth == 58.0 &&
        getDefaultScreen().polar.aspectRatio == 1.78);  // (include/ear/screen.hpp, src/screen.cpp:4-6)
  // extent (tests/extent_tests.cpp:116-138): unit power, the velocity vector stays in the median plane and
  // points forwards (4+7+0 has no layer below, so it tilts up a little), more loudspeakers than a point source
  m = base;
  m.position = PolarPosition(0.0, 0.0, 1.0);
  m.width = 20.0;
  m.height = 10.0;
  CHECK(!refuses(m));
  {
    double power = 0.0, vx = 0.0, vy = 0.0, vz = 0.0;
    int nz = 0;
    for (size_t c = 0; c < direct.size(); c++) {
      const PolarPosition pp = layout.channels()[c].polarPosition();
      const double a = -pp.azimuth * M_PI / 180.0, e = pp.elevation * M_PI / 180.0;
      power += (double)direct[c] * direct[c];
      vx += direct[c] * std::sin(a) * std::cos(e), vy += direct[c] * std::cos(a) * std::cos(e), vz += direct[c] * std::sin(e);
      nz += direct[c] > 1e-6f;
    }
    const double vn = std::sqrt(vx * vx + vy * vy + vz * vz);
    CHECK(std::fabs(power - 1.0) < 1e-5);
    CHECK(std::fabs(vx / vn) < 1e-5 && vy / vn > 0.98 && vz / vn >= 0.0);
    CHECK(nz >= 3);
  }
  m.width = m.height = 360.0;  // everywhere: every loudspeaker carries something
  calc.calculate(m, direct, diffuse);
  {
    bool all = true;
    for (float g : direct) all = all && g > 0.01f;
    CHECK(all);
  }
  m.width = 30.0, m.height = 0.0, m.depth = 0.5;
  CHECK(!refuses(m));
  // loudspeakers off their nominal positions (Channel::polarPosition): a source at a real position plays from
  // that loudspeaker alone; the screen loudspeakers' ranges (tests/point_source_panner_tests.cpp:522-551)
  {
    Layout moved = getLayout("4+7+0").withoutLfe();
    for (auto &c : moved.channels()) {
      PolarPosition q = c.polarPosition();
      if (c.name() == "M+030") q.azimuth = 26.0, q.elevation = 2.0;
      if (c.name() == "U-045") q.azimuth = -50.0, q.elevation = 33.0;
      c.polarPosition(q);
    }
    // (BS.2051 allows M+030 at 30..45 degrees on the horizontal plane and U-045 at -45..-30, 30..55 up: these
    // positions are outside, which Layout::checkPositions reports and the panner does not mind — src/layout.cpp:54-75,
    // tests/bs2051_tests.cpp:27-40)
    std::vector<std::string> complaints;
    getLayout("4+7+0").checkPositions([&](const std::string &msg) { complaints.push_back(msg); });
    CHECK(complaints.empty());
    moved.checkPositions([&](const std::string &msg) { complaints.push_back(msg); });
    CHECK(complaints.size() == 3 && complaints[0] == "M+030: azimuth 26 out of range [30, 45]" &&
          complaints[1] == "M+030: elevation 2 out of range [0, 0]" &&
          complaints[2] == "U-045: azimuth -50 out of range [-45, -30]");
    CHECK(moved.channelWithName("U-045").azimuthRange() == std::make_pair(-45.0, -30.0));
    CHECK(moved.channelWithName("U-045").elevationRange() == std::make_pair(30.0, 55.0));
    CHECK(moved.nominalPositions()[0].azimuth == 30.0 && moved.positions()[0].azimuth == 26.0);
    GainCalculatorObjects mc(moved);
    ObjectsTypeMetadata mm;
    mm.position = PolarPosition(26.0, 2.0, 1.0);
    mc.calculate(mm, direct, diffuse);
    CHECK(only(direct, "M+030", 1.0));
    mm.position = PolarPosition(-50.0, 33.0, 1.0);
    mc.calculate(mm, direct, diffuse);
    CHECK(only(direct, "U-045", 1.0));
    mm.position = PolarPosition(30.0, 0.0, 1.0);  // the nominal place is now between loudspeakers
    mc.calculate(mm, direct, diffuse);
    CHECK(!only(direct, "M+030", 1.0));
    Layout screen = getLayout("4+9+0").withoutLfe();
    for (auto &c : screen.channels())
      if (c.name() == "M+SC") c.polarPosition(PolarPosition(40.0, 0.0, 1.0));
    bool threw_ni = false, threw_ia = false;
    try {
      GainCalculatorObjects sc(screen);
    } catch (const ear::not_implemented &) {
      threw_ni = true;
    }
    for (auto &c : screen.channels())
      if (c.name() == "M+SC") c.polarPosition(PolarPosition(30.0, 0.0, 1.0));
    try {
      GainCalculatorObjects sc(screen);
    } catch (const ear::invalid_argument &) {
      threw_ia = true;
    }
    CHECK(threw_ni && threw_ia);
  }
  // with the LFE channel kept, its column is zero (gain_calculator_objects.cpp:50-52); a batch in one launch
  GainCalculatorObjects full(getLayout("4+7+0"));
  std::vector<ObjectsTypeMetadata> batch(3);
  batch[1].position = PolarPosition(-30.0, 0.0, 1.0);
  batch[2].position = PolarPosition(110.0, 20.0, 2.0);
  batch[2].diffuse = 0.3;
  std::vector<std::vector<float>> bd, bf;
  full.calculate(batch, bd, bf);
  CHECK(bd.size() == 3 && bd[0].size() == 12 && bd[0][3] == 0.0f && bd[2][3] == 0.0f);
  double power = 0;
  for (size_t c = 0; c < 12; c++) power += (double)bd[2][c] * bd[2][c] + (double)bf[2][c] * bf[2][c];
  CHECK(std::fabs(power - 1.0) < 1e-6);
}

// ---- GainCalculatorHOA (reference tests/gain_calculator_hoa_tests.cpp:39-80) ----------------------------
static void test_gain_calculator_hoa() {
  GainCalculatorHOA gc(getLayout("0+5+0"));
  HOATypeMetadata tm;
  tm.orders = {0, 1, 1, 1};
  tm.degrees = {0, -1, 0, 1};
  std::vector<std::vector<double>> gains(4, std::vector<double>(6));
  gc.calculate(tm, gains);
  bool lfe_zero = true;
  double w_sum = 0;
  for (int c = 0; c < 4; c++) lfe_zero = lfe_zero && gains[c][3] == 0.0;
  for (int s = 0; s < 6; s++) w_sum += gains[0][s];
  CHECK(lfe_zero);
  CHECK(w_sum > 0.5);  // the omnidirectional component feeds every loudspeaker with the same sign
  CHECK(gains[3][0] > 0.0 && gains[3][1] > 0.0 && gains[3][4] < 0.0);  // X (front-back): front positive, rear negative
  CHECK(gains[1][0] > 0.0 && gains[1][1] < 0.0);                        // Y (left-right): left positive, right negative
  auto throws_invalid = [&](HOATypeMetadata m) {
    try {
      gc.calculate(m, gains);
    } catch (const ear::invalid_argument &) {
      return true;
    }
    return false;
  };
  HOATypeMetadata m = tm;
  m.degrees.pop_back();
  CHECK(throws_invalid(m));
  m = tm;
  m.orders[0] = -1;
  CHECK(throws_invalid(m));
  m = tm;
  m.degrees[3] = 2;
  CHECK(throws_invalid(m));
  m = tm;
  m.degrees[3] = -2;
  CHECK(throws_invalid(m));
  m = tm;
  m.normalization = "foo";  // invalid ADM metadata: adm_error (src/hoa/gain_calculator_hoa.cpp:36-39)
  bool adm = false;
  try {
    gc.calculate(m, gains);
  } catch (const ear::adm_error &e) {
    adm = std::string(e.what()) == "ADM error: unknown normalization type: 'foo'";
  }
  CHECK(adm);
  CHECK(!throws_invalid(tm));
  // warnings (tests/gain_calculator_hoa_tests.cpp:9-37): what is ignored is reported through the callback
  {
    std::vector<ear::Warning> warnings;
    auto collect = [&](const ear::Warning &w) { warnings.push_back(w); };
    m = tm;
    m.screenRef = true;
    gc.calculate(m, gains, collect);
    CHECK(warnings.size() == 1 && warnings[0].code == ear::Warning::Code::HOA_SCREENREF_NOT_IMPLEMENTED);
    warnings.clear();
    m = tm;
    m.nfcRefDist = 1.0;
    gc.calculate(m, gains, collect);
    CHECK(warnings.size() == 1 && warnings[0].code == ear::Warning::Code::HOA_NFCREFDIST_NOT_IMPLEMENTED);
    warnings.clear();
    gc.calculate(tm, gains, collect);
    CHECK(warnings.empty());
  }
}

// ---- no heap allocation in any process() call once warmed up (the reference enforces it for
// GainInterpolator in tests/gain_interpolator_tests.cpp:89-96) ------------------------------------------
static void test_no_allocation_in_process() {
  using namespace ear::dsp::block_convolver;
  const size_t B = 512, n_in = 12, n_out = 6;
  std::vector<Vec> in(n_in), out(n_out, Vec(B * 64));
  for (size_t c = 0; c < n_in; c++) in[c] = random_vec(B * 64, 100 + (unsigned)c);
  std::vector<const float *> ip(n_in);
  std::vector<float *> op(n_out);
  for (size_t c = 0; c < n_in; c++) ip[c] = in[c].data();
  for (size_t c = 0; c < n_out; c++) op[c] = out[c].data();

  GainInterpolator<LinearInterpVector> vec_interp;
  vec_interp.interp_points.emplace_back(0, std::vector<float>(n_out, 0.25f));
  vec_interp.interp_points.emplace_back(700, std::vector<float>(n_out, 0.75f));
  GainInterpolator<LinearInterpMatrix> mat_interp;
  mat_interp.interp_points.emplace_back(0, std::vector<std::vector<float>>(n_in, std::vector<float>(n_out, 0.1f)));
  mat_interp.interp_points.emplace_back(900, std::vector<std::vector<float>>(n_in, std::vector<float>(n_out, 0.3f)));
  Context cctx(B, get_fft_hip());
  const Vec taps = random_vec(700, 5);
  Filter filt(cctx, taps.size(), taps.data());
  BlockConvolver conv(cctx, filt);
  DelayBuffer delay(n_out, 255);
  const auto dec = designDecorrelators("0+5+0");
  ObjectsRenderer block_renderer(n_in, n_out, B, dec, 255, 1), stream_renderer(n_in, n_out, B, dec, 255, 64);
  for (size_t m = 0; m < n_in; m++) {
    const std::vector<std::vector<float>> g = {std::vector<float>(n_out, 0.1f), std::vector<float>(n_out, 0.2f)};
    block_renderer.set_object_points(m, {0, 4096}, g, g);
    stream_renderer.set_object_points(m, {0, 4096}, g, g);
  }
  VariableBlockSizeAdapter adapter(B, n_in, n_out,
                                   [&](const float *const *i, float *const *o) { block_renderer.process(i, o); });
  auto all = [&](long t) {
    vec_interp.process(t, B, ip.data(), op.data());
    mat_interp.process(t, B, ip.data(), op.data());
    conv.process(in[0].data(), out[0].data());
    delay.process(B, ip.data(), op.data());
    block_renderer.process(ip.data(), op.data());
    stream_renderer.process(64, ip.data(), op.data());
    adapter.process(300, ip.data(), op.data());
    adapter.process(700, ip.data(), op.data());
  };
  for (long t = 0; t < 3; t++) all(t * (long)B);  // warm-up: staging buffers grow here
  g_heap_count = 0;
  g_heap_armed = true;
  for (long t = 3; t < 8; t++) all(t * (long)B);
  g_heap_armed = false;
  if (g_heap_count != 0) std::printf("  %ld heap allocations in steady-state process() calls\n", (long)g_heap_count);
  CHECK(g_heap_count == 0);
  // the watch itself works: an allocation of this program is seen
  g_heap_armed = true;
  { std::vector<int> v(1000); (void)v; }
  g_heap_armed = false;
  CHECK(g_heap_count >= 1);
}

// ---- channel buffers in device-reachable host memory (ear::hip::Context::alloc_host / register_host) ----
// the columns of ONE pinned matrix, addressed through PtrAdapter::set_planar-style evenly spaced pointers, take
// the renderer's short-call path without staging copies: the same bits as ordinary buffers
static void test_pinned_channel_buffers() {
  const size_t n_in = 12, n_out = 6, B = 512, T = 4;
  const auto dec = designDecorrelators("0+5+0");
  ObjectsRenderer plain(n_in, n_out, B, dec, 255, 1), pinned(n_in, n_out, B, dec, 255, 1);
  for (size_t m = 0; m < n_in; m++) {
    const std::vector<std::vector<float>> g = {std::vector<float>(n_out, 0.1f + 0.01f * m), std::vector<float>(n_out, 0.4f)};
    plain.set_object_points(m, {100, 1700}, g, g);
    pinned.set_object_points(m, {100, 1700}, g, g);
  }
  ear::hip::Context &hc = ear::hip::default_context();
  float *pin_in = hc.alloc_host(n_in * B), *pin_out = hc.alloc_host(n_out * B);
  std::vector<float> reg_out(n_out * B);  // ordinary memory, registered
  hc.register_host(reg_out.data(), reg_out.size() * sizeof(float));
  std::vector<Vec> in(n_in), out(n_out, Vec(B));
  for (size_t c = 0; c < n_in; c++) in[c] = random_vec(B * T, 300 + (unsigned)c);
  std::vector<const float *> ip(n_in), pp(n_in);
  std::vector<float *> op(n_out), qp(n_out), rp(n_out);
  for (size_t c = 0; c < n_in; c++) pp[c] = pin_in + c * B;
  for (size_t c = 0; c < n_out; c++) op[c] = out[c].data(), qp[c] = pin_out + c * B, rp[c] = reg_out.data() + c * B;
  for (size_t t = 0; t < T; t++) {
    for (size_t c = 0; c < n_in; c++) {
      ip[c] = in[c].data() + t * B;
      std::copy(ip[c], ip[c] + B, pin_in + c * B);
    }
    plain.process(ip.data(), op.data());
    pinned.process(pp.data(), (t & 1) ? rp.data() : qp.data());
    bool same = true;
    for (size_t c = 0; c < n_out; c++)
      for (size_t i = 0; i < B; i++) same = same && out[c][i] == ((t & 1) ? rp[c][i] : qp[c][i]);
    CHECK(same);
  }
  // the adapter with its FIFO in pinned memory (5-argument constructor) around the renderer == the plain adapter
  {
    ObjectsRenderer ra(n_in, n_out, B, dec, 255, 1), rb(n_in, n_out, B, dec, 255, 1);
    for (size_t m = 0; m < n_in; m++) {
      const std::vector<std::vector<float>> g = {std::vector<float>(n_out, 0.2f), std::vector<float>(n_out, 0.3f + 0.01f * m)};
      ra.set_object_points(m, {0, 900}, g, g);
      rb.set_object_points(m, {0, 900}, g, g);
    }
    VariableBlockSizeAdapter plain_ad(B, n_in, n_out, [&](const float *const *i, float *const *o) { ra.process(i, o); });
    VariableBlockSizeAdapter pinned_ad(B, n_in, n_out, [&](const float *const *i, float *const *o) { rb.process(i, o); }, hc);
    std::vector<Vec> oa(n_out, Vec(B * T)), ob(n_out, Vec(B * T));
    size_t ofs = 0;
    for (size_t k : {300u, 512u, 700u, 536u}) {
      std::vector<const float *> i2(n_in);
      std::vector<float *> a2(n_out), b2(n_out);
      for (size_t c = 0; c < n_in; c++) i2[c] = in[c].data() + ofs;
      for (size_t c = 0; c < n_out; c++) a2[c] = oa[c].data() + ofs, b2[c] = ob[c].data() + ofs;
      plain_ad.process(k, i2.data(), a2.data());
      pinned_ad.process(k, i2.data(), b2.data());
      ofs += k;
    }
    bool same = true, nonzero = false;
    for (size_t c = 0; c < n_out; c++)
      for (size_t i = 0; i < ofs; i++) same = same && oa[c][i] == ob[c][i], nonzero = nonzero || oa[c][i] != 0.0f;
    CHECK(same && nonzero);
  }
  hc.release_host(reg_out.data());
  hc.release_host(pin_out);
  hc.release_host(pin_in);
  bool threw = false;
  try {
    hc.release_host(pin_in);
  } catch (const ear::invalid_argument &) {
    threw = true;
  }
  CHECK(threw);
}

// libear's one run-time plugin point (include/ear/fft.hpp:54-62, include/ear/dsp/block_convolver.hpp:34): the
// device transform is accepted under both of its names, a foreign FFTImpl is refused — not dropped
static void test_fft_plugin_point() {
  struct ForeignFFT : FFTImpl<float> {
    std::shared_ptr<FFTPlan<float>> plan(size_t) const override { return nullptr; }
  } foreign;
  using ear::dsp::block_convolver::Context;
  bool threw = false;
  try {
    Context ctx(512, foreign);
  } catch (const ear::invalid_argument &) {
    threw = true;
  }
  CHECK(threw);
  bool ok = true;
  try {
    Context a(512, get_fft_hip());
    Context b(480, get_fft_kiss<float>());
    Context c(2 * 101, get_fft_hip());  // (a prime above 5: kissfft's generic butterfly, here too)
  } catch (const std::exception &e) {
    std::printf("  %s\n", e.what());
    ok = false;
  }
  CHECK(ok);
  CHECK(&get_fft_kiss<float>() == &get_fft_hip());
}

// the multi-GPU exchange through the mirror (ear::hip::Communicator), as far as one GPU goes: a one-rank
// communicator's reduce-scatter and gather are copies; buffers in device-reachable host memory
static void test_communicator_single_rank() {
  ear::hip::Context &hc = ear::hip::default_context();
  const int n_out = 10;
  const size_t stride = 256;
  ear::hip::Communicator comm(hc, 0, 1, ear::hip::Communicator::unique_id());
  CHECK(comm.padded_rows(n_out) == n_out);
  int lo = -1, hi = -1;
  comm.channel_range(n_out, lo, hi);
  CHECK(lo == 0 && hi == n_out);
  float *partial = hc.alloc_host(n_out * stride), *owned = hc.alloc_host(n_out * stride), *full = hc.alloc_host(n_out * stride);
  for (size_t i = 0; i < n_out * stride; i++) partial[i] = (float)i * 0.25f, owned[i] = full[i] = -1.0f;
  comm.exchange(0, partial, owned, n_out, stride);
  comm.gather(0, owned, full, n_out, stride, 0);
  comm.wait(0);
  hc.synchronize();
  bool same = true;
  for (size_t i = 0; i < n_out * stride; i++) same = same && owned[i] == partial[i] && full[i] == partial[i];
  CHECK(same);
  CHECK(comm.last_exchange_ms(0) >= 0.0);
  for (size_t i = 0; i < n_out * stride; i++) full[i] = -1.0f;
  comm.gather(1, owned, full, n_out, stride);  // all-gather form
  comm.wait(1);
  hc.synchronize();
  same = true;
  for (size_t i = 0; i < n_out * stride; i++) same = same && full[i] == partial[i];
  CHECK(same);
  hc.release_host(full);
  hc.release_host(owned);
  hc.release_host(partial);
}

// Run-time configuration: knobs are options of a context — set, read back, reset; unknown keys are refused with
// libear's invalid_argument; a gain kernel forced by option is the one that runs.
static void test_context_options() {
  hip::Context hc(0);
  int v = -1;
  CHECK(!hc.get_option("H2_TILE", v) || v >= 0);  // (set only when the environment had it at creation)
  hc.set_option("h2_tile", 256);
  CHECK(hc.get_option("EARHIP_H2_TILE", v) && v == 256);
  hc.reset_option("H2_TILE");
  CHECK(!hc.get_option("H2_TILE", v));
  bool refused = false;
  try {
    hc.set_option("NO_SUCH_KNOB", 1);
  } catch (const ear::invalid_argument &) {
    refused = true;
  }
  CHECK(refused);
  // the exact-f32 gain kernel by option: same scene, within 1e-6 of the default kernels' result
  const std::vector<std::string> names{"M+030", "M-030", "M+000", "LFE1", "M+110", "M-110"};
  const size_t M = 40, N = names.size(), B = 512, T = 2;
  const auto dec = designDecorrelators(names);
  std::vector<Vec> in(M);
  for (size_t m = 0; m < M; m++) in[m] = random_vec(B * T, 700 + (unsigned)m);
  std::vector<const float *> ip(M);
  for (size_t m = 0; m < M; m++) ip[m] = in[m].data();
  std::mt19937 g(5);
  std::vector<Vec> outs[2];
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) hc.set_option("MFMA", 1);
    ObjectsRenderer r(M, N, B, dec, decorrelatorCompensationDelay(), T, hc);
    std::mt19937 gg(5);
    for (size_t m = 0; m < M; m++) {
      std::vector<std::vector<float>> d(T + 1, Vec(N)), f(T + 1, Vec(N));
      std::vector<int64_t> times;
      for (size_t t = 0; t <= T; t++) {
        times.push_back((int64_t)(t * B));
        for (size_t c = 0; c < N; c++) d[t][c] = (float)((double)gg() / 4294967296.0), f[t][c] = (float)((double)gg() / 4294967296.0);
      }
      r.set_object_points(m, times, d, f);
    }
    outs[pass].assign(N, Vec(B * T));
    std::vector<float *> op(N);
    for (size_t c = 0; c < N; c++) op[c] = outs[pass][c].data();
    r.process(T, ip.data(), op.data());
  }
  double num = 0, den = 0;
  for (size_t c = 0; c < N; c++)
    for (size_t i = 0; i < B * T; i++) {
      const double e = (double)outs[0][c][i] - outs[1][c][i];
      num += e * e, den += (double)outs[1][c][i] * outs[1][c][i];
    }
  CHECK(den > 0 && std::sqrt(num / den) <= 1e-6);
  (void)g;
}

int main() {
  try {
    test_context_options();
    test_no_allocation_in_process();
    test_fft_plugin_point();
    test_communicator_single_rank();
    test_pinned_channel_buffers();
    test_gain_calculator_hoa();
    test_gain_calculator_objects();
    test_ptr_adapter();
    test_layout_names();
    test_gain_interpolator();
    test_block_convolver();
    test_delay_and_adapter();
    test_objects_renderer();
  } catch (const std::exception &e) {
    std::printf("FAILED: unexpected exception: %s\n", e.what());
    return 2;
  }
  std::printf("%d checks, %d failed\n", g_checks, g_failed);
  return g_failed ? 1 : 0;
}
