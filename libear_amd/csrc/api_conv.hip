// api_conv.hip — (B) FFT plugin, (C) BlockConvolver family and (D) DelayBuffer
// of include/earhip.h.  The host side keeps exactly the bookkeeping of libear's
// BlockConvolver (filter queue, spectra queues, "is all zero" flags —
// src/dsp/block_convolver_impl.{hpp,cpp}); the arithmetic runs in the kernels of
// fft_kernels.h on device-resident queues.
#include <cstring>
#include <memory>

#include "common.h"
#include "fft_kernels.h"

namespace earhip {
std::vector<cf> make_twiddles(int L);
void launch_spectrum(int L, const float *in, size_t stride, int n_valid, const cf *tw, cf *out,
                     int rows, hipStream_t s);

template <int L>
static void ifft_real_t(const cf *in, const cf *tw, float *out, int rows, hipStream_t s) {
  hipLaunchKernelGGL((k_ifft_real<L>), dim3(rows), dim3(kFftThreads), 0, s, in, tw, out);
}
template <int L>
static void conv_forward_t(const float *x, int fade, const cf *tw, cf *Xo, cf *Xn, hipStream_t s) {
  hipLaunchKernelGGL((k_conv_forward<L>), dim3(1), dim3(kFftThreads), 0, s, x, fade, tw, Xo, Xn);
}
template <int L>
static void conv_ifft_ola_t(const cf *Y, const cf *tw, float *tail, int use_tail, int mode,
                            float *out, hipStream_t s) {
  hipLaunchKernelGGL((k_conv_ifft_ola<L>), dim3(1), dim3(kFftThreads), 0, s, Y, tw, tail, use_tail,
                     mode, out);
}

// CALL(L) for the power-of-two sizes with their own instantiation, CALL_RT for every other size
#define EARHIP_DISPATCH_L(L, CALL, CALL_RT)                                     \
  switch (L) {                                                                  \
    case 64: CALL(64); break;                                                   \
    case 128: CALL(128); break;                                                 \
    case 256: CALL(256); break;                                                 \
    case 512: CALL(512); break;                                                 \
    case 1024: CALL(1024); break;                                               \
    case 2048: CALL(2048); break;                                               \
    case 4096: CALL(4096); break;                                               \
    case 8192: CALL(8192); break;                                               \
    default: CALL_RT; break;                                                    \
  }

static FftShape shape_of(int L) {
  FftShape S;
  if (!fft_make_shape(L, &S)) fail_invalid("FFT size must be in [4, 8192]");
  return S;
}
static void ifft_real_rt(int L, const cf *in, const cf *tw, float *out, int rows, hipStream_t s) {
  hipLaunchKernelGGL(k_ifft_real_rt, dim3(rows), dim3(kFftThreads), fft_rt_lds(k_ifft_real_rt, L), s, shape_of(L),
                     in, tw, out);
}
static void conv_forward_rt(int L, const float *x, int fade, const cf *tw, cf *Xo, cf *Xn, hipStream_t s) {
  hipLaunchKernelGGL(k_conv_forward_rt, dim3(1), dim3(kFftThreads), fft_rt_lds(k_conv_forward_rt, L), s,
                     shape_of(L), x, fade, tw, Xo, Xn);
}
static void conv_ifft_ola_rt(int L, const cf *Y, const cf *tw, float *tail, int use_tail, int mode, float *out,
                             hipStream_t s) {
  hipLaunchKernelGGL(k_conv_ifft_ola_rt, dim3(1), dim3(kFftThreads), fft_rt_lds(k_conv_ifft_ola_rt, L), s,
                     shape_of(L), Y, tw, tail, use_tail, mode, out);
}
}  // namespace earhip

using namespace earhip;

// ---------------------------------------------------------------------------
// (B) FFT plan
// ---------------------------------------------------------------------------
struct earhip_fft_plan {
  earhip_ctx *ctx;
  int L;
  DevBuf<cf> tw, spec;
  DevBuf<float> td;
};

// ---------------------------------------------------------------------------
// (C) BlockConvolver
// ---------------------------------------------------------------------------
struct earhip_conv_ctx {
  earhip_ctx *ctx;
  int B, L;        // block size, FFT size (block_convolver_impl.cpp:10-14)
  DevBuf<cf> tw;
  int live = 0;    // filters / convolvers that still point here
};

struct earhip_conv_filter {
  earhip_conv_ctx *cctx;
  int nblocks;
  DevBuf<cf> spec;  // [nblocks][B+1]
  // One reference for the creator's handle (dropped by earhip_conv_filter_destroy) and one per
  // convolver queue slot that points here: libear's queue holds shared_ptrs
  // (src/dsp/block_convolver_impl.hpp:154-167), so a caller may let go of a Filter that is still
  // fading out.  Not atomic: a filter and the convolvers using it belong to one thread, like libear's.
  int refs = 1;
  const cf *block(int i) const { return spec.p + (size_t)i * (cctx->B + 1); }
};

static void filter_release(const earhip_conv_filter *f) {
  if (!f) return;
  earhip_conv_filter *m = const_cast<earhip_conv_filter *>(f);
  if (--m->refs > 0) return;
  (void)hipStreamSynchronize(m->cctx->ctx->stream);  // a queued kernel may still read the spectra
  m->cctx->live--;
  delete m;
}
static const earhip_conv_filter *filter_retain(const earhip_conv_filter *f) {
  if (f) const_cast<earhip_conv_filter *>(f)->refs++;
  return f;
}

struct earhip_conv {
  earhip_conv_ctx *cctx;
  int P;  // num_blocks
  // filter queue, length P+1 (block_convolver_impl.hpp:154-167)
  std::vector<const earhip_conv_filter *> fq;
  int f_ofs = 0, s_ofs = 0;
  // spectra queues + zero flags (:169-182), device resident
  DevBuf<cf> sp_old, sp_new;  // [P][B+1]
  std::vector<char> old_zero, new_zero;
  DevBuf<float> tail;         // last_tail (:188-190)
  bool tail_zero = true;
  DevBuf<cf> Y;               // multiply_out
  DevBuf<float> d_in, d_out;
  PinBuf<float> p_io;

  const earhip_conv_filter *filt(int i) const { return fq[(f_ofs + i) % (P + 1)]; }
  // queue slot i <- f; the slot owns a reference (retain before release: f may be what the slot holds)
  void set_filt(int i, const earhip_conv_filter *f) {
    const earhip_conv_filter *&q = fq[(f_ofs + i) % (P + 1)];
    filter_retain(f);
    filter_release(q);
    q = f;
  }
  int sidx(int i) const { return (s_ofs + i) % P; }
  cf *old_at(int i) { return sp_old.p + (size_t)sidx(i) * (cctx->B + 1); }
  cf *new_at(int i) { return sp_new.p + (size_t)sidx(i) * (cctx->B + 1); }

  // block_convolver_impl.cpp:85-98
  void check_filter(const earhip_conv_filter *f) const {
    if (!f) return;
    if (f->cctx->B != cctx->B)
      fail_invalid(
          "Filter block size is not equal to BlockConvolver block size; was this created using "
          "the same context?");
    if (f->nblocks > P) fail_invalid("too many blocks in given Filter");
  }
};

// ---------------------------------------------------------------------------
// (D) DelayBuffer
// ---------------------------------------------------------------------------
struct earhip_delay {
  earhip_ctx *ctx;
  size_t nch, delay;
  DevBuf<float> mem[2];
  int cur = 0;
  DevBuf<float> d_in, d_out;
  PinBuf<float> p_in, p_out;
};

extern "C" {

int earhip_fft_plan_create(earhip_ctx *ctx, size_t n_fft, earhip_fft_plan **out) {
  return guarded([&] {
    require(ctx != nullptr && out != nullptr, "NULL argument");
    // (kissfft's real transform needs an even length, submodules/kissfft/kiss_fftr.c; any factorisation)
    FftShape shape;
    require(n_fft % 2 == 0 && n_fft >= 2 && n_fft <= 8192 && fft_make_shape((int)n_fft, &shape),
            "n_fft must be even and in [2, 8192]");
    ctx->use();
    std::unique_ptr<earhip_fft_plan> p(new earhip_fft_plan);
    p->ctx = ctx;
    p->L = (int)n_fft;
    const auto tw = make_twiddles(p->L);
    p->tw.alloc(p->L);
    EARHIP_HIP(hipMemcpy(p->tw.p, tw.data(), sizeof(cf) * p->L, hipMemcpyHostToDevice));
    p->spec.alloc(p->L);
    p->td.alloc(p->L);
    *out = p.release();
  });
}

int earhip_fft_plan_destroy(earhip_fft_plan *plan) {
  return guarded([&] {
    if (!plan) return;
    (void)hipStreamSynchronize(plan->ctx->stream);
    delete plan;
  });
}

int earhip_fft_forward(earhip_fft_plan *p, const float *in, float *out_complex) {
  return guarded([&] {
    require(p != nullptr && in != nullptr && out_complex != nullptr, "NULL argument");
    earhip_ctx *ctx = p->ctx;
    ctx->use();
    EARHIP_HIP(hipMemcpyAsync(p->td.p, in, sizeof(float) * p->L, hipMemcpyHostToDevice, ctx->stream));
    launch_spectrum(p->L, p->td.p, p->L, p->L, p->tw.p, p->spec.p, 1, ctx->stream);
    EARHIP_HIP(hipMemcpyAsync(out_complex, p->spec.p, sizeof(cf) * (p->L / 2 + 1),
                              hipMemcpyDeviceToHost, ctx->stream));
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
  });
}

int earhip_fft_reverse(earhip_fft_plan *p, const float *in_complex, float *out) {
  return guarded([&] {
    require(p != nullptr && in_complex != nullptr && out != nullptr, "NULL argument");
    earhip_ctx *ctx = p->ctx;
    ctx->use();
    EARHIP_HIP(hipMemcpyAsync(p->spec.p, in_complex, sizeof(cf) * (p->L / 2 + 1),
                              hipMemcpyHostToDevice, ctx->stream));
#define CALL(LL) ifft_real_t<LL>(p->spec.p, p->tw.p, p->td.p, 1, ctx->stream)
    EARHIP_DISPATCH_L(p->L, CALL, ifft_real_rt(p->L, p->spec.p, p->tw.p, p->td.p, 1, ctx->stream))
#undef CALL
    EARHIP_HIP(hipGetLastError());
    EARHIP_HIP(hipMemcpyAsync(out, p->td.p, sizeof(float) * p->L, hipMemcpyDeviceToHost, ctx->stream));
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
  });
}

// ---- Context / Filter --------------------------------------------------------

int earhip_conv_ctx_create(earhip_ctx *ctx, size_t block_size, earhip_conv_ctx **out) {
  return guarded([&] {
    require(ctx != nullptr && out != nullptr, "NULL argument");
    FftShape shape;
    require(block_size >= 1 && block_size <= 4096 && fft_make_shape(2 * (int)block_size, &shape),
            "block_size must be in [1, 4096]");
    ctx->use();
    std::unique_ptr<earhip_conv_ctx> c(new earhip_conv_ctx);
    c->ctx = ctx;
    c->B = (int)block_size;
    c->L = 2 * c->B;
    const auto tw = make_twiddles(c->L);
    c->tw.alloc(c->L);
    EARHIP_HIP(hipMemcpy(c->tw.p, tw.data(), sizeof(cf) * c->L, hipMemcpyHostToDevice));
    *out = c.release();
  });
}

int earhip_conv_ctx_destroy(earhip_conv_ctx *cctx) {
  return guarded([&] {
    if (!cctx) return;
    if (cctx->live != 0)
      fail_invalid("convolver context still has live filters or convolvers");
    (void)hipStreamSynchronize(cctx->ctx->stream);
    delete cctx;
  });
}

// block_convolver_impl.cpp:16-41
int earhip_conv_filter_create(earhip_conv_ctx *cctx, size_t n, const float *taps,
                              earhip_conv_filter **out) {
  return guarded([&] {
    require(cctx != nullptr && out != nullptr, "NULL argument");
    require(n == 0 || taps != nullptr, "taps must not be NULL");
    earhip_ctx *ctx = cctx->ctx;
    ctx->use();
    const int B = cctx->B, L = cctx->L;
    std::unique_ptr<earhip_conv_filter> f(new earhip_conv_filter);
    f->cctx = cctx;
    f->nblocks = (int)((n + B - 1) / B);
    if (f->nblocks > 0) {
      // rows of B taps (last one zero padded), transformed in one batch
      std::vector<float> rows((size_t)f->nblocks * B, 0.0f);
      std::memcpy(rows.data(), taps, sizeof(float) * n);
      DevBuf<float> d_rows;
      DevBuf<cf> full;
      d_rows.alloc(rows.size());
      full.alloc((size_t)f->nblocks * L);
      EARHIP_HIP(hipMemcpy(d_rows.p, rows.data(), sizeof(float) * rows.size(), hipMemcpyHostToDevice));
      launch_spectrum(L, d_rows.p, B, B, cctx->tw.p, full.p, f->nblocks, ctx->stream);
      f->spec.alloc((size_t)f->nblocks * (B + 1));
      EARHIP_HIP(hipMemcpy2DAsync(f->spec.p, sizeof(cf) * (B + 1), full.p, sizeof(cf) * L,
                                  sizeof(cf) * (B + 1), f->nblocks, hipMemcpyDeviceToDevice,
                                  ctx->stream));
      EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    }
    cctx->live++;
    *out = f.release();
  });
}

int earhip_conv_filter_destroy(earhip_conv_filter *filter) {
  return guarded([&] {
    filter_release(filter);  // the creator's reference; convolver queues may still hold theirs
  });
}

size_t earhip_conv_filter_num_blocks(const earhip_conv_filter *filter) {
  return filter ? (size_t)filter->nblocks : 0;
}

// ---- BlockConvolver ------------------------------------------------------------

int earhip_conv_create(earhip_conv_ctx *cctx, const earhip_conv_filter *filter, size_t num_blocks,
                       earhip_conv **out) {
  return guarded([&] {
    require(cctx != nullptr && out != nullptr, "NULL argument");
    if (num_blocks == 0) {
      require(filter != nullptr, "num_blocks must be given when there is no filter");
      num_blocks = (size_t)filter->nblocks;  // block_convolver_impl.cpp:63-69
    }
    require(num_blocks >= 1 && num_blocks <= 4096, "num_blocks must be in [1, 4096]");
    earhip_ctx *ctx = cctx->ctx;
    ctx->use();
    const int B = cctx->B;
    std::unique_ptr<earhip_conv> c(new earhip_conv);
    c->cctx = cctx;
    c->P = (int)num_blocks;
    c->fq.assign(c->P + 1, nullptr);
    c->sp_old.alloc_zero((size_t)c->P * (B + 1), ctx->stream);
    c->sp_new.alloc_zero((size_t)c->P * (B + 1), ctx->stream);
    c->old_zero.assign(c->P, 1);
    c->new_zero.assign(c->P, 1);
    c->tail.alloc_zero(B, ctx->stream);
    c->Y.alloc_zero(B + 1, ctx->stream);
    c->d_in.alloc(B);
    c->d_out.alloc(B);
    c->p_io.reserve(2 * (size_t)B);
    if (filter) {
      c->check_filter(filter);
      for (int i = 0; i <= c->P; i++) c->set_filt(i, filter);  // set_filter
    }
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    cctx->live++;
    *out = c.release();
  });
}

int earhip_conv_destroy(earhip_conv *conv) {
  return guarded([&] {
    if (!conv) return;
    (void)hipStreamSynchronize(conv->cctx->ctx->stream);
    earhip_conv_ctx *cctx = conv->cctx;
    for (int i = 0; i <= conv->P; i++) conv->set_filt(i, nullptr);
    cctx->live--;
    delete conv;
  });
}

// block_convolver_impl.cpp:71-76
int earhip_conv_crossfade_filter(earhip_conv *conv, const earhip_conv_filter *filter) {
  return guarded([&] {
    require(conv != nullptr, "conv must not be NULL");
    conv->check_filter(filter);
    conv->set_filt(0, filter);
  });
}

// block_convolver_impl.cpp:78-83
int earhip_conv_set_filter(earhip_conv *conv, const earhip_conv_filter *filter) {
  return guarded([&] {
    require(conv != nullptr, "conv must not be NULL");
    conv->check_filter(filter);
    for (int i = 0; i <= conv->P; i++) conv->set_filt(i, filter);
  });
}

// block_convolver_impl.cpp:143-237
int earhip_conv_process(earhip_conv *c, const float *in, float *out) {
  return guarded([&] {
    require(c != nullptr, "conv must not be NULL");
    require(out != nullptr, "out must be of size block_size");
    earhip_conv_ctx *cc = c->cctx;
    earhip_ctx *ctx = cc->ctx;
    ctx->use();
    const int B = cc->B, L = cc->L;
    hipStream_t s = ctx->stream;

    bool silent = in == nullptr;
    if (!silent) {
      silent = true;
      for (int i = 0; i < B; i++)
        if (in[i] != 0.0f) {
          silent = false;
          break;
        }
    }
    const int i0 = c->sidx(0);
    bool staged = false;  // an async copy out of p_io is in flight
    if (silent) {  // :156-159
      c->old_zero[i0] = 1;
      c->new_zero[i0] = 1;
    } else {
      std::memcpy(c->p_io.p, in, sizeof(float) * B);
      EARHIP_HIP(hipMemcpyAsync(c->d_in.p, c->p_io.p, sizeof(float) * B, hipMemcpyHostToDevice, s));
      staged = true;
      const int fade = c->filt(1) != c->filt(0) ? 1 : 0;  // :162
#define CALL(LL) conv_forward_t<LL>(c->d_in.p, fade, cc->tw.p, c->old_at(0), c->new_at(0), s)
      EARHIP_DISPATCH_L(L, CALL, conv_forward_rt(L, c->d_in.p, fade, cc->tw.p, c->old_at(0), c->new_at(0), s))
#undef CALL
      EARHIP_HIP(hipGetLastError());
      c->new_zero[i0] = 0;
      c->old_zero[i0] = fade ? 0 : 1;  // :186
    }

    // spectral multiply-accumulate in queue order, old before new (:193-209)
    MacTerms T;
    T.n = 0;
    T.overwrite = 1;
    bool any = false;
    auto flush = [&] {
      if (T.n == 0) return;
      hipLaunchKernelGGL(k_conv_mac, dim3((B + 1 + 255) / 256), dim3(256), 0, s, T, B + 1, c->Y.p);
      EARHIP_HIP(hipGetLastError());
      T.n = 0;
      T.overwrite = 0;
    };
    auto add = [&](const cf *H, const cf *X) {
      T.H[T.n] = H;
      T.X[T.n] = X;
      T.n++;
      any = true;
      if (T.n == 16) flush();
    };
    for (int i = 0; i < c->P; i++) {
      const earhip_conv_filter *fo = c->filt(i + 1), *fn = c->filt(i);
      if (fo && i < fo->nblocks && !c->old_zero[c->sidx(i)]) add(fo->block(i), c->old_at(i));
      if (fn && i < fn->nblocks && !c->new_zero[c->sidx(i)]) add(fn->block(i), c->new_at(i));
    }
    flush();

    bool have_out = true;
    if (any) {  // :217-226
#define CALL(LL) conv_ifft_ola_t<LL>(c->Y.p, cc->tw.p, c->tail.p, c->tail_zero ? 0 : 1, 0, c->d_out.p, s)
      EARHIP_DISPATCH_L(L, CALL, conv_ifft_ola_rt(L, c->Y.p, cc->tw.p, c->tail.p, c->tail_zero ? 0 : 1, 0, c->d_out.p, s))
#undef CALL
      EARHIP_HIP(hipGetLastError());
      c->tail_zero = false;
    } else if (!c->tail_zero) {  // :227-230
#define CALL(LL) conv_ifft_ola_t<LL>(c->Y.p, cc->tw.p, c->tail.p, 1, 1, c->d_out.p, s)
      EARHIP_DISPATCH_L(L, CALL, conv_ifft_ola_rt(L, c->Y.p, cc->tw.p, c->tail.p, 1, 1, c->d_out.p, s))
#undef CALL
      EARHIP_HIP(hipGetLastError());
      c->tail_zero = true;  // tail contents are ignored while the flag is set
    } else {  // :231-234
      have_out = false;
      std::memset(out, 0, sizeof(float) * B);
      // (input stored for a later filter, nothing produced: the staging buffer is the next call's too)
      if (staged) EARHIP_HIP(hipStreamSynchronize(s));
    }
    if (have_out) {
      EARHIP_HIP(hipMemcpyAsync(c->p_io.p + B, c->d_out.p, sizeof(float) * B, hipMemcpyDeviceToHost, s));
      EARHIP_HIP(hipStreamSynchronize(s));
      std::memcpy(out, c->p_io.p + B, sizeof(float) * B);
    }

    // rotate_queues (:114-122)
    c->s_ofs = (c->s_ofs + c->P - 1) % c->P;
    c->f_ofs = (c->f_ofs + c->P) % (c->P + 1);
    c->set_filt(0, c->filt(1));  // (the slot rotated in held the oldest filter: released here)
  });
}

// ---- DelayBuffer -----------------------------------------------------------------

int earhip_delay_create(earhip_ctx *ctx, size_t nchannels, size_t nsamples, earhip_delay **out) {
  return guarded([&] {
    require(ctx != nullptr && out != nullptr, "NULL argument");
    require(nchannels >= 1 && nchannels <= 65535, "nchannels must be in [1, 65535]");
    require(nsamples < ((size_t)1 << 28), "delay too long");
    ctx->use();
    std::unique_ptr<earhip_delay> d(new earhip_delay);
    d->ctx = ctx;
    d->nch = nchannels;
    d->delay = nsamples;
    for (int i = 0; i < 2; i++) d->mem[i].alloc_zero(nchannels * std::max<size_t>(nsamples, 1), ctx->stream);
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    *out = d.release();
  });
}

int earhip_delay_destroy(earhip_delay *d) {
  return guarded([&] {
    if (!d) return;
    (void)hipStreamSynchronize(d->ctx->stream);
    delete d;
  });
}

int earhip_delay_get_delay(const earhip_delay *d) { return d ? (int)d->delay : 0; }

// delay_buffer_impl.cpp:19-40
int earhip_delay_process(earhip_delay *d, size_t nsamples, const float *const *in,
                         float *const *out) {
  return guarded([&] {
    require(d != nullptr, "delay must not be NULL");
    require(in != nullptr && out != nullptr, "in and out must not be NULL");
    require(nsamples < ((size_t)1 << 28), "nsamples too large");
    if (nsamples == 0) return;
    earhip_ctx *ctx = d->ctx;
    ctx->use();
    const size_t n = nsamples, tot = n * d->nch;
    d->p_in.reserve(tot);
    d->p_out.reserve(tot);
    d->d_in.reserve(tot);
    d->d_out.reserve(tot);
    for (size_t c = 0; c < d->nch; c++) std::memcpy(d->p_in.p + c * n, in[c], sizeof(float) * n);
    EARHIP_HIP(hipMemcpyAsync(d->d_in.p, d->p_in.p, sizeof(float) * tot, hipMemcpyHostToDevice, ctx->stream));
    const dim3 grid((unsigned)((n + d->delay + 255) / 256), (unsigned)d->nch);
    hipLaunchKernelGGL(k_delay, grid, dim3(256), 0, ctx->stream, d->d_in.p, n, (int)n, (int)d->delay,
                       d->mem[d->cur].p, d->mem[d->cur ^ 1].p, d->d_out.p);
    EARHIP_HIP(hipGetLastError());
    d->cur ^= 1;
    EARHIP_HIP(hipMemcpyAsync(d->p_out.p, d->d_out.p, sizeof(float) * tot, hipMemcpyDeviceToHost, ctx->stream));
    EARHIP_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t c = 0; c < d->nch; c++) std::memcpy(out[c], d->p_out.p + c * n, sizeof(float) * n);
  });
}

}  // extern "C"
