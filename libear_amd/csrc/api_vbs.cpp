// api_vbs.cpp — (E) VariableBlockSizeAdapter of include/earhip.h: host-side FIFO
// re-blocking around a fixed-block-size callback, behaviour of libear's
// src/dsp/variable_block_size_impl.cpp:27-85 (first block_size output samples
// are zeros; the callback runs in place on the internal buffers each time a
// block has been collected).  No device work here: the callback is where the
// device chain (e.g. earhip_render_process) is invoked.  earhip_vbs_create_pinned keeps the two FIFO buffers
// in device-reachable host memory of a context, so a renderer called from the callback (the use libear's docs
// recommend, docs/dsp.rst:65-71) takes its no-staging path: the FIFO rows are evenly spaced channel buffers.
#include <algorithm>
#include <memory>
#include <vector>

#include "common.h"

using namespace earhip;

struct earhip_vbs {
  size_t B, nin, nout, fill = 0;
  earhip_process_func fn;
  void *user;
  std::vector<float> own_i, own_o;  // channel-major [nch][B] (ordinary memory) ...
  float *ibuf = nullptr, *obuf = nullptr;  // ... or pinned memory of `ctx`
  earhip_ctx *ctx = nullptr;
  std::vector<const float *> iptr;
  std::vector<float *> optr;
  bool orphaned = false;  // the context that owned the pinned buffers was destroyed first
};

namespace earhip {
void vbs_orphan(earhip_vbs *v) {
  v->ctx = nullptr;
  v->ibuf = v->obuf = nullptr;  // (freed with the context)
  v->orphaned = true;
}
}  // namespace earhip

extern "C" {

static void vbs_create(earhip_ctx *ctx, size_t block_size, size_t num_channels_in, size_t num_channels_out,
                       earhip_process_func process_func, void *user, earhip_vbs **out) {
  require(out != nullptr && process_func != nullptr, "NULL argument");
  require(block_size >= 1, "block_size must be >= 1");
  std::unique_ptr<earhip_vbs> v(new earhip_vbs);
  v->B = block_size;
  v->nin = num_channels_in;
  v->nout = num_channels_out;
  v->fn = process_func;
  v->user = user;
  const size_t ni = std::max<size_t>(block_size * num_channels_in, 4), no = std::max<size_t>(block_size * num_channels_out, 4);
  if (ctx) {
    void *p = nullptr;
    if (earhip_host_alloc(ctx, sizeof(float) * ni, &p) != EARHIP_OK) fail_internal(earhip_last_error());
    v->ibuf = static_cast<float *>(p);
    if (earhip_host_alloc(ctx, sizeof(float) * no, &p) != EARHIP_OK) {
      (void)earhip_host_release(ctx, v->ibuf);
      fail_internal(earhip_last_error());
    }
    v->obuf = static_cast<float *>(p);
    v->ctx = ctx;
    ctx->pinned_adapters.push_back(v.get());
    std::fill(v->ibuf, v->ibuf + ni, 0.0f);
    std::fill(v->obuf, v->obuf + no, 0.0f);
  } else {
    v->own_i.assign(ni, 0.0f);
    v->own_o.assign(no, 0.0f);
    v->ibuf = v->own_i.data();
    v->obuf = v->own_o.data();
  }
  v->iptr.resize(num_channels_in);
  v->optr.resize(num_channels_out);
  for (size_t c = 0; c < num_channels_in; c++) v->iptr[c] = v->ibuf + c * block_size;
  for (size_t c = 0; c < num_channels_out; c++) v->optr[c] = v->obuf + c * block_size;
  *out = v.release();
}

int earhip_vbs_create(size_t block_size, size_t num_channels_in, size_t num_channels_out,
                      earhip_process_func process_func, void *user, earhip_vbs **out) {
  return guarded([&] { vbs_create(nullptr, block_size, num_channels_in, num_channels_out, process_func, user, out); });
}

int earhip_vbs_create_pinned(earhip_ctx *ctx, size_t block_size, size_t num_channels_in, size_t num_channels_out,
                             earhip_process_func process_func, void *user, earhip_vbs **out) {
  return guarded([&] {
    require(ctx != nullptr, "ctx must not be NULL");
    vbs_create(ctx, block_size, num_channels_in, num_channels_out, process_func, user, out);
  });
}

// (either order: a context destroyed first takes the pinned buffers with it and tells its adapters)
int earhip_vbs_destroy(earhip_vbs *v) {
  if (v && v->ctx) {
    auto &list = v->ctx->pinned_adapters;
    list.erase(std::remove(list.begin(), list.end(), v), list.end());
    (void)earhip_host_release(v->ctx, v->ibuf);
    (void)earhip_host_release(v->ctx, v->obuf);
  }
  delete v;
  return EARHIP_OK;
}

int earhip_vbs_get_delay(const earhip_vbs *v) { return v ? (int)v->B : 0; }

int earhip_vbs_process(earhip_vbs *v, size_t nsamples, const float *const *in, float *const *out) {
  int cb_status = EARHIP_OK;
  const int rc = guarded([&] {
    require(v != nullptr, "adapter must not be NULL");
    require(!v->orphaned, "the context that holds this adapter's buffers has been destroyed");
    require(nsamples == 0 || (in != nullptr && out != nullptr), "in and out must not be NULL");
    size_t done = 0;
    while (done < nsamples) {
      const size_t n = std::min(nsamples - done, v->B - v->fill);
      for (size_t c = 0; c < v->nin; c++)
        std::copy(in[c] + done, in[c] + done + n, v->ibuf + c * v->B + v->fill);
      for (size_t c = 0; c < v->nout; c++)
        std::copy(v->obuf + c * v->B + v->fill, v->obuf + c * v->B + v->fill + n,
                  out[c] + done);
      done += n;
      v->fill += n;
      const bool run = v->fill == v->B;
      if (run) {
        cb_status = v->fn(v->iptr.data(), v->optr.data(), v->user);
        v->fill = 0;
        if (cb_status != EARHIP_OK) return;  // message already set by the callee
      }
      if (!run && n == 0) fail_internal("no progress made");
    }
    if (done != nsamples) fail_internal("processed more samples than expected");
  });
  return rc != EARHIP_OK ? rc : cb_status;
}

}  // extern "C"
