// common.h — shared host-side plumbing of libearhip: status codes, the
// thread-local error string, the context object and HIP error checks.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/earhip.h"

namespace earhip {

struct Error {
  int code;
  std::string msg;
};

void set_last_error(const std::string &msg);

// Runs f, translating earhip::Error / std::exception into a status code.
template <typename F>
int guarded(F &&f) {
  try {
    f();
    return EARHIP_OK;
  } catch (const Error &e) {
    set_last_error(e.msg);
    return e.code;
  } catch (const std::bad_alloc &) {
    set_last_error("out of host memory");
    return EARHIP_INTERNAL_ERROR;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return EARHIP_INTERNAL_ERROR;
  }
}

[[noreturn]] inline void fail_invalid(const std::string &m) {
  throw Error{EARHIP_INVALID_ARGUMENT, m};
}
[[noreturn]] inline void fail_internal(const std::string &m) {
  throw Error{EARHIP_INTERNAL_ERROR, "internal error: " + m};
}
inline void require(bool ok, const char *m) {
  if (!ok) fail_invalid(m);
}

#define EARHIP_HIP(expr)                                                        \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess)                                                       \
      throw ::earhip::Error{EARHIP_DEVICE_ERROR,                                \
                            std::string("internal error: HIP: ") +              \
                                hipGetErrorString(e_) + " in " #expr};          \
  } while (0)

inline bool is_pow2(size_t v) { return v && !(v & (v - 1)); }

// Owning device buffer (setup-time allocation only).
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    if (count == 0) count = 1;
    EARHIP_HIP(hipMalloc((void **)&p, count * sizeof(T)));
    n = count;
  }
  void alloc_zero(size_t count, hipStream_t s) {
    alloc(count);
    EARHIP_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
  // grows (never shrinks); contents are lost on growth
  void reserve(size_t count) {
    if (count > n) alloc(count);
  }
};

// Pinned host staging buffer.
template <typename T>
struct PinBuf {
  T *p = nullptr;
  size_t n = 0;
  PinBuf() = default;
  PinBuf(const PinBuf &) = delete;
  PinBuf &operator=(const PinBuf &) = delete;
  ~PinBuf() {
    if (p) (void)hipHostFree(p);
  }
  void reserve(size_t count) {
    if (count <= n) return;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    n = 0;
    EARHIP_HIP(hipHostMalloc((void **)&p, count * sizeof(T), hipHostMallocDefault));
    n = count;
  }
};

}  // namespace earhip

struct earhip_vbs;
namespace earhip {
// api_vbs.cpp: the context of an adapter with pinned buffers is going away (its buffers with it)
void vbs_orphan(earhip_vbs *v);
}  // namespace earhip

namespace earhip {
// Run-time configuration of a context ("a small runtime config"): every tuning knob is an OPTION of the context,
// set with earhip_ctx_set_option(ctx, key, value) or taken ONCE, at earhip_ctx_create, from the environment variable
// EARHIP_<KEY>.  Nothing on a process call's path reads the environment.  (include/earhip.h, group A, lists the keys.)
enum Opt {
  OPT_SPL, OPT_MFMA, OPT_XSCALE, OPT_WAVES, OPT_TPW, OPT_NRT,            // gain kernels' shapes (context-wide)
  OPT_H2_TILE, OPT_H2_WGS, OPT_H2_RUNS, OPT_P2_TILE, OPT_P2_PAIRS, OPT_P2_WGS, OPT_HINGE, OPT_HG_TILE, OPT_HBUILD_TPW, OPT_BUILD_TPW,  // launch plan of a call
  OPT_K2_WG, OPT_K2_OWN_BLOCK, OPT_RUN, OPT_GSPLIT,                      // decorrelator kernel / renderer creation
  OPT_PROBE_RUNS, OPT_BLOCK_GROUPS, OPT_DEBUG_TIMING, OPT_TAILCUT, OPT_HOST_CHUNK_MB, OPT_HOST_THREADS, OPT_HG_ROBUST, OPT_BUILD_2K,
  OPT_HOST_BIND, OPT_HOST_NT, OPT_HOST_FIRST, OPT_H2_PAIR, OPT_COUNT
};
struct OptVal {
  bool set = false;
  int v = 0;
};
}  // namespace earhip

struct earhip_ctx {
  earhip::OptVal opt[earhip::OPT_COUNT];
  bool has(earhip::Opt o) const { return opt[o].set; }
  int get(earhip::Opt o, int dflt = 0) const { return opt[o].set ? opt[o].v : dflt; }
  void apply_options();  // the derived fields below from opt[] (api_core.hip)
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool strict = false;
  int spl = 4;  // samples per lane of the VALU gain_mix kernel (2 or 4)
  int use_mfma = 3;  // non-strict gain stage: 0 VALU, 1 f32 MFMA, 4 f16x2 MFMA, 3 (default) f16x2
                     // when all curve points lie on tile boundaries, else f32 MFMA
  int x_scale_log2 = 14;  // f16x2 kernel: inputs are scaled by 2^x_scale_log2 before the split (gain_h2.h) ...
  bool x_scale_auto = true;  // ... unless a level estimate of the call's inputs is available (default: K0 probes them)
  earhip::DevBuf<unsigned> level;  // [2] input level words (float bits), used alternately by successive calls
  int level_idx = 0;
  int last_gate_idx = -1;  // the mode word ([2 + idx]) of the last call planned for the hinge kernel; -1: the last call was not
  bool last_hinge_robust = false;  // ... and a call beyond the packed kink products' span runs the kernel's robust form (else: the piece lists standing by)
  int last_wide_idx = -1;  // the mode word of the last call whose split-operand kernel picked its form on the device; -1: none (wide form)
  // [2][tile_slow_cap] words used alternately by successive calls of the f16x2 gain kernel: non-zero = some object
  // of the tile needs the kernel's exact path (set by K0: k_seg_prep, cleared for the call after next by K1)
  earhip::DevBuf<unsigned> tile_slow;
  size_t tile_slow_cap = 0;
  int tile_slow_idx = 0;
  earhip::DevBuf<unsigned> obj_level;  // [2][obj_level_cap] per-object input levels of the current call (float bits): the largest and the
                                       // smallest non-zero magnitude the level probe saw (k_level_probe: plain stores, no state between calls)
  int obj_level_cap = 0;
  long lazy_allocs = 0;  // process calls that had to make or grow one of the buffers above themselves (reserve_call_words did not cover them)
  unsigned *record_slot = nullptr;  // where the NEXT launch_gain_mix leaves a copy of its call's mode word (a renderer's own slot; consumed by the launch)
  int max_waves = 4;  // waves per gain_mix workgroup (column groups x object splits)
  int tiles_per_wg = 4;  // MFMA kernel: adjacent tiles per workgroup (share gain rows through L1)
  bool tiles_per_wg_forced = false;  // (EARHIP_TPW: taken as given)
  int nrt = 8;  // 16-sample row tiles per wave of the MFMA kernel (4 or 8)
  int num_cus = 256;
  size_t hinge_build_lds = 64 * 1024;  // dynamic LDS a launch of k_hinge_build may ask for on this device (earhip_ctx_create)
  // host memory the device reaches directly: earhip_host_alloc / earhip_host_register ranges (the host-pointer
  // entry points skip their staging copies for channel pointers that are evenly spaced inside one of them)
  struct HostRange {
    const char *base;
    size_t bytes;
    bool owned;  // earhip_host_alloc (freed with the context) or registered caller memory
  };
  std::vector<HostRange> host_ranges;
  // adapters whose FIFO lives in this context's pinned memory (earhip_vbs_create_pinned): told when the
  // context is destroyed first, so that their destroy does not touch a dead context
  std::vector<earhip_vbs *> pinned_adapters;
  bool host_reachable(const void *p, size_t bytes) const {
    const char *c = static_cast<const char *>(p);
    for (const auto &r : host_ranges)
      if (c >= r.base && c + bytes <= r.base + r.bytes) return true;
    return false;
  }
  // staging for the host-pointer entry points (grown at first use / create)
  earhip::PinBuf<float> pin_in, pin_out;
  earhip::DevBuf<float> dev_in, dev_out, dev_pts;
  void use() const { EARHIP_HIP(hipSetDevice(device)); }
};
