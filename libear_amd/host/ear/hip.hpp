// ear/hip.hpp — glue between the C++ mirror classes and the C ABI (earhip.h):
// status -> exception translation and the process-wide default context.
#pragma once
#include <cstdlib>
#include <string>

#include "earhip.h"
#include "exceptions.hpp"

namespace ear {
  namespace hip {
    /// Translate an earhip status into libear's exception types
    /// (include/earhip.h: 1 -> ear::invalid_argument, 2/3 -> ear::internal_error, 4 -> ear::not_implemented, 5 -> ear::unknown_layout, 6 -> ear::adm_error).
    inline void check(int status) {
      if (status == EARHIP_OK) return;
      const std::string msg = earhip_last_error();
      if (status == EARHIP_INVALID_ARGUMENT) throw invalid_argument(msg);
      if (status == EARHIP_NOT_IMPLEMENTED) throw not_implemented(msg);
      // (the C ABI's messages carry the prefix these types add themselves)
      auto strip = [&](const char *prefix) {
        const std::string p(prefix);
        return msg.compare(0, p.size(), p) == 0 ? msg.substr(p.size()) : msg;
      };
      if (status == EARHIP_UNKNOWN_LAYOUT) throw unknown_layout(strip("unknown layout: "));
      if (status == EARHIP_ADM_ERROR) throw adm_error(strip("ADM error: "));
      throw internal_error(msg);
    }

    /// One device + stream.  Not thread-safe, like libear's stateful DSP objects.
    class Context {
     public:
      explicit Context(int device = 0, void *stream = nullptr) {
        check(earhip_ctx_create(device, stream, &ctx_));
      }
      ~Context() { earhip_ctx_destroy(ctx_); }
      Context(const Context &) = delete;
      Context &operator=(const Context &) = delete;
      earhip_ctx *get() const { return ctx_; }
      /// bit-exact libear arithmetic in the gain kernels (slower)
      void set_strict(bool strict) { check(earhip_ctx_set_strict(ctx_, strict ? 1 : 0)); }
      void synchronize() { check(earhip_ctx_synchronize(ctx_)); }
      /// A tuning knob of this context (include/earhip.h, earhip_ctx_set_option: "MFMA", "H2_TILE", "TAILCUT" ...); the
      /// environment variable EARHIP_<KEY> is only read when a context is created.  reset_option: back to the library's choice.
      void set_option(const char *key, int value) { check(earhip_ctx_set_option(ctx_, key, std::to_string(value).c_str())); }
      void reset_option(const char *key) { check(earhip_ctx_set_option(ctx_, key, nullptr)); }
      /// false: the option is at its default
      bool get_option(const char *key, int &value) const {
        int is_set = 0;
        check(earhip_ctx_get_option(ctx_, key, &is_set, &value));
        return is_set != 0;
      }
      /// Host memory the device reaches directly.  Channel buffers taken from here (or registered once with
      /// register_host: e.g. the storage of an Eigen matrix whose columns PtrAdapter points at,
      /// include/ear/dsp/ptr_adapter.hpp:17-24) let ObjectsRenderer::process skip its staging copies for
      /// block-sized calls when the channel pointers are evenly spaced.  Freed with the context or release_host.
      float *alloc_host(size_t count) {
        void *p = nullptr;
        check(earhip_host_alloc(ctx_, count * sizeof(float), &p));
        return static_cast<float *>(p);
      }
      void register_host(void *ptr, size_t bytes) { check(earhip_host_register(ctx_, ptr, bytes)); }
      void release_host(void *ptr) { check(earhip_host_release(ctx_, ptr)); }

     private:
      earhip_ctx *ctx_ = nullptr;
    };

    /// The exchange step of the multi-GPU render (earhip group J): one Communicator per rank, created
    /// collectively from the 128-byte id rank 0 makes with unique_id() and the caller distributes.
    class Communicator {
     public:
      struct Id {
        char bytes[128];
      };
      static Id unique_id() {
        Id id;
        check(earhip_comm_unique_id(id.bytes));
        return id;
      }
      Communicator(Context &ctx, int rank, int world, const Id &id) : rank_(rank), world_(world) {
        check(earhip_comm_create(ctx.get(), rank, world, id.bytes, &comm_));
      }
      ~Communicator() { earhip_comm_destroy(comm_); }
      Communicator(const Communicator &) = delete;
      Communicator &operator=(const Communicator &) = delete;
      /// rows of the exchange buffers for n_out channels
      int padded_rows(int n_out) const {
        int rows = 0;
        check(earhip_comm_channel_range(n_out, rank_, world_, &rows, nullptr, nullptr));
        return rows;
      }
      /// channels [lo, hi) of the shared bus this rank owns after the exchange
      void channel_range(int n_out, int &lo, int &hi) const {
        check(earhip_comm_channel_range(n_out, rank_, world_, nullptr, &lo, &hi));
      }
      /// sum the ranks' partial outputs (device buffers, [padded_rows][row_stride] -> [padded_rows / world][row_stride])
      void exchange(int slot, const float *partial_dev, float *owned_dev, int n_out, size_t row_stride) {
        check(earhip_render_exchange_device(comm_, slot, partial_dev, owned_dev, (size_t)(padded_rows(n_out) / world_), row_stride));
      }
      /// the shared loudspeaker bus in one place: the owned slices -> full_dev [padded_rows][row_stride] on every
      /// rank (root < 0) or on rank `root` only (full_dev may be null elsewhere); its first n_out rows are the bus
      void gather(int slot, const float *owned_dev, float *full_dev, int n_out, size_t row_stride, int root = -1) {
        check(earhip_comm_gather_device(comm_, slot, owned_dev, full_dev, (size_t)(padded_rows(n_out) / world_), row_stride, root));
      }
      void wait(int slot) { check(earhip_comm_wait(comm_, slot)); }
      /// milliseconds the collectives of `slot` took on the communicator's stream (waits for them)
      double last_exchange_ms(int slot) {
        double ms = 0.0;
        check(earhip_comm_last_exchange_ms(comm_, slot, &ms));
        return ms;
      }
      int rank() const { return rank_; }
      int world() const { return world_; }

     private:
      earhip_comm *comm_ = nullptr;
      int rank_, world_;
    };

    /// Context used by the drop-in classes whose libear signature has no place
    /// for one.  Device from $EARHIP_DEVICE (default 0).  Throws
    /// ear::internal_error when no GPU is present: there is no CPU fallback.
    inline Context &default_context() {
      static Context ctx([] {
        const char *e = std::getenv("EARHIP_DEVICE");
        return e ? std::atoi(e) : 0;
      }());
      return ctx;
    }
  }  // namespace hip
}  // namespace ear
