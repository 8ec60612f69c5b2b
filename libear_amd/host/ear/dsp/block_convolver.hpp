// ear/dsp/block_convolver.hpp — Context / Filter / BlockConvolver with libear's
// interface (include/ear/dsp/block_convolver.hpp:28-112) over the device
// implementation (earhip_conv_*).
#pragma once
#include <complex>
#include <cstddef>
#include <memory>

#include "../fft.hpp"
#include "../hip.hpp"

namespace ear {
  namespace dsp {
    namespace block_convolver {
      using real_t = float;
      using complex_t = std::complex<real_t>;

      namespace detail {
        struct CtxHandle {
          earhip_conv_ctx *h = nullptr;
          ~CtxHandle() { earhip_conv_ctx_destroy(h); }
        };
        struct FilterHandle {
          std::shared_ptr<CtxHandle> ctx;  // keeps the context alive
          earhip_conv_filter *h = nullptr;
          ~FilterHandle() { earhip_conv_filter_destroy(h); }
        };
      }  // namespace detail

      /// Static data for one block size; shareable between Filters and BlockConvolvers.
      class Context {
       public:
        /// libear's signature (include/ear/dsp/block_convolver.hpp:34).  The device path transforms on the GPU:
        /// the plugin that IS that transform — get_fft_hip(), which get_fft_kiss<float>() also returns here — is
        /// accepted; any other FFTImpl would have to run on the host (the CPU fallback this library does not
        /// have) and is refused with ear::invalid_argument instead of being dropped silently.
        Context(size_t block_size, FFTImpl<real_t> &fft) : Context(block_size) {
          if (&fft != &get_fft_hip())
            throw invalid_argument(
                "block_convolver::Context: this FFTImpl is not the device transform (pass ear::get_fft_hip() or "
                "ear::get_fft_kiss<float>(); a host FFT plugin cannot run inside the device BlockConvolver)");
        }
        explicit Context(size_t block_size) : impl(std::make_shared<detail::CtxHandle>()) {
          hip::check(earhip_conv_ctx_create(hip::default_context().get(), block_size, &impl->h));
        }

       private:
        std::shared_ptr<detail::CtxHandle> impl;
        friend class Filter;
        friend class BlockConvolver;
      };

      /// Pre-transformed filter partitions; shareable between BlockConvolvers.
      class Filter {
       public:
        Filter(const Context &ctx, size_t n, const real_t *filter)
            : impl(std::make_shared<detail::FilterHandle>()) {
          impl->ctx = ctx.impl;
          hip::check(earhip_conv_filter_create(ctx.impl->h, n, filter, &impl->h));
        }
        size_t num_blocks() const { return earhip_conv_filter_num_blocks(impl->h); }

       private:
        std::shared_ptr<detail::FilterHandle> impl;
        friend class BlockConvolver;
      };

      /// Partitioned overlap-add convolution with click-free filter changes.
      class BlockConvolver {
       public:
        BlockConvolver(const Context &ctx, size_t num_blocks) : ctx_(ctx.impl) {
          hip::check(earhip_conv_create(ctx_->h, nullptr, num_blocks, &h_));
        }
        BlockConvolver(const Context &ctx, const Filter &filter, size_t num_blocks = 0)
            : ctx_(ctx.impl) {
          hip::check(earhip_conv_create(ctx_->h, filter.impl->h, num_blocks, &h_));
        }
        ~BlockConvolver() { earhip_conv_destroy(h_); }
        BlockConvolver(const BlockConvolver &) = delete;
        BlockConvolver &operator=(const BlockConvolver &) = delete;

        /// in: block_size samples or nullptr (silence); out: block_size samples
        void process(const float *in, float *out) { hip::check(earhip_conv_process(h_, in, out)); }
        void crossfade_filter(const Filter &filter) {
          hip::check(earhip_conv_crossfade_filter(h_, filter.impl->h));
        }
        void fade_down() { hip::check(earhip_conv_crossfade_filter(h_, nullptr)); }
        void set_filter(const Filter &filter) {
          hip::check(earhip_conv_set_filter(h_, filter.impl->h));
        }
        void unset_filter() { hip::check(earhip_conv_set_filter(h_, nullptr)); }

       private:
        // (the convolver's queue holds its own references to the filters it uses, like libear's
        // shared_ptr queue, src/dsp/block_convolver_impl.hpp:154-167: a Filter may be destroyed while
        // it is still fading out)
        std::shared_ptr<detail::CtxHandle> ctx_;
        earhip_conv *h_ = nullptr;
      };
    }  // namespace block_convolver
  }  // namespace dsp
}  // namespace ear
