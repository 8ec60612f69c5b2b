// ear/dsp/objects_renderer.hpp — the composed Objects render block libear
// documents but does not ship (docs/dsp.rst:40-71): interpolated direct/diffuse
// gains -> buses -> decorrelators + compensation delay -> mix.  This is the
// batched GPU boundary: it has the shape of VariableBlockSizeAdapter's
// ProcessFunc (one block in, one block out) and a multi-block stream variant.
#pragma once
#include <cstdint>
#include <vector>
#include "../hip.hpp"

namespace ear {
  namespace dsp {
    class ObjectsRenderer {
     public:
      /// decorrelators: designDecorrelators(channel names); empty = direct bus only
      ObjectsRenderer(size_t n_objects, size_t n_out, size_t block_size,
                      const std::vector<std::vector<float>> &decorrelators, int delay,
                      size_t max_blocks = 1, hip::Context &ctx = hip::default_context())
          : n_objects_(n_objects), n_out_(n_out), block_size_(block_size), two_buses_(!decorrelators.empty()) {
        earhip_render_config cfg;
        cfg.n_objects = (int)n_objects;
        cfg.n_out = (int)n_out;
        cfg.block_size = (int)block_size;
        cfg.n_buses = decorrelators.empty() ? 1 : 2;
        std::vector<float> flat;
        cfg.n_taps = 0;
        if (!decorrelators.empty()) {
          if (decorrelators.size() != n_out) throw invalid_argument("one decorrelator per output");
          cfg.n_taps = (int)decorrelators[0].size();
          for (auto &d : decorrelators) {
            if ((int)d.size() != cfg.n_taps) throw invalid_argument("decorrelators differ in length");
            flat.insert(flat.end(), d.begin(), d.end());
          }
        }
        cfg.decorrelators = flat.empty() ? nullptr : flat.data();
        cfg.delay = delay;
        cfg.max_blocks = (int)max_blocks;
        hip::check(earhip_render_create(ctx.get(), &cfg, &h_));
      }
      ~ObjectsRenderer() { earhip_render_destroy(h_); }
      ObjectsRenderer(const ObjectsRenderer &) = delete;
      ObjectsRenderer &operator=(const ObjectsRenderer &) = delete;

      /// Gain curve of one object: same meaning as GainInterpolator::interp_points
      /// of its direct and diffuse LinearInterpVector interpolators.
      void set_object_points(size_t object, const std::vector<int64_t> &times,
                             const std::vector<std::vector<float>> &direct,
                             const std::vector<std::vector<float>> &diffuse) {
        if (direct.size() != times.size()) throw invalid_argument("one direct gain vector per time");
        if (two_buses_ ? diffuse.size() != times.size() : !diffuse.empty())
          throw invalid_argument(two_buses_ ? "one diffuse gain vector per time"
                                            : "diffuse gains given to a renderer without a diffuse bus");
        std::vector<float> d, f;
        for (auto &p : direct) {
          if (p.size() != n_out_) throw invalid_argument("gain vector length != number of outputs");
          d.insert(d.end(), p.begin(), p.end());
        }
        for (auto &p : diffuse) {
          if (p.size() != n_out_) throw invalid_argument("gain vector length != number of outputs");
          f.insert(f.end(), p.begin(), p.end());
        }
        hip::check(earhip_render_set_object_points(h_, (int)object, (int)times.size(), times.data(),
                                                   d.data(), f.empty() ? nullptr : f.data()));
      }
      /// one block: in[n_objects][block_size] -> out[n_out][block_size] (ProcessFunc shape)
      void process(const float *const *in, float *const *out) {
        hip::check(earhip_render_process(h_, 1, in, out));
      }
      /// nblocks consecutive blocks per call
      void process(size_t nblocks, const float *const *in, float *const *out) {
        hip::check(earhip_render_process(h_, nblocks, in, out));
      }
      /// device-resident planar buffers, asynchronous on the context's stream
      void process_device(size_t nblocks, const float *in_dev, size_t in_stride, float *out_dev,
                          size_t out_stride) {
        hip::check(earhip_render_process_device(h_, nblocks, in_dev, in_stride, out_dev, out_stride));
      }
      void reset(int64_t sample_time = 0) { hip::check(earhip_render_reset(h_, sample_time)); }
      size_t block_size() const { return block_size_; }

     private:
      size_t n_objects_, n_out_, block_size_;
      bool two_buses_;
      earhip_render *h_ = nullptr;
    };
  }  // namespace dsp
}  // namespace ear
