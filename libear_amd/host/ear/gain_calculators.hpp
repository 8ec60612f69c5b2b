// ear/gain_calculators.hpp — GainCalculatorObjects with libear's interface
// (include/ear/gain_calculators.hpp:45-56) over the device batch panner (earhip group I), plus the batched
// call a renderer wants: all metadata blocks of all objects in one launch.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "hip.hpp"
#include "metadata.hpp"
#include "warnings.hpp"

namespace ear {
  namespace detail {
    /// which channels of the full BS.2051 layout the caller's layout keeps (Layout::withoutLfe drops the LFEs),
    /// and the real positions of all channels of the full layout (the caller's where it has the channel)
    inline std::vector<int> match_layout(const Layout &layout, std::vector<double> &az, std::vector<double> &el) {
      const Layout full = getLayout(layout.name());
      std::vector<int> keep;
      for (auto &c : full.channels()) {
        az.push_back(c.polarPosition().azimuth);
        el.push_back(c.polarPosition().elevation);
      }
      for (auto &c : layout.channels()) {
        const int i = full.indexForName(c.name());
        if (i < 0) throw invalid_argument("channel " + c.name() + " is not part of layout " + layout.name());
        az[i] = c.polarPosition().azimuth;
        el[i] = c.polarPosition().elevation;
        keep.push_back(i);
      }
      return keep;
    }
  }  // namespace detail

  class GainCalculatorObjects {
   public:
    /// layout: an ITU-R BS.2051 layout (getLayout), with or without its LFE channels
    explicit GainCalculatorObjects(const Layout &layout, hip::Context &ctx = hip::default_context()) {
      std::vector<double> az, el;
      keep_ = detail::match_layout(layout, az, el);
      n_full_ = az.size();
      // the loudspeakers' real positions (Channel::polarPosition) shape the panner, the nominal ones its layers
      hip::check(earhip_panner_create_positions(ctx.get(), layout.name().c_str(), (int)n_full_, az.data(), el.data(), &h_));
    }
    ~GainCalculatorObjects() { earhip_panner_destroy(h_); }
    GainCalculatorObjects(const GainCalculatorObjects &) = delete;
    GainCalculatorObjects &operator=(const GainCalculatorObjects &) = delete;

    /// one metadata block -> direct and diffuse gain vectors, which must have the layout's channel count
    /// (libear: OutputGainsT::check_size, include/ear/helpers/output_gains.hpp:40-43)
    template <typename T>
    void calculate(const ObjectsTypeMetadata &metadata, std::vector<T> &directGains, std::vector<T> &diffuseGains,
                   const WarningCB & = default_warning_cb) {  // (libear's Objects calculator emits no warnings either)
      if (directGains.size() != keep_.size() || diffuseGains.size() != keep_.size())
        throw invalid_argument("incorrect size for output vector");
      std::vector<std::vector<T>> d, f;
      calculate(std::vector<ObjectsTypeMetadata>(1, metadata), d, f);
      directGains = d[0];
      diffuseGains = f[0];
    }
    /// a batch of metadata blocks in one device launch
    template <typename T>
    void calculate(const std::vector<ObjectsTypeMetadata> &metadata, std::vector<std::vector<T>> &directGains,
                   std::vector<std::vector<T>> &diffuseGains) {
      const size_t n = metadata.size();
      std::vector<double> az(n), el(n), dist(n), gain(n), diffuse(n), width(n), height(n), depth(n);
      bool extent = false;
      for (size_t i = 0; i < n; i++) {
        const ObjectsTypeMetadata &m = metadata[i];
        // libear's own refusals (src/object_based/gain_calculator_objects.cpp:37-44) ...
        if (m.cartesian || m.position.isCartesian) throw not_implemented("cartesian");
        if (m.objectDivergence.divergence != 0.0) throw not_implemented("divergence");
        if (m.channelLock.flag) throw not_implemented("channelLock");
        if (m.zoneExclusion.zones.size()) throw not_implemented("zoneExclusion");
        if (m.screenRef) throw not_implemented("screenRef");
        // width / height / depth: the polar extent panner (src/object_based/polar_extent.cpp:290-302)
        extent = extent || m.width != 0.0 || m.height != 0.0 || m.depth != 0.0;
        width[i] = m.width;
        height[i] = m.height;
        depth[i] = m.depth;
        az[i] = m.position.polar.azimuth;
        el[i] = m.position.polar.elevation;
        dist[i] = m.position.polar.distance;
        gain[i] = m.gain;
        diffuse[i] = m.diffuse;
      }
      std::vector<float> d(n * n_full_), f(n * n_full_);
      if (extent)
        hip::check(earhip_panner_calculate_extent(h_, n, az.data(), el.data(), dist.data(), width.data(), height.data(),
                                                  depth.data(), gain.data(), diffuse.data(), d.data(), f.data()));
      else
        hip::check(earhip_panner_calculate(h_, n, az.data(), el.data(), dist.data(), gain.data(), diffuse.data(), d.data(),
                                           f.data()));
      directGains.assign(n, std::vector<T>(keep_.size()));
      diffuseGains.assign(n, std::vector<T>(keep_.size()));
      for (size_t i = 0; i < n; i++)
        for (size_t c = 0; c < keep_.size(); c++) {
          directGains[i][c] = (T)d[i * n_full_ + keep_[c]];
          diffuseGains[i][c] = (T)f[i * n_full_ + keep_[c]];
        }
    }

   private:
    earhip_panner *h_ = nullptr;
    std::vector<int> keep_;
    size_t n_full_ = 0;
  };

  /// libear: include/ear/gain_calculators.hpp:58-70
  class GainCalculatorHOA {
   public:
    explicit GainCalculatorHOA(const Layout &layout, hip::Context &ctx = hip::default_context())
        : name_(layout.name()), ctx_(ctx) {
      keep_ = detail::match_layout(layout, az_, el_);
      n_full_ = az_.size();
    }
    /// gains[coefficient][loudspeaker] (libear's column-major vector of vectors): must have that shape
    template <typename T>
    void calculate(const HOATypeMetadata &metadata, std::vector<std::vector<T>> &gains,
                   const WarningCB &warning_cb = default_warning_cb) {
      if (metadata.orders.size() != metadata.degrees.size())
        throw invalid_argument("orders and degrees must be the same size");
      const size_t C = metadata.orders.size();
      if (gains.size() != C) throw invalid_argument("incorrect number of cols in output matrix");
      for (auto &col : gains)
        if (col.size() != keep_.size()) throw invalid_argument("incorrect number of rows in output matrix column");
      std::vector<float> D(n_full_ * (C ? C : 1));
      // (an unknown normalization is refused by the C ABI before anything is ignored, as in libear:
      // src/hoa/gain_calculator_hoa.cpp:36-48)
      if (metadata.normalization == "N3D" || metadata.normalization == "SN3D" || metadata.normalization == "FuMa") {
        if (metadata.screenRef)
          warning_cb({Warning::Code::HOA_SCREENREF_NOT_IMPLEMENTED, "screenRef for HOA is not implemented; ignoring"});
        if (metadata.nfcRefDist != 0.0)
          warning_cb({Warning::Code::HOA_NFCREFDIST_NOT_IMPLEMENTED, "nfcRefDist is not implemented; ignoring"});
      }
      hip::check(earhip_hoa_decode_matrix_positions(ctx_.get(), name_.c_str(), (int)n_full_, az_.data(), el_.data(), (int)C,
                                                    metadata.orders.data(), metadata.degrees.data(),
                                                    metadata.normalization.c_str(), D.data()));
      for (size_t c = 0; c < C; c++)
        for (size_t r = 0; r < keep_.size(); r++) gains[c][r] = (T)D[(size_t)keep_[r] * C + c];
    }

   private:
    std::string name_;
    hip::Context &ctx_;
    std::vector<int> keep_;
    std::vector<double> az_, el_;  // real loudspeaker positions, full layout
    size_t n_full_ = 0;
  };
}  // namespace ear
