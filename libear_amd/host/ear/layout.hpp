// ear/layout.hpp — what the render path needs of libear's Layout / Channel (include/ear/layout.hpp:12-95)
// and of ear::loadLayouts / ear::getLayout (include/ear/bs2051.hpp:8-11): the ITU-R BS.2051 layouts from
// the native table (earhip group H).  Same class and accessor names; getLayout gives the nominal positions, a
// caller may move the real ones (Channel::polarPosition) as with libear.
#pragma once
#include <functional>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "hip.hpp"

namespace ear {
  struct PolarPosition {
    PolarPosition(double az = 0.0, double el = 0.0, double dist = 1.0) : azimuth(az), elevation(el), distance(dist) {}
    double azimuth, elevation, distance;
  };
  struct CartesianPosition {
    CartesianPosition(double X = 0.0, double Y = 0.0, double Z = 0.0) : X(X), Y(Y), Z(Z) {}
    double X, Y, Z;
  };

  class Channel {
   public:
    Channel() = default;
    /// real position; the nominal one defaults to it (libear: boost::optional, include/ear/layout.hpp:32-38)
    Channel(const std::string &name, PolarPosition pos, bool isLfe = false) : name_(name), pos_(pos), nominal_(pos), lfe_(isLfe) {}
    Channel(const std::string &name, PolarPosition pos, PolarPosition nominal, bool isLfe = false)
        : name_(name), pos_(pos), nominal_(nominal), lfe_(isLfe) {}
    const std::string &name() const { return name_; }
    PolarPosition polarPosition() const { return pos_; }
    PolarPosition polarPositionNominal() const { return nominal_; }
    bool isLfe() const { return lfe_; }
    /// where the loudspeaker really stands (the gain calculators pan with it; :47)
    void polarPosition(PolarPosition p) { pos_ = p; }
    void polarPositionNominal(PolarPosition p) { nominal_ = p; }
    /// allowed ranges of the real position: azimuth from .first anticlockwise to .second, elevation from
    /// .first up to .second; default: the nominal position itself (:21-30, src/layout.cpp:26-33)
    std::pair<double, double> azimuthRange() const {
      return has_az_ ? az_range_ : std::make_pair(pos_.azimuth, pos_.azimuth);  // (libear: the REAL position, src/layout.cpp:24-28)
    }
    std::pair<double, double> elevationRange() const {
      return has_el_ ? el_range_ : std::make_pair(pos_.elevation, pos_.elevation);  // (src/layout.cpp:29-33)
    }
    void azimuthRange(std::pair<double, double> r) { az_range_ = r, has_az_ = true; }
    void elevationRange(std::pair<double, double> r) { el_range_ = r, has_el_ = true; }
    /// reports a real position outside the allowed ranges through the callback (src/layout.cpp:54-75)
    void checkPosition(std::function<void(const std::string &)> callback) const {
      if (has_az_ && !inside_angle_range(pos_.azimuth, az_range_.first, az_range_.second)) {
        std::stringstream ss;
        ss << name_ << ": azimuth " << pos_.azimuth << " out of range [" << az_range_.first << ", " << az_range_.second << "]";
        callback(ss.str());
      }
      if (has_el_ && !(el_range_.first <= pos_.elevation && pos_.elevation <= el_range_.second)) {
        std::stringstream ss;
        ss << name_ << ": elevation " << pos_.elevation << " out of range [" << el_range_.first << ", " << el_range_.second
           << "]";
        callback(ss.str());
      }
    }

   private:
    // src/common/geom.cpp:7-28: is x within the range from start anticlockwise to end?  (-180, 180) is any
    // angle, (-180, -180) a single one
    static bool inside_angle_range(double x, double start, double end, double tol = 0.0) {
      while (end - 360.0 > start) end -= 360.0;
      while (end < start) end += 360.0;
      const double start_tol = start - tol;
      while (x - 360.0 >= start_tol) x -= 360.0;
      while (x < start_tol) x += 360.0;
      return x <= end + tol;
    }
    std::string name_;
    PolarPosition pos_, nominal_;
    std::pair<double, double> az_range_{0.0, 0.0}, el_range_{0.0, 0.0};
    bool has_az_ = false, has_el_ = false;
    bool lfe_ = false;
  };

  class Layout {
   public:
    Layout(std::string name = "", std::vector<Channel> channels = std::vector<Channel>())
        : name_(std::move(name)), channels_(std::move(channels)) {}
    std::string name() const { return name_; }
    std::vector<Channel> &channels() { return channels_; }
    const std::vector<Channel> &channels() const { return channels_; }
    Layout withoutLfe() const {
      Layout l(name_);
      for (auto &c : channels_)
        if (!c.isLfe()) l.channels_.push_back(c);
      return l;
    }
    std::vector<bool> isLfe() const {
      std::vector<bool> v;
      for (auto &c : channels_) v.push_back(c.isLfe());
      return v;
    }
    std::vector<std::string> channelNames() const {
      std::vector<std::string> v;
      for (auto &c : channels_) v.push_back(c.name());
      return v;
    }
    void checkPositions(std::function<void(const std::string &)> callback) const {
      for (auto &c : channels_) c.checkPosition(callback);
    }
    /// (libear returns a copy of the first match; an unknown name is undefined there, an exception here)
    Channel channelWithName(const std::string &name) const {
      for (auto &c : channels_)
        if (c.name() == name) return c;
      throw invalid_argument("no channel named " + name + " in layout " + name_);
    }
    std::vector<PolarPosition> positions() const {
      std::vector<PolarPosition> v;
      for (auto &c : channels_) v.push_back(c.polarPosition());
      return v;
    }
    std::vector<PolarPosition> nominalPositions() const {
      std::vector<PolarPosition> v;
      for (auto &c : channels_) v.push_back(c.polarPositionNominal());
      return v;
    }
    /// index of the channel with this name, or -1 (libear: boost::optional<int>)
    int indexForName(const std::string &name) const {
      for (size_t i = 0; i < channels_.size(); i++)
        if (channels_[i].name() == name) return (int)i;
      return -1;
    }

   private:
    std::string name_;
    std::vector<Channel> channels_;
  };

  /// Get a layout given its ITU-R BS.2051 name (e.g. `4+5+0`); unknown names throw (libear: unknown_layout).
  inline Layout getLayout(const std::string &name) {
    int n = 0;
    hip::check(earhip_layout_num_channels(name.c_str(), &n));
    std::vector<Channel> ch;
    for (int i = 0; i < n; i++) {
      const char *cn = nullptr;
      double az = 0, el = 0;
      int lfe = 0;
      hip::check(earhip_layout_channel(name.c_str(), i, &cn, &az, &el, &lfe));
      ch.emplace_back(cn, PolarPosition(az, el, 1.0), lfe != 0);
      double azr[2], elr[2];
      hip::check(earhip_layout_channel_ranges(name.c_str(), i, azr, elr));
      ch.back().azimuthRange(std::make_pair(azr[0], azr[1]));
      ch.back().elevationRange(std::make_pair(elr[0], elr[1]));
    }
    return Layout(name, ch);
  }
  /// Get all ITU-R BS.2051 layouts.
  inline std::vector<Layout> loadLayouts() {
    std::vector<Layout> v;
    for (int i = 0; i < earhip_layout_count(); i++) v.push_back(getLayout(earhip_layout_name(i)));
    return v;
  }
}  // namespace ear
