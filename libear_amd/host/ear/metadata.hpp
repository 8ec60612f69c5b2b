// ear/metadata.hpp — the Objects metadata the gain producer reads, field names and defaults of libear's
// ObjectsTypeMetadata (include/ear/metadata.hpp:88-160).  libear's variants (boost) are plain structs with a
// discriminating flag here.
#pragma once
#include <vector>

#include "layout.hpp"

namespace ear {
  /// libear: boost::variant<PolarPosition, CartesianPosition>
  struct Position {
    Position(PolarPosition p = PolarPosition()) : isCartesian(false), polar(p) {}
    Position(CartesianPosition c) : isCartesian(true), cartesian(c) {}
    bool isCartesian;
    PolarPosition polar;
    CartesianPosition cartesian;
  };
  struct ChannelLock {
    ChannelLock(bool flag = false) : flag(flag) {}
    bool flag;
  };
  /// libear: variant of PolarObjectDivergence / CartesianObjectDivergence; only `divergence` is read
  struct ObjectDivergence {
    ObjectDivergence(double divergence = 0.0, double range = 45.0) : divergence(divergence), range(range) {}
    double divergence, range;
  };
  struct ExclusionZone {
    float min[3], max[3];
  };
  struct ZoneExclusion {
    std::vector<ExclusionZone> zones;
  };
  struct ObjectsTypeMetadata {
    Position position = {};
    double width = 0.0;
    double height = 0.0;
    double depth = 0.0;
    bool cartesian = false;
    double gain = 1.0;
    double diffuse = 0.0;
    ChannelLock channelLock = {};
    ObjectDivergence objectDivergence = {};
    ZoneExclusion zoneExclusion = {};
    bool screenRef = false;
  };
}  // namespace ear
