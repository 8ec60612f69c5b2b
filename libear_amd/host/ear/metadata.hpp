// ear/metadata.hpp — the metadata the gain producers read: type names, field names and defaults of libear's
// ObjectsTypeMetadata and HOATypeMetadata (include/ear/metadata.hpp:66-171).  libear's boost::variant members
// (position, objectDivergence, exclusion zones) are plain structs here that convert from the same alternative
// types, so the lines a libear application writes — `otm.position = PolarPosition(...)`, `otm.objectDivergence =
// PolarObjectDivergence(0.5)`, `zones.push_back(PolarExclusionZone{...})` — compile unchanged.
#pragma once
#include <string>
#include <vector>

#include "layout.hpp"
#include "screen.hpp"

namespace ear {
  /// libear: boost::variant<CartesianPosition, PolarPosition> (common_types.hpp:26)
  struct Position {
    Position(PolarPosition p = PolarPosition()) : isCartesian(false), polar(p) {}
    Position(CartesianPosition c) : isCartesian(true), cartesian(c) {}
    bool isCartesian;
    PolarPosition polar;
    CartesianPosition cartesian;
  };
  struct ChannelLock {
    ChannelLock(bool flag = false) : flag(flag) {}
    /// (libear: boost::optional<double> maxDistance)
    ChannelLock(bool flag, double maxDistance) : flag(flag), hasMaxDistance(true), maxDistance(maxDistance) {}
    bool flag;
    bool hasMaxDistance = false;
    double maxDistance = 0.0;
  };
  struct PolarObjectDivergence {
    PolarObjectDivergence(double divergence = 0.0, double azimuthRange = 45.0)
        : divergence(divergence), azimuthRange(azimuthRange) {}
    double divergence;
    double azimuthRange;
  };
  struct CartesianObjectDivergence {
    CartesianObjectDivergence(double divergence = 0.0, double positionRange = 0.0)
        : divergence(divergence), positionRange(positionRange) {}
    double divergence;
    double positionRange;
  };
  /// libear: boost::variant<PolarObjectDivergence, CartesianObjectDivergence>; the calculators only read
  /// `divergence` (and refuse anything but 0)
  struct ObjectDivergence {
    ObjectDivergence(double divergence = 0.0, double range = 45.0) : divergence(divergence), range(range) {}
    ObjectDivergence(PolarObjectDivergence d) : divergence(d.divergence), range(d.azimuthRange) {}
    ObjectDivergence(CartesianObjectDivergence d) : isCartesian(true), divergence(d.divergence), range(d.positionRange) {}
    bool isCartesian = false;
    double divergence, range;
  };
  struct PolarExclusionZone {
    float minAzimuth;
    float maxAzimuth;
    float minElevation;
    float maxElevation;
    float minDistance;
    float maxDistance;
    std::string label;
  };
  struct CartesianExclusionZone {
    float minX;
    float maxX;
    float minY;
    float maxY;
    float minZ;
    float maxZ;
    std::string label;
  };
  /// libear: boost::variant<PolarExclusionZone, CartesianExclusionZone>
  struct ExclusionZone {
    ExclusionZone() = default;
    ExclusionZone(PolarExclusionZone z) : isCartesian(false), polar(std::move(z)) {}
    ExclusionZone(CartesianExclusionZone z) : isCartesian(true), cartesian(std::move(z)) {}
    bool isCartesian = false;
    PolarExclusionZone polar = {};
    CartesianExclusionZone cartesian = {};
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
    bool screenRef = false;  ///< refused when set, as in libear
    Screen referenceScreen = getDefaultScreen();
  };

  struct HOATypeMetadata {
    std::vector<int> orders;
    std::vector<int> degrees;
    std::string normalization = std::string("SN3D");
    double nfcRefDist = 0.0;  ///< ignored, as in libear (which warns)
    bool screenRef = false;   ///< ignored, as in libear (which warns)
    Screen referenceScreen = getDefaultScreen();
  };
}  // namespace ear
