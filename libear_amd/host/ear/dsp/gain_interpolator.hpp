// ear/dsp/gain_interpolator.hpp — GainInterpolator and its interpolation
// policies with libear's interface (include/ear/dsp/gain_interpolator.hpp), the
// policies' arithmetic running on the GPU through earhip_interp_apply_*.
//
// The segment walk (which curve segment a sample range falls into, constant vs
// ramp) is host control flow exactly as in libear (:53-129); each segment is one
// device call on host pointers.  For throughput use ObjectsRenderer
// (objects_renderer.hpp), which keeps curves and audio on the device.
#pragma once
#include <algorithm>
#include <utility>
#include <vector>

#include "../helpers/assert.hpp"
#include "../hip.hpp"

namespace ear {
  namespace dsp {
    using SampleIndex = long int;

    template <typename InterpType>
    class GainInterpolator {
     public:
      /// (sample index, gain values) pairs sorted by time; duplicates = steps
      std::vector<std::pair<SampleIndex, typename InterpType::Point>> interp_points;

      void process(SampleIndex block_start, size_t nsamples, const float *const *in,
                   float *const *out) {
        // libear reads interp_points[-1] when empty (undefined); defined as an error here
        if (interp_points.empty()) throw invalid_argument("interp_points must not be empty");
        const SampleIndex block_end = block_start + (SampleIndex)nsamples;
        const size_t n = interp_points.size();
        SampleIndex cur = block_start;
        while (cur < block_end) {
          const size_t k = locate(cur);
          const SampleIndex seg_end =
              k == n ? block_end : std::min(interp_points[k].first, block_end);
          ear_assert(cur < seg_end, "found block ends before processed block starts");
          if (k == 0 || k == n ||
              InterpType::constant_interp(interp_points[k - 1].second, interp_points[k].second)) {
            InterpType::apply_constant(in, out, cur - block_start, seg_end - block_start,
                                       interp_points[k == n ? k - 1 : k].second);
          } else {
            InterpType::apply_interp(in, out, cur - block_start, seg_end - block_start,
                                     block_start, interp_points[k - 1].first,
                                     interp_points[k].first, interp_points[k - 1].second,
                                     interp_points[k].second);
          }
          cur = seg_end;
        }
      }

     private:
      size_t hint_ = 0;
      int side(size_t k, SampleIndex t) const {
        if (k > 0 && t < interp_points[k - 1].first) return -1;
        if (k < interp_points.size() && t >= interp_points[k].first) return 1;
        return 0;
      }
      size_t locate(SampleIndex t) {
        if (hint_ > interp_points.size()) hint_ = 0;
        const int first = side(hint_, t);
        int dir = first;
        while (dir != 0) {
          hint_ += dir;
          if (dir != first) throw invalid_argument("interpolation points are not sorted");
          dir = side(hint_, t);
        }
        return hint_;
      }
    };

    template <typename PointT>
    struct InterpType {
      using Point = PointT;
      static bool constant_interp(const Point &a, const Point &b) { return a == b; }
    };

    /// 1 -> 1
    struct LinearInterpSingle : public InterpType<float> {
      static void apply_interp(const float *const *in, float *const *out, SampleIndex range_start,
                               SampleIndex range_end, SampleIndex block_start, SampleIndex start,
                               SampleIndex end, const Point &start_point, const Point &end_point) {
        hip::check(earhip_interp_apply_interp(hip::default_context().get(), 1, 1, in, out,
                                              range_start, range_end, block_start, start, end,
                                              &start_point, &end_point));
      }
      static void apply_constant(const float *const *in, float *const *out, SampleIndex range_start,
                                 SampleIndex range_end, const Point &point) {
        hip::check(earhip_interp_apply_constant(hip::default_context().get(), 1, 1, in, out,
                                                range_start, range_end, &point));
      }
    };

    /// 1 -> N
    struct LinearInterpVector : public InterpType<std::vector<float>> {
      static void apply_interp(const float *const *in, float *const *out, SampleIndex range_start,
                               SampleIndex range_end, SampleIndex block_start, SampleIndex start,
                               SampleIndex end, const Point &start_point, const Point &end_point) {
        if (start_point.empty()) return;
        ear_assert(start_point.size() == end_point.size(), "points differ in size");
        hip::check(earhip_interp_apply_interp(hip::default_context().get(), 1,
                                              (int)start_point.size(), in, out, range_start,
                                              range_end, block_start, start, end,
                                              start_point.data(), end_point.data()));
      }
      static void apply_constant(const float *const *in, float *const *out, SampleIndex range_start,
                                 SampleIndex range_end, const Point &point) {
        if (point.empty()) return;
        hip::check(earhip_interp_apply_constant(hip::default_context().get(), 1, (int)point.size(),
                                                in, out, range_start, range_end, point.data()));
      }
    };

    /// M -> N; points hold one vector of per-output gains per input channel
    struct LinearInterpMatrix : public InterpType<std::vector<std::vector<float>>> {
      /// dense row-major copy of a point in per-thread scratch `which` (grown once: no allocation in
      /// process(), like libear — tests/gain_interpolator_tests.cpp:1,89,96)
      static const std::vector<float> &flatten(const Point &p, int which) {
        static thread_local std::vector<float> scratch[2];
        std::vector<float> &flat = scratch[which];
        const size_t n_out = p.empty() ? 0 : p[0].size();
        flat.resize(p.size() * n_out);
        for (size_t i = 0; i < p.size(); i++) {
          ear_assert(p[i].size() == n_out, "ragged gain matrix");
          std::copy(p[i].begin(), p[i].end(), flat.begin() + i * n_out);
        }
        return flat;
      }
      static void apply_interp(const float *const *in, float *const *out, SampleIndex range_start,
                               SampleIndex range_end, SampleIndex block_start, SampleIndex start,
                               SampleIndex end, const Point &start_point, const Point &end_point) {
        if (start_point.empty() || start_point[0].empty()) return;
        const std::vector<float> &s = flatten(start_point, 0), &e = flatten(end_point, 1);
        ear_assert(s.size() == e.size(), "points differ in size");
        hip::check(earhip_interp_apply_interp(hip::default_context().get(), (int)start_point.size(),
                                              (int)start_point[0].size(), in, out, range_start,
                                              range_end, block_start, start, end, s.data(),
                                              e.data()));
      }
      static void apply_constant(const float *const *in, float *const *out, SampleIndex range_start,
                                 SampleIndex range_end, const Point &point) {
        if (point.empty() || point[0].empty()) return;
        const std::vector<float> &p = flatten(point, 0);
        hip::check(earhip_interp_apply_constant(hip::default_context().get(), (int)point.size(),
                                                (int)point[0].size(), in, out, range_start,
                                                range_end, p.data()));
      }
    };
  }  // namespace dsp
}  // namespace ear
