// vsf_frontend.hpp -- C++17 host-side mirror of the reference's feature interfaces over the C ABI
// (include/vsf.h).  Header only; needs nothing but the C++ standard library and libvsf_hip.so.
//
// Mirrors, with the same names, argument meaning and state:
//   vslam::feature::FeatureDetectorBase            ref: src/vslam/feature/FeatureDetectorBase.hpp:13-21
//   vslam::feature::FeatureDetectorShiTomasiBucketing   FeatureDetectorShiTomasiBucketing.hpp:30-59
//   vslam::feature::FeatureTrackerBase              ref: src/vslam/feature/FeatureTrackerBase.hpp:16-26
//   vslam::feature::FeatureTrackerLucasKanade            FeatureTrackerLucasKanade.hpp:17-48
//   vslam::FeaturePositionDatabase / FeatureAgeDatabase   src/vslam/type_defines.hpp:19,95-99
// The reference passes cv::Mat and Eigen::Vector2d; neither OpenCV nor Eigen exists on this image,
// so the mirror takes an ImageView (pointer, size, step: exactly what a CV_8UC1 cv::Mat holds) and
// a two-double Vec2_t.  With -DVSF_WITH_OPENCV (where OpenCV and Eigen are installed) overloads
// taking cv::Mat are compiled as well, which makes the classes literal drop-ins.
//
// Error behaviour: the reference's methods return void and only OpenCV assertions throw
// (cv::Exception).  Here every VSF_E_* code becomes a std::runtime_error carrying vsf_last_error().
// Unlike the reference there is no cv::imshow side effect and no spdlog output; the
// three counts Track logs are available from last_stats().
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "vsf.h"

#ifdef VSF_WITH_OPENCV
#include <opencv2/core.hpp>
#endif

namespace vslam {

using database_index_t = uint64_t;

struct Vec2_t {
  double v[2] = {0, 0};
  Vec2_t() = default;
  Vec2_t(double x, double y) : v{x, y} {}
  double& operator[](int i) { return v[i]; }
  const double& operator[](int i) const { return v[i]; }
};

using FeaturePositionDatabase = std::unordered_map<database_index_t, Vec2_t>;
using FeatureAgeDatabase = std::unordered_map<database_index_t, uint32_t>;

namespace feature {

// What an 8-bit cv::Mat provides: data, cols, rows, step, channels.  The detectors take 1 channel, or 3 / 4
// interleaved (BGR / BGRA), which they convert like the reference does (cv::cvtColor CV_BGR2GRAY,
// FeatureDetectorShiTomasiBucketing.cpp:105-110); the trackers take 1 channel.
struct ImageView {
  const uint8_t* data = nullptr;
  int cols = 0, rows = 0;
  int step = 0;
  int channels = 1;
  ImageView() = default;
  ImageView(const uint8_t* d, int c, int r, int s = 0, int cn = 1)
      : data(d), cols(c), rows(r), step(s ? s : c * cn), channels(cn) {}
  bool empty() const { return data == nullptr || cols == 0 || rows == 0; }
#ifdef VSF_WITH_OPENCV
  ImageView(const cv::Mat& m) : data(m.data), cols(m.cols), rows(m.rows), step((int)m.step), channels(m.channels()) {  // NOLINT
    if (!m.empty() && m.depth() != CV_8U) throw std::runtime_error("vsf: 8-bit image expected");
  }
#endif
};

namespace detail {

struct CtxDeleter {
  void operator()(vsf_ctx* c) const { vsf_destroy(c); }
};
using CtxPtr = std::unique_ptr<vsf_ctx, CtxDeleter>;

inline void check(vsf_ctx* c, int rc, const char* what) {
  if (rc != VSF_OK)
    throw std::runtime_error(std::string(what) + ": vsf error " + std::to_string(rc) + " " +
                             (c ? vsf_last_error(c) : ""));
}

// unordered_map pair -> SoA (the key sets of the two maps are the same in the reference)
inline void flatten(const FeaturePositionDatabase& pos, const FeatureAgeDatabase& age, size_t capacity,
                    std::vector<uint64_t>& ids, std::vector<double>& xy, std::vector<uint32_t>& ag) {
  ids.assign(capacity, 0);
  xy.assign(2 * capacity, 0.0);
  ag.assign(capacity, 0);
  size_t i = 0;
  for (const auto& [id, p] : pos) {
    ids[i] = id;
    xy[2 * i] = p[0];
    xy[2 * i + 1] = p[1];
    auto it = age.find(id);
    ag[i] = it == age.end() ? 0u : it->second;
    i++;
  }
}

inline void unflatten(const std::vector<uint64_t>& ids, const std::vector<double>& xy, const std::vector<uint32_t>& ag,
                      int n, FeaturePositionDatabase& pos, FeatureAgeDatabase& age) {
  pos.clear();
  age.clear();
  for (int i = 0; i < n; i++) {
    pos[ids[i]] = Vec2_t(xy[2 * i], xy[2 * i + 1]);
    age[ids[i]] = ag[i];
  }
}

}  // namespace detail

class FeatureDetectorBase {
 public:
  FeatureDetectorBase() = default;
  virtual ~FeatureDetectorBase() = default;
  virtual void UpdateDetection(FeaturePositionDatabase& feature_position, FeatureAgeDatabase& feature_age,
                               const ImageView& current_image, const ImageView& mask_image = ImageView()) = 0;
};

class FeatureDetectorShiTomasiBucketing : public FeatureDetectorBase {
 public:
  FeatureDetectorShiTomasiBucketing(int32_t division_number_row, int32_t division_number_col,
                                    int32_t max_feature_number, double min_feature_distance, int device_id = 0)
      : division_number_row_(division_number_row),
        division_number_col_(division_number_col),
        max_feature_number_(max_feature_number),
        min_feature_distance_(min_feature_distance),
        device_id_(device_id) {}

  void UpdateDetection(FeaturePositionDatabase& feature_position, FeatureAgeDatabase& feature_age,
                       const ImageView& current_image, const ImageView& mask_image = ImageView()) override {
    const size_t need = feature_position.size() + (size_t)max_feature_number_ + 64;
    ensure(current_image.cols, current_image.rows, need);
    std::vector<uint64_t> ids;
    std::vector<double> xy;
    std::vector<uint32_t> ag;
    detail::flatten(feature_position, feature_age, capacity_, ids, xy, ag);
    int32_t n = (int32_t)feature_position.size();
    detail::check(ctx_.get(),
                  current_image.channels == 1
                      ? vsf_detect(ctx_.get(), current_image.data, current_image.cols, current_image.rows,
                                   current_image.step, mask_image.empty() ? nullptr : mask_image.data, mask_image.step,
                                   ids.data(), xy.data(), ag.data(), &n, (int32_t)capacity_)
                      : vsf_detect_bgr(ctx_.get(), current_image.data, current_image.cols, current_image.rows,
                                       current_image.step, current_image.channels,
                                       mask_image.empty() ? nullptr : mask_image.data, mask_image.step, ids.data(),
                                       xy.data(), ag.data(), &n, (int32_t)capacity_),
                  "UpdateDetection");
    detail::unflatten(ids, xy, ag, n, feature_position, feature_age);
  }

  database_index_t max_feature_index() {  // max_feature_index_ of the reference object
    uint64_t v = 0;
    if (ctx_) detail::check(ctx_.get(), vsf_get_max_feature_index(ctx_.get(), 0, &v), "max_feature_index");
    return v;
  }

 private:
  void ensure(int w, int h, size_t need) {
    if (ctx_ && w == w_ && h == h_ && need <= capacity_) return;
    uint64_t keep = 0;
    if (ctx_) vsf_get_max_feature_index(ctx_.get(), 0, &keep);  // the counter survives a re-size
    vsf_params p;
    vsf_default_params(&p);
    p.grid_rows = division_number_row_;
    p.grid_cols = division_number_col_;
    p.max_features = max_feature_number_;
    p.min_distance = min_feature_distance_;
    p.width = w;
    p.height = h;
    p.device_id = device_id_;
    p.capacity = (int32_t)(need + need / 2);
    vsf_ctx* c = nullptr;
    detail::check(nullptr, vsf_create(&p, &c), "vsf_create(detector)");
    ctx_.reset(c);
    w_ = w;
    h_ = h;
    capacity_ = (size_t)p.capacity;
    detail::check(c, vsf_set_max_feature_index(c, 0, keep), "restore counter");
  }

  int32_t division_number_row_, division_number_col_, max_feature_number_;
  double min_feature_distance_;
  int device_id_;
  detail::CtxPtr ctx_;
  int w_ = 0, h_ = 0;
  size_t capacity_ = 0;
};

// Mirror of vslam::feature::FeatureDetectorANMS (ref: src/vslam/feature/FeatureDetectorANMS.hpp,
// FeatureDetectorANMS.cpp:13-67): same Parameter fields, same UpdateDetection.
class FeatureDetectorANMS : public FeatureDetectorBase {
 public:
  class Parameter {
   public:
    Parameter()  // FeatureDetectorANMS.cpp:13-35
        : max_feature_number_(300),
          min_feature_distance_(10.0),
          detection_feature_number_(600),
          detection_min_feature_distance_(1.0),
          detection_quality_level_(0.001),
          detection_block_size_(4),
          detection_use_harris_corner_(false),
          detection_k_(0.04),
          detection_ignore_edge_(20),
          non_max_suppression_tolerance_(0.1) {}
    int32_t max_feature_number_;
    double min_feature_distance_;
    int32_t detection_feature_number_;
    double detection_min_feature_distance_;
    double detection_quality_level_;
    int32_t detection_block_size_;
    bool detection_use_harris_corner_;
    double detection_k_;
    int32_t detection_ignore_edge_;
    double non_max_suppression_tolerance_;
  };

  explicit FeatureDetectorANMS(const Parameter parameter, int device_id = 0)
      : parameter_(parameter), device_id_(device_id) {}

  void UpdateDetection(FeaturePositionDatabase& feature_position, FeatureAgeDatabase& feature_age,
                       const ImageView& current_image, const ImageView& mask_image = ImageView()) override {
    const size_t need = feature_position.size() + (size_t)parameter_.detection_feature_number_ + 64;
    ensure(current_image.cols, current_image.rows, need);
    std::vector<uint64_t> ids;
    std::vector<double> xy;
    std::vector<uint32_t> ag;
    detail::flatten(feature_position, feature_age, capacity_, ids, xy, ag);
    int32_t n = (int32_t)feature_position.size();
    detail::check(ctx_.get(),
                  current_image.channels == 1
                      ? vsf_detect(ctx_.get(), current_image.data, current_image.cols, current_image.rows,
                                   current_image.step, mask_image.empty() ? nullptr : mask_image.data, mask_image.step,
                                   ids.data(), xy.data(), ag.data(), &n, (int32_t)capacity_)
                      : vsf_detect_bgr(ctx_.get(), current_image.data, current_image.cols, current_image.rows,
                                       current_image.step, current_image.channels,
                                       mask_image.empty() ? nullptr : mask_image.data, mask_image.step, ids.data(),
                                       xy.data(), ag.data(), &n, (int32_t)capacity_),
                  "UpdateDetection");
    detail::unflatten(ids, xy, ag, n, feature_position, feature_age);
  }

  database_index_t max_feature_index() {
    uint64_t v = 0;
    if (ctx_) detail::check(ctx_.get(), vsf_get_max_feature_index(ctx_.get(), 0, &v), "max_feature_index");
    return v;
  }

 private:
  void ensure(int w, int h, size_t need) {
    if (ctx_ && w == w_ && h == h_ && need <= capacity_) return;
    uint64_t keep = 0;
    if (ctx_) vsf_get_max_feature_index(ctx_.get(), 0, &keep);
    vsf_params p;
    vsf_default_params(&p);
    p.detector_kind = VSF_DETECTOR_ANMS;
    p.max_features = parameter_.max_feature_number_;
    p.min_distance = parameter_.min_feature_distance_;
    p.anms_detection_features = parameter_.detection_feature_number_;
    p.anms_quality_level = parameter_.detection_quality_level_;
    p.anms_block_size = parameter_.detection_block_size_;
    p.anms_ignore_edge = parameter_.detection_ignore_edge_;
    p.anms_tolerance = parameter_.non_max_suppression_tolerance_;
    p.anms_detection_min_distance = parameter_.detection_min_feature_distance_;
    p.anms_use_harris = parameter_.detection_use_harris_corner_ ? 1 : 0;
    p.anms_harris_k = parameter_.detection_k_;
    p.width = w;
    p.height = h;
    p.device_id = device_id_;
    p.capacity = (int32_t)(need + need / 2);
    vsf_ctx* c = nullptr;
    detail::check(nullptr, vsf_create(&p, &c), "vsf_create(ANMS detector)");
    ctx_.reset(c);
    w_ = w;
    h_ = h;
    capacity_ = (size_t)p.capacity;
    detail::check(c, vsf_set_max_feature_index(c, 0, keep), "restore counter");
  }

  Parameter parameter_;
  int device_id_;
  detail::CtxPtr ctx_;
  int w_ = 0, h_ = 0;
  size_t capacity_ = 0;
};

class FeatureTrackerBase {
 public:
  FeatureTrackerBase() = default;
  virtual ~FeatureTrackerBase() = default;
  virtual void Track(FeaturePositionDatabase& feature_position, FeatureAgeDatabase& feature_age,
                     const ImageView& current_frame) = 0;
};

class FeatureTrackerLucasKanade : public FeatureTrackerBase {
 public:
  FeatureTrackerLucasKanade(int32_t klt_max_iteration, double klt_epsilon, int32_t klt_window_size,
                            int32_t klt_max_level, double backtrack_distance_threshold, int device_id = 0)
      : klt_max_iteration_(klt_max_iteration),
        klt_epsilon_(klt_epsilon),
        klt_window_size_(klt_window_size),
        klt_max_level_(klt_max_level),
        backtrack_distance_threshold_(backtrack_distance_threshold),
        device_id_(device_id) {}

  // 3-argument stateful form: the first call only stores the frame.
  void Track(FeaturePositionDatabase& feature_position, FeatureAgeDatabase& feature_age,
             const ImageView& current_frame) override {
    run(feature_position, feature_age, nullptr, current_frame);
  }
  // 4-argument stateless form.
  void Track(FeaturePositionDatabase& feature_position, FeatureAgeDatabase& feature_age, const ImageView& prev_image,
             const ImageView& current_frame) {
    run(feature_position, feature_age, &prev_image, current_frame);
  }

  const vsf_track_stats& last_stats() const { return stats_; }

 private:
  void run(FeaturePositionDatabase& pos, FeatureAgeDatabase& age, const ImageView* prev, const ImageView& cur) {
    ensure(cur.cols, cur.rows, pos.size() + 64);
    std::vector<uint64_t> ids;
    std::vector<double> xy;
    std::vector<uint32_t> ag;
    detail::flatten(pos, age, capacity_, ids, xy, ag);
    int32_t n = (int32_t)pos.size();
    const bool first = !prev && !has_prev_;
    if (prev)
      detail::check(ctx_.get(),
                    vsf_track_pair(ctx_.get(), prev->data, prev->step, cur.data, cur.step, cur.cols, cur.rows,
                                   ids.data(), xy.data(), ag.data(), &n, &stats_),
                    "Track(prev, cur)");
    else
      detail::check(ctx_.get(),
                    vsf_track(ctx_.get(), cur.data, cur.cols, cur.rows, cur.step, ids.data(), xy.data(), ag.data(),
                              &n, &stats_),
                    "Track");
    if (!prev) has_prev_ = true;
    if (first) return;                         // ref :29-32: databases untouched
    if (pos.empty() || age.empty()) return;    // ref :195-198: empty inputs are returned as they are
    detail::unflatten(ids, xy, ag, n, pos, age);
  }

  void ensure(int w, int h, size_t need) {
    if (ctx_ && w == w_ && h == h_ && need <= capacity_) return;
    if (ctx_ && (w != w_ || h != h_)) has_prev_ = false;  // a frame of another size cannot be tracked into
    const bool grow_only = ctx_ && w == w_ && h == h_;
    if (grow_only && has_prev_)
      throw std::runtime_error("vsf: feature table grew beyond the tracker capacity between frames");
    vsf_params p;
    vsf_default_params(&p);
    p.klt_max_iter = klt_max_iteration_;
    p.klt_eps = klt_epsilon_;
    p.klt_win = klt_window_size_;
    p.klt_max_level = klt_max_level_;
    p.backtrack_thr = backtrack_distance_threshold_;
    p.width = w;
    p.height = h;
    p.device_id = device_id_;
    p.capacity = (int32_t)(need < 4096 ? 4096 : need * 2);
    vsf_ctx* c = nullptr;
    detail::check(nullptr, vsf_create(&p, &c), "vsf_create(tracker)");
    ctx_.reset(c);
    w_ = w;
    h_ = h;
    capacity_ = (size_t)p.capacity;
  }

  int32_t klt_max_iteration_;
  double klt_epsilon_;
  int32_t klt_window_size_, klt_max_level_;
  double backtrack_distance_threshold_;
  int device_id_;
  detail::CtxPtr ctx_;
  int w_ = 0, h_ = 0;
  size_t capacity_ = 0;
  bool has_prev_ = false;
  vsf_track_stats stats_{};
};

// ref: FeatureTrackerLSSDLucasKanade.hpp:17-54 -- Parameter by its reference field names, the
// stateful 3-argument Track only (the class has no frame-pair form).
class FeatureTrackerLSSDLucasKanade : public FeatureTrackerBase {
 public:
  class Parameter {
   public:
    Parameter()  // FeatureTrackerLSSDLucasKanade.cpp:13-23
        : optical_flow_type_("frame_to_frame"),
          optical_flow_detection_grid_size_(32),
          optical_flow_max_recovered_dist_(0.09),
          optical_flow_pattern_(51),
          optical_flow_max_iterations_(5),
          optical_flow_levels_(3) {}
    std::string optical_flow_type_;
    int optical_flow_detection_grid_size_;
    double optical_flow_max_recovered_dist_;
    int optical_flow_pattern_;  // not read: the frame-to-frame class fixes Pattern24
    int optical_flow_max_iterations_;
    int optical_flow_levels_;
  };

  explicit FeatureTrackerLSSDLucasKanade(const Parameter& parameter, int device_id = 0)
      : parameter_(parameter), device_id_(device_id) {}

  void Track(FeaturePositionDatabase& feature_position, FeatureAgeDatabase& feature_age,
             const ImageView& current_frame) override {
    ensure(current_frame.cols, current_frame.rows, feature_position.size() + 64);
    std::vector<uint64_t> ids;
    std::vector<double> xy;
    std::vector<uint32_t> ag;
    detail::flatten(feature_position, feature_age, capacity_, ids, xy, ag);
    int32_t n = (int32_t)feature_position.size();
    const bool first = !has_prev_;
    detail::check(ctx_.get(),
                  vsf_track(ctx_.get(), current_frame.data, current_frame.cols, current_frame.rows,
                            current_frame.step, ids.data(), xy.data(), ag.data(), &n, &stats_),
                  "Track");
    has_prev_ = true;
    if (first) return;  // ref :57-68: pyramid and transforms stored, databases untouched
    detail::unflatten(ids, xy, ag, n, feature_position, feature_age);  // ref :105-113
  }

  const vsf_track_stats& last_stats() const { return stats_; }

 private:
  void ensure(int w, int h, size_t need) {
    if (ctx_ && w == w_ && h == h_ && need <= capacity_) return;
    if (ctx_ && w == w_ && h == h_ && has_prev_)
      throw std::runtime_error("vsf: feature table grew beyond the tracker capacity between frames");
    has_prev_ = false;
    vsf_params p;
    vsf_default_params(&p);
    p.tracker_kind = VSF_TRACKER_LSSD;
    p.lssd_levels = parameter_.optical_flow_levels_;
    p.lssd_max_iterations = parameter_.optical_flow_max_iterations_;
    p.lssd_max_recovered_dist = parameter_.optical_flow_max_recovered_dist_;
    p.width = w;
    p.height = h;
    p.device_id = device_id_;
    p.capacity = (int32_t)(need < 4096 ? 4096 : need * 2);
    vsf_ctx* c = nullptr;
    detail::check(nullptr, vsf_create(&p, &c), "vsf_create(lssd tracker)");
    ctx_.reset(c);
    w_ = w;
    h_ = h;
    capacity_ = (size_t)p.capacity;
  }

  Parameter parameter_;
  int device_id_;
  detail::CtxPtr ctx_;
  int w_ = 0, h_ = 0;
  size_t capacity_ = 0;
  bool has_prev_ = false;
  vsf_track_stats stats_{};
};

}  // namespace feature
}  // namespace vslam
