// vsf_euroc.hpp -- C++ host mirror of vslam::dataprovider::EurocKimeraDataProvider
// (ref: src/vslam/io/EurocKimeraDataProvider.hpp:14-72, .cpp:21-228) over the C ABI of libvsf_io.so
// (include/vsf_io.h): same constructor arguments and calls, no OpenCV / yaml-cpp / nlohmann / csv.h.
// GetInput returns std::nullopt past the last line, as the reference does; other failures throw.
#pragma once

#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vsf_frontend.hpp"
#include "vsf_io.h"

namespace vslam {

struct Pose_t {  // pose_body_T_sensor_: rotation (row-major) and translation
  double rotation[9];
  double translation[3];
};

namespace frontend {

// ref: src/vslam/frontend/KimeraFrontend.hpp -- KimeraFrontendInput {timestamp_, frame_, mask_, camera_model_}
struct KimeraFrontendInput {
  double timestamp_ = 0;
  std::vector<uint8_t> frame_storage_;  // owns the pixels frame_ views (a cv::Mat owns its own)
  feature::ImageView frame_;
  feature::ImageView mask_;  // data == nullptr when the provider has no mask
  vsf_euroc_camera camera_model_{};

  KimeraFrontendInput() = default;
  KimeraFrontendInput(const KimeraFrontendInput& o) { *this = o; }
  KimeraFrontendInput(KimeraFrontendInput&& o) noexcept { *this = std::move(o); }
  KimeraFrontendInput& operator=(const KimeraFrontendInput& o) {
    if (this == &o) return *this;
    timestamp_ = o.timestamp_;
    frame_storage_ = o.frame_storage_;
    frame_ = feature::ImageView(frame_storage_.data(), o.frame_.cols, o.frame_.rows, o.frame_.step);
    mask_ = o.mask_;
    camera_model_ = o.camera_model_;
    return *this;
  }
  KimeraFrontendInput& operator=(KimeraFrontendInput&& o) noexcept {
    timestamp_ = o.timestamp_;
    frame_storage_ = std::move(o.frame_storage_);
    frame_ = feature::ImageView(frame_storage_.data(), o.frame_.cols, o.frame_.rows, o.frame_.step);
    mask_ = o.mask_;
    camera_model_ = o.camera_model_;
    return *this;
  }
};

}  // namespace frontend

namespace dataprovider {

class EurocKimeraDataProvider {
 public:
  EurocKimeraDataProvider(const std::string& path_to_dataset_root, const std::string& path_to_calibration_file,
                          const std::string& path_to_mask_image = "") {
    check(vsf_euroc_open(path_to_dataset_root.c_str(), path_to_calibration_file.c_str(), path_to_mask_image.c_str(),
                         &handle_));
    check(vsf_euroc_size(handle_, &size_));
    check(vsf_euroc_camera_model(handle_, &camera_));
    int32_t has = 0;
    check(vsf_euroc_mask(handle_, &has, &mask_w_, &mask_h_, nullptr, 0));
    if (has) {
      mask_.resize((size_t)mask_w_ * mask_h_);
      check(vsf_euroc_mask(handle_, &has, &mask_w_, &mask_h_, mask_.data(), mask_w_));
    }
  }
  ~EurocKimeraDataProvider() { vsf_euroc_close(handle_); }
  EurocKimeraDataProvider(const EurocKimeraDataProvider&) = delete;
  EurocKimeraDataProvider& operator=(const EurocKimeraDataProvider&) = delete;

  // Get data and increment the internal line index (.cpp:43-68)
  std::optional<frontend::KimeraFrontendInput> GetInput() {
    auto r = GetInput(last_index_);
    if (r) last_index_++;
    return r;
  }
  // Get data by the index specified (.cpp:70-91)
  std::optional<frontend::KimeraFrontendInput> GetInput(uint64_t index) {
    if (index >= size_) return std::nullopt;
    int32_t w = 0, h = 0;
    check(vsf_euroc_frame_size(handle_, &w, &h));
    frontend::KimeraFrontendInput in;
    in.frame_storage_.resize((size_t)w * h);
    check(vsf_euroc_frame(handle_, index, &in.timestamp_, in.frame_storage_.data(), w, w, h));
    in.frame_ = feature::ImageView(in.frame_storage_.data(), w, h);
    if (!mask_.empty()) in.mask_ = feature::ImageView(mask_.data(), mask_w_, mask_h_);
    in.camera_model_ = camera_;
    return in;
  }

  Pose_t GetSensorPose() {
    Pose_t p;
    check(vsf_euroc_sensor_pose(handle_, p.rotation, p.translation));
    return p;
  }
  uint64_t size() const { return size_; }

 private:
  static void check(int rc) {
    if (rc != VSF_IO_OK) throw std::runtime_error(std::string("vsf_io: ") + vsf_io_last_error());
  }
  vsf_euroc* handle_ = nullptr;
  uint64_t size_ = 0, last_index_ = 0;
  vsf_euroc_camera camera_{};
  std::vector<uint8_t> mask_;
  int32_t mask_w_ = 0, mask_h_ = 0;
};

}  // namespace dataprovider
}  // namespace vslam
