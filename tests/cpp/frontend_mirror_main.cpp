// Drives the C++ mirror classes the way example/vslam/example_feature_detection_and_tracking.cpp:41-56
// drives the reference: detect on the first frame, then Track (4-argument form) and a periodic
// UpdateDetection.  Reads raw 8-bit frames, writes the databases after every frame as text; the
// pytest wrapper (tests/test_gpu_cpp_mirror.py) compares them with the CPU oracle.
//   usage: frontend_mirror_main frames.raw W H N out.txt rows cols maxN minDist iter eps win lvl back
// rows == 0 selects FeatureDetectorANMS (Parameter defaults with max_feature_number_ = maxN,
// min_feature_distance_ = minDist, detection_feature_number_ = 2 * maxN) instead of the bucketing detector;
// with cols == 1 also detection_use_harris_corner_, detection_k_ = 0.06, detection_min_feature_distance_ = 3.5.
// iter == 0 selects FeatureTrackerLSSDLucasKanade (Parameter defaults) driven the way
// KimeraFrontend.cpp drives a tracker: the 3-argument Track on every frame, then the detector.
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <vector>

#include "vsf_frontend.hpp"

int main(int argc, char** argv) {
  if (argc != 15) {
    std::fprintf(stderr, "bad arguments\n");
    return 2;
  }
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), N = std::atoi(argv[4]);
  std::vector<uint8_t> buf((size_t)W * H * N);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(buf.data(), 1, buf.size(), f) != buf.size()) return 3;
  std::fclose(f);
  try {
    std::unique_ptr<vslam::feature::FeatureDetectorBase> detector_ptr;
    std::function<uint64_t()> max_index;
    if (std::atoi(argv[6]) == 0) {
      vslam::feature::FeatureDetectorANMS::Parameter prm;
      prm.max_feature_number_ = std::atoi(argv[8]);
      prm.min_feature_distance_ = std::atof(argv[9]);
      prm.detection_feature_number_ = prm.max_feature_number_ * 2;
      if (std::atoi(argv[7]) == 1) {
        prm.detection_use_harris_corner_ = true;
        prm.detection_k_ = 0.06;
        prm.detection_min_feature_distance_ = 3.5;
      }
      auto* d = new vslam::feature::FeatureDetectorANMS(prm);
      detector_ptr.reset(d);
      max_index = [d]() { return d->max_feature_index(); };
    } else {
      auto* d = new vslam::feature::FeatureDetectorShiTomasiBucketing(std::atoi(argv[6]), std::atoi(argv[7]),
                                                                      std::atoi(argv[8]), std::atof(argv[9]));
      detector_ptr.reset(d);
      max_index = [d]() { return d->max_feature_index(); };
    }
    vslam::feature::FeatureDetectorBase& shi_tomasi_detector = *detector_ptr;
    const bool use_lssd = std::atoi(argv[10]) == 0;
    vslam::feature::FeatureTrackerLucasKanade kl_tracker(use_lssd ? 30 : std::atoi(argv[10]), std::atof(argv[11]),
                                                         std::atoi(argv[12]), std::atoi(argv[13]), std::atof(argv[14]));
    vslam::feature::FeatureTrackerLSSDLucasKanade lssd_tracker{vslam::feature::FeatureTrackerLSSDLucasKanade::Parameter()};
    vslam::FeaturePositionDatabase prev_feature_position;
    vslam::FeatureAgeDatabase prev_feature_age;
    FILE* out = std::fopen(argv[5], "w");
    for (int k = 0; k < N; k++) {
      vslam::feature::ImageView frame(buf.data() + (size_t)k * W * H, W, H);
      if (use_lssd) {
        lssd_tracker.Track(prev_feature_position, prev_feature_age, frame);
        if (k % 2 == 0) shi_tomasi_detector.UpdateDetection(prev_feature_position, prev_feature_age, frame);
      } else if (k == 0) {
        shi_tomasi_detector.UpdateDetection(prev_feature_position, prev_feature_age, frame);
      } else {
        if (k % 2 == 0) shi_tomasi_detector.UpdateDetection(prev_feature_position, prev_feature_age, frame);
        vslam::feature::ImageView prev(buf.data() + (size_t)(k - 1) * W * H, W, H);
        kl_tracker.Track(prev_feature_position, prev_feature_age, prev, frame);
      }
      std::map<uint64_t, vslam::Vec2_t> sorted(prev_feature_position.begin(), prev_feature_position.end());
      std::fprintf(out, "frame %d %zu %llu\n", k, sorted.size(), (unsigned long long)max_index());
      for (const auto& [id, p] : sorted)
        std::fprintf(out, "%llu %.17g %.17g %u\n", (unsigned long long)id, p[0], p[1], prev_feature_age[id]);
    }
    std::fclose(out);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
