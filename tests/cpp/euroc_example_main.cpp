// The reference's example/vslam/example_feature_detection_and_tracking.cpp:12-80 on the C++ mirror
// classes: an EuRoC directory is read from disk (EurocKimeraDataProvider), the first frame is
// detected on, then every frame is tracked with the 4-argument Track and every 10th one tops the
// table up.  Visualisation is replaced by a text dump of the databases that the pytest wrapper
// compares with the CPU oracle run on the raw frames.
//   usage: euroc_example_main <dataset root> <calibration.json> <out.txt>
#include <cstdio>
#include <map>

#include "vsf_euroc.hpp"
#include "vsf_frontend.hpp"

int main(int argc, char** argv) {
  if (argc != 4) return 2;
  try {
    vslam::dataprovider::EurocKimeraDataProvider euroc_kimera_data_provider(argv[1], argv[2]);
    vslam::feature::FeatureDetectorShiTomasiBucketing shi_tomasi_detector(4, 4, 300, 2.0);
    vslam::feature::FeatureTrackerLucasKanade kl_tracker(30, 0.01, 15, 3, 1.0);
    vslam::FeatureAgeDatabase prev_feature_age;
    vslam::FeaturePositionDatabase prev_feature_position;
    vslam::frontend::KimeraFrontendInput prev_input;
    bool is_initialized = false;
    bool is_reach_the_last = false;
    uint64_t counter = 0;
    FILE* out = std::fopen(argv[3], "w");
    while (!is_reach_the_last) {
      auto input = euroc_kimera_data_provider.GetInput();
      if (input == std::nullopt) {
        is_reach_the_last = true;
        continue;
      }
      if (!is_initialized) {
        shi_tomasi_detector.UpdateDetection(prev_feature_position, prev_feature_age, input.value().frame_);
        is_initialized = true;
      } else {
        if (counter++ % 10 == 0)
          shi_tomasi_detector.UpdateDetection(prev_feature_position, prev_feature_age, input.value().frame_);
        kl_tracker.Track(prev_feature_position, prev_feature_age, prev_input.frame_, input.value().frame_);
      }
      std::map<uint64_t, vslam::Vec2_t> sorted(prev_feature_position.begin(), prev_feature_position.end());
      std::fprintf(out, "frame %.9f %zu\n", input.value().timestamp_, sorted.size());
      for (const auto& [id, p] : sorted)
        std::fprintf(out, "%llu %.17g %.17g %u\n", (unsigned long long)id, p[0], p[1], prev_feature_age[id]);
      prev_input = input.value();
    }
    std::fclose(out);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
