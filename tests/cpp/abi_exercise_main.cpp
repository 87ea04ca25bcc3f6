// Walks the C ABI (include/vsf.h) the way a C++ host would -- every kind of entry, error paths included -- for
// runs with the HOST code under AddressSanitizer (scratch/asan/run.sh builds library and driver that way; device
// code is compiled as always).  It checks return codes and a few invariants, not results: parity is the pytest
// suite's business.   usage: abi_exercise_main <frames.raw> W H N
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vsf.h"

#define REQ(cond)                                                              \
  do {                                                                         \
    if (!(cond)) {                                                             \
      std::fprintf(stderr, "%s:%d: %s failed\n", __FILE__, __LINE__, #cond);   \
      return 1;                                                                \
    }                                                                          \
  } while (0)

int main(int argc, char** argv) {
  if (argc != 5) return 2;
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), N = std::atoi(argv[4]);
  std::vector<uint8_t> buf((size_t)W * H * N);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(buf.data(), 1, buf.size(), f) != buf.size()) return 3;
  std::fclose(f);
  std::printf("%s\n", vsf_version());
  vsf_params p;
  // ---- refused parameters leave nothing behind
  for (int k = 0; k < 6; k++) {
    vsf_default_params(&p);
    p.width = W, p.height = H;
    if (k == 0) p.klt_win = 33;
    if (k == 1) p.grid_rows = 0;
    if (k == 2) p.streams = 0;
    if (k == 3) p.detector_kind = 7;
    if (k == 4) p.device_id = 99;
    if (k == 5) { p.detector_kind = VSF_DETECTOR_ANMS; p.anms_detection_min_distance = 40.0; }
    vsf_ctx* bad = nullptr;
    REQ(vsf_create(&p, &bad) != VSF_OK && bad == nullptr);
  }
  // ---- one stream, host-pointer API, both detectors and both trackers
  for (int det = 0; det < 2; det++)
    for (int trk = 0; trk < 2; trk++) {
      vsf_default_params(&p);
      p.width = W, p.height = H, p.max_features = 150, p.grid_rows = 3, p.grid_cols = 3, p.min_distance = 6.0;
      p.detector_kind = det ? VSF_DETECTOR_ANMS : VSF_DETECTOR_BUCKETING;
      p.tracker_kind = trk ? VSF_TRACKER_LSSD : VSF_TRACKER_LK;
      if (det) { p.anms_use_harris = 1; p.anms_detection_min_distance = 2.5; }
      vsf_ctx* c = nullptr;
      REQ(vsf_create(&p, &c) == VSF_OK);
      vsf_info info;
      REQ(vsf_get_info(c, &info) == VSF_OK && info.width == W && info.streams == 1);
      const int cap = info.capacity;
      std::vector<uint64_t> ids(cap);
      std::vector<double> xy(2 * (size_t)cap);
      std::vector<uint32_t> age(cap);
      int32_t n = 0;
      vsf_track_stats st;
      std::vector<uint8_t> mask((size_t)W * H, 255);
      for (int y = 0; y < H / 4; y++) std::memset(mask.data() + (size_t)y * W, 0, W);
      for (int k = 0; k < N; k++) {
        const uint8_t* fr = buf.data() + (size_t)k * W * H;
        if (k % 2 == 0) {
          REQ(vsf_step(c, fr, W, H, W, k == 2 ? mask.data() : nullptr, W, ids.data(), xy.data(), age.data(), &n, cap, &st) == VSF_OK);
        } else {
          REQ(vsf_track(c, fr, W, H, W, ids.data(), xy.data(), age.data(), &n, &st) == VSF_OK);
          REQ(vsf_detect(c, fr, W, H, W, nullptr, 0, ids.data(), xy.data(), age.data(), &n, cap) == VSF_OK);
        }
        REQ(n > 0 && n <= cap);
      }
      // arrays too short: refused before anything runs, the table untouched
      int32_t n2 = n;
      REQ(vsf_step(c, buf.data(), W, H, W, nullptr, 0, ids.data(), xy.data(), age.data(), &n2, n / 2, &st) == VSF_E_CAPACITY);
      REQ(n2 == n && std::strlen(vsf_last_error(c)) > 0);
      // a 3-channel frame, then a frame of another size (the context re-creates its buffers; the tracker starts over)
      std::vector<uint8_t> bgr((size_t)W * H * 3);
      for (size_t i = 0; i < (size_t)W * H; i++) bgr[3 * i] = bgr[3 * i + 1] = bgr[3 * i + 2] = buf[i];
      n2 = 0;
      REQ(vsf_detect_bgr(c, bgr.data(), W, H, 3 * W, 3, nullptr, 0, ids.data(), xy.data(), age.data(), &n2, cap) == VSF_OK);
      REQ(n2 > 0);
      const int w2 = W - 36, h2 = H - 20;
      std::vector<uint8_t> crop((size_t)w2 * h2);
      for (int y = 0; y < h2; y++) std::memcpy(crop.data() + (size_t)y * w2, buf.data() + (size_t)y * W, w2);
      REQ(vsf_reset_tracker(c) == VSF_OK);
      n2 = 0;
      REQ(vsf_detect(c, crop.data(), w2, h2, w2, nullptr, 0, ids.data(), xy.data(), age.data(), &n2, cap) == VSF_OK);
      REQ(vsf_get_info(c, &info) == VSF_OK && info.width == w2 && info.height == h2);
      uint64_t mi = 0;
      REQ(vsf_get_max_feature_index(c, 0, &mi) == VSF_OK && mi > 0);
      REQ(vsf_get_max_feature_index(c, 5, &mi) != VSF_OK);
      if (!det && !trk) {
        std::vector<float> lam((size_t)W * H);
        int32_t rect[4];
        REQ(vsf_debug_min_eig(c, crop.data(), w2, h2, w2, 0, lam.data(), rect) == VSF_OK);
        REQ(vsf_debug_min_eig(c, crop.data(), w2, h2, w2, 99, lam.data(), rect) != VSF_OK);
      }
      vsf_destroy(c);
    }
  // ---- several streams, device-pointer API
  {
    const int B = 3;
    vsf_default_params(&p);
    p.width = W, p.height = H, p.streams = B, p.max_features = 100, p.grid_rows = 2, p.grid_cols = 2, p.min_distance = 6.0;
    vsf_ctx* c = nullptr;
    REQ(vsf_create(&p, &c) == VSF_OK);
    vsf_info info;
    REQ(vsf_get_info(c, &info) == VSF_OK);
    const int cap = info.capacity;
    uint8_t* d_fr = nullptr;
    void* d_rec = nullptr;
    REQ(hipMalloc((void**)&d_fr, (size_t)B * W * H) == hipSuccess);
    REQ(hipMalloc(&d_rec, (size_t)B * cap * 24) == hipSuccess);
    REQ(vsf_profile_enable(c, 1 << (1 + VSF_K_LK)) == VSF_OK);
    for (int k = 0; k + 1 < N; k++) {
      for (int b = 0; b < B; b++)
        REQ(hipMemcpy(d_fr + (size_t)b * W * H, buf.data() + (size_t)((k + b) % N) * W * H, (size_t)W * H, hipMemcpyHostToDevice) == hipSuccess);
      REQ(vsf_batch_step(c, d_fr, (int64_t)W * H, W, nullptr) == VSF_OK);
    }
    REQ(vsf_batch_export_tracks(c, d_rec, nullptr) == VSF_OK);
    REQ(vsf_gather_tracks(c, nullptr, d_rec, nullptr) == VSF_OK);   // a world of one: export only
    std::vector<uint64_t> ids((size_t)B * cap);
    std::vector<double> xy(2 * (size_t)B * cap);
    std::vector<uint32_t> age((size_t)B * cap);
    std::vector<int32_t> n(B), status(B);
    std::vector<vsf_track_stats> st(B);
    REQ(vsf_batch_download(c, 0, B, ids.data(), xy.data(), age.data(), n.data(), st.data()) == VSF_OK);
    REQ(vsf_batch_download(c, 2, B, ids.data(), xy.data(), age.data(), n.data(), st.data()) != VSF_OK);  // past the end
    REQ(vsf_batch_upload(c, 1, 1, ids.data(), xy.data(), age.data(), n.data()) == VSF_OK);
    REQ(vsf_batch_status(c, status.data()) == VSF_OK && status[0] == 0);
    float ms[VSF_K_COUNT];
    int32_t cnt[VSF_K_COUNT];
    REQ(vsf_profile_enable(c, 0) == VSF_OK && vsf_profile_read(c, ms, cnt) == VSF_OK && cnt[VSF_K_LK] == N - 2 && cnt[VSF_K_SCORE] == 0);
    // the caller's current device survives the calls
    int dev = -1;
    REQ(hipGetDevice(&dev) == hipSuccess && dev == 0);
    vsf_camera_ds cam = {458.654, 457.296, 367.215, 248.375, -0.2, 0.59};
    std::vector<double> bearing(3 * 4), pts = {10.0, 20.0, 300.5, 100.25, 5000.0, 5000.0, 0.0, 0.0};
    std::vector<uint8_t> ok(4);
    REQ(vsf_unproject(c, &cam, pts.data(), 4, bearing.data(), ok.data()) == VSF_OK);
    // the entry that allocates nothing
    float *d_p = nullptr, *d_q = nullptr;
    uint8_t* d_s = nullptr;
    REQ(hipMalloc((void**)&d_p, 64 * 8) == hipSuccess && hipMalloc((void**)&d_q, 64 * 8) == hipSuccess && hipMalloc((void**)&d_s, 64) == hipSuccess);
    std::vector<float> hp(128);
    for (int i = 0; i < 64; i++) hp[2 * i] = 30.f + 5.f * (i % 8), hp[2 * i + 1] = 30.f + 4.f * (i / 8);
    REQ(hipMemcpy(d_p, hp.data(), 64 * 8, hipMemcpyHostToDevice) == hipSuccess);
    REQ(vsf_lk_points_noalloc(&p, d_fr, W, d_fr + (size_t)W * H, W, d_p, d_q, d_s, 64, nullptr) == VSF_OK);
    REQ(vsf_lk_points_noalloc(&p, d_fr, W, d_fr + (size_t)W * H, W, d_p, d_q, d_s, -1, nullptr) != VSF_OK);
    REQ(hipDeviceSynchronize() == hipSuccess);
    hipFree(d_p), hipFree(d_q), hipFree(d_s), hipFree(d_rec), hipFree(d_fr);
    vsf_destroy(c);
  }
  // ---- communicator bookkeeping without a second rank
  {
    vsf_comm* cm = nullptr;
    REQ(vsf_comm_create(nullptr, 0, 1, 0, &cm) == VSF_OK && cm != nullptr);
    int32_t r = -1, nr = -1;
    REQ(vsf_comm_rank(cm, &r, &nr) == VSF_OK && r == 0 && nr == 1);
    vsf_comm_destroy(cm);
    REQ(vsf_comm_create(nullptr, 0, 2, 0, &cm) != VSF_OK);   // more than one rank needs an id
  }
  vsf_destroy(nullptr);
  std::printf("abi exercise ok\n");
  return 0;
}
