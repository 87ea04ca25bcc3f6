/*
 * vsf_oracle.h -- CPU ORACLE for the vision-frontend hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm the reference (eryeden/vi-slam) runs on the CPU
 * for image pyramid -> Shi-Tomasi grid detection -> pyramidal LK tracking with forward-backward
 * check.  It is used ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
 * the checker / reported baseline.  The product path (vi-slam_amd/csrc) never links or calls it.
 *
 * PARITY UNPINNED.  The reference does none of this arithmetic itself: its detector and tracker
 * call OpenCV 3.4.9 (docs/setup.md:27; call sites FeatureDetectorShiTomasiBucketing.cpp:158-163,
 * FeatureTrackerLucasKanade.cpp:216-224).  OpenCV is a third-party dependency that is not vendored
 * under /root/reference and is not installed here, and the reference's own tests are placeholders
 * (test/sample/test_sample1.cpp:38-45), so no golden vector of the reference pins these results.
 * The OpenCV parts below restate the published OpenCV 3.4 algorithms (imgproc: pyrDown, Sobel,
 * boxFilter, cornerMinEigenVal, goodFeaturesToTrack; video: buildOpticalFlowPyramid,
 * calcOpticalFlowPyrLK scalar path) per SURVEY.md Appendix A; the wrapper parts follow the
 * reference's own sources line by line (citations at each function in vsf_oracle.c).
 * What pins it instead: analytic known-answer tests (tests/test_oracle_known_answers.py) and
 * committed golden vectors generated from this file (tests/golden/, tests/golden/make_golden.py).
 */
#ifndef VSF_ORACLE_H
#define VSF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16

/* LK accumulation modes for sum(Ix*Ix) etc. (lines marked with a warning sign):
 *   ORC_ACC_FLOAT_SEQ : float32 accumulator, raster order, (float)(int product) per pixel.  This is
 *                       OpenCV 3.4's scalar (non-SIMD) path, the "restated reference".
 *   ORC_ACC_EXACT     : exact integer sum, converted to float32 once (round-to-nearest-even).
 *                       Order independent, so a parallel implementation can match it bit for bit.
 * tests/test_oracle_lk_modes.py measures the distance between the two modes (<= 0.01 px, same
 * status) so that a bit-exact match against ORC_ACC_EXACT carries over to the reference order
 * within the tolerance BASELINE.json states. */
#define ORC_ACC_FLOAT_SEQ 0
#define ORC_ACC_EXACT 1

typedef struct {
  int w, h;
  uint8_t* data; /* dense, stride == w */
} orc_level;

typedef struct {
  int max_level; /* effective maxLevel (levels 0..max_level are valid) */
  orc_level lv[ORC_MAX_LEVELS];
} orc_pyramid;

/* ---- OpenCV imgproc / video restatements ---- */
void orc_pyr_down_u8(const uint8_t* src, int w, int h, int stride, uint8_t* dst);
int orc_build_pyramid(const uint8_t* img, int w, int h, int stride, int win, int max_level,
                      orc_pyramid* out);
void orc_free_pyramid(orc_pyramid* p);
void orc_scharr_s16(const uint8_t* src, int w, int h, int stride, int16_t* dxy);

int orc_calc_optical_flow_pyr_lk(const uint8_t* prev, int prev_stride, const uint8_t* next,
                                 int next_stride, int w, int h, const float* prev_pts,
                                 float* next_pts, uint8_t* status, float* err, int n, int win,
                                 int max_level, int max_count, double eps, double min_eig_thr,
                                 int acc_mode);

void orc_corner_min_eigen_val(const uint8_t* img, int img_w, int img_h, int stride, int rx, int ry,
                              int rw, int rh, int block_size, float* eig);

/* cv::cornerHarris on the ROI (blockSize, ksize 3, k): same covariance sums as the min-eigenvalue map, response
 * a*c - b*b - k*(a + c)^2 as OpenCV 3.4's calcHarris forms it (float32 in groups of four pixels, the row's last
 * width % 4 pixels with k in double).  Recalled; PARITY UNPINNED like the rest of the OpenCV arithmetic. */
void orc_corner_harris(const uint8_t* img, int img_w, int img_h, int stride, int rx, int ry, int rw, int rh,
                       int block_size, double k, float* dst);
/* goodFeaturesToTrack with useHarrisDetector / k (what cv::GFTTDetector passes on, FeatureDetectorANMS.cpp:41-47) */
int orc_good_features_to_track_ex(const uint8_t* img, int img_w, int img_h, int stride, int rx, int ry,
                                  int rw, int rh, int max_corners, double quality, double min_dist,
                                  const uint8_t* mask, int mask_stride, int block_size, int use_harris,
                                  double harris_k, float* out_xy, int cap);
int orc_good_features_to_track(const uint8_t* img, int img_w, int img_h, int stride, int rx, int ry,
                               int rw, int rh, int max_corners, double quality, double min_dist,
                               const uint8_t* mask, int mask_stride, int block_size, float* out_xy,
                               int cap);

/* ---- reference wrappers ---- */
typedef struct {
  int32_t division_number_row;
  int32_t division_number_col;
  int32_t max_feature_number;
  double min_feature_distance;
  uint64_t max_feature_index;
} orc_detector;

void orc_detector_init(orc_detector* d, int32_t rows, int32_t cols, int32_t max_features,
                       double min_distance);
/* cv::cvtColor(src, dst, CV_BGR2GRAY) for 8-bit input of cn = 3 or 4 interleaved channels, B first: what the
 * detectors do to a multi-channel frame (FeatureDetectorShiTomasiBucketing.cpp:105-110, FeatureDetectorANMS.cpp:
 * 98-99).  OpenCV 3.4 imgproc/color_rgb.cpp RGB2Gray<uchar>: fixed-point weights B2Y = 1868, G2Y = 9617,
 * R2Y = 4899, yuv_shift = 14, rounding constant 1 << 13 (recalled; parity unpinned like the rest of the OpenCV
 * arithmetic -- an IPP build may round differently). */
void orc_bgr2gray(const uint8_t* src, int w, int h, int stride, int cn, uint8_t* dst, int dst_stride);
int orc_generate_grid(int w, int h, int cols, int rows, int* rects /* 4*cells: x,y,w,h */,
                      float* origin /* 2*cells Rect2f x,y; may be NULL */);
int orc_detector_update(orc_detector* d, const uint8_t* img, int w, int h, int stride,
                        const uint8_t* mask, int mask_stride, uint64_t* ids, double* xy,
                        uint32_t* age, int* n_inout, int capacity);

typedef struct {
  int32_t klt_max_iteration;
  double klt_epsilon;
  int32_t klt_window_size;
  int32_t klt_max_level;
  double backtrack_distance_threshold;
  int acc_mode;
  uint8_t* old_image; /* owned copy; NULL == cv::Mat::empty() */
  int old_w, old_h;
} orc_tracker;

typedef struct {
  int32_t input, forward, backward, verified;
} orc_track_stats;

void orc_tracker_init(orc_tracker* t, int32_t max_iter, double eps, int32_t win, int32_t max_level,
                      double backtrack_thr, int acc_mode);
void orc_tracker_free(orc_tracker* t);
int orc_tracker_track(orc_tracker* t, const uint8_t* cur, int w, int h, int stride, uint64_t* ids,
                      double* xy, uint32_t* age, int* n_inout, orc_track_stats* stats);
int orc_tracker_track_pair(const orc_tracker* t, const uint8_t* prev, int prev_stride,
                           const uint8_t* cur, int cur_stride, int w, int h, uint64_t* ids,
                           double* xy, uint32_t* age, int* n_inout, orc_track_stats* stats);

/* ---- ANMS detector (ref: src/vslam/feature/FeatureDetectorANMS.cpp) ---- */
/* cv::circle(img, center, radius, color, FILLED) for 8-bit single-channel images, thickness < 0,
 * LINE_8, shift 0 -- OpenCV 3.4 drawing.cpp Circle() (midpoint algorithm, horizontal spans). */
void orc_circle_filled(uint8_t* img, int w, int h, int stride, int cx, int cy, int radius,
                       uint8_t color);
/* Half-width of that disc on the rows cy +- d, d = 0..radius: the disc is { |dy| <= radius,
 * |dx| <= halfw[|dy|] }.  Returns 0. */
int orc_circle_halfwidths(int radius, int* halfw /* radius+1 */);
/* cv::sortIdx(std::vector<int>, idx, CV_SORT_DESCENDING): idx = iota; std::sort(idx, LessThanIdx);
 * reverse.  std::sort restated from libstdc++ (introsort: median-of-3 to first, unguarded partition,
 * heapsort fallback, final insertion sort); NOT stable, so equal keys come out permuted (Q9). */
void orc_sort_idx_descending(const int* keys, int n, int* idx);
/* anms::Ssc (NonMaximumSuppression/anms/anms.cpp:344-438) on keypoints given in sweep order.
 * result receives indices into kxy; returns their number, or -1 when num_ret <= 0 (the reference
 * divides by zero there: undefined). */
int orc_ssc(const float* kxy, int n, int num_ret, float tolerance, int cols, int rows, int* result);

typedef struct {
  int32_t max_feature_number;             /* Parameter::max_feature_number_            :17 */
  double min_feature_distance;            /* min_feature_distance_                     :18 */
  int32_t detection_feature_number;       /* detection_feature_number_ = 2 * max       :23 */
  double detection_min_feature_distance;  /* 1.0                                       :24 */
  double detection_quality_level;         /* 0.001                                     :25 */
  int32_t detection_block_size;           /* 4                                         :26 */
  int32_t detection_ignore_edge;          /* 20                                        :29 */
  double non_max_suppression_tolerance;   /* 0.1                                       :34 */
  uint64_t max_feature_index;             /* member, starts at 0                       :39 */
  int32_t detection_use_harris_corner;    /* false                                     :27 */
  double detection_k;                     /* 0.04                                      :28 */
} orc_anms_detector;

void orc_anms_detector_init(orc_anms_detector* d, int32_t max_features, double min_distance);
/* The mask GFTT sees (FeatureDetectorANMS.cpp:107-132): user mask (or ones), discs zeroed at the
 * existing in-image features, AND-ed bitwise with the edge rectangle of ones. */
void orc_anms_build_mask(const orc_anms_detector* d, int w, int h, const uint8_t* mask,
                         int mask_stride, const double* xy, int n, uint8_t* flag /* w*h */);
/* FeatureDetectorANMS::Detect (:69-212).  Returns 0, -1 when capacity is exhausted.  When the
 * database already holds max_feature_number entries and corners are detected the reference calls
 * Ssc with numRetPoints = 0 (undefined behaviour); this restatement adds nothing then and
 * returns 1. */
int orc_anms_detector_update(orc_anms_detector* d, const uint8_t* img, int w, int h, int stride,
                             const uint8_t* mask, int mask_stride, uint64_t* ids, double* xy,
                             uint32_t* age, int* n_inout, int capacity);

/* ---- LSSD SE(2) patch tracker (ref: src/vslam/feature/FeatureTrackerLSSDLucasKanade.cpp,
 *      LSSDOpticalFlow/frame_to_frame_optical_flow_single_thread.cpp, patch.h, patterns.h) ----
 * In-repo parts (TrackPoints / TrackPoint / TrackPointAtLevel, OpticalFlowPatch, Pattern24) are followed
 * line by line.  The parts that live in the absent basalt-headers / Sophus / Eigen submodules are
 * restated from their published sources as recalled: Image::InBounds / interp /
 * interpGrad, ManagedImagePyr::setFromImage (u16 mip levels, floor-half sizes, [1 4 6 4 1]^2 with
 * one rounding), Sophus::SE2f::exp, Eigen's 3x3 LDLT solve and fixed-size products.  PARITY UNPINNED.
 * Float summation order:
 *   ORC_LSSD_SEQ  : sums over the pattern in index order, libm sinf/cosf ("restated reference");
 *   ORC_LSSD_TREE : sums as a 32-slot butterfly tree (slot i = pattern point i, zeros elsewhere;
 *                   steps i^1, i^2, i^4, i^8, i^16) and a polynomial sin/cos, which a 32-lane
 *                   half-wavefront reproduces bit for bit. */
#define ORC_LSSD_SEQ 0
#define ORC_LSSD_TREE 1
#define ORC_LSSD_PATTERN 24

typedef struct {
  int w, h;
  uint16_t* data; /* dense, stride == w */
} orc_level16;
typedef struct {
  int levels; /* optical_flow_levels: images 0..levels */
  orc_level16 lv[ORC_MAX_LEVELS];
} orc_pyramid16;

/* u8 frame -> u16 (<< 8, FeatureTrackerLSSDLucasKanade.cpp:42-51) -> ManagedImagePyr::setFromImage */
int orc_lssd_build_pyramid(const uint8_t* img, int w, int h, int stride, int levels, orc_pyramid16* out);
void orc_lssd_free_pyramid(orc_pyramid16* p);
/* FrameToFrameOpticalFlow::TrackPoint (forward only): returns validity, xy_out = translation */
int orc_lssd_track_point(const orc_pyramid16* old_pyr, const orc_pyramid16* pyr, const float* xy_in,
                         float* xy_out, int levels, int max_iterations, int mode);

typedef struct {
  double optical_flow_max_recovered_dist; /* 0.09, compared with the SQUARED distance (:67) */
  int32_t optical_flow_max_iterations;    /* 5 */
  int32_t optical_flow_levels;            /* 3 */
  int mode;
  orc_pyramid16 old_pyr;
  int has_old;
  /* transform_ (FeatureTrackerLSSDLucasKanade.hpp:51): id -> float translation, kept between calls */
  uint64_t* map_ids;
  float* map_xy;
  int map_n, map_cap;
} orc_lssd_tracker;

void orc_lssd_tracker_init(orc_lssd_tracker* t, double max_recovered_dist, int32_t levels,
                           int32_t max_iterations, int mode);
void orc_lssd_tracker_free(orc_lssd_tracker* t);
/* FeatureTrackerLSSDLucasKanade::Track (:35-119): first call stores the pyramid only. */
int orc_lssd_tracker_track(orc_lssd_tracker* t, const uint8_t* cur, int w, int h, int stride,
                           uint64_t* ids, double* xy, uint32_t* age, int* n_inout);

/* ---- frontend step around the path (ref: src/vslam/frontend/KimeraFrontend.cpp:247-304) ---- */
typedef struct {
  double fx, fy, cx, cy, xi, alpha; /* DoubleSphereCameraModel (src/vslam/data/Camera.hpp:134) */
} orc_camera_ds;
/* basalt::DoubleSphereCamera<double>::unproject (basalt-headers, absent submodule: restated from
 * Usenko et al., "The Double Sphere Camera Model", 3DV 2018, eq. 47-50, in basalt's operation
 * order).  Returns 0 when the pixel is outside the model's domain; out is then left untouched. */
int orc_ds_unproject(const orc_camera_ds* cam, double u, double v, double* out3);
/* KimeraFrontend::ProcessFrame between Track and RemoveOutlier: bearings of every tracked feature
 * ({0,0,1} when Unproject throws, :249-256), the number of features of age >= counting_feature_age
 * (:266-269), the keyframe decision (:272-293) and, for a keyframe, the age increment (:302-304).
 * Returns 1 for a keyframe, 0 otherwise. */
int orc_frontend_post_track(const orc_camera_ds* cam, const double* xy, uint32_t* age, int n,
                            int32_t counting_feature_age, uint32_t low_keyframe_feature_number,
                            double minimum_keyframe_interval, double keyframe_interval,
                            double dt_since_keyframe, double* bearings /* 3n */,
                            uint32_t* aged_out);

/* One frontend step on `streams` independent streams, used as the timed CPU baseline:
 * track (3-arg) then detect on the same frame.  Parallel over streams with OpenMP when built
 * with -fopenmp (threads = omp default), serial otherwise. */
int orc_step_streams(orc_tracker* trackers, orc_detector* detectors, int streams,
                     const uint8_t* frames, long frame_stride_bytes, int w, int h, int stride,
                     uint64_t* ids, double* xy, uint32_t* age, int* n, int capacity);
/* Q5 nearest-neighbour rejection against the pre-existing features (FeatureDetectorShiTomasiBucketing.cpp:168-185) */
int orc_nn_accept(const double* kd_points, int n0, double cx, double cy, double min_feature_distance);
void orc_set_threads(int threads); /* <= 0: the OpenMP default */
int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
