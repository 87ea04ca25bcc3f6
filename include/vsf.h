/*
 * vsf.h -- C ABI of the MI355X-native vision-frontend hot path (libvsf_hip.so).
 *
 * Drop-in boundary for the reference's two C++ interfaces (eryeden/vi-slam, paths relative to the
 * reference root):
 *   vslam::feature::FeatureDetectorBase::UpdateDetection   src/vslam/feature/FeatureDetectorBase.hpp:17-20
 *   vslam::feature::FeatureTrackerBase::Track              src/vslam/feature/FeatureTrackerBase.hpp:21-23
 * as implemented by FeatureDetectorShiTomasiBucketing (FeatureDetectorShiTomasiBucketing.cpp:38-239)
 * and FeatureTrackerLucasKanade (FeatureTrackerLucasKanade.cpp:8-240).
 *
 * The reference's in/out containers FeaturePositionDatabase / FeatureAgeDatabase
 * (src/vslam/type_defines.hpp:95-99: unordered_map<uint64, Vector2d>, unordered_map<uint64,uint32>
 * with one key set) cross this boundary as SoA arrays: ids[n] (uint64), xy[2n] (double, x then y),
 * age[n] (uint32).  Order is not significant (the reference's maps are unordered); this library
 * keeps surviving entries in their input order and appends new ones in id order.
 *
 * Plain pointers and sizes only; no C++/torch types.  Every call returns VSF_OK (0) or a negative
 * VSF_E_* code; nothing throws.  A context is not thread-safe (the reference objects are not
 * either: old_image_, max_feature_index_); use one context per thread.  All kernels are hand
 * written HIP for gfx950; there is no CPU fallback: without a GPU vsf_create fails.
 */
#ifndef VSF_H
#define VSF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSF_OK 0
#define VSF_E_INVALID (-1)   /* bad argument / parameter out of the supported range */
#define VSF_E_HIP (-2)       /* a HIP runtime call failed (see vsf_last_error) */
#define VSF_E_CAPACITY (-3)  /* feature table capacity exhausted */
#define VSF_E_GEOMETRY (-4)  /* a grid cell leaves the image (cv::Mat::operator()(Rect) would throw in the reference);
                                a frame whose size differs from the stored previous frame's (cv::calcOpticalFlowPyrLK
                                would throw); a frame size a batched context was not created for */
#define VSF_E_NOMEM (-5)

typedef struct vsf_ctx vsf_ctx;

/* Parameters = the constructor arguments of the two reference classes plus the OpenCV defaults
 * the reference relies on. */
typedef struct vsf_params {
  /* FeatureDetectorShiTomasiBucketing ctor (FeatureDetectorShiTomasiBucketing.hpp:32-35) */
  int32_t grid_rows;      /* division_number_row (rows x cols <= 65535 cells) */
  int32_t grid_cols;      /* division_number_col */
  int32_t max_features;   /* max_feature_number  */
  double min_distance;    /* min_feature_distance */
  double quality_level;   /* 0.1, hard-coded at FeatureDetectorShiTomasiBucketing.cpp:161 */
  int32_t block_size;     /* 3, cv::goodFeaturesToTrack default */
  /* FeatureTrackerLucasKanade ctor (FeatureTrackerLucasKanade.hpp:19-23) */
  int32_t klt_max_iter;   /* klt_max_iteration */
  double klt_eps;         /* klt_epsilon */
  int32_t klt_win;        /* klt_window_size (square), 3..31 */
  int32_t klt_max_level;  /* klt_max_level, 0..7 */
  double min_eig_thr;     /* 1e-4, cv::calcOpticalFlowPyrLK default */
  double backtrack_thr;   /* backtrack_distance_threshold */
  /* geometry and sizing of this context */
  int32_t width, height;  /* frame size (8-bit, single channel); each <= 65535 (ANMS detector: <= 32767) */
  int32_t streams;        /* independent streams held by this context (>= 1) */
  int32_t capacity;       /* feature-table capacity per stream; 0 = max_features + 64 */
  int32_t device_id;      /* HIP device ordinal */
  /* Which FeatureDetectorBase implementation vsf_detect / vsf_batch_detect / vsf_batch_step run:
   * VSF_DETECTOR_BUCKETING = FeatureDetectorShiTomasiBucketing (fields above),
   * VSF_DETECTOR_ANMS      = FeatureDetectorANMS (FeatureDetectorANMS.cpp:13-212): max_features and
   * min_distance above are its Parameter::max_feature_number_ / min_feature_distance_ (:17-18; the disc
   * radius is int(min_distance) <= 31); the fields below are the rest of its Parameter (:23-34), continued
   * by anms_detection_min_distance / anms_use_harris / anms_harris_k at the end of the struct (:24, :27-28). */
  int32_t detector_kind;
  int32_t anms_detection_features; /* detection_feature_number_; 0 = 2 * max_features (:23); <= 8192 */
  double anms_quality_level;       /* detection_quality_level_ 0.001 (:25) */
  int32_t anms_block_size;         /* detection_block_size_ 4 (:26); 3 or 4 */
  int32_t anms_ignore_edge;        /* detection_ignore_edge_ 20 (:29) */
  double anms_tolerance;           /* non_max_suppression_tolerance_ 0.1 (:34) */
  /* Which FeatureTrackerBase implementation vsf_track / vsf_batch_track / vsf_batch_step run:
   * VSF_TRACKER_LK = FeatureTrackerLucasKanade (klt_* above), VSF_TRACKER_LSSD =
   * FeatureTrackerLSSDLucasKanade (FeatureTrackerLSSDLucasKanade.cpp:13-119; basalt frame-to-frame SE(2)
   * patch tracker, Pattern24, float): fields below are its Parameter (:17-22). */
  int32_t tracker_kind;
  int32_t lssd_levels;             /* optical_flow_levels_ 3 (images 0..levels); <= 7 */
  int32_t lssd_max_iterations;     /* optical_flow_max_iterations_ 5 */
  double lssd_max_recovered_dist;  /* optical_flow_max_recovered_dist_ 0.09, compared with the SQUARED distance */
  /* LK template cache (an optimisation of this library, no counterpart in the reference; results are bit-identical
   * with and without it): 2 x streams x capacity x levels x 2576 bytes of device memory (6.3 GB at 1024 streams of
   * 752x480 / 300 features, 6 MB for one stream), allocated by vsf_create after everything else.  0 = allocate it
   * when it stays below 16 GiB; > 0 = that many bytes at most; < 0 = never.  When a later allocation of the context
   * (the host-pointer API's staging, the frame-pair buffers) does not fit, the cache is given back first.
   * vsf_get_info reports what the context holds. */
  int64_t lk_cache_max_bytes;
  /* The rest of FeatureDetectorANMS::Parameter, as passed to cv::GFTTDetector::create (FeatureDetectorANMS.cpp:41-47).
   * anms_detection_min_distance: detection_min_feature_distance_ 1.0 (:24), cv's minDistance; <= 1 (0 included)
   * never rejects a corner, above 1 the sorted corners are accepted greedily (<= 31).
   * anms_use_harris / anms_harris_k: detection_use_harris_corner_ false (:27) / detection_k_ 0.04 (:28): the
   * response map is cv::cornerHarris instead of cv::cornerMinEigenVal. */
  double anms_detection_min_distance;
  int32_t anms_use_harris;
  double anms_harris_k;
} vsf_params;

#define VSF_TRACKER_LK 0
#define VSF_TRACKER_LSSD 1

#define VSF_DETECTOR_BUCKETING 0
#define VSF_DETECTOR_ANMS 1

/* The three counts FeatureTrackerLucasKanade::Track logs (FeatureTrackerLucasKanade.cpp:87-104). */
typedef struct vsf_track_stats {
  int32_t input, forward, backward, verified;
} vsf_track_stats;

/* What a context holds (vsf_get_info). */
typedef struct vsf_info {
  int64_t lk_cache_bytes; /* device bytes of the LK template cache; 0 = not held (disabled, too large, given back) */
  int32_t width, height;  /* current frame size (the host-pointer API may have re-created the context's buffers) */
  int32_t streams, capacity;
  int32_t levels;         /* effective pyramid levels (maxLevel + 1 after cv's window rule) */
  int32_t reserved;
} vsf_info;

/* Fills *p with the reference's example parameters (example_feature_detection_and_tracking.cpp:19-23
 * grid 4x4 / 300 features, OpenCV's 21x21 window, 4 levels, 30 iterations, eps 0.01) at 752x480. */
void vsf_default_params(vsf_params* p);

int vsf_create(const vsf_params* p, vsf_ctx** out);
void vsf_destroy(vsf_ctx* ctx);
const char* vsf_last_error(const vsf_ctx* ctx);
const char* vsf_version(void);
int vsf_get_info(const vsf_ctx* ctx, vsf_info* out);

/* ---- single-stream host-pointer API: the drop-in for the reference interfaces (uses stream 0) ---- */

/* FeatureTrackerLucasKanade::Track, 3-argument stateful form (FeatureTrackerLucasKanade.cpp:25-107).
 * The first call only stores the frame and leaves the table untouched (:29-32).  The frame is
 * copied to the device during the call (the reference keeps a shallow cv::Mat reference instead).
 * stats may be NULL. */
int vsf_track(vsf_ctx* ctx, const uint8_t* img, int32_t w, int32_t h, int32_t stride, uint64_t* ids,
              double* xy, uint32_t* age, int32_t* n_inout, vsf_track_stats* stats);

/* 4-argument stateless form (FeatureTrackerLucasKanade.cpp:109-185). */
int vsf_track_pair(vsf_ctx* ctx, const uint8_t* prev, int32_t prev_stride, const uint8_t* cur,
                   int32_t cur_stride, int32_t w, int32_t h, uint64_t* ids, double* xy,
                   uint32_t* age, int32_t* n_inout, vsf_track_stats* stats);

/* FeatureDetectorShiTomasiBucketing::UpdateDetection (FeatureDetectorShiTomasiBucketing.cpp:50-239)
 * including its quirks (column-major cells, early break, id counter), without
 * the cv::imshow side effect.  mask may be NULL (no mask); non-zero = allowed.  `capacity` is the
 * length of the caller's arrays: it must hold what the call can produce, min(context capacity, n + rows * cols *
 * (max_features / (rows * cols))) entries (n + anms_detection_features for the ANMS detector); a shorter array is
 * refused with VSF_E_CAPACITY BEFORE anything runs, so the call can be repeated (vsf_step likewise). */
int vsf_detect(vsf_ctx* ctx, const uint8_t* img, int32_t w, int32_t h, int32_t stride,
               const uint8_t* mask, int32_t mask_stride, uint64_t* ids, double* xy, uint32_t* age,
               int32_t* n_inout, int32_t capacity);

/* UpdateDetection on a multi-channel frame: the reference converts any frame with channels() != 1 by
 * cv::cvtColor(CV_BGR2GRAY) first (FeatureDetectorShiTomasiBucketing.cpp:105-110, FeatureDetectorANMS.cpp:98-99).
 * img holds `channels` = 3 (BGR) or 4 (BGRA) interleaved bytes per pixel, rows `stride` bytes apart; the conversion
 * (cv's fixed-point weights, (1868 B + 9617 G + 4899 R + 2^13) >> 14) runs on the device.  Otherwise vsf_detect. */
int vsf_detect_bgr(vsf_ctx* ctx, const uint8_t* img, int32_t w, int32_t h, int32_t stride, int32_t channels,
                   const uint8_t* mask, int32_t mask_stride, uint64_t* ids, double* xy, uint32_t* age,
                   int32_t* n_inout, int32_t capacity);

/* Frame size per call.  The reference interfaces take any cv::Mat per call; a context is created for one size
 * (vsf_params.width / height) and, when it holds ONE stream, follows the caller: vsf_detect, vsf_detect_bgr,
 * vsf_track_pair, and vsf_track / vsf_step while no previous frame is stored, re-create the context's device
 * buffers for a frame of another size (about a millisecond; max_feature_index_ is kept, the previous frame and the
 * LK template cache are not).  vsf_track / vsf_step with a stored previous frame of another size return
 * VSF_E_GEOMETRY, where cv::calcOpticalFlowPyrLK throws; so does a grid that does not fit the new size, and any
 * size change on a context of several streams. */

/* Track, then UpdateDetection on the SAME frame -- the loop body of the reference's example
 * (example/vslam/example_feature_detection_and_tracking.cpp:43-55) -- as one call: the frame crosses PCIe once, the
 * table once in each direction, one synchronisation.  Identical results to vsf_track followed by vsf_detect.
 * (KimeraFrontend runs RemoveOutlier between the two on keyframes, KimeraFrontend.cpp:306-335: use the two calls
 * there.)  mask may be NULL; stats may be NULL. */
int vsf_step(vsf_ctx* ctx, const uint8_t* img, int32_t w, int32_t h, int32_t stride, const uint8_t* mask,
             int32_t mask_stride, uint64_t* ids, double* xy, uint32_t* age, int32_t* n_inout, int32_t capacity,
             vsf_track_stats* stats);

/* max_feature_index_ of the detector (FeatureDetectorShiTomasiBucketing.hpp:53). */
int vsf_get_max_feature_index(vsf_ctx* ctx, int32_t stream, uint64_t* out);
int vsf_set_max_feature_index(vsf_ctx* ctx, int32_t stream, uint64_t value);

/* ---- batched device-resident API (multi-stream configs; frames already in HBM) ----
 * d_frames points to `streams` frames in device memory, frame s at d_frames + s*frame_stride_bytes,
 * rows `row_stride` bytes apart.  Frames are BORROWED: a batch passed to vsf_batch_track /
 * vsf_batch_step must stay valid and unmodified until the next such call has completed, because it
 * is level 0 of the "previous" pyramid of that next call (the reference holds the previous cv::Mat
 * the same way, FeatureTrackerLucasKanade.cpp:30,106).  `hip_stream` is a hipStream_t (NULL = the
 * default stream); calls only enqueue work and do not synchronise. */
int vsf_batch_track(vsf_ctx* ctx, const uint8_t* d_frames, int64_t frame_stride_bytes,
                    int32_t row_stride, void* hip_stream);
int vsf_batch_detect(vsf_ctx* ctx, const uint8_t* d_frames, int64_t frame_stride_bytes,
                     int32_t row_stride, const uint8_t* d_masks, int64_t mask_stride_bytes,
                     int32_t mask_row_stride, void* hip_stream);
/* cv::cvtColor(CV_BGR2GRAY) of `streams` device-resident multi-channel frames (3 or 4 interleaved bytes per pixel,
 * B first) into 8-bit frames the batched calls take; one kernel on hip_stream. */
int vsf_batch_bgr_to_gray(vsf_ctx* ctx, const uint8_t* d_bgr, int64_t bgr_frame_stride_bytes, int32_t bgr_row_stride,
                          int32_t channels, uint8_t* d_gray, int64_t gray_frame_stride_bytes, int32_t gray_row_stride,
                          void* hip_stream);
/* One frontend step per stream: Track, then UpdateDetection on the same frame. */
int vsf_batch_step(vsf_ctx* ctx, const uint8_t* d_frames, int64_t frame_stride_bytes,
                   int32_t row_stride, void* hip_stream);

/* Host <-> device copies of the feature tables of streams [first, first+count).  Arrays are
 * [count][capacity] (xy: [count][capacity][2]); n is [count].  Synchronous. */
int vsf_batch_upload(vsf_ctx* ctx, int32_t first, int32_t count, const uint64_t* ids,
                     const double* xy, const uint32_t* age, const int32_t* n);
int vsf_batch_download(vsf_ctx* ctx, int32_t first, int32_t count, uint64_t* ids, double* xy,
                       uint32_t* age, int32_t* n, vsf_track_stats* stats /* [count] or NULL */);
/* Device pointers of the resident tables (for a collective over xGMI): ids [streams][capacity]
 * u64, xy [streams][capacity][2] f64, age [streams][capacity] u32, n [streams] i32. */
int vsf_batch_device_tables(vsf_ctx* ctx, void** ids, void** xy, void** age, void** n,
                            int32_t* capacity);
/* Packs every stream's table into fixed-size 24-byte records {u64 id; f32 x, y; u32 age; u32 valid}
 * [streams][capacity] in the caller's device buffer (streams*capacity*24 bytes), padded with
 * valid = 0, for the RCCL gather of tracks.  Enqueues one kernel on hip_stream. */
int vsf_batch_export_tracks(vsf_ctx* ctx, void* d_records, void* hip_stream);
/* ---- multi-GPU: one process per GPU, streams sharded over the ranks, no data-path collective.
 * The one exchange is an all-gather of the 24-byte track records over RCCL / xGMI.  No counterpart in the
 * reference (single process, single device; its loop is example/vslam/example_serialization.cpp:159-171): this is
 * what a C++ host running that loop on one GPU per process adds to see every rank's tracks.
 * librccl is opened at run time (dlopen), so single-GPU users need not have it.
 *   rank 0: vsf_comm_unique_id(id), hand the VSF_COMM_ID_BYTES bytes to the other ranks by any means (file, socket,
 *   MPI); every rank: vsf_comm_create(id, rank, nranks, device, &comm) -- collective, like ncclCommInitRank;
 *   or vsf_comm_wrap() around an ncclComm_t the host already has.  nranks == 1 needs neither id nor librccl. */
#define VSF_COMM_ID_BYTES 128
typedef struct vsf_comm vsf_comm;
int vsf_comm_unique_id(uint8_t* id /* [VSF_COMM_ID_BYTES] */);
int vsf_comm_create(const uint8_t* id, int32_t rank, int32_t nranks, int32_t device_id, vsf_comm** out);
int vsf_comm_wrap(void* nccl_comm /* ncclComm_t */, int32_t device_id, vsf_comm** out);
void vsf_comm_destroy(vsf_comm* comm);
int vsf_comm_rank(const vsf_comm* comm, int32_t* rank, int32_t* nranks);
const char* vsf_comm_last_error(void);
/* vsf_batch_export_tracks of this rank into slot `rank` of d_recv, then ncclAllGather in place: afterwards d_recv
 * holds nranks x streams x capacity records, rank-major, on every rank (every rank's context must have the same
 * streams and capacity).  Enqueues on hip_stream, does not synchronise.  comm == NULL: a world of one (export only). */
int vsf_gather_tracks(vsf_ctx* ctx, vsf_comm* comm, void* d_recv, void* hip_stream);

/* Sticky per-stream device status of the batched calls: 0 or VSF_E_CAPACITY.  Synchronous. */
int vsf_batch_status(vsf_ctx* ctx, int32_t* status /* [streams] */);
/* Forget the previous frame (next vsf_batch_track / vsf_track only stores its frame). */
int vsf_reset_tracker(vsf_ctx* ctx);

/* ---- building blocks exposed for parity tests (synchronous, stream 0, host pointers) ---- */
/* OpenCV buildOpticalFlowPyramid levels; out_levels[l] receives level l densely (w_l*h_l bytes),
 * l = 1..*max_level_out; returns the effective maxLevel in *max_level_out. */
int vsf_debug_pyramid(vsf_ctx* ctx, const uint8_t* img, int32_t w, int32_t h, int32_t stride,
                      uint8_t** out_levels, int32_t* max_level_out);
/* cv::calcOpticalFlowPyrLK on n points (status as OpenCV: 1 = found). */
int vsf_debug_lk(vsf_ctx* ctx, const uint8_t* prev, const uint8_t* next, int32_t w, int32_t h,
                 const float* prev_pts, float* next_pts, uint8_t* status, int32_t n);
/* cv::cornerMinEigenVal of grid cell `cell` (column-major index): out receives the dense
 * [cell_h][cell_w] map (at most w*h floats), rect (may be NULL) the cell's x, y, w, h. */
int vsf_debug_min_eig(vsf_ctx* ctx, const uint8_t* img, int32_t w, int32_t h, int32_t stride,
                      int32_t cell, float* out, int32_t* rect);

/* Test hook: lowers the candidate-key slots select_kernel keeps per grid cell (never above what vsf_create
 * allocated; >= 16), so that a test can force the exact but slow overflow path of the selection. */
int vsf_debug_set_cand_cap(vsf_ctx* ctx, int32_t slots);

/* ---- frontend step around the path (KimeraFrontend::ProcessFrame, src/vslam/frontend/KimeraFrontend.cpp:247-304) ---- */
typedef struct vsf_camera_ds {
  double fx, fy, cx, cy, xi, alpha; /* DoubleSphereCameraModel ctor (src/vslam/data/Camera.cpp:101-122) */
} vsf_camera_ds;
typedef struct vsf_keyframe_params {
  double keyframe_interval;             /* KimeraFrontendParameter.cpp:11  5.0 */
  double minimum_keyframe_interval;     /* :12  0.5 */
  uint32_t low_keyframe_feature_number; /* :13  200 */
  int32_t counting_feature_age;         /* :14  2 */
} vsf_keyframe_params;
/* What ProcessFrame does with the tracker's output before RemoveOutlier, for every stream of the
 * context: bearings = camera_model->Unproject(pos), {0,0,1} where Unproject throws (:247-257); count of
 * features with age >= counting_feature_age (:266-269); keyframe decision from that count and
 * dt_since_keyframe[s] = timestamp - last_keyframe->timestamp (:272-293); for a keyframe every age of
 * the stream is incremented (:302-304).  d_bearings: device buffer of streams*capacity*3 doubles (same
 * indexing as the tables); dt_since_keyframe, is_keyframe_out, aged_out: host arrays of `streams`
 * entries (the last two may be NULL).  Synchronises the stream before returning. */
int vsf_batch_frontend_post_track(vsf_ctx* ctx, const vsf_camera_ds* cam, const vsf_keyframe_params* kp,
                                  const double* dt_since_keyframe, double* d_bearings, int32_t* is_keyframe_out,
                                  uint32_t* aged_out, void* hip_stream);
/* DoubleSphereCameraModel::Unproject on n host points (src/vslam/data/Camera.cpp:181-189): bearings
 * receives 3n doubles, valid n bytes (0 where the reference throws ProjectionErrorException; the
 * bearing is then {0,0,1} as at KimeraFrontend.cpp:255). */
int vsf_unproject(vsf_ctx* ctx, const vsf_camera_ds* cam, const double* xy, int32_t n, double* bearings,
                  uint8_t* valid);

/* FrameToFrameOpticalFlow::TrackPoint (forward only, frame_to_frame_optical_flow_single_thread.cpp:84-112)
 * on n points of a frame pair, with the context's lssd_* parameters: next_pts 2n floats, valid n bytes. */
int vsf_debug_lssd(vsf_ctx* ctx, const uint8_t* prev, const uint8_t* next, int32_t w, int32_t h,
                   const float* prev_pts, float* next_pts, uint8_t* valid, int32_t n);
/* The 3x3 inverse the patch set-up takes (patch.h: H_se2.ldlt().solve(Identity), Eigen's pivoted LDLT restated)
 * on n row-major matrices: every pivot order can be checked against the oracle. */
int vsf_debug_lssd_inverse3(vsf_ctx* ctx, const float* H, float* Hinv, int32_t n);
/* Level `level` (1..lssd_levels) of the u16 mip pyramid (basalt ManagedImagePyr::setFromImage of the frame
 * shifted left by 8): out receives (w >> level) * (h >> level) uint16. */
int vsf_debug_lssd_pyramid(vsf_ctx* ctx, const uint8_t* img, int32_t w, int32_t h, int32_t stride, int32_t level,
                           uint16_t* out);

/* Intermediates of the last vsf_detect / vsf_batch_detect of an ANMS context, for one stream: the mask
 * cv::GFTTDetector saw (FeatureDetectorANMS.cpp:107-132; [h][w] bytes, may be NULL) and the detected
 * keypoints in cv order as x, y int16 pairs (kp may be NULL; at most anms_detection_features pairs). */
int vsf_debug_anms(vsf_ctx* ctx, int32_t stream, uint8_t* flag_out, int16_t* kp_out, int32_t* n_kp_out);

/* ---- one entry point that allocates nothing ----
 * No counterpart in the reference: it exists for callers that cannot call hipMalloc (code under HIP graph
 * capture, tight serving loops, harnesses that forbid allocation).  cv::calcOpticalFlowPyrLK on n points like
 * vsf_debug_lk, but frames, points, next points and status are DEVICE pointers the caller owns, there is no
 * context, and the scratch the kernels need (levels >= 1 of both pyramids) lives in a 4 MiB device array inside
 * the library (one copy per GPU, held until the process ends).  p supplies width, height, klt_* , min_eig_thr and
 * device_id.  Frames whose two pyramids do not fit 4 MiB (above about 5.9 Mpixel) return VSF_E_CAPACITY.  The
 * call only enqueues kernels on hip_stream; it calls no hipMalloc / hipFree.  All calls on one GPU share that
 * GPU's array: calls that overlap on one GPU (different streams, threads) race; serialise them. */
int vsf_lk_points_noalloc(const vsf_params* p, const uint8_t* d_prev, int32_t prev_stride, const uint8_t* d_next,
                          int32_t next_stride, const float* d_prev_pts, float* d_next_pts, uint8_t* d_status,
                          int32_t n, void* hip_stream);

/* ---- in-library kernel timing (HIP events on the launch stream), used by bench.py ---- */
#define VSF_K_PYRAMID 0
#define VSF_K_LK 1
#define VSF_K_COMPACT 2
#define VSF_K_OCCUPANCY 3
#define VSF_K_SCORE 4
#define VSF_K_CANDIDATES 5
#define VSF_K_SELECT 6
#define VSF_K_APPEND 7
#define VSF_K_COUNT 8
/* on: 0 = off, 1 = every kernel class above, otherwise a mask with bit (1 + VSF_K_x) set for the classes to time
 * (e.g. 1 << (1 + VSF_K_LK): the tracker only).  An event pair around a kernel drains the stream before and after
 * it -- about 8 us of idle GPU each; sixteen of them cost the C2 step 3-5 % --, so a caller that times a whole
 * run asks for the one class it needs and takes the full breakdown from a few extra steps. */
int vsf_profile_enable(vsf_ctx* ctx, int32_t on);
/* Diagnostic: the shader clock the GPU holds right now.  A one-lane kernel on a stream of its own spins for spin_us
 * microseconds of the constant-rate counter (s_memrealtime) and reports the advance of the shader cycle counter
 * (s_memtime) over it; called while a step's kernels are queued on another stream it measures the clock under that
 * load.  Blocks for spin_us; allocates 32 bytes.  realtime_mhz (may be NULL) = the constant counter's rate. */
int vsf_debug_sclk_probe(vsf_ctx* ctx, int32_t spin_us, double* sclk_mhz, double* realtime_mhz);
/* Sums the recorded event pairs: total milliseconds and launch count per kernel id.  Synchronous;
 * clears the record. */
int vsf_profile_read(vsf_ctx* ctx, float* total_ms /* [VSF_K_COUNT] */,
                     int32_t* launches /* [VSF_K_COUNT] */);

#ifdef __cplusplus
}
#endif
#endif
