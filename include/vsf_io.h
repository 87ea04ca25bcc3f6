/* vsf_io.h -- C ABI of libvsf_io.so: the ingestion format in front of the hot path.
 *
 * Host-only code (no HIP): an EuRoC "mav0/cam0" directory plus a basalt calibration file, as read by
 * the reference's vslam::dataprovider::EurocKimeraDataProvider
 * (src/vslam/io/EurocKimeraDataProvider.cpp:21-228), turned into what the frontend takes per frame:
 * timestamp [s], 8-bit frame, mask, double-sphere camera.  Frames decoded here are handed to
 * vsf_track / vsf_detect (vsf.h) or staged into a batch for vsf_batch_step.
 *
 * All functions return VSF_IO_OK (0) or a negative VSF_IO_E_* code; vsf_io_last_error() gives the text of
 * the calling thread's last failure.  Plain pointers and sizes only.
 */
#ifndef VSF_IO_H_
#define VSF_IO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSF_IO_OK 0
#define VSF_IO_E_INVALID (-1)     /* bad argument */
#define VSF_IO_E_FILE (-2)        /* file missing / unreadable */
#define VSF_IO_E_FORMAT (-3)      /* malformed PNG / CSV / JSON */
#define VSF_IO_E_UNSUPPORTED (-4) /* valid file, but a kind this library does not decode */
#define VSF_IO_E_RANGE (-5)       /* index past the last line (the reference returns std::nullopt) */

const char* vsf_io_last_error(void);

/* ---- PNG -> 8-bit grayscale ---------------------------------------------------------------------
 * replaces cv::imread(path, cv::IMREAD_GRAYSCALE) at EurocKimeraDataProvider.cpp:36-37,59,83 for the
 * files EuRoC holds (8-bit grayscale, non-interlaced).  Decoded: colour type 0 (gray) at 1/2/4/8/16
 * bits and colour type 4 (gray + alpha) at 8/16 bits, non-interlaced; 16-bit samples keep their high
 * byte, 1/2/4-bit samples are scaled to 0..255, alpha is dropped (libpng strip_16 /
 * expand_gray_1_2_4_to_8 / strip_alpha, which is what OpenCV's decoder asks for).  Chunk CRCs and the
 * zlib checksum are verified.  RGB / palette / interlaced files: VSF_IO_E_UNSUPPORTED (OpenCV would
 * convert RGB with libpng's rgb_to_gray; EuRoC has none). */
int vsf_png_info(const uint8_t* file, size_t nbytes, int32_t* width, int32_t* height, int32_t* bit_depth,
                 int32_t* color_type);
int vsf_png_decode_gray8(const uint8_t* file, size_t nbytes, uint8_t* out, int32_t out_stride);
int vsf_png_read_gray8(const char* path, uint8_t* out, int32_t out_stride, int32_t expect_width,
                       int32_t expect_height);

/* ---- EuRoC provider -------------------------------------------------------------------------------
 * vsf_euroc_open   = EurocKimeraDataProvider ctor (:21-41): LogParser(root/mav0/cam0/data.csv),
 *                    ParseSensorPose(root/mav0/cam0/sensor.yaml), ParseCameraParameters(calibration),
 *                    optional mask PNG thresholded at 10 (cv::threshold(.., 10, 255, THRESH_BINARY)).
 * vsf_euroc_frame  = GetInput(index) (:70-91): timestamp_s = double(timestamp_ns) * 1e-9, image =
 *                    imread(root/mav0/cam0/data/<filename>, GRAYSCALE).
 * The stateful GetInput() (:43-68) is `index = last_index_++` on the host side of this ABI. */
typedef struct vsf_euroc vsf_euroc;

typedef struct {
  double fx, fy, cx, cy, xi, alpha; /* same layout as vsf_camera_ds (vsf.h) */
  int32_t width, height;            /* value0.resolution[0] */
} vsf_euroc_camera;

int vsf_euroc_open(const char* path_to_dataset_root, const char* path_to_calibration_file,
                   const char* path_to_mask_image /* NULL or "" = none */, vsf_euroc** out);
void vsf_euroc_close(vsf_euroc* e);
int vsf_euroc_size(const vsf_euroc* e, uint64_t* n_frames);
/* line `index` of data.csv; *filename points into the provider and stays valid until close */
int vsf_euroc_entry(const vsf_euroc* e, uint64_t index, uint64_t* timestamp_ns, const char** filename);
/* frame size from the first image (EuRoC: 752x480) */
int vsf_euroc_frame_size(vsf_euroc* e, int32_t* width, int32_t* height);
int vsf_euroc_frame(vsf_euroc* e, uint64_t index, double* timestamp_s, uint8_t* out, int32_t out_stride,
                    int32_t width, int32_t height);
int vsf_euroc_camera_model(const vsf_euroc* e, vsf_euroc_camera* out);
/* mask: 0 / 255, width x height of the mask file; has_mask = 0 when none was given */
int vsf_euroc_mask(const vsf_euroc* e, int32_t* has_mask, int32_t* width, int32_t* height, uint8_t* out,
                   int32_t out_stride);
/* pose_body_T_sensor_ (:168-226): rotation row-major [9] (re-normalised through a quaternion), translation [3];
 * identity when sensor.yaml is missing or malformed, like the reference's catch block.  The reference fills
 * matrix element (1,2) from data[5] instead of data[6] (:177-178); reproduced. */
int vsf_euroc_sensor_pose(const vsf_euroc* e, double* rotation9, double* translation3);

#ifdef __cplusplus
}
#endif
#endif /* VSF_IO_H_ */
