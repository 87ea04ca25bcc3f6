// vsf_lssd.hpp -- FeatureTrackerLSSDLucasKanade on the device (ref: src/vslam/feature/
// FeatureTrackerLSSDLucasKanade.cpp:35-119, LSSDOpticalFlow/frame_to_frame_optical_flow_single_thread.cpp,
// patch.h, patterns.h): basalt's frame-to-frame SE(2) patch tracker with Pattern24, float arithmetic,
// forward + backward tracking and the recovered-distance test.
// One 32-lane half-wavefront per feature, lane i = pattern point i (24 of 32 lanes carry a point).
// Every sum over the pattern is a 32-slot butterfly (xor 1, 2, 4, 8, 16; DPP and permlane moves) -- the order the oracle's
// ORC_LSSD_TREE mode uses -- and every other expression is written exactly as in oracle/vsf_oracle.c,
// so the results are equal bit for bit (the build uses -ffp-contract=off; float division and sqrt are
// correctly rounded).  Included by vsf_kernels.hpp.
#pragma once

namespace vsf {

constexpr int LSSD_PATTERN = 24;
constexpr int LSSD_MAXL = 8;

// ref: LSSDOpticalFlow/patterns.h:60-73 (Pattern24::pattern_raw)
__device__ const float LSSD_PAT[32][2] = {
    {-1, 5},  {1, 5},  {-3, 3},  {-1, 3},  {1, 3},   {3, 3},   {-5, 1},  {-3, 1}, {-1, 1}, {1, 1}, {3, 1},
    {5, 1},   {-5, -1}, {-3, -1}, {-1, -1}, {1, -1}, {3, -1},  {5, -1},  {-3, -3}, {-1, -3}, {1, -3}, {3, -3},
    {-1, -5}, {1, -5}, {0, 0},   {0, 0},   {0, 0},   {0, 0},   {0, 0},   {0, 0},  {0, 0},  {0, 0}};

// u16 mip pyramid of `streams` frames: level 0 is the caller's u8 frame read as (v << 8)
// (FeatureTrackerLSSDLucasKanade.cpp:42-51), levels >= 1 are dense u16 [stream][h][w].
struct Pyr16View {
  const uint8_t* base0;
  long long sstride0;
  int pitch0;
  const uint16_t* lv[LSSD_MAXL];
  int w[LSSD_MAXL], h[LSSD_MAXL];
  int levels;  // optical_flow_levels: images 0..levels
};

struct LssdParams {
  int levels;
  int max_iterations;
  double max_recovered_dist;  // compared with the SQUARED distance (frame_to_frame_..._single_thread.cpp:67)
};

// basalt::ManagedImagePyr<uint16_t>::subsample: [1 4 6 4 1]^2 in int, REFLECT_101, (v + 128) >> 8, size (w/2, h/2).
// SRC8: source is the u8 frame read as v << 8.
// 64x16 output tile per block: the 131x35 source tile is staged in LDS, filtered along x into u32, then along y
// (integer arithmetic, so the order of the two passes does not change the result).
template <bool SRC8>
__global__ __launch_bounds__(256) void lssd_subsample_kernel(const void* __restrict__ src, long long src_sstride,
                                                             int src_pitch, int sw, int sh,
                                                             uint16_t* __restrict__ dst, int dw, int dh) {
  using src_t = typename std::conditional<SRC8, uint8_t, uint16_t>::type;
  __shared__ src_t s_src[PD_SH * PD_SP];
  __shared__ uint32_t s_h[PD_SH * PD_TW];
  const int stream = blockIdx.z;
  const src_t* S = reinterpret_cast<const src_t*>(src) + (long long)stream * src_sstride;
  const int ox = blockIdx.x * PD_TW, oy = blockIdx.y * PD_TH;
  const int sx0 = 2 * ox - 2, sy0 = 2 * oy - 2;
  for (int i = threadIdx.x; i < PD_SH * PD_SW; i += 256) {
    const int y = i / PD_SW, x = i - y * PD_SW;
    const int gy = reflect101(sy0 + y, sh), gx = reflect101(sx0 + x, sw);
    s_src[y * PD_SP + x] = S[(long long)gy * src_pitch + gx];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PD_SH * PD_TW; i += 256) {
    const int y = i / PD_TW, x = i - y * PD_TW;
    const src_t* r = s_src + y * PD_SP + 2 * x;
    const uint32_t v = (uint32_t)r[0] + 4u * r[1] + 6u * r[2] + 4u * r[3] + r[4];
    s_h[i] = SRC8 ? (v << 8) : v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PD_TH * PD_TW; i += 256) {
    const int y = i / PD_TW, x = i - y * PD_TW;
    const int gx = ox + x, gy = oy + y;
    if (gx < dw && gy < dh) {
      const uint32_t* c = s_h + (2 * y) * PD_TW + x;
      const uint32_t v = c[0] + 4u * c[PD_TW] + 6u * c[2 * PD_TW] + 4u * c[3 * PD_TW] + c[4 * PD_TW];
      dst[((long long)stream * dh + gy) * dw + gx] = (uint16_t)((v + 128u) >> 8);
    }
  }
}

struct Lvl16 {  // one level of one stream
  const uint8_t* p8;
  const uint16_t* p16;
  int pitch, w, h;
};
__device__ __forceinline__ Lvl16 lssd_level(const Pyr16View& P, int stream, int level) {
  Lvl16 L;
  L.w = P.w[level];
  L.h = P.h[level];
  if (level == 0) {
    L.p8 = P.base0 + (long long)stream * P.sstride0;
    L.p16 = nullptr;
    L.pitch = P.pitch0;
  } else {
    L.p8 = nullptr;
    L.p16 = P.lv[level] + (long long)stream * L.w * L.h;
    L.pitch = L.w;
  }
  return L;
}
// L0: the level is the caller's u8 frame read as v << 8; otherwise a u16 level.  Offsets fit 32 bits (a level of
// one stream is far below 4 GB), which keeps the address arithmetic to one multiply-add per pixel.
template <bool L0>
__device__ __forceinline__ float lssd_px(const Lvl16& im, int x, int y) {
  const unsigned int off = (unsigned int)(y * im.pitch + x);
  if (L0) return (float)((int)im.p8[off] << 8);
  return (float)im.p16[off];
}
// basalt::Image::InBounds(p, border), floating point p
__device__ __forceinline__ bool lssd_in_bounds(const Lvl16& im, float x, float y, float border) {
  return border <= x && x < ((float)im.w - border - 1.f) && border <= y && y < ((float)im.h - border - 1.f);
}
// basalt::Image::interp<float>
template <bool L0>
__device__ __forceinline__ float lssd_interp(const Lvl16& im, float x, float y) {
  const int ix = (int)x, iy = (int)y;
  const float dx = x - (float)ix, dy = y - (float)iy;
  const float ddx = 1.f - dx, ddy = 1.f - dy;
  return ddx * ddy * lssd_px<L0>(im, ix, iy) + ddx * dy * lssd_px<L0>(im, ix, iy + 1) + dx * ddy * lssd_px<L0>(im, ix + 1, iy) +
         dx * dy * lssd_px<L0>(im, ix + 1, iy + 1);
}
// basalt::Image::interpGrad<float>
template <bool L0>
__device__ __forceinline__ void lssd_interp_grad(const Lvl16& im, float x, float y, float (&out)[3]) {
  const int ix = (int)x, iy = (int)y;
  const float dx = x - (float)ix, dy = y - (float)iy;
  const float ddx = 1.f - dx, ddy = 1.f - dy;
  const float px0y0 = lssd_px<L0>(im, ix, iy), px1y0 = lssd_px<L0>(im, ix + 1, iy);
  const float px0y1 = lssd_px<L0>(im, ix, iy + 1), px1y1 = lssd_px<L0>(im, ix + 1, iy + 1);
  out[0] = ddx * ddy * px0y0 + ddx * dy * px0y1 + dx * ddy * px1y0 + dx * dy * px1y1;
  const float pxm1y0 = lssd_px<L0>(im, ix - 1, iy), pxm1y1 = lssd_px<L0>(im, ix - 1, iy + 1);
  const float res_mx = ddx * ddy * pxm1y0 + ddx * dy * pxm1y1 + dx * ddy * px0y0 + dx * dy * px0y1;
  const float px2y0 = lssd_px<L0>(im, ix + 2, iy), px2y1 = lssd_px<L0>(im, ix + 2, iy + 1);
  const float res_px = ddx * ddy * px1y0 + ddx * dy * px1y1 + dx * ddy * px2y0 + dx * dy * px2y1;
  out[1] = 0.5f * (res_px - res_mx);
  const float px0ym1 = lssd_px<L0>(im, ix, iy - 1), px1ym1 = lssd_px<L0>(im, ix + 1, iy - 1);
  const float res_my = ddx * ddy * px0ym1 + ddx * dy * px0y0 + dx * ddy * px1ym1 + dx * dy * px1y0;
  const float px0y2 = lssd_px<L0>(im, ix, iy + 2), px1y2 = lssd_px<L0>(im, ix + 1, iy + 2);
  const float res_py = ddx * ddy * px0y1 + ddx * dy * px0y2 + dx * ddy * px1y1 + dx * dy * px1y2;
  out[2] = 0.5f * (res_py - res_my);
}

// Butterfly sum over a 32-lane half, every lane of the half gets the total.  The partner of step k is lane i ^ k;
// after the steps 1 and 2 a quad is uniform, so lane i ^ 7 (row_half_mirror) holds what lane i ^ 4 holds, and
// lane i ^ 15 (row_mirror) what lane i ^ 8 holds; the last step adds the two 16-lane rows of the half (float
// addition commutes, so both rows get the same bits).  DPP / permlane moves only: no LDS round trips.
template <int CTRL>
__device__ __forceinline__ float lssd_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lssd_sum32(float v) {
  v = v + lssd_dpp<0xB1>(v);   // quad_perm [1,0,3,2]   : i ^ 1
  v = v + lssd_dpp<0x4E>(v);   // quad_perm [2,3,0,1]   : i ^ 2
  v = v + lssd_dpp<0x141>(v);  // row_half_mirror       : i ^ 7  (== i ^ 4 here)
  v = v + lssd_dpp<0x140>(v);  // row_mirror            : i ^ 15 (== i ^ 8 here)
  // rows are uniform now; v_permlane16_swap (gfx950) exchanges row 1 of one copy with row 0 of the other
  // (and row 3 with row 2), so the two copies hold the two row totals of each half
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ int lssd_count32(bool pred) {
  const unsigned long long b = __ballot(pred);
  const unsigned int half = (threadIdx.x & 32) ? (unsigned int)(b >> 32) : (unsigned int)b;
  return __popc(half);
}

__device__ __forceinline__ void lssd_sincos(float t, float& s, float& c) {  // oracle lssd_sincos, TREE mode
  const float t2 = t * t;
  s = t * (1.f + t2 * (-1.f / 6.f + t2 * (1.f / 120.f + t2 * (-1.f / 5040.f + t2 * (1.f / 362880.f)))));
  c = 1.f + t2 * (-0.5f + t2 * (1.f / 24.f + t2 * (-1.f / 720.f + t2 * (1.f / 40320.f + t2 * (-1.f / 3628800.f)))));
}

// Eigen::LDLT<Matrix3f> compute + solveInPlace(Identity), the arithmetic of oracle lssd_ldlt_inverse3 without
// its dynamically indexed arrays (which cost ~200 select instructions per call here).
//   * The in-place factorisation touches the lower triangle only; with the pivot of step 0 in {0,1,2} and of
//     step 1 in {1,2} the symmetric swaps are fixed register exchanges under two small branches.
//   * solve(Identity) is P^T (L D L^T)^-1 P column by column.  A column of P is a unit vector, and running
//     the substitutions on unit vector e_r performs exactly the operations the oracle performs on the column of
//     P that equals e_r, so W = (L D L^T)^-1 is formed once on e_0, e_1, e_2 and the result is the symmetric
//     re-indexing Hinv[i][c] = W[rho(i)][rho(c)], rho = inverse of the pivot permutation.
__device__ __forceinline__ float lssd_sel3(int i, float a, float b, float c) { return i == 0 ? a : (i == 1 ? b : c); }

__device__ inline void lssd_ldlt_inverse3(const float (&H)[3][3], float (&Hinv)[3][3]) {
  float a00 = H[0][0], a10 = H[1][0], a11 = H[1][1], a20 = H[2][0], a21 = H[2][1], a22 = H[2][2];
  auto swp = [](float& x, float& y) {
    const float t = x;
    x = y;
    y = t;
  };
  // k = 0: largest |diagonal| (first one wins ties)
  int t0 = 0;
  float bv = fabsf(a00);
  if (fabsf(a11) > bv) {
    bv = fabsf(a11);
    t0 = 1;
  }
  if (fabsf(a22) > bv) t0 = 2;
  if (t0 == 1) {
    swp(a20, a21);
    swp(a00, a11);
  } else if (t0 == 2) {
    swp(a00, a22);
    swp(a10, a21);
  }
  if (fabsf(a00) > 0.f) {
    a10 /= a00;
    a20 /= a00;
  }
  // k = 1: the remaining diagonal entries are compared as they stand (left-looking update)
  const int t1 = fabsf(a22) > fabsf(a11) ? 2 : 1;
  if (t1 == 2) {
    swp(a10, a20);
    swp(a11, a22);
  }
  {
    const float temp0 = a00 * a10;
    a11 -= a10 * temp0;
    a21 -= a20 * temp0;
    if (fabsf(a11) > 0.f) a21 /= a11;
  }
  // k = 2
  {
    const float temp0 = a00 * a20, temp1 = a11 * a21;
    a22 -= a20 * temp0 + a21 * temp1;
  }
  // W = (L D L^T)^-1 on the unit vectors
  const float tol = 1.f / 3.4028234663852886e+38f;
  const float d[3] = {a00, a11, a22};
  float W[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    float x[3] = {r == 0 ? 1.f : 0.f, r == 1 ? 1.f : 0.f, r == 2 ? 1.f : 0.f};
    // run-time values to the optimiser, as the entries of P are in the oracle: folding 0 - a * 1 into -a
    // would turn +0 into -0 on exactly diagonal matrices
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
    x[1] -= a10 * x[0];
    x[2] -= a20 * x[0];
    x[2] -= a21 * x[1];
#pragma unroll
    for (int i = 0; i < 3; i++) x[i] = fabsf(d[i]) > tol ? x[i] / d[i] : 0.f;
    x[1] -= a21 * x[2];
    x[0] -= a10 * x[1];
    x[0] -= a20 * x[2];
    W[0][r] = x[0];
    W[1][r] = x[1];
    W[2][r] = x[2];
  }
  // rho for (t0, t1) = (0,1) (0,2) (1,1) (1,2) (2,1) (2,2): two bits per case
  const int code = 2 * (2 * t0 + (t1 - 1));
  const int rho[3] = {(0x690 >> code) & 3, (0x909 >> code) & 3, (0x066 >> code) & 3};
  float Tm[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int c = 0; c < 3; c++) Tm[i][c] = lssd_sel3(rho[i], W[0][c], W[1][c], W[2][c]);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int c = 0; c < 3; c++) Hinv[i][c] = lssd_sel3(rho[c], Tm[i][0], Tm[i][1], Tm[i][2]);
}

// One pyramid level of TrackPoint for the feature of this half-wavefront: the patch of the old image
// (OpticalFlowPatch::setFromImage, patch.h:75-131) and TrackPointAtLevel (:114-146).
struct LssdTrack {
  float L00, L01, L10, L11;  // transform.linear()
  float t0, t1;              // transform.translation(), full-resolution pixels
  bool valid;                // patch_valid
};

template <bool L0>
__device__ __forceinline__ void lssd_level_pass(const Lvl16& im1, const Lvl16& im2, int level, float pinx, float piny,
                                                float patx, float paty, bool has_pt, LssdTrack& T,
                                                const LssdParams& prm) {
  const float scale = (float)(1 << level);
  const float inv_scale = 1.f / scale;  // a power of two: x * inv_scale == x / scale bit for bit
  float data = -1.f, M0 = 0.f, M1 = 0.f, M2 = 0.f;
  bool valid = T.valid;
  const bool lvl_active = valid;  // the reference leaves the level loop once the patch is invalid
  float L00 = T.L00, L01 = T.L01, L10 = T.L10, L11 = T.L11;
  float s0 = T.t0, s1 = T.t1;  // translation at this level's scale
  if (lvl_active) {
    s0 = T.t0 * inv_scale;
    s1 = T.t1 * inv_scale;
  }
  // ---- OpticalFlowPatch::setFromImage (patch.h:75-131)
  {
    const float x = pinx * inv_scale + patx, y = piny * inv_scale + paty;
    const bool inb = valid && has_pt && lssd_in_bounds(im1, x, y, 2.f);
    float val = 0.f, gx = 0.f, gy = 0.f;
    if (inb) {
      float vg[3];
      lssd_interp_grad<L0>(im1, x, y, vg);
      data = vg[0];
      val = vg[0];
      gx = vg[1];
      gy = vg[2];
    }
    const int num_valid = lssd_count32(inb);
    const float sum = lssd_sum32(val), gsx = lssd_sum32(gx), gsy = lssd_sum32(gy);
    const float mean_inv = (float)num_valid / sum;
    float jx = 0.f, jy = 0.f;
    if (data >= 0.f) {
      const float d = data;
      jx = ((float)num_valid * (gx * sum - gsx * d)) / (sum * sum);
      jy = ((float)num_valid * (gy * sum - gsy * d)) / (sum * sum);
      data = d * mean_inv;
    }
    const float J0 = jx, J1 = jy, J2 = jx * (-paty) + jy * patx;
    float H[3][3], Hinv[3][3];
    H[0][0] = lssd_sum32(J0 * J0);
    H[0][1] = H[1][0] = lssd_sum32(J0 * J1);
    H[0][2] = H[2][0] = lssd_sum32(J0 * J2);
    H[1][1] = lssd_sum32(J1 * J1);
    H[1][2] = H[2][1] = lssd_sum32(J1 * J2);
    H[2][2] = lssd_sum32(J2 * J2);
    lssd_ldlt_inverse3(H, Hinv);
    M0 = (Hinv[0][0] * J0 + Hinv[0][1] * J1) + Hinv[0][2] * J2;
    M1 = (Hinv[1][0] * J0 + Hinv[1][1] * J1) + Hinv[1][2] * J2;
    M2 = (Hinv[2][0] * J0 + Hinv[2][1] * J1) + Hinv[2][2] * J2;
    if (!has_pt) {
      M0 = 0.f;
      M1 = 0.f;
      M2 = 0.f;
    }
  }
  // ---- TrackPointAtLevel (:114-146)
  for (int it = 0; it < prm.max_iterations; it++) {
    if (!__any(valid)) break;
    const float px = (L00 * patx + L01 * paty) + s0, py = (L10 * patx + L11 * paty) + s1;
    const bool inb = valid && has_pt && lssd_in_bounds(im2, px, py, 2.f);
    float res = -1.f, val = 0.f;
    if (inb) {
      res = lssd_interp<L0>(im2, px, py);
      val = res;
    }
    const int num_valid = lssd_count32(inb);
    const float sum = lssd_sum32(val);
    const bool use = has_pt && res >= 0.f && data >= 0.f;
    float r = 0.f;
    if (use) r = ((float)num_valid * res) / sum - data;
    const int num_res = lssd_count32(use);
    // slots without a pattern point hold +0 like the oracle's padding (not the -0 of (-0) * 0)
    const float inc0 = lssd_sum32(has_pt ? (-M0) * r : 0.f), inc1 = lssd_sum32(has_pt ? (-M1) * r : 0.f),
                inc2 = lssd_sum32(has_pt ? (-M2) * r : 0.f);
    if (valid) {
      if (num_res > LSSD_PATTERN / 2) {
        const float theta = inc2;  // Sophus::SE2f::exp
        float sn, cs;
        lssd_sincos(theta, sn, cs);
        float sbt, omcbt;
        if (fabsf(theta) < 1e-5f) {
          const float theta_sq = theta * theta;
          sbt = 1.f - (1.f / 6.f) * theta_sq;
          omcbt = 0.5f * theta - (1.f / 24.f) * theta * theta_sq;
        } else {
          sbt = sn / theta;
          omcbt = (1.f - cs) / theta;
        }
        const float ex = sbt * inc0 - omcbt * inc1;
        const float ey = omcbt * inc0 + sbt * inc1;
        // transform *= SE2::exp(inc).matrix(): (2x3) * (3x3) with last row (0 0 1)
        const float n00 = (L00 * cs + L01 * sn) + s0 * 0.f, n01 = (L00 * (-sn) + L01 * cs) + s0 * 0.f;
        const float n10 = (L10 * cs + L11 * sn) + s1 * 0.f, n11 = (L10 * (-sn) + L11 * cs) + s1 * 0.f;
        const float nt0 = (L00 * ex + L01 * ey) + s0 * 1.f, nt1 = (L10 * ex + L11 * ey) + s1 * 1.f;
        L00 = n00;
        L01 = n01;
        L10 = n10;
        L11 = n11;
        s0 = nt0;
        s1 = nt1;
        if (!lssd_in_bounds(im2, s0, s1, 2.f)) valid = false;
      } else {
        valid = false;
      }
    }
  }
  T.L00 = L00;
  T.L01 = L01;
  T.L10 = L10;
  T.L11 = L11;
  T.valid = valid;
  // TrackPoint :98/:106 scale the translation of a level that ran, whatever its outcome
  if (lvl_active) {
    T.t0 = s0 * scale;
    T.t1 = s1 * scale;
  }
}

// FrameToFrameOpticalFlow::TrackPoint (frame_to_frame_optical_flow_single_thread.cpp:84-146) for the
// feature of this half-wavefront.  `active`: the half holds a feature (control flow below is per half:
// the two halves of a wave may run different trip counts, lanes of a finished half idle).
__device__ bool lssd_track_point(const Pyr16View& OLD, const Pyr16View& CUR, int stream, bool active, float pinx,
                                 float piny, float& tx, float& ty, const LssdParams& prm) {
  const int li = threadIdx.x & 31;
  const float patx = LSSD_PAT[li][0], paty = LSSD_PAT[li][1];
  const bool has_pt = li < LSSD_PATTERN;
  LssdTrack T = {1.f, 0.f, 0.f, 1.f, tx, ty, active};
  for (int level = prm.levels; level >= 0; level--) {
    if (!__any(T.valid)) break;
    const Lvl16 im1 = lssd_level(OLD, stream, level), im2 = lssd_level(CUR, stream, level);
    if (level == 0)
      lssd_level_pass<true>(im1, im2, 0, pinx, piny, patx, paty, has_pt, T, prm);
    else
      lssd_level_pass<false>(im1, im2, level, pinx, piny, patx, paty, has_pt, T, prm);
  }
  tx = T.t0;
  ty = T.t1;
  return T.valid;
}

// The reference tracker keeps its own id -> transform map between calls
// (FeatureTrackerLSSDLucasKanade.cpp:70-91): an id that was in the map after the previous Track keeps
// the float translation stored there, whatever the caller's table says now (the bucketing detector
// overwrites the position of the feature whose id equals its counter, Q6); only ids new to the map
// start from the table position.  The map after a Track is the table after that Track, so the state
// is a snapshot of (id, float xy) per stream.
struct LssdState {
  const uint64_t* ids;  // [streams][cap]
  const float2* xy;     // [streams][cap]
  const int* n;         // [streams]
};

// flags as lk_fb_kernel: bit0 forward valid, bit1 backward valid, bit2 kept.  Two features per wave.
__global__ __launch_bounds__(64, 6) void lssd_fb_kernel(Pyr16View P, Pyr16View C, const uint64_t* __restrict__ ids,
                                                     const double2* __restrict__ xy, const int* __restrict__ n,
                                                     int cap, int fgrid, int nstreams, LssdParams prm, LssdState st,
                                                     float2* __restrict__ fwd, uint8_t* __restrict__ flags) {
  // same XCD-aware map as lk_fb_kernel: grid x = 8 * fgrid, y = groups of eight streams
  const int xcd = blockIdx.x & 7, pair0 = blockIdx.x >> 3;
  const int stream = (int)blockIdx.y * 8 + xcd;
  if (stream >= nstreams) return;
  const int nf = n[stream];
  const int half = (threadIdx.x >> 5) & 1, sub = threadIdx.x & 31;
  const long long sb = (long long)stream * cap;
  const int st_n = st.n[stream];
  for (int f0 = 2 * pair0; f0 < nf; f0 += 2 * fgrid) {
    const int f = f0 + half;
    const bool active = f < nf;
    double2 p0 = make_double2(0.0, 0.0);
    int hit = -1;
    if (active) {
      p0 = xy[sb + f];
      const uint64_t id = ids[sb + f];
      if (f < st_n && st.ids[sb + f] == id) {
        hit = f;  // the usual case: survivors keep their slots, the detector appends behind them
      } else {
        for (int j = sub; j < st_n; j += 32)
          if (st.ids[sb + j] == id) hit = j;
      }
    }
#pragma unroll
    for (int step = 1; step < 32; step <<= 1) hit = max(hit, __shfl_xor(hit, step, 32));
    float p1x = (float)p0.x, p1y = (float)p0.y;  // transform.translation() = pos.cast<float>() (:64,:77)
    if (hit >= 0) {
      const float2 q = st.xy[sb + hit];
      p1x = q.x;
      p1y = q.y;
    }
    float qx = p1x, qy = p1y;
    uint8_t flag = 0;
    const bool v1 = lssd_track_point(P, C, stream, active, p1x, p1y, qx, qy, prm);
    float rx = qx, ry = qy;
    const bool v2 = lssd_track_point(C, P, stream, v1, qx, qy, rx, ry, prm);
    if (v1) flag |= 1;
    if (v2) {
      flag |= 2;
      const float ddx = p1x - rx, ddy = p1y - ry;
      const float dist2 = ddx * ddx + ddy * ddy;
      if ((double)dist2 < prm.max_recovered_dist) flag |= 4;  // :67
    }
    if (active && sub == 0) {
      fwd[sb + f] = make_float2(qx, qy);
      flags[sb + f] = flag;
    }
  }
}

// state <- table (after the first call's early return :57-68, and after every later Track :102)
__global__ void lssd_snapshot_kernel(const uint64_t* __restrict__ ids, const double2* __restrict__ xy,
                                     const int* __restrict__ n, int cap, uint64_t* __restrict__ st_ids,
                                     float2* __restrict__ st_xy, int* __restrict__ st_n) {
  const int stream = blockIdx.y;
  const int nf = n[stream];
  const long long sb = (long long)stream * cap;
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += gridDim.x * blockDim.x) {
    const double2 p = xy[sb + f];
    st_ids[sb + f] = ids[sb + f];
    st_xy[sb + f] = make_float2((float)p.x, (float)p.y);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) st_n[stream] = nf;
}

// lssd_ldlt_inverse3 on n matrices (row-major 3x3), for parity tests of every pivot order
__global__ void lssd_inverse3_kernel(const float* __restrict__ H, float* __restrict__ Hinv, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float h[3][3], o[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) h[r][c] = H[9 * i + 3 * r + c];
  lssd_ldlt_inverse3(h, o);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) Hinv[9 * i + 3 * r + c] = o[r][c];
}

// TrackPoint (forward only) on a point list, for parity tests
__global__ __launch_bounds__(64) void lssd_points_kernel(Pyr16View P, Pyr16View C, const float2* __restrict__ pts, int n,
                                                         LssdParams prm, float2* __restrict__ out,
                                                         uint8_t* __restrict__ valid) {
  const int f = blockIdx.x * 2 + ((threadIdx.x >> 5) & 1);
  const bool active = f < n;
  float2 p = make_float2(0.f, 0.f);
  if (active) p = pts[f];
  float qx = p.x, qy = p.y;
  const bool v = lssd_track_point(P, C, 0, active, p.x, p.y, qx, qy, prm);
  if (active && (threadIdx.x & 31) == 0) {
    out[f] = make_float2(qx, qy);
    valid[f] = v ? 1 : 0;
  }
}

}  // namespace vsf
