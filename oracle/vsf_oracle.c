/*
 * vsf_oracle.c -- CPU ORACLE (test infrastructure only; see vsf_oracle.h for the "parity
 * unpinned" statement).  Plain C99, no dependencies.  Build: see oracle/Makefile
 * (-O2 -ffp-contract=off, no fast-math, so every float expression below is evaluated exactly
 * as written).
 *
 * Citations: "ref:" = file:line under /root/reference (the wrapper logic this follows);
 * "cv:" = the OpenCV 3.4 function whose published algorithm is restated (SURVEY.md Appendix A;
 * OpenCV sources are not available offline).
 */
#include "vsf_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* cv: borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0)
      p = -p;
    else
      p = 2 * len - 2 - p;
  }
  return p;
}

/* cv: cvRound(float) == round-half-to-even under the default rounding mode */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) { return (int)floorf(v); }

#define CV_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

/* ------------------------------------------------------------------------------------------ */
/* cv: pyrDown (8-bit, 5x5 [1 4 6 4 1]^2 / 256, BORDER_REFLECT_101) -- SURVEY A.2             */
/* ------------------------------------------------------------------------------------------ */
void orc_pyr_down_u8(const uint8_t* src, int w, int h, int stride, uint8_t* dst) {
  const int dw = (w + 1) / 2, dh = (h + 1) / 2;
  /* Rows are independent: threads share them when the caller is not already inside a parallel region (OpenMP
   * leaves a nested region with one thread), which is how the single-stream CPU baseline uses its cores --
   * OpenCV's parallel_for_ over rows / points.  Same arithmetic per row either way. */
#pragma omp parallel if (dh >= 32)
  {
  int* rows = (int*)malloc(sizeof(int) * (size_t)dw * 5);
#pragma omp for schedule(static)
  for (int y = 0; y < dh; y++) {
    for (int k = 0; k < 5; k++) {
      const int sy = reflect101(2 * y - 2 + k, h);
      const uint8_t* s = src + (size_t)sy * stride;
      int* r = rows + (size_t)k * dw;
      for (int x = 0; x < dw; x++) {
        const int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w);
        const int x2 = reflect101(2 * x, w), x3 = reflect101(2 * x + 1, w);
        const int x4 = reflect101(2 * x + 2, w);
        r[x] = s[x0] + 4 * s[x1] + 6 * s[x2] + 4 * s[x3] + s[x4];
      }
    }
    for (int x = 0; x < dw; x++) {
      const int v = rows[x] + 4 * rows[dw + x] + 6 * rows[2 * dw + x] + 4 * rows[3 * dw + x] +
                    rows[4 * dw + x];
      dst[(size_t)y * dw + x] = (uint8_t)((v + 128) >> 8);
    }
  }
  free(rows);
  }
}

/* cv: buildOpticalFlowPyramid(img, pyr, winSize, maxLevel, withDerivatives=false).
 * Level 0 is the image; level l = pyrDown(level l-1); after building level l the next size is
 * computed and the build stops (returning l) when it is <= the window in either dimension.
 * The per-level REFLECT_101 border OpenCV stores is replaced by index reflection at the readers
 * (identical values). */
int orc_build_pyramid(const uint8_t* img, int w, int h, int stride, int win, int max_level,
                      orc_pyramid* out) {
  memset(out, 0, sizeof(*out));
  if (max_level >= ORC_MAX_LEVELS) max_level = ORC_MAX_LEVELS - 1;
  int cw = w, ch = h;
  for (int level = 0; level <= max_level; level++) {
    orc_level* L = &out->lv[level];
    L->w = cw;
    L->h = ch;
    L->data = (uint8_t*)malloc((size_t)cw * ch);
    if (level == 0) {
      for (int y = 0; y < ch; y++) memcpy(L->data + (size_t)y * cw, img + (size_t)y * stride, cw);
    } else {
      const orc_level* P = &out->lv[level - 1];
      orc_pyr_down_u8(P->data, P->w, P->h, P->w, L->data);
    }
    out->max_level = level;
    cw = (cw + 1) / 2;
    ch = (ch + 1) / 2;
    if (cw <= win || ch <= win) break;
  }
  return out->max_level;
}

void orc_free_pyramid(orc_pyramid* p) {
  for (int i = 0; i < ORC_MAX_LEVELS; i++) {
    free(p->lv[i].data);
    p->lv[i].data = NULL;
  }
}

/* cv: calcSharrDeriv (video/lkpyramid.cpp) -- SURVEY A.3.  dxy is interleaved (Ix, Iy) int16. */
void orc_scharr_s16(const uint8_t* src, int w, int h, int stride, int16_t* dxy) {
#pragma omp parallel if (h >= 32)
  {
  int* t0 = (int*)malloc(sizeof(int) * (size_t)(w + 2) * 2);
  int* t1 = t0 + (w + 2);
#pragma omp for schedule(static)
  for (int y = 0; y < h; y++) {
    const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : (h > 1 ? 1 : 0)) * stride;
    const uint8_t* r1 = src + (size_t)y * stride;
    const uint8_t* r2 = src + (size_t)(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0)) * stride;
    for (int x = 0; x < w; x++) {
      t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
      t1[x + 1] = r2[x] - r0[x];
    }
    const int xl = (w > 1 ? 1 : 0), xr = (w > 1 ? w - 2 : 0);
    t0[0] = t0[xl + 1];
    t0[w + 1] = t0[xr + 1];
    t1[0] = t1[xl + 1];
    t1[w + 1] = t1[xr + 1];
    int16_t* d = dxy + (size_t)y * w * 2;
    for (int x = 0; x < w; x++) {
      d[2 * x] = (int16_t)(t0[x + 2] - t0[x]);
      d[2 * x + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
    }
  }
  free(t0);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* cv: calcOpticalFlowPyrLK / LKTrackerInvoker scalar path -- SURVEY A.4                       */
/* ------------------------------------------------------------------------------------------ */
static inline int lvl_I(const orc_level* L, int x, int y) {
  return L->data[(size_t)reflect101(y, L->h) * L->w + reflect101(x, L->w)];
}
/* derivative image is padded with zeros (BORDER_CONSTANT) */
static inline int lvl_D(const orc_level* L, const int16_t* d, int x, int y, int c) {
  if (x < 0 || y < 0 || x >= L->w || y >= L->h) return 0;
  return d[((size_t)y * L->w + x) * 2 + c];
}

/* debug statistics (not part of any result): iterations and level passes executed */
long long orc_dbg_iters = 0, orc_dbg_level_passes = 0, orc_dbg_restage_moves = 0;

static void lk_level(const orc_level* I, const int16_t* dI, const orc_level* J,
                     const float* prev_pts, float* next_pts, uint8_t* status, float* err, int n,
                     int win, int level, int max_level, int max_count, double epsilon,
                     double min_eig_thr, int acc_mode) {
  const float half = (win - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (1 << 20);
  const int W_BITS = 14;
  /* points are independent (cv: parallel_for_ over the points, LKTrackerInvoker) */
#pragma omp parallel if (n >= 8)
  {
  int16_t* Iw = (int16_t*)malloc(sizeof(int16_t) * (size_t)win * win * 3);
  int16_t* dIw = Iw + (size_t)win * win;
#pragma omp for schedule(dynamic, 4)
  for (int i = 0; i < n; i++) {
    float px = prev_pts[2 * i] * (float)(1. / (1 << level));
    float py = prev_pts[2 * i + 1] * (float)(1. / (1 << level));
    float nx, ny;
    if (level == max_level) {
      nx = px;
      ny = py;
    } else {
      nx = next_pts[2 * i] * 2.f;
      ny = next_pts[2 * i + 1] * 2.f;
    }
    next_pts[2 * i] = nx;
    next_pts[2 * i + 1] = ny;

    px -= half;
    py -= half;
    int ipx = cv_floor_f(px), ipy = cv_floor_f(py);
    if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
      if (level == 0) {
        status[i] = 0;
        if (err) err[i] = 0;
      }
      continue;
    }
    float a = px - ipx, b = py - ipy;
    int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
    int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
    int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
    int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

    float fA11 = 0, fA12 = 0, fA22 = 0;
    int64_t eA11 = 0, eA12 = 0, eA22 = 0;
    for (int y = 0; y < win; y++) {
      for (int x = 0; x < win; x++) {
        const int X = ipx + x, Y = ipy + y;
        int ival = CV_DESCALE(lvl_I(I, X, Y) * iw00 + lvl_I(I, X + 1, Y) * iw01 +
                                  lvl_I(I, X, Y + 1) * iw10 + lvl_I(I, X + 1, Y + 1) * iw11,
                              W_BITS - 5);
        int ixval = CV_DESCALE(lvl_D(I, dI, X, Y, 0) * iw00 + lvl_D(I, dI, X + 1, Y, 0) * iw01 +
                                   lvl_D(I, dI, X, Y + 1, 0) * iw10 +
                                   lvl_D(I, dI, X + 1, Y + 1, 0) * iw11,
                               W_BITS);
        int iyval = CV_DESCALE(lvl_D(I, dI, X, Y, 1) * iw00 + lvl_D(I, dI, X + 1, Y, 1) * iw01 +
                                   lvl_D(I, dI, X, Y + 1, 1) * iw10 +
                                   lvl_D(I, dI, X + 1, Y + 1, 1) * iw11,
                               W_BITS);
        Iw[y * win + x] = (int16_t)ival;
        dIw[(y * win + x) * 2] = (int16_t)ixval;
        dIw[(y * win + x) * 2 + 1] = (int16_t)iyval;
        if (acc_mode == ORC_ACC_FLOAT_SEQ) {
          fA11 += (float)(ixval * ixval);
          fA12 += (float)(ixval * iyval);
          fA22 += (float)(iyval * iyval);
        } else {
          eA11 += (int64_t)ixval * ixval;
          eA12 += (int64_t)ixval * iyval;
          eA22 += (int64_t)iyval * iyval;
        }
      }
    }
    if (acc_mode != ORC_ACC_FLOAT_SEQ) {
      fA11 = (float)eA11;
      fA12 = (float)eA12;
      fA22 = (float)eA22;
    }
    float A11 = fA11 * FLT_SCALE, A12 = fA12 * FLT_SCALE, A22 = fA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                   (float)(2 * win * win);
    if (minEig < min_eig_thr || D < FLT_EPSILON) {
      if (level == 0) status[i] = 0;
      continue;
    }
    D = 1.f / D;
    nx -= half;
    ny -= half;
    float pdx = 0, pdy = 0;
#pragma omp atomic
    orc_dbg_level_passes++;
    for (int j = 0; j < max_count; j++) {
      int inx = cv_floor_f(nx), iny = cv_floor_f(ny);
#pragma omp atomic
      orc_dbg_iters++;
      if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
        if (level == 0) status[i] = 0;
        break;
      }
      a = nx - inx;
      b = ny - iny;
      iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
      iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
      iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
      iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      float fb1 = 0, fb2 = 0;
      int64_t eb1 = 0, eb2 = 0;
      for (int y = 0; y < win; y++) {
        for (int x = 0; x < win; x++) {
          const int X = inx + x, Y = iny + y;
          int diff = CV_DESCALE(lvl_I(J, X, Y) * iw00 + lvl_I(J, X + 1, Y) * iw01 +
                                    lvl_I(J, X, Y + 1) * iw10 + lvl_I(J, X + 1, Y + 1) * iw11,
                                W_BITS - 5) -
                     Iw[y * win + x];
          if (acc_mode == ORC_ACC_FLOAT_SEQ) {
            fb1 += (float)(diff * dIw[(y * win + x) * 2]);
            fb2 += (float)(diff * dIw[(y * win + x) * 2 + 1]);
          } else {
            eb1 += (int64_t)diff * dIw[(y * win + x) * 2];
            eb2 += (int64_t)diff * dIw[(y * win + x) * 2 + 1];
          }
        }
      }
      if (acc_mode != ORC_ACC_FLOAT_SEQ) {
        fb1 = (float)eb1;
        fb2 = (float)eb2;
      }
      float b1 = fb1 * FLT_SCALE, b2 = fb2 * FLT_SCALE;
      float dx = (float)((A12 * b2 - A22 * b1) * D);
      float dy = (float)((A12 * b1 - A11 * b2) * D);
      nx += dx;
      ny += dy;
      next_pts[2 * i] = nx + half;
      next_pts[2 * i + 1] = ny + half;
      if ((double)dx * dx + (double)dy * dy <= epsilon) break;
      if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
        next_pts[2 * i] -= dx * 0.5f;
        next_pts[2 * i + 1] -= dy * 0.5f;
        break;
      }
      pdx = dx;
      pdy = dy;
    }
    /* default flags: err requested, OPTFLOW_LK_GET_MIN_EIGENVALS not set */
    if (status[i] && err && level == 0) {
      float fx = next_pts[2 * i] - half, fy = next_pts[2 * i + 1] - half;
      int inx = cv_floor_f(fx), iny = cv_floor_f(fy);
      if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
        status[i] = 0;
        continue;
      }
      float aa = fx - inx, bb = fy - iny;
      iw00 = cv_round_f((1.f - aa) * (1.f - bb) * (1 << W_BITS));
      iw01 = cv_round_f(aa * (1.f - bb) * (1 << W_BITS));
      iw10 = cv_round_f((1.f - aa) * bb * (1 << W_BITS));
      iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      float errval = 0.f;
      for (int y = 0; y < win; y++)
        for (int x = 0; x < win; x++) {
          const int X = inx + x, Y = iny + y;
          int diff = CV_DESCALE(lvl_I(J, X, Y) * iw00 + lvl_I(J, X + 1, Y) * iw01 +
                                    lvl_I(J, X, Y + 1) * iw10 + lvl_I(J, X + 1, Y + 1) * iw11,
                                W_BITS - 5) -
                     Iw[y * win + x];
          errval += (float)abs(diff);
        }
      err[i] = errval * 1.f / (32 * win * win);
    }
  }
  free(Iw);
  }
}

int orc_calc_optical_flow_pyr_lk(const uint8_t* prev, int prev_stride, const uint8_t* next,
                                 int next_stride, int w, int h, const float* prev_pts,
                                 float* next_pts, uint8_t* status, float* err, int n, int win,
                                 int max_level, int max_count, double eps, double min_eig_thr,
                                 int acc_mode) {
  if (n <= 0) return 0;
  /* cv: criteria clamp -- maxCount in [0,100], epsilon in [0,10], epsilon *= epsilon */
  if (max_count < 0) max_count = 0;
  if (max_count > 100) max_count = 100;
  if (eps < 0.) eps = 0.;
  if (eps > 10.) eps = 10.;
  eps *= eps;

  orc_pyramid P, N;
  int ml = orc_build_pyramid(prev, w, h, prev_stride, win, max_level, &P);
  int ml2 = orc_build_pyramid(next, w, h, next_stride, win, max_level, &N);
  if (ml2 < ml) ml = ml2;
  for (int i = 0; i < n; i++) {
    status[i] = 1;
    if (err) err[i] = 0;
  }
  int16_t* deriv = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)w * h);
  for (int level = ml; level >= 0; level--) {
    const orc_level* I = &P.lv[level];
    orc_scharr_s16(I->data, I->w, I->h, I->w, deriv);
    lk_level(I, deriv, &N.lv[level], prev_pts, next_pts, status, err, n, win, level, ml, max_count,
             eps, min_eig_thr, acc_mode);
  }
  free(deriv);
  orc_free_pyramid(&P);
  orc_free_pyramid(&N);
  return ml;
}

/* ------------------------------------------------------------------------------------------ */
/* cv: cornerMinEigenVal on a sub-matrix (ROI) of a parent image -- SURVEY A.1 steps 1-4       */
/* ------------------------------------------------------------------------------------------ */
/* Sobel reads the parent's real pixels outside the ROI and reflects only at the true image
 * border.  Dx: row kernel [-1 0 1] (exact), column kernel [1 2 1]*scale as float32, evaluated
 * (S[y-1]+S[y+1])*k1 + S[y]*k0 (cv: SymmColumnSmallFilter).  Dy: row kernel [1 2 1]*scale applied
 * to the 8-bit source as k[0]*S[x-1]; += k[1]*S[x]; += k[2]*S[x+1] (cv: RowFilter<uchar,float>),
 * column kernel [-1 0 1] (exact).
 * Box filter: OpenCV's boxFilter accumulates 32F input in 64F (createBoxFilter: sumType=CV_64F),
 * so the block sum is formed in double -- row sums left to right, then the column of row sums top
 * to bottom -- and rounded to float32 once.  (SURVEY A.1 step 3 words this with float sums; the
 * double form is what OpenCV does.  OpenCV's ColumnSum -- and its RowSum for block sizes other than 3 and 5 -- is a
 * RUNNING sum: add the entering element, subtract the leaving one.  The two forms give the same double unless an
 * addition is inexact, which needs summands some 29 bits apart in magnitude: a Dy that should be 0 and is a
 * rounding residue of ~7e-9 beside gradients of 0.1.  That moves the double by its last bit and the float32 result
 * about once in 1e11 pixels; restating the running form would tie every pixel to the whole column above it.  Part of
 * what "PARITY UNPINNED" in the header covers.)
 * The box filter reflects (REFLECT_101) at the borders of the ROI's own covariance image. */
static void corner_response(const uint8_t* img, int img_w, int img_h, int stride, int rx, int ry, int rw, int rh,
                            int block_size, int use_harris, double harris_k, float* eig) {
  const double scale_d = 1.0 / ((double)(1 << 2) * block_size * 255.0);
  const float k_side = (float)(1.0 * scale_d), k_mid = (float)(2.0 * scale_d);
  float* cov = (float*)malloc(sizeof(float) * 3 * (size_t)rw * rh);
#pragma omp parallel for schedule(static) if (rh >= 32)
  for (int y = 0; y < rh; y++) {
    const int Y = ry + y;
    const int ym = reflect101(Y - 1, img_h), yp = reflect101(Y + 1, img_h);
    for (int x = 0; x < rw; x++) {
      const int X = rx + x;
      const int xm = reflect101(X - 1, img_w), xp = reflect101(X + 1, img_w);
      /* Dx */
      const float r_m = (float)(img[(size_t)ym * stride + xp] - img[(size_t)ym * stride + xm]);
      const float r_0 = (float)(img[(size_t)Y * stride + xp] - img[(size_t)Y * stride + xm]);
      const float r_p = (float)(img[(size_t)yp * stride + xp] - img[(size_t)yp * stride + xm]);
      const float dx = (r_m + r_p) * k_side + r_0 * k_mid;
      /* Dy */
      float s_m = k_side * (float)img[(size_t)ym * stride + xm];
      s_m += k_mid * (float)img[(size_t)ym * stride + X];
      s_m += k_side * (float)img[(size_t)ym * stride + xp];
      float s_p = k_side * (float)img[(size_t)yp * stride + xm];
      s_p += k_mid * (float)img[(size_t)yp * stride + X];
      s_p += k_side * (float)img[(size_t)yp * stride + xp];
      const float dy = s_p - s_m;
      float* c = cov + ((size_t)y * rw + x) * 3;
      c[0] = dx * dx;
      c[1] = dx * dy;
      c[2] = dy * dy;
    }
  }
  const int a0 = -(block_size / 2); /* anchor = block/2: offsets a0 .. a0+block-1 */
#pragma omp parallel for schedule(static) if (rh >= 32)
  for (int y = 0; y < rh; y++) {
    for (int x = 0; x < rw; x++) {
      double s[3] = {0, 0, 0};
      for (int j = 0; j < block_size; j++) {
        const int yy = reflect101(y + a0 + j, rh);
        double r[3] = {0, 0, 0};
        for (int i = 0; i < block_size; i++) {
          const int xx = reflect101(x + a0 + i, rw);
          const float* c = cov + ((size_t)yy * rw + xx) * 3;
          if (i == 0) {
            r[0] = (double)c[0];
            r[1] = (double)c[1];
            r[2] = (double)c[2];
          } else {
            r[0] += (double)c[0];
            r[1] += (double)c[1];
            r[2] += (double)c[2];
          }
        }
        if (j == 0) {
          s[0] = r[0];
          s[1] = r[1];
          s[2] = r[2];
        } else {
          s[0] += r[0];
          s[1] += r[1];
          s[2] += r[2];
        }
      }
      if (!use_harris) {
        const float a = (float)s[0] * 0.5f, b = (float)s[1], c = (float)s[2] * 0.5f;
        eig[(size_t)y * rw + x] = (a + c) - sqrtf((a - c) * (a - c) + b * b);
      } else {
        /* cv::cornerHarris -> calcHarris (imgproc/corner.cpp, recalled): the universal-intrinsics loop takes
         * four pixels at a time in float32 with (float)k -- ac_bb = a*c - b*b; t = a + c; dst = ac_bb - k*t*t,
         * products left to right -- and the scalar tail (the last width % 4 pixels of a row) evaluates
         * (float)(a*c - b*b - k*(a + c)*(a + c)) with k a double: a*c - b*b in float, the k term in double. */
        const float a = (float)s[0], b = (float)s[1], c = (float)s[2];
        float r;
        if (x < (rw & ~3)) {
          const float kf = (float)harris_k;
          const float ac_bb = a * c - b * b;
          const float t = a + c;
          r = ac_bb - kf * t * t;
        } else {
          r = (float)((double)(a * c - b * b) - harris_k * (double)(a + c) * (double)(a + c));
        }
        eig[(size_t)y * rw + x] = r;
      }
    }
  }
  free(cov);
}

void orc_corner_min_eigen_val(const uint8_t* img, int img_w, int img_h, int stride, int rx, int ry,
                              int rw, int rh, int block_size, float* eig) {
  corner_response(img, img_w, img_h, stride, rx, ry, rw, rh, block_size, 0, 0.0, eig);
}

void orc_corner_harris(const uint8_t* img, int img_w, int img_h, int stride, int rx, int ry, int rw, int rh,
                       int block_size, double k, float* dst) {
  corner_response(img, img_w, img_h, stride, rx, ry, rw, rh, block_size, 1, k, dst);
}

/* cv: goodFeaturesToTrack -- SURVEY A.1 steps 5-9.  ROI of a parent image; mask (if any) is the
 * matching ROI of the parent mask (mask points at the ROI origin).  Returns the corner count;
 * out_xy holds ROI-relative integer-valued float coordinates. */
typedef struct {
  float v;
  int idx;
} orc_cand;

static int cand_greater(const void* pa, const void* pb) {
  const orc_cand* a = (const orc_cand*)pa;
  const orc_cand* b = (const orc_cand*)pb;
  /* cv: greaterThanPtr -- value descending, ties by larger address first */
  if (a->v > b->v) return -1;
  if (a->v < b->v) return 1;
  return (a->idx > b->idx) ? -1 : (a->idx < b->idx ? 1 : 0);
}

int orc_good_features_to_track(const uint8_t* img, int img_w, int img_h, int stride, int rx, int ry,
                               int rw, int rh, int max_corners, double quality, double min_dist,
                               const uint8_t* mask, int mask_stride, int block_size, float* out_xy,
                               int cap) {
  return orc_good_features_to_track_ex(img, img_w, img_h, stride, rx, ry, rw, rh, max_corners, quality, min_dist,
                                       mask, mask_stride, block_size, 0, 0.04, out_xy, cap);
}

int orc_good_features_to_track_ex(const uint8_t* img, int img_w, int img_h, int stride, int rx, int ry,
                                  int rw, int rh, int max_corners, double quality, double min_dist,
                                  const uint8_t* mask, int mask_stride, int block_size, int use_harris,
                                  double harris_k, float* out_xy, int cap) {
  if (rw <= 0 || rh <= 0) return 0;
  float* eig = (float*)malloc(sizeof(float) * (size_t)rw * rh);
  corner_response(img, img_w, img_h, stride, rx, ry, rw, rh, block_size, use_harris, harris_k, eig);

  /* cv: minMaxLoc(eig, 0, &maxVal, 0, 0, mask) */
  double maxVal = 0;
  int any = 0;
  for (int y = 0; y < rh; y++)
    for (int x = 0; x < rw; x++) {
      if (mask && !mask[(size_t)y * mask_stride + x]) continue;
      const double v = eig[(size_t)y * rw + x];
      if (!any || v > maxVal) {
        maxVal = v;
        any = 1;
      }
    }
  if (!any) maxVal = 0;
  /* cv: threshold(eig, eig, maxVal*quality, 0, THRESH_TOZERO) */
  const float thr = (float)(maxVal * quality);
  for (size_t i = 0; i < (size_t)rw * rh; i++) eig[i] = eig[i] > thr ? eig[i] : 0.f;

  /* cv: dilate 3x3 (outside ignored) + local-maximum test on the inner (rw-2)x(rh-2) pixels */
  orc_cand* cand = (orc_cand*)malloc(sizeof(orc_cand) * ((size_t)rw * rh + 1));
  int nc = 0;
  for (int y = 1; y < rh - 1; y++)
    for (int x = 1; x < rw - 1; x++) {
      const float val = eig[(size_t)y * rw + x];
      if (val == 0) continue;
      float m = val;
      for (int j = -1; j <= 1; j++)
        for (int i = -1; i <= 1; i++) {
          const float q = eig[(size_t)(y + j) * rw + (x + i)];
          if (q > m) m = q;
        }
      if (val == m && (!mask || mask[(size_t)y * mask_stride + x])) {
        cand[nc].v = val;
        cand[nc].idx = y * rw + x;
        nc++;
      }
    }
  qsort(cand, (size_t)nc, sizeof(orc_cand), cand_greater);

  int ncorners = 0;
  if (nc > 0) {
    if (min_dist >= 1) {
      const int cell = cv_round_d(min_dist);
      const int gw = (rw + cell - 1) / cell, gh = (rh + cell - 1) / cell;
      /* buckets as linked lists over accepted corners */
      int* head = (int*)malloc(sizeof(int) * (size_t)gw * gh);
      int* nxt = (int*)malloc(sizeof(int) * ((size_t)nc + 1));
      for (int i = 0; i < gw * gh; i++) head[i] = -1;
      const double md2d = min_dist * min_dist; /* cv: minDistance *= minDistance (double) */
      for (int i = 0; i < nc; i++) {
        const int y = cand[i].idx / rw, x = cand[i].idx - y * rw;
        int good = 1;
        const int xc = x / cell, yc = y / cell;
        int x1 = xc - 1, y1 = yc - 1, x2 = xc + 1, y2 = yc + 1;
        if (x1 < 0) x1 = 0;
        if (y1 < 0) y1 = 0;
        if (x2 > gw - 1) x2 = gw - 1;
        if (y2 > gh - 1) y2 = gh - 1;
        for (int yy = y1; yy <= y2 && good; yy++)
          for (int xx = x1; xx <= x2 && good; xx++)
            for (int k = head[yy * gw + xx]; k >= 0; k = nxt[k]) {
              const float ddx = (float)x - out_xy[2 * k], ddy = (float)y - out_xy[2 * k + 1];
              /* cv: float dx*dx + dy*dy compared with the double minDistance^2 */
              if ((double)(ddx * ddx + ddy * ddy) < md2d) {
                good = 0;
                break;
              }
            }
        if (good) {
          if (ncorners >= cap) break;
          out_xy[2 * ncorners] = (float)x;
          out_xy[2 * ncorners + 1] = (float)y;
          nxt[ncorners] = head[yc * gw + xc];
          head[yc * gw + xc] = ncorners;
          ncorners++;
          if (max_corners > 0 && ncorners == max_corners) break;
        }
      }
      free(head);
      free(nxt);
    } else {
      for (int i = 0; i < nc; i++) {
        if (ncorners >= cap) break;
        const int y = cand[i].idx / rw, x = cand[i].idx - y * rw;
        out_xy[2 * ncorners] = (float)x;
        out_xy[2 * ncorners + 1] = (float)y;
        ncorners++;
        if (max_corners > 0 && ncorners == max_corners) break;
      }
    }
  }
  free(cand);
  free(eig);
  return ncorners;
}

/* ------------------------------------------------------------------------------------------ */
/* reference wrapper: FeatureDetectorShiTomasiBucketing                                        */
/* ------------------------------------------------------------------------------------------ */
/* ref: src/vslam/feature/FeatureDetectorShiTomasiBucketing.cpp:38-48 (ctor; index starts at 0) */
void orc_detector_init(orc_detector* d, int32_t rows, int32_t cols, int32_t max_features,
                       double min_distance) {
  d->division_number_row = rows;
  d->division_number_col = cols;
  d->max_feature_number = max_features;
  d->min_feature_distance = min_distance;
  d->max_feature_index = 0;
}

/* ref: FeatureDetectorShiTomasiBucketing.cpp:13-36 (GenerateGrid, column-major, Rect2f) and
 * :144-149 (Rect2f -> Rect conversion = cv::saturate_cast<int>(float) = cvRound on x, y, width,
 * height independently).  Returns the number of cells. */
void orc_bgr2gray(const uint8_t* src, int w, int h, int stride, int cn, uint8_t* dst, int dst_stride) {
  /* cv RGB2Gray<uchar>: tab[b] + tab[g + 256] + tab[r + 512] with the rounding constant folded into the r table */
  for (int y = 0; y < h; y++) {
    const uint8_t* s = src + (size_t)y * stride;
    uint8_t* d = dst + (size_t)y * dst_stride;
    for (int x = 0; x < w; x++, s += cn)
      d[x] = (uint8_t)((s[0] * 1868 + s[1] * 9617 + s[2] * 4899 + (1 << 13)) >> 14);
  }
}

int orc_generate_grid(int w, int h, int cols, int rows, int* rects, float* origin) {
  const double dx = (double)w / (double)cols, dy = (double)h / (double)rows;
  int k = 0;
  for (int x = 0; x < cols; x++)
    for (int y = 0; y < rows; y++) {
      const float lx = (float)(x * dx), ly = (float)(y * dy);
      const float rxf = lx + (float)dx, ryf = ly + (float)dy; /* Point2f + Point2f */
      const float wf = rxf - lx, hf = ryf - ly;               /* Rect2f(pt1, pt2) */
      rects[4 * k + 0] = cv_round_f(lx);
      rects[4 * k + 1] = cv_round_f(ly);
      rects[4 * k + 2] = cv_round_f(wf);
      rects[4 * k + 3] = cv_round_f(hf);
      if (origin) { /* Rect2f::x, ::y -- what :169 adds to the cell-local corner */
        origin[2 * k] = lx;
        origin[2 * k + 1] = ly;
      }
      k++;
    }
  return k;
}

/* ref: FeatureDetectorShiTomasiBucketing.cpp:67-239 (DetectShiTomasiCorners).  SoA in/out:
 * ids/xy/age hold *n_inout entries on entry; new features are upserted (an id that already exists
 * is overwritten in place, exactly like unordered_map::operator[]), others appended.
 * Returns 0, or -1 when capacity is exhausted, -2 when a cell rectangle leaves the image
 * (cv::Mat::operator() would throw). */
/* The detector's nearest-neighbour rejection (Q5), FeatureDetectorShiTomasiBucketing.cpp:168-185: a corner is kept
 * when the kd-tree over the PRE-EXISTING features is empty (:172-176; KDTree.cpp:100-102,163-165) or when its
 * distance to the nearest of them is strictly above min_feature_distance (:178-180).  kd-tree nearest_point ==
 * exact minimum of the squared distance (kdtree/KDTree.cpp:154-233), so brute force decides the same; pinned
 * against the reference's own KDTree.cpp by tests/test_ref_kdtree.py (oracle/_ref). */
int orc_nn_accept(const double* kd_points, int n0, double cx, double cy, double min_feature_distance) {
  if (n0 <= 0) return 1;
  double best = 0;
  for (int i = 0; i < n0; i++) {
    const double ddx = cx - kd_points[2 * i], ddy = cy - kd_points[2 * i + 1];
    const double d2 = ddx * ddx + ddy * ddy;
    if (i == 0 || d2 < best) best = d2;
  }
  return sqrt(best) > min_feature_distance; /* :178-180 strict > on the L2 norm */
}

int orc_detector_update(orc_detector* d, const uint8_t* img, int w, int h, int stride,
                        const uint8_t* mask, int mask_stride, uint64_t* ids, double* xy,
                        uint32_t* age, int* n_inout, int capacity) {
  const int n0 = *n_inout;
  int n = n0;
  const int cols = d->division_number_col, rows = d->division_number_row;
  const int ncell = cols * rows;
  int* rects = (int*)malloc(sizeof(int) * 4 * (size_t)ncell);
  float* origin = (float*)malloc(sizeof(float) * 2 * (size_t)ncell);
  orc_generate_grid(w, h, cols, rows, rects, origin); /* :114-115 */
  const int grid_max = d->max_feature_number / (rows * cols); /* :116-117 */

  /* :120-129 flag image; every pre-existing feature is a kd-tree point (brute force here: only
   * the nearest distance is used, kdtree/KDTree.cpp:154-233 is an exact NN search) */
  uint8_t* flag = (uint8_t*)calloc((size_t)w * h, 1);
  for (int i = 0; i < n0; i++) {
    const double px = xy[2 * i], py = xy[2 * i + 1];
    if (px > 0 && px < w && py > 0 && py < h) flag[(size_t)((int)py) * w + (int)px] = 1;
  }
  /* :121-130 the kd-tree is built ONCE from the positions at entry: a feature whose slot the
   * first new id overwrites (:142 starts at max_feature_index, not +1) keeps its old position in it */
  double* kd_points = (double*)malloc(sizeof(double) * 2 * (size_t)(n0 > 0 ? n0 : 1));
  memcpy(kd_points, xy, sizeof(double) * 2 * (size_t)n0);

  int rc = 0;
  uint64_t feature_index = d->max_feature_index; /* :142 */
  float* pts = (float*)malloc(sizeof(float) * 2 * (size_t)(grid_max > 0 ? grid_max : 1));
  for (int c = 0; c < ncell; c++) {
    const int rx = rects[4 * c], ry = rects[4 * c + 1], rw = rects[4 * c + 2],
              rh = rects[4 * c + 3];
    if (rx < 0 || ry < 0 || rw < 0 || rh < 0 || rx + rw > w || ry + rh > h) {
      rc = -2;
      break;
    }
    int num = 0; /* :152 countNonZero */
    for (int y = 0; y < rh; y++)
      for (int x = 0; x < rw; x++) num += flag[(size_t)(ry + y) * w + rx + x] != 0;
    const int query = grid_max - num;
    if (query <= 0) break; /* :154 -- break, not continue */
    const uint8_t* m = mask ? mask + (size_t)ry * mask_stride + rx : NULL;
    const int got = orc_good_features_to_track(img, w, h, stride, rx, ry, rw, rh, query, 0.1,
                                               d->min_feature_distance, m, mask_stride, 3, pts,
                                               grid_max); /* :158-163 */
    for (int k = 0; k < got; k++) { /* :168-230 */
      /* :169 kp.pt.x + grect.x: float + float with the UNROUNDED Rect2f origin */
      const double cx = (double)(pts[2 * k] + origin[2 * c]);
      const double cy = (double)(pts[2 * k + 1] + origin[2 * c + 1]);
      const int accept = orc_nn_accept(kd_points, n0, cx, cy, d->min_feature_distance);
      if (accept) {
        int slot = -1;
        for (int i = 0; i < n; i++)
          if (ids[i] == feature_index) {
            slot = i;
            break;
          }
        if (slot < 0) {
          if (n >= capacity) {
            rc = -1;
            break;
          }
          slot = n++;
        }
        ids[slot] = feature_index;
        xy[2 * slot] = cx;
        xy[2 * slot + 1] = cy;
        age[slot] = 1;
        feature_index++;
      }
    }
    if (rc) break;
  }
  d->max_feature_index = feature_index - 1; /* :238 */
  *n_inout = n;
  free(pts);
  free(kd_points);
  free(flag);
  free(origin);
  free(rects);
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* reference wrapper: FeatureDetectorANMS (+ cv::circle, cv::sortIdx, anms::Ssc)               */
/* ------------------------------------------------------------------------------------------ */
/* cv: drawing.cpp Circle(), fill != 0.  Each step of the midpoint walk (dx from radius down, dy
 * from 0 up, while dx >= dy) paints the rows cy-+dy over [cx-dx, cx+dx] and the rows cy-+dx over
 * [cx-dy, cx+dy]; the clipped branch of the original paints the same spans cut to the image. */
static void hline_clip(uint8_t* img, int w, int h, int stride, int y, int x1, int x2, uint8_t c) {
  if (y < 0 || y >= h) return;
  if (x1 < 0) x1 = 0;
  if (x2 > w - 1) x2 = w - 1;
  for (int x = x1; x <= x2; x++) img[(size_t)y * stride + x] = c;
}

void orc_circle_filled(uint8_t* img, int w, int h, int stride, int cx, int cy, int radius,
                       uint8_t color) {
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
    hline_clip(img, w, h, stride, cy - dy, cx - dx, cx + dx, color);
    hline_clip(img, w, h, stride, cy + dy, cx - dx, cx + dx, color);
    hline_clip(img, w, h, stride, cy - dx, cx - dy, cx + dy, color);
    hline_clip(img, w, h, stride, cy + dx, cx - dy, cx + dy, color);
    dy++;
    err += plus;
    plus += 2;
    const int mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
}

int orc_circle_halfwidths(int radius, int* halfw) {
  for (int i = 0; i <= radius; i++) halfw[i] = -1;
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
    if (dx > halfw[dy]) halfw[dy] = dx;
    if (dy > halfw[dx]) halfw[dx] = dy;
    dy++;
    err += plus;
    plus += 2;
    const int mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
  return 0;
}

/* libstdc++ std::sort(first, last, comp) on an int index array with comp(a, b) = keys[a] < keys[b]
 * (cv: LessThanIdx), restated from bits/stl_algo.h / bits/stl_heap.h. */
typedef struct {
  const int* keys;
} idx_less;
static inline int il(const idx_less* c, int a, int b) { return c->keys[a] < c->keys[b]; }
static inline void iswap(int* a, int* b) {
  const int t = *a;
  *a = *b;
  *b = t;
}
static void ss_push_heap(int* first, long hole, long top, int value, const idx_less* c) {
  long parent = (hole - 1) / 2;
  while (hole > top && il(c, first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}
static void ss_adjust_heap(int* first, long hole, long len, int value, const idx_less* c) {
  const long top = hole;
  long child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (il(c, first[child], first[child - 1])) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  ss_push_heap(first, hole, top, value, c);
}
static void ss_make_heap(int* first, int* last, const idx_less* c) {
  const long len = last - first;
  if (len < 2) return;
  long parent = (len - 2) / 2;
  for (;;) {
    ss_adjust_heap(first, parent, len, first[parent], c);
    if (parent == 0) return;
    parent--;
  }
}
static void ss_pop_heap(int* first, int* last, int* result, const idx_less* c) {
  const int value = *result;
  *result = *first;
  ss_adjust_heap(first, 0, last - first, value, c);
}
static void ss_partial_sort_all(int* first, int* last, const idx_less* c) { /* middle == last */
  ss_make_heap(first, last, c);
  while (last - first > 1) {
    --last;
    ss_pop_heap(first, last, last, c);
  }
}
static void ss_move_median_to_first(int* result, int* a, int* b, int* cc, const idx_less* c) {
  if (il(c, *a, *b)) {
    if (il(c, *b, *cc))
      iswap(result, b);
    else if (il(c, *a, *cc))
      iswap(result, cc);
    else
      iswap(result, a);
  } else if (il(c, *a, *cc))
    iswap(result, a);
  else if (il(c, *b, *cc))
    iswap(result, cc);
  else
    iswap(result, b);
}
static int* ss_unguarded_partition(int* first, int* last, int* pivot, const idx_less* c) {
  for (;;) {
    while (il(c, *first, *pivot)) ++first;
    --last;
    while (il(c, *pivot, *last)) --last;
    if (!(first < last)) return first;
    iswap(first, last);
    ++first;
  }
}
static void ss_introsort_loop(int* first, int* last, long depth, const idx_less* c) {
  while (last - first > 16) {
    if (depth == 0) {
      ss_partial_sort_all(first, last, c);
      return;
    }
    --depth;
    int* mid = first + (last - first) / 2;
    ss_move_median_to_first(first, first + 1, mid, last - 1, c);
    int* cut = ss_unguarded_partition(first + 1, last, first, c);
    ss_introsort_loop(cut, last, depth, c);
    last = cut;
  }
}
static void ss_unguarded_linear_insert(int* last, const idx_less* c) {
  const int val = *last;
  int* next = last - 1;
  while (il(c, val, *next)) {
    *last = *next;
    last = next;
    --next;
  }
  *last = val;
}
static void ss_insertion_sort(int* first, int* last, const idx_less* c) {
  if (first == last) return;
  for (int* i = first + 1; i != last; ++i) {
    if (il(c, *i, *first)) {
      const int val = *i;
      memmove(first + 1, first, (size_t)(i - first) * sizeof(int));
      *first = val;
    } else
      ss_unguarded_linear_insert(i, c);
  }
}
static void ss_sort(int* first, int* last, const idx_less* c) {
  if (first == last) return;
  long n = last - first, lg = 0;
  while (n > 1) { /* std::__lg */
    n >>= 1;
    lg++;
  }
  ss_introsort_loop(first, last, lg * 2, c);
  if (last - first > 16) {
    ss_insertion_sort(first, first + 16, c);
    for (int* i = first + 16; i != last; ++i) ss_unguarded_linear_insert(i, c);
  } else
    ss_insertion_sort(first, last, c);
}

/* cv: sort.cpp sortIdx_<int>, one row, CV_SORT_DESCENDING -- SURVEY A.5 */
void orc_sort_idx_descending(const int* keys, int n, int* idx) {
  idx_less c = {keys};
  for (int j = 0; j < n; j++) idx[j] = j;
  ss_sort(idx, idx + n, &c);
  for (int j = 0; j < n / 2; j++) iswap(&idx[j], &idx[n - 1 - j]);
}

/* ref: NonMaximumSuppression/anms/anms.cpp:344-438 */
int orc_ssc(const float* kxy, int n, int num_ret, float tolerance, int cols, int rows, int* result) {
  if (num_ret <= 0) return -1;
  const int exp1 = rows + cols + 2 * num_ret; /* :350 */
  const long long exp2 = ((long long)4 * cols + (long long)4 * num_ret + (long long)4 * rows * num_ret +
                          (long long)rows * rows + (long long)cols * cols - (long long)2 * rows * cols +
                          (long long)4 * rows * cols * num_ret);
  const double exp3 = sqrt((double)exp2);
  const double exp4 = num_ret - 1;
  const double sol1 = -round((exp1 + exp3) / exp4); /* :359-360 */
  const double sol2 = -round((exp1 - exp3) / exp4);
  /* :362-365 `int high = sol1 > sol2 ? sol1 : sol2`.  With num_ret == 1 exp4 is 0 and the larger solution +inf: the
   * conversion is undefined in C++; the reference's x86-64 binary (cvttsd2si) gets INT_MIN for anything outside int,
   * which is what is restated here without the undefined behaviour. */
  const double hsol = (sol1 > sol2) ? sol1 : sol2;
  int high = (hsol >= -2147483648.0 && hsol < 2147483648.0) ? (int)hsol : (-2147483647 - 1);
  int low = (int)floor(sqrt((double)n / num_ret)); /* :366 */
  int width, prev_width = -1;
  const unsigned int K = (unsigned int)num_ret;
  /* :374-375  K - (K * tolerance): unsigned * float -> float arithmetic */
  const unsigned int Kmin = (unsigned int)roundf((float)K - ((float)K * tolerance));
  const unsigned int Kmax = (unsigned int)roundf((float)K + ((float)K * tolerance));
  int nres = 0, nfinal = 0;
  int* res = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (;;) {
    /* :380 `low + (high - low) / 2`; with high = INT_MIN (above) the subtraction overflows: undefined in C++, a
     * wrapping subtract in the reference's binary -- restated with unsigned arithmetic.  The width is then about
     * 2^30, every keypoint falls into cell (0, 0), the first one is the result: what the HIP kernel returns too
     * (its conversion saturates to INT_MAX, a width within 1 of this one). */
    width = (int)((unsigned int)low + (unsigned int)((int)((unsigned int)high - (unsigned int)low) / 2));
    if (width == prev_width || low > high) { /* :381-386 previous iteration's result */
      nfinal = nres;
      memcpy(result, res, sizeof(int) * (size_t)nres);
      break;
    }
    nres = 0;
    const double c = (double)width / 2.0; /* :388 */
    const int ncc = (int)floor(cols / c), ncr = (int)floor(rows / c);
    uint8_t* cov = (uint8_t*)calloc((size_t)(ncr + 1) * (ncc + 1), 1);
    const int reach = (int)floor(width / c);
    for (int i = 0; i < n; i++) {
      const int row = (int)floor(kxy[2 * i + 1] / c), col = (int)floor(kxy[2 * i] / c);
      if (!cov[(size_t)row * (ncc + 1) + col]) {
        res[nres++] = i;
        const int r0 = row - reach >= 0 ? row - reach : 0, r1 = row + reach <= ncr ? row + reach : ncr;
        const int c0 = col - reach >= 0 ? col - reach : 0, c1 = col + reach <= ncc ? col + reach : ncc;
        for (int rr = r0; rr <= r1; rr++)
          for (int cc = c0; cc <= c1; cc++) cov[(size_t)rr * (ncc + 1) + cc] = 1;
      }
    }
    free(cov);
    if ((unsigned int)nres >= Kmin && (unsigned int)nres <= Kmax) { /* :423-425 */
      nfinal = nres;
      memcpy(result, res, sizeof(int) * (size_t)nres);
      break;
    } else if ((unsigned int)nres < Kmin)
      high = width - 1;
    else
      low = width + 1;
    prev_width = width;
  }
  free(res);
  return nfinal;
}

/* ref: FeatureDetectorANMS.cpp:13-35 (Parameter defaults) and :37-53 (ctor) */
void orc_anms_detector_init(orc_anms_detector* d, int32_t max_features, double min_distance) {
  d->max_feature_number = max_features;
  d->min_feature_distance = min_distance;
  d->detection_feature_number = max_features * 2;
  d->detection_min_feature_distance = 1.0;
  d->detection_quality_level = 0.001;
  d->detection_block_size = 4;
  d->detection_ignore_edge = 20;
  d->non_max_suppression_tolerance = 0.1;
  d->max_feature_index = 0;
  d->detection_use_harris_corner = 0;
  d->detection_k = 0.04;
}

/* ref: FeatureDetectorANMS.cpp:107-132 */
void orc_anms_build_mask(const orc_anms_detector* d, int w, int h, const uint8_t* mask,
                         int mask_stride, const double* xy, int n, uint8_t* flag) {
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) flag[(size_t)y * w + x] = mask ? mask[(size_t)y * mask_stride + x] : 1;
  const int radius = (int)d->min_feature_distance; /* :127 static_cast<int> */
  for (int i = 0; i < n; i++) {
    const double px = xy[2 * i], py = xy[2 * i + 1];
    if (px > 0 && px < w && py > 0 && py < h) /* :123-124 */
      orc_circle_filled(flag, w, h, w, (int)px, (int)py, radius, 0); /* cv::Point(double,double): truncation */
  }
  /* :107-113 edge_mask: cv::rectangle(Rect(e, e, w-2e, h-2e), 1, FILLED) covers [e, w-e-1] x [e, h-e-1]
   * (nothing when the rectangle is empty); :132 bitwise_and -- BITWISE, so only bit 0 of the mask survives */
  const int e = d->detection_ignore_edge;
  const int ew = w - 2 * e, eh = h - 2 * e;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int em = 0;
      if (ew > 0 && eh > 0) {
        /* cv::rectangle clips to the image; with e < 0 the whole image is covered */
        em = (x >= e && x <= e + ew - 1 && y >= e && y <= e + eh - 1) ? 1 : 0;
      }
      flag[(size_t)y * w + x] &= (uint8_t)em;
    }
}

/* ref: FeatureDetectorANMS.cpp:69-212 */
int orc_anms_detector_update(orc_anms_detector* d, const uint8_t* img, int w, int h, int stride,
                             const uint8_t* mask, int mask_stride, uint64_t* ids, double* xy,
                             uint32_t* age, int* n_inout, int capacity) {
  const int n0 = *n_inout;
  int n = n0, rc = 0;
  uint8_t* flag = (uint8_t*)malloc((size_t)w * h);
  orc_anms_build_mask(d, w, h, mask, mask_stride, xy, n0, flag);
  /* :135-137 GFTTDetector::detect = goodFeaturesToTrack(nfeatures, quality, minDistance, mask, blockSize);
   * KeyPoint(pt, blockSize): response stays 0 (Q9) */
  const int kcap = d->detection_feature_number > 0 ? d->detection_feature_number : w * h;
  float* kp = (float*)malloc(sizeof(float) * 2 * (size_t)kcap);
  const int ndet = orc_good_features_to_track_ex(img, w, h, stride, 0, 0, w, h, d->detection_feature_number,
                                                 d->detection_quality_level, d->detection_min_feature_distance,
                                                 flag, w, d->detection_block_size, d->detection_use_harris_corner,
                                                 d->detection_k, kp, kcap);
  int need = d->max_feature_number - n0; /* :174-175 */
  if (need < 0) need = 0;
  int* sel = (int*)malloc(sizeof(int) * (size_t)(ndet > 0 ? ndet : 1));
  float* sorted = (float*)malloc(sizeof(float) * 2 * (size_t)(ndet > 0 ? ndet : 1));
  int nsel = 0;
  const float* out_pts = kp;
  if (need < ndet) { /* :177-183 suppressNonMax */
    if (need == 0) {
      rc = 1; /* Ssc(numRetPoints = 0): undefined in the reference; nothing is added here */
      nsel = 0;
    } else {
      /* NonMaximumSuppression.cpp:43-53: int-truncated responses (all 0), sortIdx descending */
      int* keys = (int*)calloc((size_t)ndet, sizeof(int));
      int* idx = (int*)malloc(sizeof(int) * (size_t)ndet);
      orc_sort_idx_descending(keys, ndet, idx);
      for (int i = 0; i < ndet; i++) {
        sorted[2 * i] = kp[2 * idx[i]];
        sorted[2 * i + 1] = kp[2 * idx[i] + 1];
      }
      nsel = orc_ssc(sorted, ndet, need, (float)d->non_max_suppression_tolerance, w, h, sel);
      out_pts = sorted;
      free(keys);
      free(idx);
    }
  } else { /* :184-186 */
    nsel = ndet;
    for (int i = 0; i < ndet; i++) sel[i] = i;
  }
  uint64_t feature_index = d->max_feature_index + 1; /* :198 */
  for (int k = 0; k < nsel; k++) {
    int slot = -1;
    for (int i = 0; i < n; i++)
      if (ids[i] == feature_index) {
        slot = i;
        break;
      }
    if (slot < 0) {
      if (n >= capacity) {
        rc = -1;
        break;
      }
      slot = n++;
    }
    ids[slot] = feature_index;
    xy[2 * slot] = (double)out_pts[2 * sel[k]];
    xy[2 * slot + 1] = (double)out_pts[2 * sel[k] + 1];
    age[slot] = 1;
    feature_index++;
  }
  d->max_feature_index = feature_index - 1; /* :211 */
  *n_inout = n;
  free(sorted);
  free(sel);
  free(kp);
  free(flag);
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* LSSD SE(2) patch tracker (FeatureTrackerLSSDLucasKanade, basalt frame-to-frame optical flow) */
/* ------------------------------------------------------------------------------------------ */
/* ref: LSSDOpticalFlow/patterns.h:60-73 (Pattern24::pattern_raw); frame_to_frame_optical_flow_single_thread.hpp:25 */
static const float LSSD_PATTERN[ORC_LSSD_PATTERN][2] = {
    {-1, 5},  {1, 5},  {-3, 3},  {-1, 3},  {1, 3},   {3, 3},  {-5, 1},  {-3, 1},
    {-1, 1},  {1, 1},  {3, 1},   {5, 1},   {-5, -1}, {-3, -1}, {-1, -1}, {1, -1},
    {3, -1},  {5, -1}, {-3, -3}, {-1, -3}, {1, -3},  {3, -3}, {-1, -5}, {1, -5}};

/* basalt::ManagedImagePyr<uint16_t>::subsample: vertical then horizontal [1 4 6 4 1] in int, borders
 * abs() / border101(), (v + 128) >> 8 once; size (w/2, h/2) */
static void lssd_subsample(const orc_level16* s, orc_level16* d) {
  d->w = s->w / 2;
  d->h = s->h / 2;
  d->data = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)(d->w > 0 ? d->w : 1) * (d->h > 0 ? d->h : 1));
  static const int k[5] = {1, 4, 6, 4, 1};
  for (int y = 0; y < d->h; y++)
    for (int x = 0; x < d->w; x++) {
      int acc = 0;
      for (int j = -2; j <= 2; j++) {
        const int yy = reflect101(2 * y + j, s->h);
        int r = 0;
        for (int i = -2; i <= 2; i++) r += k[i + 2] * (int)s->data[(size_t)yy * s->w + reflect101(2 * x + i, s->w)];
        acc += k[j + 2] * r;
      }
      d->data[(size_t)y * d->w + x] = (uint16_t)((acc + (1 << 7)) >> 8);
    }
}

int orc_lssd_build_pyramid(const uint8_t* img, int w, int h, int stride, int levels, orc_pyramid16* out) {
  if (levels < 0 || levels >= ORC_MAX_LEVELS) return -1;
  memset(out, 0, sizeof(*out));
  out->levels = levels;
  out->lv[0].w = w;
  out->lv[0].h = h;
  out->lv[0].data = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) out->lv[0].data[(size_t)y * w + x] = (uint16_t)((int)img[(size_t)y * stride + x] << 8);
  for (int l = 0; l < levels; l++) lssd_subsample(&out->lv[l], &out->lv[l + 1]);
  return 0;
}

void orc_lssd_free_pyramid(orc_pyramid16* p) {
  for (int l = 0; l < ORC_MAX_LEVELS; l++) {
    free(p->lv[l].data);
    p->lv[l].data = NULL;
  }
}

/* basalt::Image::InBounds(p, border) for floating point p */
static inline int lssd_in_bounds(const orc_level16* im, float x, float y, float border) {
  return border <= x && x < ((float)im->w - border - 1.f) && border <= y && y < ((float)im->h - border - 1.f);
}
static inline float lssd_px(const orc_level16* im, int x, int y) { return (float)im->data[(size_t)y * im->w + x]; }
/* basalt::Image::interp<float> */
static inline float lssd_interp(const orc_level16* im, float x, float y) {
  const int ix = (int)x, iy = (int)y;
  const float dx = x - (float)ix, dy = y - (float)iy;
  const float ddx = 1.f - dx, ddy = 1.f - dy;
  return ddx * ddy * lssd_px(im, ix, iy) + ddx * dy * lssd_px(im, ix, iy + 1) + dx * ddy * lssd_px(im, ix + 1, iy) +
         dx * dy * lssd_px(im, ix + 1, iy + 1);
}
/* basalt::Image::interpGrad<float>: value and central differences of the interpolated image */
static inline void lssd_interp_grad(const orc_level16* im, float x, float y, float* out3) {
  const int ix = (int)x, iy = (int)y;
  const float dx = x - (float)ix, dy = y - (float)iy;
  const float ddx = 1.f - dx, ddy = 1.f - dy;
  const float px0y0 = lssd_px(im, ix, iy), px1y0 = lssd_px(im, ix + 1, iy);
  const float px0y1 = lssd_px(im, ix, iy + 1), px1y1 = lssd_px(im, ix + 1, iy + 1);
  out3[0] = ddx * ddy * px0y0 + ddx * dy * px0y1 + dx * ddy * px1y0 + dx * dy * px1y1;
  const float pxm1y0 = lssd_px(im, ix - 1, iy), pxm1y1 = lssd_px(im, ix - 1, iy + 1);
  const float res_mx = ddx * ddy * pxm1y0 + ddx * dy * pxm1y1 + dx * ddy * px0y0 + dx * dy * px0y1;
  const float px2y0 = lssd_px(im, ix + 2, iy), px2y1 = lssd_px(im, ix + 2, iy + 1);
  const float res_px = ddx * ddy * px1y0 + ddx * dy * px1y1 + dx * ddy * px2y0 + dx * dy * px2y1;
  out3[1] = 0.5f * (res_px - res_mx);
  const float px0ym1 = lssd_px(im, ix, iy - 1), px1ym1 = lssd_px(im, ix + 1, iy - 1);
  const float res_my = ddx * ddy * px0ym1 + ddx * dy * px0y0 + dx * ddy * px1ym1 + dx * dy * px1y0;
  const float px0y2 = lssd_px(im, ix, iy + 2), px1y2 = lssd_px(im, ix + 1, iy + 2);
  const float res_py = ddx * ddy * px0y1 + ddx * dy * px0y2 + dx * ddy * px1y1 + dx * dy * px1y2;
  out3[2] = 0.5f * (res_py - res_my);
}

/* sum of v[0..23] in the selected order */
static float lssd_sum(const float* v, int mode) {
  if (mode == ORC_LSSD_SEQ) {
    float s = 0.f;
    for (int i = 0; i < ORC_LSSD_PATTERN; i++) s += v[i];
    return s;
  }
  float a[32], b[32];
  for (int i = 0; i < 32; i++) a[i] = i < ORC_LSSD_PATTERN ? v[i] : 0.f;
  for (int step = 1; step < 32; step <<= 1) {
    for (int i = 0; i < 32; i++) b[i] = a[i] + a[i ^ step];
    memcpy(a, b, sizeof(a));
  }
  return a[0];
}

/* sin / cos for the TREE mode: Taylor polynomials in float (|theta| is far below 1 in this use);
 * the SEQ mode calls libm like Sophus does */
static void lssd_sincos(float t, int mode, float* s, float* c) {
  if (mode == ORC_LSSD_SEQ) {
    *s = sinf(t);
    *c = cosf(t);
    return;
  }
  const float t2 = t * t;
  *s = t * (1.f + t2 * (-1.f / 6.f + t2 * (1.f / 120.f + t2 * (-1.f / 5040.f + t2 * (1.f / 362880.f)))));
  *c = 1.f + t2 * (-0.5f + t2 * (1.f / 24.f + t2 * (-1.f / 720.f + t2 * (1.f / 40320.f + t2 * (-1.f / 3628800.f)))));
}

/* Eigen::LDLT<Matrix3f>::compute + solveInPlace(Identity): lower-triangular, diagonal pivoting
 * (Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked, restated for n = 3) */
static void lssd_ldlt_inverse3(const float H[3][3], float Hinv[3][3]) {
  float m[3][3];
  int tr[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) m[i][j] = H[i][j];
  for (int k = 0; k < 3; k++) {
    int big = k;
    float bv = fabsf(m[k][k]);
    for (int i = k + 1; i < 3; i++)
      if (fabsf(m[i][i]) > bv) {
        bv = fabsf(m[i][i]);
        big = i;
      }
    tr[k] = big;
    if (big != k) { /* symmetric swap of rows/columns k and big, lower triangle only */
      const int s = 3 - big - 1;
      for (int j = 0; j < k; j++) {
        const float t = m[k][j];
        m[k][j] = m[big][j];
        m[big][j] = t;
      }
      for (int i = 0; i < s; i++) {
        const float t = m[big + 1 + i][k];
        m[big + 1 + i][k] = m[big + 1 + i][big];
        m[big + 1 + i][big] = t;
      }
      {
        const float t = m[k][k];
        m[k][k] = m[big][big];
        m[big][big] = t;
      }
      for (int i = k + 1; i < big; i++) {
        const float t = m[i][k];
        m[i][k] = m[big][i];
        m[big][i] = t;
      }
    }
    const int rs = 3 - k - 1;
    if (k > 0) {
      float temp[3] = {0.f, 0.f, 0.f};
      for (int j = 0; j < k; j++) temp[j] = m[j][j] * m[k][j];
      float dot = 0.f;
      for (int j = 0; j < k; j++) dot = (j == 0) ? m[k][j] * temp[j] : dot + m[k][j] * temp[j];
      m[k][k] -= dot;
      for (int i = 0; i < rs; i++) {
        float d2 = 0.f;
        for (int j = 0; j < k; j++) d2 = (j == 0) ? m[k + 1 + i][j] * temp[j] : d2 + m[k + 1 + i][j] * temp[j];
        m[k + 1 + i][k] -= d2;
      }
    }
    const float akk = m[k][k];
    if (rs > 0 && fabsf(akk) > 0.f)
      for (int i = 0; i < rs; i++) m[k + 1 + i][k] /= akk;
  }
  /* solve for the identity: dst = P b; L^-1; D^-1 (pseudo-inverse with Eigen's tolerance); L^-T; P^T */
  float X[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) X[i][j] = (i == j) ? 1.f : 0.f;
  for (int k = 0; k < 3; k++)
    if (tr[k] != k)
      for (int j = 0; j < 3; j++) {
        const float t = X[k][j];
        X[k][j] = X[tr[k]][j];
        X[tr[k]][j] = t;
      }
  for (int c = 0; c < 3; c++) { /* unit lower triangular forward substitution */
    for (int i = 1; i < 3; i++)
      for (int j = 0; j < i; j++) X[i][c] -= m[i][j] * X[j][c];
  }
  const float tol = 1.f / 3.4028234663852886e+38f; /* RealScalar(1) / NumTraits<RealScalar>::highest() */
  for (int i = 0; i < 3; i++)
    for (int c = 0; c < 3; c++) {
      if (fabsf(m[i][i]) > tol)
        X[i][c] /= m[i][i];
      else
        X[i][c] = 0.f;
    }
  for (int c = 0; c < 3; c++) { /* unit upper triangular (L^T) back substitution */
    for (int i = 1; i >= 0; i--)
      for (int j = i + 1; j < 3; j++) X[i][c] -= m[j][i] * X[j][c];
  }
  for (int k = 2; k >= 0; k--)
    if (tr[k] != k)
      for (int j = 0; j < 3; j++) {
        const float t = X[k][j];
        X[k][j] = X[tr[k]][j];
        X[tr[k]][j] = t;
      }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Hinv[i][j] = X[i][j];
}

typedef struct {
  float data[ORC_LSSD_PATTERN]; /* negative if the point is not valid */
  float HinvJT[3][ORC_LSSD_PATTERN];
} lssd_patch;

/* ref: patch.h:75-131 OpticalFlowPatch::setFromImage */
static void lssd_patch_set(lssd_patch* p, const orc_level16* img, float posx, float posy, int mode) {
  float val[ORC_LSSD_PATTERN], gx[ORC_LSSD_PATTERN], gy[ORC_LSSD_PATTERN];
  int num_valid = 0;
  for (int i = 0; i < ORC_LSSD_PATTERN; i++) {
    const float x = posx + LSSD_PATTERN[i][0], y = posy + LSSD_PATTERN[i][1];
    if (lssd_in_bounds(img, x, y, 2.f)) {
      float vg[3];
      lssd_interp_grad(img, x, y, vg);
      p->data[i] = vg[0];
      val[i] = vg[0];
      gx[i] = vg[1];
      gy[i] = vg[2];
      num_valid++;
    } else {
      p->data[i] = -1.f;
      val[i] = 0.f;
      gx[i] = 0.f;
      gy[i] = 0.f;
    }
  }
  const float sum = lssd_sum(val, mode), gsx = lssd_sum(gx, mode), gsy = lssd_sum(gy, mode);
  const float mean_inv = (float)num_valid / sum; /* :100 */
  float J[ORC_LSSD_PATTERN][3];
  for (int i = 0; i < ORC_LSSD_PATTERN; i++) {
    float jx = 0.f, jy = 0.f;
    if (p->data[i] >= 0.f) {
      const float d = p->data[i];
      /* :111-112 num_valid_points * (grad_i * sum - grad_sum * data_i) / (sum * sum) */
      jx = ((float)num_valid * (gx[i] * sum - gsx * d)) / (sum * sum);
      jy = ((float)num_valid * (gy[i] * sum - gsy * d)) / (sum * sum);
      p->data[i] = d * mean_inv;
    }
    /* :120-122 J_se2.row(i) = grad.row(i) * Jw_se2, Jw = [I | (-py, px)^T] */
    J[i][0] = jx;
    J[i][1] = jy;
    J[i][2] = jx * (-LSSD_PATTERN[i][1]) + jy * LSSD_PATTERN[i][0];
  }
  float H[3][3], Hinv[3][3], prod[ORC_LSSD_PATTERN];
  for (int a = 0; a < 3; a++)
    for (int b = a; b < 3; b++) {
      for (int i = 0; i < ORC_LSSD_PATTERN; i++) prod[i] = J[i][a] * J[i][b];
      H[a][b] = H[b][a] = lssd_sum(prod, mode);
    }
  lssd_ldlt_inverse3(H, Hinv); /* :125-128 */
  for (int a = 0; a < 3; a++)  /* :130 */
    for (int i = 0; i < ORC_LSSD_PATTERN; i++)
      p->HinvJT[a][i] = (Hinv[a][0] * J[i][0] + Hinv[a][1] * J[i][1]) + Hinv[a][2] * J[i][2];
}

/* ref: patch.h:133-164 OpticalFlowPatch::residual */
static int lssd_patch_residual(const lssd_patch* p, const orc_level16* img, const float tp[ORC_LSSD_PATTERN][2],
                               float* res, int mode) {
  float val[ORC_LSSD_PATTERN];
  int num_valid = 0;
  for (int i = 0; i < ORC_LSSD_PATTERN; i++) {
    if (lssd_in_bounds(img, tp[i][0], tp[i][1], 2.f)) {
      res[i] = lssd_interp(img, tp[i][0], tp[i][1]);
      val[i] = res[i];
      num_valid++;
    } else {
      res[i] = -1.f;
      val[i] = 0.f;
    }
  }
  const float sum = lssd_sum(val, mode);
  int num_res = 0;
  for (int i = 0; i < ORC_LSSD_PATTERN; i++) {
    if (res[i] >= 0.f && p->data[i] >= 0.f) {
      res[i] = ((float)num_valid * res[i]) / sum - p->data[i];
      num_res++;
    } else {
      res[i] = 0.f;
    }
  }
  return num_res > ORC_LSSD_PATTERN / 2;
}

/* ref: frame_to_frame_optical_flow_single_thread.cpp:114-146 TrackPointAtLevel.
 * T = [L | t] (AffineCompact2f): L row-major 2x2, t */
static int lssd_track_at_level(const orc_level16* img2, const lssd_patch* dp, float L[2][2], float t[2],
                               int max_iterations, int mode) {
  int patch_valid = 1;
  for (int it = 0; patch_valid && it < max_iterations; it++) {
    float tp[ORC_LSSD_PATTERN][2], res[ORC_LSSD_PATTERN];
    for (int i = 0; i < ORC_LSSD_PATTERN; i++) { /* :126-128 */
      tp[i][0] = (L[0][0] * LSSD_PATTERN[i][0] + L[0][1] * LSSD_PATTERN[i][1]) + t[0];
      tp[i][1] = (L[1][0] * LSSD_PATTERN[i][0] + L[1][1] * LSSD_PATTERN[i][1]) + t[1];
    }
    if (lssd_patch_residual(dp, img2, tp, res, mode)) {
      float inc[3], prod[ORC_LSSD_PATTERN];
      for (int a = 0; a < 3; a++) { /* :133 inc = -H_se2_inv_J_se2_T * res */
        for (int i = 0; i < ORC_LSSD_PATTERN; i++) prod[i] = (-dp->HinvJT[a][i]) * res[i];
        inc[a] = lssd_sum(prod, mode);
      }
      /* Sophus::SE2f::exp(inc) */
      const float theta = inc[2];
      float sn, cs;
      lssd_sincos(theta, mode, &sn, &cs);
      float sbt, omcbt;
      if (fabsf(theta) < 1e-5f) { /* Sophus::Constants<float>::epsilon() */
        const float theta_sq = theta * theta;
        sbt = 1.f - (1.f / 6.f) * theta_sq;
        omcbt = 0.5f * theta - (1.f / 24.f) * theta * theta_sq;
      } else {
        sbt = sn / theta;
        omcbt = (1.f - cs) / theta;
      }
      const float ex = sbt * inc[0] - omcbt * inc[1];
      const float ey = omcbt * inc[0] + sbt * inc[1];
      /* :134 transform *= SE2::exp(inc).matrix(): (2x3) * (3x3), last row of the right factor (0 0 1) */
      const float R[3][3] = {{cs, -sn, ex}, {sn, cs, ey}, {0.f, 0.f, 1.f}};
      float nL[2][2], nt[2];
      for (int r = 0; r < 2; r++) {
        for (int c = 0; c < 2; c++) nL[r][c] = (L[r][0] * R[0][c] + L[r][1] * R[1][c]) + t[r] * R[2][c];
        nt[r] = (L[r][0] * R[0][2] + L[r][1] * R[1][2]) + t[r] * R[2][2];
      }
      memcpy(L, nL, sizeof(nL));
      t[0] = nt[0];
      t[1] = nt[1];
      if (!lssd_in_bounds(img2, t[0], t[1], 2.f)) patch_valid = 0; /* :136-139 filter_margin */
    } else {
      patch_valid = 0;
    }
  }
  return patch_valid;
}

/* ref: frame_to_frame_optical_flow_single_thread.cpp:84-112 TrackPoint */
int orc_lssd_track_point(const orc_pyramid16* old_pyr, const orc_pyramid16* pyr, const float* xy_in,
                         float* xy_out, int levels, int max_iterations, int mode) {
  float L[2][2] = {{1.f, 0.f}, {0.f, 1.f}};
  float t[2] = {xy_out[0], xy_out[1]}; /* transform starts as the caller's initial guess */
  int patch_valid = 1;
  for (int level = levels; level >= 0 && patch_valid; level--) {
    const float scale = (float)(1 << level);
    t[0] /= scale;
    t[1] /= scale;
    lssd_patch p;
    lssd_patch_set(&p, &old_pyr->lv[level], xy_in[0] / scale, xy_in[1] / scale, mode);
    patch_valid &= lssd_track_at_level(&pyr->lv[level], &p, L, t, max_iterations, mode);
    t[0] *= scale;
    t[1] *= scale;
  }
  xy_out[0] = t[0];
  xy_out[1] = t[1];
  return patch_valid;
}

void orc_lssd_tracker_init(orc_lssd_tracker* t, double max_recovered_dist, int32_t levels,
                           int32_t max_iterations, int mode) {
  memset(t, 0, sizeof(*t));
  t->optical_flow_max_recovered_dist = max_recovered_dist;
  t->optical_flow_levels = levels;
  t->optical_flow_max_iterations = max_iterations;
  t->mode = mode;
}

void orc_lssd_tracker_free(orc_lssd_tracker* t) {
  if (t->has_old) orc_lssd_free_pyramid(&t->old_pyr);
  t->has_old = 0;
  free(t->map_ids);
  free(t->map_xy);
  t->map_ids = NULL;
  t->map_xy = NULL;
  t->map_n = t->map_cap = 0;
}

/* transform_ <- the database as it stands (ids, float(position)) */
static void lssd_map_store(orc_lssd_tracker* t, const uint64_t* ids, const double* xy, int n) {
  if (n > t->map_cap) {
    t->map_cap = n + 64;
    t->map_ids = (uint64_t*)realloc(t->map_ids, sizeof(uint64_t) * (size_t)t->map_cap);
    t->map_xy = (float*)realloc(t->map_xy, sizeof(float) * 2 * (size_t)t->map_cap);
  }
  for (int i = 0; i < n; i++) {
    t->map_ids[i] = ids[i];
    t->map_xy[2 * i] = (float)xy[2 * i];
    t->map_xy[2 * i + 1] = (float)xy[2 * i + 1];
  }
  t->map_n = n;
}

/* ref: FeatureTrackerLSSDLucasKanade.cpp:35-119 + TrackPoints (frame_to_frame_..._single_thread.cpp:11-82).
 * transform_ holds one AffineCompact2f per id.  Its linear part is reset to identity at the start of
 * each TrackPoint (:93) and never read.  Its translation is float(position) when the id enters the
 * map (:61-66 first call, :73-80 later) and is NOT refreshed from the database afterwards: an id that
 * survived the previous Track is tracked from the translation stored then, even if the caller has
 * rewritten its position since (FeatureDetectorShiTomasiBucketing does, to the feature whose id
 * equals its counter, Q6).  Ids absent from the database are erased (:82-90), so after each call the
 * map is the database (ids, float positions). */
int orc_lssd_tracker_track(orc_lssd_tracker* t, const uint8_t* cur, int w, int h, int stride,
                           uint64_t* ids, double* xy, uint32_t* age, int* n_inout) {
  orc_pyramid16 pyr;
  if (orc_lssd_build_pyramid(cur, w, h, stride, t->optical_flow_levels, &pyr)) return -1;
  const int n0 = *n_inout;
  if (!t->has_old) { /* :57-68 */
    t->old_pyr = pyr;
    t->has_old = 1;
    lssd_map_store(t, ids, xy, n0);
    return 0;
  }
  int n = 0;
  for (int i = 0; i < n0; i++) {
    float p1[2] = {(float)xy[2 * i], (float)xy[2 * i + 1]}; /* :73-80 new to the map */
    for (int j = 0; j < t->map_n; j++)
      if (t->map_ids[j] == ids[i]) { /* already in the map: stored translation */
        p1[0] = t->map_xy[2 * j];
        p1[1] = t->map_xy[2 * j + 1];
        break;
      }
    float p2[2] = {p1[0], p1[1]};
    int valid = orc_lssd_track_point(&t->old_pyr, &pyr, p1, p2, t->optical_flow_levels,
                                     t->optical_flow_max_iterations, t->mode);
    if (valid) {
      float rec[2] = {p2[0], p2[1]};
      valid = orc_lssd_track_point(&pyr, &t->old_pyr, p2, rec, t->optical_flow_levels,
                                   t->optical_flow_max_iterations, t->mode);
      if (valid) {
        const float ddx = p1[0] - rec[0], ddy = p1[1] - rec[1];
        const float dist2 = ddx * ddx + ddy * ddy;
        valid = (double)dist2 < t->optical_flow_max_recovered_dist; /* :67 */
      }
    }
    if (valid) {
      ids[n] = ids[i];
      xy[2 * n] = (double)p2[0]; /* :108 translation().cast<double>() */
      xy[2 * n + 1] = (double)p2[1];
      age[n] = age[i];
      n++;
    }
  }
  *n_inout = n;
  lssd_map_store(t, ids, xy, n); /* :102 transform_ = new_transform */
  orc_lssd_free_pyramid(&t->old_pyr);
  t->old_pyr = pyr;
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* frontend step: bearings, aged count, keyframe decision                                      */
/* ------------------------------------------------------------------------------------------ */
int orc_ds_unproject(const orc_camera_ds* cam, double u, double v, double* out3) {
  const double mx = (u - cam->cx) / cam->fx;
  const double my = (v - cam->cy) / cam->fy;
  const double r2 = mx * mx + my * my;
  if (cam->alpha > 0.5 && r2 >= 1.0 / (2.0 * cam->alpha - 1.0)) return 0;
  const double xi2_2 = cam->alpha * cam->alpha;
  const double xi1_2 = cam->xi * cam->xi;
  const double sqrt2 = sqrt(1.0 - (2.0 * cam->alpha - 1.0) * r2);
  const double norm2 = cam->alpha * sqrt2 + 1.0 - cam->alpha;
  const double mz = (1.0 - xi2_2 * r2) / norm2;
  const double mz2 = mz * mz;
  const double norm1 = mz2 + r2;
  const double sqrt1 = sqrt(mz2 + (1.0 - xi1_2) * r2);
  const double k = (mz * cam->xi + sqrt1) / norm1;
  out3[0] = k * mx;
  out3[1] = k * my;
  out3[2] = k * mz - cam->xi;
  return 1;
}

int orc_frontend_post_track(const orc_camera_ds* cam, const double* xy, uint32_t* age, int n,
                            int32_t counting_feature_age, uint32_t low_keyframe_feature_number,
                            double minimum_keyframe_interval, double keyframe_interval,
                            double dt_since_keyframe, double* bearings, uint32_t* aged_out) {
  for (int i = 0; i < n; i++) { /* :247-257 */
    double* b = bearings + 3 * i;
    if (!orc_ds_unproject(cam, xy[2 * i], xy[2 * i + 1], b)) {
      b[0] = 0;
      b[1] = 0;
      b[2] = 1;
    }
  }
  uint32_t aged = 0; /* :266-269: uint32 age against int32 parameter -> unsigned comparison */
  for (int i = 0; i < n; i++)
    if (age[i] >= (uint32_t)counting_feature_age) aged++;
  int low = aged < low_keyframe_feature_number;                 /* :275-276 */
  if (dt_since_keyframe < minimum_keyframe_interval) low = 0;   /* :278-281 */
  const int keyframe = low || (dt_since_keyframe > keyframe_interval); /* :291-293 */
  if (keyframe)
    for (int i = 0; i < n; i++) age[i]++; /* :302-304 */
  if (aged_out) *aged_out = aged;
  return keyframe;
}

/* ------------------------------------------------------------------------------------------ */
/* reference wrapper: FeatureTrackerLucasKanade                                                */
/* ------------------------------------------------------------------------------------------ */
/* ref: src/vslam/feature/FeatureTrackerLucasKanade.cpp:8-23 */
void orc_tracker_init(orc_tracker* t, int32_t max_iter, double eps, int32_t win, int32_t max_level,
                      double backtrack_thr, int acc_mode) {
  memset(t, 0, sizeof(*t));
  t->klt_max_iteration = max_iter;
  t->klt_epsilon = eps;
  t->klt_window_size = win;
  t->klt_max_level = max_level;
  t->backtrack_distance_threshold = backtrack_thr;
  t->acc_mode = acc_mode;
}

void orc_tracker_free(orc_tracker* t) {
  free(t->old_image);
  t->old_image = NULL;
}

/* ref: FeatureTrackerLucasKanade.cpp:187-240 (TrackByLucasKanade): flatten, calcOpticalFlowPyrLK,
 * keep status==1.  Works on compacted SoA arrays in place; returns the kept count.
 * `keep_idx[k]` = index into the input arrays of the k-th kept feature. */
static int track_by_lk(const orc_tracker* t, const uint8_t* prev, int prev_stride,
                       const uint8_t* cur, int cur_stride, int w, int h, const double* xy_in, int n,
                       double* xy_out, int* keep_idx) {
  if (n == 0) return 0; /* :195-198 empty inputs: untouched */
  float* pp = (float*)malloc(sizeof(float) * 2 * (size_t)n * 2);
  float* np_ = pp + 2 * (size_t)n;
  uint8_t* st = (uint8_t*)malloc((size_t)n);
  float* er = (float*)malloc(sizeof(float) * (size_t)n);
  for (int i = 0; i < n; i++) { /* :208-211 double -> float narrowing */
    pp[2 * i] = (float)xy_in[2 * i];
    pp[2 * i + 1] = (float)xy_in[2 * i + 1];
    np_[2 * i] = np_[2 * i + 1] = 0.f;
  }
  orc_calc_optical_flow_pyr_lk(prev, prev_stride, cur, cur_stride, w, h, pp, np_, st, er, n,
                               t->klt_window_size, t->klt_max_level, t->klt_max_iteration,
                               t->klt_epsilon, 1e-4, t->acc_mode); /* :216-224 */
  int m = 0;
  for (int i = 0; i < n; i++)
    if (st[i] == 1) { /* :229-235 */
      xy_out[2 * m] = (double)np_[2 * i];
      xy_out[2 * m + 1] = (double)np_[2 * i + 1];
      keep_idx[m] = i;
      m++;
    }
  free(er);
  free(st);
  free(pp);
  return m;
}

/* ref: FeatureTrackerLucasKanade.cpp:109-185 (4-arg Track) */
int orc_tracker_track_pair(const orc_tracker* t, const uint8_t* prev, int prev_stride,
                           const uint8_t* cur, int cur_stride, int w, int h, uint64_t* ids,
                           double* xy, uint32_t* age, int* n_inout, orc_track_stats* stats) {
  const int n = *n_inout;
  orc_track_stats s = {n, 0, 0, 0};
  double* fwd = (double*)malloc(sizeof(double) * 2 * (size_t)(n + 1) * 2);
  double* back = fwd + 2 * (size_t)(n + 1);
  int* fidx = (int*)malloc(sizeof(int) * (size_t)(n + 1) * 2);
  int* bidx = fidx + (n + 1);
  const int nf = track_by_lk(t, prev, prev_stride, cur, cur_stride, w, h, xy, n, fwd, fidx);
  s.forward = nf;
  const int nb = track_by_lk(t, cur, cur_stride, prev, prev_stride, w, h, fwd, nf, back, bidx);
  s.backward = nb;
  int m = 0;
  for (int k = 0; k < nb; k++) { /* :155-160 */
    const int f = bidx[k];  /* index into forward arrays */
    const int o = fidx[f];  /* index into the original arrays */
    const double ddx = back[2 * k] - xy[2 * o], ddy = back[2 * k + 1] - xy[2 * o + 1];
    if (sqrt(ddx * ddx + ddy * ddy) <= t->backtrack_distance_threshold) {
      /* o is increasing in k, and m <= o, so the in-place compaction is safe */
      ids[m] = ids[o];
      age[m] = age[o];
      fidx[m] = f;
      m++;
    }
  }
  for (int k = 0; k < m; k++) {
    xy[2 * k] = fwd[2 * fidx[k]];
    xy[2 * k + 1] = fwd[2 * fidx[k] + 1];
  }
  s.verified = m;
  *n_inout = m;
  if (stats) *stats = s;
  free(fidx);
  free(fwd);
  return 0;
}

/* ref: FeatureTrackerLucasKanade.cpp:25-107 (3-arg stateful Track).  The reference keeps a
 * shallow reference to the caller's frame; this copy is value-equivalent. */
int orc_tracker_track(orc_tracker* t, const uint8_t* cur, int w, int h, int stride, uint64_t* ids,
                      double* xy, uint32_t* age, int* n_inout, orc_track_stats* stats) {
  int rc = 0;
  if (t->old_image == NULL) { /* :29-32 */
    if (stats) {
      stats->input = *n_inout;
      stats->forward = stats->backward = stats->verified = *n_inout;
    }
  } else {
    if (t->old_w != w || t->old_h != h) return -3;
    rc = orc_tracker_track_pair(t, t->old_image, w, cur, stride, w, h, ids, xy, age, n_inout,
                                stats);
  }
  if (t->old_image == NULL || t->old_w != w || t->old_h != h) {
    free(t->old_image);
    t->old_image = (uint8_t*)malloc((size_t)w * h);
    t->old_w = w;
    t->old_h = h;
  }
  for (int y = 0; y < h; y++) memcpy(t->old_image + (size_t)y * w, cur + (size_t)y * stride, w);
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* timed CPU baseline: one frontend step (Track, then UpdateDetection) on independent streams   */
/* ------------------------------------------------------------------------------------------ */
int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* threads <= 0: the OpenMP default (all cores) */
void orc_set_threads(int threads) {
#ifdef _OPENMP
  static int dflt = 0;
  if (!dflt) dflt = omp_get_max_threads();
  omp_set_num_threads(threads > 0 ? threads : dflt);
#else
  (void)threads;
#endif
}

int orc_step_streams(orc_tracker* trackers, orc_detector* detectors, int streams,
                     const uint8_t* frames, long frame_stride_bytes, int w, int h, int stride,
                     uint64_t* ids, double* xy, uint32_t* age, int* n, int capacity) {
  int rc_all = 0;
  /* one stream: no team here, so that the row / point loops inside get the cores (single-stream latency) */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) if (streams > 1)
#endif
  for (int s = 0; s < streams; s++) {
    const uint8_t* f = frames + (size_t)s * frame_stride_bytes;
    uint64_t* sid = ids + (size_t)s * capacity;
    double* sxy = xy + 2 * (size_t)s * capacity;
    uint32_t* sage = age + (size_t)s * capacity;
    int rc = orc_tracker_track(&trackers[s], f, w, h, stride, sid, sxy, sage, &n[s], NULL);
    if (!rc)
      rc = orc_detector_update(&detectors[s], f, w, h, stride, NULL, 0, sid, sxy, sage, &n[s],
                               capacity);
    if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
      rc_all = rc;
    }
  }
  return rc_all;
}

/* test hook: the 3x3 LDLT inverse used by the LSSD patch */
/* test hook: the sin / cos of SE2::exp in either mode */
void orc_lssd_debug_sincos(float theta, int mode, float* s, float* c) { lssd_sincos(theta, mode, s, c); }

void orc_lssd_debug_inverse3(const float* H9, float* Hinv9) {
  float H[3][3], Hi[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) H[i][j] = H9[3 * i + j];
  lssd_ldlt_inverse3(H, Hi);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Hinv9[3 * i + j] = Hi[i][j];
}
