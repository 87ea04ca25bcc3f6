// vsf_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the vision-frontend hot path.
//
// Algorithms follow SURVEY.md Appendix A (OpenCV 3.4 semantics the reference relies on) and the
// wrapper logic of the reference (file:line citations at each kernel).  Integer stages are exact;
// float stages are written so that, compiled with -ffp-contract=off, every expression is evaluated
// in the order the CPU oracle uses (oracle/vsf_oracle.c), which makes detector results bit-exact.
//
// Layout in HBM (per context, B = streams):
//   frames      : caller's 8-bit frames, frame s at base + s*frame_stride, rows row_stride apart
//   pyramid     : levels >= 1, level l as [B][h_l][w_l] u8 (pitch = w_l), two sets (prev / cur)
//   table       : ids [B][cap] u64, xy [B][cap] double2, age [B][cap] u32, n [B] i32
//   lambda map  : per stream, per grid cell, min-eigenvalue map f32 [cell_h][cell_w]
//   candidates  : per stream, per cell, up to cand_cap 64-bit keys (lambda bits << 32 | raster idx)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#ifndef VSF_LK_WAVES
#define VSF_LK_WAVES 7  // launch bound of the LK kernels (window <= 21): 72 registers, 4608 bytes of LDS: seven waves per SIMD (eight run at a lower clock and spill more: 2.77 against 2.61 ms)
#endif
#ifndef VSF_SUM_LDS
#define VSF_SUM_LDS 1  // exact sums through the LDS atomic unit (0: DPP + v_readlane only)
#endif

// Instrumented build only (-DVSF_LK_STATS, scratch/lk_stats.py): per-level counters of the LK kernel.
//   [level*16 + 0] levels entered in range   [+1] levels that passed the min-eigenvalue test   [+2] iterations
//   [+3] J region stagings   [+4] interior (register Scharr) set-ups   [+5] ended by |delta| <= eps
//   [+6] ended by the oscillation rule   [+7] ended by max_count   [+8] left the image
//   [128 + level*32 + min(iterations, 31)] histogram of iterations per level
#ifdef VSF_LK_STATS
__device__ unsigned long long g_lk_stats[512];
#define VSF_LKSTAT(i)                                                        \
  do {                                                                       \
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_lk_stats[(i)], 1ull);          \
  } while (0)
#else
#define VSF_LKSTAT(i) \
  do {                \
  } while (0)
#endif

namespace vsf {

constexpr int MAXL = 8;      // pyramid levels supported (klt_max_level <= 7)
constexpr int MAX_WIN = 31;  // LK window side supported
constexpr int W_BITS = 14;

// One level = one 32-byte record: a kernel that walks the levels fetches a level's description with ONE scalar load
// (as five arrays indexed by the level it took five loads and five address computations per level and pyramid).
struct PyrLevel {
  const uint8_t* base;  // level l of stream 0
  long long sstride;    // bytes between consecutive streams at level l
  int pitch, w, h;
  int aligned;          // base, stream stride and pitch are multiples of 4: regions can be fetched as dwords
};
struct PyrView {
  PyrLevel lv[MAXL];
  int nlevels;          // effective maxLevel + 1
};

struct LkParams {
  int win;        // window side
  int nseg;       // segments per window row (64 / win)
  int seglen;     // pixels per lane
  int nseg_inv;   // ceil(2^16 / nseg): (lane * nseg_inv) >> 16 == lane / nseg for lane < 64
  int max_count;  // clamped iteration count
  double eps2;    // clamped eps, squared
  float eps2_lo;  // float s = dx^2 + dy^2 <= eps2_lo  =>  the double test is true
  float eps2_hi;  // s >= eps2_hi  =>  the double test is false; in between the double test runs
  double min_eig_thr;
  float min_eig_c1;  // max(threshold, 0) * 2 win^2 * 1.001: see the shortcut in lk_track_point
  double backtrack_thr;
  int jmargin;    // extra margin staged around the J window
};

// cv: borderInterpolate(p, len, BORDER_REFLECT_101), one reflection, then clamped for memory safety
__device__ __forceinline__ int reflect101(int p, int len) {
  if (p < 0) p = -p;
  if (p >= len) p = 2 * len - 2 - p;
  return min(max(p, 0), len - 1);
}

// ------------------------------------------------------------------------------------------------
// Pyramid: cv::pyrDown, 8-bit, [1 4 6 4 1]^2, (v+128)>>8, REFLECT_101.
// One block = 64x16 output pixels of one stream.  The (131 x 35) source tile is staged in LDS with
// coalesced row reads, filtered horizontally into LDS (int16), then vertically.
// ------------------------------------------------------------------------------------------------
constexpr int PD_TW = 64, PD_TH = 16;
constexpr int PD_SW = 2 * PD_TW + 3, PD_SH = 2 * PD_TH + 3;
constexpr int PD_SP = 136;  // LDS pitch of the source tile

__global__ __launch_bounds__(256) void pyr_down_kernel(const uint8_t* __restrict__ src, long long src_sstride,
                                                       int src_pitch, int sw, int sh,
                                                       uint8_t* __restrict__ dst, long long dst_sstride,
                                                       int dst_pitch, int dw, int dh) {
  __shared__ uint8_t s_src[PD_SH * PD_SP];
  __shared__ uint16_t s_h[PD_SH * PD_TW];
  const int stream = blockIdx.z;
  const uint8_t* S = src + (long long)stream * src_sstride;
  uint8_t* D = dst + (long long)stream * dst_sstride;
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
    const uint8_t* r = s_src + y * PD_SP + 2 * x;
    s_h[i] = (uint16_t)(r[0] + 4 * r[1] + 6 * r[2] + 4 * r[3] + r[4]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PD_TH * PD_TW; i += 256) {
    const int y = i / PD_TW, x = i - y * PD_TW;
    const int gx = ox + x, gy = oy + y;
    if (gx < dw && gy < dh) {
      const uint16_t* c = s_h + (2 * y) * PD_TW + x;
      const int v = c[0] + 4 * c[PD_TW] + 6 * c[2 * PD_TW] + 4 * c[3 * PD_TW] + c[4 * PD_TW];
      D[(long long)gy * dst_pitch + gx] = (uint8_t)((v + 128) >> 8);
    }
  }
}

// Dword variant of pyr_down_kernel for 4-byte aligned sources (base, pitch, stream stride and width all
// multiples of 4; the host checks): same arithmetic, a quarter of the memory instructions.
//   stage  : the 35 x 136-byte source tile starting at column 2*ox - 4 is read as 34 dwords per row; rows are
//            reflected by index, the (at most 2 + 2) columns that fall outside the image are patched from
//            their REFLECT_101 partners, which the tile already holds;
//   x pass : a thread makes two adjacent outputs from three dwords and stores them as one dword of two u16
//            (a [1 4 6 4 1] sum of bytes is <= 4080);
//   y pass : a thread makes four adjacent outputs; the two u16 halves of a dword are filtered together as one
//            32-bit integer (the sum is <= 65280 + 128, no carry into the upper half) and written as one dword.
constexpr int PD_SD = 34;  // dwords per staged row

// OUT16: the destination is a dense u16 level holding the UNROUNDED double sum.  That is level 1 of basalt's
// u16 mip pyramid of the frame read as v << 8 (vsf_lssd.hpp): (sum * 256 + 128) >> 8 == sum.  dst_sstride and
// dst_pitch are then in u16 elements.
// A block walks `tiles_y` vertically adjacent tiles and keeps the NEXT tile's source dwords in flight (five
// registers per thread) while it filters the current one: the kernel is HBM-bound, and with load, filter and
// store phases strictly in turn a block had bytes in flight for well under half of its life (level 1 of 1024
// 752x480 frames: 2.1 TB/s).
template <bool OUT16>
__global__ __launch_bounds__(256) void pyr_down_dword_kernel(const uint8_t* __restrict__ src, long long src_sstride,
                                                             int src_pitch, int sw, int sh,
                                                             void* __restrict__ dst, long long dst_sstride,
                                                             int dst_pitch, int dw, int dh, int tiles_y) {
  __shared__ uint32_t s_src[PD_SH * PD_SD];
  __shared__ uint32_t s_h[PD_SH * (PD_TW / 2 + 2)];
  constexpr int HP = PD_TW / 2 + 2;  // pitch of s_h in dwords (even: 8-byte reads stay aligned)
  constexpr int NR = (PD_SH * PD_SD + 255) / 256;  // staged dwords per thread
  const int stream = blockIdx.z;
  const uint8_t* S = src + (long long)stream * src_sstride;
  const int ox = blockIdx.x * PD_TW;
  const int gx_base = 2 * ox - 4;
  // this thread's dwords of a tile: fixed (row, column) slots, only the tile's first row changes
  int sl_y[NR], sl_x[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const int i = threadIdx.x + 256 * k;
    const int y = i / PD_SD, j = i - y * PD_SD;
    const int gx = gx_base + 4 * j;
    sl_y[k] = (i < PD_SH * PD_SD) ? y : -1;
    sl_x[k] = (gx >= 0 && gx + 4 <= sw) ? gx : -1;
  }
  uint32_t nxt[NR];
  auto fetch = [&](int oy) {
    const int sy0 = 2 * oy - 2;
#pragma unroll
    for (int k = 0; k < NR; k++) {
      uint32_t v = 0;
      if (sl_y[k] >= 0 && sl_x[k] >= 0) {
        int gy = reflect101(sy0 + sl_y[k], sh);
        gy = min(max(gy, 0), sh - 1);  // rows that only feed outputs below the image
        v = *reinterpret_cast<const uint32_t*>(S + (long long)gy * src_pitch + sl_x[k]);
      }
      nxt[k] = v;
    }
  };
  const int ty0 = blockIdx.y * tiles_y;
  if (ty0 * PD_TH >= dh) return;
  fetch(ty0 * PD_TH);
  for (int t = 0; t < tiles_y; t++) {
    const int oy = (ty0 + t) * PD_TH;
    if (oy >= dh) break;  // block-uniform
#pragma unroll
    for (int k = 0; k < NR; k++)
      if (sl_y[k] >= 0) s_src[threadIdx.x + 256 * k] = nxt[k];
    __syncthreads();
    if (t + 1 < tiles_y && oy + PD_TH < dh) fetch(oy + PD_TH);  // in flight during the two passes below
    if (gx_base + 2 < 0 || gx_base + PD_SW + 2 > sw) {  // block-uniform: the tile crosses the left / right image edge
      uint8_t* sb = reinterpret_cast<uint8_t*>(s_src);
      for (int i = threadIdx.x; i < PD_SH * 8; i += 256) {
        const int y = i >> 3, k = i & 7;
        // candidates: tile columns 2, 3 (gx = 2ox-2, 2ox-1) on the left, and the first four past sw on the right
        const int tc = k < 2 ? 2 + k : (sw - gx_base) + (k - 2);
        const int gx = gx_base + tc;
        if (tc >= 2 && tc < 2 + PD_SW && (gx < 0 || gx >= sw) && k < 6) {
          const int tr = min(max(reflect101(gx, sw) - gx_base, 0), 4 * PD_SD - 1);
          sb[y * 4 * PD_SD + tc] = sb[y * 4 * PD_SD + tr];
        }
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < PD_SH * (PD_TW / 2); i += 256) {
      const int y = i >> 5, p = i & 31;  // outputs 2p, 2p+1 of row y: tile columns 4p+2 .. 4p+8
      const uint32_t* r = s_src + y * PD_SD + p;
      const uint32_t a = r[0], b = r[1], c = r[2];
      const uint32_t b0 = a >> 16 & 255, b1 = a >> 24, b2 = b & 255, b3 = b >> 8 & 255, b4 = b >> 16 & 255, b5 = b >> 24,
                     b6 = c & 255;
      const uint32_t h0 = b0 + 4 * b1 + 6 * b2 + 4 * b3 + b4, h1 = b2 + 4 * b3 + 6 * b4 + 4 * b5 + b6;
      s_h[y * HP + p] = h0 | (h1 << 16);
    }
    __syncthreads();  // also: every read of s_src is done before the next tile overwrites it
    {
      const int y = threadIdx.x >> 4, q = threadIdx.x & 15;  // outputs 4q .. 4q+3 of row y
      const int gx = ox + 4 * q, gy = oy + y;
      if (gx < dw && gy < dh) {
        const uint32_t* c = s_h + (2 * y) * HP + 2 * q;
        uint32_t lo = c[0] + c[4 * HP] + 4 * (c[HP] + c[3 * HP]) + 6 * c[2 * HP];
        uint32_t hi = c[1] + c[4 * HP + 1] + 4 * (c[HP + 1] + c[3 * HP + 1]) + 6 * c[2 * HP + 1];
        if (OUT16) {
          uint16_t* D = reinterpret_cast<uint16_t*>(dst) + (long long)stream * dst_sstride + (long long)gy * dst_pitch + gx;
          if ((dst_pitch & 3) == 0) {  // gx is a multiple of 4 and < dw: the four outputs exist, 8-byte aligned
            *reinterpret_cast<uint2*>(D) = make_uint2(lo, hi);
          } else {
            D[0] = (uint16_t)lo;
            if (gx + 1 < dw) D[1] = (uint16_t)(lo >> 16);
            if (gx + 2 < dw) D[2] = (uint16_t)hi;
            if (gx + 3 < dw) D[3] = (uint16_t)(hi >> 16);
          }
        } else {
          uint8_t* D = reinterpret_cast<uint8_t*>(dst) + (long long)stream * dst_sstride;
          lo = (lo + 0x00800080u) >> 8 & 0x00ff00ffu;
          hi = (hi + 0x00800080u) >> 8 & 0x00ff00ffu;
          const uint32_t out = (lo & 255) | (lo >> 16 << 8) | (hi & 255) << 16 | (hi >> 16 << 24);
          *reinterpret_cast<uint32_t*>(D + (long long)gy * dst_pitch + gx) = out;  // dst_pitch = roundup4(dw): pad bytes
        }
      }
    }
    // s_h is rewritten only after the next tile's first barrier, which every thread reaches after these reads
  }
}

}  // namespace vsf
#include "vsf_lk.hpp"
namespace vsf {

// Stable in-place compaction of one stream's table by the verified flag; also the three counts
// the reference logs (FeatureTrackerLucasKanade.cpp:87-104).  One block of 256 threads per stream.
__global__ __launch_bounds__(256) void compact_kernel(uint64_t* __restrict__ ids, double2* __restrict__ xy,
                                                      uint32_t* __restrict__ age, int* __restrict__ n, int cap,
                                                      const float2* __restrict__ fwd,
                                                      const uint8_t* __restrict__ flags, int* __restrict__ stats,
                                                      int* __restrict__ tslot) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  __shared__ int s_cnt[3];
  const int stream = blockIdx.x;
  const long long o = (long long)stream * cap;
  const int n0 = n[stream];
  if (threadIdx.x == 0) {
    s_base = 0;
    s_cnt[0] = s_cnt[1] = s_cnt[2] = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c = 0; c < n0; c += 256) {
    const int i = c + threadIdx.x;
    uint8_t fl = 0;
    uint64_t id = 0;
    uint32_t ag = 0;
    float2 q = make_float2(0.f, 0.f);
    if (i < n0) {
      fl = flags[o + i];
      id = ids[o + i];
      ag = age[o + i];
      q = fwd[o + i];
    }
    const bool keep = (fl & 4) != 0;
    const unsigned long long bal = __ballot(keep);
    const unsigned long long bf = __ballot((fl & 1) != 0), bb = __ballot((fl & 2) != 0);
    if (lane == 0) {
      s_wave[wv] = __popcll(bal);
      atomicAdd(&s_cnt[0], __popcll(bf));
      atomicAdd(&s_cnt[1], __popcll(bb));
      atomicAdd(&s_cnt[2], __popcll(bal));
    }
    __syncthreads();  // all reads of this chunk done, wave counts visible
    int off = s_base;
    for (int k = 0; k < wv; k++) off += s_wave[k];
    off += __popcll(bal & ((1ull << lane) - 1ull));
    if (keep) {
      ids[o + off] = id;
      age[o + off] = ag;
      xy[o + off] = make_double2((double)q.x, (double)q.y);  // :231-232 float -> double
      tslot[o + off] = i;  // where the tracker kernel left this feature's template (LK template cache)
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    n[stream] = s_base;
    stats[stream * 4 + 0] = n0;
    stats[stream * 4 + 1] = s_cnt[0];
    stats[stream * 4 + 2] = s_cnt[1];
    stats[stream * 4 + 3] = s_cnt[2];
  }
}

// ------------------------------------------------------------------------------------------------
// Detector (ref: FeatureDetectorShiTomasiBucketing.cpp:67-239; cv::goodFeaturesToTrack)
// ------------------------------------------------------------------------------------------------
struct GridView {
  const int4* rects;      // [ncell] x, y, w, h (cvRound of the Rect2f, Q1)
  const float2* origin;   // [ncell] unrounded Rect2f x, y (what :169 adds to the cell-local corner)
  const int* lam_off;     // [ncell+1] offset of each cell's lambda map (floats)
  int ncell, rows, cols;
  int grid_max;           // max_feature_number / (rows*cols)  (Q3)
  int lam_total;          // floats per stream
};

// Occupancy (Q4): number of distinct flagged pixels per cell, need per cell, and the number of
// cells processed before the reference's early `break` (Q2).  One block per stream (any multiple of 64 threads:
// the host takes 1024 when few streams would leave a stream's thousands of features to 256 threads); an LDS hash
// set de-duplicates pixels exactly like the reference's flag image.  LDS: hash | per-cell counts | column and row
// intervals of the grid.
__global__ __launch_bounds__(1024) void occupancy_kernel(const double2* __restrict__ xy, const int* __restrict__ n,
                                                        int cap, GridView G, int w, int h, int hash_size,
                                                        int* __restrict__ need, int* __restrict__ n_active,
                                                        unsigned int* __restrict__ cell_max,
                                                        int* __restrict__ cand_cnt) {
  extern __shared__ __align__(16) uint8_t smem[];
  int* s_hash = reinterpret_cast<int*>(smem);
  int* s_cnt = s_hash + hash_size;
  int* s_col = s_cnt + G.ncell;    // [cols][2] x0, x1 (exclusive)
  int* s_row = s_col + 2 * G.cols;  // [rows][2] y0, y1
  const int NT = (int)blockDim.x;
  const int stream = blockIdx.x;
  const int n0 = n[stream];
  for (int i = threadIdx.x; i < hash_size; i += NT) s_hash[i] = -1;
  for (int i = threadIdx.x; i < G.ncell; i += NT) s_cnt[i] = 0;
  for (int i = threadIdx.x; i < G.cols; i += NT) {
    const int4 r = G.rects[i * G.rows];
    s_col[2 * i] = r.x;
    s_col[2 * i + 1] = r.x + r.z;
  }
  for (int i = threadIdx.x; i < G.rows; i += NT) {
    const int4 r = G.rects[i];
    s_row[2 * i] = r.y;
    s_row[2 * i + 1] = r.y + r.w;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n0; i += NT) {
    const double2 p = xy[(long long)stream * cap + i];
    if (p.x > 0 && p.x < (double)w && p.y > 0 && p.y < (double)h) {  // :123-124 strict, in double
      const int ix = (int)p.x, iy = (int)p.y;                        // :125 at<uint8_t>(pos[1], pos[0])
      const int key = iy * w + ix;
      unsigned int slot = ((unsigned int)key * 2654435761u) & (unsigned int)(hash_size - 1);
      bool fresh = false;
      for (;;) {
        const int old = atomicCAS(&s_hash[slot], -1, key);
        if (old == -1) {
          fresh = true;
          break;
        }
        if (old == key) break;
        slot = (slot + 1) & (unsigned int)(hash_size - 1);
      }
      if (fresh) {
        // GenerateGrid's cells are a product of column intervals and row intervals (cell = column * rows + row; the
        // rounded intervals may overlap or leave gaps, Q1): test cols + rows intervals, not cols * rows rectangles
        for (int cx = 0; cx < G.cols; cx++) {
          if (ix >= s_col[2 * cx] && ix < s_col[2 * cx + 1]) {
            for (int cy = 0; cy < G.rows; cy++) {
              if (iy >= s_row[2 * cy] && iy < s_row[2 * cy + 1]) atomicAdd(&s_cnt[cx * G.rows + cy], 1);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int act = G.ncell;
    for (int c = 0; c < G.ncell; c++) {
      const int q = G.grid_max - s_cnt[c];  // :152-153
      need[stream * G.ncell + c] = q;
      if (q <= 0 && act == G.ncell) act = c;  // :154 break
    }
    n_active[stream] = act;
  }
  for (int c = threadIdx.x; c < G.ncell; c += NT) {
    cell_max[stream * G.ncell + c] = 0u;
    cand_cnt[stream * G.ncell + c] = 0;
  }
}

__device__ __forceinline__ unsigned int float_order_encode(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_order_decode(unsigned int e) {
  return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

// Correctly rounded square root of a finite x >= 0: v_sqrt_f32 (1 ulp) and the two residual tests the compiler's
// own lowering of sqrtf ends with, without its range scaling (eight more instructions per call, a quarter of the
// box stage's vector work).  The residuals x - s * (s -+ 1 ulp) are of the order s * ulp(s) and would be lost
// to underflow for 0 < x < 2^-96: the caller sends such an x (a wave-uniform test) to sqrtf.
__device__ __forceinline__ float sqrt_rn_midrange(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __int_as_float(__float_as_int(s) - 1), su = __int_as_float(__float_as_int(s) + 1);
  const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
  float r = rd <= 0.f ? sd : s;
  r = ru > 0.f ? su : r;
  return r;
}

// Min-eigenvalue score of one grid cell tile (cv::cornerMinEigenVal on the cell ROI of the parent image;
// SURVEY A.1 steps 1-4), the masked per-cell maximum, AND the candidate corners of the tile (steps 5-6), in one
// pass over the frame: the lambda map is never written to memory.
//   * A block computes lambda on a 64x16 block of cell positions and emits candidates from its inner 62x14
//     (every candidate's 3x3 neighbourhood lies inside the block); tiles step by 62 x 14.
//   * cv thresholds the map at quality * max and then keeps pixels equal to their 3x3 dilation.  The threshold is
//     not known before every tile of the cell has been scored, but it commutes with the local-maximum test: for
//     quality in [0, 1] a surviving pixel has lambda > thr >= 0, every neighbour the threshold zeroes is below it,
//     and the others keep their value, so "thresholded pixel equals its thresholded dilation" <=> "lambda > thr
//     and lambda >= its eight neighbours".  The block therefore emits a key for every positive 3x3 local maximum
//     (interior ring, mask) and select_kernel / anms_topk_kernel apply `lambda > thr` when they read the list.
//     Keys at or below quality * (the cell maximum seen SO FAR: an atomic read at block start joined with the
//     block's own maximum, a lower bound of the final one) can never pass and are dropped at once.
//   * Sobel reads the parent image's real pixels (REFLECT_101 only at the true image border); the box filter
//     reflects at the CELL border and sums in double like cv::boxFilter does for 32F input.
// (Round 3 measured 32-row tiles in blocks of 512 threads: 11 % fewer lambda rows, 16 % fewer covariance rows per
// candidate, the same waves per CU -- and 1.50 against 1.31 ms per 1024 frames at C2, 3.25 against 2.52 ms with the
// ANMS detector: eight waves meeting at every barrier and four blocks a CU to overlap their phases cost more than
// the halo rows saved.  -DVSF_SC_TH=32 still builds and passes the parity tests.)
#ifndef VSF_SC_TH
#define VSF_SC_TH 16
#endif
constexpr int SC_TW = 64, SC_TH = VSF_SC_TH;   // lambda block of a tile (16 or 32 rows)
constexpr int SC_NT = SC_TW * SC_TH / 4;       // threads of a block: four lambda rows of one column per thread
constexpr int SC_IW = SC_TW - 2, SC_IH = SC_TH - 2;  // candidate positions of a tile (tile step)
constexpr int SC_FP = 72;  // LDS pitch (floats) of the staged tile (tile width + up to 3 pixels of dword alignment, rounded up to dwords)
constexpr int SC_LP = SC_TW + 1;               // LDS pitch of the lambda block
constexpr int SC_LCAP = SC_IH * SC_IW / 4 + 64;  // keys gathered in LDS per block; beyond it they go straight to the list

// Tiles along one axis of a cell: tile t holds lambda at [t * step, t * step + step + 2) and candidates one in
// from both ends, so T tiles cover the candidates 1 .. T * step and lambda 0 .. T * step + 1.
__host__ __device__ inline int sc_tiles(int cells_extent, int step) {
  const int t = (cells_extent - 2 + step - 1) / step;
  return t > 1 ? t : 1;
}

// BS = blockSize of the box filter: 3 (offsets -1..+1; cv::goodFeaturesToTrack default, the bucketing
// detector) or 4 (offsets -2..+1: anchor = blockSize / 2; the ANMS detector's GFTTDetector).
// WRITE_LAM: also store the lambda block to the cell's map (fallback / debug); EMIT: emit candidate keys.
// HARRIS: the response is cv::cornerHarris's a c - b^2 - k (a + c)^2 instead of the smaller eigenvalue (the ANMS
// detector's detection_use_harris_corner_ / detection_k_, FeatureDetectorANMS.cpp:27-28, :41-47); everything
// before and after the per-pixel formula is shared.
template <int BS, bool WRITE_LAM, bool EMIT, bool HARRIS = false>
__device__ __forceinline__ void score_tile(const uint8_t* __restrict__ F, int row_stride, int w, int h,
                                           const uint8_t* __restrict__ M, int mask_row_stride, int4 rc, int tx, int ty,
                                           float k_side, float k_mid, float* __restrict__ lam_cell,
                                           unsigned int* __restrict__ cmax, double quality,
                                           unsigned long long* __restrict__ cand, int cand_cap,
                                           int* __restrict__ ccnt, double harris_k = 0.0) {
  constexpr int HL = BS / 2, HR = BS - 1 - HL;             // covariance halo left/top, right/bottom
  constexpr int CW = SC_TW + HL + HR, CH = SC_TH + HL + HR;  // covariance tile
  constexpr int IW = CW + 2, IH = CH + 2;                  // staged image tile (Sobel reads +-1 more)
  constexpr int NDW = (IW + 3 + 3) / 4;                    // dwords of a staged row (up to 3 bytes of alignment slack)
  static_assert(4 * NDW <= SC_FP && SC_FP % 4 == 0, "tile pitch");
  // LDS: the staged image (as float32: every pixel is converted once, not once per use) is dead once the
  // covariance products exist, and those once lambda exists; the lambda block and the mask bits reuse the
  // image's space, the gathered keys the products'.  20.0 KB a block with the 3x3 box (8 blocks per CU), 22.9 KB with
  // the 4x4 box (7: LDS is granted in 1280-byte units, and a row pitch of 76 floats was one unit too many).  (The products stay
  // float32 although the box filter sums them in double: held as float64 the conversions drop to a third,
  // but the 8-byte LDS reads of the box stage then cost more than the conversions saved -- the LDS pipe is
  // nearly as busy as the VALU here -- and the kernel ran 1.7x slower.)
  constexpr int NSTRIP = 3 * (SC_TH / 16);         // waves that walk columns 0..63 of the covariance tile
  constexpr int RS = (CH + NSTRIP - 1) / NSTRIP;   // rows of a covariance strip; the product planes are padded to
  constexpr int CHP = NSTRIP * RS;                 // NSTRIP * RS rows so that the last strip needs no row test
  constexpr int IMG_BYTES = IH * SC_FP * 4, COV_BYTES = 3 * CHP * CW * 4;
  constexpr int LAM_BYTES = (SC_TH * SC_LP * 4 + 15) & ~15;
  static_assert(IMG_BYTES % 16 == 0 && LAM_BYTES + SC_TH * SC_TW <= IMG_BYTES && SC_LCAP * 8 <= COV_BYTES,
                "aliased regions");
  __shared__ __align__(16) uint8_t s_raw[IMG_BYTES + COV_BYTES];
  float* s_img = reinterpret_cast<float*>(s_raw);
  float* s_lam = reinterpret_cast<float*>(s_raw);  // after the covariance stage
  uint8_t* s_msk = s_raw + LAM_BYTES;              // after the covariance stage
  // dx*dx, dx*dy, dy*dy: float32 products as cv forms them
  float(*s_cov)[CHP * CW] = reinterpret_cast<float(*)[CHP * CW]>(s_raw + IMG_BYTES);
  unsigned long long* s_keys = reinterpret_cast<unsigned long long*>(s_raw + IMG_BYTES);  // after the lambda stage
  __shared__ unsigned int s_max;
  __shared__ int s_n, s_base;
  const int cw = rc.z, ch = rc.w;
  const int cx0 = tx * SC_IW, cy0 = ty * SC_IH;  // origin of the lambda block in cell coordinates
  if (tx >= sc_tiles(cw, SC_IW) || ty >= sc_tiles(ch, SC_IH)) return;  // block-uniform: a smaller cell needs fewer tiles
  // cv::boxFilter reflects (REFLECT_101) at the CELL border.  Unless the cell is tiny, the covariance stage stores
  // every product it forms one or two rows / columns inside the border a second time at its mirror position outside,
  // and the box stage of border tiles then runs the same straight-line code as that of inner tiles.
  const bool mirror = cw >= 4 && ch >= 4;
  // block-uniform: this tile holds a mirrored row (only the cell's first and last tile rows do)
  const bool edge_rows = cy0 == 0 || ch - cy0 < SC_TH + HR + 1;
  unsigned int seen = 0u;
  if (threadIdx.x == 0) {
    s_max = 0u;
    s_n = 0;
  }
  // staged image tile: parent coords [rc.x + cx0 - HL - 1, +IW) x [rc.y + cy0 - HL - 1, +IH)
  const int gx0 = rc.x + cx0 - HL - 1, gy0 = rc.y + cy0 - HL - 1;
  // tiles away from the image border are fetched as aligned dwords (sh = bytes of alignment slack);
  // border tiles byte by byte with REFLECT_101
  const int sh = gx0 & 3;
  const bool img_interior = gx0 >= 0 && gy0 >= 0 && gx0 + IW <= w && gy0 + IH <= h && (row_stride & 3) == 0 &&
                            (reinterpret_cast<uintptr_t>(F) & 3) == 0 && (gx0 - sh) + 4 * NDW <= row_stride;
  if (img_interior) {
    const unsigned int* Fw = reinterpret_cast<const unsigned int*>(F + (long long)gy0 * row_stride + (gx0 - sh));
    const unsigned int rsw = (unsigned int)row_stride >> 2;
    // every load of the thread is in flight (and the cell maximum seen so far behind them) before the first
    // conversion waits: a block spends its first microseconds on latency, during which its LDS keeps another from
    // the CU
    constexpr int NLD = (IH * NDW + SC_NT - 1) / SC_NT;
    unsigned int v[NLD];
#pragma unroll
    for (int k = 0; k < NLD; k++) {
      // (lanes past the end load the last dword again: no branch between the loads)
      const int i = min((int)threadIdx.x + k * SC_NT, IH * NDW - 1), y = i / NDW, xd = i - y * NDW;
      v[k] = Fw[(unsigned int)y * rsw + (unsigned int)xd];
    }
    unsigned int seen_v = 0u;
    if (EMIT) seen_v = __hip_atomic_load(cmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < NLD; k++) {
      const int i = (int)threadIdx.x + k * SC_NT, y = i / NDW, xd = i - y * NDW;
      if (i < IH * NDW)
        *reinterpret_cast<float4*>(s_img + y * SC_FP + 4 * xd) = make_float4(
            (float)(v[k] & 255u), (float)((v[k] >> 8) & 255u), (float)((v[k] >> 16) & 255u), (float)(v[k] >> 24));
    }
    asm volatile("" : "+v"(seen_v));  // (stays a vector value until here: no wait for it in front of the conversions)
    seen = (unsigned int)__builtin_amdgcn_readfirstlane((int)seen_v);
  } else {
    if (EMIT) seen = __hip_atomic_load(cmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = threadIdx.x; i < IH * IW; i += SC_NT) {
      const int y = i / IW, x = i - y * IW;
      const int gy = reflect101(gy0 + y, h), gx = reflect101(gx0 + x, w);
      s_img[y * SC_FP + sh + x] = (float)F[(long long)gy * row_stride + gx];
    }
  }
  __syncthreads();
  // covariance at cell coords [cx0-HL, cx0+SC_TW+HR) x [cy0-HL, cy0+SC_TH+HR) that lie inside the cell.
  // A thread walks one column of the tile down a strip of rows: per image row it forms the horizontal
  // difference r = I(x+1) - I(x-1) and the horizontal smooth s = k I(x-1) + 2k I(x) + k I(x+1) ONCE, and every
  // covariance row combines three consecutive (r, s) pairs -- the same float operations in the same order as
  // the direct 3x3 form, a third of the multiplies.  Waves 0..2 take columns 0..63 in three strips of RS rows;
  // the two or three columns left over go to wave 3, one (column, row) per lane (spread over the 198 or 201
  // threads of a plain column-major split, the fourth wave would run the whole strip for half a dozen lanes).
  {
    static_assert(NSTRIP * RS >= CH && (CW - 64) * CH <= 64 * (SC_NT / 64 - NSTRIP) && CW > 64,
                  "strip mapping of the covariance stage");
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    int x, y0, nrows;
    if (wv < NSTRIP) {
      x = ln, y0 = wv * RS, nrows = min(RS, CH - y0);
    } else {
      const int idx = (wv - NSTRIP) * 64 + ln, c = idx / CH;
      x = 64 + c, y0 = idx - c * CH, nrows = (idx < (CW - 64) * CH) ? 1 : 0;
    }
    const int cx = cx0 - HL + x;
    const bool xin = mirror ? (cx >= -HL && cx < cw + HR) : (cx >= 0 && cx < cw);
    // a column outside the cell carries its mirror's products: it reads the mirror's image columns
    const float* col = s_img + (sh + x) + ((mirror && xin) ? reflect101(cx, cw) - cx : 0);
    auto rowvals = [&](int iy, float& rr, float& ss) {
      const float* c = col + iy * SC_FP;
      const float f0 = c[0], f1 = c[1], f2 = c[2];
      rr = f2 - f0;  // exact: equals (float)(int difference)
      float sv = k_side * f0;
      sv += k_mid * f1;
      sv += k_side * f2;
      ss = sv;
    };
    auto put = [&](int yy, float pxx, float pxy, float pyy) {
      s_cov[0][yy * CW + x] = pxx;
      s_cov[1][yy * CW + x] = pxy;
      s_cov[2][yy * CW + x] = pyy;
    };
    if (wv < NSTRIP && mirror && !edge_rows) {
      // every row of the tile lies inside the cell: straight-line code, no per-row test the compiler would
      // turn into a branch (as basic blocks per row the LDS reads of a row no longer overlap the arithmetic
      // of the one before: +12 % on the whole kernel, measured).  The last strip runs past row CH - 1: it reads
      // whatever follows the image tile in LDS and stores the result to the padding rows, which nobody reads.
      if (xin) {
        float r0, s0, r1, s1;
        rowvals(y0, r0, s0);
        rowvals(y0 + 1, r1, s1);
#pragma unroll
        for (int k = 0; k < RS; k++) {
          const int y = y0 + k;
          float r2, s2;
          rowvals(y + 2, r2, s2);
          const float dx = (r0 + r2) * k_side + r1 * k_mid;
          const float dy = s2 - s0;
          put(y, dx * dx, dx * dy, dy * dy);
          r0 = r1;
          s0 = s1;
          r1 = r2;
          s1 = s2;
        }
      }
    } else if (nrows > 0 && (xin || !mirror)) {
      float r0, s0, r1, s1;
      rowvals(y0, r0, s0);
      rowvals(y0 + 1, r1, s1);
#pragma unroll
      for (int k = 0; k < RS; k++) {
        const int y = y0 + k;
        if (k < nrows) {
          float r2, s2;
          rowvals(y + 2, r2, s2);
          const float dx = (r0 + r2) * k_side + r1 * k_mid;
          const float dy = s2 - s0;
          const int cy = cy0 - HL + y;
          const bool in = xin && cy >= 0 && cy < ch;
          if (!mirror) {
            put(y, in ? dx * dx : 0.f, in ? dx * dy : 0.f, in ? dy * dy : 0.f);
          } else if (in) {
            const float pxx = dx * dx, pxy = dx * dy, pyy = dy * dy;
            put(y, pxx, pxy, pyy);
            // rows -d and ch - 1 + d mirror rows d and ch - 1 - d (their own walkers store nothing)
            if (edge_rows) {
#pragma unroll
              for (int d = 1; d <= HL; d++)
                if (cy == d && y - 2 * d >= 0) put(y - 2 * d, pxx, pxy, pyy);
#pragma unroll
              for (int d = 1; d <= HR; d++)
                if (cy == ch - 1 - d && y + 2 * d < CH) put(y + 2 * d, pxx, pxy, pyy);
            }
          }
          r0 = r1;
          s0 = s1;
          r1 = r2;
          s1 = s2;
        }
      }
    }
  }
  __syncthreads();
  // the block's masked maximum as a float (-inf: none yet); lambda is finite and never -0 (a, c >= +0), so the
  // order encoding commutes with the maximum and is applied once per thread, not once per pixel
  float local_maxf = -INFINITY;
  const float thr_seen = (EMIT && seen) ? (float)((double)float_order_decode(seen) * quality) : -INFINITY;
  // (x, y): position inside the lambda block; (cx, cy) = (cx0 + x, cy0 + y) lies inside the cell
  auto mask_at = [&](int cx, int cy) -> bool {
    return !M || M[(long long)(rc.y + cy) * mask_row_stride + (rc.x + cx)];
  };
  // lam_p / msk_p: the lambda-block / mask slot the value goes to (strip threads pass a base + constant offset)
  // (valid: the position lies inside the cell.  The strip code of the box stage finishes all four rows of a thread
  // without a row test -- a row below the cell costs nothing extra in a wave that runs anyway, and four tests
  // are a dozen scalar instructions of mask bookkeeping -- so an invalid row arrives here with sums of whatever the
  // padding held: it is kept out of the maximum and of the cell's map, and what it leaves in the lambda block is
  // never read, a candidate's neighbours being rows of the cell.)
  auto finish = [&](int x, int y, const double (&sum)[3], bool mk, bool valid, float* lam_p, uint8_t* msk_p) {
    const int cx = cx0 + x, cy = cy0 + y;
    if (HARRIS) {
      // cv's calcHarris: groups of four pixels of a row in float32 with (float)k, products left to right; the last
      // width % 4 pixels of the row with k as a double (the oracle's corner_response states both forms).  The
      // response may be negative, so nothing is skipped and 0 stands in for nothing.
      const float a = (float)sum[0], b = (float)sum[1], c = (float)sum[2];
      const float ac_bb = a * c - b * b, t = a + c;
      float lm;
      if (cx < (cw & ~3))
        lm = ac_bb - (float)harris_k * t * t;
      else
        lm = (float)((double)ac_bb - harris_k * (double)t * (double)t);
      if (WRITE_LAM && valid) lam_cell[cy * cw + cx] = lm;
      if (EMIT) {
        *lam_p = lm;
        *msk_p = mk ? 1 : 0;
      }
      local_maxf = fmaxf(local_maxf, (mk && valid) ? lm : -INFINITY);
      return;
    }
    const float a = (float)sum[0] * 0.5f, b = (float)sum[1], c = (float)sum[2] * 0.5f;
    // lambda <= a + c.  Where a + c is not above quality * (the cell maximum seen at block start) the pixel passes
    // neither the final threshold nor the maximum, and as a neighbour any value below a surviving candidate's
    // (which is above that bound) decides the same: 0 stands in, and the correctly rounded square root (a dozen
    // instructions) is skipped -- for whole waves in smooth regions.
    float lm = 0.f;
    const bool full = WRITE_LAM || a + c > thr_seen;
    if (full) {
      const float q = (a - c) * (a - c) + b * b;
      float sq;
      if (__ballot(q < 0x1p-96f && q != 0.f) == 0ull) {
        sq = sqrt_rn_midrange(q);
      } else {
        asm volatile("");  // (rare: a block of its own)
        sq = sqrtf(q);
      }
      lm = (a + c) - sq;
    }
    if (WRITE_LAM && valid) lam_cell[cy * cw + cx] = lm;
    if (EMIT) {
      *lam_p = lm;
      *msk_p = mk ? 1 : 0;
    }
    // (a skipped pixel's stand-in 0 may join the maximum: it is skipped only below a threshold seen >= 0, and a
    // block maximum at or below the one seen changes nothing.)  Plain v_max_f32: fmaxf would first quiet both
    // operands, and neither can be a NaN.
    float cand_v = (mk && valid) ? lm : -INFINITY;
    asm("v_max_f32 %0, %0, %1" : "+v"(local_maxf) : "v"(cand_v));
  };
  // Box filter in double like cv::boxFilter for 32F input: row sum left to right, then the column of
  // row sums top to bottom.
  if (mirror) {
    // a thread owns one column and four consecutive rows, forms the BS + 3 row sums they share once, and adds BS
    // of them per pixel (same additions, same order); positions outside the cell are skipped
    const int x = threadIdx.x & (SC_TW - 1), y0 = (threadIdx.x / SC_TW) * 4;
    static_assert(SC_TW == 64 && SC_TH * SC_TW == SC_NT * 4, "strip mapping");
    if (cx0 + x < cw && cy0 + y0 < ch) {
      // the four mask bytes first (rows clamped into the cell), so that their global latency runs under the row
      // sums (loaded per pixel behind the row test the kernel also needs 6 to 20 registers more)
      bool mk4[4];
#pragma unroll
      for (int o = 0; o < 4; o++) mk4[o] = mask_at(cx0 + x, min(cy0 + y0 + o, ch - 1));
      float* lam_base = s_lam + y0 * SC_LP + x;
      uint8_t* msk_base = s_msk + y0 * SC_TW + x;
      double rs[3][BS + 3];
#pragma unroll
      for (int k = 0; k < BS + 3; k++) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const float* row = s_cov[q] + (y0 + k) * CW + x;
          double t = (double)row[0];
#pragma unroll
          for (int i = 1; i < BS; i++) t += (double)row[i];
          rs[q][k] = t;
        }
      }
      // the image tile is dead (barrier above); lambda and mask bits go where it was.  No thread reads
      // s_lam / s_msk before the barrier after this stage.
#pragma unroll
      for (int o = 0; o < 4; o++) {
        double sum[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
          sum[q] = rs[q][o];
#pragma unroll
          for (int j = 1; j < BS; j++) sum[q] += rs[q][o + j];
        }
        finish(x, y0 + o, sum, mk4[o], cy0 + y0 + o < ch, lam_base + o * SC_LP, msk_base + o * SC_TW);
      }
    }
  } else {
    for (int i = threadIdx.x; i < SC_TH * SC_TW; i += SC_NT) {
      const int y = i / SC_TW, x = i - y * SC_TW;
      const int cx = cx0 + x, cy = cy0 + y;
      if (cx >= 0 && cx < cw && cy >= 0 && cy < ch) {
        int xi[BS];
#pragma unroll
        for (int q = 0; q < BS; q++) xi[q] = reflect101(cx - HL + q, cw) - (cx0 - HL);
        double sum[3];
#pragma unroll
        for (int j = 0; j < BS; j++) {
          const int yy = reflect101(cy - HL + j, ch) - (cy0 - HL);
#pragma unroll
          for (int q = 0; q < 3; q++) {
            const float* row = s_cov[q] + yy * CW;
            double t = (double)row[xi[0]];
#pragma unroll
            for (int k = 1; k < BS; k++) t += (double)row[xi[k]];
            sum[q] = (j == 0) ? t : sum[q] + t;
          }
        }
        finish(x, y, sum, mask_at(cx, cy), true, s_lam + y * SC_LP + x, s_msk + y * SC_TW + x);
      }
    }
  }
  // wave max, then one LDS atomic per wave, one global atomic per block
  unsigned int local_max = local_maxf == -INFINITY ? 0u : float_order_encode(local_maxf);
  for (int m = 32; m >= 1; m >>= 1) local_max = max(local_max, (unsigned int)__shfl_xor((int)local_max, m, 64));
  if ((threadIdx.x & 63) == 0 && local_max) atomicMax(&s_max, local_max);
  __syncthreads();
  const unsigned int blk_max = s_max;
  // (a block that cannot raise the maximum it saw at its start sends no atomic: with the whole frame as ONE cell
  // -- the ANMS detector -- thousands of tiles would otherwise queue on one address)
  if (threadIdx.x == 0 && blk_max > seen) atomicMax(cmax, blk_max);
  if (!EMIT) return;
  // ---- candidates of the inner 62 x 14 positions
  const unsigned int lb = max(seen, blk_max);  // lower bound of the cell's final masked maximum (0: none yet)
  const float thr_lb = lb ? (float)((double)float_order_decode(lb) * quality) : -INFINITY;
  // A thread looks at the four rows of one column it finished in the box stage.  The four values and mask bytes
  // are read together and tested together -- a candidate is rare, most waves leave here --; the 3 x 3 maxima of
  // the four rows share their six row maxima.  (Row and column indices that fall off the block are clamped or read
  // the pad column: they only feed positions on the block's rim, which are no candidates of this tile.)
  {
    const int x = threadIdx.x & (SC_TW - 1), y0 = (int)(threadIdx.x >> 6) * 4, cx = cx0 + x;
    const float* col = s_lam + x;
    float vc[4];
    uint8_t mc[4];
#pragma unroll
    for (int o = 0; o < 4; o++) {
      vc[o] = col[(y0 + o) * SC_LP];
      mc[o] = s_msk[(y0 + o) * SC_TW + x];
    }
    const bool xok = x >= 1 && x <= SC_TW - 2 && cx >= 1 && cx < cw - 1;
    bool pass[4], any = false;
#pragma unroll
    for (int o = 0; o < 4; o++) {
      const int y = y0 + o, cy = cy0 + y;
      pass[o] = xok && y >= 1 && y <= SC_TH - 2 && cy >= 1 && cy < ch - 1 && vc[o] > 0.f && vc[o] > thr_lb && mc[o] != 0;
      any |= pass[o];
    }
    if (any) {
      float hm[6];
#pragma unroll
      for (int r = 0; r < 6; r++) {
        const int yy = min(max(y0 - 1 + r, 0), SC_TH - 1);
        const float* c = col + yy * SC_LP;
        hm[r] = fmaxf(fmaxf(c[-1], c[0]), c[1]);
      }
#pragma unroll
      for (int o = 0; o < 4; o++) {
        if (pass[o] && vc[o] == fmaxf(fmaxf(hm[o], hm[o + 1]), hm[o + 2])) {
          // key = lambda bits << 32 | y << 16 | x: unsigned order = cv's greaterThanPtr order (value, then address:
          // the raster order of (y, x), kept as two fields so that no reader has to divide by the cell width)
          const unsigned long long key =
              ((unsigned long long)__float_as_uint(vc[o]) << 32) | (unsigned int)(((cy0 + y0 + o) << 16) | cx);
          const int ls = atomicAdd(&s_n, 1);
          if (ls < SC_LCAP) {
            s_keys[ls] = key;
          } else {
            const int slot = atomicAdd(ccnt, 1);
            if (slot < cand_cap) cand[slot] = key;
          }
        }
      }
    }
  }
  __syncthreads();
  // ONE global atomic per block reserves the slots of its keys in the cell's list (order does not matter)
  const int nl = min(s_n, SC_LCAP);
  if (threadIdx.x == 0 && nl > 0) s_base = atomicAdd(ccnt, nl);
  __syncthreads();
  for (int i = threadIdx.x; i < nl; i += SC_NT) {
    const int slot = s_base + i;
    if (slot < cand_cap) cand[slot] = s_keys[i];
  }
}

// Fused score + candidates.  Grid: (tiles_x, tiles_y, streams * cells) -- blockIdx.z = (stream - stream0) * cells +
// cell; the host splits the streams over several launches when that exceeds the grid's 65535 -- or with
// stream_major (streams, tiles_x, tiles_y): the ANMS detector's single whole-frame cell, where the blocks in
// flight should belong to many streams -- every block ends with one returning atomic on its cell's counter, and
// with the tiles of ONE cell in flight together a few hundred blocks queue on one address (the L2's same-address
// atomic rate: 1.67 against 1.30 ms per 1024 frames, measured on the former separate candidates kernel).
// The block's coordinates come from the three grid indices and ONE multiply-high (cell_magic = ceil(2^32 / cells),
// exact for blockIdx.z * cells < 2^32; 0 stands for a single cell, whose reciprocal does not fit 32 bits): decoded from a flat index they took two integer divisions, some fifty scalar
// instructions in front of every wave -- a tenth of the kernel's issue slots at four waves a block.
#ifndef VSF_SC_W4
#define VSF_SC_W4 7  // waves per SIMD the 4x4-box instance is compiled for (at 8 it spills)
#endif
template <int BS, bool HARRIS = false>
__global__ __launch_bounds__(SC_NT, (BS == 4 ? VSF_SC_W4 : 8)) void score_kernel(const uint8_t* __restrict__ frames, long long frame_sstride,
                                                    int row_stride, int w, int h,
                                                    const uint8_t* __restrict__ masks, long long mask_sstride,
                                                    int mask_row_stride, GridView G,
                                                    const int* __restrict__ n_active, int tiles_x, int tiles_y,
                                                    float k_side, float k_mid, double quality,
                                                    unsigned int* __restrict__ cell_max,
                                                    unsigned long long* __restrict__ cand, int cand_cap,
                                                    int* __restrict__ cand_cnt, int stream_major,
                                                    double harris_k, int stream0, unsigned int cell_magic) {
  // every argument in a scalar register NOW: one batch of wide scalar loads and one wait, where the compiler would
  // fetch them a few at a time as the code reaches them, a memory round trip each
  asm volatile("" ::"s"(frames), "s"(frame_sstride), "s"(row_stride), "s"(w), "s"(h), "s"(masks), "s"(mask_sstride),
               "s"(mask_row_stride), "s"(G.rects), "s"(G.ncell), "s"(n_active), "s"(k_side), "s"(k_mid), "s"(quality),
               "s"(cell_max), "s"(cand), "s"(cand_cap), "s"(cand_cnt), "s"(stream_major), "s"(stream0), "s"(cell_magic));
  int stream, cell, tx, ty;
  if (stream_major) {
    stream = blockIdx.x, cell = 0, tx = blockIdx.y, ty = blockIdx.z;
  } else {
    const int ds = cell_magic ? (int)__umulhi(blockIdx.z, cell_magic) : (int)blockIdx.z;  // (0: one cell a stream)
    stream = stream0 + ds, cell = (int)blockIdx.z - ds * G.ncell, tx = blockIdx.x, ty = blockIdx.y;
  }
  // (read through the constant address space: scalar loads whatever the compiler can prove about the asm above;
  // both arrays were written by kernels that ended before this one started)
  typedef const int __attribute__((address_space(4))) * const_int_p;
  const int nact = ((const_int_p)(uintptr_t)n_active)[stream];
  const const_int_p rp = (const_int_p)(uintptr_t)(G.rects + cell);
  const int4 rect = make_int4(rp[0], rp[1], rp[2], rp[3]);
  asm volatile("" ::"s"(nact), "s"(rect.x), "s"(rect.y), "s"(rect.z), "s"(rect.w));  // (both loads, then one wait)
  if (cell >= nact) return;
  const int sc = stream * G.ncell + cell;
  score_tile<BS, false, true, HARRIS>(frames + (long long)stream * frame_sstride, row_stride, w, h,
                                      masks ? masks + (long long)stream * mask_sstride : nullptr, mask_row_stride,
                                      rect, tx, ty, k_side, k_mid, nullptr, cell_max + sc, quality,
                                      cand + (long long)sc * cand_cap, cand_cap, cand_cnt + sc, harris_k);
}

// The lambda map of the cells whose candidate list overflowed (select_kernel / anms_topk_kernel then re-derive the
// candidates from the map: exact, slow), or of ONE cell regardless (force_cell >= 0: vsf_debug_min_eig).  Same tile
// routine, so the same arithmetic.  Grid: (cells, streams); a block walks the tiles of its cell.
// (HARRIS: a negative masked maximum -- an image of edges only -- needs nothing special: cv's threshold
// max * quality, quality in [0, 1], is then at or above the maximum, so no response passes it; the candidate list
// holds positive local maxima only and is empty as well.)
template <int BS, bool HARRIS = false>
__global__ __launch_bounds__(SC_NT) void lam_fill_kernel(const uint8_t* __restrict__ frames, long long frame_sstride,
                                                       int row_stride, int w, int h,
                                                       const uint8_t* __restrict__ masks, long long mask_sstride,
                                                       int mask_row_stride, GridView G,
                                                       const int* __restrict__ n_active, int tiles_x, int tiles_y,
                                                       float k_side, float k_mid, float* __restrict__ lam,
                                                       unsigned int* __restrict__ cell_max,
                                                       const int* __restrict__ cand_cnt, int cand_cap, int force_cell,
                                                       double harris_k) {
  const int stream = blockIdx.y, cell = blockIdx.x;
  const int sc = stream * G.ncell + cell;
  if (force_cell >= 0) {
    if (cell != force_cell) return;
  } else if (cell >= n_active[stream] || cand_cnt[sc] <= cand_cap) {
    return;
  }
  float* lam_cell = lam + (long long)stream * G.lam_total + G.lam_off[cell];
  for (int t = 0; t < tiles_x * tiles_y; t++) {
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    score_tile<BS, true, false, HARRIS>(frames + (long long)stream * frame_sstride, row_stride, w, h,
                                        masks ? masks + (long long)stream * mask_sstride : nullptr, mask_row_stride,
                                        G.rects[cell], tx, ty, k_side, k_mid, lam_cell, cell_max + sc, 0.0, nullptr, 0,
                                        nullptr, harris_k);
    __syncthreads();  // the tile's LDS is reused by the next one
  }
}

// Greedy min-distance selection + the reference's nearest-neighbour
// rejection against PRE-EXISTING features (Q5).  One block per (stream, cell).
// "Visit candidates in sorted order, accept when no accepted corner is closer than minDist" is
// evaluated as "repeatedly take the largest remaining key, then drop everything within minDist of
// it": the same accepted sequence, without a sort.  When the candidate list overflowed cand_cap
// the same loop runs over the whole lambda map instead (slow, exact).
struct SelParams {
  double min_dist;   // detector min_feature_distance
  double quality;
  int cap;           // table capacity
  int cand_cap;
  int sel_cap;       // accepted-corner slots per cell (= grid_max)
  int lds_keys;      // candidate keys the block's LDS holds (<= cand_cap); more survivors stay in the global list
};

__global__ __launch_bounds__(256) void select_kernel(const float* __restrict__ lam, GridView G,
                                                     const int* __restrict__ n_active,
                                                     const int* __restrict__ need,
                                                     const unsigned int* __restrict__ cell_max,
                                                     const uint8_t* __restrict__ masks, long long mask_sstride,
                                                     int mask_row_stride,
                                                     unsigned long long* __restrict__ cand,
                                                     const int* __restrict__ cand_cnt, SelParams sp,
                                                     const double2* __restrict__ xy, const int* __restrict__ n,
                                                     double2* __restrict__ sel_xy, int* __restrict__ sel_cnt) {
  extern __shared__ __align__(16) uint8_t smem[];
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(smem);  // [lds_keys], 0 = dead
  int* s_acc_x = reinterpret_cast<int*>(s_key + sp.lds_keys);               // [sel_cap]
  int* s_acc_y = s_acc_x + sp.sel_cap;                                      // [sel_cap]
  int* s_ok = s_acc_y + sp.sel_cap;                                         // [sel_cap]
  __shared__ unsigned long long s_red[4];
  const int stream = blockIdx.y, cell = blockIdx.x;
  const int sc = stream * G.ncell + cell;
  if (cell >= n_active[stream]) {
    if (threadIdx.x == 0) sel_cnt[sc] = 0;
    return;
  }
  const int4 rc = G.rects[cell];
  const int cw = rc.z, ch = rc.w;
  const int want = min(need[sc], sp.sel_cap);
  const int total = cand_cnt[sc];
  const bool overflow = total > sp.cand_cap;
  const float* L = lam + (long long)stream * G.lam_total + G.lam_off[cell];
  const uint8_t* Mk = masks ? masks + (long long)stream * mask_sstride : nullptr;
  const unsigned int enc = cell_max[sc];
  const float thr = (float)((enc ? (double)float_order_decode(enc) : 0.0) * sp.quality);
  const double md2 = sp.min_dist * sp.min_dist;  // cv: minDistance *= minDistance (double)
  const bool use_dist = sp.min_dist >= 1.0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

  // the list holds every positive local maximum the score kernel could not rule out; cv's threshold
  // (lambda > (float)(max * quality)) is applied here, and the survivors are compacted in LDS.  The LDS holds
  // lds_keys of them -- a few hundred survive in a cell of any size, and LDS sized for the list's capacity (30 KB at
  // C2) kept all but five blocks off a CU --; when more survive (in_lds false, rare) they stay where they are, in
  // the cell's global list, with the keys under the threshold zeroed in place.
  __shared__ int s_m;
  if (threadIdx.x == 0) s_m = 0;
  __syncthreads();
  unsigned long long* gkey = cand + (long long)sc * sp.cand_cap;
  if (!overflow)
    for (int i = threadIdx.x; i < total; i += 256) {
      const unsigned long long k = gkey[i];
      if (__uint_as_float((unsigned int)(k >> 32)) > thr) {
        const int slot = atomicAdd(&s_m, 1);
        if (slot < sp.lds_keys) s_key[slot] = k;
      }
    }
  __syncthreads();
  const bool in_lds = s_m <= sp.lds_keys;
  if (!overflow && !in_lds)
    for (int i = threadIdx.x; i < total; i += 256)  // (a thread owns the same elements in every pass below)
      if (!(__uint_as_float((unsigned int)(gkey[i] >> 32)) > thr)) gkey[i] = 0ull;
  const int M = overflow ? cw * ch : (in_lds ? s_m : total);

  int accepted = 0;
  int lastx = -1000000, lasty = -1000000;
  unsigned long long lastkey = ~0ull;
  while (accepted < want) {
    // ---- arg-max over the remaining candidates
    unsigned long long best = 0ull;
    if (!overflow) {
      // fast path: the suppression by the last accepted corner is folded into the scan
      auto scan = [&](auto* keys) {
        for (int i = threadIdx.x; i < M; i += 256) {
          const unsigned long long k = keys[i];
          if (k) {
            const int y = (int)((unsigned int)k >> 16), x = (int)((unsigned int)k & 0xffffu);
            const float ddx = (float)x - (float)lastx, ddy = (float)y - (float)lasty;
            const bool dead = (k == lastkey) || (use_dist && (double)(ddx * ddx + ddy * ddy) < md2);
            if (dead)
              keys[i] = 0ull;
            else if (k > best)
              best = k;
          }
        }
      };
      if (in_lds)
        scan(s_key);
      else
        scan(gkey);
    } else {
      // overflow path: candidates are re-derived from the lambda map; keys are unique and visited in
      // descending order, so everything >= lastkey has been decided already
      for (int i = threadIdx.x; i < M; i += 256) {
        const int y = i / cw, x = i - y * cw;
        if (x < 1 || x >= cw - 1 || y < 1 || y >= ch - 1) continue;
        float v = L[i];
        v = v > thr ? v : 0.f;
        if (v == 0.f) continue;
        const unsigned long long k = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned int)((y << 16) | x);
        if (k >= lastkey || k <= best) continue;
        float m = v;
        for (int j = -1; j <= 1; j++)
          for (int q = -1; q <= 1; q++) {
            float t = L[(y + j) * cw + (x + q)];
            t = t > thr ? t : 0.f;
            m = fmaxf(m, t);
          }
        if (v != m) continue;
        if (Mk && !Mk[(long long)(rc.y + y) * mask_row_stride + (rc.x + x)]) continue;
        bool dead = false;
        if (use_dist)
          for (int a = 0; a < accepted && !dead; a++) {
            const float ddx = (float)x - (float)s_acc_x[a], ddy = (float)y - (float)s_acc_y[a];
            dead = (double)(ddx * ddx + ddy * ddy) < md2;
          }
        if (!dead) best = k;
      }
    }
    for (int m = 32; m >= 1; m >>= 1) {
      const unsigned long long o = __shfl_xor(best, m, 64);
      best = o > best ? o : best;
    }
    __syncthreads();
    if (lane == 0) s_red[wv] = best;
    __syncthreads();
    best = s_red[0];
    for (int k = 1; k < 4; k++) best = s_red[k] > best ? s_red[k] : best;
    if (best == 0ull) break;
    lasty = (int)((unsigned int)best >> 16);
    lastx = (int)((unsigned int)best & 0xffffu);
    lastkey = best;
    if (threadIdx.x == 0) {
      s_acc_x[accepted] = lastx;
      s_acc_y[accepted] = lasty;
    }
    accepted++;
    __syncthreads();  // s_acc visible to the overflow path's distance test
  }
  // ---- NN rejection against the pre-existing features (Q5), in double; all of them, in or out of
  // the image (:127).  kd-tree nearest_point == exact minimum distance (kdtree/KDTree.cpp:154-233).
  const int n0 = n[stream];
  const float2 org = G.origin[cell];
  for (int a = threadIdx.x; a < accepted; a += 256) s_ok[a] = 1;
  __syncthreads();
  const double nn_lo = md2 * (1.0 - 1e-12), nn_hi = md2 * (1.0 + 1e-12);
  for (int t = threadIdx.x; t < accepted * n0; t += 256) {
    const int a = t / n0, i = t - a * n0;
    const double px = (double)((float)s_acc_x[a] + org.x), py = (double)((float)s_acc_y[a] + org.y);
    const double2 e = xy[(long long)stream * sp.cap + i];
    const double ddx = px - e.x, ddy = py - e.y;
    // :178-180 keep iff norm > minDist.  sqrt is monotone and correctly rounded, so away from a 1e-12 relative band
    // around minDist^2 the squared distance decides; only inside the band the square root itself is taken.
    const double d2 = ddx * ddx + ddy * ddy;
    bool far = d2 > nn_hi;
    if (!far && d2 >= nn_lo) {
      asm volatile("");  // keeps the rare block a block (no if-conversion of the double square root)
      far = sqrt(d2) > sp.min_dist;
    }
    if (!far) s_ok[a] = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int written = 0;
    for (int a = 0; a < accepted; a++)
      if (s_ok[a]) {
        // :169 kp.pt.x + grect.x in float with the unrounded Rect2f origin, then widened
        sel_xy[(long long)sc * sp.sel_cap + written] =
            make_double2((double)((float)s_acc_x[a] + org.x), (double)((float)s_acc_y[a] + org.y));
        written++;
      }
    sel_cnt[sc] = written;
  }
}

// Id assignment and upsert (Q6, Q7; ref :142, :173-184, :238).  One block per stream.
// New ids are the contiguous range [start, start+K) mod 2^64 in column-major cell order; an id that
// already exists is overwritten in place (unordered_map::operator[]), the rest are appended.
__global__ __launch_bounds__(256) void append_kernel(GridView G, const int* __restrict__ n_active,
                                                     const double2* __restrict__ sel_xy,
                                                     const int* __restrict__ sel_cnt, int sel_cap,
                                                     uint64_t* __restrict__ ids, double2* __restrict__ xy,
                                                     uint32_t* __restrict__ age, int* __restrict__ n, int cap,
                                                     uint64_t* __restrict__ max_feature_index,
                                                     uint64_t id_bias, int* __restrict__ status,
                                                     int* __restrict__ tslot) {
  extern __shared__ __align__(16) uint8_t smem[];
  int* s_off = reinterpret_cast<int*>(smem);       // [ncell+1] prefix of selected counts
  int* s_slot = s_off + (G.ncell + 1);             // [K] slot of each new feature (-1 = append)
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int stream = blockIdx.x;
  const int act = n_active[stream];
  if (threadIdx.x == 0) {
    int o = 0;
    for (int c = 0; c < G.ncell; c++) {
      s_off[c] = o;
      if (c < act) o += sel_cnt[stream * G.ncell + c];
    }
    s_off[G.ncell] = o;
  }
  __syncthreads();
  const int K = s_off[G.ncell];
  // first id issued: the stored index itself for the bucketing detector (Q6: the previous call's last
  // id is overwritten), stored index + 1 for the ANMS detector (FeatureDetectorANMS.cpp:198)
  const uint64_t start = max_feature_index[stream] + id_bias;
  const int n0 = n[stream];
  const long long o = (long long)stream * cap;
  for (int k = threadIdx.x; k < K; k += 256) s_slot[k] = -1;
  __syncthreads();
  for (int i = threadIdx.x; i < n0; i += 256) {
    const uint64_t d = ids[o + i] - start;  // wraps like the reference's uint64 counter
    if (d < (uint64_t)K) s_slot[(int)d] = i;
  }
  __syncthreads();
  // stable ranks of the appended ones
  if (threadIdx.x == 0) s_base = n0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c0 = 0; c0 < K; c0 += 256) {
    const int k = c0 + threadIdx.x;
    const bool app = (k < K) && (s_slot[k] < 0);
    const unsigned long long bal = __ballot(app);
    if (lane == 0) s_wave[wv] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int q = 0; q < wv; q++) off += s_wave[q];
    off += __popcll(bal & ((1ull << lane) - 1ull));
    if (app) s_slot[k] = off;
    __syncthreads();
    if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  // scatter: cell of k by scanning the prefix (ncell is small)
  for (int k = threadIdx.x; k < K; k += 256) {
    int c = 0;
    while (s_off[c + 1] <= k) c++;
    const int slot = s_slot[k];
    if (slot < cap) {
      ids[o + slot] = start + (uint64_t)k;
      xy[o + slot] = sel_xy[((long long)stream * G.ncell + c) * sel_cap + (k - s_off[c])];
      age[o + slot] = 1u;
      tslot[o + slot] = -1;  // new (or overwritten) position: no cached template
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int nn = s_base;
    if (nn > cap) {
      nn = cap;
      status[stream] = -3;  // VSF_E_CAPACITY
    }
    n[stream] = nn;
    max_feature_index[stream] = start + (uint64_t)K - 1ull;  // :238 feature_index - 1 (wraps)
  }
}

// Frontend step around the path (ref: src/vslam/frontend/KimeraFrontend.cpp:247-304), per stream:
// bearings of every tracked feature through the double-sphere model ({0,0,1} where the reference's
// Unproject throws), number of features of age >= counting_feature_age, keyframe decision, and the
// age increment of a keyframe.  All in double, in the operation order of the CPU code.
struct CameraDS {
  double fx, fy, cx, cy, xi, alpha;  // DoubleSphereCameraModel (src/vslam/data/Camera.hpp:134)
};
struct KeyframeParams {                  // KimeraFrontend::Parameter (KimeraFrontendParameter.cpp:11-14)
  double keyframe_interval;
  double minimum_keyframe_interval;
  unsigned int low_keyframe_feature_number;
  int counting_feature_age;
};

// basalt::DoubleSphereCamera<double>::unproject (Usenko et al., 3DV 2018, eq. 47-50)
__device__ __forceinline__ bool ds_unproject(const CameraDS& cam, double u, double v, double (&out)[3]) {
  const double mx = (u - cam.cx) / cam.fx;
  const double my = (v - cam.cy) / cam.fy;
  const double r2 = mx * mx + my * my;
  if (cam.alpha > 0.5 && r2 >= 1.0 / (2.0 * cam.alpha - 1.0)) return false;
  const double xi2_2 = cam.alpha * cam.alpha;
  const double xi1_2 = cam.xi * cam.xi;
  const double sqrt2 = sqrt(1.0 - (2.0 * cam.alpha - 1.0) * r2);
  const double norm2 = cam.alpha * sqrt2 + 1.0 - cam.alpha;
  const double mz = (1.0 - xi2_2 * r2) / norm2;
  const double mz2 = mz * mz;
  const double norm1 = mz2 + r2;
  const double sqrt1 = sqrt(mz2 + (1.0 - xi1_2) * r2);
  const double k = (mz * cam.xi + sqrt1) / norm1;
  out[0] = k * mx;
  out[1] = k * my;
  out[2] = k * mz - cam.xi;
  return true;
}

__global__ __launch_bounds__(256) void frontend_kernel(const double2* __restrict__ xy, uint32_t* __restrict__ age,
                                                       const int* __restrict__ n, int cap, CameraDS cam,
                                                       KeyframeParams kp, const double* __restrict__ dt_since_keyframe,
                                                       double* __restrict__ bearings, int* __restrict__ is_keyframe,
                                                       unsigned int* __restrict__ aged_out) {
  __shared__ unsigned int s_aged;
  __shared__ int s_kf;
  const int stream = blockIdx.x;
  const int n0 = n[stream];
  const long long o = (long long)stream * cap;
  if (threadIdx.x == 0) s_aged = 0u;
  __syncthreads();
  unsigned int aged = 0u;
  for (int i = threadIdx.x; i < n0; i += 256) {
    const double2 p = xy[o + i];
    double b[3];
    if (!ds_unproject(cam, p.x, p.y, b)) {  // :252-256
      b[0] = 0.0;
      b[1] = 0.0;
      b[2] = 1.0;
    }
    double* out = bearings + (o + i) * 3;
    out[0] = b[0];
    out[1] = b[1];
    out[2] = b[2];
    aged += age[o + i] >= (unsigned int)kp.counting_feature_age;  // :268 unsigned comparison
  }
  for (int m = 32; m >= 1; m >>= 1) aged += (unsigned int)__shfl_xor((int)aged, m, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_aged, aged);
  __syncthreads();
  if (threadIdx.x == 0) {
    const double dt = dt_since_keyframe[stream];
    bool low = s_aged < kp.low_keyframe_feature_number;  // :275-276
    if (dt < kp.minimum_keyframe_interval) low = false;  // :278-281
    s_kf = (low || dt > kp.keyframe_interval) ? 1 : 0;   // :291-293
    is_keyframe[stream] = s_kf;
    aged_out[stream] = s_aged;
  }
  __syncthreads();
  if (s_kf)
    for (int i = threadIdx.x; i < n0; i += 256) age[o + i] += 1u;  // :302-304
}

__global__ __launch_bounds__(256) void unproject_kernel(const double2* __restrict__ xy, int n, CameraDS cam,
                                                        double* __restrict__ bearings, uint8_t* __restrict__ valid) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double2 p = xy[i];
  double b[3];
  const bool ok = ds_unproject(cam, p.x, p.y, b);
  if (!ok) {
    b[0] = 0.0;
    b[1] = 0.0;
    b[2] = 1.0;
  }
  bearings[3 * i] = b[0];
  bearings[3 * i + 1] = b[1];
  bearings[3 * i + 2] = b[2];
  valid[i] = ok ? 1 : 0;
}

// Host-pointer API staging (vsf_track / vsf_detect / vsf_step, stream 0): the table crosses PCIe as ONE packed block
// per direction instead of four copies.  Layout: int hdr[16] {n, stats[4], status} | double2 xy[cap] | u64 ids[cap]
// | u32 age[cap].
__host__ __device__ inline size_t table_pack_bytes(int cap) { return 64 + (size_t)cap * 28; }

__global__ __launch_bounds__(256) void table_unpack_kernel(const uint8_t* __restrict__ pack, int cap,
                                                           uint64_t* __restrict__ ids, double2* __restrict__ xy,
                                                           uint32_t* __restrict__ age, int* __restrict__ n,
                                                           int* __restrict__ status, int* __restrict__ tslot) {
  const int* hdr = reinterpret_cast<const int*>(pack);
  const double2* pxy = reinterpret_cast<const double2*>(pack + 64);
  const uint64_t* pid = reinterpret_cast<const uint64_t*>(pack + 64 + (size_t)cap * 16);
  const uint32_t* pag = reinterpret_cast<const uint32_t*>(pack + 64 + (size_t)cap * 24);
  const int m = hdr[0];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
    xy[i] = pxy[i];
    ids[i] = pid[i];
    age[i] = pag[i];
    tslot[i] = -1;  // a table from the host: no cached template
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    n[0] = m;
    status[0] = 0;
  }
}

__global__ __launch_bounds__(256) void table_pack_kernel(const uint64_t* __restrict__ ids, const double2* __restrict__ xy,
                                                         const uint32_t* __restrict__ age, const int* __restrict__ n,
                                                         const int* __restrict__ stats, const int* __restrict__ status,
                                                         int cap, uint8_t* __restrict__ pack) {
  int* hdr = reinterpret_cast<int*>(pack);
  double2* pxy = reinterpret_cast<double2*>(pack + 64);
  uint64_t* pid = reinterpret_cast<uint64_t*>(pack + 64 + (size_t)cap * 16);
  uint32_t* pag = reinterpret_cast<uint32_t*>(pack + 64 + (size_t)cap * 24);
  const int m = min(n[0], cap);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
    pxy[i] = xy[i];
    pid[i] = ids[i];
    pag[i] = age[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    hdr[0] = n[0];
    hdr[1] = stats[0];
    hdr[2] = stats[1];
    hdr[3] = stats[2];
    hdr[4] = stats[3];
    hdr[5] = status[0];
  }
}

// cv::cvtColor(BGR2GRAY) on 8-bit interleaved frames of 3 or 4 channels (what both detectors do to a multi-channel
// frame, FeatureDetectorShiTomasiBucketing.cpp:105-110, FeatureDetectorANMS.cpp:98-99): cv's fixed-point weights, (1868 B + 9617 G + 4899 R +
// 2^13) >> 14.  Four pixels per thread, one dword store; z = stream.  HBM-bound (cn bytes read, one written).
__global__ __launch_bounds__(256) void bgr2gray_kernel(const uint8_t* __restrict__ src, long long src_sstride,
                                                       int src_pitch, int cn, int w, int h,
                                                       uint8_t* __restrict__ dst, long long dst_sstride, int dst_pitch) {
  const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x4 >= w || y >= h) return;
  const uint8_t* s = src + (long long)blockIdx.z * src_sstride + (long long)y * src_pitch + (long long)x4 * cn;
  uint8_t* d = dst + (long long)blockIdx.z * dst_sstride + (long long)y * dst_pitch + x4;
  unsigned int g[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (x4 + k < w) g[k] = ((unsigned int)s[k * cn] * 1868u + (unsigned int)s[k * cn + 1] * 9617u +
                            (unsigned int)s[k * cn + 2] * 4899u + (1u << 13)) >> 14;
  if (x4 + 3 < w && ((reinterpret_cast<uintptr_t>(d) & 3) == 0)) {
    *reinterpret_cast<unsigned int*>(d) = g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24);
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (x4 + k < w) d[k] = (uint8_t)g[k];
  }
}

// Shader-clock probe (diagnostic, bench.py): one lane spins for `spin_ticks` of the constant-rate counter
// (s_memrealtime, hipDeviceAttributeWallClockRate) and reports how far the shader cycle counter (s_memtime)
// moved meanwhile.  Run on a second stream while the step's kernels run, it gives the clock the chip actually
// holds under that load (DVFS), which the VALU issue figures of DESIGN.md section 5 are stated in.  The loop is
// bounded by an iteration count as well, so the wave ends whatever the counters do.
__global__ __launch_bounds__(64) void sclk_probe_kernel(unsigned long long* __restrict__ out,
                                                        unsigned long long spin_ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  unsigned int n = 0;
  while (r1 - r0 < spin_ticks && n < (1u << 28)) {
    __builtin_amdgcn_s_sleep(8);
    r1 = __builtin_amdgcn_s_memrealtime();
    n++;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[0] = t1 - t0;
  out[1] = r1 - r0;
  out[2] = n;
}

// Fixed-size track records for the end-of-step collective over xGMI: 24 bytes,
// padded to the table capacity so every rank contributes the same byte count.
struct TrackRecord {
  uint64_t id;
  float x, y;
  uint32_t age;
  uint32_t valid;
};

__global__ __launch_bounds__(256) void export_kernel(const uint64_t* __restrict__ ids,
                                                     const double2* __restrict__ xy,
                                                     const uint32_t* __restrict__ age, const int* __restrict__ n,
                                                     int cap, long long total, TrackRecord* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int stream = (int)(i / cap), k = (int)(i - (long long)stream * cap);
  TrackRecord r;
  if (k < n[stream]) {
    const double2 p = xy[i];
    r.id = ids[i];
    r.x = (float)p.x;
    r.y = (float)p.y;
    r.age = age[i];
    r.valid = 1u;
  } else {
    r.id = 0;
    r.x = r.y = 0.f;
    r.age = 0u;
    r.valid = 0u;
  }
  out[i] = r;
}

}  // namespace vsf
#include "vsf_anms.hpp"
#include "vsf_lssd.hpp"
