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
#define VSF_LK_WAVES 7  // waves per SIMD the LK kernels are register-limited to (window <= 21): 72 registers
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

struct PyrView {
  const uint8_t* base[MAXL];   // level l of stream 0
  long long sstride[MAXL];     // bytes between consecutive streams at level l
  int pitch[MAXL], w[MAXL], h[MAXL];
  int nlevels;                 // effective maxLevel + 1
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
template <bool OUT16>
__global__ __launch_bounds__(256) void pyr_down_dword_kernel(const uint8_t* __restrict__ src, long long src_sstride,
                                                             int src_pitch, int sw, int sh,
                                                             void* __restrict__ dst, long long dst_sstride,
                                                             int dst_pitch, int dw, int dh) {
  __shared__ uint32_t s_src[PD_SH * PD_SD];
  __shared__ uint32_t s_h[PD_SH * (PD_TW / 2 + 2)];
  constexpr int HP = PD_TW / 2 + 2;  // pitch of s_h in dwords (even: 8-byte reads stay aligned)
  const int stream = blockIdx.z;
  const uint8_t* S = src + (long long)stream * src_sstride;
  const int ox = blockIdx.x * PD_TW, oy = blockIdx.y * PD_TH;
  const int gx_base = 2 * ox - 4, sy0 = 2 * oy - 2;
  for (int i = threadIdx.x; i < PD_SH * PD_SD; i += 256) {
    const int y = i / PD_SD, j = i - y * PD_SD;
    int gy = reflect101(sy0 + y, sh);
    gy = min(max(gy, 0), sh - 1);  // rows that only feed outputs below the image
    const int gx = gx_base + 4 * j;
    uint32_t v = 0;
    if (gx >= 0 && gx + 4 <= sw) v = *reinterpret_cast<const uint32_t*>(S + (long long)gy * src_pitch + gx);
    s_src[i] = v;
  }
  __syncthreads();
  if (gx_base + 2 < 0 || gx_base + PD_SW + 2 > sw) {  // block-uniform: the tile crosses the left / right image edge
    uint8_t* sb = reinterpret_cast<uint8_t*>(s_src);
    for (int i = threadIdx.x; i < PD_SH * 8; i += 256) {
      const int y = i >> 3, k = i & 7;
      // candidates: tile columns 2, 3 (gx = 2ox-2, 2ox-1) on the left, and the first four past sw on the right
      const int t = k < 2 ? 2 + k : (sw - gx_base) + (k - 2);
      const int gx = gx_base + t;
      if (t >= 2 && t < 2 + PD_SW && (gx < 0 || gx >= sw) && k < 6) {
        const int tr = min(max(reflect101(gx, sw) - gx_base, 0), 4 * PD_SD - 1);
        sb[y * 4 * PD_SD + t] = sb[y * 4 * PD_SD + tr];
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
  __syncthreads();
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
}

// ------------------------------------------------------------------------------------------------
// Pyramidal LK (cv::calcOpticalFlowPyrLK scalar semantics), one wavefront per
// feature.  Lane l owns window row r = l / nseg and the seglen consecutive pixels starting at
// (l % nseg) * seglen, so the template (I, Ix, Iy as int16 values) lives in registers for the whole
// level and only the J patch is re-read per iteration, from an LDS-staged region.  The Scharr
// derivative is fused: the wave stages the (win+3)^2 patch of I in LDS and derives the (win+1)^2
// gradient patch there; no derivative image is ever written to HBM.
// Sums of products are accumulated exactly (int32 per lane, int64 across the wave) and converted
// to float32 once: order independent, equal to the oracle's ORC_ACC_EXACT mode bit for bit.
// ------------------------------------------------------------------------------------------------
// ---- exact wave-wide integer sums -------------------------------------------------------------
// Every lane ends up holding the sum of its 16-lane row; DPP adds, no LDS traffic.
__device__ __forceinline__ int row_sum_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
  return v;
}
// Wave-uniform (scalar) sum of a per-lane int32; the caller guarantees no overflow.
__device__ __forceinline__ int wave_sum_i32(int v) {
  v = row_sum_dpp(v);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}
// float32(sum over the wave of p), correctly rounded from the EXACT integer sum: the per-lane value
// is split into a 16-bit low limb and a signed high limb so that both 64-lane sums fit int32; the
// recombination in double is exact (< 2^53) and rounds to float32 once.
__device__ __forceinline__ float wave_sum_exact_f32(int p) {
  const int slo = wave_sum_i32(p & 0xffff);
  const int shi = wave_sum_i32(p >> 16);
  return (float)((double)shi * 65536.0 + (double)slo);
}
// The same for N values at once.  Each |p| must be < 2^29 (true for every sum here: <= 16 pixels x
// 8160 x 4080), so the first two DPP steps (sums over quads) still fit int32; only then is each value
// split into the two limbs, which halves the DPP adds of those steps.  The limb chains are
// interleaved so that no chain waits on its own previous step.  The recombination takes the int32
// route (one v_cvt_f32_i32, exact round-to-nearest-even) whenever the total fits 32 bits and the
// double route otherwise; both round the EXACT integer sum once.
template <int N>
__device__ __forceinline__ void wave_sums_exact_f32(const int (&p)[N], float (&out)[N]) {
  int q[N];
#pragma unroll
  for (int i = 0; i < N; i++) q[i] = p[i] + __builtin_amdgcn_update_dpp(0, p[i], 0xB1, 0xf, 0xf, false);
#pragma unroll
  for (int i = 0; i < N; i++) q[i] += __builtin_amdgcn_update_dpp(0, q[i], 0x4E, 0xf, 0xf, false);
  int v[2 * N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    v[2 * i] = q[i] & 0xffff;
    v[2 * i + 1] = q[i] >> 16;
  }
#pragma unroll
  for (int i = 0; i < 2 * N; i++) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0x141, 0xf, 0xf, false);
#pragma unroll
  for (int i = 0; i < 2 * N; i++) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0x140, 0xf, 0xf, false);
#pragma unroll
  for (int i = 0; i < N; i++) {
    const int slo = __builtin_amdgcn_readlane(v[2 * i], 0) + __builtin_amdgcn_readlane(v[2 * i], 16) +
                    __builtin_amdgcn_readlane(v[2 * i], 32) + __builtin_amdgcn_readlane(v[2 * i], 48);
    const int shi = __builtin_amdgcn_readlane(v[2 * i + 1], 0) + __builtin_amdgcn_readlane(v[2 * i + 1], 16) +
                    __builtin_amdgcn_readlane(v[2 * i + 1], 32) + __builtin_amdgcn_readlane(v[2 * i + 1], 48);
    // slo < 2^22 here, so shi in [-32768, 32766] keeps shi*65536 + slo inside int32
    if (shi >= -32768 && shi <= 32766)
      out[i] = (float)(shi * 65536 + slo);
    else
      out[i] = (float)((double)shi * 65536.0 + (double)slo);
  }
}

// v_readfirstlane the compiler cannot look through: the result is an SGPR, and everything computed
// from it stays on the scalar unit instead of being folded back into vector instructions.
// gfx940/950 need one wait state between a VALU write of a VGPR and a v_readlane of it, and two
// between a VALU write of an SGPR and a VALU read of it; the compiler inserts them for its own
// instructions but cannot see inside an asm statement, so they are part of it (measured: without
// the leading s_nop the lane read returns the register's previous value).
__device__ __forceinline__ int to_sgpr(int v) {
  int s;
  asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 1" : "=s"(s) : "v"(v));
  return s;
}
__device__ __forceinline__ int to_sgpr(float v) {
  int s;
  asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 1" : "=s"(s) : "v"(v));
  return s;
}

// Several lane reads behind ONE pair of wait states (the leading one covers the first read, the later reads
// are further away from their producers; the trailing two cover the last SGPR written).
__device__ __forceinline__ void to_sgpr3(float a, float b, float c, int& sa, int& sb, int& sc) {
  asm volatile(
      "s_nop 0\n\tv_readfirstlane_b32 %0, %3\n\tv_readfirstlane_b32 %1, %4\n\tv_readfirstlane_b32 %2, %5\n\ts_nop 1"
      : "=&s"(sa), "=&s"(sb), "=&s"(sc)
      : "v"(a), "v"(b), "v"(c));
}
__device__ __forceinline__ void to_sgpr2(int a, int b, int& sa, int& sb) {
  asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3\n\ts_nop 1"
               : "=&s"(sa), "=&s"(sb)
               : "v"(a), "v"(b));
}
__device__ __forceinline__ void to_sgpr4(int a, int b, int c, int d, int& sa, int& sb, int& sc, int& sd) {
  asm volatile(
      "s_nop 0\n\tv_readfirstlane_b32 %0, %4\n\tv_readfirstlane_b32 %1, %5\n\tv_readfirstlane_b32 %2, %6\n\t"
      "v_readfirstlane_b32 %3, %7\n\ts_nop 1"
      : "=&s"(sa), "=&s"(sb), "=&s"(sc), "=&s"(sd)
      : "v"(a), "v"(b), "v"(c), "v"(d));
}

// The same sums through the LDS ATOMIC unit.  The kernel is VALU-issue bound while the LDS pipe idles, so
// only the first NRAW butterfly steps run as DPP adds (the values still fit int32: NRAW = 3 needs
// |p| < 2^28, NRAW = 2 needs |p| < 2^29); then one lane of every 2^NRAW adds its partial sum as a
// 64-bit integer into an LDS accumulator (lane 0 stores, the others ds_add_u64: LDS operations of one
// wave execute in issue order), and every lane reads the exact total back.  The total is converted
// to float32 with ONE rounding: v_cvt_f32_i32 when it fits 32 bits, through double otherwise.
template <int N, int NRAW>
__device__ __forceinline__ void wave_sums_exact_lds(const int (&p)[N], float (&out)[N], long long* acc, int lane) {
  int q[N];
#pragma unroll
  for (int i = 0; i < N; i++) q[i] = p[i] + __builtin_amdgcn_update_dpp(0, p[i], 0xB1, 0xf, 0xf, false);
#pragma unroll
  for (int i = 0; i < N; i++) q[i] += __builtin_amdgcn_update_dpp(0, q[i], 0x4E, 0xf, 0xf, false);
  if constexpr (NRAW == 3) {
#pragma unroll
    for (int i = 0; i < N; i++) q[i] += __builtin_amdgcn_update_dpp(0, q[i], 0x141, 0xf, 0xf, false);
  }
  constexpr int LM = (1 << NRAW) - 1;
  // LDS byte address of the accumulators.  The three steps are written as inline asm: left to the
  // compiler, the atomic is "optimised" into a serial v_readlane loop over the active lanes, which
  // is exactly the VALU work this routine exists to avoid.
  const unsigned int a32 =
      (unsigned int)(size_t)(__attribute__((address_space(3))) void*)acc;
  static_assert(N == 2 || N == 3, "two or three sums");
  long long q64[N];
#pragma unroll
  for (int i = 0; i < N; i++) q64[i] = (long long)q[i];
  if (lane == 0) {
    if constexpr (N == 2)
      asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:8" ::"v"(a32), "v"(q64[0]), "v"(q64[1])
                   : "memory");
    else
      asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:8\n\tds_write_b64 %0, %3 offset:16" ::"v"(a32),
                   "v"(q64[0]), "v"(q64[1]), "v"(q64[N - 1])
                   : "memory");
  }
  // convergent barriers keep the store, the adds and the loads in this order in the instruction
  // stream (per lane they are mutually exclusive, so nothing else orders them for the compiler)
  __builtin_amdgcn_wave_barrier();
  if (lane != 0 && (lane & LM) == 0) {
    if constexpr (N == 2)
      asm volatile("ds_add_u64 %0, %1\n\tds_add_u64 %0, %2 offset:8" ::"v"(a32), "v"(q64[0]), "v"(q64[1]) : "memory");
    else
      asm volatile("ds_add_u64 %0, %1\n\tds_add_u64 %0, %2 offset:8\n\tds_add_u64 %0, %3 offset:16" ::"v"(a32),
                   "v"(q64[0]), "v"(q64[1]), "v"(q64[N - 1])
                   : "memory");
  }
  __builtin_amdgcn_wave_barrier();
  long long t[N];
  if constexpr (N == 2) {
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1])
                 : "v"(a32)
                 : "memory");
  } else {
    asm volatile(
        "ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:8\n\tds_read_b64 %2, %3 offset:16\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[N - 1])
        : "v"(a32)
        : "memory");
  }
  int los[N], his[N];
  if constexpr (N == 2) {
    to_sgpr4((int)(unsigned int)(unsigned long long)t[0], (int)(t[0] >> 32), (int)(unsigned int)(unsigned long long)t[1],
             (int)(t[1] >> 32), los[0], his[0], los[1], his[1]);
  } else {
    to_sgpr4((int)(unsigned int)(unsigned long long)t[0], (int)(t[0] >> 32), (int)(unsigned int)(unsigned long long)t[1],
             (int)(t[1] >> 32), los[0], his[0], los[1], his[1]);
    to_sgpr2((int)(unsigned int)(unsigned long long)t[N - 1], (int)(t[N - 1] >> 32), los[N - 1], his[N - 1]);
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    // scalar from here on: the range test and the recombination cost no vector instruction
    const int lo = los[i], hi = his[i];
    if (hi == (lo >> 31))
      out[i] = (float)lo;
    else
      out[i] = (float)((double)hi * 4294967296.0 + (double)(unsigned int)lo);
  }
}

// The four weights of a wave-uniform sub-pixel offset, as scalars.  cvRound(x * 2^14), x in [0, 1]:
// scaling by 2^14 is exact, so it is applied to one factor before the product; adding 1.5 * 2^23
// rounds the product to an integer with ties to even (the parity of the sum is the parity of the
// integer part), which is what cvRound / rint do; the integer is then the low mantissa bits.
__device__ __forceinline__ void bilinear_weights_uniform(float a, float b, int& w00, int& w01, int& w10, int& w11) {
  const float MAGIC = 12582912.f;  // 1.5 * 2^23
  const float a1 = 1.f - a;
  const float b1 = (1.f - b) * (float)(1 << W_BITS), b0 = b * (float)(1 << W_BITS);
  int s00, s01, s10;
  to_sgpr3(a1 * b1 + MAGIC, a * b1 + MAGIC, a1 * b0 + MAGIC, s00, s01, s10);
  w00 = s00 - 0x4B400000;
  w01 = s01 - 0x4B400000;
  w10 = s10 - 0x4B400000;
  w11 = (1 << W_BITS) - w00 - w01 - w10;
}

__device__ __forceinline__ void bilinear_weights(float a, float b, int& w00, int& w01, int& w10, int& w11) {
  w00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
  w01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
  w10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
  w11 = (1 << W_BITS) - w00 - w01 - w10;
}

#define VSF_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

typedef short vsf_i16x2 __attribute__((ext_vector_type(2)));

// a.lo*b.lo + a.hi*b.hi + c on packed SIGNED 16-bit pairs (v_dot2_i32_i16).  Signed because the
// fourth bilinear weight, 2^14 - w00 - w01 - w10, can come out as -1 after rounding.
__device__ __forceinline__ int sdot2(unsigned int a, unsigned int b, int c) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(vsf_i16x2, a), __builtin_bit_cast(vsf_i16x2, b), c, false);
}

// (lo & 0xffff) | (hi << 16) in one v_perm_b32
__device__ __forceinline__ unsigned int pack_lo16(int lo, int hi) {
  return __builtin_amdgcn_perm((unsigned int)hi, (unsigned int)lo, 0x05040100u);
}

// Reads NA+... dwords of an LDS byte row starting at byte offset `boff` and returns, in A[], the byte
// stream re-aligned so that A[0] byte 0 is the byte at `boff`.
template <int NA>
__device__ __forceinline__ void lds_stream(const uint8_t* row, int boff, unsigned int (&A)[NA]) {
  const unsigned int* w = reinterpret_cast<const unsigned int*>(row) + (boff >> 2);
  const unsigned int o = (unsigned int)boff & 3u;
  unsigned int d[NA + 1];
#pragma unroll
  for (int i = 0; i < NA + 1; i++) d[i] = w[i];
#pragma unroll
  for (int i = 0; i < NA; i++) A[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], o);
}

// Two consecutive LDS rows (pitch a multiple of 4, rows dword aligned) read from the byte address p
// of the first one: one address split serves both rows.
template <int NA>
__device__ __forceinline__ void lds_stream_at(const uint8_t* p, int pitch, unsigned int (&A0)[NA],
                                              unsigned int (&A1)[NA]) {
  const unsigned int o = (unsigned int)reinterpret_cast<uintptr_t>(p) & 3u;
  const unsigned int* w0 = reinterpret_cast<const unsigned int*>(p - o);
  const unsigned int* w1 = reinterpret_cast<const unsigned int*>(p - o + pitch);
  unsigned int d0[NA + 1], d1[NA + 1];
#pragma unroll
  for (int i = 0; i < NA + 1; i++) d0[i] = w0[i];
#pragma unroll
  for (int i = 0; i < NA + 1; i++) d1[i] = w1[i];
#pragma unroll
  for (int i = 0; i < NA; i++) A0[i] = __builtin_amdgcn_alignbyte(d0[i + 1], d0[i], o);
#pragma unroll
  for (int i = 0; i < NA; i++) A1[i] = __builtin_amdgcn_alignbyte(d1[i + 1], d1[i], o);
}

// (byte k) | (byte k+1) << 16 of a dword stream (v_perm_b32; selector 0x0c = zero byte)
template <int K, int NA>
__device__ __forceinline__ unsigned int byte_pair(const unsigned int (&A)[NA]) {
  constexpr int j = K >> 2, m = K & 3;
  if constexpr (m < 3) {
    return __builtin_amdgcn_perm(0u, A[j], 0x0c000c00u | (unsigned int)m | ((unsigned int)(m + 1) << 16));
  } else {
    return __builtin_amdgcn_perm(A[j + 1], A[j], 0x0c040c03u);
  }
}

template <int K, int N, int NA, int C0>
struct BilinearRow {
  // out[k] = C0 + sum over the 2x2 neighbourhood, for k = K..N-1 (compile-time recursion keeps every
  // index a constant, so the arrays stay in registers).  C0 carries DESCALE's rounding constant.
  static __device__ __forceinline__ void run(const unsigned int (&A0)[NA], const unsigned int (&A1)[NA],
                                             unsigned int W0, unsigned int W1, int (&out)[N]) {
    out[K] = sdot2(byte_pair<K, NA>(A1), W1, sdot2(byte_pair<K, NA>(A0), W0, C0));
    if constexpr (K + 1 < N) BilinearRow<K + 1, N, NA, C0>::run(A0, A1, W0, W1, out);
  }
};

typedef unsigned short vsf_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int udot2(unsigned int a, unsigned int b, int c) {
  return (int)__builtin_amdgcn_udot2(__builtin_bit_cast(vsf_u16x2, a), __builtin_bit_cast(vsf_u16x2, b),
                                     (unsigned int)c, false);
}

// Unsigned twin of BilinearRow for the common case of four non-negative weights: v_dot2_u32_u16 is a
// three-operand instruction, so the accumulator constant needs no v_mov per pixel (the signed form
// on gfx950 is the two-operand v_dot2c_i32_i16).
template <int K, int N, int NA, int C0>
struct BilinearRowU {
  static __device__ __forceinline__ void run(const unsigned int (&A0)[NA], const unsigned int (&A1)[NA],
                                             unsigned int W0, unsigned int W1, int (&out)[N]) {
    out[K] = udot2(byte_pair<K, NA>(A1), W1, udot2(byte_pair<K, NA>(A0), W0, C0));
    if constexpr (K + 1 < N) BilinearRowU<K + 1, N, NA, C0>::run(A0, A1, W0, W1, out);
  }
};

template <int N, int NA, int C0>
__device__ __forceinline__ void bilinear_row(bool nonneg, const unsigned int (&A0)[NA], const unsigned int (&A1)[NA],
                                             unsigned int W0, unsigned int W1, int (&out)[N]) {
  if (nonneg)
    BilinearRowU<0, N, NA, C0>::run(A0, A1, W0, W1, out);
  else
    BilinearRow<0, N, NA, C0>::run(A0, A1, W0, W1, out);
}

// LDS carve-up per wave and the geometry of the two staged regions
struct RegionGeom {
  int ndw;    // staged dwords per row
  int inv;    // ceil(2^16 / ndw): (t * inv) >> 16 == t / ndw for every task index used
  int rows;   // staged rows
  int pitch;  // LDS bytes per row: 4 * (ndw | 1), odd in dwords so that window rows hit distinct banks
};

struct LkSmem {
  RegionGeom gi;  // template patch of I: (win+3) rows, covers win+3 columns at any 4-byte phase
  RegionGeom gj;  // J region: window + margin
  int d_pitch;    // gradient patch (win+1)^2, packed (Ix | Iy<<16) int32 (generic path only)
  int jp_pitch;   // J region in PAIR layout: dwords per row (4 * gj.ndw pairs + 2 of padding)
  int off_j, off_d;
  int off_acc;   // 4 x int64 accumulators of the exact sums
  int off_mask;  // per lane: NP dwords, 16-bit masks of the window pixels the lane owns (pair-packed)
  int total;
};

// Pixels per lane: the window row is cut into nseg = 64 / win segments of seglen pixels.
__host__ __device__ inline int lk_seglen(int win) {
  const int nseg = 64 / win;
  return (win + nseg - 1) / nseg;
}
// Kernel instantiation (compile-time pixels per lane) that serves a given seglen.
__host__ __device__ inline int lk_seg_template(int seglen) { return seglen <= 7 ? 7 : (seglen <= 8 ? 8 : 16); }

__host__ __device__ inline RegionGeom vsf_region_geom(int cols_needed, int rows) {
  RegionGeom g;
  g.ndw = (cols_needed + 3 + 3) / 4;  // any 4-byte phase of the origin
  g.inv = (65536 + g.ndw - 1) / g.ndw;
  g.rows = rows;
  g.pitch = 4 * (g.ndw | 1);
  return g;
}

__host__ __device__ inline LkSmem lk_smem_layout(int win, int jmargin) {
  LkSmem s;
  s.gi = vsf_region_geom(win + 3, win + 3);
  s.gj = vsf_region_geom(win + 1 + 2 * jmargin, win + 1 + 2 * jmargin);
  s.d_pitch = win + 1;
  const int si = s.gi.pitch * s.gi.rows + 32;  // +slack: dword-granular readers may run past a row end
  s.off_j = (si + 15) & ~15;
  // The J region is kept as pixel PAIRS, one dword per pixel: byte(x) | byte(x + 1) << 16, the operand v_dot2
  // takes, so the iterations spend no v_perm / v_alignbyte on it.  The gradient patch of the border path (sD)
  // is dead before the first J staging of a level and shares the space.
  s.jp_pitch = 4 * s.gj.ndw + 2;
  const int sj = s.jp_pitch * 4 * s.gj.rows + 64, sd = s.d_pitch * (win + 1) * 4;
  s.off_d = s.off_j;
  s.off_acc = s.off_j + (((sj > sd ? sj : sd) + 15) & ~15);
  s.off_mask = s.off_acc + 32;
  const int np = (lk_seg_template(lk_seglen(win)) + 1) / 2;
  s.total = s.off_mask + 64 * 4 * np;
#ifdef VSF_LDS_PAD  // occupancy experiments
  s.total += VSF_LDS_PAD;
#endif
  return s;
}

// A region [xa, xa + 4*ndw) x [y0, y0 + rows) can be fetched as aligned dwords when it lies inside
// the level and rows are dword aligned; otherwise it is assembled byte by byte with REFLECT_101.
__device__ __forceinline__ bool region_is_fast(const uint8_t* src, int sp, int sw, int sh, int xa, int y0,
                                               const RegionGeom& g) {
  return xa >= 0 && y0 >= 0 && xa + 4 * g.ndw <= sw && y0 + g.rows <= sh && (sp & 3) == 0 &&
         ((reinterpret_cast<uintptr_t>(src) & 3) == 0);
}

// Staging map (no division): lane -> dword column xd = lane % C and row yl = lane / C, with C = 8 when
// the region is at most 8 dwords wide and 16 otherwise; task i covers row yl + i * (64 / C).
// Issues the aligned dword loads of a fast region into registers (no wait).  The source pointer is
// wave-uniform, so each load is a scalar base plus a 32-bit lane offset.
template <int N>
__device__ __forceinline__ void region_load_regs(unsigned int (&reg)[N], const uint8_t* __restrict__ src, int sp,
                                                 int xa, int y0, const RegionGeom& g, int lane) {
  const int cl = g.ndw <= 8 ? 3 : 4;
  const int xd = lane & ((1 << cl) - 1), yl = lane >> cl, ystep = 64 >> cl;
  const unsigned int* s = reinterpret_cast<const unsigned int*>(src + (long long)y0 * sp + xa);
  const unsigned int spw = (unsigned int)sp >> 2;
  const unsigned int off = (unsigned int)yl * spw + (unsigned int)xd, step = (unsigned int)ystep * spw;
  const bool xin = xd < g.ndw;
#pragma unroll
  for (int i = 0; i < N; i++) {
    reg[i] = 0u;
    if (xin && yl + i * ystep < g.rows) reg[i] = s[off + (unsigned int)i * step];
  }
}

template <int N>
__device__ __forceinline__ void region_store_regs(uint8_t* dst, const unsigned int (&reg)[N], const RegionGeom& g,
                                                  int lane) {
  const int cl = g.ndw <= 8 ? 3 : 4;
  const int xd = lane & ((1 << cl) - 1), yl = lane >> cl, ystep = 64 >> cl;
  unsigned int* d = reinterpret_cast<unsigned int*>(dst);
  const int pw = g.pitch >> 2;
  const int off = yl * pw + xd, step = ystep * pw;
  const bool xin = xd < g.ndw;
#pragma unroll
  for (int i = 0; i < N; i++) {
    if (xin && yl + i * ystep < g.rows) d[off + i * step] = reg[i];
  }
}

// Border path: values identical to OpenCV's reflect-padded pyramid level.
template <int N>
__device__ __forceinline__ void region_stage_slow(uint8_t* dst, const uint8_t* __restrict__ src, int sp, int sw,
                                                  int sh, int xa, int y0, const RegionGeom& g, int lane) {
  const int cl = g.ndw <= 8 ? 3 : 4;
  const int xd = lane & ((1 << cl) - 1), yl = lane >> cl, ystep = 64 >> cl;
  unsigned int* d = reinterpret_cast<unsigned int*>(dst);
  const int pw = g.pitch >> 2;
  if (xd < g.ndw) {
    for (int y = yl; y < g.rows; y += ystep) {
      const uint8_t* sr = src + (long long)reflect101(y0 + y, sh) * sp;
      unsigned int v = 0u;
#pragma unroll
      for (int b = 0; b < 4; b++) v |= (unsigned int)sr[reflect101(xa + 4 * xd + b, sw)] << (8 * b);
      d[y * pw + xd] = v;
    }
  }
}

// ---- pair layout (J region): LDS dword (row, x) = byte(x) | byte(x + 1) << 16
__device__ __forceinline__ void pairs_of_dword(unsigned int d0, unsigned int d1, unsigned int (&P)[4]) {
  P[0] = __builtin_amdgcn_perm(0u, d0, 0x0c010c00u);
  P[1] = __builtin_amdgcn_perm(0u, d0, 0x0c020c01u);
  P[2] = __builtin_amdgcn_perm(0u, d0, 0x0c030c02u);
  P[3] = __builtin_amdgcn_perm(d1, d0, 0x0c040c03u);
}

// Registers of region_load_regs -> pair layout.  The dword to the right is the next lane's (the staging map puts
// consecutive columns on consecutive lanes of a 16-lane DPP row); the last column's fourth pair would need a
// byte outside the region and is never read (the iterations restage before the window reaches it).
template <int N>
__device__ __forceinline__ void region_store_pairs(unsigned int* dst, const unsigned int (&reg)[N], const RegionGeom& g,
                                                   int pitch_dw, int lane) {
  const int cl = g.ndw <= 8 ? 3 : 4;
  const int xd = lane & ((1 << cl) - 1), yl = lane >> cl, ystep = 64 >> cl;
  const int off = yl * pitch_dw + 4 * xd, step = ystep * pitch_dw;
  const bool xin = xd < g.ndw;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const unsigned int right = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)reg[i], 0x101, 0xf, 0xf, true);  // row_shl:1
    unsigned int P[4];
    pairs_of_dword(reg[i], right, P);
    if (xin && yl + i * ystep < g.rows) {
      unsigned int* d = dst + off + i * step;
      d[0] = P[0];
      d[1] = P[1];
      d[2] = P[2];
      d[3] = P[3];
    }
  }
}

template <int N>
__device__ __forceinline__ void region_stage_slow_pairs(unsigned int* dst, const uint8_t* __restrict__ src, int sp,
                                                        int sw, int sh, int xa, int y0, const RegionGeom& g,
                                                        int pitch_dw, int lane) {
  const int cl = g.ndw <= 8 ? 3 : 4;
  const int xd = lane & ((1 << cl) - 1), yl = lane >> cl, ystep = 64 >> cl;
  if (xd < g.ndw) {
    for (int y = yl; y < g.rows; y += ystep) {
      const uint8_t* sr = src + (long long)reflect101(y0 + y, sh) * sp;
      unsigned int b[5];
#pragma unroll
      for (int k = 0; k < 5; k++) b[k] = (unsigned int)sr[reflect101(xa + 4 * xd + k, sw)];
      unsigned int* d = dst + y * pitch_dw + 4 * xd;
#pragma unroll
      for (int k = 0; k < 4; k++) d[k] = b[k] | (b[k + 1] << 16);
    }
  }
}

// Tracks one point from pyramid A (template, derivatives) to pyramid B.  Wave-uniform control flow.
// Returns the OpenCV status (true = found); out = nextPts.
// SEG = compile-time pixels per lane (>= prm.seglen).  Gradients are held pair-packed
// (IxP[k2] = Ix[2k2] | Ix[2k2+1] << 16, same for Iy), so that every sum of products is a v_dot2.
template <int SEG>
__device__ bool lk_track_point(const PyrView& A, const PyrView& B, int stream, float ptx, float pty,
                               float& outx, float& outy, const LkParams& prm, uint8_t* smem,
                               const LkSmem& L) {
  constexpr int NA = (SEG + 1 + 3) / 4;     // aligned dwords holding SEG+1 bytes
  constexpr int NI = (SEG <= 8) ? 3 : 9;    // I-patch staging tasks per lane (win <= 21: 24 rows / 8; <= 31: 34 / 4)
  constexpr int NJ = (SEG <= 8) ? 4 : 10;   // J-region staging tasks per lane (jmargin 3: 28 rows / 8; 38 / 4)
  constexpr int NP = (SEG + 1) / 2;         // pixel pairs per lane
  constexpr int NRAW = (SEG <= 8) ? 3 : 2;  // raw DPP steps of the exact sums (see wave_sums_exact_lds)
  const int lane0 = threadIdx.x & 63;
  const int win = prm.win;
  uint8_t* sI = smem;
  unsigned int* sJ = reinterpret_cast<unsigned int*>(smem + L.off_j);
  int* sD = reinterpret_cast<int*>(smem + L.off_d);
  long long* sAcc = reinterpret_cast<long long*>(smem + L.off_acc);
  const float half = (float)(win - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (float)(1 << 20);
  const int maxLevel = A.nlevels - 1;
  bool status = true;
  float nx = 0.f, ny = 0.f;  // nextPts

  int jx0 = 0, jy0 = 0;

  for (int level = maxLevel; level >= 0; level--) {
    // An opaque zero, new in every level: whatever derives from (lane0 + oz) cannot be hoisted out of
    // the level loop.  Hoisted, these dozen addresses and masks stay live through the whole kernel and
    // push the 96-register build into scratch spills; recomputing them costs ~15 instructions a level.
    int oz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
    const int lane = lane0 + oz;
    const int r = (lane * prm.nseg_inv) >> 16;  // lane / nseg
    const int x0 = (lane - r * prm.nseg) * prm.seglen;
    const bool owner = (r < win) && (x0 < win);                // lane owns at least one window pixel
    const int rr = owner ? r : 0, xx0 = owner ? x0 : 0;        // idle lanes read a valid (unused) place
    const unsigned int* sMask = reinterpret_cast<const unsigned int*>(smem + L.off_mask) + lane * NP;
    const int rowI = __mul24(rr, L.gi.pitch);
    const unsigned int* rowJ0 = sJ + __mul24(rr, L.jp_pitch) + xx0;
    const float lscale = 1.f / (float)(1 << level);
    float px = ptx * lscale, py = pty * lscale;
    if (level == maxLevel) {
      nx = px;
      ny = py;
    } else {
      nx = nx * 2.f;
      ny = ny * 2.f;
    }
    px -= half;
    py -= half;
    const int ipx = __builtin_amdgcn_readfirstlane((int)floorf(px));
    const int ipy = __builtin_amdgcn_readfirstlane((int)floorf(py));
    const int aw = A.w[level], ah = A.h[level];
    const bool in_range = !(ipx < -win || ipx >= aw || ipy < -win || ipy >= ah);
    const int xaI = (ipx - 1) & ~3;
    const int oI = (ipx - 1) - xaI;

    // ---- this level's I patch: aligned dwords when interior, assembled byte-wise at the border
    if (in_range) {
      const uint8_t* Ab = A.base[level] + (long long)stream * A.sstride[level];
      if (region_is_fast(Ab, A.pitch[level], aw, ah, xaI, ipy - 1, L.gi)) {
        unsigned int ri[NI];
        region_load_regs<NI>(ri, Ab, A.pitch[level], xaI, ipy - 1, L.gi, lane);
        region_store_regs<NI>(sI, ri, L.gi, lane);
      } else {
        region_stage_slow<NI>(sI, Ab, A.pitch[level], aw, ah, xaI, ipy - 1, L.gi, lane);
      }
    }
    __syncthreads();
    if (in_range) VSF_LKSTAT(level * 16 + 0);

    const uint8_t* Bb = B.base[level] + (long long)stream * B.sstride[level];
    const int bp = B.pitch[level], bw = B.w[level], bh = B.h[level];
    bool level_ok = in_range;
    if (!in_range && level == 0) status = false;

    unsigned int IxP[NP], IyP[NP];  // template gradients, pair-packed, zero where the lane owns no pixel
    int cI1 = 0, cI2 = 0;
    float A11 = 0.f, A12 = 0.f, A22 = 0.f, D = 0.f;
    if (level_ok) {
      int Iw[SEG + 1];
      Iw[SEG] = 0;
      const float fa = px - (float)ipx, fb = py - (float)ipy;
      int iw00, iw01, iw10, iw11;
      bilinear_weights_uniform(fa, fb, iw00, iw01, iw10, iw11);
      const unsigned int W0 = ((unsigned int)iw00 & 0xffffu) | ((unsigned int)iw01 << 16);
      const unsigned int W1 = ((unsigned int)iw10 & 0xffffu) | ((unsigned int)iw11 << 16);
      unsigned int pm[NP];
#pragma unroll
      for (int k2 = 0; k2 < NP; k2++) pm[k2] = sMask[k2];
      // every tap of every window pixel lies inside the level image: no zero-padded gradient involved
      const bool interior = ipx >= 0 && ipy >= 0 && ipx + win < aw && ipy + win < ah;
      if (interior) {
        VSF_LKSTAT(level * 16 + 4);
        // Fast path.  The Scharr operator and the bilinear interpolation are both linear and integer,
        // so sum_w(w * Scharr(I)) == Scharr(V) with V = sum_w(w * I) (22 bits) evaluated on the window
        // grid plus a one-pixel ring.  Each lane forms V on 3 rows x (SEG+2) columns from 4 patch
        // rows, then applies the separable Scharr stencil; nothing goes through LDS again.
        constexpr int NC = SEG + 2;            // V columns: window columns xx0-1 .. xx0+SEG
        constexpr int NAV = (NC + 1 + 3) / 4;  // aligned dwords holding NC+1 bytes
        // The three V rows are produced one after the other and folded into the vertical Scharr sums
        // at once (T0 = 3 V0 + 10 V1 + 3 V2, T1 = V2 - V0), so at most one V row is live.
        const uint8_t* rI = sI + rowI;
        const bool wpos = iw11 >= 0;  // the other three weights are rounded products of non-negatives
        int T0[NC], T1[NC];
        {
          unsigned int Ra[NAV], Rb[NAV];
          int V[NC];
          lds_stream<NAV>(rI, oI + xx0, Ra);
          lds_stream<NAV>(rI + L.gi.pitch, oI + xx0, Rb);
          bilinear_row<NC, NAV, 0>(wpos, Ra, Rb, W0, W1, V);  // V0
#pragma unroll
          for (int c = 0; c < NC; c++) {
            T0[c] = __mul24(V[c], 3);
            T1[c] = -V[c];
          }
          lds_stream<NAV>(rI + 2 * L.gi.pitch, oI + xx0, Ra);
          bilinear_row<NC, NAV, 0>(wpos, Rb, Ra, W0, W1, V);  // V1
#pragma unroll
          for (int c = 0; c < NC; c++) T0[c] += __mul24(V[c], 10);
#pragma unroll
          for (int k = 0; k < SEG; k++) Iw[k] = VSF_DESCALE(V[k + 1], W_BITS - 5);
          lds_stream<NAV>(rI + 3 * L.gi.pitch, oI + xx0, Rb);
          bilinear_row<NC, NAV, 0>(wpos, Ra, Rb, W0, W1, V);  // V2
#pragma unroll
          for (int c = 0; c < NC; c++) {
            T0[c] += __mul24(V[c], 3);  // < 2^27
            T1[c] += V[c];
          }
        }
        int ixv[SEG + 1], iyv[SEG + 1];
        ixv[SEG] = 0;
        iyv[SEG] = 0;
#pragma unroll
        for (int k = 0; k < SEG; k++) {
          ixv[k] = VSF_DESCALE(T0[k + 2] - T0[k], W_BITS);
          iyv[k] = VSF_DESCALE(3 * (T1[k] + T1[k + 2]) + 10 * T1[k + 1], W_BITS);
        }
        // pixels a lane does not own carry zero gradients, so they add nothing to any sum
#pragma unroll
        for (int k2 = 0; k2 < NP; k2++) {
          IxP[k2] = pack_lo16(ixv[2 * k2], ixv[2 * k2 + 1]) & pm[k2];
          IyP[k2] = pack_lo16(iyv[2 * k2], iyv[2 * k2 + 1]) & pm[k2];
        }
      } else {
        // Generic path (window touches the level border): Scharr gradient patch (win+1)^2 in LDS,
        // zero outside the level image (BORDER_CONSTANT), then the bilinear template per lane.
        {
          const int side = win + 1;
          const int xs = lane & 31;
          if (xs < side) {
            const int gx = ipx + xs;
            const bool xin = gx >= 0 && gx < aw;
            for (int y = lane >> 5; y < side; y += 2) {
              const int gy = ipy + y;
              int packed = 0;
              if (xin && gy >= 0 && gy < ah) {
                const uint8_t* c = sI + (y + 1) * L.gi.pitch + (oI + xs + 1);
                const int p = L.gi.pitch;
                const int i00 = c[-p - 1], i01 = c[-p], i02 = c[-p + 1];
                const int i10 = c[-1], i12 = c[1];
                const int i20 = c[p - 1], i21 = c[p], i22 = c[p + 1];
                const int ix = 3 * (i02 - i00) + 10 * (i12 - i10) + 3 * (i22 - i20);
                const int iy = 3 * (i20 - i00) + 10 * (i21 - i01) + 3 * (i22 - i02);
                packed = (ix & 0xffff) | (iy << 16);
              }
              sD[y * L.d_pitch + xs] = packed;
            }
          }
        }
        __syncthreads();
        unsigned int R0[NA], R1[NA];
        lds_stream<NA>(sI + rowI + L.gi.pitch, oI + 1 + xx0, R0);
        lds_stream<NA>(sI + rowI + 2 * L.gi.pitch, oI + 1 + xx0, R1);
        int Iv[SEG];
        BilinearRow<0, SEG, NA, (1 << (W_BITS - 5 - 1))>::run(R0, R1, W0, W1, Iv);
        const int* d0 = sD + rr * L.d_pitch + xx0;
        const int* d1 = d0 + L.d_pitch;
        int ixv[SEG + 1], iyv[SEG + 1];
        ixv[SEG] = 0;
        iyv[SEG] = 0;
#pragma unroll
        for (int k = 0; k < SEG; k++) {
          Iw[k] = Iv[k] >> (W_BITS - 5);
          ixv[k] = 0;
          iyv[k] = 0;
          if (k < prm.seglen) {  // d0[k + 1] stays inside the (win+1)-wide patch for every pixel read
            const int p00 = d0[k], p01 = d0[k + 1], p10 = d1[k], p11 = d1[k + 1];
            ixv[k] = VSF_DESCALE(__mul24((int)(short)(p00 & 0xffff), iw00) + __mul24((int)(short)(p01 & 0xffff), iw01) +
                                     __mul24((int)(short)(p10 & 0xffff), iw10) +
                                     __mul24((int)(short)(p11 & 0xffff), iw11),
                                 W_BITS);
            iyv[k] = VSF_DESCALE(__mul24(p00 >> 16, iw00) + __mul24(p01 >> 16, iw01) + __mul24(p10 >> 16, iw10) +
                                     __mul24(p11 >> 16, iw11),
                                 W_BITS);
          }
        }
#pragma unroll
        for (int k2 = 0; k2 < NP; k2++) {
          IxP[k2] = pack_lo16(ixv[2 * k2], ixv[2 * k2 + 1]) & pm[k2];
          IyP[k2] = pack_lo16(iyv[2 * k2], iyv[2 * k2 + 1]) & pm[k2];
        }
      }
      // b = sum (Jw - Iw) * (Ix, Iy) = sum Jw * (Ix, Iy) - (cI1, cI2): the template's own part is a
      // per-lane constant of the level, so the iterations neither subtract per pixel nor keep Iw live
      int a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
      for (int k2 = 0; k2 < NP; k2++) {
        const unsigned int iwp = pack_lo16(Iw[2 * k2], Iw[2 * k2 + 1]);  // 0 <= Iw <= 8160
        a11 = sdot2(IxP[k2], IxP[k2], a11);
        a12 = sdot2(IxP[k2], IyP[k2], a12);
        a22 = sdot2(IyP[k2], IyP[k2], a22);
        cI1 = sdot2(iwp, IxP[k2], cI1);
        cI2 = sdot2(iwp, IyP[k2], cI2);
      }
      float Asum[3];
      {
        const int part[3] = {a11, a12, a22};
#if VSF_SUM_LDS
        wave_sums_exact_lds<3, NRAW>(part, Asum, sAcc, lane);
#else
        wave_sums_exact_f32<3>(part, Asum);
#endif
      }
      A11 = Asum[0] * FLT_SCALE;
      A12 = Asum[1] * FLT_SCALE;
      A22 = Asum[2] * FLT_SCALE;
      const float P = A11 * A22;
      D = P - A12 * A12;
      // cv rejects the level when minEig < threshold or D < FLT_EPSILON.  lambda_min >= det / trace, so a
      // determinant well above trace * threshold settles the first test without the square root and
      // the division; the margins (1e-3 relative on the threshold, 1e-6 * trace^2 and 1e-6 * A11 * A22
      // absolute, i.e. > 10 ulp of every float rounding in either evaluation) keep the shortcut on the
      // safe side, and whatever it does not settle takes cv's own expression.
      const float T = A11 + A22;
      bool reject = D < 1.1920928955078125e-7f;
      if (!reject && !(2.f * D >= T * (prm.min_eig_c1 + 1e-6f * T) + 1e-6f * P)) {
        const float minEig =
            (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
        reject = (double)minEig < prm.min_eig_thr;
      }
      if (reject) {
        if (level == 0) status = false;
        level_ok = false;
      }
    }

    if (level_ok) {
      VSF_LKSTAT(level * 16 + 1);
      int n_iter = 0, end_kind = 7;
      D = 1.f / D;
      bool staged = false;
      float cx = nx - half, cy = ny - half;  // nextPt - halfWin
      float pdx = 0.f, pdy = 0.f;
      for (int j = 0; j < prm.max_count; j++) {
        const int inx = __builtin_amdgcn_readfirstlane((int)floorf(cx));
        const int iny = __builtin_amdgcn_readfirstlane((int)floorf(cy));
        if (inx < -win || inx >= bw || iny < -win || iny >= bh) {
          if (level == 0) status = false;
          end_kind = 8;
          break;
        }
        n_iter++;
        if (!staged || inx < jx0 || iny < jy0 || inx + win + 1 > jx0 + 4 * L.gj.ndw ||
            iny + win + 1 > jy0 + L.gj.rows) {
          __syncthreads();  // previous iteration's reads are done
          jx0 = (inx - prm.jmargin) & ~3;
          jy0 = iny - prm.jmargin;
          if (region_is_fast(Bb, bp, bw, bh, jx0, jy0, L.gj)) {
            unsigned int rj[NJ];
            region_load_regs<NJ>(rj, Bb, bp, jx0, jy0, L.gj, lane);
            region_store_pairs<NJ>(sJ, rj, L.gj, L.jp_pitch, lane);
          } else {
            region_stage_slow_pairs<NJ>(sJ, Bb, bp, bw, bh, jx0, jy0, L.gj, L.jp_pitch, lane);
          }
          staged = true;
          VSF_LKSTAT(level * 16 + 3);
          __syncthreads();
        }
        int iw00, iw01, iw10, iw11;
        bilinear_weights_uniform(cx - (float)inx, cy - (float)iny, iw00, iw01, iw10, iw11);
        int b1 = -cI1, b2 = -cI2;
        {
          unsigned int R0[SEG], R1[SEG];
          // rowJ0 = sJ + rr * pitch + xx0 per lane; the rest of the address is wave-uniform
          const unsigned int* p0 = rowJ0 + ((iny - jy0) * L.jp_pitch + (inx - jx0));
#pragma unroll
          for (int k = 0; k < SEG; k++) R0[k] = p0[k];
#pragma unroll
          for (int k = 0; k < SEG; k++) R1[k] = p0[L.jp_pitch + k];
          const unsigned int V0 = ((unsigned int)iw00 & 0xffffu) | ((unsigned int)iw01 << 16);
          const unsigned int V1 = ((unsigned int)iw10 & 0xffffu) | ((unsigned int)iw11 << 16);
          int Jw[SEG + 1];
          Jw[SEG] = 0;
          {
            int Jr[SEG];
            if (iw11 >= 0) {
#pragma unroll
              for (int k = 0; k < SEG; k++) Jr[k] = udot2(R1[k], V1, udot2(R0[k], V0, (1 << (W_BITS - 5 - 1))));
            } else {
#pragma unroll
              for (int k = 0; k < SEG; k++) Jr[k] = sdot2(R1[k], V1, sdot2(R0[k], V0, (1 << (W_BITS - 5 - 1))));
            }
#pragma unroll
            for (int k = 0; k < SEG; k++) Jw[k] = Jr[k] >> (W_BITS - 5);  // 0 <= Jw <= 8160
          }
#pragma unroll
          for (int k2 = 0; k2 < NP; k2++) {
            const unsigned int jp = pack_lo16(Jw[2 * k2], Jw[2 * k2 + 1]);
            b1 = sdot2(jp, IxP[k2], b1);
            b2 = sdot2(jp, IyP[k2], b2);
          }
        }
        float bsum[2];
        {
          const int part[2] = {b1, b2};
#if VSF_SUM_LDS
          wave_sums_exact_lds<2, NRAW>(part, bsum, sAcc, lane);
#else
          wave_sums_exact_f32<2>(part, bsum);
#endif
        }
        const float fb1 = bsum[0] * FLT_SCALE, fb2 = bsum[1] * FLT_SCALE;
        const float dx = (A12 * fb2 - A22 * fb1) * D;
        const float dy = (A12 * fb1 - A11 * fb2) * D;
        cx += dx;
        cy += dy;
        nx = cx + half;
        ny = cy + half;
        // cv: dx*dx + dy*dy <= eps^2 evaluated in double.  The float value s is within 3 ulp of the
        // exact sum, so outside a 1e-6 relative band around eps^2 the float compare already decides.
        {
          const float s = __fmaf_rn(dx, dx, dy * dy);
          bool conv;
          if (s <= prm.eps2_lo)
            conv = true;
          else if (s >= prm.eps2_hi)
            conv = false;
          else
            conv = (double)dx * (double)dx + (double)dy * (double)dy <= prm.eps2;
          if (conv) {
            end_kind = 5;
            break;
          }
        }
        // cv compares the float sums with the double 0.01; (float)0.01 is the largest float below
        // 0.01, so `<= 0.01f` in float is the same predicate
        if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
          nx -= dx * 0.5f;
          ny -= dy * 0.5f;
          end_kind = 6;
          break;
        }
        pdx = dx;
        pdy = dy;
      }
#ifdef VSF_LK_STATS
      if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_lk_stats[level * 16 + 2], (unsigned long long)n_iter);
        atomicAdd(&g_lk_stats[level * 16 + end_kind], 1ull);
        atomicAdd(&g_lk_stats[128 + level * 32 + (n_iter < 31 ? n_iter : 31)], 1ull);
      }
#else
      (void)n_iter;
      (void)end_kind;
#endif
      // cv: with err requested (the reference always passes an err vector) the final position is
      // range-checked once more at level 0
      if (status && level == 0) {
        const int fx = (int)floorf(nx - half), fy = (int)floorf(ny - half);
        if (fx < -win || fx >= bw || fy < -win || fy >= bh) status = false;
      }
    }
    __syncthreads();  // LDS is reused by the next level
  }
  outx = nx;
  outy = ny;
  return status;
}

// Fills the per-lane ownership masks in LDS (once per kernel): 16 bits per window pixel the lane owns.
template <int SEG>
__device__ __forceinline__ void lk_init_masks(const LkParams& prm, uint8_t* smem, const LkSmem& L) {
  constexpr int NP = (SEG + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int r = lane / prm.nseg;
  const int x0 = (lane - r * prm.nseg) * prm.seglen;
  const bool owner = (r < prm.win) && (x0 < prm.win);
  const int cnt = owner ? min(prm.seglen, prm.win - x0) : 0;
  unsigned int* m = reinterpret_cast<unsigned int*>(smem + L.off_mask) + lane * NP;
#pragma unroll
  for (int k2 = 0; k2 < NP; k2++)
    m[k2] = (2 * k2 + 1 < cnt) ? 0xffffffffu : ((2 * k2 < cnt) ? 0x0000ffffu : 0u);
  __syncthreads();
}

// flags bit0 = forward found, bit1 = backward found, bit2 = verified (kept).
// ref: FeatureTrackerLucasKanade.cpp:36-80 (forward, backward on the forward survivors, then
// ||back - original||_2 <= threshold in double against the ORIGINAL double position).
template <int SEG>
__global__ __launch_bounds__(64, (SEG <= 8 ? VSF_LK_WAVES : 2)) void lk_fb_kernel(PyrView P, PyrView C, const double2* __restrict__ xy,
                                                   const int* __restrict__ n, int cap, int fgrid, int nstreams,
                                                   LkParams prm, float2* __restrict__ fwd,
                                                   uint8_t* __restrict__ flags) {
  extern __shared__ __align__(16) uint8_t smem[];
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (id % 8 share an L2), so all
  // features of one stream are given to ONE residue class: their overlapping window lines are then
  // fetched into a single L2 instead of eight.  Placement is a speed matter only.
  // fgrid workgroups serve a stream; feature f0 + k * fgrid goes to workgroup f0 (k > 0 only when a
  // table holds more than fgrid entries, which the usual table size never does).
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int grp = local / fgrid;
  const int stream = grp * 8 + xcd, f0 = local - grp * fgrid;
  if (stream >= nstreams) return;
  const int nf = n[stream];
  if (f0 >= nf) return;
  const LkSmem L = lk_smem_layout(prm.win, prm.jmargin);
  lk_init_masks<SEG>(prm, smem, L);
  for (int f = f0; f < nf; f += fgrid) {
    const double2 p0 = xy[(long long)stream * cap + f];
    const float p0x = (float)p0.x, p0y = (float)p0.y;  // :210 double -> float narrowing
    float qx, qy;
    uint8_t flag = 0;
    if (lk_track_point<SEG>(P, C, stream, p0x, p0y, qx, qy, prm, smem, L)) {
      flag |= 1;
      float rx, ry;
      if (lk_track_point<SEG>(C, P, stream, qx, qy, rx, ry, prm, smem, L)) {
        flag |= 2;
        const double ddx = (double)rx - p0.x, ddy = (double)ry - p0.y;
        if (sqrt(ddx * ddx + ddy * ddy) <= prm.backtrack_thr) flag |= 4;
      }
    }
    if ((threadIdx.x & 63) == 0) {
      fwd[(long long)stream * cap + f] = make_float2(qx, qy);
      flags[(long long)stream * cap + f] = flag;
    }
  }
}

// plain cv::calcOpticalFlowPyrLK on a point list (parity tests): one wave per point
template <int SEG>
__global__ __launch_bounds__(64, (SEG <= 8 ? VSF_LK_WAVES : 2)) void lk_points_kernel(PyrView P, PyrView C, const float2* __restrict__ pts,
                                                       int n, LkParams prm, float2* __restrict__ out,
                                                       uint8_t* __restrict__ status) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int f = blockIdx.x;
  if (f >= n) return;
  const LkSmem L = lk_smem_layout(prm.win, prm.jmargin);
  lk_init_masks<SEG>(prm, smem, L);
  const float2 p = pts[f];
  float qx, qy;
  const bool st = lk_track_point<SEG>(P, C, 0, p.x, p.y, qx, qy, prm, smem, L);
  if ((threadIdx.x & 63) == 0) {
    out[f] = make_float2(qx, qy);
    status[f] = st ? 1 : 0;
  }
}

// Stable in-place compaction of one stream's table by the verified flag; also the three counts
// the reference logs (FeatureTrackerLucasKanade.cpp:87-104).  One block of 256 threads per stream.
__global__ __launch_bounds__(256) void compact_kernel(uint64_t* __restrict__ ids, double2* __restrict__ xy,
                                                      uint32_t* __restrict__ age, int* __restrict__ n, int cap,
                                                      const float2* __restrict__ fwd,
                                                      const uint8_t* __restrict__ flags, int* __restrict__ stats) {
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
// cells processed before the reference's early `break` (Q2).  One block per stream; an LDS hash
// set de-duplicates pixels exactly like the reference's flag image.
__global__ __launch_bounds__(256) void occupancy_kernel(const double2* __restrict__ xy, const int* __restrict__ n,
                                                        int cap, GridView G, int w, int h, int hash_size,
                                                        int* __restrict__ need, int* __restrict__ n_active,
                                                        unsigned int* __restrict__ cell_max,
                                                        int* __restrict__ cand_cnt) {
  extern __shared__ __align__(16) uint8_t smem[];
  int* s_hash = reinterpret_cast<int*>(smem);
  int* s_cnt = s_hash + hash_size;
  const int stream = blockIdx.x;
  const int n0 = n[stream];
  for (int i = threadIdx.x; i < hash_size; i += 256) s_hash[i] = -1;
  for (int i = threadIdx.x; i < G.ncell; i += 256) s_cnt[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n0; i += 256) {
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
        for (int c = 0; c < G.ncell; c++) {
          const int4 rc = G.rects[c];
          if (ix >= rc.x && ix < rc.x + rc.z && iy >= rc.y && iy < rc.y + rc.w) atomicAdd(&s_cnt[c], 1);
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
  for (int c = threadIdx.x; c < G.ncell; c += 256) {
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
constexpr int SC_TW = 64, SC_TH = 16;          // lambda block of a tile
constexpr int SC_IW = SC_TW - 2, SC_IH = SC_TH - 2;  // candidate positions of a tile (tile step)
constexpr int SC_IP = 80;  // LDS pitch of the staged 8-bit tile (tile width + up to 3 bytes of dword alignment)
constexpr int SC_LP = SC_TW + 1;               // LDS pitch of the lambda block
constexpr int SC_LCAP = SC_IH * SC_IW / 4 + 64;  // keys gathered in LDS per block; beyond it they go straight to the list

__host__ __device__ inline int sc_tiles(int cells_extent, int step) {  // tiles covering positions -1 .. extent
  const int t = (cells_extent - 1 + step - 1) / step;
  return t > 1 ? t : 1;
}

// BS = blockSize of the box filter: 3 (offsets -1..+1; cv::goodFeaturesToTrack default, the bucketing
// detector) or 4 (offsets -2..+1: anchor = blockSize / 2; the ANMS detector's GFTTDetector).
// WRITE_LAM: also store the lambda block to the cell's map (fallback / debug); EMIT: emit candidate keys.
template <int BS, bool WRITE_LAM, bool EMIT>
__device__ __forceinline__ void score_tile(const uint8_t* __restrict__ F, int row_stride, int w, int h,
                                           const uint8_t* __restrict__ M, int mask_row_stride, int4 rc, int tx, int ty,
                                           float k_side, float k_mid, float* __restrict__ lam_cell,
                                           unsigned int* __restrict__ cmax, double quality,
                                           unsigned long long* __restrict__ cand, int cand_cap,
                                           int* __restrict__ ccnt) {
  constexpr int HL = BS / 2, HR = BS - 1 - HL;             // covariance halo left/top, right/bottom
  constexpr int CW = SC_TW + HL + HR, CH = SC_TH + HL + HR;  // covariance tile
  constexpr int IW = CW + 2, IH = CH + 2;                  // staged 8-bit tile (Sobel reads +-1 more)
  static_assert(IW + 3 <= SC_IP && SC_IP % 4 == 0, "tile pitch");
  // LDS: the staged image is dead once the covariances exist, and those once lambda exists; the mask bits and
  // the gathered keys reuse their space (the kernel runs on occupancy: three barriers and dependent LDS round
  // trips per tile, 8 blocks per CU at 20 KB each)
  constexpr int IMG_BYTES = (IH * SC_IP + 15) & ~15, COV_BYTES = 3 * CH * CW * 4;
  static_assert(SC_TH * SC_TW <= IMG_BYTES && SC_LCAP * 8 <= COV_BYTES, "aliased regions");
  __shared__ __align__(16) uint8_t s_raw[IMG_BYTES + COV_BYTES];
  __shared__ float s_lam[EMIT ? SC_TH * SC_LP : 1];
  uint8_t* s_img = s_raw;
  uint8_t* s_msk = s_raw;                                                         // after the covariance stage
  float(*s_cov)[CH * CW] = reinterpret_cast<float(*)[CH * CW]>(s_raw + IMG_BYTES);  // dx*dx, dx*dy, dy*dy (float32 products, as cv)
  unsigned long long* s_keys = reinterpret_cast<unsigned long long*>(s_raw + IMG_BYTES);  // after the lambda stage
  __shared__ unsigned int s_max;
  __shared__ int s_n, s_base;
  const int cw = rc.z, ch = rc.w;
  const int cx0 = tx * SC_IW - 1, cy0 = ty * SC_IH - 1;  // origin of the lambda block in cell coordinates
  if (cx0 + 1 >= cw || cy0 + 1 >= ch) return;            // block-uniform: the tile lies outside this cell
  unsigned int seen = 0u;
  if (EMIT) seen = __hip_atomic_load(cmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) {
    s_max = 0u;
    s_n = 0;
  }
  // staged image tile: parent coords [rc.x + cx0 - HL - 1, +IW) x [rc.y + cy0 - HL - 1, +IH)
  const int gx0 = rc.x + cx0 - HL - 1, gy0 = rc.y + cy0 - HL - 1;
  // tiles away from the image border are fetched as aligned dwords (sh = bytes of alignment slack);
  // border tiles byte by byte with REFLECT_101
  const int sh = gx0 & 3;
  constexpr int NDW = (IW + 3 + 3) / 4;
  const bool img_interior = gx0 >= 0 && gy0 >= 0 && gx0 + IW <= w && gy0 + IH <= h && (row_stride & 3) == 0 &&
                            (reinterpret_cast<uintptr_t>(F) & 3) == 0 && (gx0 - sh) + 4 * NDW <= row_stride;
  if (img_interior) {
    const unsigned int* Fw = reinterpret_cast<const unsigned int*>(F + (long long)gy0 * row_stride + (gx0 - sh));
    const unsigned int rsw = (unsigned int)row_stride >> 2;
    unsigned int* sw = reinterpret_cast<unsigned int*>(s_img);
    for (int i = threadIdx.x; i < IH * NDW; i += 256) {
      const int y = i / NDW, xd = i - y * NDW;
      sw[y * (SC_IP / 4) + xd] = Fw[(unsigned int)y * rsw + (unsigned int)xd];
    }
  } else {
    for (int i = threadIdx.x; i < IH * IW; i += 256) {
      const int y = i / IW, x = i - y * IW;
      const int gy = reflect101(gy0 + y, h), gx = reflect101(gx0 + x, w);
      s_img[y * SC_IP + sh + x] = F[(long long)gy * row_stride + gx];
    }
  }
  __syncthreads();
  // covariance at cell coords [cx0-HL, cx0+SC_TW+HR) x [cy0-HL, cy0+SC_TH+HR) that lie inside the cell.
  // A thread walks one column of the tile down a strip of RS rows: per image row it forms the horizontal
  // difference r = I(x+1) - I(x-1) and the horizontal smooth s = k I(x-1) + 2k I(x) + k I(x+1) ONCE, and every
  // covariance row combines three consecutive (r, s) pairs -- the same float operations in the same order as
  // the direct 3x3 form, a third of the conversions and multiplies.
  {
    constexpr int RS = (CH + 2) / 3;
    static_assert(3 * CW <= 256 && 3 * RS >= CH, "strip mapping of the covariance stage");
    const int t = threadIdx.x;
    const int strip = (t >= CW ? 1 : 0) + (t >= 2 * CW ? 1 : 0);
    const int x = t - strip * CW;
    if (t < 3 * CW) {
      const int y0 = strip * RS;
      const uint8_t* col = s_img + (sh + x);
      auto rowvals = [&](int iy, float& rr, float& ss) {
        const uint8_t* c = col + iy * SC_IP;
        const float f0 = (float)c[0], f1 = (float)c[1], f2 = (float)c[2];
        rr = f2 - f0;  // exact: equals (float)(int difference)
        float sv = k_side * f0;
        sv += k_mid * f1;
        sv += k_side * f2;
        ss = sv;
      };
      float r0, s0, r1, s1;
      rowvals(y0, r0, s0);
      rowvals(y0 + 1, r1, s1);
      const int cx = cx0 - HL + x;
      const bool xin = cx >= 0 && cx < cw;
#pragma unroll
      for (int k = 0; k < RS; k++) {
        const int y = y0 + k;
        if (y < CH) {
          float r2, s2;
          rowvals(y + 2, r2, s2);
          const float dx = (r0 + r2) * k_side + r1 * k_mid;
          const float dy = s2 - s0;
          const int cy = cy0 - HL + y;
          const bool in = xin && cy >= 0 && cy < ch;
          s_cov[0][y * CW + x] = in ? dx * dx : 0.f;
          s_cov[1][y * CW + x] = in ? dx * dy : 0.f;
          s_cov[2][y * CW + x] = in ? dy * dy : 0.f;
          r0 = r1;
          s0 = s1;
          r1 = r2;
          s1 = s2;
        }
      }
    }
  }
  __syncthreads();
  unsigned int local_max = 0u;
  const float thr_seen = (EMIT && seen) ? (float)((double)float_order_decode(seen) * quality) : -INFINITY;
  // (x, y): position inside the lambda block; (cx, cy) = (cx0 + x, cy0 + y) lies inside the cell
  auto finish = [&](int x, int y, const double (&sum)[3]) {
    const int cx = cx0 + x, cy = cy0 + y;
    const float a = (float)sum[0] * 0.5f, b = (float)sum[1], c = (float)sum[2] * 0.5f;
    // lambda <= a + c.  Where a + c is not above quality * (the cell maximum seen at block start) the pixel passes
    // neither the final threshold nor the maximum, and as a neighbour any value below a surviving candidate's
    // (which is above that bound) decides the same: 0 stands in, and the correctly rounded square root (a dozen
    // instructions) is skipped -- for whole waves in smooth regions.
    float lm = 0.f;
    const bool full = WRITE_LAM || a + c > thr_seen;
    if (full) lm = (a + c) - sqrtf((a - c) * (a - c) + b * b);
    if (WRITE_LAM) lam_cell[cy * cw + cx] = lm;
    const bool mk = !M || M[(long long)(rc.y + cy) * mask_row_stride + (rc.x + cx)];
    if (EMIT) {
      s_lam[y * SC_LP + x] = lm;
      s_msk[y * SC_TW + x] = mk ? 1 : 0;
    }
    if (mk && full) local_max = max(local_max, float_order_encode(lm));
  };
  // Box filter in double like cv::boxFilter for 32F input: row sum left to right, then the column of
  // row sums top to bottom.
  const bool box_interior = cx0 - HL >= 0 && cx0 + SC_TW + HR <= cw && cy0 - HL >= 0 && cy0 + SC_TH + HR <= ch;
  if (box_interior) {
    // no reflection inside the tile: a thread owns one column and four consecutive rows, forms the
    // BS + 3 row sums they share once, and adds BS of them per pixel (same additions, same order)
    const int x = threadIdx.x & (SC_TW - 1), y0 = (threadIdx.x / SC_TW) * 4;
    static_assert(SC_TH == 16 && SC_TW * 4 == 256, "strip mapping");
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
#pragma unroll
    for (int o = 0; o < 4; o++) {
      double sum[3];
#pragma unroll
      for (int q = 0; q < 3; q++) {
        sum[q] = rs[q][o];
#pragma unroll
        for (int j = 1; j < BS; j++) sum[q] += rs[q][o + j];
      }
      finish(x, y0 + o, sum);
    }
  } else {
    for (int i = threadIdx.x; i < SC_TH * SC_TW; i += 256) {
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
        finish(x, y, sum);
      }
    }
  }
  // wave max, then one LDS atomic per wave, one global atomic per block
  for (int m = 32; m >= 1; m >>= 1) local_max = max(local_max, (unsigned int)__shfl_xor((int)local_max, m, 64));
  if ((threadIdx.x & 63) == 0 && local_max) atomicMax(&s_max, local_max);
  __syncthreads();
  const unsigned int blk_max = s_max;
  if (threadIdx.x == 0 && blk_max) atomicMax(cmax, blk_max);
  if (!EMIT) return;
  // ---- candidates of the inner 62 x 14 positions
  const unsigned int lb = max(seen, blk_max);  // lower bound of the cell's final masked maximum (0: none yet)
  const float thr_lb = lb ? (float)((double)float_order_decode(lb) * quality) : -INFINITY;
#pragma unroll
  for (int k = 0; k < SC_TH * SC_TW / 256; k++) {
    const int x = threadIdx.x & (SC_TW - 1), y = (threadIdx.x >> 6) + 4 * k;  // position in the lambda block
    const int cx = cx0 + x, cy = cy0 + y;
    if (x >= 1 && x <= SC_TW - 2 && y >= 1 && y <= SC_TH - 2 && cx >= 1 && cx < cw - 1 && cy >= 1 && cy < ch - 1) {
      const float* c = s_lam + y * SC_LP + x;
      const float v = c[0];
      if (v > 0.f && v > thr_lb && s_msk[y * SC_TW + x]) {
        float m = v;
#pragma unroll
        for (int j = -1; j <= 1; j++)
#pragma unroll
          for (int q = -1; q <= 1; q++) m = fmaxf(m, c[j * SC_LP + q]);
        if (v == m) {
          // key = lambda bits << 32 | raster index: unsigned order = cv's greaterThanPtr order (value, then address)
          const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned int)(cy * cw + cx);
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
  for (int i = threadIdx.x; i < nl; i += 256) {
    const int slot = s_base + i;
    if (slot < cand_cap) cand[slot] = s_keys[i];
  }
}

// Fused score + candidates.  Grid: (tiles * cells, streams), or with stream_major (streams, tiles * cells): the ANMS
// detector's single whole-frame cell, where the blocks in flight should belong to many streams -- every block ends
// with one returning atomic on its cell's counter, and with the tiles of ONE cell in flight together a few hundred
// blocks queue on one address (the L2's same-address atomic rate: 1.67 against 1.30 ms per 1024 frames, measured on
// the former separate candidates kernel).
template <int BS>
__global__ __launch_bounds__(256) void score_kernel(const uint8_t* __restrict__ frames, long long frame_sstride,
                                                    int row_stride, int w, int h,
                                                    const uint8_t* __restrict__ masks, long long mask_sstride,
                                                    int mask_row_stride, GridView G,
                                                    const int* __restrict__ n_active, int tiles_x, int tiles_y,
                                                    float k_side, float k_mid, double quality,
                                                    unsigned int* __restrict__ cell_max,
                                                    unsigned long long* __restrict__ cand, int cand_cap,
                                                    int* __restrict__ cand_cnt, int stream_major) {
  const int stream = stream_major ? blockIdx.x : blockIdx.y;
  const int tc = stream_major ? blockIdx.y : blockIdx.x;
  const int tpc = tiles_x * tiles_y;
  const int cell = tc / tpc, tile = tc - cell * tpc;
  if (cell >= n_active[stream]) return;
  const int sc = stream * G.ncell + cell;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  score_tile<BS, false, true>(frames + (long long)stream * frame_sstride, row_stride, w, h,
                              masks ? masks + (long long)stream * mask_sstride : nullptr, mask_row_stride,
                              G.rects[cell], tx, ty, k_side, k_mid, nullptr, cell_max + sc, quality,
                              cand + (long long)sc * cand_cap, cand_cap, cand_cnt + sc);
}

// The lambda map of the cells whose candidate list overflowed (select_kernel / anms_topk_kernel then re-derive the
// candidates from the map: exact, slow), or of ONE cell regardless (force_cell >= 0: vsf_debug_min_eig).  Same tile
// routine, so the same arithmetic.  Grid: (cells, streams); a block walks the tiles of its cell.
template <int BS>
__global__ __launch_bounds__(256) void lam_fill_kernel(const uint8_t* __restrict__ frames, long long frame_sstride,
                                                       int row_stride, int w, int h,
                                                       const uint8_t* __restrict__ masks, long long mask_sstride,
                                                       int mask_row_stride, GridView G,
                                                       const int* __restrict__ n_active, int tiles_x, int tiles_y,
                                                       float k_side, float k_mid, float* __restrict__ lam,
                                                       unsigned int* __restrict__ cell_max,
                                                       const int* __restrict__ cand_cnt, int cand_cap, int force_cell) {
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
    score_tile<BS, true, false>(frames + (long long)stream * frame_sstride, row_stride, w, h,
                                masks ? masks + (long long)stream * mask_sstride : nullptr, mask_row_stride,
                                G.rects[cell], tx, ty, k_side, k_mid, lam_cell, cell_max + sc, 0.0, nullptr, 0, nullptr);
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
};

__global__ __launch_bounds__(256) void select_kernel(const float* __restrict__ lam, GridView G,
                                                     const int* __restrict__ n_active,
                                                     const int* __restrict__ need,
                                                     const unsigned int* __restrict__ cell_max,
                                                     const uint8_t* __restrict__ masks, long long mask_sstride,
                                                     int mask_row_stride,
                                                     const unsigned long long* __restrict__ cand,
                                                     const int* __restrict__ cand_cnt, SelParams sp,
                                                     const double2* __restrict__ xy, const int* __restrict__ n,
                                                     double2* __restrict__ sel_xy, int* __restrict__ sel_cnt) {
  extern __shared__ __align__(16) uint8_t smem[];
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(smem);  // [cand_cap], 0 = dead
  int* s_acc_x = reinterpret_cast<int*>(s_key + sp.cand_cap);               // [sel_cap]
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
  // (lambda > (float)(max * quality)) is applied here, and the survivors are compacted in LDS
  __shared__ int s_m;
  if (threadIdx.x == 0) s_m = 0;
  __syncthreads();
  if (!overflow)
    for (int i = threadIdx.x; i < total; i += 256) {
      const unsigned long long k = cand[(long long)sc * sp.cand_cap + i];
      if (__uint_as_float((unsigned int)(k >> 32)) > thr) s_key[atomicAdd(&s_m, 1)] = k;
    }
  __syncthreads();
  const int M = overflow ? cw * ch : s_m;

  int accepted = 0;
  int lastx = -1000000, lasty = -1000000;
  unsigned long long lastkey = ~0ull;
  while (accepted < want) {
    // ---- arg-max over the remaining candidates
    unsigned long long best = 0ull;
    if (!overflow) {
      // fast path: the suppression by the last accepted corner is folded into the scan
      for (int i = threadIdx.x; i < M; i += 256) {
        const unsigned long long k = s_key[i];
        if (k) {
          const int idx = (int)(k & 0xffffffffull);
          const int y = idx / cw, x = idx - y * cw;
          const float ddx = (float)x - (float)lastx, ddy = (float)y - (float)lasty;
          const bool dead = (k == lastkey) || (use_dist && (double)(ddx * ddx + ddy * ddy) < md2);
          if (dead)
            s_key[i] = 0ull;
          else if (k > best)
            best = k;
        }
      }
    } else {
      // overflow path: candidates are re-derived from the lambda map; keys are unique and visited in
      // descending order, so everything >= lastkey has been decided already
      for (int i = threadIdx.x; i < M; i += 256) {
        const int y = i / cw, x = i - y * cw;
        if (x < 1 || x >= cw - 1 || y < 1 || y >= ch - 1) continue;
        float v = L[i];
        v = v > thr ? v : 0.f;
        if (v == 0.f) continue;
        const unsigned long long k = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned int)i;
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
    const int idx = (int)(best & 0xffffffffull);
    lasty = idx / cw;
    lastx = idx - lasty * cw;
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
  for (int t = threadIdx.x; t < accepted * n0; t += 256) {
    const int a = t / n0, i = t - a * n0;
    const double px = (double)((float)s_acc_x[a] + org.x), py = (double)((float)s_acc_y[a] + org.y);
    const double2 e = xy[(long long)stream * sp.cap + i];
    const double ddx = px - e.x, ddy = py - e.y;
    if (!(sqrt(ddx * ddx + ddy * ddy) > sp.min_dist)) s_ok[a] = 0;  // :178-180 keep iff norm > minDist
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
                                                     uint64_t id_bias, int* __restrict__ status) {
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
