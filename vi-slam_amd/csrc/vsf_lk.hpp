// vsf_lk.hpp -- pyramidal Lucas-Kanade (cv::calcOpticalFlowPyrLK) on gfx950.
// Included by vsf_kernels.hpp after the common declarations (PyrView, LkParams, reflect101).
#pragma once

namespace vsf {

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
// Accumulate half: DPP steps, then lane 0 stores and one lane of every 2^NRAW adds (ds_add_u64) into acc[0..N).
// Nothing is read back here; the reader (this wave or, in the four-wave kernel, the solver wave after a
// workgroup barrier) finds the exact 64-bit totals in acc.
// ONE asm statement per group of sums: lane 0 stores its partial, one lane of every 2^NRAW adds (ds_add_u64), every
// lane reads the totals back.  The three phases differ only in the exec mask, which the statement sets by hand from
// two constants (as three C++ regions each phase cost a v_mov of the LDS address, an s_and_saveexec / s_or pair and a
// branch around it -- vector and scalar instructions this kernel is short of).  LDS operations of one wave execute in
// issue order, so no wait separates the phases.  The wave runs this with every lane active (wave-uniform control
// flow, 64 threads per workgroup); exec is restored to all ones.
template <int N, int NRAW>
__device__ __forceinline__ void wave_sums_lds_roundtrip(const int (&p)[N], long long (&t)[N], long long* acc) {
  static_assert(N == 2 || N == 3, "two or three sums");
  int q[N];
#pragma unroll
  for (int i = 0; i < N; i++) q[i] = p[i] + __builtin_amdgcn_update_dpp(0, p[i], 0xB1, 0xf, 0xf, false);
#pragma unroll
  for (int i = 0; i < N; i++) q[i] += __builtin_amdgcn_update_dpp(0, q[i], 0x4E, 0xf, 0xf, false);
  if constexpr (NRAW == 3) {
#pragma unroll
    for (int i = 0; i < N; i++) q[i] += __builtin_amdgcn_update_dpp(0, q[i], 0x141, 0xf, 0xf, false);
  }
  const unsigned int a32 = (unsigned int)(size_t)(__attribute__((address_space(3))) void*)acc;
  long long q64[N];
#pragma unroll
  for (int i = 0; i < N; i++) q64[i] = (long long)q[i];
  // adders: one lane of every 2^NRAW, lane 0 excepted (it stores)
  constexpr unsigned int ADD_LO = NRAW == 3 ? 0x01010100u : 0x11111110u, ADD_HI = NRAW == 3 ? 0x01010101u : 0x11111111u;
  if constexpr (N == 2) {
    asm volatile(
        "s_mov_b64 exec, 1\n\t"
        "ds_write_b64 %2, %3\n\t"
        "ds_write_b64 %2, %4 offset:8\n\t"
        "s_mov_b32 exec_lo, %5\n\t"
        "s_mov_b32 exec_hi, %6\n\t"
        "ds_add_u64 %2, %3\n\t"
        "ds_add_u64 %2, %4 offset:8\n\t"
        "s_mov_b64 exec, -1\n\t"
        "ds_read_b64 %0, %2\n\t"
        "ds_read_b64 %1, %2 offset:8\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(t[0]), "=&v"(t[1])
        : "v"(a32), "v"(q64[0]), "v"(q64[1]), "s"(ADD_LO), "s"(ADD_HI)
        : "memory");
  } else {
    asm volatile(
        "s_mov_b64 exec, 1\n\t"
        "ds_write_b64 %3, %4\n\t"
        "ds_write_b64 %3, %5 offset:8\n\t"
        "ds_write_b64 %3, %6 offset:16\n\t"
        "s_mov_b32 exec_lo, %7\n\t"
        "s_mov_b32 exec_hi, %8\n\t"
        "ds_add_u64 %3, %4\n\t"
        "ds_add_u64 %3, %5 offset:8\n\t"
        "ds_add_u64 %3, %6 offset:16\n\t"
        "s_mov_b64 exec, -1\n\t"
        "ds_read_b64 %0, %3\n\t"
        "ds_read_b64 %1, %3 offset:8\n\t"
        "ds_read_b64 %2, %3 offset:16\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[N - 1])
        : "v"(a32), "v"(q64[0]), "v"(q64[1]), "v"(q64[N - 1]), "s"(ADD_LO), "s"(ADD_HI)
        : "memory");
  }
}

// float32 of an exact 64-bit integer total, ONE rounding: v_cvt_f32_i32 when it fits 32 bits, through double otherwise
__device__ __forceinline__ float exact_i64_to_f32(int lo, int hi) {
  if (hi == (lo >> 31)) return (float)lo;
  return (float)((double)hi * 4294967296.0 + (double)(unsigned int)lo);
}

// Totals as scalars: low words in lo[], and `wide` != 0 when some total does not fit 32 bits (hi != sign of lo).  The
// totals arrive from the LDS (no vector write precedes the lane reads, so no wait state), and the scalar test puts
// more than two instructions between the lane reads and any later vector use of lo[].
__device__ __forceinline__ void sums_to_scalars2(const long long (&t)[2], int (&lo)[2], int (&hi)[2], int& wide) {
  int x0, x1;
  asm volatile(
      "v_readfirstlane_b32 %0, %7\n\t"
      "v_readfirstlane_b32 %1, %8\n\t"
      "v_readfirstlane_b32 %2, %9\n\t"
      "v_readfirstlane_b32 %3, %10\n\t"
      "s_ashr_i32 %5, %0, 31\n\t"
      "s_ashr_i32 %6, %2, 31\n\t"
      "s_xor_b32 %5, %5, %1\n\t"
      "s_xor_b32 %6, %6, %3\n\t"
      "s_or_b32 %4, %5, %6"
      : "=&s"(lo[0]), "=&s"(hi[0]), "=&s"(lo[1]), "=&s"(hi[1]), "=&s"(wide), "=&s"(x0), "=&s"(x1)
      : "v"((int)(unsigned int)(unsigned long long)t[0]), "v"((int)(t[0] >> 32)),
        "v"((int)(unsigned int)(unsigned long long)t[1]), "v"((int)(t[1] >> 32))
      : "scc");
}

__device__ __forceinline__ void sums_to_scalars3(const long long (&t)[3], int (&lo)[3], int (&hi)[3], int& wide) {
  int x0, x1;
  asm volatile(
      "v_readfirstlane_b32 %0, %9\n\t"
      "v_readfirstlane_b32 %1, %10\n\t"
      "v_readfirstlane_b32 %2, %11\n\t"
      "v_readfirstlane_b32 %3, %12\n\t"
      "v_readfirstlane_b32 %4, %13\n\t"
      "v_readfirstlane_b32 %5, %14\n\t"
      "s_ashr_i32 %7, %0, 31\n\t"
      "s_ashr_i32 %8, %2, 31\n\t"
      "s_xor_b32 %7, %7, %1\n\t"
      "s_xor_b32 %8, %8, %3\n\t"
      "s_or_b32 %6, %7, %8\n\t"
      "s_ashr_i32 %7, %4, 31\n\t"
      "s_xor_b32 %7, %7, %5\n\t"
      "s_or_b32 %6, %6, %7"
      : "=&s"(lo[0]), "=&s"(hi[0]), "=&s"(lo[1]), "=&s"(hi[1]), "=&s"(lo[2]), "=&s"(hi[2]), "=&s"(wide), "=&s"(x0),
        "=&s"(x1)
      : "v"((int)(unsigned int)(unsigned long long)t[0]), "v"((int)(t[0] >> 32)),
        "v"((int)(unsigned int)(unsigned long long)t[1]), "v"((int)(t[1] >> 32)),
        "v"((int)(unsigned int)(unsigned long long)t[2]), "v"((int)(t[2] >> 32))
      : "scc");
}

template <int N, int NRAW>
__device__ __forceinline__ void wave_sums_exact_lds(const int (&p)[N], float (&out)[N], long long* acc, int lane) {
  (void)lane;
  long long t[N];
  wave_sums_lds_roundtrip<N, NRAW>(p, t, acc);
  if constexpr (N == 2) {
    int lo[2], hi[2], wide;
    sums_to_scalars2(t, lo, hi, wide);
    if (wide == 0) {  // the common case: both totals fit 32 bits, v_cvt_f32_i32 rounds once
      out[0] = (float)lo[0];
      out[1] = (float)lo[1];
    } else {
      asm volatile("");  // a block of its own (rare)
      out[0] = exact_i64_to_f32(lo[0], hi[0]);
      out[1] = exact_i64_to_f32(lo[1], hi[1]);
    }
  } else {
    int lo[3], hi[3], wide;
    sums_to_scalars3(t, lo, hi, wide);
    if (wide == 0) {
#pragma unroll
      for (int i = 0; i < 3; i++) out[i] = (float)lo[i];
    } else {
      asm volatile("");
#pragma unroll
      for (int i = 0; i < 3; i++) out[i] = exact_i64_to_f32(lo[i], hi[i]);
    }
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

// The iterations' form: the two packed operands the dot2 instructions take, W0 = w00 | w01 << 16 and
// W1 = w10 | w11 << 16, as scalars.  Lane reads and scalar arithmetic in ONE statement: the low 16 bits of the
// biased floats ARE the weights (0x4B400000 has none), s_pack_ll_b32_b16 joins two of them, and
// w11 = 2^14 - w00 - w01 - w10 = 0xE1C04000 - (s00 + s01 + s10) mod 2^32 (it may be -1; the dot2 is signed).
// No trailing wait state: five scalar instructions separate the lane reads from any vector use of the results.
__device__ __forceinline__ void bilinear_weights_packed(float a, float b, unsigned int& W0, unsigned int& W1) {
  const float MAGIC = 12582912.f;  // 1.5 * 2^23
  const float a1 = 1.f - a;
  const float b1 = (1.f - b) * (float)(1 << W_BITS), b0 = b * (float)(1 << W_BITS);
  const float p00 = a1 * b1 + MAGIC, p01 = a * b1 + MAGIC, p10 = a1 * b0 + MAGIC;
  unsigned int t0, t1, t2;
  asm volatile(
      "s_nop 0\n\t"
      "v_readfirstlane_b32 %2, %5\n\t"
      "v_readfirstlane_b32 %3, %6\n\t"
      "v_readfirstlane_b32 %4, %7\n\t"
      "s_pack_ll_b32_b16 %0, %2, %3\n\t"
      "s_add_i32 %2, %2, %3\n\t"
      "s_add_i32 %2, %2, %4\n\t"
      "s_sub_i32 %2, 0xe1c04000, %2\n\t"
      "s_pack_ll_b32_b16 %1, %4, %2"
      : "=&s"(W0), "=&s"(W1), "=&s"(t0), "=&s"(t1), "=&s"(t2)
      : "v"(p00), "v"(p01), "v"(p10)
      : "scc");
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

// The same as a THREE-operand instruction (v_dot2_i32_i16, VOP3P): the compiler only ever selects the two-operand
// v_dot2c_i32_i16 for the builtin and copies the accumulator first when it is still needed (or is a constant).
// First link of an accumulation chain that starts from a live value or from zero: no v_mov.
__device__ __forceinline__ int sdot2_acc(unsigned int a, unsigned int b, int c) {
  int r;
  asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// first link of a bilinear sample: pixel pair (VGPR) x packed weights (SGPR) + rounding constant (VGPR)
__device__ __forceinline__ int sdot2_sw(unsigned int px, unsigned int w_uniform, int c) {
  int r;
  asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(px), "s"(w_uniform), "v"(c));
  return r;
}
__device__ __forceinline__ int sdot2_zero(unsigned int a, unsigned int b) {
  int r;
  asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// ((lo >> SH) & 0xffff) | ((hi >> SH) << 16) for values whose shifted form fits 16 bits (signed or not): a shift,
// then a shift that writes the upper word of the same register (SDWA) -- two instructions instead of two shifts
// and a v_perm.
template <int SH>
__device__ __forceinline__ unsigned int pack_shr16(int lo, int hi) {
  unsigned int r = (unsigned int)(lo >> SH) & 0xffffu;
  asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
      : "+v"(r)
      : "n"(SH), "v"(hi));
  return r;
}
// the odd last pixel of a lane: (lo >> SH) & 0xffff
template <int SH>
__device__ __forceinline__ unsigned int pack_shr16_single(int lo) {
  return (unsigned int)(lo >> SH) & 0xffffu;
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
  // index a constant, so the arrays stay in registers).  C0 carries DESCALE's rounding constant.  Signed dot2 for
  // every weight set (w11 may be -1): the first link is the three-operand form on a register that holds C0 (the
  // two-operand form the compiler selects copies the constant into every pixel's accumulator first), the second
  // accumulates in place.  W0, W1 wave-uniform.
  static __device__ __forceinline__ void run(const unsigned int (&A0)[NA], const unsigned int (&A1)[NA],
                                             unsigned int W0, unsigned int W1, int c0, int (&out)[N]) {
    out[K] = sdot2(byte_pair<K, NA>(A1), W1, sdot2_sw(byte_pair<K, NA>(A0), W0, c0));
    if constexpr (K + 1 < N) BilinearRow<K + 1, N, NA, C0>::run(A0, A1, W0, W1, c0, out);
  }
};

template <int N, int NA, int C0>
__device__ __forceinline__ void bilinear_row(const unsigned int (&A0)[NA], const unsigned int (&A1)[NA],
                                             unsigned int W0, unsigned int W1, int (&out)[N]) {
  int c0 = C0;
  asm("" : "+v"(c0));  // one register for the whole row, not a literal per pixel
  BilinearRow<0, N, NA, C0>::run(A0, A1, W0, W1, c0, out);
}

typedef unsigned short vsf_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int udot2(unsigned int a, unsigned int b, int c) {
  return (int)__builtin_amdgcn_udot2(__builtin_bit_cast(vsf_u16x2, a), __builtin_bit_cast(vsf_u16x2, b),
                                     (unsigned int)c, false);
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
  int total;
};

// Pixels per lane: the window row is cut into nseg = 64 / win segments of seglen pixels.
constexpr int LK_JMARGIN = 3;  // extra margin staged around the J window (make_lk_params)
__host__ __device__ constexpr int lk_seglen(int win) {
  const int nseg = 64 / win;
  return (win + nseg - 1) / nseg;
}
// Kernel instantiation (compile-time pixels per lane) that serves a given seglen.
__host__ __device__ constexpr int lk_seg_template(int seglen) { return seglen <= 7 ? 7 : (seglen <= 8 ? 8 : 16); }

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
  s.total = s.off_acc + 32;  // 4608 bytes at 21x21: four 1280-byte units, eight waves per SIMD as far as LDS goes
#ifdef VSF_LDS_PAD  // occupancy experiments
  s.total += VSF_LDS_PAD;
#endif
  return s;
}

// A region [xa, xa + 4*ndw) x [y0, y0 + rows) can be fetched as aligned dwords when it lies inside
// the level and rows are dword aligned; otherwise it is assembled byte by byte with REFLECT_101.
// 0 <= xa && xa + 4 ndw <= sw  <=>  (unsigned)xa <= sw - 4 ndw (a level narrower than the region makes the right side
// negative = huge, so it is tested too); the alignment part is a property of the level (PyrLevel::aligned).
__device__ __forceinline__ bool region_is_fast(int aligned, int sw, int sh, int xa, int y0, const RegionGeom& g) {
  const int mx = sw - 4 * g.ndw, my = sh - g.rows;
  return aligned && (mx | my) >= 0 && (unsigned int)xa <= (unsigned int)mx && (unsigned int)y0 <= (unsigned int)my;
}

// Staging map (no division): lane -> dword column xd = lane % C and row yl = lane / C, with C = 8 when
// the region is at most 8 dwords wide and 16 otherwise; task i covers row yl + i * (64 / C).
// Issues the aligned dword loads of a fast region into registers (no wait).  The source pointer is
// wave-uniform, so each load is a scalar base plus a 32-bit lane offset.
template <int N>
__device__ __forceinline__ void region_load_regs(unsigned int (&reg)[N], const uint8_t* __restrict__ src, int sp,
                                                 int xa, int y0, const RegionGeom& g, int lane) {
  const int cl = g.ndw <= 8 ? 3 : 4;
  const int xd = lane & ((1 << cl) - 1), yl = (lane >> cl) & ((64 >> cl) - 1), ystep = 64 >> cl;  // (the mask tells the compiler yl's range: `lane` is opaque)
  const unsigned int* s = reinterpret_cast<const unsigned int*>(src + (long long)y0 * sp + xa);
  const unsigned int spw = (unsigned int)sp >> 2;
  // rows and pitches are far below 2^24: v_mad_u32_u24, not the 64-bit multiply-add a plain 32-bit product takes.
  // (Buffer loads with a wave-uniform descriptor and out-of-range offsets for the idle lanes -- no exec-mask
  // branches, 32-bit offsets -- were tried here: bit-exact, 6 % SLOWER on the whole kernel.)
  const unsigned int off = __umul24((unsigned int)yl, spw) + (unsigned int)xd, step = (unsigned int)ystep * spw;
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
  const int xd = lane & ((1 << cl) - 1), yl = (lane >> cl) & ((64 >> cl) - 1), ystep = 64 >> cl;  // (the mask tells the compiler yl's range: `lane` is opaque)
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
  const int xd = lane & ((1 << cl) - 1), yl = (lane >> cl) & ((64 >> cl) - 1), ystep = 64 >> cl;  // (the mask tells the compiler yl's range: `lane` is opaque)
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
  const int xd = lane & ((1 << cl) - 1), yl = (lane >> cl) & ((64 >> cl) - 1), ystep = 64 >> cl;  // (the mask tells the compiler yl's range: `lane` is opaque)
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
  const int xd = lane & ((1 << cl) - 1), yl = (lane >> cl) & ((64 >> cl) - 1), ystep = 64 >> cl;  // (the mask tells the compiler yl's range: `lane` is opaque)
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

// (int)f of a wave-uniform float as a scalar.  Written through an opaque VGPR: left alone, the compiler moves the
// lane read in front of the conversion (v_readfirstlane, v_cvt with a scalar operand, v_readfirstlane again).
__device__ __forceinline__ int uniform_f2i(float f) {
  int i = (int)f;
  asm volatile("" : "+v"(i));
  return __builtin_amdgcn_readfirstlane(i);
}

// One wave's LDS ordering point: LDS operations of a wave execute in issue order, so a store by one lane is seen
// by a later load of another lane of the SAME wave without any wait; the fence only keeps the compiler from moving
// memory operations across it.  (The single-wave kernels used __syncthreads() here; the four-wave kernel cannot:
// its waves reach these points independently.)
__device__ __forceinline__ void lk_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Per-level lane map: lane l owns window row r = l / nseg and seglen pixels from x0 = (l % nseg) * seglen.
struct LkLaneMap {
  int rowI;                  // byte offset of the lane's row in the I patch
  const unsigned int* rowJ0; // the lane's first pixel in the J pair region (row rr, column xx0)
  int cnt;                   // window pixels the lane owns (0 for an idle lane)
  int rr, xx0;
};

template <int SEG>
__device__ __forceinline__ LkLaneMap lk_lane_map(int lane, const LkParams& prm, uint8_t* smem, const LkSmem& L) {
  LkLaneMap m;
  const int r = (lane * prm.nseg_inv) >> 16;  // lane / nseg
  const int x0 = (lane - r * prm.nseg) * prm.seglen;
  const bool owner = (r < prm.win) && (x0 < prm.win);  // lane owns at least one window pixel
  m.rr = owner ? r : 0;                                // idle lanes read a valid (unused) place
  m.xx0 = owner ? x0 : 0;
  m.cnt = owner ? min(prm.seglen, prm.win - x0) : 0;
  m.rowI = __mul24(m.rr, L.gi.pitch);
  m.rowJ0 = reinterpret_cast<unsigned int*>(smem + L.off_j) + __mul24(m.rr, L.jp_pitch) + m.xx0;
  return m;
}

// This level's I patch into LDS: aligned dwords when interior, assembled byte-wise at the border.
// ipx, ipy wave-uniform.
template <int SEG>
__device__ __forceinline__ void lk_stage_i(const PyrLevel& a, int stream, int ipx, int ipy, uint8_t* sI,
                                           const LkSmem& L, int lane) {
  constexpr int NI = (SEG <= 8) ? 3 : 9;  // I-patch staging tasks per lane (win <= 21: 24 rows / 8; <= 31: 34 / 4)
  const int aw = a.w, ah = a.h;
  const int xaI = (ipx - 1) & ~3;
  const uint8_t* Ab = a.base + (long long)stream * a.sstride;
  if (region_is_fast(a.aligned, aw, ah, xaI, ipy - 1, L.gi)) {
    unsigned int ri[NI];
    region_load_regs<NI>(ri, Ab, a.pitch, xaI, ipy - 1, L.gi, lane);
    region_store_regs<NI>(sI, ri, L.gi, lane);
  } else {
    region_stage_slow<NI>(sI, Ab, a.pitch, aw, ah, xaI, ipy - 1, L.gi, lane);
  }
}

// The template of a level from the staged I patch: interpolated intensities and Scharr gradients of the lane's
// pixels.  Out: gradients pair-packed (IxP[k2] = Ix[2k2] | Ix[2k2+1] << 16, zero where the lane owns no pixel), the
// lane's share of -sum I*Ix, -sum I*Iy (ncI1, ncI2) and of A11, A12, A22 (part).  All weights wave-uniform.
template <int SEG>
__device__ __forceinline__ void lk_template(uint8_t* smem, const LkSmem& L, const LkParams& prm, const LkLaneMap& m,
                                            int lane, int aw, int ah, int ipx, int ipy, bool interior, int iw00,
                                            int iw01, int iw10, int iw11, unsigned int (&IxP)[(SEG + 1) / 2],
                                            unsigned int (&IyP)[(SEG + 1) / 2], int& ncI1, int& ncI2, int (&part)[3]) {
  constexpr int NA = (SEG + 1 + 3) / 4;  // aligned dwords holding SEG+1 bytes
  constexpr int NP = (SEG + 1) / 2;      // pixel pairs per lane
  const int win = prm.win;
  uint8_t* sI = smem;
  int* sD = reinterpret_cast<int*>(smem + L.off_d);
  const int oI = (ipx - 1) - ((ipx - 1) & ~3);
  const int rowI = m.rowI, rr = m.rr, xx0 = m.xx0;
  unsigned int IwP[NP];
  const unsigned int W0 = ((unsigned int)iw00 & 0xffffu) | ((unsigned int)iw01 << 16);
  const unsigned int W1 = ((unsigned int)iw10 & 0xffffu) | ((unsigned int)iw11 << 16);
  // ownership masks, 16 bits per window pixel the lane owns (pixels it does not own carry zero gradients): pair k2
  // holds pixels 2 k2 and 2 k2 + 1, so its mask is the low 32 bits of (1 << 16 * clamp(cnt - 2 k2, 0, 2)) - 1.
  // (They were a 1 KB table in LDS; without it the kernel's LDS fits eight waves per SIMD.)
  unsigned int pm[NP];
  if (prm.win == prm.nseg * prm.seglen && prm.seglen == SEG) {
    // full segments cover the row exactly (21 = 3 x 7): a lane owns all of its pixels or none, and slot SEG of an
    // odd SEG is zero by construction
    const unsigned int own = m.cnt > 0 ? 0xffffffffu : 0u;
#pragma unroll
    for (int k2 = 0; k2 < NP; k2++) pm[k2] = own;
  } else {
#pragma unroll
    for (int k2 = 0; k2 < NP; k2++) {
      const int c = min(max(m.cnt - 2 * k2, 0), 2);
      pm[k2] = (unsigned int)((1ull << (16 * c)) - 1ull);
    }
  }
  if (interior) {
    // Fast path.  The Scharr operator and the bilinear interpolation are both linear and integer,
    // so sum_w(w * Scharr(I)) == Scharr(V) with V = sum_w(w * I) (22 bits) evaluated on the window
    // grid plus a one-pixel ring.  Each lane forms V on 3 rows x (SEG+2) columns from 4 patch
    // rows, then applies the separable Scharr stencil; nothing goes through LDS again.
    constexpr int NC = SEG + 2;            // V columns: window columns xx0-1 .. xx0+SEG
    constexpr int NAV = (NC + 1 + 3) / 4;  // aligned dwords holding NC+1 bytes
    // The three V rows are produced one after the other and folded into the vertical Scharr sums
    // at once (T0 = 3 V0 + 10 V1 + 3 V2, T1 = V2 - V0), so at most one V row is live.
    const uint8_t* rI = sI + rowI;
    int T0[NC], T1[NC];
    {
      unsigned int Ra[NAV], Rb[NAV];
      int V[NC];
      lds_stream<NAV>(rI, oI + xx0, Ra);
      lds_stream<NAV>(rI + L.gi.pitch, oI + xx0, Rb);
      bilinear_row<NC, NAV, 0>(Ra, Rb, W0, W1, V);  // V0
#pragma unroll
      for (int c = 0; c < NC; c++) {
        T0[c] = __mul24(V[c], 3);
        T1[c] = -V[c];
      }
      lds_stream<NAV>(rI + 2 * L.gi.pitch, oI + xx0, Ra);
      bilinear_row<NC, NAV, 0>(Rb, Ra, W0, W1, V);  // V1
#pragma unroll
      for (int c = 0; c < NC; c++) T0[c] += __mul24(V[c], 10);
      {
        int Ir[SEG + 1];  // DESCALE's rounding constant added here, the shift happens in the pack
        Ir[SEG] = 0;
#pragma unroll
        for (int k = 0; k < SEG; k++) Ir[k] = V[k + 1] + (1 << (W_BITS - 5 - 1));
#pragma unroll
        for (int k2 = 0; k2 < NP; k2++)
          IwP[k2] = (2 * k2 + 1 < SEG) ? pack_shr16<W_BITS - 5>(Ir[2 * k2], Ir[2 * k2 + 1])
                                       : pack_shr16_single<W_BITS - 5>(Ir[2 * k2]);
      }
      lds_stream<NAV>(rI + 3 * L.gi.pitch, oI + xx0, Rb);
      bilinear_row<NC, NAV, 0>(Ra, Rb, W0, W1, V);  // V2
#pragma unroll
      for (int c = 0; c < NC; c++) {
        T0[c] += __mul24(V[c], 3);  // < 2^27
        T1[c] += V[c];
      }
    }
    int ixr[SEG + 1], iyr[SEG + 1];  // before DESCALE's shift (the pack shifts)
    ixr[SEG] = 0;
    iyr[SEG] = 0;
#pragma unroll
    for (int k = 0; k < SEG; k++) {
      ixr[k] = T0[k + 2] - T0[k] + (1 << (W_BITS - 1));
      iyr[k] = 3 * (T1[k] + T1[k + 2]) + 10 * T1[k + 1] + (1 << (W_BITS - 1));
    }
    // pixels a lane does not own carry zero gradients, so they add nothing to any sum
#pragma unroll
    for (int k2 = 0; k2 < NP; k2++) {
      if (2 * k2 + 1 < SEG) {
        IxP[k2] = pack_shr16<W_BITS>(ixr[2 * k2], ixr[2 * k2 + 1]) & pm[k2];
        IyP[k2] = pack_shr16<W_BITS>(iyr[2 * k2], iyr[2 * k2 + 1]) & pm[k2];
      } else {
        IxP[k2] = pack_shr16_single<W_BITS>(ixr[2 * k2]) & pm[k2];
        IyP[k2] = pack_shr16_single<W_BITS>(iyr[2 * k2]) & pm[k2];
      }
    }
  } else {
    // Generic path (window touches the level border): Scharr gradient patch (win+1)^2 in LDS,
    // zero outside the level image (BORDER_CONSTANT), then the bilinear template per lane.
    {
      // an opaque zero defined HERE: what derives from it cannot be scheduled above the branch, where the
      // interior path (the common one) would pay for this path's two dozen address instructions
      int oz;
      asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
      const int lane_g = lane + oz;
      const int side = win + 1;
      const int xs = lane_g & 31;
      if (xs < side) {
        const int gx = ipx + xs;
        const bool xin = gx >= 0 && gx < aw;
        for (int y = lane_g >> 5; y < side; y += 2) {
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
    lk_wave_sync();
    unsigned int R0[NA], R1[NA];
    lds_stream<NA>(sI + rowI + L.gi.pitch, oI + 1 + xx0, R0);
    lds_stream<NA>(sI + rowI + 2 * L.gi.pitch, oI + 1 + xx0, R1);
    int Iv[SEG];
    bilinear_row<SEG, NA, (1 << (W_BITS - 5 - 1))>(R0, R1, W0, W1, Iv);
    const int* d0 = sD + rr * L.d_pitch + xx0;
    const int* d1 = d0 + L.d_pitch;
    int ixv[SEG + 1], iyv[SEG + 1];
    ixv[SEG] = 0;
    iyv[SEG] = 0;
#pragma unroll
    for (int k = 0; k < SEG; k++) {
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
    {
      int Iw[SEG + 1];
      Iw[SEG] = 0;
#pragma unroll
      for (int k = 0; k < SEG; k++) Iw[k] = Iv[k] >> (W_BITS - 5);
#pragma unroll
      for (int k2 = 0; k2 < NP; k2++) IwP[k2] = pack_lo16(Iw[2 * k2], Iw[2 * k2 + 1]);
    }
    lk_wave_sync();  // the gradient patch shares its space with the J region: reads done before any staging
  }
  // b = sum (Jw - Iw) * (Ix, Iy) = sum Jw * (Ix, Iy) + (ncI1, ncI2): the template's own part is a per-lane constant of
  // the level (kept NEGATED: the iterations' accumulation chains start from it), so the iterations neither subtract
  // per pixel nor keep Iw live.  IwP[k2] = Iw[2 k2] | Iw[2 k2 + 1] << 16, 0 <= Iw <= 8160.
  int a11 = 0, a12 = 0, a22 = 0, c1 = 0, c2 = 0;
#pragma unroll
  for (int k2 = 0; k2 < NP; k2++) {
    if (k2 == 0) {
      a11 = sdot2_zero(IxP[0], IxP[0]);
      a12 = sdot2_zero(IxP[0], IyP[0]);
      a22 = sdot2_zero(IyP[0], IyP[0]);
      c1 = sdot2_zero(IwP[0], IxP[0]);
      c2 = sdot2_zero(IwP[0], IyP[0]);
    } else {
      a11 = sdot2(IxP[k2], IxP[k2], a11);
      a12 = sdot2(IxP[k2], IyP[k2], a12);
      a22 = sdot2(IyP[k2], IyP[k2], a22);
      c1 = sdot2(IwP[k2], IxP[k2], c1);
      c2 = sdot2(IwP[k2], IyP[k2], c2);
    }
  }
  ncI1 = -c1;
  ncI2 = -c2;
  part[0] = a11;
  part[1] = a12;
  part[2] = a22;
}

// cv rejects the level when minEig < threshold or D < FLT_EPSILON.  lambda_min >= det / trace, so a
// determinant well above trace * threshold settles the first test without the square root and
// the division; the margins (1e-3 relative on the threshold, 1e-6 * trace^2 and 1e-6 * A11 * A22
// absolute, i.e. > 10 ulp of every float rounding in either evaluation) keep the shortcut on the
// safe side, and whatever it does not settle takes cv's own expression.
__device__ __forceinline__ bool lk_reject_level(float A11, float A12, float A22, float& D, const LkParams& prm) {
  const float P = A11 * A22;
  D = P - A12 * A12;
  const float T = A11 + A22;
  bool reject = D < 1.1920928955078125e-7f;
  if (!reject && !(2.f * D >= T * (prm.min_eig_c1 + 1e-6f * T) + 1e-6f * P)) {
    const float minEig =
        (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * prm.win * prm.win);
    reject = (double)minEig < prm.min_eig_thr;
  }
  return reject;
}

// The J region [jx0, jx0 + 4 ndw) x [jy0, jy0 + rows) of level image Bb into LDS, in pair layout.
template <int SEG>
__device__ __forceinline__ void lk_stage_j(const uint8_t* __restrict__ Bb, int bp, int bw, int bh, int aligned, int jx0,
                                           int jy0, unsigned int* sJ, const LkSmem& L, int lane) {
  constexpr int NJ = (SEG <= 8) ? 4 : 10;  // J-region staging tasks per lane (jmargin 3: 28 rows / 8; 38 / 4)
  if (region_is_fast(aligned, bw, bh, jx0, jy0, L.gj)) {
    unsigned int rj[NJ];
    region_load_regs<NJ>(rj, Bb, bp, jx0, jy0, L.gj, lane);
    region_store_pairs<NJ>(sJ, rj, L.gj, L.jp_pitch, lane);
  } else {
    region_stage_slow_pairs<NJ>(sJ, Bb, bp, bw, bh, jx0, jy0, L.gj, L.jp_pitch, lane);
  }
}

// One iteration's pixel work: the lane's share of b1 = sum Jw * Ix + ncI1, b2 = sum Jw * Iy + ncI2, with Jw the
// bilinear sample of J at the lane's pixels.  p0 = the lane's first pixel in the pair region for this window
// position; V0, V1 = packed weights (w00 | w01 << 16, w10 | w11 << 16), wave-uniform.
template <int SEG>
__device__ __forceinline__ void lk_iter_pixels(const unsigned int* p0, int jp_pitch, unsigned int V0, unsigned int V1,
                                               const unsigned int (&IxP)[(SEG + 1) / 2],
                                               const unsigned int (&IyP)[(SEG + 1) / 2], int ncI1, int ncI2, int c0,
                                               int& b1, int& b2) {
  constexpr int NP = (SEG + 1) / 2;
  unsigned int R0[SEG], R1[SEG];
#pragma unroll
  for (int k = 0; k < SEG; k++) R0[k] = p0[k];
#pragma unroll
  for (int k = 0; k < SEG; k++) R1[k] = p0[jp_pitch + k];
  // Jw = Jr >> 9 (0 <= Jw <= 8160; Jr > 0 even when w11 = -1), two pixels to a dword: a bit-field extract, then a
  // shift that writes the upper word (pack_shr16); the accumulation chains open with the three-operand dot2 on the
  // level's constants ncI1 = -sum I*Ix, ncI2 = -sum I*Iy (this lane's share), so nothing is copied
  int Jr[SEG + 1];
  Jr[SEG] = 0;
#pragma unroll
  for (int k = 0; k < SEG; k++) Jr[k] = sdot2(R1[k], V1, sdot2_sw(R0[k], V0, c0));  // c0 = DESCALE's rounding constant
#pragma unroll
  for (int k2 = 0; k2 < NP; k2++) {
    const unsigned int jp = (2 * k2 + 1 < SEG) ? pack_shr16<W_BITS - 5>(Jr[2 * k2], Jr[2 * k2 + 1])
                                               : pack_shr16_single<W_BITS - 5>(Jr[2 * k2]);
    if (k2 == 0) {
      b1 = sdot2_acc(jp, IxP[0], ncI1);
      b2 = sdot2_acc(jp, IyP[0], ncI2);
    } else {
      b1 = sdot2(jp, IxP[k2], b1);
      b2 = sdot2(jp, IyP[k2], b2);
    }
  }
}

// Head of an iteration: the window position as scalars and ONE flag for "needs attention" (outside the box of
// positions that are both in range and served by the staged J region; see lk_track_point).  Lane reads and the
// two unsigned range tests in one statement: the compiler turns the same tests into two dozen mask instructions.
__device__ __forceinline__ void lk_iter_head(int vinx, int viny, int lox, int loy, unsigned int nbx, unsigned int nby,
                                             int& inx, int& iny, int& rare) {
  int tmp;
  asm volatile(
      "s_nop 0\n\t"
      "v_readfirstlane_b32 %0, %4\n\t"
      "v_readfirstlane_b32 %1, %5\n\t"
      "s_sub_i32 %2, %0, %6\n\t"
      "s_sub_i32 %3, %1, %7\n\t"
      "s_cmp_le_u32 %2, %8\n\t"
      "s_cselect_b32 %3, %3, -1\n\t"
      "s_cmp_le_u32 %3, %9\n\t"
      "s_cselect_b32 %2, 0, 1"
      : "=&s"(inx), "=&s"(iny), "=&s"(rare), "=&s"(tmp)
      : "v"(vinx), "v"(viny), "s"(lox), "s"(loy), "s"(nbx), "s"(nby)
      : "scc");
}

// cv: dx*dx + dy*dy <= eps^2 evaluated in double.  The float value s is within 3 ulp of the
// exact sum, so outside a 1e-6 relative band around eps^2 the float compare already decides.
// The arguments are wave-uniform values held in vector registers; every decision goes through a ballot, which makes
// the branch a scalar one in the compiler's eyes too (a branch on a vector compare counts as divergent, and a loop
// with divergent exits is rebuilt around mask registers: a dozen scalar instructions on every trip).
__device__ __forceinline__ bool lk_converged(float dx, float dy, const LkParams& prm) {
  const float s = __fmaf_rn(dx, dx, dy * dy);
  if (__ballot(s <= prm.eps2_lo) != 0ull) return true;
  if (__ballot(s >= prm.eps2_hi) != 0ull) return false;
  // an empty volatile asm gives the rare block a side effect: it cannot be if-converted into the common path (where
  // its seven double-precision instructions ran in every iteration that did not converge)
  asm volatile("");
  return __ballot((double)dx * (double)dx + (double)dy * (double)dy <= prm.eps2) != 0ull;
}

// ---- template cache -------------------------------------------------------------------------------------
// The backward pass of step t builds, for every surviving feature, the template of the CURRENT image at the tracked
// position q; the forward pass of step t+1 needs the template of the same image (now the previous one) at the same
// float position -- two thirds of the kernel's vector instructions are template set-up, and the HBM pipe idles
// while it issues them.  So the backward pass stores what a level set-up produces (pair-packed gradients, the
// lane's share of sum I*Ix / I*Iy, A11 / A12 / A22, the range flag: 2.5 KB per level) and the next step's forward
// pass loads it instead of recomputing it.  A hit needs the same float position, the slot the feature had when the
// entry was written (tslot, carried through compaction) and the epoch of the call that wrote it; anything else
// (new features, host-uploaded tables, a reset tracker, the stateless pair form) recomputes.  The values are the ones
// the set-up would compute from the same image and position, so results are bit-identical with or without it.
// Two buffers ping-pong with the pyramids (a launch reads one and writes the other).
constexpr int LK_TPL_U4 = 160;  // uint4 per (slot, level): IxP [64], IyP [64], (cI1, cI2) as 32 uint4 = 64 uint2

struct LkCacheBuf {
  uint4* tpl;   // [streams][cap][levels][LK_TPL_U4]
  float4* hdr;  // [streams][cap][levels]  {1 / det (0: level rejected or out of range), A11, A12, A22}
  float4* tag;  // [streams][cap]          {qx, qy, epoch as int bits, 0}
};
struct LkCache {
  LkCacheBuf rd, wr;
  const int* tslot;  // [streams][cap] slot of each table entry in `rd` (-1 = none)
  int rd_epoch, wr_epoch;
  int levels;        // entries per slot
  int enabled;
};
// this feature's entries (level 0 first); null = miss / do not write
struct LkTplRef {
  const uint4* rd_tpl;
  const float4* rd_hdr;
  uint4* wr_tpl;
  float4* wr_hdr;
};

// Tracks one point from pyramid A (template, derivatives) to pyramid B.  One wave, wave-uniform control flow.
// Returns the OpenCV status (true = found); out = nextPts.
// SEG = compile-time pixels per lane (>= prm.seglen).
template <int SEG>
__device__ bool lk_track_point(const PyrView& A, const PyrView& B, int stream, float ptx, float pty,
                               float& outx, float& outy, const LkParams& prm, uint8_t* smem,
                               const LkSmem& L, const LkTplRef tc = LkTplRef{nullptr, nullptr, nullptr, nullptr}) {
  constexpr int NP = (SEG + 1) / 2;         // pixel pairs per lane
  constexpr int NRAW = (SEG <= 8) ? 3 : 2;  // raw DPP steps of the exact sums (see wave_sums_exact_lds)
  const int win = prm.win;
  uint8_t* sI = smem;
  unsigned int* sJ = reinterpret_cast<unsigned int*>(smem + L.off_j);
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
    const int lane = (int)(threadIdx.x & 63) + oz;
    const LkLaneMap m = lk_lane_map<SEG>(lane, prm, smem, L);
    const float lscale = __int_as_float((127 - level) << 23);  // (float)(1.0 / 2^level), exact, no division
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
    // the floor as a float serves the fraction below: (float)ip == floorf(p) wherever the point is in range
    const float fpx = floorf(px), fpy = floorf(py);
    const int ipx = uniform_f2i(fpx);
    const int ipy = uniform_f2i(fpy);
    const PyrLevel la = A.lv[level], lb = B.lv[level];  // one scalar load each
    const int aw = la.w, ah = la.h;
    // cv: ip < -win || ip >= size is out of range  <=>  (unsigned)(ip + win) >= size + win
    const bool in_range = (unsigned int)(ipx + win) < (unsigned int)(aw + win) && (unsigned int)(ipy + win) < (unsigned int)(ah + win);
    const bool cached = (SEG <= 8) && tc.rd_hdr != nullptr;  // wave-uniform

    if (in_range && !cached) lk_stage_i<SEG>(la, stream, ipx, ipy, sI, L, lane);
    lk_wave_sync();
    if (in_range) VSF_LKSTAT(level * 16 + 0);

    const uint8_t* Bb = lb.base + (long long)stream * lb.sstride;
    const int bp = lb.pitch, bw = lb.w, bh = lb.h;
    bool level_ok = in_range;
    if (!in_range && level == 0) status = false;

    unsigned int IxP[NP], IyP[NP];  // template gradients, pair-packed, zero where the lane owns no pixel
    int ncI1 = 0, ncI2 = 0;  // this lane's share of -sum I*Ix, -sum I*Iy
    float A11 = 0.f, A12 = 0.f, A22 = 0.f, D = 0.f;
    if (level_ok && cached) {
      VSF_LKSTAT(level * 16 + 9);
      if constexpr (SEG <= 8) {
        // the entry the previous step's backward pass left for this image and position: gradients, constants, the
        // normal matrix and 1 / det (0 = the level was rejected: cv's min-eigenvalue / determinant test), so neither the
        // test nor the division is repeated here
        const float4 h = tc.rd_hdr[level];
        const uint4* t = tc.rd_tpl + level * LK_TPL_U4;
        const uint4 gx = t[lane], gy = t[64 + lane];
        const uint2 ci = reinterpret_cast<const uint2*>(t + 128)[lane];
        IxP[0] = gx.x, IxP[1] = gx.y, IxP[2] = gx.z, IxP[3] = gx.w;
        IyP[0] = gy.x, IyP[1] = gy.y, IyP[2] = gy.z, IyP[3] = gy.w;
        ncI1 = (int)ci.x;
        ncI2 = (int)ci.y;
        A11 = h.y;
        A12 = h.z;
        A22 = h.w;
        D = h.x;
        if (__ballot(D == 0.f) != 0ull) {
          if (level == 0) status = false;
          level_ok = false;
        }
      }
    } else if (level_ok) {
      const float fa = px - fpx, fb = py - fpy;
      int iw00, iw01, iw10, iw11;
      bilinear_weights_uniform(fa, fb, iw00, iw01, iw10, iw11);
      // every tap of every window pixel lies inside the level image: no zero-padded gradient involved
      const bool interior = ipx >= 0 && ipy >= 0 && ipx + win < aw && ipy + win < ah;
      int part[3];
      lk_template<SEG>(smem, L, prm, m, lane, aw, ah, ipx, ipy, interior, iw00, iw01, iw10, iw11, IxP, IyP, ncI1, ncI2,
                       part);
      float Asum[3];
#if VSF_SUM_LDS
      wave_sums_exact_lds<3, NRAW>(part, Asum, sAcc, lane);
#else
      wave_sums_exact_f32<3>(part, Asum);
#endif
      A11 = Asum[0] * FLT_SCALE;
      A12 = Asum[1] * FLT_SCALE;
      A22 = Asum[2] * FLT_SCALE;
      if (lk_reject_level(A11, A12, A22, D, prm)) {
        if (level == 0) status = false;
        level_ok = false;
        D = 0.f;
      } else {
        D = 1.f / D;
      }
      if constexpr (SEG <= 8) {
        if (tc.wr_hdr != nullptr) {  // leave the template for the next step's forward pass (fire and forget)
          uint4* t = tc.wr_tpl + level * LK_TPL_U4;
          t[lane] = make_uint4(IxP[0], IxP[1], IxP[2], IxP[3]);
          t[64 + lane] = make_uint4(IyP[0], IyP[1], IyP[2], IyP[3]);
          reinterpret_cast<uint2*>(t + 128)[lane] = make_uint2((unsigned int)ncI1, (unsigned int)ncI2);
          if (lane == 0) tc.wr_hdr[level] = make_float4(D, A11, A12, A22);
        }
      }
    }
    if constexpr (SEG <= 8) {
      // a level out of range leaves an entry too: the reader takes the same path from its own range test
      if (!in_range && tc.wr_hdr != nullptr && lane == 0) tc.wr_hdr[level] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    if (level_ok) {
      VSF_LKSTAT(level * 16 + 1);
      int n_iter = 0, end_kind = 7;
      float cx = nx - half, cy = ny - half;  // nextPt - halfWin
      // cv tests the oscillation rule from the second iteration on (j > 0): a first "previous delta" this large
      // fails it by itself
      float pdx = 1e30f, pdy = 1e30f;
      // An iteration at window position (inx, iny) needs no attention when the position lies inside the level's
      // range AND inside the staged J region: both in one test against [lox, lox + nx] x [loy, loy + ny], the
      // intersection of the two, kept as scalars and recomputed when a region is staged.  Outside it (rare) the
      // position has either left the image or needs a new region.  Nothing staged yet = an empty box far away.
      int lox = 0x40000000, loy = 0x40000000;
      unsigned int nbx = 0u, nby = 0u;
      int c256 = 1 << (W_BITS - 5 - 1);  // the samples' rounding constant, in a register for the whole level
      asm volatile("" : "+v"(c256));
      // One scalar state word steers the loop (0 = keep iterating; else how the level ended: 5 converged, 6 the
      // oscillation rule, 7 max_count, 8 left the image) and every ending goes through the single latch at the
      // bottom: with exits from the middle the compiler rebuilds the loop around mask registers (the staging code
      // inside it has lane-dependent branches) and spends a dozen scalar instructions per trip on them.
      int state = prm.max_count > 0 ? 0 : 7;
      int j = 0;
      while (state == 0) {
        const float fcx = floorf(cx), fcy = floorf(cy);
        int inx, iny, rare;
        lk_iter_head((int)fcx, (int)fcy, lox, loy, nbx, nby, inx, iny, rare);
        if (rare) {
          asm volatile("");  // a block of its own
          // cv: in < -win || in >= size  <=>  (unsigned)(in + win) >= size + win
          if ((unsigned int)(inx + win) >= (unsigned int)(bw + win) || (unsigned int)(iny + win) >= (unsigned int)(bh + win)) {
            state = 8;
          } else {
            lk_wave_sync();  // previous iteration's reads are done
            jx0 = (inx - prm.jmargin) & ~3;
            jy0 = iny - prm.jmargin;
            lk_stage_j<SEG>(Bb, bp, bw, bh, lb.aligned, jx0, jy0, sJ, L, lane);
            VSF_LKSTAT(level * 16 + 3);
            lk_wave_sync();
            // window positions the region serves: jx0 .. jx0 + 4 ndw - win - 1, cut to the level's range -win .. bw - 1
            lox = max(jx0, -win);
            loy = max(jy0, -win);
            nbx = (unsigned int)(min(jx0 + 4 * L.gj.ndw - win - 1, bw - 1) - lox);
            nby = (unsigned int)(min(jy0 + L.gj.rows - win - 1, bh - 1) - loy);
          }
        }
        if (state == 0) {
          n_iter++;
          unsigned int V0, V1;
          bilinear_weights_packed(cx - fcx, cy - fcy, V0, V1);
          int b1, b2;
          {
            // rowJ0 = sJ + rr * pitch + xx0 per lane; the rest of the address is wave-uniform
            const unsigned int* p0 = m.rowJ0 + ((iny - jy0) * L.jp_pitch + (inx - jx0));
            lk_iter_pixels<SEG>(p0, L.jp_pitch, V0, V1, IxP, IyP, ncI1, ncI2, c256, b1, b2);
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
          // The common outcome first, in ONE branch: clearly not converged (|delta|^2 at or above the upper edge of
          // the band around eps^2) and not oscillating.  cv compares |dx + prev| and |dy + prev| with the double
          // 0.01; (float)0.01 is the largest float below 0.01, so `<= 0.01f` in float is the same predicate, and "not
          // both <= 0.01" is "the larger one > 0.01" (the deltas are finite: D >= FLT_EPSILON, sums bounded).
          const float s2 = __fmaf_rn(dx, dx, dy * dy);
          const float osc = fmaxf(fabsf(dx + pdx), fabsf(dy + pdy));
          j++;
          state = j < prm.max_count ? 0 : 7;
          if ((__ballot(s2 >= prm.eps2_hi) & __ballot(osc > 0.01f)) == 0ull) {
            if (lk_converged(dx, dy, prm)) {
              state = 5;
            } else if (__ballot(osc <= 0.01f) != 0ull) {
              nx -= dx * 0.5f;
              ny -= dy * 0.5f;
              state = 6;
            }
          }
          pdx = dx;
          pdy = dy;
        }
      }
      end_kind = state;
      if (state == 8 && level == 0) status = false;
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
    lk_wave_sync();  // LDS is reused by the next level
  }
  outx = nx;
  outy = ny;
  return status;
}

// flags bit0 = forward found, bit1 = backward found, bit2 = verified (kept).
// ref: FeatureTrackerLucasKanade.cpp:36-80 (forward, backward on the forward survivors, then
// ||back - original||_2 <= threshold in double against the ORIGINAL double position).
// WIN > 0: an instance for that window side (and the default J margin).  The window geometry and the LDS layout
// are then compile-time constants instead of two dozen scalar registers (the generic instance spills SGPRs into
// VGPR lanes and reads them back with v_readlane, a vector instruction each) and fold into the address arithmetic.
template <int SEG, int WIN = 0>
__global__ __launch_bounds__(64, (SEG <= 8 ? VSF_LK_WAVES : 2)) void lk_fb_kernel(PyrView P, PyrView C, const double2* __restrict__ xy,
                                                   const int* __restrict__ n, int cap, int fgrid, int nstreams,
                                                   LkParams prm_in, float2* __restrict__ fwd,
                                                   uint8_t* __restrict__ flags, LkCache cache) {
  extern __shared__ __align__(16) uint8_t smem[];
  LkParams prm = prm_in;
  if constexpr (WIN > 0) {
    static_assert(lk_seg_template(lk_seglen(WIN)) == SEG, "window and pixels per lane");
    prm.win = WIN;
    prm.nseg = 64 / WIN;
    prm.seglen = lk_seglen(WIN);
    prm.nseg_inv = (65536 + 64 / WIN - 1) / (64 / WIN);
    prm.jmargin = LK_JMARGIN;
  }
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (id % 8 share an L2), so all
  // features of one stream are given to ONE residue class: their overlapping window lines are then
  // fetched into a single L2 instead of eight.  Placement is a speed matter only.
  // fgrid workgroups serve a stream; feature f0 + k * fgrid goes to workgroup f0 (k > 0 only when a
  // table holds more than fgrid entries, which the usual table size never does).
  // (grid: x = 8 * fgrid, y = groups of eight streams; the dispatch order is x + gridDim.x * y)
  const int xcd = blockIdx.x & 7, f0 = blockIdx.x >> 3;
  const int stream = (int)blockIdx.y * 8 + xcd;
  if (stream >= nstreams) return;
  const int nf = n[stream];
  if (f0 >= nf) return;
  const LkSmem L = lk_smem_layout(prm.win, prm.jmargin);
  const long long so = (long long)stream * cap;
  for (int f = f0; f < nf; f += fgrid) {
    const double2 p0 = xy[so + f];
    const float p0x = (float)p0.x, p0y = (float)p0.y;  // :210 double -> float narrowing
    LkTplRef fw = {nullptr, nullptr, nullptr, nullptr}, bw = {nullptr, nullptr, nullptr, nullptr};
    if (SEG <= 8 && cache.enabled) {
      const int sl = cache.tslot[so + f];
      if (sl >= 0 && sl < cap) {
        const float4 tg = cache.rd.tag[so + sl];
        if (__float_as_int(tg.z) == cache.rd_epoch && __float_as_int(tg.x) == __float_as_int(p0x) &&
            __float_as_int(tg.y) == __float_as_int(p0y)) {
          fw.rd_tpl = cache.rd.tpl + (so + sl) * cache.levels * LK_TPL_U4;
          fw.rd_hdr = cache.rd.hdr + (so + sl) * cache.levels;
        }
      }
      bw.wr_tpl = cache.wr.tpl + (so + f) * cache.levels * LK_TPL_U4;
      bw.wr_hdr = cache.wr.hdr + (so + f) * cache.levels;
    }
    float qx, qy;
    uint8_t flag = 0;
    if (lk_track_point<SEG>(P, C, stream, p0x, p0y, qx, qy, prm, smem, L, fw)) {
      flag |= 1;
      float rx, ry;
      if (lk_track_point<SEG>(C, P, stream, qx, qy, rx, ry, prm, smem, L, bw)) {
        flag |= 2;
        const double ddx = (double)rx - p0.x, ddy = (double)ry - p0.y;
        if (sqrt(ddx * ddx + ddy * ddy) <= prm.backtrack_thr) flag |= 4;
      }
    }
    if ((threadIdx.x & 63) == 0) {
      fwd[so + f] = make_float2(qx, qy);
      flags[so + f] = flag;
      // the slot is valid for the position the backward pass started from, in the image that is "previous" next time
      if (SEG <= 8 && cache.enabled && (flag & 1))
        cache.wr.tag[so + f] = make_float4(qx, qy, __int_as_float(cache.wr_epoch), 0.f);
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
  const float2 p = pts[f];
  float qx, qy;
  const bool st = lk_track_point<SEG>(P, C, 0, p.x, p.y, qx, qy, prm, smem, L);
  if ((threadIdx.x & 63) == 0) {
    out[f] = make_float2(qx, qy);
    status[f] = st ? 1 : 0;
  }
}

}  // namespace vsf
