// vsf_lk_quad.hpp -- pyramidal LK with SEVERAL features per wavefront: G lanes per feature, G = 16
// (four features, one DPP row each) or G = 32 (two features, two DPP rows each).
//
// Why: the wave-per-feature kernel is VALU-issue bound and ~70 % of its instructions are per-feature
// "scalar" work (bilinear weights, 2x2 solve, convergence tests, reduction tails, addressing) that a
// wave executes once however many lanes need it.  Giving each feature one DPP row makes that work
// vector work over four features, and a row-wide sum needs only the four DPP adds (every lane of the
// row ends up with the row's sum: no v_readlane, no SALU round trip).  Each lane then owns NS window
// "slots" (row segments of <= 8 pixels) instead of one.
//
// Semantics are exactly those of lk_track_point (cv::calcOpticalFlowPyrLK scalar path, SURVEY A.3/A.4,
// exact integer sums); control flow that was wave-uniform there is per-row here (EXEC masks), and the
// level / iteration loops run until no row is active.  Included by vsf_kernels.hpp.
#pragma once

namespace vsf {

struct LkQuadSmem {
  RegionGeom gi, gj;
  int d_pitch;
  int off_j;      // J region inside a group's block
  int grp_bytes;  // bytes per group (I patch + J region)
  int off_d;      // shared gradient patch (border windows only)
  int total;
};

__host__ __device__ inline LkQuadSmem lkq_smem_layout(int win, int jmargin, int groups) {
  LkQuadSmem s;
  s.gi = vsf_region_geom(win + 3, win + 3);
  s.gj = vsf_region_geom(win + 1 + 2 * jmargin, win + 1 + 2 * jmargin);
  s.d_pitch = win + 1;
  s.off_j = (s.gi.pitch * s.gi.rows + 32 + 15) & ~15;
  s.grp_bytes = s.off_j + ((s.gj.pitch * s.gj.rows + 32 + 15) & ~15);
  s.off_d = groups * s.grp_bytes;
  s.total = s.off_d + ((s.d_pitch * (win + 1) * 4 + 15) & ~15);
  return s;
}

// float32(sum over the G lanes of a feature of p[i]), exact; every lane of the group receives its
// group's sums.  |p| < 2^29, so the two quad steps run on the raw value; then each value is split into
// a 16-bit low limb and a signed high limb whose group sums fit int32 (DPP row steps, plus one
// cross-row exchange for G = 32).  float(hi) * 65536 and float(lo) are both exact, so ONE fused
// multiply-add rounds the exact integer sum to float32 once (= the oracle's (float)int64).
template <int N, int G>
__device__ __forceinline__ void group_sums_exact_f32(const int (&p)[N], float (&out)[N]) {
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
  if constexpr (G == 32) {
#pragma unroll
    for (int i = 0; i < 2 * N; i++) v[i] += __shfl_xor(v[i], 16, 64);
  }
#pragma unroll
  for (int i = 0; i < N; i++) out[i] = __fmaf_rn((float)v[2 * i + 1], 65536.f, (float)v[2 * i]);
}

// Region staging by the 16 lanes of a row.  Lane gl takes dword column gl % cols (cols = 8 when the
// region is at most 8 dwords wide, else 16) and every (16 / cols)-th row; rows advance in a run-time
// loop in chunks of CH loads issued back to back (one exposed latency per chunk), and nothing about
// the mapping is loop invariant enough to be hoisted into long-lived registers.
// xa / y0 are per row (per feature); src is common to the wave (one stream per wave).
template <int CH, int G>
__device__ __forceinline__ void regionq_load_store(uint8_t* dst, const uint8_t* __restrict__ src, int sp, int xa,
                                                   int y0, const RegionGeom& g, int gl) {
  const int cl = g.ndw <= 8 ? 3 : 4;
  const int xd = gl & ((1 << cl) - 1), yl = gl >> cl, ystep = G >> cl;
  if (xd < g.ndw) {
    const unsigned int* s = reinterpret_cast<const unsigned int*>(src + (long long)y0 * sp + xa) + xd;
    unsigned int* d = reinterpret_cast<unsigned int*>(dst) + xd;
    const int spw = sp >> 2, pw = g.pitch >> 2;
    for (int yb = yl; yb < g.rows; yb += CH * ystep) {
      unsigned int reg[CH];
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const int y = yb + c * ystep;
        reg[c] = 0u;
        if (y < g.rows) reg[c] = s[(long long)y * spw];
      }
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const int y = yb + c * ystep;
        if (y < g.rows) d[y * pw] = reg[c];
      }
    }
  }
}

// Border path: values identical to OpenCV's reflect-padded pyramid level.
template <int G>
__device__ __forceinline__ void regionq_stage_slow(uint8_t* dst, const uint8_t* __restrict__ src, int sp, int sw,
                                                   int sh, int xa, int y0, const RegionGeom& g, int gl) {
  const int cl = g.ndw <= 8 ? 3 : 4;
  const int xd = gl & ((1 << cl) - 1), yl = gl >> cl, ystep = G >> cl;
  if (xd < g.ndw) {
    unsigned int* d = reinterpret_cast<unsigned int*>(dst) + xd;
    const int pw = g.pitch >> 2;
    int gx[4];
#pragma unroll
    for (int b = 0; b < 4; b++) gx[b] = reflect101(xa + 4 * xd + b, sw);
    for (int y = yl; y < g.rows; y += ystep) {
      const uint8_t* sr = src + (long long)reflect101(y0 + y, sh) * sp;
      d[y * pw] = (unsigned int)sr[gx[0]] | ((unsigned int)sr[gx[1]] << 8) | ((unsigned int)sr[gx[2]] << 16) |
                  ((unsigned int)sr[gx[3]] << 24);
    }
  }
}

// Tracks the 64/G features of a wave (group g = lanes G*g .. G*g+G-1) from pyramid A to pyramid B.
// `alive` : the group holds a feature.  Outputs are per lane (identical within a group).
template <int NS, int G>
__device__ void lkq_track(const PyrView& A, const PyrView& B, int stream, bool alive, float ptx, float pty,
                          float& outx, float& outy, bool& status_out, const LkParams& prm, uint8_t* smem,
                          const LkQuadSmem& L) {
  constexpr int SEG = 8;
  constexpr int NA = (SEG + 1 + 3) / 4;
  constexpr int CH = 8;  // staging loads in flight per lane
  const int lane = threadIdx.x & 63;
  constexpr int F = 64 / G;  // features per wave
  const int g = lane / G, gl = lane % G;
  const int win = prm.win;
  uint8_t* sI = smem + g * L.grp_bytes;
  uint8_t* sJ = sI + L.off_j;
  int* sD = reinterpret_cast<int*>(smem + L.off_d);
  const float half = (float)(win - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (float)(1 << 20);
  const int maxLevel = A.nlevels - 1;

  // window slots of this lane: slot j covers segment q = gl + 16 j (row q / nseg, pixels x0 .. x0+cnt-1)
  int s_row[NS], s_x0[NS], s_cnt[NS];
#pragma unroll
  for (int j = 0; j < NS; j++) {
    const int q = gl + G * j;
    const int row = q / prm.q_nseg;
    const int x0 = (q - row * prm.q_nseg) * prm.q_seglen;
    const bool own = row < win && x0 < win;
    s_row[j] = own ? row : 0;
    s_x0[j] = own ? x0 : 0;
    s_cnt[j] = own ? min(prm.q_seglen, win - x0) : 0;
  }

  bool status = alive;
  float nx = 0.f, ny = 0.f;

  for (int level = maxLevel; level >= 0; level--) {
    // An opaque zero, new in every level: whatever is derived from (gl + oz) cannot be hoisted out of
    // the level loop.  Without it LICM parks ~150 loop-invariant addresses and masks in VGPRs for the
    // whole kernel (measured: 283 -> see DESIGN.md), which costs far more than recomputing them.
    int oz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
    const int glv = gl + oz;
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
    const int ipx = (int)floorf(px), ipy = (int)floorf(py);
    const int aw = A.w[level], ah = A.h[level];
    const bool in_range = alive && !(ipx < -win || ipx >= aw || ipy < -win || ipy >= ah);
    const int xaI = (ipx - 1) & ~3;
    const int oI = (ipx - 1) - xaI;
    const uint8_t* Ab = A.base[level] + (long long)stream * A.sstride[level];

    // ---- I patch per row
    if (in_range) {
      if (region_is_fast(Ab, A.pitch[level], aw, ah, xaI, ipy - 1, L.gi))
        regionq_load_store<CH, G>(sI, Ab, A.pitch[level], xaI, ipy - 1, L.gi, glv);
      else
        regionq_stage_slow<G>(sI, Ab, A.pitch[level], aw, ah, xaI, ipy - 1, L.gi, glv);
    }
    __syncthreads();
    if (alive && !in_range && level == 0) status = false;
    bool level_ok = in_range;

    int iw00 = 0, iw01 = 0, iw10 = 0, iw11 = 0;
    if (in_range) bilinear_weights(px - (float)ipx, py - (float)ipy, iw00, iw01, iw10, iw11);
    const unsigned int W0 = ((unsigned int)iw00 & 0xffffu) | ((unsigned int)iw01 << 16);
    const unsigned int W1 = ((unsigned int)iw10 & 0xffffu) | ((unsigned int)iw11 << 16);
    // every tap of every window pixel inside the level image: the fused (linear) template path applies
    const bool interior = in_range && ipx >= 0 && ipy >= 0 && ipx + win < aw && ipy + win < ah;
    const bool generic = in_range && !interior;

    // level-local slot geometry (recomputed from the opaque lane id, see above)
    int t_row[NS], t_x0[NS], t_cnt[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
      t_row[j] = s_row[j] + oz;
      t_x0[j] = s_x0[j] + oz;
      t_cnt[j] = s_cnt[j] + oz;
    }
    int IxIy[NS][SEG];  // template gradients, (Ix & 0xffff) | (Iy << 16); zero for pixels not owned
    int cI1 = 0, cI2 = 0;  // sum of Iw*Ix, Iw*Iy over the lane's pixels (the constant part of b)
    int a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
    for (int j = 0; j < NS; j++)
#pragma unroll
      for (int k = 0; k < SEG; k++) IxIy[j][k] = 0;

#ifndef VSFQ_NO_INTERIOR
    if (interior) {
      constexpr int NC = SEG + 2;
      constexpr int NAV = (NC + 1 + 3) / 4;
#pragma unroll
      for (int j = 0; j < NS; j++) {
        unsigned int R0[NAV], R1[NAV], R2[NAV], R3[NAV];
        const uint8_t* base = sI + t_row[j] * L.gi.pitch;
        lds_stream<NAV>(base, oI + t_x0[j], R0);
        lds_stream<NAV>(base + L.gi.pitch, oI + t_x0[j], R1);
        lds_stream<NAV>(base + 2 * L.gi.pitch, oI + t_x0[j], R2);
        lds_stream<NAV>(base + 3 * L.gi.pitch, oI + t_x0[j], R3);
        int V0[NC], V1[NC], V2[NC];
        BilinearRow<0, NC, NAV, 0>::run(R0, R1, W0, W1, V0);
        BilinearRow<0, NC, NAV, 0>::run(R1, R2, W0, W1, V1);
        BilinearRow<0, NC, NAV, 0>::run(R2, R3, W0, W1, V2);
        int T0[NC], T1[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
          T0[c] = __mul24(V0[c] + V2[c], 3) + __mul24(V1[c], 10);
          T1[c] = V2[c] - V0[c];
        }
#pragma unroll
        for (int k = 0; k < SEG; k++) {
          const bool own = k < t_cnt[j];
          const int ixval = own ? VSF_DESCALE(T0[k + 2] - T0[k], W_BITS) : 0;
          const int iyval = own ? VSF_DESCALE(3 * (T1[k] + T1[k + 2]) + 10 * T1[k + 1], W_BITS) : 0;
          const int ival = VSF_DESCALE(V1[k + 1], W_BITS - 5);
          IxIy[j][k] = (ixval & 0xffff) | (iyval << 16);
          cI1 += __mul24(ival, ixval);
          cI2 += __mul24(ival, iyval);
          a11 += __mul24(ixval, ixval);
          a12 += __mul24(ixval, iyval);
          a22 += __mul24(iyval, iyval);
        }
        // keep the slots sequential: their transients (4 streams, 30 V, 20 T values) must not overlap
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
      }
    }
#endif
    // ---- border windows: one row at a time through the shared gradient patch (whole wave helps)
#ifndef VSFQ_NO_GENERIC
    {
      const unsigned long long gmask = __ballot(generic);
      if (gmask != 0ull) {
#pragma unroll 1
        for (int gg = 0; gg < F; gg++) {
          if ((gmask >> (G * gg)) & ((G == 32) ? 0xffffffffull : 0xffffull)) {
            const int b_ipx = __builtin_amdgcn_readlane(ipx, G * gg), b_ipy = __builtin_amdgcn_readlane(ipy, G * gg);
            const int b_oI = __builtin_amdgcn_readlane(oI, G * gg);
            const uint8_t* sIg = smem + gg * L.grp_bytes;
            const int side = win + 1;
            const int xs = lane & 31;
            if (xs < side) {
              const int gx = b_ipx + xs;
              const bool xin = gx >= 0 && gx < aw;
              for (int y = lane >> 5; y < side; y += 2) {
                const int gy = b_ipy + y;
                int packed = 0;
                if (xin && gy >= 0 && gy < ah) {
                  const uint8_t* c = sIg + (y + 1) * L.gi.pitch + (b_oI + xs + 1);
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
            __syncthreads();
            if (g == gg) {
#pragma unroll
              for (int j = 0; j < NS; j++) {
                unsigned int R0[NA], R1[NA];
                const uint8_t* base = sI + (t_row[j] + 1) * L.gi.pitch;
                lds_stream<NA>(base, oI + 1 + t_x0[j], R0);
                lds_stream<NA>(base + L.gi.pitch, oI + 1 + t_x0[j], R1);
                int Iw[SEG];
                BilinearRow<0, SEG, NA, (1 << (W_BITS - 5 - 1))>::run(R0, R1, W0, W1, Iw);
                const int* d0 = sD + t_row[j] * L.d_pitch + t_x0[j];
                const int* d1 = d0 + L.d_pitch;
#pragma unroll
                for (int k = 0; k < SEG; k++) {
                  int ixval = 0, iyval = 0;
                  if (k < t_cnt[j]) {
                    const int p00 = d0[k], p01 = d0[k + 1], p10 = d1[k], p11 = d1[k + 1];
                    ixval = VSF_DESCALE(
                        __mul24((int)(short)(p00 & 0xffff), iw00) + __mul24((int)(short)(p01 & 0xffff), iw01) +
                            __mul24((int)(short)(p10 & 0xffff), iw10) + __mul24((int)(short)(p11 & 0xffff), iw11),
                        W_BITS);
                    iyval = VSF_DESCALE(__mul24(p00 >> 16, iw00) + __mul24(p01 >> 16, iw01) +
                                            __mul24(p10 >> 16, iw10) + __mul24(p11 >> 16, iw11),
                                        W_BITS);
                  }
                  const int ival = Iw[k] >> (W_BITS - 5);
                  IxIy[j][k] = (ixval & 0xffff) | (iyval << 16);
                  cI1 += __mul24(ival, ixval);
                  cI2 += __mul24(ival, iyval);
                  a11 += __mul24(ixval, ixval);
                  a12 += __mul24(ixval, iyval);
                  a22 += __mul24(iyval, iyval);
                }
                asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
              }
            }
            __syncthreads();
          }
        }
      }
    }
#endif

    float A11 = 0.f, A12 = 0.f, A22 = 0.f, D = 0.f;
    {
      float Asum[3];
      const int part[3] = {a11, a12, a22};
      group_sums_exact_f32<3, G>(part, Asum);
      A11 = Asum[0] * FLT_SCALE;
      A12 = Asum[1] * FLT_SCALE;
      A22 = Asum[2] * FLT_SCALE;
    }
    if (level_ok) {
      D = A11 * A22 - A12 * A12;
      const float minEig =
          (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
      if ((double)minEig < prm.min_eig_thr || D < 1.1920928955078125e-7f) {
        if (level == 0) status = false;
        level_ok = false;
      } else {
        D = 1.f / D;
      }
    }

    // ---- iterations (per row; the loop runs while any row is active)
    const uint8_t* Bb = B.base[level] + (long long)stream * B.sstride[level];
    const int bp = B.pitch[level], bw = B.w[level], bh = B.h[level];
    float cx = nx - half, cy = ny - half;
    float pdx = 0.f, pdy = 0.f;
    int jx0 = 0, jy0 = 0;
    bool staged = false;
    bool active = level_ok && prm.max_count > 0;
    int j = 0;
    while (__any(active)) {
      int inx = 0, iny = 0;
      bool need = false;
      if (active) {
        inx = (int)floorf(cx);
        iny = (int)floorf(cy);
        if (inx < -win || inx >= bw || iny < -win || iny >= bh) {
          if (level == 0) status = false;
          active = false;
        } else {
          need = !staged || inx < jx0 || iny < jy0 || inx + win + 1 > jx0 + 4 * L.gj.ndw ||
                 iny + win + 1 > jy0 + L.gj.rows;
        }
      }
      if (__any(need)) {
        __syncthreads();  // earlier reads of the region are done
        if (need) {
          jx0 = (inx - prm.jmargin) & ~3;
          jy0 = iny - prm.jmargin;
          if (region_is_fast(Bb, bp, bw, bh, jx0, jy0, L.gj))
            regionq_load_store<CH, G>(sJ, Bb, bp, jx0, jy0, L.gj, glv);
          else
            regionq_stage_slow<G>(sJ, Bb, bp, bw, bh, jx0, jy0, L.gj, glv);
          staged = true;
        }
        __syncthreads();
      }
      if (active) {
        int jw00, jw01, jw10, jw11;
        bilinear_weights(cx - (float)inx, cy - (float)iny, jw00, jw01, jw10, jw11);
        const unsigned int V0 = ((unsigned int)jw00 & 0xffffu) | ((unsigned int)jw01 << 16);
        const unsigned int V1 = ((unsigned int)jw10 & 0xffffu) | ((unsigned int)jw11 << 16);
        int b1 = -cI1, b2 = -cI2;
#ifndef VSFQ_NO_ITER
#pragma unroll
        for (int s = 0; s < NS; s++) {
          unsigned int R0[NA], R1[NA];
          const uint8_t* row0 = sJ + (s_row[s] + iny - jy0) * L.gj.pitch;
          lds_stream<NA>(row0, s_x0[s] + inx - jx0, R0);
          lds_stream<NA>(row0 + L.gj.pitch, s_x0[s] + inx - jx0, R1);
          int Jw[SEG];
          BilinearRow<0, SEG, NA, (1 << (W_BITS - 5 - 1))>::run(R0, R1, V0, V1, Jw);
#pragma unroll
          for (int k = 0; k < SEG; k++) {
            const int jv = Jw[k] >> (W_BITS - 5);
            b1 += __mul24(jv, (int)(short)(IxIy[s][k] & 0xffff));
            b2 += __mul24(jv, IxIy[s][k] >> 16);
          }
          asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        }
#endif
        float bsum[2];
        {
          const int part[2] = {b1, b2};
          group_sums_exact_f32<2, G>(part, bsum);
        }
        const float fb1 = bsum[0] * FLT_SCALE, fb2 = bsum[1] * FLT_SCALE;
        const float dx = (A12 * fb2 - A22 * fb1) * D;
        const float dy = (A12 * fb1 - A11 * fb2) * D;
        cx += dx;
        cy += dy;
        nx = cx + half;
        ny = cy + half;
        if ((double)dx * (double)dx + (double)dy * (double)dy <= prm.eps2) {
          active = false;
        } else if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {  // == "< 0.01" in double
          nx -= dx * 0.5f;
          ny -= dy * 0.5f;
          active = false;
        } else {
          pdx = dx;
          pdy = dy;
          j++;
          if (j >= prm.max_count) active = false;
        }
      }
    }
    if (level_ok && status && level == 0) {
      const int fx = (int)floorf(nx - half), fy = (int)floorf(ny - half);
      if (fx < -win || fx >= bw || fy < -win || fy >= bh) status = false;
    }
    __syncthreads();  // LDS is reused by the next level
  }
  outx = nx;
  outy = ny;
  status_out = status;
}

// flags bit0 = forward found, bit1 = backward found, bit2 = verified (kept); see lk_fb_kernel.
template <int NS, int G>
__global__ __launch_bounds__(64, 3) void lkq_fb_kernel(PyrView P, PyrView C, const double2* __restrict__ xy,
                                                    const int* __restrict__ n, int cap, int quads, int nstreams,
                                                    LkParams prm, float2* __restrict__ fwd,
                                                    uint8_t* __restrict__ flags) {
  extern __shared__ __align__(16) uint8_t smem[];
  // XCD-aware mapping (see lk_fb_kernel): all quads of a stream share one residue class mod 8
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int grp = local / quads;
  const int stream = grp * 8 + xcd, quad = local - grp * quads;
  if (stream >= nstreams) return;
  const int n0 = n[stream];
  constexpr int F = 64 / G;
  if (F * quad >= n0) return;
  const int lane = threadIdx.x & 63;
  const int f = F * quad + lane / G;
  const bool alive = f < n0;
  const LkQuadSmem L = lkq_smem_layout(prm.win, prm.jmargin, F);
  double2 p0 = make_double2(0.0, 0.0);
  if (alive) p0 = xy[(long long)stream * cap + f];
  const float p0x = (float)p0.x, p0y = (float)p0.y;
  float qx, qy, rx, ry;
  bool st1, st2;
  lkq_track<NS, G>(P, C, stream, alive, p0x, p0y, qx, qy, st1, prm, smem, L);
  lkq_track<NS, G>(C, P, stream, alive && st1, qx, qy, rx, ry, st2, prm, smem, L);
  uint8_t flag = 0;
  if (alive && st1) {
    flag |= 1;
    if (st2) {
      flag |= 2;
      const double ddx = (double)rx - p0.x, ddy = (double)ry - p0.y;
      if (sqrt(ddx * ddx + ddy * ddy) <= prm.backtrack_thr) flag |= 4;
    }
  }
  if (alive && (lane % G) == 0) {
    fwd[(long long)stream * cap + f] = make_float2(qx, qy);
    flags[(long long)stream * cap + f] = flag;
  }
}

template <int NS, int G>
__global__ __launch_bounds__(64, 3) void lkq_points_kernel(PyrView P, PyrView C, const float2* __restrict__ pts, int n,
                                                        LkParams prm, float2* __restrict__ out,
                                                        uint8_t* __restrict__ status) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  constexpr int F = 64 / G;
  const int f = F * blockIdx.x + lane / G;
  const bool alive = f < n;
  const LkQuadSmem L = lkq_smem_layout(prm.win, prm.jmargin, F);
  float2 p = make_float2(0.f, 0.f);
  if (alive) p = pts[f];
  float qx, qy;
  bool st;
  lkq_track<NS, G>(P, C, 0, alive, p.x, p.y, qx, qy, st, prm, smem, L);
  if (alive && (lane % G) == 0) {
    out[f] = make_float2(qx, qy);
    status[f] = st ? 1 : 0;
  }
}

}  // namespace vsf
