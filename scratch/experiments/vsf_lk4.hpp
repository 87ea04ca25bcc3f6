// Experiment (round 2), NOT part of the product: four waves per workgroup, one feature per wave, one solver wave.
// Bit-exact on the whole GPU suite, but 5.29 ms against 3.45 ms per 1024-frame launch at C2: the solver in
// per-lane form with its state in LDS costs ~110 VALU per round, the pixel phase ~70 with the LDS parameter reads,
// so the instruction count per feature-iteration stays near the single-wave kernel's 100 while two workgroup
// barriers per round are added; 80 VGPRs with 15 spilled.  Appended to vsf_lk.hpp inside namespace vsf to build.
// ================================================================================================
// Four waves per workgroup, one feature per wave, ONE solver.
//
// The single-wave kernel above spends most of its vector instructions on arithmetic that is uniform across the
// wave: sub-pixel weights, the 2x2 solve, convergence tests, range tests (about 58 of the ~100 VALU instructions
// of an iteration and ~180 of the ~660 of a level set-up, counted in the ISA), and gfx950 has no scalar float unit
// to take it.  A wavefront instruction costs the same 4 cycles of its SIMD with 1 lane or 64 active, and the
// kernel runs AT the integer issue rate, so the only way to pay less for uniform work is to do it for several
// features in one instruction stream.  Here a workgroup is four waves = four features; every wave does the pixel
// work of its own feature (staging, template, bilinear samples, partial sums into LDS accumulators), and between
// two workgroup barriers wave 0 runs the uniform arithmetic of all four in lanes 0..3 (the "solver").  Waves whose
// feature has converged wait at the barriers, which costs occupancy but no issue slots.
// Every float operation is the one the single-wave kernel performs, in the same order: results are identical.
// ================================================================================================
struct Lk4Slot {  // per feature of the workgroup, in LDS; 40 dwords
  float ptx, pty;  // start point of this pass (level 0 coordinates)
  float nx, ny;    // nextPts estimate
  int status;      // OpenCV status so far (1 = found)
  int alive;       // the slot takes part in this pass
  // level set-up: solver -> wave
  int ctl;         // bit0 build the template (in range), bit1 interior
  int ipx, ipy;
  int iw00, iw01, iw10, iw11;
  // iteration: solver -> wave
  int ictl;        // bit0 active, bit1 stage the J region first, bit2 w11 >= 0
  int joff;        // (iny - jy0) * jp_pitch + (inx - jx0)
  int jx0, jy0;
  unsigned int V0, V1;
  int pad0;
  long long acc[3];  // wave -> solver: exact sums (byte offset 80)
  // solver state
  float A11, A12, A22, D;
  float cx, cy, pdx, pdy;
  int j, level_ok, staged, pad1;
  int pad2[2];
};
static_assert(sizeof(Lk4Slot) == 160, "slot layout");

__host__ __device__ inline int lk4_smem_bytes(const LkSmem& L) { return 4 * L.total + 4 * (int)sizeof(Lk4Slot) + 16; }

// cvRound(x * 2^14) of the four bilinear products per lane (bilinear_weights_uniform without the scalar moves)
__device__ __forceinline__ void bilinear_weights_vec(float a, float b, int& w00, int& w01, int& w10, int& w11) {
  const float MAGIC = 12582912.f;  // 1.5 * 2^23
  const float a1 = 1.f - a;
  const float b1 = (1.f - b) * (float)(1 << W_BITS), b0 = b * (float)(1 << W_BITS);
  w00 = __float_as_int(a1 * b1 + MAGIC) - 0x4B400000;
  w01 = __float_as_int(a * b1 + MAGIC) - 0x4B400000;
  w10 = __float_as_int(a1 * b0 + MAGIC) - 0x4B400000;
  w11 = (1 << W_BITS) - w00 - w01 - w10;
}

__device__ __forceinline__ float lk4_read_sum(const long long* acc) {
  const long long t = *acc;
  return exact_i64_to_f32((int)(unsigned int)(unsigned long long)t, (int)(t >> 32));
}

// End of a level's iterations (any exit of cv's loop): with err requested the final position is range-checked
// once more at level 0.  Returns the iteration control word (0 = idle).
__device__ __forceinline__ int lk4_finish_level(Lk4Slot& s, int level, int bw, int bh, float half, int win) {
  if (s.status && level == 0) {
    const int fx = (int)floorf(s.nx - half), fy = (int)floorf(s.ny - half);
    if (fx < -win || fx >= bw || fy < -win || fy >= bh) s.status = 0;
  }
  return 0;
}

// Head of iteration j: range test, restaging decision, weights.  Returns the control word for the wave.
__device__ __forceinline__ int lk4_prepare_iteration(Lk4Slot& s, int level, int bw, int bh, float half,
                                                     const LkParams& prm, const LkSmem& L) {
  const int win = prm.win;
  if (s.j >= prm.max_count) return lk4_finish_level(s, level, bw, bh, half, win);
  const float cx = s.cx, cy = s.cy;
  const int inx = (int)floorf(cx), iny = (int)floorf(cy);
  if (inx < -win || inx >= bw || iny < -win || iny >= bh) {
    if (level == 0) s.status = 0;
    return lk4_finish_level(s, level, bw, bh, half, win);
  }
  int stage = 0;
  int jx0 = s.jx0, jy0 = s.jy0;
  if (!s.staged || inx < jx0 || iny < jy0 || inx + win + 1 > jx0 + 4 * L.gj.ndw || iny + win + 1 > jy0 + L.gj.rows) {
    jx0 = (inx - prm.jmargin) & ~3;
    jy0 = iny - prm.jmargin;
    s.jx0 = jx0;
    s.jy0 = jy0;
    s.staged = 1;
    stage = 2;
  }
  int iw00, iw01, iw10, iw11;
  bilinear_weights_vec(cx - (float)inx, cy - (float)iny, iw00, iw01, iw10, iw11);
  s.V0 = ((unsigned int)iw00 & 0xffffu) | ((unsigned int)iw01 << 16);
  s.V1 = ((unsigned int)iw10 & 0xffffu) | ((unsigned int)iw11 << 16);
  s.joff = (iny - jy0) * L.jp_pitch + (inx - jx0);
  return 1 | stage | (iw11 >= 0 ? 4 : 0);
}

// One pass (A -> B) over the workgroup's four slots.  Slots hold ptx, pty, alive, status = 1 on entry (written by
// the solver lanes themselves); on exit nx, ny, status.  Every wave of the workgroup must call it.
template <int SEG>
__device__ __forceinline__ void lk4_pass(const PyrView& A, const PyrView& B, int stream, const LkParams& prm,
                                         uint8_t* wsm, const LkSmem& L, Lk4Slot* slots, int* any_flag, int wave,
                                         int lane0) {
  constexpr int NP = (SEG + 1) / 2;
  constexpr int NRAW = (SEG <= 8) ? 3 : 2;
  const int win = prm.win;
  const float half = (float)(win - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (float)(1 << 20);
  const int maxLevel = A.nlevels - 1;
  const bool solver = wave == 0 && lane0 < 4;
  Lk4Slot& mine = slots[wave];                   // the slot this wave works for
  Lk4Slot& sl = slots[lane0 < 4 ? lane0 : 0];    // the slot this lane solves (solver lanes only)
  uint8_t* sI = wsm;
  unsigned int* sJ = reinterpret_cast<unsigned int*>(wsm + L.off_j);

  for (int level = maxLevel; level >= 0; level--) {
    const int aw = A.w[level], ah = A.h[level];
    const int bw = B.w[level], bh = B.h[level], bp = B.pitch[level];
    // ---- solver: position at this level, range, template weights
    if (solver) {
      int ctl = 0;
      sl.level_ok = 0;
      if (sl.alive) {
        const float lscale = 1.f / (float)(1 << level);
        float px = sl.ptx * lscale, py = sl.pty * lscale;
        float nx, ny;
        if (level == maxLevel) {
          nx = px;
          ny = py;
        } else {
          nx = sl.nx * 2.f;
          ny = sl.ny * 2.f;
        }
        sl.nx = nx;
        sl.ny = ny;
        px -= half;
        py -= half;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        const bool in_range = !(ipx < -win || ipx >= aw || ipy < -win || ipy >= ah);
        if (!in_range && level == 0) sl.status = 0;
        if (in_range) {
          int iw00, iw01, iw10, iw11;
          bilinear_weights_vec(px - (float)ipx, py - (float)ipy, iw00, iw01, iw10, iw11);
          // every tap of every window pixel lies inside the level image: no zero-padded gradient involved
          const bool interior = ipx >= 0 && ipy >= 0 && ipx + win < aw && ipy + win < ah;
          sl.ipx = ipx;
          sl.ipy = ipy;
          sl.iw00 = iw00;
          sl.iw01 = iw01;
          sl.iw10 = iw10;
          sl.iw11 = iw11;
          sl.level_ok = 1;
          ctl = 1 | (interior ? 2 : 0);
        }
      }
      sl.ctl = ctl;
    }
    __syncthreads();
    // ---- every wave: template of its feature, partial sums of A into the slot
    int oz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(oz));  // see lk_track_point: keeps the lane map out of the outer loops
    const int lane = lane0 + oz;
    const LkLaneMap m = lk_lane_map<SEG>(lane, prm, wsm, L);
    unsigned int IxP[NP], IyP[NP];
    int cI1 = 0, cI2 = 0;
#pragma unroll
    for (int k2 = 0; k2 < NP; k2++) IxP[k2] = IyP[k2] = 0u;
    const int ctl = __builtin_amdgcn_readfirstlane(mine.ctl);
    if (ctl & 1) {
      const int ipx = __builtin_amdgcn_readfirstlane(mine.ipx), ipy = __builtin_amdgcn_readfirstlane(mine.ipy);
      lk_stage_i<SEG>(A, level, stream, ipx, ipy, sI, L, lane);
      lk_wave_sync();
      int part[3];
      const int iw11 = __builtin_amdgcn_readfirstlane(mine.iw11);  // its sign selects the dot2 form: scalar branch
      lk_template<SEG>(wsm, L, prm, m, lane, aw, ah, ipx, ipy, (ctl & 2) != 0, mine.iw00, mine.iw01, mine.iw10, iw11,
                       IxP, IyP, cI1, cI2, part);
      wave_sums_to_lds<3, NRAW>(part, mine.acc, lane);
    }
    __syncthreads();
    // ---- solver: A, determinant, min-eigenvalue test, head of the first iteration
    if (solver) {
      int ictl = 0;
      if (sl.level_ok) {
        const float A11 = lk4_read_sum(&sl.acc[0]) * FLT_SCALE;
        const float A12 = lk4_read_sum(&sl.acc[1]) * FLT_SCALE;
        const float A22 = lk4_read_sum(&sl.acc[2]) * FLT_SCALE;
        float D;
        if (lk_reject_level(A11, A12, A22, D, prm)) {
          if (level == 0) sl.status = 0;
          sl.level_ok = 0;
        } else {
          sl.A11 = A11;
          sl.A12 = A12;
          sl.A22 = A22;
          sl.D = 1.f / D;
          sl.cx = sl.nx - half;  // nextPt - halfWin
          sl.cy = sl.ny - half;
          sl.pdx = 0.f;
          sl.pdy = 0.f;
          sl.j = 0;
          sl.staged = 0;
          ictl = lk4_prepare_iteration(sl, level, bw, bh, half, prm, L);
        }
      }
      sl.ictl = ictl;
    }
    if (wave == 0) {
      const bool act = solver && (sl.ictl & 1);
      const unsigned long long bal = __ballot(act);
      if (lane0 == 0) *any_flag = bal != 0ull;
    }
    __syncthreads();
    // ---- rounds: pixel work of the active slots, then one solver step for all of them
    const uint8_t* Bb = B.base[level] + (long long)stream * B.sstride[level];
    while (__builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(any_flag))) {
      const int ictl = __builtin_amdgcn_readfirstlane(mine.ictl);
      if (ictl & 1) {
        if (ictl & 2) {
          const int jx0 = __builtin_amdgcn_readfirstlane(mine.jx0), jy0 = __builtin_amdgcn_readfirstlane(mine.jy0);
          lk_wave_sync();  // previous iteration's reads are done
          lk_stage_j<SEG>(Bb, bp, bw, bh, jx0, jy0, sJ, L, lane);
          lk_wave_sync();
        }
        int b1, b2;
        lk_iter_pixels<SEG>(m.rowJ0 + mine.joff, L.jp_pitch, mine.V0, mine.V1, (ictl & 4) != 0, IxP, IyP, cI1, cI2, b1,
                            b2);
        const int part[2] = {b1, b2};
        wave_sums_to_lds<2, NRAW>(part, mine.acc, lane);
      }
      __syncthreads();
      if (solver) {
        if (sl.ictl & 1) {
          const float fb1 = lk4_read_sum(&sl.acc[0]) * FLT_SCALE, fb2 = lk4_read_sum(&sl.acc[1]) * FLT_SCALE;
          const float A11 = sl.A11, A12 = sl.A12, A22 = sl.A22, D = sl.D;
          const float dx = (A12 * fb2 - A22 * fb1) * D;
          const float dy = (A12 * fb1 - A11 * fb2) * D;
          const float cx = sl.cx + dx, cy = sl.cy + dy;
          sl.cx = cx;
          sl.cy = cy;
          sl.nx = cx + half;
          sl.ny = cy + half;
          int ictl2;
          if (lk_converged(dx, dy, prm)) {
            ictl2 = lk4_finish_level(sl, level, bw, bh, half, win);
          } else if (sl.j > 0 && fabsf(dx + sl.pdx) <= 0.01f && fabsf(dy + sl.pdy) <= 0.01f) {
            // cv compares the float sums with the double 0.01; (float)0.01 is the largest float below 0.01
            sl.nx -= dx * 0.5f;
            sl.ny -= dy * 0.5f;
            ictl2 = lk4_finish_level(sl, level, bw, bh, half, win);
          } else {
            sl.pdx = dx;
            sl.pdy = dy;
            sl.j += 1;
            ictl2 = lk4_prepare_iteration(sl, level, bw, bh, half, prm, L);
          }
          sl.ictl = ictl2;
        }
      }
      if (wave == 0) {
        const bool act = solver && (sl.ictl & 1);
        const unsigned long long bal = __ballot(act);
        if (lane0 == 0) *any_flag = bal != 0ull;
      }
      __syncthreads();
    }
    lk_wave_sync();  // the wave's LDS is reused by the next level
  }
}

// flags bit0 = forward found, bit1 = backward found, bit2 = verified (kept).
// ref: FeatureTrackerLucasKanade.cpp:36-80 (forward, backward on the forward survivors, then
// ||back - original||_2 <= threshold in double against the ORIGINAL double position).
// Grid: 8 XCD residue classes x ceil(streams / 8) x ggrid workgroups; workgroup g0 of a stream takes the
// feature quadruples g0, g0 + ggrid, ...  (XCD-aware mapping: see lk_fb_kernel.)
template <int SEG>
__global__ __launch_bounds__(256, (SEG <= 8 ? 6 : 2)) void lk4_fb_kernel(PyrView P, PyrView C,
                                                                         const double2* __restrict__ xy,
                                                                         const int* __restrict__ n, int cap, int ggrid,
                                                                         int nstreams, LkParams prm,
                                                                         float2* __restrict__ fwd,
                                                                         uint8_t* __restrict__ flags) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int grp = local / ggrid;
  const int stream = grp * 8 + xcd, g0 = local - grp * ggrid;
  if (stream >= nstreams) return;
  const int nf = n[stream];
  if (4 * g0 >= nf) return;
  const int wave = threadIdx.x >> 6, lane0 = threadIdx.x & 63;
  const LkSmem L = lk_smem_layout(prm.win, prm.jmargin);
  uint8_t* wsm = smem + wave * L.total;
  Lk4Slot* slots = reinterpret_cast<Lk4Slot*>(smem + 4 * L.total);
  int* any_flag = reinterpret_cast<int*>(slots + 4);
  lk_init_masks<SEG>(prm, wsm, L);
  const bool solver = wave == 0 && lane0 < 4;
  Lk4Slot& sl = slots[lane0 < 4 ? lane0 : 0];
  for (int g = g0; 4 * g < nf; g += ggrid) {
    const int f = 4 * g + lane0;  // solver lanes: the feature of slot lane0
    const bool valid = solver && f < nf;
    double2 p0 = make_double2(0.0, 0.0);
    if (solver) {
      if (valid) p0 = xy[(long long)stream * cap + f];
      sl.alive = valid ? 1 : 0;
      sl.status = 1;
      sl.ptx = (float)p0.x;  // :210 double -> float narrowing
      sl.pty = (float)p0.y;
      sl.nx = 0.f;
      sl.ny = 0.f;
    }
    lk4_pass<SEG>(P, C, stream, prm, wsm, L, slots, any_flag, wave, lane0);
    float qx = 0.f, qy = 0.f;
    bool st1 = false;
    if (solver) {
      qx = sl.nx;
      qy = sl.ny;
      st1 = valid && sl.status != 0;
      sl.alive = st1 ? 1 : 0;
      sl.status = 1;
      sl.ptx = qx;
      sl.pty = qy;
      sl.nx = 0.f;
      sl.ny = 0.f;
    }
    lk4_pass<SEG>(C, P, stream, prm, wsm, L, slots, any_flag, wave, lane0);
    if (valid) {
      uint8_t flag = 0;
      if (st1) {
        flag |= 1;
        if (sl.status != 0) {
          flag |= 2;
          const double ddx = (double)sl.nx - p0.x, ddy = (double)sl.ny - p0.y;
          if (sqrt(ddx * ddx + ddy * ddy) <= prm.backtrack_thr) flag |= 4;
        }
      }
      fwd[(long long)stream * cap + f] = make_float2(qx, qy);
      flags[(long long)stream * cap + f] = flag;
    }
  }
}

// plain cv::calcOpticalFlowPyrLK on a point list (vsf_debug_lk, vsf_lk_points_noalloc): four points per workgroup
template <int SEG>
__global__ __launch_bounds__(256, (SEG <= 8 ? 6 : 2)) void lk4_points_kernel(PyrView P, PyrView C,
                                                                             const float2* __restrict__ pts, int n,
                                                                             LkParams prm, float2* __restrict__ out,
                                                                             uint8_t* __restrict__ status) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int wave = threadIdx.x >> 6, lane0 = threadIdx.x & 63;
  if (4 * (int)blockIdx.x >= n) return;
  const LkSmem L = lk_smem_layout(prm.win, prm.jmargin);
  uint8_t* wsm = smem + wave * L.total;
  Lk4Slot* slots = reinterpret_cast<Lk4Slot*>(smem + 4 * L.total);
  int* any_flag = reinterpret_cast<int*>(slots + 4);
  lk_init_masks<SEG>(prm, wsm, L);
  const bool solver = wave == 0 && lane0 < 4;
  Lk4Slot& sl = slots[lane0 < 4 ? lane0 : 0];
  const int f = 4 * blockIdx.x + lane0;
  const bool valid = solver && f < n;
  if (solver) {
    float2 p = make_float2(0.f, 0.f);
    if (valid) p = pts[f];
    sl.alive = valid ? 1 : 0;
    sl.status = 1;
    sl.ptx = p.x;
    sl.pty = p.y;
    sl.nx = 0.f;
    sl.ny = 0.f;
  }
  lk4_pass<SEG>(P, C, 0, prm, wsm, L, slots, any_flag, wave, lane0);
  if (valid) {
    out[f] = make_float2(sl.nx, sl.ny);
    status[f] = sl.status != 0 ? 1 : 0;
  }
}

