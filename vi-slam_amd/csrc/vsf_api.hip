// vsf_api.hip -- C ABI (include/vsf.h) over the HIP kernels in vsf_kernels.hpp.
// Host side only orchestrates: allocation, geometry, kernel launches on the caller's HIP stream.
// No CPU fallback exists; every numeric result comes from the gfx950 kernels.
#include "../../include/vsf.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is opened at run time (vsf_comm_*)

#include <mutex>
#include <utility>
#include <vector>

#include "vsf_kernels.hpp"

using namespace vsf;

namespace {

struct PyrSet {            // levels >= 1 of `streams` pyramids
  uint8_t* lv[MAXL] = {};  // lv[0] unused
};

struct FrameRef {  // level 0 of a pyramid
  const uint8_t* ptr = nullptr;
  long long sstride = 0;
  int pitch = 0;
};

struct EvPair {
  int kid;
  hipEvent_t a, b;
};

}  // namespace

struct vsf_ctx {
  vsf_params p;
  int cap = 0;
  int B = 0;
  // pyramid geometry
  int nlevels = 0;
  int lw[MAXL], lh[MAXL];
  int lp[MAXL];  // pitch of levels >= 1 (width rounded up to 4 so rows stay dword aligned)
  // tracker state
  PyrSet track_pyr[2];
  FrameRef track_frame[2];
  int cur = 0;
  bool has_prev = false;
  uint8_t* own_frame[2] = {};  // host-API staging of stream 0 (ping-pong: previous frame stays resident)
  uint8_t* h_stage = nullptr;  // pinned host block of the host-pointer API: frame | mask | table pack (first use)
  uint8_t* d_pack = nullptr;   // device side of the table pack
  uint8_t* h_bgr = nullptr;    // multi-channel detector input (vsf_detect_bgr): pinned host + device copy, first use
  uint8_t* d_bgr = nullptr;
  size_t bgr_bytes = 0;
  // stateless pair / debug buffers (single stream), allocated lazily
  PyrSet pair_pyr[2];
  uint8_t* pair_frame[2] = {};
  bool pair_ready = false;
  // detector frame / mask staging for the host API
  uint8_t* det_frame = nullptr;
  uint8_t* det_mask = nullptr;
  // table
  uint64_t* d_ids = nullptr;
  double2* d_xy = nullptr;
  uint32_t* d_age = nullptr;
  int* d_n = nullptr;
  float2* d_fwd = nullptr;
  uint8_t* d_flags = nullptr;
  int* d_stats = nullptr;
  int* d_tslot = nullptr;      // [B][cap] LK template cache slot of each table entry
  LkCacheBuf lkc[2] = {};      // LK template cache, ping-pong with the pyramids (tpl == nullptr: disabled)
  size_t lkc_bytes = 0;        // device bytes the cache holds (0: none)
  int epoch = 0;               // bumped by every Track call
  int* d_status = nullptr;
  uint64_t* d_maxidx = nullptr;
  // detector
  int ncell = 0, grid_max = 0, lam_total = 0, cand_cap = 0, cand_cap_alloc = 0, hash_size = 0;
  int max_cw = 0, max_ch = 0;
  int4* d_rects = nullptr;
  float2* d_origin = nullptr;
  int* d_lam_off = nullptr;
  int* d_need = nullptr;
  int* d_nact = nullptr;
  unsigned int* d_cellmax = nullptr;
  int* d_candcnt = nullptr;
  unsigned long long* d_cand = nullptr;
  float* d_lam = nullptr;
  double2* d_selxy = nullptr;
  int* d_selcnt = nullptr;
  float k_side = 0.f, k_mid = 0.f;
  // ANMS detector (detector_kind == VSF_DETECTOR_ANMS)
  AnmsParams anms;
  float ak_side = 0.f, ak_mid = 0.f;
  uint8_t* d_aflag = nullptr;             // [B][H][W] the mask GFTT sees
  int4* d_arect = nullptr;                // the frame as a single cell
  float2* d_aorigin = nullptr;
  int* d_alamoff = nullptr;
  int* d_anact = nullptr;                 // [B] all ones
  unsigned long long* d_acand = nullptr;  // [B][cand_cap]
  short2* d_akp = nullptr;                // [B][kcap] detected keypoints, cv order
  int* d_andet = nullptr;                 // [B]
  double2* d_aselxy = nullptr;            // [B][kcap]
  int* d_aselcnt = nullptr;               // [B]
  // LSSD tracker (tracker_kind == VSF_TRACKER_LSSD): u16 mip levels >= 1, two sets (prev / cur)
  LssdParams lssd;
  int l16w[LSSD_MAXL] = {}, l16h[LSSD_MAXL] = {};
  uint16_t* lssd_pyr[2][LSSD_MAXL] = {};
  uint16_t* lssd_pair_pyr[2][LSSD_MAXL] = {};  // single-stream debug pair, allocated lazily
  uint64_t* d_lssd_ids = nullptr;  // the tracker's own id -> translation map (see LssdState)
  float2* d_lssd_xy = nullptr;
  int* d_lssd_n = nullptr;
  // frontend step (allocated on first use)
  double* d_fe_dt = nullptr;
  int* d_fe_kf = nullptr;
  unsigned int* d_fe_aged = nullptr;
  // LK
  LkParams lk;
  LkSmem lks;
  // profiling
  unsigned int prof = 0;  // bit k: kernel class k is timed (vsf_profile_enable)
  std::vector<EvPair> ev_used;
  std::vector<EvPair> ev_free;
  char err[256] = {0};
};

namespace {

int fail(vsf_ctx* c, int code, const char* fmt, ...) {
  if (c) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define HIPCHK(c, call)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) return fail(c, VSF_E_HIP, "%s: %s", #call, hipGetErrorString(e_));    \
  } while (0)

// Every extern "C" entry runs on the context's device and leaves the calling thread's current device as it found it.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define VSF_ON_DEVICE(c)                 \
  DeviceGuard dev_guard_((c)->p.device_id); \
  HIPCHK(c, dev_guard_.err)

// The host-pointer entries move everything through ONE pinned block (h_stage) with asynchronous copies on the null
// stream.  Whatever way such an entry returns, no copy may still be reading or writing that block: an immediate
// retry would memcpy into it under a running H2D copy.  The success path synchronises explicitly and disarms.
struct StageSync {
  bool armed = true;
  ~StageSync() {
    if (armed) (void)hipStreamSynchronize(nullptr);
  }
};

void free_lk_cache(vsf_ctx* c) {
  for (int i = 0; i < 2; i++) {
    if (c->lkc[i].tpl) (void)hipFree(c->lkc[i].tpl);
    if (c->lkc[i].hdr) (void)hipFree(c->lkc[i].hdr);
    if (c->lkc[i].tag) (void)hipFree(c->lkc[i].tag);
    c->lkc[i] = LkCacheBuf{};
  }
  c->lkc_bytes = 0;
}

// The LK template cache is the one optional allocation of a context (results are identical without it): when a
// mandatory or lazily allocated buffer does not fit, the cache is given back and the allocation tried again.
template <typename T>
int dalloc(vsf_ctx* c, T** p, size_t count) {
  const size_t bytes = count * sizeof(T) ? count * sizeof(T) : 16;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), bytes);
  if (e == hipErrorOutOfMemory && c->lkc[0].tpl) {
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();  // kernels in flight may still use the cache
    free_lk_cache(c);
    e = hipMalloc(reinterpret_cast<void**>(p), bytes);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(c, e == hipErrorOutOfMemory ? VSF_E_NOMEM : VSF_E_HIP, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
  }
  return VSF_OK;
}

struct ProfScope {
  vsf_ctx* c;
  hipStream_t s;
  EvPair e;
  bool on;
  ProfScope(vsf_ctx* c_, int kid, hipStream_t s_) : c(c_), s(s_), on((c_->prof >> kid) & 1u) {
    if (!on) return;
    if (!c->ev_free.empty()) {
      e = c->ev_free.back();
      c->ev_free.pop_back();
    } else {
      if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) {
        on = false;
        return;
      }
    }
    e.kid = kid;
    (void)hipEventRecord(e.a, s);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e.b, s);
    c->ev_used.push_back(e);
  }
};

PyrView make_view(const vsf_ctx* c, const FrameRef& f, const PyrSet& ps) {
  PyrView v;
  memset(&v, 0, sizeof(v));
  v.nlevels = c->nlevels;
  for (int l = 0; l < c->nlevels; l++) {
    v.lv[l].w = c->lw[l];
    v.lv[l].h = c->lh[l];
    if (l == 0) {
      v.lv[0].base = f.ptr;
      v.lv[0].sstride = f.sstride;
      v.lv[0].pitch = f.pitch;
    } else {
      v.lv[l].base = ps.lv[l];
      v.lv[l].sstride = (long long)c->lp[l] * c->lh[l];
      v.lv[l].pitch = c->lp[l];
    }
    v.lv[l].aligned = ((uintptr_t)v.lv[l].base % 4 == 0) && v.lv[l].sstride % 4 == 0 && v.lv[l].pitch % 4 == 0;
  }
  return v;
}

// pyramid geometry: cv::buildOpticalFlowPyramid stops when the NEXT level would be <= the window
int lk_pyramid_geometry(int width, int height, int win, int max_level, int* lw, int* lh, int* lp) {
  int w = width, h = height, l = 0;
  for (;; l++) {
    lw[l] = w;
    lh[l] = h;
    lp[l] = (w + 3) & ~3;
    w = (w + 1) / 2;
    h = (h + 1) / 2;
    if (l == max_level || w <= win || h <= win) break;
  }
  return l + 1;
}

// LK lane mapping and the clamped termination criteria
LkParams make_lk_params(const vsf_params* p) {
  LkParams k;
  memset(&k, 0, sizeof(k));
  k.win = p->klt_win;
  k.nseg = 64 / k.win;
  k.seglen = (k.win + k.nseg - 1) / k.nseg;
  k.nseg_inv = (65536 + k.nseg - 1) / k.nseg;
  int mc = p->klt_max_iter;  // cv: TermCriteria clamp
  mc = mc < 0 ? 0 : (mc > 100 ? 100 : mc);
  k.max_count = mc;
  double e = p->klt_eps;
  e = e < 0. ? 0. : (e > 10. ? 10. : e);
  k.eps2 = e * e;
  if (k.eps2 >= 1e-30) {
    k.eps2_lo = nextafterf((float)(k.eps2 * (1.0 - 1e-6)), 0.f);
    k.eps2_hi = nextafterf((float)(k.eps2 * (1.0 + 1e-6)), INFINITY);
  } else {  // degenerate eps: always decide in double
    k.eps2_lo = -1.f;
    k.eps2_hi = INFINITY;
  }
  k.min_eig_thr = p->min_eig_thr;
  k.min_eig_c1 = nextafterf((float)((p->min_eig_thr > 0. ? p->min_eig_thr : 0.) * 2.0 * k.win * k.win * 1.001), INFINITY);
  k.backtrack_thr = p->backtrack_thr;
  k.jmargin = LK_JMARGIN;
  return k;
}

int alloc_pyr(vsf_ctx* c, PyrSet* ps, int streams) {
  for (int l = 1; l < c->nlevels; l++) {
    int rc = dalloc(c, &ps->lv[l], (size_t)streams * c->lp[l] * c->lh[l] + 16);
    if (rc) return rc;
  }
  return VSF_OK;
}

// Vertically adjacent tiles one block of pyr_down_dword_kernel walks (it prefetches the next tile while it filters
// the current one): up to 5, fewer while that would leave the chip short of blocks.
int pyr_tiles_y(int tiles_x, int tiles_y_total, int streams) {
  int t = tiles_y_total < 5 ? tiles_y_total : 5;
  while (t > 1 && (long long)tiles_x * ((tiles_y_total + t - 1) / t) * streams < 8192) t--;
  return t;
}

// cv::buildOpticalFlowPyramid levels >= 1 (level 0 is the frame itself); lv[l] = level l of stream 0
void launch_pyramid(const FrameRef& f, uint8_t* const* lv, int nlevels, const int* lw, const int* lh, const int* lp,
                    int streams, hipStream_t s) {
  for (int l = 1; l < nlevels; l++) {
    const uint8_t* src = (l == 1) ? f.ptr : lv[l - 1];
    const long long sst = (l == 1) ? f.sstride : (long long)lp[l - 1] * lh[l - 1];
    const int sp = (l == 1) ? f.pitch : lp[l - 1];
    dim3 grid((lw[l] + PD_TW - 1) / PD_TW, (lh[l] + PD_TH - 1) / PD_TH, streams);
    const long long dst_sst = (long long)lp[l] * lh[l];
    const bool aligned = ((uintptr_t)src % 4 == 0) && sst % 4 == 0 && sp % 4 == 0 && lw[l - 1] % 4 == 0 &&
                         lw[l - 1] >= 8 && ((uintptr_t)lv[l] % 4 == 0) && dst_sst % 4 == 0 && lp[l] % 4 == 0;
    if (aligned) {
      const int ty = pyr_tiles_y(grid.x, grid.y, streams);
      hipLaunchKernelGGL(pyr_down_dword_kernel<false>, dim3(grid.x, (grid.y + ty - 1) / ty, streams), dim3(256), 0, s, src,
                         sst, sp, lw[l - 1], lh[l - 1], (void*)lv[l], dst_sst, lp[l], lw[l], lh[l], ty);
    } else
      hipLaunchKernelGGL(pyr_down_kernel, grid, dim3(256), 0, s, src, sst, sp, lw[l - 1], lh[l - 1], lv[l], dst_sst,
                         lp[l], lw[l], lh[l]);
  }
}

int build_pyramid(vsf_ctx* c, const FrameRef& f, PyrSet& ps, int streams, hipStream_t s) {
  ProfScope prof(c, VSF_K_PYRAMID, s);
  launch_pyramid(f, ps.lv, c->nlevels, c->lw, c->lh, c->lp, streams, s);
  HIPCHK(c, hipGetLastError());
  return VSF_OK;
}

int launch_lk_fb(vsf_ctx* c, const PyrView& P, const PyrView& C, int streams, hipStream_t s, const LkCache& cache) {
  ProfScope prof(c, VSF_K_LK, s);
  // workgroups per stream: the table size a detector maintains (+ slack), not the (larger) capacity
  int fgrid = c->p.max_features + 64;
  if (fgrid > c->cap) fgrid = c->cap;
  // x = 8 XCD residue classes x features, y = groups of eight streams: the linear workgroup id (x + gridDim.x * y,
  // the order of dispatch) modulo 8 is the stream's residue class, and the kernel reads stream and feature off the two
  // indices without a division
  dim3 grid((unsigned)(8 * fgrid), (unsigned)((streams + 7) / 8));
#define VSF_LAUNCH_W(SEG)                                                                                       \
  hipLaunchKernelGGL(lk_fb_kernel<SEG>, grid, dim3(64), c->lks.total, s, P, C, c->d_xy, c->d_n, c->cap, fgrid, \
                     streams, c->lk, c->d_fwd, c->d_flags, cache)
#define VSF_LAUNCH_WIN(SEG, WIN)                                                                                \
  hipLaunchKernelGGL((lk_fb_kernel<SEG, WIN>), grid, dim3(64), c->lks.total, s, P, C, c->d_xy, c->d_n, c->cap, \
                     fgrid, streams, c->lk, c->d_fwd, c->d_flags, cache)
  // the windows the reference's configurations use (cv's default 21; 15 and 24 in its examples): geometry folded
  if (c->lk.jmargin == LK_JMARGIN && c->lk.win == 21) {
    VSF_LAUNCH_WIN(7, 21);
  } else if (c->lk.jmargin == LK_JMARGIN && c->lk.win == 15) {
    VSF_LAUNCH_WIN(7, 15);
  } else if (c->lk.jmargin == LK_JMARGIN && c->lk.win == 24) {
    VSF_LAUNCH_WIN(16, 24);
  } else {
    switch (lk_seg_template(c->lk.seglen)) {
      case 7: VSF_LAUNCH_W(7); break;
      case 8: VSF_LAUNCH_W(8); break;
      default: VSF_LAUNCH_W(16); break;
    }
  }
#undef VSF_LAUNCH_W
#undef VSF_LAUNCH_WIN
  HIPCHK(c, hipGetLastError());
  return VSF_OK;
}

int launch_compact(vsf_ctx* c, int streams, hipStream_t s) {
  ProfScope prof(c, VSF_K_COMPACT, s);
  hipLaunchKernelGGL(compact_kernel, dim3(streams), dim3(256), 0, s, c->d_ids, c->d_xy, c->d_age, c->d_n, c->cap,
                     c->d_fwd, c->d_flags, c->d_stats, c->d_tslot);
  HIPCHK(c, hipGetLastError());
  return VSF_OK;
}

GridView grid_view(const vsf_ctx* c) {
  GridView g;
  g.rects = c->d_rects;
  g.origin = c->d_origin;
  g.lam_off = c->d_lam_off;
  g.ncell = c->ncell;
  g.rows = c->p.grid_rows;
  g.cols = c->p.grid_cols;
  g.grid_max = c->grid_max;
  g.lam_total = c->lam_total;
  return g;
}

// basalt ManagedImagePyr::setFromImage on `streams` frames: u16 levels 1..levels
int lssd_build_pyramid(vsf_ctx* c, const FrameRef& f, uint16_t* const* lv, int streams, hipStream_t s) {
  ProfScope prof(c, VSF_K_PYRAMID, s);
  for (int l = 1; l <= c->lssd.levels; l++) {
    const int dw = c->l16w[l], dh = c->l16h[l];
    if (dw <= 0 || dh <= 0) continue;
    dim3 grid((dw + PD_TW - 1) / PD_TW, (dh + PD_TH - 1) / PD_TH, streams);
    // level 1 from a 4-byte aligned u8 frame of even width: the dword pyrDown kernel, unrounded (see its comment).
    // basalt's level is (w >> 1) wide, cv's is (w + 1) >> 1: the same for even w, and then the right border
    // (column w -> w - 2) is the same reflection too.
    if (l == 1 && ((uintptr_t)f.ptr % 4 == 0) && f.sstride % 4 == 0 && f.pitch % 4 == 0 && c->l16w[0] % 4 == 0 &&
        c->l16w[0] >= 8 && ((uintptr_t)lv[1] % 8 == 0) && ((long long)dw * dh) % 4 == 0) {
      const int ty = pyr_tiles_y(grid.x, grid.y, streams);
      hipLaunchKernelGGL(pyr_down_dword_kernel<true>, dim3(grid.x, (grid.y + ty - 1) / ty, streams), dim3(256), 0, s, f.ptr,
                         f.sstride, f.pitch, c->l16w[0], c->l16h[0], (void*)lv[1], (long long)dw * dh, dw, dw, dh, ty);
    } else if (l == 1)
      hipLaunchKernelGGL(lssd_subsample_kernel<true>, grid, dim3(256), 0, s, f.ptr, f.sstride, f.pitch, c->l16w[0],
                         c->l16h[0], lv[1], dw, dh);
    else
      hipLaunchKernelGGL(lssd_subsample_kernel<false>, grid, dim3(256), 0, s, lv[l - 1],
                         (long long)c->l16w[l - 1] * c->l16h[l - 1], c->l16w[l - 1], c->l16w[l - 1], c->l16h[l - 1],
                         lv[l], dw, dh);
  }
  HIPCHK(c, hipGetLastError());
  return VSF_OK;
}

Pyr16View make_view16(const vsf_ctx* c, const FrameRef& f, uint16_t* const* lv) {
  Pyr16View v;
  memset(&v, 0, sizeof(v));
  v.base0 = f.ptr;
  v.sstride0 = f.sstride;
  v.pitch0 = f.pitch;
  v.levels = c->lssd.levels;
  for (int l = 0; l <= c->lssd.levels; l++) {
    v.w[l] = c->l16w[l];
    v.h[l] = c->l16h[l];
    v.lv[l] = l ? lv[l] : nullptr;
  }
  return v;
}

int lssd_track_impl(vsf_ctx* c, const FrameRef& f, int streams, hipStream_t s) {
  const int nxt = c->cur ^ 1;
  int rc = lssd_build_pyramid(c, f, c->lssd_pyr[nxt], streams, s);
  if (rc) return rc;
  c->track_frame[nxt] = f;
  if (c->has_prev) {  // ref: FeatureTrackerLSSDLucasKanade.cpp:57-68 -- first call only stores the pyramid
    const Pyr16View P = make_view16(c, c->track_frame[c->cur], c->lssd_pyr[c->cur]);
    const Pyr16View C = make_view16(c, f, c->lssd_pyr[nxt]);
    {
      ProfScope prof(c, VSF_K_LK, s);
      int fgrid = (c->p.max_features + 64 + 1) / 2;  // feature pairs per stream
      if (fgrid > (c->cap + 1) / 2) fgrid = (c->cap + 1) / 2;
      dim3 grid((unsigned)(8 * fgrid), (unsigned)((streams + 7) / 8));
      const LssdState st = {c->d_lssd_ids, c->d_lssd_xy, c->d_lssd_n};
      hipLaunchKernelGGL(lssd_fb_kernel, grid, dim3(64), 0, s, P, C, c->d_ids, c->d_xy, c->d_n, c->cap, fgrid, streams,
                         c->lssd, st, c->d_fwd, c->d_flags);
      HIPCHK(c, hipGetLastError());
    }
    rc = launch_compact(c, streams, s);
    if (rc) return rc;
  }
  {
    int gx = (c->p.max_features + 64 + 255) / 256;
    if (gx > (c->cap + 255) / 256) gx = (c->cap + 255) / 256;
    hipLaunchKernelGGL(lssd_snapshot_kernel, dim3(gx, streams), dim3(256), 0, s, c->d_ids, c->d_xy, c->d_n, c->cap,
                       c->d_lssd_ids, c->d_lssd_xy, c->d_lssd_n);
    HIPCHK(c, hipGetLastError());
  }
  c->cur = nxt;
  c->has_prev = true;
  return VSF_OK;
}

// Track on `streams` streams given the current frames (already on the device).
int track_impl(vsf_ctx* c, const FrameRef& f, int streams, hipStream_t s) {
  if (c->p.tracker_kind == VSF_TRACKER_LSSD) return lssd_track_impl(c, f, streams, s);
  const int nxt = c->cur ^ 1;
  int rc = build_pyramid(c, f, c->track_pyr[nxt], streams, s);
  if (rc) return rc;
  c->track_frame[nxt] = f;
  const int rd_epoch = c->epoch;
  c->epoch += 1;  // every Track call: entries written before a store-only call or a reset can never match again
  if (c->has_prev) {  // ref: FeatureTrackerLucasKanade.cpp:29-32 -- first call only stores the frame
    const PyrView P = make_view(c, c->track_frame[c->cur], c->track_pyr[c->cur]);
    const PyrView C = make_view(c, f, c->track_pyr[nxt]);
    // template cache: read what the previous call's backward pass left for the image that is now "previous",
    // write this call's into the other buffer (whole batch only: the buffers are indexed by stream)
    LkCache cache;
    memset(&cache, 0, sizeof(cache));
    if (c->lkc[0].tpl && streams == c->B) {
      cache.rd = c->lkc[c->cur];
      cache.wr = c->lkc[nxt];
      cache.tslot = c->d_tslot;
      cache.rd_epoch = rd_epoch;
      cache.wr_epoch = c->epoch;
      cache.levels = c->nlevels;
      cache.enabled = 1;
    }
    rc = launch_lk_fb(c, P, C, streams, s, cache);
    if (rc) return rc;
    rc = launch_compact(c, streams, s);
    if (rc) return rc;
  }
  c->cur = nxt;
  c->has_prev = true;
  return VSF_OK;
}

int detect_anms_impl(vsf_ctx* c, const FrameRef& f, const uint8_t* masks, long long mask_sstride, int mask_pitch,
                     int streams, hipStream_t s);

// candidate keys select_kernel keeps in LDS (more survivors of a cell stay in its global list)
#ifndef VSF_SEL_LDS_KEYS
#define VSF_SEL_LDS_KEYS 1024
#endif

int detect_impl(vsf_ctx* c, const FrameRef& f, const uint8_t* masks, long long mask_sstride, int mask_pitch,
                int streams, hipStream_t s) {
  if (c->p.detector_kind == VSF_DETECTOR_ANMS) return detect_anms_impl(c, f, masks, mask_sstride, mask_pitch, streams, s);
  const GridView G = grid_view(c);
  const int W = c->p.width, H = c->p.height;
  {
    ProfScope prof(c, VSF_K_OCCUPANCY, s);
    const size_t sm = (size_t)(c->hash_size + c->ncell + 2 * (G.cols + G.rows)) * sizeof(int);
    const int nt = streams >= 1024 ? 256 : (streams >= 512 ? 512 : 1024);
    hipLaunchKernelGGL(occupancy_kernel, dim3(streams), dim3(nt), sm, s, c->d_xy, c->d_n, c->cap, G, W, H,
                       c->hash_size, c->d_need, c->d_nact, c->d_cellmax, c->d_candcnt);
  }
  const int tiles_x = sc_tiles(c->max_cw, SC_IW), tiles_y = sc_tiles(c->max_ch, SC_IH);
  {
    ProfScope prof(c, VSF_K_SCORE, s);
    // (cell-major on purpose: with the ANMS detector's stream-major order -- the tiles of a cell spread in time, so
    // that later tiles start with the cell maximum the earlier ones found and skip more square roots -- the kernel
    // took 1.43 instead of 1.26 ms per 1024 frames at C2: the tiles of 1024 frames in flight together miss the L2)
    // blockIdx.z = (stream - stream0) * cells + cell: at most 65535 of them a launch
    const unsigned int cell_magic =
        c->ncell == 1 ? 0u : (unsigned int)((0x100000000ull + (unsigned)c->ncell - 1) / (unsigned)c->ncell);
    const int per_launch = 65535 / c->ncell;
    for (int s0 = 0; s0 < streams; s0 += per_launch) {
      const int ns = streams - s0 < per_launch ? streams - s0 : per_launch;
      hipLaunchKernelGGL(score_kernel<3>, dim3(tiles_x, tiles_y, ns * c->ncell), dim3(SC_NT), 0, s, f.ptr, f.sstride,
                         f.pitch, W, H, masks, mask_sstride, mask_pitch, G, c->d_nact, tiles_x, tiles_y, c->k_side,
                         c->k_mid, c->p.quality_level, c->d_cellmax, c->d_cand, c->cand_cap, c->d_candcnt, 0, 0.0, s0,
                         cell_magic);
    }
  }
  {  // cells whose candidate list overflowed (rare): their lambda map, for select_kernel's exact fallback
    ProfScope prof(c, VSF_K_CANDIDATES, s);
    hipLaunchKernelGGL(lam_fill_kernel<3>, dim3(c->ncell, streams), dim3(SC_NT), 0, s, f.ptr, f.sstride, f.pitch, W, H,
                       masks, mask_sstride, mask_pitch, G, c->d_nact, tiles_x, tiles_y, c->k_side, c->k_mid, c->d_lam,
                       c->d_cellmax, c->d_candcnt, c->cand_cap, -1, 0.0);
  }
  {
    ProfScope prof(c, VSF_K_SELECT, s);
    SelParams sp;
    sp.min_dist = c->p.min_distance;
    sp.quality = c->p.quality_level;
    sp.cap = c->cap;
    sp.cand_cap = c->cand_cap;
    sp.sel_cap = c->grid_max > 0 ? c->grid_max : 1;
    sp.lds_keys = c->cand_cap < VSF_SEL_LDS_KEYS ? c->cand_cap : VSF_SEL_LDS_KEYS;
    const size_t sm = (size_t)sp.lds_keys * 8 + (size_t)sp.sel_cap * 12;
    hipLaunchKernelGGL(select_kernel, dim3(c->ncell, streams), dim3(256), sm, s, c->d_lam, G, c->d_nact, c->d_need,
                       c->d_cellmax, masks, mask_sstride, mask_pitch, c->d_cand, c->d_candcnt, sp, c->d_xy, c->d_n,
                       c->d_selxy, c->d_selcnt);
  }
  {
    ProfScope prof(c, VSF_K_APPEND, s);
    const int sel_cap = c->grid_max > 0 ? c->grid_max : 1;
    const size_t sm = (size_t)(c->ncell + 1 + c->ncell * sel_cap) * sizeof(int);
    hipLaunchKernelGGL(append_kernel, dim3(streams), dim3(256), sm, s, G, c->d_nact, c->d_selxy, c->d_selcnt,
                       sel_cap, c->d_ids, c->d_xy, c->d_age, c->d_n, c->cap, c->d_maxidx, (uint64_t)0, c->d_status,
                       c->d_tslot);
  }
  HIPCHK(c, hipGetLastError());
  return VSF_OK;
}

int anms_ssc_smem(const AnmsParams& a) { return a.kcap * (4 + 2 + 2 + 1) + a.hash_size * 4 + 16; }

// FeatureDetectorANMS::Detect on `streams` streams (ref: FeatureDetectorANMS.cpp:69-212)
int detect_anms_impl(vsf_ctx* c, const FrameRef& f, const uint8_t* masks, long long mask_sstride, int mask_pitch,
                     int streams, hipStream_t s) {
  const AnmsParams& ap = c->anms;
  const int W = ap.w, H = ap.h;
  GridView G;
  G.rects = c->d_arect;
  G.origin = c->d_aorigin;
  G.lam_off = c->d_alamoff;
  G.ncell = 1;
  G.rows = G.cols = 1;
  G.grid_max = ap.kcap;
  G.lam_total = c->lam_total;
  {
    ProfScope prof(c, VSF_K_OCCUPANCY, s);
    hipLaunchKernelGGL(anms_mask_kernel, dim3((W + 1023) / 1024, H, streams), dim3(256), 0, s, masks, mask_sstride,
                       mask_pitch, ap, c->d_aflag);
    int chunks = (c->cap * (2 * ap.radius + 1) + 1023) / 1024;  // four disc rows per thread at a full table
    chunks = chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
    hipLaunchKernelGGL(anms_disc_kernel, dim3(chunks, streams), dim3(256), 0, s, c->d_xy, c->d_n, c->cap, ap,
                       c->d_aflag, c->d_cellmax, c->d_candcnt);
  }
  const int tiles_x = sc_tiles(W, SC_IW), tiles_y = sc_tiles(H, SC_IH);
  const long long fsz = (long long)W * H;
  {
    ProfScope prof(c, VSF_K_SCORE, s);
    auto launch = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(streams, tiles_x, tiles_y), dim3(SC_NT), 0, s, f.ptr, f.sstride, f.pitch, W, H,
                         c->d_aflag, fsz, W, G, c->d_anact, tiles_x, tiles_y, c->ak_side, c->ak_mid, ap.quality,
                         c->d_cellmax, c->d_acand, ap.cand_cap, c->d_candcnt, 1, c->p.anms_harris_k, 0, 0u);
    };
    const bool b4 = c->p.anms_block_size == 4, hr = c->p.anms_use_harris != 0;
    if (b4 && !hr) launch(score_kernel<4, false>);
    if (!b4 && !hr) launch(score_kernel<3, false>);
    if (b4 && hr) launch(score_kernel<4, true>);
    if (!b4 && hr) launch(score_kernel<3, true>);
  }
  {
    ProfScope prof(c, VSF_K_CANDIDATES, s);
    auto launch = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(1, streams), dim3(SC_NT), 0, s, f.ptr, f.sstride, f.pitch, W, H, c->d_aflag, fsz, W,
                         G, c->d_anact, tiles_x, tiles_y, c->ak_side, c->ak_mid, c->d_lam, c->d_cellmax, c->d_candcnt,
                         ap.cand_cap, -1, c->p.anms_harris_k);
    };
    const bool b4 = c->p.anms_block_size == 4, hr = c->p.anms_use_harris != 0;
    if (b4 && !hr) launch(lam_fill_kernel<4, false>);
    if (!b4 && !hr) launch(lam_fill_kernel<3, false>);
    if (b4 && hr) launch(lam_fill_kernel<4, true>);
    if (!b4 && hr) launch(lam_fill_kernel<3, true>);
  }
  {
    ProfScope prof(c, VSF_K_SELECT, s);
    // one workgroup per stream in both kernels: few streams get large workgroups (a stream's few hundred thousand
    // keys behind 256 threads took 3.6 of C4 + ANMS's 10.6 ms per 32 frames), many streams fill the chip anyway
    const int nt = streams >= 1024 ? 256 : (streams >= 512 ? 512 : 1024);
    if (c->p.anms_detection_min_distance > 1.0)  // cv: minDistance >= 1 runs the distance test, which 1 never fails
      hipLaunchKernelGGL(anms_topk_kernel<true>, dim3(streams), dim3(nt), (size_t)ap.gcap * 8, s, c->d_lam,
                         (long long)c->lam_total, c->d_aflag, c->d_cellmax, c->d_acand, c->d_candcnt, ap, c->d_akp,
                         c->d_andet);
    else
      hipLaunchKernelGGL(anms_topk_kernel<false>, dim3(streams), dim3(nt), (size_t)ap.kcap * 8, s, c->d_lam,
                         (long long)c->lam_total, c->d_aflag, c->d_cellmax, c->d_acand, c->d_candcnt, ap, c->d_akp,
                         c->d_andet);
    hipLaunchKernelGGL(anms_ssc_kernel, dim3(streams), dim3(nt), (size_t)anms_ssc_smem(ap), s, c->d_akp, c->d_andet,
                       c->d_n, ap, c->d_aselxy, c->d_aselcnt);
  }
  {
    ProfScope prof(c, VSF_K_APPEND, s);
    const size_t sm = (size_t)(2 + ap.kcap) * sizeof(int);
    hipLaunchKernelGGL(append_kernel, dim3(streams), dim3(256), sm, s, G, c->d_anact, c->d_aselxy, c->d_aselcnt,
                       ap.kcap, c->d_ids, c->d_xy, c->d_age, c->d_n, c->cap, c->d_maxidx, (uint64_t)1, c->d_status,
                       c->d_tslot);
  }
  HIPCHK(c, hipGetLastError());
  return VSF_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is per kernel and process-wide: keep the running maximum.
hipError_t raise_dynamic_lds(const void* fn, int slot, int bytes) {
  static std::mutex mu;
  static int cur[8] = {};
  if (bytes <= 48 * 1024) return hipSuccess;
  std::lock_guard<std::mutex> lock(mu);
  if (bytes <= cur[slot]) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) cur[slot] = bytes;
  return e;
}

// ref: FeatureDetectorShiTomasiBucketing.cpp:13-36 (GenerateGrid) and :144-149 (Rect2f -> Rect)
void host_grid(int w, int h, int cols, int rows, std::vector<int4>& rects, std::vector<float2>& origin) {
  const double dx = (double)w / (double)cols, dy = (double)h / (double)rows;
  for (int x = 0; x < cols; x++)
    for (int y = 0; y < rows; y++) {
      const float lx = (float)(x * dx), ly = (float)(y * dy);
      const float rx = lx + (float)dx, ry = ly + (float)dy;
      const float wf = rx - lx, hf = ry - ly;
      int4 r;
      r.x = (int)lrintf(lx);
      r.y = (int)lrintf(ly);
      r.z = (int)lrintf(wf);
      r.w = (int)lrintf(hf);
      rects.push_back(r);
      float2 o;
      o.x = lx;
      o.y = ly;
      origin.push_back(o);
    }
}

// Host-pointer API staging.  Everything a call moves goes through ONE pinned host block (frame | mask | table
// pack) with asynchronous copies on the default stream and a single synchronisation at the end of the call: a
// 752x480 frame from pageable memory, four table copies up and four down cost 0.49 ms per Track + UpdateDetection
// before (round 1); hipMemcpy2D from pageable memory copies row by row (4.5 ms per frame, measured).
int ensure_stage(vsf_ctx* c) {
  if (c->h_stage) return VSF_OK;
  const size_t fs = (size_t)c->p.width * c->p.height;
  HIPCHK(c, hipHostMalloc((void**)&c->h_stage, 2 * fs + table_pack_bytes(c->cap) + 64, hipHostMallocDefault));
  return dalloc(c, &c->d_pack, table_pack_bytes(c->cap));
}

// slot 0 = frame, 1 = mask
int upload_frame(vsf_ctx* c, uint8_t* dst, const uint8_t* img, int w, int h, int stride, int slot = 0) {
  int rc = ensure_stage(c);
  if (rc) return rc;
  uint8_t* hp = c->h_stage + (size_t)slot * c->p.width * c->p.height;
  if (stride == w)
    memcpy(hp, img, (size_t)w * h);
  else
    for (int y = 0; y < h; y++) memcpy(hp + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
  HIPCHK(c, hipMemcpyAsync(dst, hp, (size_t)w * h, hipMemcpyHostToDevice, nullptr));
  return VSF_OK;
}

uint8_t* host_pack(vsf_ctx* c) { return c->h_stage + 2 * (size_t)c->p.width * c->p.height; }

// the caller's table -> stream 0 of the device table (one copy + one small kernel); also clears the status word
int upload_table0(vsf_ctx* c, const uint64_t* ids, const double* xy, const uint32_t* age, int n) {
  if (n < 0 || n > c->cap) return fail(c, VSF_E_CAPACITY, "table of %d entries exceeds capacity %d", n, c->cap);
  int rc = ensure_stage(c);
  if (rc) return rc;
  uint8_t* hp = host_pack(c);
  memset(hp, 0, 64);
  reinterpret_cast<int*>(hp)[0] = n;
  if (n > 0) {
    memcpy(hp + 64, xy, (size_t)n * 16);
    memcpy(hp + 64 + (size_t)c->cap * 16, ids, (size_t)n * 8);
    memcpy(hp + 64 + (size_t)c->cap * 24, age, (size_t)n * 4);
  }
  // three ranges of the pack hold data; below ~64 KB one copy of the whole block is cheaper than three
  HIPCHK(c, hipMemcpyAsync(c->d_pack, hp, table_pack_bytes(c->cap), hipMemcpyHostToDevice, nullptr));
  hipLaunchKernelGGL(table_unpack_kernel, dim3((n + 255) / 256 > 0 ? (n + 255) / 256 : 1), dim3(256), 0, nullptr, c->d_pack,
                     c->cap, c->d_ids, c->d_xy, c->d_age, c->d_n, c->d_status, c->d_tslot);
  HIPCHK(c, hipGetLastError());
  return VSF_OK;
}

// stream 0 of the device table -> the pinned pack (asynchronous; read it with read_table0 after a synchronisation)
int download_table0_async(vsf_ctx* c) {
  int rc = ensure_stage(c);
  if (rc) return rc;
  hipLaunchKernelGGL(table_pack_kernel, dim3((c->cap + 255) / 256), dim3(256), 0, nullptr, c->d_ids, c->d_xy, c->d_age, c->d_n,
                     c->d_stats, c->d_status, c->cap, c->d_pack);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(host_pack(c), c->d_pack, table_pack_bytes(c->cap), hipMemcpyDeviceToHost, nullptr));
  return VSF_OK;
}

int read_table0(vsf_ctx* c, uint64_t* ids, double* xy, uint32_t* age, int* n, int capacity, vsf_track_stats* stats,
                int* status) {
  const uint8_t* hp = host_pack(c);
  const int* hdr = reinterpret_cast<const int*>(hp);
  const int m = hdr[0];
  if (status) *status = hdr[5];
  if (stats) {
    stats->input = hdr[1];
    stats->forward = hdr[2];
    stats->backward = hdr[3];
    stats->verified = hdr[4];
  }
  if (m > capacity) return fail(c, VSF_E_CAPACITY, "result of %d entries exceeds caller capacity %d", m, capacity);
  if (m > 0) {
    memcpy(xy, hp + 64, (size_t)m * 16);
    memcpy(ids, hp + 64 + (size_t)c->cap * 16, (size_t)m * 8);
    memcpy(age, hp + 64 + (size_t)c->cap * 24, (size_t)m * 4);
  }
  *n = m;
  return VSF_OK;
}

// The reference takes any cv::Mat per call (FeatureDetectorBase.hpp:17-20): a SINGLE-STREAM context whose frame size
// differs from the one it was created for re-creates its buffers for the new size (a fresh context with the same
// parameters, swapped in).  What survives: the detector's max_feature_index_ and the profiling switch.  What does
// not: the tracker's previous frame (cv::calcOpticalFlowPyrLK throws when prev and next differ in size, so a
// tracker that has one refuses the new size instead: `keeps_prev`), test overrides.  Batched contexts keep
// their size (their device-resident batches are sized by it).
int regeom(vsf_ctx* c, int w, int h) {
  if (c->B != 1)
    return fail(c, VSF_E_GEOMETRY, "frame %dx%d does not match the batched context's %dx%d", w, h, c->p.width, c->p.height);
  vsf_params np = c->p;
  np.width = w;
  np.height = h;
  np.capacity = c->cap;
  vsf_ctx* nc = nullptr;
  int rc = vsf_create(&np, &nc);
  if (rc)
    return fail(c, rc == VSF_E_INVALID ? VSF_E_GEOMETRY : rc, "frame %dx%d: the context cannot be re-created for this size (code %d)",
                w, h, rc);
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(nc->d_maxidx, c->d_maxidx, sizeof(uint64_t), hipMemcpyDeviceToDevice) != hipSuccess) {
    vsf_destroy(nc);
    return fail(c, VSF_E_HIP, "regeom: copy of the detector state failed");
  }
  nc->prof = c->prof;
  nc->ev_used.swap(c->ev_used);  // recorded timings stay readable
  nc->ev_free.swap(c->ev_free);
  std::swap(*c, *nc);
  vsf_destroy(nc);  // the old buffers
  return VSF_OK;
}

// keeps_prev: the call continues from a stored previous frame, which pins the size
int check_geom(vsf_ctx* c, int w, int h, int stride, bool keeps_prev = false) {
  if (w <= 0 || h <= 0) return fail(c, VSF_E_INVALID, "frame %dx%d", w, h);
  if (stride < w) return fail(c, VSF_E_INVALID, "stride %d < width %d", stride, w);
  if (w == c->p.width && h == c->p.height) return VSF_OK;
  if (keeps_prev)
    return fail(c, VSF_E_GEOMETRY, "frame %dx%d after a previous frame of %dx%d (cv::calcOpticalFlowPyrLK would throw); "
                "vsf_reset_tracker first", w, h, c->p.width, c->p.height);
  DeviceGuard dg(c->p.device_id);
  if (dg.err != hipSuccess) return fail(c, VSF_E_HIP, "hipSetDevice");
  return regeom(c, w, h);
}

// Checked BEFORE anything runs: once the tracker and the detector have run, the previous frame, the id counter and
// the device table have advanced, and a caller whose arrays turn out too short could not retry the frame.
int check_caller_capacity(vsf_ctx* c, int n, int capacity) {
  if (n < 0 || n > c->cap) return fail(c, VSF_E_CAPACITY, "table of %d entries exceeds capacity %d", n, c->cap);
  if (n > capacity) return fail(c, VSF_E_CAPACITY, "table of %d entries exceeds the caller's capacity %d", n, capacity);
  // what the call can leave in the table: the entries it came with plus what the detector can add
  long long most = (long long)n + (c->p.detector_kind == VSF_DETECTOR_ANMS
                                       ? (long long)c->anms.nfeatures
                                       : (long long)c->ncell * (c->grid_max > 0 ? c->grid_max : 0));
  if (most > c->cap) most = c->cap;
  if (capacity < most)
    return fail(c, VSF_E_CAPACITY, "caller capacity %d below the %lld entries this call may produce; nothing was run",
                capacity, most);
  return VSF_OK;
}

int ensure_pair(vsf_ctx* c) {
  if (c->pair_ready) return VSF_OK;
  const size_t fs = (size_t)c->p.width * c->p.height;
  for (int i = 0; i < 2; i++) {
    int rc = dalloc(c, &c->pair_frame[i], fs);
    if (rc) return rc;
    rc = alloc_pyr(c, &c->pair_pyr[i], 1);
    if (rc) return rc;
  }
  c->pair_ready = true;
  return VSF_OK;
}

FrameRef own_ref(const vsf_ctx* c, const uint8_t* p) {
  FrameRef f;
  f.ptr = p;
  f.sstride = (long long)c->p.width * c->p.height;
  f.pitch = c->p.width;
  return f;
}

}  // namespace

extern "C" {

const char* vsf_version(void) { return "vsf-hip 0.1 (gfx950)"; }

void vsf_default_params(vsf_params* p) {
  memset(p, 0, sizeof(*p));
  p->grid_rows = 4;
  p->grid_cols = 4;
  p->max_features = 300;
  p->min_distance = 2.0;
  p->quality_level = 0.1;
  p->block_size = 3;
  p->klt_max_iter = 30;
  p->klt_eps = 0.01;
  p->klt_win = 21;
  p->klt_max_level = 3;
  p->min_eig_thr = 1e-4;
  p->backtrack_thr = 1.0;
  p->width = 752;
  p->height = 480;
  p->streams = 1;
  p->capacity = 0;
  p->device_id = 0;
  p->detector_kind = VSF_DETECTOR_BUCKETING;
  p->anms_detection_features = 0;  // 2 * max_features
  p->anms_quality_level = 0.001;
  p->anms_block_size = 4;
  p->anms_ignore_edge = 20;
  p->anms_tolerance = 0.1;
  p->anms_detection_min_distance = 1.0;
  p->anms_use_harris = 0;
  p->anms_harris_k = 0.04;
  p->tracker_kind = VSF_TRACKER_LK;
  p->lssd_levels = 3;
  p->lssd_max_iterations = 5;
  p->lssd_max_recovered_dist = 0.09;
  p->lk_cache_max_bytes = 0;  // default limit (16 GiB)
}

const char* vsf_last_error(const vsf_ctx* ctx) { return ctx ? ctx->err : "null context"; }

void vsf_destroy(vsf_ctx* c) {
  if (!c) return;
  DeviceGuard dev_guard_(c->p.device_id);
  (void)hipDeviceSynchronize();
  for (int i = 0; i < 2; i++) {
    for (int l = 0; l < MAXL; l++) {
      if (c->track_pyr[i].lv[l]) (void)hipFree(c->track_pyr[i].lv[l]);
      if (c->pair_pyr[i].lv[l]) (void)hipFree(c->pair_pyr[i].lv[l]);
    }
    for (int l = 0; l < LSSD_MAXL; l++) {
      if (c->lssd_pyr[i][l]) (void)hipFree(c->lssd_pyr[i][l]);
      if (c->lssd_pair_pyr[i][l]) (void)hipFree(c->lssd_pair_pyr[i][l]);
    }
    if (c->own_frame[i]) (void)hipFree(c->own_frame[i]);
    if (c->pair_frame[i]) (void)hipFree(c->pair_frame[i]);
  }
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->d_pack) (void)hipFree(c->d_pack);
  if (c->h_bgr) (void)hipHostFree(c->h_bgr);
  if (c->d_bgr) (void)hipFree(c->d_bgr);
  if (c->d_tslot) (void)hipFree(c->d_tslot);
  free_lk_cache(c);
  void* lssd_bufs[] = {c->d_lssd_ids, c->d_lssd_xy, c->d_lssd_n};
  for (void* b : lssd_bufs)
    if (b) (void)hipFree(b);
  void* bufs[] = {c->det_frame, c->det_mask, c->d_ids,    c->d_xy,      c->d_age,   c->d_n,      c->d_fwd,
                  c->d_flags,   c->d_stats,  c->d_status, c->d_maxidx,  c->d_rects, c->d_origin, c->d_lam_off,
                  c->d_need,    c->d_nact,   c->d_cellmax, c->d_candcnt, c->d_cand,  c->d_lam,    c->d_selxy,
                  c->d_selcnt,  c->d_aflag,  c->d_arect,  c->d_aorigin, c->d_alamoff, c->d_anact, c->d_acand,
                  c->d_akp,     c->d_andet,  c->d_aselxy, c->d_aselcnt, c->d_fe_dt,  c->d_fe_kf,  c->d_fe_aged};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  for (auto& e : c->ev_used) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (auto& e : c->ev_free) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  delete c;
}

int vsf_create(const vsf_params* p, vsf_ctx** out) {
  if (!p || !out) return VSF_E_INVALID;
  *out = nullptr;
  if (p->width <= 0 || p->height <= 0 || p->streams < 1 || p->streams > 65535) return VSF_E_INVALID;
  if (p->width > 65535 || p->height > 65535) return VSF_E_INVALID;  // candidate keys carry y << 16 | x
  if (p->klt_win < 3 || p->klt_win > MAX_WIN || p->klt_max_level < 0 || p->klt_max_level >= MAXL)
    return VSF_E_INVALID;
  // the LK window must fit the frame; an LSSD context never runs the LK kernels
  if (p->tracker_kind != VSF_TRACKER_LSSD && (p->width <= p->klt_win || p->height <= p->klt_win)) return VSF_E_INVALID;
  if (p->grid_rows < 1 || p->grid_cols < 1 || p->max_features < 0 || p->block_size != 3) return VSF_E_INVALID;
  if ((long long)p->grid_rows * p->grid_cols > 65535) return VSF_E_INVALID;  // a cell index travels in blockIdx.z
  // quality in [0, 1]: cv accepts more, but a threshold above the maximum keeps nothing, and the fused score kernel's
  // "local maximum first, threshold later" is equivalent to cv's order only in this range (vsf_kernels.hpp)
  if (!(p->min_distance >= 0.0) || !(p->quality_level >= 0.0) || p->quality_level > 1.0) return VSF_E_INVALID;
  if ((long long)p->width * p->height > 0x7fffffffLL / 4) return VSF_E_INVALID;  // 32-bit raster / map indices
  const bool use_anms = p->detector_kind == VSF_DETECTOR_ANMS;
  if (p->detector_kind != VSF_DETECTOR_BUCKETING && !use_anms) return VSF_E_INVALID;
  int anms_k = 0;
  if (use_anms) {
    anms_k = p->anms_detection_features > 0 ? p->anms_detection_features : 2 * p->max_features;
    if (anms_k < 1 || anms_k > 8192 || (int)p->min_distance > ANMS_MAX_RADIUS || (int)p->min_distance < 0)
      return VSF_E_INVALID;
    if (p->anms_block_size != 3 && p->anms_block_size != 4) return VSF_E_INVALID;
    if (!(p->anms_quality_level >= 0.0) || p->anms_quality_level > 1.0 || !(p->anms_tolerance >= 0.0) ||
        p->anms_ignore_edge < 0)
      return VSF_E_INVALID;
    if (p->width > 32767 || p->height > 32767) return VSF_E_INVALID;  // keypoints travel as short2
    // cv's minDistance: anything up to 1 (0 and negatives included) never rejects a corner
    if (!(p->anms_detection_min_distance <= (double)ANMS_MAX_RADIUS)) return VSF_E_INVALID;
    if (p->anms_use_harris && !std::isfinite(p->anms_harris_k)) return VSF_E_INVALID;
    if ((long long)sc_tiles(p->width, SC_IW) * sc_tiles(p->height, SC_IH) > 65535) return VSF_E_INVALID;  // grid.y
  }
  const bool use_lssd = p->tracker_kind == VSF_TRACKER_LSSD;
  if (p->tracker_kind != VSF_TRACKER_LK && !use_lssd) return VSF_E_INVALID;
  if (use_lssd && (p->lssd_levels < 0 || p->lssd_levels >= LSSD_MAXL || p->lssd_max_iterations < 0 ||
                   (p->width >> p->lssd_levels) < 1 || (p->height >> p->lssd_levels) < 1))
    return VSF_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p->device_id < 0 || p->device_id >= ndev)
    return VSF_E_HIP;  // no GPU: fail loudly, there is no CPU path
  DeviceGuard dev_guard_(p->device_id);
  if (dev_guard_.err != hipSuccess) return VSF_E_HIP;

  vsf_ctx* c = new (std::nothrow) vsf_ctx();
  if (!c) return VSF_E_NOMEM;
  c->p = *p;
  c->B = p->streams;
  // the ANMS detector may add up to detection_feature_number corners in one call (:184-186)
  c->cap = p->capacity > 0 ? p->capacity : p->max_features + 64 + (use_anms ? anms_k : 0);
  const int B = c->B;

  c->nlevels = lk_pyramid_geometry(p->width, p->height, p->klt_win, p->klt_max_level, c->lw, c->lh, c->lp);
  c->lk = make_lk_params(p);
  c->lks = lk_smem_layout(c->lk.win, c->lk.jmargin);
  // detector grid
  std::vector<int4> rects;
  std::vector<float2> origin;
  host_grid(p->width, p->height, p->grid_cols, p->grid_rows, rects, origin);
  c->ncell = (int)rects.size();
  c->grid_max = p->max_features / (p->grid_rows * p->grid_cols);  // Q3
  std::vector<int> lam_off(c->ncell + 1, 0);
  for (int i = 0; i < c->ncell; i++) {
    const int4 r = rects[i];
    if (r.x < 0 || r.y < 0 || r.z < 0 || r.w < 0 || r.x + r.z > p->width || r.y + r.w > p->height) {
      delete c;
      return VSF_E_GEOMETRY;
    }
    lam_off[i + 1] = lam_off[i] + r.z * r.w;
    if (r.z > c->max_cw) c->max_cw = r.z;
    if (r.w > c->max_ch) c->max_ch = r.w;
  }
  c->lam_total = lam_off[c->ncell];
  if (c->max_cw < 1) c->max_cw = 1;
  if (c->max_ch < 1) c->max_ch = 1;
  if (c->grid_max > 8192) {
    delete c;
    return VSF_E_INVALID;
  }
  // Candidate slots per cell = LDS-resident keys of select_kernel.  Strict 3x3 local maxima above the
  // quality threshold are a small fraction of a cell; a sixth of its pixels covers textured content with
  // margin (beyond it select_kernel falls back to re-deriving candidates from the lambda map: exact, but two
  // orders of magnitude slower -- with the former fixed 2048 the 320x180 cells of C3 hit that every frame).
  {
    long long want = ((long long)c->max_cw * c->max_ch / 6 + 255) & ~255LL;
    const long long lds_room = (150 * 1024 - (long long)(c->grid_max > 0 ? c->grid_max : 1) * 12) / 8;
    if (want < 2048) want = 2048;
    if (want > 16384) want = 16384;
    if (want > lds_room) want = lds_room & ~255LL;
    c->cand_cap = (int)want;
    c->cand_cap_alloc = c->cand_cap;
  }
  c->hash_size = 64;
  while (c->hash_size < 2 * c->cap) c->hash_size <<= 1;
  if ((size_t)(c->hash_size + c->ncell + 2 * (p->grid_cols + p->grid_rows)) * 4 > 150 * 1024) {
    delete c;
    return VSF_E_INVALID;
  }
  // append_kernel keeps one slot per selectable corner in LDS
  if ((size_t)(c->ncell + 1 + (size_t)c->ncell * (c->grid_max > 0 ? c->grid_max : 1)) * 4 > 150 * 1024 ||
      (use_anms && (size_t)(2 + 2 * (size_t)anms_k) * 4 > 150 * 1024)) {
    delete c;
    return VSF_E_INVALID;
  }
  {
    const double scale = 1.0 / ((double)(1 << 2) * p->block_size * 255.0);  // cv: cornerEigenValsVecs
    c->k_side = (float)(1.0 * scale);
    c->k_mid = (float)(2.0 * scale);
  }

  if (use_lssd) {
    c->lssd.levels = p->lssd_levels;
    c->lssd.max_iterations = p->lssd_max_iterations;
    c->lssd.max_recovered_dist = p->lssd_max_recovered_dist;
    for (int l = 0; l <= p->lssd_levels; l++) {  // ManagedImagePyr::lvl(l): (w >> l, h >> l)
      c->l16w[l] = p->width >> l;
      c->l16h[l] = p->height >> l;
    }
  }
  if (use_anms) {
    AnmsParams& a = c->anms;
    memset(&a, 0, sizeof(a));
    a.w = p->width;
    a.h = p->height;
    a.max_features = p->max_features;
    a.radius = (int)p->min_distance;  // FeatureDetectorANMS.cpp:127 static_cast<int>
    a.edge = p->anms_ignore_edge;
    a.nfeatures = anms_k;
    a.tolerance = (float)p->anms_tolerance;  // :181 static_cast<float>
    a.kcap = 64;
    while (a.kcap < anms_k) a.kcap <<= 1;
    a.hash_size = 2 * a.kcap;
    a.hash_bits = 0;
    while ((1 << a.hash_bits) < a.hash_size) a.hash_bits++;
    a.cand_cap = (p->width * p->height) / 6;
    if (a.cand_cap < 4096) a.cand_cap = 4096;
    a.quality = p->anms_quality_level;
    a.gcap = a.kcap < 4096 ? 4096 : a.kcap;
    a.grad = (int)std::ceil(p->anms_detection_min_distance > 1.0 ? p->anms_detection_min_distance : 1.0);
    a.gmd2 = p->anms_detection_min_distance * p->anms_detection_min_distance;
    {  // cv::Circle (drawing.cpp), fill: midpoint walk; span on rows +-dy is +-dx and on rows +-dx is +-dy
      for (int i = 0; i <= ANMS_MAX_RADIUS; i++) a.halfw[i] = -1;
      int err = 0, dx = a.radius, dy = 0, plus = 1, minus = (a.radius << 1) - 1;
      while (dx >= dy) {
        if (dx > a.halfw[dy]) a.halfw[dy] = dx;
        if (dy > a.halfw[dx]) a.halfw[dx] = dy;
        dy++;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
      }
    }
    const double scale = 1.0 / ((double)(1 << 2) * p->anms_block_size * 255.0);
    c->ak_side = (float)(1.0 * scale);
    c->ak_mid = (float)(2.0 * scale);
    if (c->lam_total < p->width * p->height) c->lam_total = p->width * p->height;
  }

  int rc = VSF_OK;
#define A(expr)                 \
  do {                          \
    if (!rc) rc = (expr);       \
  } while (0)
  if (!use_lssd)  // the LSSD tracker keeps its own u16 levels
    for (int i = 0; i < 2; i++) A(alloc_pyr(c, &c->track_pyr[i], B));
  const size_t fs = (size_t)p->width * p->height;
  A(dalloc(c, &c->own_frame[0], fs));
  A(dalloc(c, &c->own_frame[1], fs));
  A(dalloc(c, &c->det_frame, fs));
  A(dalloc(c, &c->det_mask, fs));
  const size_t T = (size_t)B * c->cap;
  A(dalloc(c, &c->d_ids, T));
  A(dalloc(c, &c->d_xy, T));
  A(dalloc(c, &c->d_age, T));
  A(dalloc(c, &c->d_n, (size_t)B));
  A(dalloc(c, &c->d_fwd, T));
  A(dalloc(c, &c->d_flags, T));
  A(dalloc(c, &c->d_stats, (size_t)B * 4));
  A(dalloc(c, &c->d_tslot, T));
  A(dalloc(c, &c->d_status, (size_t)B));
  A(dalloc(c, &c->d_maxidx, (size_t)B));
  A(dalloc(c, &c->d_rects, (size_t)c->ncell));
  A(dalloc(c, &c->d_origin, (size_t)c->ncell));
  A(dalloc(c, &c->d_lam_off, (size_t)c->ncell + 1));
  const size_t BC = (size_t)B * c->ncell;
  A(dalloc(c, &c->d_need, BC));
  A(dalloc(c, &c->d_nact, (size_t)B));
  A(dalloc(c, &c->d_cellmax, BC));
  A(dalloc(c, &c->d_candcnt, BC));
  A(dalloc(c, &c->d_cand, BC * c->cand_cap));
  A(dalloc(c, &c->d_lam, (size_t)B * c->lam_total));
  A(dalloc(c, &c->d_selxy, BC * (size_t)(c->grid_max > 0 ? c->grid_max : 1)));
  A(dalloc(c, &c->d_selcnt, BC));
  if (use_lssd) {
    A(dalloc(c, &c->d_lssd_ids, (size_t)B * c->cap));
    A(dalloc(c, &c->d_lssd_xy, (size_t)B * c->cap));
    A(dalloc(c, &c->d_lssd_n, (size_t)B));
    if (!rc && hipMemset(c->d_lssd_n, 0, sizeof(int) * (size_t)B) != hipSuccess) rc = VSF_E_HIP;
    for (int i = 0; i < 2; i++)
      for (int l = 1; l <= p->lssd_levels; l++) A(dalloc(c, &c->lssd_pyr[i][l], (size_t)B * c->l16w[l] * c->l16h[l] + 8));
  }
  if (use_anms) {
    A(dalloc(c, &c->d_aflag, (size_t)B * fs));
    A(dalloc(c, &c->d_arect, (size_t)1));
    A(dalloc(c, &c->d_aorigin, (size_t)1));
    A(dalloc(c, &c->d_alamoff, (size_t)2));
    A(dalloc(c, &c->d_anact, (size_t)B));
    A(dalloc(c, &c->d_acand, (size_t)B * c->anms.cand_cap));
    A(dalloc(c, &c->d_akp, (size_t)B * c->anms.kcap));
    A(dalloc(c, &c->d_andet, (size_t)B));
    A(dalloc(c, &c->d_aselxy, (size_t)B * c->anms.kcap));
    A(dalloc(c, &c->d_aselcnt, (size_t)B));
  }
#undef A
  // LK template cache (vsf_lk.hpp): 2 x streams x capacity x levels x 2.5 KB.  Optional: a context that cannot
  // have it (other tracker, windows above 21, above the byte limit of vsf_params.lk_cache_max_bytes -- 16 GiB by
  // default, negative = never --, allocation failure) just recomputes templates.  It is allocated LAST and is the
  // first thing given back when a later allocation of this context does not fit (dalloc).
  if (!rc && !use_lssd && lk_seg_template(c->lk.seglen) <= 8 && p->lk_cache_max_bytes >= 0) {
    const size_t slots = (size_t)B * c->cap;
    const size_t tpl_b = slots * c->nlevels * LK_TPL_U4 * sizeof(uint4), hdr_b = slots * c->nlevels * sizeof(float4),
                 tag_b = slots * sizeof(float4);
    const size_t limit = p->lk_cache_max_bytes > 0 ? (size_t)p->lk_cache_max_bytes : ((size_t)16 << 30);
    if (2 * (tpl_b + hdr_b + tag_b) <= limit) {
      bool ok = true;
      for (int i = 0; i < 2 && ok; i++) {
        ok = hipMalloc((void**)&c->lkc[i].tpl, tpl_b) == hipSuccess && hipMalloc((void**)&c->lkc[i].hdr, hdr_b) == hipSuccess &&
             hipMalloc((void**)&c->lkc[i].tag, tag_b) == hipSuccess && hipMemset(c->lkc[i].tag, 0, tag_b) == hipSuccess;
      }
      if (!ok) {
        (void)hipGetLastError();
        free_lk_cache(c);
      } else {
        c->lkc_bytes = 2 * (tpl_b + hdr_b + tag_b);
      }
    }
  }
  c->epoch = 1;
  if (!rc) {
    hipError_t e = hipSuccess;
    auto acc = [&](hipError_t x) {
      if (e == hipSuccess) e = x;
    };
    acc(hipMemset(c->d_n, 0, (size_t)B * 4));
    acc(hipMemset(c->d_stats, 0, (size_t)B * 16));
    acc(hipMemset(c->d_status, 0, (size_t)B * 4));
    acc(hipMemset(c->d_maxidx, 0, (size_t)B * 8));  // ref ctor: max_feature_index_(0)
    acc(hipMemset(c->d_flags, 0, T));
    acc(hipMemset(c->d_tslot, 0xff, T * 4));
    acc(hipMemcpy(c->d_rects, rects.data(), rects.size() * sizeof(int4), hipMemcpyHostToDevice));
    acc(hipMemcpy(c->d_origin, origin.data(), origin.size() * sizeof(float2), hipMemcpyHostToDevice));
    acc(hipMemcpy(c->d_lam_off, lam_off.data(), lam_off.size() * sizeof(int), hipMemcpyHostToDevice));
    // dynamic LDS above the default limit needs an explicit opt-in; the attribute belongs to the kernel, not to
    // the context, so only ever raise it (an earlier, larger context still launches with its own size)
    const int occ_sm = (c->hash_size + c->ncell + 2 * (c->p.grid_cols + c->p.grid_rows)) * 4;
    const int sel_sm = (c->cand_cap < VSF_SEL_LDS_KEYS ? c->cand_cap : VSF_SEL_LDS_KEYS) * 8 +
                       (c->grid_max > 0 ? c->grid_max : 1) * 12;
    int app_sm = (c->ncell + 1 + c->ncell * (c->grid_max > 0 ? c->grid_max : 1)) * 4;
    if (use_anms && (2 + c->anms.kcap) * 4 > app_sm) app_sm = (2 + c->anms.kcap) * 4;
    acc(raise_dynamic_lds(reinterpret_cast<const void*>(occupancy_kernel), 0, occ_sm));
    acc(raise_dynamic_lds(reinterpret_cast<const void*>(select_kernel), 1, sel_sm));
    acc(raise_dynamic_lds(reinterpret_cast<const void*>(append_kernel), 2, app_sm));
    if (use_anms) {
      acc(raise_dynamic_lds(reinterpret_cast<const void*>(anms_ssc_kernel), 3, anms_ssc_smem(c->anms)));
      acc(raise_dynamic_lds(reinterpret_cast<const void*>(anms_topk_kernel<false>), 4, c->anms.kcap * 8));
      acc(raise_dynamic_lds(reinterpret_cast<const void*>(anms_topk_kernel<true>), 5, c->anms.gcap * 8));
    }
    if (use_anms) {
      const int4 r = make_int4(0, 0, p->width, p->height);
      const float2 o = make_float2(0.f, 0.f);
      const int lo[2] = {0, p->width * p->height};
      std::vector<int> ones((size_t)B, 1);
      acc(hipMemcpy(c->d_arect, &r, sizeof(r), hipMemcpyHostToDevice));
      acc(hipMemcpy(c->d_aorigin, &o, sizeof(o), hipMemcpyHostToDevice));
      acc(hipMemcpy(c->d_alamoff, lo, sizeof(lo), hipMemcpyHostToDevice));
      acc(hipMemcpy(c->d_anact, ones.data(), (size_t)B * 4, hipMemcpyHostToDevice));
    }
    if (e != hipSuccess) rc = VSF_E_HIP;
  }
  if (rc) {
    vsf_destroy(c);
    return rc;
  }
  *out = c;
  return VSF_OK;
}

int vsf_get_info(const vsf_ctx* c, vsf_info* out) {
  if (!c || !out) return VSF_E_INVALID;
  memset(out, 0, sizeof(*out));
  out->lk_cache_bytes = (int64_t)c->lkc_bytes;
  out->width = c->p.width;
  out->height = c->p.height;
  out->streams = c->B;
  out->capacity = c->cap;
  out->levels = c->nlevels;
  return VSF_OK;
}

int vsf_reset_tracker(vsf_ctx* c) {
  if (!c) return VSF_E_INVALID;
  c->has_prev = false;
  return VSF_OK;
}

int vsf_get_max_feature_index(vsf_ctx* c, int32_t stream, uint64_t* out) {
  if (!c || !out || stream < 0 || stream >= c->B) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  HIPCHK(c, hipMemcpy(out, c->d_maxidx + stream, 8, hipMemcpyDeviceToHost));
  return VSF_OK;
}

int vsf_set_max_feature_index(vsf_ctx* c, int32_t stream, uint64_t value) {
  if (!c || stream < 0 || stream >= c->B) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  HIPCHK(c, hipMemcpy(c->d_maxidx + stream, &value, 8, hipMemcpyHostToDevice));
  return VSF_OK;
}

// ------------------------------------------------------------------------------------------------
// single-stream host API
// ------------------------------------------------------------------------------------------------
int vsf_track(vsf_ctx* c, const uint8_t* img, int32_t w, int32_t h, int32_t stride, uint64_t* ids, double* xy,
              uint32_t* age, int32_t* n_inout, vsf_track_stats* stats) {
  if (!c || !img || !n_inout) return VSF_E_INVALID;
  int rc = check_geom(c, w, h, stride, c->has_prev);
  if (rc) return rc;
  VSF_ON_DEVICE(c);
  const int n = *n_inout;
  if (n > 0 && (!ids || !xy || !age)) return VSF_E_INVALID;
  if (n < 0 || n > c->cap) return fail(c, VSF_E_CAPACITY, "table of %d entries exceeds capacity %d", n, c->cap);
  uint8_t* slot = c->own_frame[c->cur ^ 1];
  StageSync stage_sync;  // from here on copies may be in flight on the pinned block
  rc = upload_frame(c, slot, img, w, h, stride);
  if (rc) return rc;
  const bool first = !c->has_prev;
  if (!first) rc = upload_table0(c, ids, xy, age, n);
  if (!rc) rc = track_impl(c, own_ref(c, slot), 1, nullptr);
  if (!rc && !first) rc = download_table0_async(c);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(nullptr));
  stage_sync.armed = false;
  if (first) {  // :29-32 table untouched
    if (stats) stats->input = stats->forward = stats->backward = stats->verified = n;
    return VSF_OK;
  }
  return read_table0(c, ids, xy, age, n_inout, n, stats, nullptr);
}

int vsf_track_pair(vsf_ctx* c, const uint8_t* prev, int32_t prev_stride, const uint8_t* cur, int32_t cur_stride,
                   int32_t w, int32_t h, uint64_t* ids, double* xy, uint32_t* age, int32_t* n_inout,
                   vsf_track_stats* stats) {
  if (!c || !prev || !cur || !n_inout) return VSF_E_INVALID;
  if (c->p.tracker_kind == VSF_TRACKER_LSSD)  // the reference's LSSD tracker has the stateful 3-argument Track only
    return fail(c, VSF_E_INVALID, "vsf_track_pair: FeatureTrackerLSSDLucasKanade has no frame-pair form");
  int rc = check_geom(c, w, h, prev_stride);
  if (!rc) rc = check_geom(c, w, h, cur_stride);
  if (rc) return rc;
  VSF_ON_DEVICE(c);
  const int n = *n_inout;
  if (n > 0 && (!ids || !xy || !age)) return VSF_E_INVALID;
  StageSync stage_sync;
  rc = ensure_pair(c);
  if (!rc) rc = upload_frame(c, c->pair_frame[0], prev, w, h, prev_stride);
  if (!rc) rc = upload_frame(c, c->pair_frame[1], cur, w, h, cur_stride, 1);
  if (!rc) rc = upload_table0(c, ids, xy, age, n);
  if (rc) return rc;
  const FrameRef fp = own_ref(c, c->pair_frame[0]), fc = own_ref(c, c->pair_frame[1]);
  rc = build_pyramid(c, fp, c->pair_pyr[0], 1, nullptr);
  if (!rc) rc = build_pyramid(c, fc, c->pair_pyr[1], 1, nullptr);
  LkCache nocache;  // the stateless pair form has no previous call
  memset(&nocache, 0, sizeof(nocache));
  if (!rc) rc = launch_lk_fb(c, make_view(c, fp, c->pair_pyr[0]), make_view(c, fc, c->pair_pyr[1]), 1, nullptr, nocache);
  if (!rc) rc = launch_compact(c, 1, nullptr);
  if (!rc) rc = download_table0_async(c);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(nullptr));
  stage_sync.armed = false;
  return read_table0(c, ids, xy, age, n_inout, n, stats, nullptr);
}

}  // extern "C"

namespace {
// vsf_detect / vsf_detect_bgr: cn = 1 (grey) or 3 / 4 interleaved channels, B first (converted on the device)
int detect_host(vsf_ctx* c, const uint8_t* img, int w, int h, int stride, int cn, const uint8_t* mask, int mask_stride,
                uint64_t* ids, double* xy, uint32_t* age, int32_t* n_inout, int capacity) {
  if (!c || !img || !n_inout || !ids || !xy || !age) return VSF_E_INVALID;
  if (cn != 1 && cn != 3 && cn != 4) return fail(c, VSF_E_INVALID, "%d channels (1, 3 or 4)", cn);
  if (stride < w * cn) return fail(c, VSF_E_INVALID, "stride %d < %d bytes per row", stride, w * cn);
  int rc = check_geom(c, w, h, w);
  if (rc) return rc;
  if (mask && mask_stride < w) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  rc = check_caller_capacity(c, *n_inout, capacity);
  if (rc) return rc;
  StageSync stage_sync;
  if (cn == 1) {
    rc = upload_frame(c, c->det_frame, img, w, h, stride);
  } else {
    rc = ensure_stage(c);
    const size_t row = (size_t)w * cn, bytes = row * h;
    if (!rc && c->bgr_bytes < bytes) {  // first multi-channel frame (or a larger one): its own pinned + device pair
      (void)hipStreamSynchronize(nullptr);
      if (c->h_bgr) (void)hipHostFree(c->h_bgr);
      if (c->d_bgr) (void)hipFree(c->d_bgr);
      c->h_bgr = nullptr, c->d_bgr = nullptr, c->bgr_bytes = 0;
      if (hipHostMalloc((void**)&c->h_bgr, bytes, hipHostMallocDefault) != hipSuccess) rc = fail(c, VSF_E_NOMEM, "pinned BGR staging");
      if (!rc) rc = dalloc(c, &c->d_bgr, bytes);
      if (!rc) c->bgr_bytes = bytes;
    }
    if (!rc) {
      if ((size_t)stride == row)
        memcpy(c->h_bgr, img, bytes);
      else
        for (int y = 0; y < h; y++) memcpy(c->h_bgr + (size_t)y * row, img + (size_t)y * stride, row);
      if (hipMemcpyAsync(c->d_bgr, c->h_bgr, bytes, hipMemcpyHostToDevice, nullptr) != hipSuccess) rc = fail(c, VSF_E_HIP, "BGR upload");
    }
    if (!rc) {
      hipLaunchKernelGGL(bgr2gray_kernel, dim3((w + 255) / 256, (h + 3) / 4, 1), dim3(256), 0, nullptr, c->d_bgr,
                         (long long)bytes, (int)row, cn, w, h, c->det_frame, (long long)w * h, w);
      if (hipGetLastError() != hipSuccess) rc = fail(c, VSF_E_HIP, "bgr2gray launch");
    }
  }
  if (!rc && mask) rc = upload_frame(c, c->det_mask, mask, w, h, mask_stride, 1);
  if (!rc) rc = upload_table0(c, ids, xy, age, *n_inout);  // also clears the status word
  if (!rc) rc = detect_impl(c, own_ref(c, c->det_frame), mask ? c->det_mask : nullptr, (long long)w * h, w, 1, nullptr);
  if (!rc) rc = download_table0_async(c);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(nullptr));
  stage_sync.armed = false;
  int st = 0;
  rc = read_table0(c, ids, xy, age, n_inout, capacity, nullptr, &st);
  if (rc) return rc;
  if (st) return fail(c, st, "feature table capacity %d exhausted during detection", c->cap);
  return VSF_OK;
}
}  // namespace

extern "C" {

int vsf_detect(vsf_ctx* c, const uint8_t* img, int32_t w, int32_t h, int32_t stride, const uint8_t* mask,
               int32_t mask_stride, uint64_t* ids, double* xy, uint32_t* age, int32_t* n_inout, int32_t capacity) {
  return detect_host(c, img, w, h, stride, 1, mask, mask_stride, ids, xy, age, n_inout, capacity);
}

int vsf_detect_bgr(vsf_ctx* c, const uint8_t* img, int32_t w, int32_t h, int32_t stride, int32_t channels,
                   const uint8_t* mask, int32_t mask_stride, uint64_t* ids, double* xy, uint32_t* age, int32_t* n_inout,
                   int32_t capacity) {
  if (channels == 1) return VSF_E_INVALID;  // vsf_detect
  return detect_host(c, img, w, h, stride, channels, mask, mask_stride, ids, xy, age, n_inout, capacity);
}

int vsf_batch_bgr_to_gray(vsf_ctx* c, const uint8_t* d_bgr, int64_t bgr_frame_stride, int32_t bgr_row_stride,
                          int32_t channels, uint8_t* d_gray, int64_t gray_frame_stride, int32_t gray_row_stride,
                          void* hip_stream) {
  if (!c || !d_bgr || !d_gray || (channels != 3 && channels != 4)) return VSF_E_INVALID;
  if (bgr_row_stride < c->p.width * channels || gray_row_stride < c->p.width) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  hipLaunchKernelGGL(bgr2gray_kernel, dim3((c->p.width + 255) / 256, (c->p.height + 3) / 4, c->B), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(hip_stream), d_bgr, (long long)bgr_frame_stride, bgr_row_stride, channels,
                     c->p.width, c->p.height, d_gray, (long long)gray_frame_stride, gray_row_stride);
  HIPCHK(c, hipGetLastError());
  return VSF_OK;
}

// Track, then UpdateDetection on the same frame (the loop of example_feature_detection_and_tracking.cpp:43-55) with
// ONE frame upload, one table round trip and one synchronisation.
int vsf_step(vsf_ctx* c, const uint8_t* img, int32_t w, int32_t h, int32_t stride, const uint8_t* mask,
             int32_t mask_stride, uint64_t* ids, double* xy, uint32_t* age, int32_t* n_inout, int32_t capacity,
             vsf_track_stats* stats) {
  if (!c || !img || !n_inout || !ids || !xy || !age) return VSF_E_INVALID;
  int rc = check_geom(c, w, h, stride, c->has_prev);
  if (rc) return rc;
  if (mask && mask_stride < w) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  const int n = *n_inout;
  uint8_t* slot = c->own_frame[c->cur ^ 1];
  const bool first = !c->has_prev;
  rc = check_caller_capacity(c, n, capacity);
  if (rc) return rc;
  StageSync stage_sync;
  rc = upload_frame(c, slot, img, w, h, stride);
  if (!rc && mask) rc = upload_frame(c, c->det_mask, mask, w, h, mask_stride, 1);
  if (!rc) rc = upload_table0(c, ids, xy, age, n);
  if (!rc) rc = track_impl(c, own_ref(c, slot), 1, nullptr);
  if (!rc) rc = detect_impl(c, own_ref(c, slot), mask ? c->det_mask : nullptr, (long long)w * h, w, 1, nullptr);
  if (!rc) rc = download_table0_async(c);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(nullptr));
  stage_sync.armed = false;
  int st = 0;
  rc = read_table0(c, ids, xy, age, n_inout, capacity, first ? nullptr : stats, &st);
  if (rc) return rc;
  if (first && stats) stats->input = stats->forward = stats->backward = stats->verified = n;
  if (st) return fail(c, st, "feature table capacity %d exhausted during detection", c->cap);
  return VSF_OK;
}

// ------------------------------------------------------------------------------------------------
// batched device-resident API
// ------------------------------------------------------------------------------------------------
int vsf_batch_track(vsf_ctx* c, const uint8_t* d_frames, int64_t frame_stride_bytes, int32_t row_stride,
                    void* hip_stream) {
  if (!c || !d_frames || row_stride < c->p.width) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  FrameRef f;
  f.ptr = d_frames;
  f.sstride = frame_stride_bytes;
  f.pitch = row_stride;
  return track_impl(c, f, c->B, reinterpret_cast<hipStream_t>(hip_stream));
}

int vsf_batch_detect(vsf_ctx* c, const uint8_t* d_frames, int64_t frame_stride_bytes, int32_t row_stride,
                     const uint8_t* d_masks, int64_t mask_stride_bytes, int32_t mask_row_stride, void* hip_stream) {
  if (!c || !d_frames || row_stride < c->p.width) return VSF_E_INVALID;
  if (d_masks && mask_row_stride < c->p.width) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  FrameRef f;
  f.ptr = d_frames;
  f.sstride = frame_stride_bytes;
  f.pitch = row_stride;
  return detect_impl(c, f, d_masks, mask_stride_bytes, mask_row_stride, c->B,
                     reinterpret_cast<hipStream_t>(hip_stream));
}

int vsf_batch_step(vsf_ctx* c, const uint8_t* d_frames, int64_t frame_stride_bytes, int32_t row_stride,
                   void* hip_stream) {
  int rc = vsf_batch_track(c, d_frames, frame_stride_bytes, row_stride, hip_stream);
  if (rc) return rc;
  return vsf_batch_detect(c, d_frames, frame_stride_bytes, row_stride, nullptr, 0, 0, hip_stream);
}

int vsf_batch_upload(vsf_ctx* c, int32_t first, int32_t count, const uint64_t* ids, const double* xy,
                     const uint32_t* age, const int32_t* n) {
  if (!c || first < 0 || count < 0 || first + count > c->B || !ids || !xy || !age || !n) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  for (int i = 0; i < count; i++)
    if (n[i] < 0 || n[i] > c->cap) return fail(c, VSF_E_CAPACITY, "stream %d: n=%d exceeds capacity", first + i, n[i]);
  const size_t o = (size_t)first * c->cap, T = (size_t)count * c->cap;
  HIPCHK(c, hipMemcpy(c->d_ids + o, ids, T * 8, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_xy + o, xy, T * 16, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_age + o, age, T * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_n + first, n, (size_t)count * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemset(c->d_tslot + o, 0xff, T * 4));  // a table from the host: no cached templates
  return VSF_OK;
}

int vsf_batch_download(vsf_ctx* c, int32_t first, int32_t count, uint64_t* ids, double* xy, uint32_t* age,
                       int32_t* n, vsf_track_stats* stats) {
  if (!c || first < 0 || count < 0 || first + count > c->B || !ids || !xy || !age || !n) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  HIPCHK(c, hipDeviceSynchronize());
  const size_t o = (size_t)first * c->cap, T = (size_t)count * c->cap;
  HIPCHK(c, hipMemcpy(ids, c->d_ids + o, T * 8, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(xy, c->d_xy + o, T * 16, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(age, c->d_age + o, T * 4, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(n, c->d_n + first, (size_t)count * 4, hipMemcpyDeviceToHost));
  if (stats) HIPCHK(c, hipMemcpy(stats, c->d_stats + (size_t)first * 4, (size_t)count * 16, hipMemcpyDeviceToHost));
  return VSF_OK;
}

int vsf_batch_device_tables(vsf_ctx* c, void** ids, void** xy, void** age, void** n, int32_t* capacity) {
  if (!c) return VSF_E_INVALID;
  if (ids) *ids = c->d_ids;
  if (xy) *xy = c->d_xy;
  if (age) *age = c->d_age;
  if (n) *n = c->d_n;
  if (capacity) *capacity = c->cap;
  return VSF_OK;
}

int vsf_batch_export_tracks(vsf_ctx* c, void* d_records, void* hip_stream) {
  if (!c || !d_records) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  const long long total = (long long)c->B * c->cap;
  hipLaunchKernelGGL(export_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(hip_stream), c->d_ids, c->d_xy, c->d_age, c->d_n, c->cap, total,
                     reinterpret_cast<TrackRecord*>(d_records));
  HIPCHK(c, hipGetLastError());
  return VSF_OK;
}

int vsf_batch_status(vsf_ctx* c, int32_t* status) {
  if (!c || !status) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(status, c->d_status, (size_t)c->B * 4, hipMemcpyDeviceToHost));
  return VSF_OK;
}

// ------------------------------------------------------------------------------------------------
// building blocks for parity tests
// ------------------------------------------------------------------------------------------------
int vsf_debug_pyramid(vsf_ctx* c, const uint8_t* img, int32_t w, int32_t h, int32_t stride, uint8_t** out_levels,
                      int32_t* max_level_out) {
  if (!c || !img || !out_levels || !max_level_out) return VSF_E_INVALID;
  int rc = check_geom(c, w, h, stride);
  if (rc) return rc;
  VSF_ON_DEVICE(c);
  StageSync stage_sync;
  rc = ensure_pair(c);
  if (!rc) rc = upload_frame(c, c->pair_frame[0], img, w, h, stride);
  if (!rc) rc = build_pyramid(c, own_ref(c, c->pair_frame[0]), c->pair_pyr[0], 1, nullptr);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(nullptr));
  for (int l = 1; l < c->nlevels; l++)
    if (out_levels[l])
      HIPCHK(c, hipMemcpy2D(out_levels[l], (size_t)c->lw[l], c->pair_pyr[0].lv[l], (size_t)c->lp[l], (size_t)c->lw[l],
                            (size_t)c->lh[l], hipMemcpyDeviceToHost));
  *max_level_out = c->nlevels - 1;
  return VSF_OK;
}

int vsf_debug_lk(vsf_ctx* c, const uint8_t* prev, const uint8_t* next, int32_t w, int32_t h, const float* prev_pts,
                 float* next_pts, uint8_t* status, int32_t n) {
  if (!c || !prev || !next || n < 0) return VSF_E_INVALID;
  if (c->p.tracker_kind == VSF_TRACKER_LSSD)  // such a context is not held to the LK window / frame size rule
    return fail(c, VSF_E_INVALID, "vsf_debug_lk: not an LK context");
  if (n == 0) return VSF_OK;
  if (!prev_pts || !next_pts || !status) return VSF_E_INVALID;
  int rc = check_geom(c, w, h, w);
  if (rc) return rc;
  VSF_ON_DEVICE(c);
  StageSync stage_sync;  // its destructor covers every return below
  rc = ensure_pair(c);
  if (!rc) rc = upload_frame(c, c->pair_frame[0], prev, w, h, w);
  if (!rc) rc = upload_frame(c, c->pair_frame[1], next, w, h, w, 1);
  if (rc) return rc;
  float2 *d_in = nullptr, *d_out = nullptr;
  uint8_t* d_st = nullptr;
  rc = dalloc(c, &d_in, (size_t)n);
  if (!rc) rc = dalloc(c, &d_out, (size_t)n);
  if (!rc) rc = dalloc(c, &d_st, (size_t)n);
  if (!rc && hipMemcpy(d_in, prev_pts, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) rc = VSF_E_HIP;
  const FrameRef fp = own_ref(c, c->pair_frame[0]), fc = own_ref(c, c->pair_frame[1]);
  if (!rc) rc = build_pyramid(c, fp, c->pair_pyr[0], 1, nullptr);
  if (!rc) rc = build_pyramid(c, fc, c->pair_pyr[1], 1, nullptr);
  if (!rc) {
    const PyrView P = make_view(c, fp, c->pair_pyr[0]), C = make_view(c, fc, c->pair_pyr[1]);
    {
#define VSF_LAUNCH_WP(SEG)                                                                                      \
  hipLaunchKernelGGL(lk_points_kernel<SEG>, dim3(n), dim3(64), c->lks.total, nullptr, P, C, d_in, n, c->lk, d_out, \
                     d_st)
      switch (lk_seg_template(c->lk.seglen)) {
        case 7: VSF_LAUNCH_WP(7); break;
        case 8: VSF_LAUNCH_WP(8); break;
        default: VSF_LAUNCH_WP(16); break;
      }
#undef VSF_LAUNCH_WP
    }
    if (hipGetLastError() != hipSuccess) rc = VSF_E_HIP;
  }
  if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) rc = VSF_E_HIP;
  if (!rc && hipMemcpy(next_pts, d_out, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = VSF_E_HIP;
  if (!rc && hipMemcpy(status, d_st, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = VSF_E_HIP;
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (d_st) (void)hipFree(d_st);
  if (rc == VSF_E_HIP) return fail(c, rc, "vsf_debug_lk: HIP failure");
  return rc;
}

int vsf_debug_set_cand_cap(vsf_ctx* c, int32_t slots) {
  if (!c || slots < 16) return VSF_E_INVALID;
  c->cand_cap = slots < c->cand_cap_alloc ? slots : c->cand_cap_alloc;
  // (the ANMS detector's whole-frame list only ever shrinks: its buffer was sized at creation)
  if (c->p.detector_kind == VSF_DETECTOR_ANMS && slots < c->anms.cand_cap) c->anms.cand_cap = slots;
  return VSF_OK;
}

int vsf_debug_min_eig(vsf_ctx* c, const uint8_t* img, int32_t w, int32_t h, int32_t stride, int32_t cell,
                      float* out, int32_t* rect) {
  if (!c || !img || !out || cell < 0 || cell >= c->ncell) return VSF_E_INVALID;
  int rc = check_geom(c, w, h, stride);
  if (rc) return rc;
  VSF_ON_DEVICE(c);
  StageSync stage_sync;
  rc = upload_frame(c, c->det_frame, img, w, h, stride);
  if (rc) return rc;
  const GridView G = grid_view(c);
  const int tiles_x = sc_tiles(c->max_cw, SC_IW), tiles_y = sc_tiles(c->max_ch, SC_IH);
  const FrameRef f = own_ref(c, c->det_frame);
  // the map of this one cell through the score kernels' own tile routine; the feature table is not touched
  hipLaunchKernelGGL(lam_fill_kernel<3>, dim3(c->ncell, 1), dim3(SC_NT), 0, nullptr, f.ptr, f.sstride, f.pitch, w, h,
                     (const uint8_t*)nullptr, 0ll, 0, G, c->d_nact, tiles_x, tiles_y, c->k_side, c->k_mid, c->d_lam,
                     c->d_cellmax, c->d_candcnt, c->cand_cap, cell, 0.0);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(nullptr));
  std::vector<int4> rects(c->ncell);
  std::vector<int> off(c->ncell + 1);
  HIPCHK(c, hipMemcpy(rects.data(), c->d_rects, rects.size() * sizeof(int4), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(off.data(), c->d_lam_off, off.size() * 4, hipMemcpyDeviceToHost));
  const int4 r = rects[cell];
  HIPCHK(c, hipMemcpy(out, c->d_lam + off[cell], (size_t)r.z * r.w * 4, hipMemcpyDeviceToHost));
  if (rect) {
    rect[0] = r.x;
    rect[1] = r.y;
    rect[2] = r.z;
    rect[3] = r.w;
  }
  return VSF_OK;
}

// ------------------------------------------------------------------------------------------------
// One entry point that allocates nothing (callers under HIP graph capture, tight serving loops, harnesses
// that forbid hipMalloc): cv::calcOpticalFlowPyrLK on device-resident frames and points, like vsf_debug_lk,
// with the two pyramids (levels >= 1) in a file-scope device array instead of hipMalloc'ed buffers.
// Every call on a GPU shares that GPU's array: calls that overlap on one GPU race.
// ------------------------------------------------------------------------------------------------
}  // extern "C"

#define VSF_NOALLOC_SCRATCH_BYTES (4u << 20)
static __device__ __align__(256) uint8_t vsf_noalloc_scratch[VSF_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int NOALLOC_MAX_DEV = 64;
uint8_t* g_noalloc_ptr[NOALLOC_MAX_DEV];  // device address of the array on each GPU (each holds its own copy)
std::mutex g_noalloc_mu;

// `vsf_noalloc_scratch` in host code is a host-side stand-in, not a device address: the address comes from
// hipGetSymbolAddress, once per device, with that device current.
int noalloc_scratch_ptr(int dev, uint8_t** out) {
  if (dev < 0 || dev >= NOALLOC_MAX_DEV) return VSF_E_INVALID;
  std::lock_guard<std::mutex> lock(g_noalloc_mu);
  if (!g_noalloc_ptr[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(vsf_noalloc_scratch)) != hipSuccess || !p) return VSF_E_HIP;
    g_noalloc_ptr[dev] = static_cast<uint8_t*>(p);
  }
  *out = g_noalloc_ptr[dev];
  return VSF_OK;
}
}  // namespace

extern "C" {

int vsf_lk_points_noalloc(const vsf_params* p, const uint8_t* d_prev, int32_t prev_stride, const uint8_t* d_next,
                          int32_t next_stride, const float* d_prev_pts, float* d_next_pts, uint8_t* d_status,
                          int32_t n, void* hip_stream) {
  if (!p || !d_prev || !d_next || n < 0) return VSF_E_INVALID;
  if (n == 0) return VSF_OK;
  if (!d_prev_pts || !d_next_pts || !d_status) return VSF_E_INVALID;
  if (p->width <= 0 || p->height <= 0 || prev_stride < p->width || next_stride < p->width) return VSF_E_INVALID;
  if (p->klt_win < 3 || p->klt_win > MAX_WIN || p->klt_max_level < 0 || p->klt_max_level >= MAXL ||
      p->width <= p->klt_win || p->height <= p->klt_win)
    return VSF_E_INVALID;
  int lw[MAXL], lh[MAXL], lp[MAXL];
  const int nlevels = lk_pyramid_geometry(p->width, p->height, p->klt_win, p->klt_max_level, lw, lh, lp);
  // carve the two pyramids out of the array: level l of pyramid k, 256-byte aligned
  size_t off = 0;
  size_t lv_off[2][MAXL] = {};
  for (int k = 0; k < 2; k++)
    for (int l = 1; l < nlevels; l++) {
      lv_off[k][l] = off;
      off += ((size_t)lp[l] * lh[l] + 16 + 255) & ~(size_t)255;
    }
  if (off > VSF_NOALLOC_SCRATCH_BYTES) return VSF_E_CAPACITY;  // frame too large for the 4 MiB array (about 5.9 Mpixel)
  DeviceGuard dev_guard_(p->device_id);  // the caller's current device is restored on return
  if (dev_guard_.err != hipSuccess) return VSF_E_HIP;
  (void)hipGetLastError();  // an error left by earlier work of the caller is not this call's
  uint8_t* base = nullptr;
  int rc = noalloc_scratch_ptr(p->device_id, &base);
  if (rc) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  uint8_t* lv[2][MAXL] = {};
  for (int k = 0; k < 2; k++)
    for (int l = 1; l < nlevels; l++) lv[k][l] = base + lv_off[k][l];
  const FrameRef fr[2] = {{d_prev, (long long)prev_stride * p->height, prev_stride},
                          {d_next, (long long)next_stride * p->height, next_stride}};
  PyrView V[2];
  for (int k = 0; k < 2; k++) {
    launch_pyramid(fr[k], lv[k], nlevels, lw, lh, lp, 1, s);
    memset(&V[k], 0, sizeof(PyrView));
    V[k].nlevels = nlevels;
    for (int l = 0; l < nlevels; l++) {
      V[k].lv[l].w = lw[l];
      V[k].lv[l].h = lh[l];
      V[k].lv[l].base = l ? lv[k][l] : fr[k].ptr;
      V[k].lv[l].sstride = l ? (long long)lp[l] * lh[l] : fr[k].sstride;
      V[k].lv[l].pitch = l ? lp[l] : fr[k].pitch;
      V[k].lv[l].aligned = ((uintptr_t)V[k].lv[l].base % 4 == 0) && V[k].lv[l].sstride % 4 == 0 && V[k].lv[l].pitch % 4 == 0;
    }
  }
  const LkParams lk = make_lk_params(p);
  const LkSmem lks = lk_smem_layout(lk.win, lk.jmargin);
#define VSF_LAUNCH_NP(SEG)                                                                                   \
  hipLaunchKernelGGL(lk_points_kernel<SEG>, dim3(n), dim3(64), lks.total, s, V[0], V[1],                      \
                     reinterpret_cast<const float2*>(d_prev_pts), n, lk, reinterpret_cast<float2*>(d_next_pts), \
                     d_status)
  switch (lk_seg_template(lk.seglen)) {
    case 7: VSF_LAUNCH_NP(7); break;
    case 8: VSF_LAUNCH_NP(8); break;
    default: VSF_LAUNCH_NP(16); break;
  }
#undef VSF_LAUNCH_NP
  return hipGetLastError() == hipSuccess ? VSF_OK : VSF_E_HIP;
}

// ------------------------------------------------------------------------------------------------
// profiling
// ------------------------------------------------------------------------------------------------
int vsf_batch_frontend_post_track(vsf_ctx* c, const vsf_camera_ds* cam, const vsf_keyframe_params* kp,
                                  const double* dt_since_keyframe, double* d_bearings, int32_t* is_keyframe_out,
                                  uint32_t* aged_out, void* hip_stream) {
  if (!c || !cam || !kp || !dt_since_keyframe || !d_bearings) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  const int B = c->B;
  if (!c->d_fe_dt) {
    int rc = dalloc(c, &c->d_fe_dt, (size_t)B);
    if (!rc) rc = dalloc(c, &c->d_fe_kf, (size_t)B);
    if (!rc) rc = dalloc(c, &c->d_fe_aged, (size_t)B);
    if (rc) return rc;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_fe_dt, dt_since_keyframe, (size_t)B * 8, hipMemcpyHostToDevice, s));
  CameraDS cd = {cam->fx, cam->fy, cam->cx, cam->cy, cam->xi, cam->alpha};
  KeyframeParams k = {kp->keyframe_interval, kp->minimum_keyframe_interval, kp->low_keyframe_feature_number,
                      kp->counting_feature_age};
  hipLaunchKernelGGL(frontend_kernel, dim3(B), dim3(256), 0, s, c->d_xy, c->d_age, c->d_n, c->cap, cd, k, c->d_fe_dt,
                     d_bearings, c->d_fe_kf, c->d_fe_aged);
  HIPCHK(c, hipGetLastError());
  if (is_keyframe_out) HIPCHK(c, hipMemcpyAsync(is_keyframe_out, c->d_fe_kf, (size_t)B * 4, hipMemcpyDeviceToHost, s));
  if (aged_out) HIPCHK(c, hipMemcpyAsync(aged_out, c->d_fe_aged, (size_t)B * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  return VSF_OK;
}

int vsf_unproject(vsf_ctx* c, const vsf_camera_ds* cam, const double* xy, int32_t n, double* bearings,
                  uint8_t* valid) {
  if (!c || !cam || n < 0 || (n > 0 && (!xy || !bearings))) return VSF_E_INVALID;
  if (n == 0) return VSF_OK;
  VSF_ON_DEVICE(c);
  double *d_in = nullptr, *d_out = nullptr;
  uint8_t* d_ok = nullptr;
  int rc = dalloc(c, &d_in, (size_t)n * 2);
  if (!rc) rc = dalloc(c, &d_out, (size_t)n * 3);
  if (!rc) rc = dalloc(c, &d_ok, (size_t)n);
  if (!rc && hipMemcpy(d_in, xy, (size_t)n * 16, hipMemcpyHostToDevice) != hipSuccess) rc = VSF_E_HIP;
  if (!rc) {
    CameraDS cd = {cam->fx, cam->fy, cam->cx, cam->cy, cam->xi, cam->alpha};
    hipLaunchKernelGGL(unproject_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr,
                       reinterpret_cast<const double2*>(d_in), n, cd, d_out, d_ok);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) rc = VSF_E_HIP;
  }
  if (!rc && hipMemcpy(bearings, d_out, (size_t)n * 24, hipMemcpyDeviceToHost) != hipSuccess) rc = VSF_E_HIP;
  if (!rc && valid && hipMemcpy(valid, d_ok, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = VSF_E_HIP;
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (d_ok) (void)hipFree(d_ok);
  if (rc == VSF_E_HIP) return fail(c, rc, "vsf_unproject: HIP failure");
  return rc;
}

namespace {
int lssd_ensure_pair(vsf_ctx* c) {
  int rc = ensure_pair(c);
  for (int i = 0; i < 2 && !rc; i++)
    for (int l = 1; l <= c->lssd.levels && !rc; l++)
      if (!c->lssd_pair_pyr[i][l]) rc = dalloc(c, &c->lssd_pair_pyr[i][l], (size_t)c->l16w[l] * c->l16h[l] + 8);
  return rc;
}
}  // namespace

int vsf_debug_lssd(vsf_ctx* c, const uint8_t* prev, const uint8_t* next, int32_t w, int32_t h, const float* prev_pts,
                   float* next_pts, uint8_t* valid, int32_t n) {
  if (!c || !prev || !next || n < 0 || (n > 0 && (!prev_pts || !next_pts || !valid))) return VSF_E_INVALID;
  if (c->p.tracker_kind != VSF_TRACKER_LSSD) return fail(c, VSF_E_INVALID, "not an LSSD context");
  int rc = check_geom(c, w, h, w);
  if (rc) return rc;
  if (n == 0) return VSF_OK;
  VSF_ON_DEVICE(c);
  StageSync stage_sync;
  rc = lssd_ensure_pair(c);
  if (!rc) rc = upload_frame(c, c->pair_frame[0], prev, w, h, w);
  if (!rc) rc = upload_frame(c, c->pair_frame[1], next, w, h, w, 1);
  if (rc) return rc;
  float2 *d_in = nullptr, *d_out = nullptr;
  uint8_t* d_ok = nullptr;
  rc = dalloc(c, &d_in, (size_t)n);
  if (!rc) rc = dalloc(c, &d_out, (size_t)n);
  if (!rc) rc = dalloc(c, &d_ok, (size_t)n);
  if (!rc && hipMemcpy(d_in, prev_pts, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) rc = VSF_E_HIP;
  const FrameRef fp = own_ref(c, c->pair_frame[0]), fc = own_ref(c, c->pair_frame[1]);
  if (!rc) rc = lssd_build_pyramid(c, fp, c->lssd_pair_pyr[0], 1, nullptr);
  if (!rc) rc = lssd_build_pyramid(c, fc, c->lssd_pair_pyr[1], 1, nullptr);
  if (!rc) {
    const Pyr16View P = make_view16(c, fp, c->lssd_pair_pyr[0]), C = make_view16(c, fc, c->lssd_pair_pyr[1]);
    hipLaunchKernelGGL(lssd_points_kernel, dim3((n + 1) / 2), dim3(64), 0, nullptr, P, C, d_in, n, c->lssd, d_out, d_ok);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) rc = VSF_E_HIP;
  }
  if (!rc && hipMemcpy(next_pts, d_out, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = VSF_E_HIP;
  if (!rc && hipMemcpy(valid, d_ok, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = VSF_E_HIP;
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (d_ok) (void)hipFree(d_ok);
  if (rc == VSF_E_HIP) return fail(c, rc, "vsf_debug_lssd: HIP failure");
  return rc;
}

int vsf_debug_lssd_inverse3(vsf_ctx* c, const float* H, float* Hinv, int32_t n) {
  if (!c || n < 0 || (n > 0 && (!H || !Hinv))) return VSF_E_INVALID;
  if (n == 0) return VSF_OK;
  VSF_ON_DEVICE(c);
  float *d_in = nullptr, *d_out = nullptr;
  int rc = dalloc(c, &d_in, (size_t)n * 9);
  if (!rc) rc = dalloc(c, &d_out, (size_t)n * 9);
  if (!rc && hipMemcpy(d_in, H, (size_t)n * 36, hipMemcpyHostToDevice) != hipSuccess) rc = VSF_E_HIP;
  if (!rc) {
    hipLaunchKernelGGL(lssd_inverse3_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, d_in, d_out, n);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) rc = VSF_E_HIP;
  }
  if (!rc && hipMemcpy(Hinv, d_out, (size_t)n * 36, hipMemcpyDeviceToHost) != hipSuccess) rc = VSF_E_HIP;
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (rc == VSF_E_HIP) return fail(c, rc, "vsf_debug_lssd_inverse3: HIP failure");
  return rc;
}

int vsf_debug_lssd_pyramid(vsf_ctx* c, const uint8_t* img, int32_t w, int32_t h, int32_t stride, int32_t level,
                           uint16_t* out) {
  if (!c || !img || !out) return VSF_E_INVALID;
  if (c->p.tracker_kind != VSF_TRACKER_LSSD) return fail(c, VSF_E_INVALID, "not an LSSD context");
  if (level < 1 || level > c->lssd.levels) return VSF_E_INVALID;
  int rc = check_geom(c, w, h, stride);
  if (rc) return rc;
  VSF_ON_DEVICE(c);
  StageSync stage_sync;
  rc = lssd_ensure_pair(c);
  if (!rc) rc = upload_frame(c, c->pair_frame[0], img, w, h, stride);
  if (!rc) rc = lssd_build_pyramid(c, own_ref(c, c->pair_frame[0]), c->lssd_pair_pyr[0], 1, nullptr);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(nullptr));
  HIPCHK(c, hipMemcpy(out, c->lssd_pair_pyr[0][level], (size_t)c->l16w[level] * c->l16h[level] * 2,
                      hipMemcpyDeviceToHost));
  return VSF_OK;
}

int vsf_debug_anms(vsf_ctx* c, int32_t stream, uint8_t* flag_out, int16_t* kp_out, int32_t* n_kp_out) {
  if (!c || stream < 0 || stream >= c->B) return VSF_E_INVALID;
  if (c->p.detector_kind != VSF_DETECTOR_ANMS) return fail(c, VSF_E_INVALID, "not an ANMS context");
  VSF_ON_DEVICE(c);
  HIPCHK(c, hipDeviceSynchronize());
  const size_t fs = (size_t)c->p.width * c->p.height;
  if (flag_out) HIPCHK(c, hipMemcpy(flag_out, c->d_aflag + (size_t)stream * fs, fs, hipMemcpyDeviceToHost));
  int nd = 0;
  HIPCHK(c, hipMemcpy(&nd, c->d_andet + stream, 4, hipMemcpyDeviceToHost));
  if (kp_out && nd > 0)
    HIPCHK(c, hipMemcpy(kp_out, c->d_akp + (size_t)stream * c->anms.kcap, (size_t)nd * 4, hipMemcpyDeviceToHost));
  if (n_kp_out) *n_kp_out = nd;
  return VSF_OK;
}

#ifdef VSF_LK_STATS
// instrumented build only (not declared in include/vsf.h): reads and optionally clears the LK counters
int vsf_debug_lk_stats(unsigned long long* out /* [512] */, int32_t reset) {
  if (hipDeviceSynchronize() != hipSuccess) return VSF_E_HIP;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lk_stats), sizeof(g_lk_stats)) != hipSuccess) return VSF_E_HIP;
  if (reset) {
    static unsigned long long zeros[512];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lk_stats), zeros, sizeof(zeros)) != hipSuccess) return VSF_E_HIP;
  }
  return VSF_OK;
}
#endif

int vsf_debug_sclk_probe(vsf_ctx* c, int32_t spin_us, double* sclk_mhz, double* realtime_mhz) {
  if (!c || !sclk_mhz || spin_us < 1 || spin_us > 2000000) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  int wall_khz = 0;
  HIPCHK(c, hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, c->p.device_id));
  if (wall_khz <= 0) return fail(c, VSF_E_HIP, "no wall clock rate");
  unsigned long long* d = nullptr;
  int rc = dalloc(c, &d, (size_t)4);
  if (rc) return rc;
  hipStream_t s = nullptr;
  unsigned long long h[4] = {0, 0, 0, 0};
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);  // runs beside whatever the other streams hold
  if (e == hipSuccess) {
    const unsigned long long ticks = (unsigned long long)spin_us * (unsigned long long)wall_khz / 1000ull;
    hipLaunchKernelGGL(sclk_probe_kernel, dim3(1), dim3(64), 0, s, d, ticks);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    (void)hipStreamDestroy(s);
  }
  (void)hipFree(d);
  if (e != hipSuccess) return fail(c, VSF_E_HIP, "sclk probe: %s", hipGetErrorString(e));
  if (h[1] == 0) return fail(c, VSF_E_HIP, "sclk probe: the constant-rate counter did not move");
  *sclk_mhz = (double)h[0] / (double)h[1] * (double)wall_khz / 1000.0;
  if (realtime_mhz) *realtime_mhz = (double)wall_khz / 1000.0;
  return VSF_OK;
}

int vsf_profile_enable(vsf_ctx* c, int32_t on) {
  if (!c) return VSF_E_INVALID;
  c->prof = on == 0 ? 0u : (on == 1 ? (1u << VSF_K_COUNT) - 1u : ((unsigned int)on >> 1) & ((1u << VSF_K_COUNT) - 1u));
  return VSF_OK;
}

int vsf_profile_read(vsf_ctx* c, float* total_ms, int32_t* launches) {
  if (!c || !total_ms || !launches) return VSF_E_INVALID;
  VSF_ON_DEVICE(c);
  HIPCHK(c, hipDeviceSynchronize());
  for (int k = 0; k < VSF_K_COUNT; k++) {
    total_ms[k] = 0.f;
    launches[k] = 0;
  }
  for (auto& e : c->ev_used) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      total_ms[e.kid] += ms;
      launches[e.kid] += 1;
    }
    c->ev_free.push_back(e);
  }
  c->ev_used.clear();
  return VSF_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Native multi-GPU gather: one RCCL all-gather of the fixed-size track records over xGMI.
// The path shards by independent streams, so this is the ONLY collective; a C++ host (the reference's
// example loop, example/vslam/example_serialization.cpp:159-171, one process per GPU) needs nothing but
// include/vsf.h for it.  librccl is opened at run time (dlopen by soname: a process that already
// holds an RCCL -- PyTorch brings its own copy -- gets THAT one, so communicators and streams stay
// compatible), never linked: single-GPU users do not need the library at all.
// ------------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclCommCount) comm_count = nullptr;
  decltype(&ncclCommUserRank) comm_user_rank = nullptr;
  decltype(&ncclGetErrorString) get_error_string = nullptr;
};
std::mutex g_rccl_mu;
RcclApi g_rccl;
char g_comm_err[256] = "";

const RcclApi* rccl_api() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl.handle) return &g_rccl;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    snprintf(g_comm_err, sizeof(g_comm_err), "librccl not found: %s", dlerror());
    return nullptr;
  }
  RcclApi a;
  a.handle = h;
#define VSF_SYM(field, name)                                        \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));    \
  if (!a.field) {                                                   \
    snprintf(g_comm_err, sizeof(g_comm_err), "librccl lacks %s", name); \
    return nullptr;                                                 \
  }
  VSF_SYM(get_unique_id, "ncclGetUniqueId")
  VSF_SYM(comm_init_rank, "ncclCommInitRank")
  VSF_SYM(comm_destroy, "ncclCommDestroy")
  VSF_SYM(all_gather, "ncclAllGather")
  VSF_SYM(comm_count, "ncclCommCount")
  VSF_SYM(comm_user_rank, "ncclCommUserRank")
  VSF_SYM(get_error_string, "ncclGetErrorString")
#undef VSF_SYM
  g_rccl = a;
  return &g_rccl;
}
}  // namespace

struct vsf_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1, device = 0;
  bool owned = false;
};

extern "C" {

const char* vsf_comm_last_error(void) { return g_comm_err; }

int vsf_comm_unique_id(uint8_t* id /* [VSF_COMM_ID_BYTES] */) {
  if (!id) return VSF_E_INVALID;
  static_assert(sizeof(ncclUniqueId) == VSF_COMM_ID_BYTES, "ncclUniqueId size");
  const RcclApi* a = rccl_api();
  if (!a) return VSF_E_HIP;
  ncclUniqueId u;
  const ncclResult_t r = a->get_unique_id(&u);
  if (r != ncclSuccess) {
    snprintf(g_comm_err, sizeof(g_comm_err), "ncclGetUniqueId: %s", a->get_error_string(r));
    return VSF_E_HIP;
  }
  memcpy(id, &u, sizeof(u));
  return VSF_OK;
}

int vsf_comm_create(const uint8_t* id, int32_t rank, int32_t nranks, int32_t device_id, vsf_comm** out) {
  if (!out) return VSF_E_INVALID;
  *out = nullptr;
  if (nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !id)) return VSF_E_INVALID;
  vsf_comm* c = new (std::nothrow) vsf_comm();
  if (!c) return VSF_E_NOMEM;
  c->rank = rank;
  c->nranks = nranks;
  c->device = device_id;
  if (nranks > 1) {  // a world of one needs no communicator (and no librccl)
    const RcclApi* a = rccl_api();
    if (!a) {
      delete c;
      return VSF_E_HIP;
    }
    DeviceGuard dg(device_id);
    if (dg.err != hipSuccess) {
      delete c;
      return VSF_E_HIP;
    }
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    const ncclResult_t r = a->comm_init_rank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) {
      snprintf(g_comm_err, sizeof(g_comm_err), "ncclCommInitRank: %s", a->get_error_string(r));
      delete c;
      return VSF_E_HIP;
    }
    c->owned = true;
  }
  *out = c;
  return VSF_OK;
}

int vsf_comm_wrap(void* nccl_comm, int32_t device_id, vsf_comm** out) {
  if (!out || !nccl_comm) return VSF_E_INVALID;
  *out = nullptr;
  const RcclApi* a = rccl_api();
  if (!a) return VSF_E_HIP;
  vsf_comm* c = new (std::nothrow) vsf_comm();
  if (!c) return VSF_E_NOMEM;
  c->comm = static_cast<ncclComm_t>(nccl_comm);
  c->device = device_id;
  if (a->comm_count(c->comm, &c->nranks) != ncclSuccess || a->comm_user_rank(c->comm, &c->rank) != ncclSuccess) {
    snprintf(g_comm_err, sizeof(g_comm_err), "ncclCommCount / ncclCommUserRank failed on the given communicator");
    delete c;
    return VSF_E_HIP;
  }
  *out = c;
  return VSF_OK;
}

void vsf_comm_destroy(vsf_comm* c) {
  if (!c) return;
  if (c->owned && c->comm) {
    const RcclApi* a = rccl_api();
    DeviceGuard dg(c->device);
    if (a) (void)a->comm_destroy(c->comm);
  }
  delete c;
}

int vsf_comm_rank(const vsf_comm* c, int32_t* rank, int32_t* nranks) {
  if (!c) return VSF_E_INVALID;
  if (rank) *rank = c->rank;
  if (nranks) *nranks = c->nranks;
  return VSF_OK;
}

int vsf_gather_tracks(vsf_ctx* c, vsf_comm* comm, void* d_recv, void* hip_stream) {
  if (!c || !d_recv) return VSF_E_INVALID;
  const int nranks = comm ? comm->nranks : 1, rank = comm ? comm->rank : 0;
  if (comm && comm->device != c->p.device_id) return fail(c, VSF_E_INVALID, "communicator of device %d, context of device %d", comm->device, c->p.device_id);
  VSF_ON_DEVICE(c);
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  const size_t bytes = (size_t)c->B * c->cap * sizeof(TrackRecord);
  // this rank's records straight into its slot of the receive buffer: the all-gather then runs in place
  uint8_t* mine = static_cast<uint8_t*>(d_recv) + (size_t)rank * bytes;
  const long long total = (long long)c->B * c->cap;
  hipLaunchKernelGGL(export_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, c->d_ids, c->d_xy, c->d_age,
                     c->d_n, c->cap, total, reinterpret_cast<TrackRecord*>(mine));
  HIPCHK(c, hipGetLastError());
  if (nranks > 1) {
    const RcclApi* a = rccl_api();
    if (!a || !comm->comm) return fail(c, VSF_E_HIP, "no RCCL communicator: %s", g_comm_err);
    const ncclResult_t r = a->all_gather(mine, d_recv, bytes, ncclUint8, comm->comm, s);
    if (r != ncclSuccess) return fail(c, VSF_E_HIP, "ncclAllGather: %s", a->get_error_string(r));
  }
  return VSF_OK;
}

}  // extern "C"
