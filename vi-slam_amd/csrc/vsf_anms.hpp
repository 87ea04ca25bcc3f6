// vsf_anms.hpp -- FeatureDetectorANMS on the device (ref: src/vslam/feature/FeatureDetectorANMS.cpp:69-212):
//   mask  = (user mask or ones) with a cv::circle disc zeroed at every in-image feature, AND the edge
//           rectangle of ones (:107-132)
//   GFTT  = cv::GFTTDetector(nfeatures, quality 0.001, minDistance 1, blockSize 4) on the whole frame:
//           score_kernel<4> + candidate_kernel with the frame as one cell, then the nfeatures largest
//           keys in cv's greaterThanPtr order (minDistance 1 never rejects distinct pixels)
//   ANMS  = the reference's degenerate cv::sortIdx permutation (Q9) followed by anms::Ssc
//           (NonMaximumSuppression.cpp:43-53, anms/anms.cpp:344-438)
//   ids   = max_feature_index + 1 ... (:198-211), through append_kernel with id_bias = 1.
// Included by vsf_kernels.hpp (inside no namespace).
#pragma once

namespace vsf {

constexpr int ANMS_MAX_RADIUS = 31;

struct AnmsParams {
  int w, h;
  int max_features;  // Parameter::max_feature_number_
  int radius;        // static_cast<int>(min_feature_distance)  (:127)
  int edge;          // detection_ignore_edge_
  int nfeatures;     // detection_feature_number_
  float tolerance;   // non_max_suppression_tolerance_ as float (:181)
  int kcap;          // power of two >= nfeatures (sort / LDS array size)
  int hash_size;     // power of two >= 2 * kcap
  int hash_bits;
  int cand_cap;      // candidate keys kept per stream
  double quality;    // detection_quality_level_
  // detection_min_feature_distance_ > 1 (cv's greedy minDistance rule, anms_topk_kernel<true>)
  int gcap;          // keys of a sorted chunk (power of two >= kcap)
  int grad;          // ceil(minDistance): half side of the square that holds the suppression disc
  double gmd2;       // minDistance squared, as cv squares it (double)
  int halfw[ANMS_MAX_RADIUS + 1];  // cv::Circle half widths: disc = { |dy| <= r, |dx| <= halfw[|dy|] }
};

// (user mask & 1) inside the edge rectangle, 0 outside; one thread per pixel quad of a row.
__global__ __launch_bounds__(256) void anms_mask_kernel(const uint8_t* __restrict__ masks, long long mask_sstride,
                                                        int mask_row_stride, AnmsParams ap,
                                                        uint8_t* __restrict__ flag) {
  const int stream = blockIdx.z, y = blockIdx.y;
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x0 >= ap.w) return;
  const int ew = ap.w - 2 * ap.edge, eh = ap.h - 2 * ap.edge;
  const bool yin = ew > 0 && eh > 0 && y >= ap.edge && y <= ap.edge + eh - 1;
  const uint8_t* M = masks ? masks + (long long)stream * mask_sstride + (long long)y * mask_row_stride : nullptr;
  uint8_t* F = flag + ((long long)stream * ap.h + y) * ap.w;
  uint8_t v[4];
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const int x = x0 + b;
    v[b] = 0;
    if (x < ap.w && yin && x >= ap.edge && x <= ap.edge + ew - 1) v[b] = M ? (uint8_t)(M[x] & 1u) : (uint8_t)1;  // :132 bitwise_and
  }
  if ((ap.w & 3) == 0) {  // rows of the flag image are dword aligned: one store per quad
    *reinterpret_cast<unsigned int*>(F + x0) =
        (unsigned int)v[0] | ((unsigned int)v[1] << 8) | ((unsigned int)v[2] << 16) | ((unsigned int)v[3] << 24);
  } else {
#pragma unroll
    for (int b = 0; b < 4; b++)
      if (x0 + b < ap.w) F[x0 + b] = v[b];
  }
}

// Zeroes the cv::circle disc of every in-image feature (:122-131) and resets the per-stream counters.
// Grid: (chunks of disc rows, streams).
__global__ __launch_bounds__(256) void anms_disc_kernel(const double2* __restrict__ xy, const int* __restrict__ n,
                                                        int cap, AnmsParams ap, uint8_t* __restrict__ flag,
                                                        unsigned int* __restrict__ cell_max,
                                                        int* __restrict__ cand_cnt) {
  const int stream = blockIdx.y;
  const int n0 = n[stream];
  const int rows = 2 * ap.radius + 1;
  uint8_t* F = flag + (long long)stream * ap.h * ap.w;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < n0 * rows; t += gridDim.x * 256) {
    const int f = t / rows, dy = t - f * rows - ap.radius;
    const double2 p = xy[(long long)stream * cap + f];
    if (p.x > 0 && p.x < (double)ap.w && p.y > 0 && p.y < (double)ap.h) {  // :123-124
      const int cx = (int)p.x, cy = (int)p.y;                              // cv::Point(double, double)
      const int y = cy + dy;
      if (y >= 0 && y < ap.h) {
        const int hw = ap.halfw[dy < 0 ? -dy : dy];
        const int x1 = max(cx - hw, 0), x2 = min(cx + hw, ap.w - 1);
        for (int x = x1; x <= x2; x++) F[(long long)y * ap.w + x] = 0;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    cell_max[stream] = 0u;
    cand_cnt[stream] = 0;
  }
}

// Candidate key of raster position idx straight from the lambda map (used when the candidate list
// overflowed): threshold, 3x3 local maximum, interior only, mask -- SURVEY A.1 steps 5-6.
__device__ __forceinline__ unsigned long long anms_key_from_map(const float* __restrict__ L, int w, int h, float thr,
                                                                const uint8_t* __restrict__ M, int idx) {
  const int y = idx / w, x = idx - y * w;
  if (x < 1 || x >= w - 1 || y < 1 || y >= h - 1) return 0ull;
  float v = L[idx];
  v = v > thr ? v : 0.f;
  if (!(v > 0.f)) return 0ull;
  float m = v;
#pragma unroll
  for (int j = -1; j <= 1; j++)
#pragma unroll
    for (int q = -1; q <= 1; q++) {
      float t = L[idx + j * w + q];
      t = t > thr ? t : 0.f;
      m = fmaxf(m, t);
    }
  if (v != m || !M[idx]) return 0ull;
  // (v > 0: with a negative masked maximum -- Harris on an image of edges only -- cv's threshold max * quality,
  // quality in [0, 1], is at or above that maximum, and no response at or below the maximum passes it)
  return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned int)((y << 16) | x);
}

// The nfeatures largest candidate keys of a stream, in descending order (= cv's sorted corner list cut
// at maxCorners).  MSB-first radix select (up to 8 passes of 8 bits) finds the K-th largest key, the keys at
// or above it are gathered into LDS and sorted there (bitonic, descending).  One block per stream, any block
// size that is a multiple of 64 (the host takes 1024 threads when few streams would otherwise leave the one
// block of a stream alone with a few hundred thousand keys: C4 with 32 streams, the single-stream host API).
//   * The first pass also counts the keys.  A pass whose selected bin holds exactly the keys still wanted ends the
//     search: every key with that prefix belongs to the top K (the K-th lambda is nearly always unique, so the
//     four passes over the position half of the key are rarely needed).
//   * Keys are read four per thread and iteration (loads in flight together), and a wave adds the keys that
//     share the bin of its first lane with ONE LDS atomic: the upper bytes of lambda are the same for most keys
//     of a frame, and sixteen waves adding one by one to the same address queue behind each other.
// GREEDY (detection_min_feature_distance_ > 1): cv::goodFeaturesToTrack visits the sorted corners and accepts one
// when no accepted corner lies closer than minDistance (float dx*dx + dy*dy < double minDistance^2), until
// maxCorners are accepted.  Here the sorted list is produced in chunks of gcap keys (the same select + sort, over
// the keys below the previous chunk's smallest), and a chunk is swept a window of blockDim.x keys at a time: a
// thread holds one key of the window and whether it is still alive; the first alive key of the window is accepted
// (one ballot per wave, one LDS atomicMin, ONE barrier per accepted corner), every thread tests its own key against
// it in registers, and the corner's disc is zeroed in the stream's flag image -- dead for the score kernel by now --
// which is what tells later windows and chunks that a key is blocked.  The flag image read back by vsf_debug_anms
// therefore has those discs zeroed too.
template <bool GREEDY>
__global__ __launch_bounds__(1024) void anms_topk_kernel(const float* __restrict__ lam, long long lam_sstride,
                                                         uint8_t* flag, const unsigned int* __restrict__ cell_max,
                                                         const unsigned long long* __restrict__ cand,
                                                         const int* __restrict__ cand_cnt, AnmsParams ap,
                                                         short2* __restrict__ kp, int* __restrict__ ndet) {
  extern __shared__ __align__(16) uint8_t smem[];
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(smem);  // [kcap], GREEDY: [gcap]
  __shared__ unsigned int s_hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_remaining, s_count, s_stop, s_ncand;
  __shared__ int s_first[3];
  const int NT = (int)blockDim.x, tid = (int)threadIdx.x;
  const int stream = blockIdx.x;
  const int total = cand_cnt[stream];
  const unsigned int enc = cell_max[stream];
  const bool overflow = total > ap.cand_cap;
  const int M = overflow ? ap.w * ap.h : total;
  const float* L = lam + (long long)stream * lam_sstride;
  uint8_t* F = flag + (long long)stream * ap.w * ap.h;
  const unsigned long long* C = cand + (long long)stream * ap.cand_cap;
  const float thr = (float)((enc ? (double)float_order_decode(enc) : 0.0) * ap.quality);
  const int want = GREEDY ? ap.gcap : ap.nfeatures;   // keys of one sorted chunk
  const int arr = GREEDY ? ap.gcap : ap.kcap;         // slots of s_key
  short2* out = kp + (long long)stream * ap.kcap;
  unsigned long long U = ~0ull;  // GREEDY: keys at or above U belong to earlier chunks
  int accepted = 0;
  // the list holds every positive local maximum the score kernel could not rule out: cv's threshold is applied here
  auto key_at = [&](int i) -> unsigned long long {
    unsigned long long k;
    if (overflow) {
      k = anms_key_from_map(L, ap.w, ap.h, thr, F, i);  // (reads the flag: blocked positions give no key)
    } else {
      k = C[i];
      if (!(__uint_as_float((unsigned int)(k >> 32)) > thr)) k = 0ull;
    }
    if (GREEDY && U != ~0ull) {
      if (k >= U) k = 0ull;
      if (k != 0ull && !overflow && F[(int)((unsigned int)k >> 16) * ap.w + (int)((unsigned int)k & 0xffffu)] == 0)
        k = 0ull;  // blocked by an accepted corner
    }
    return k;
  };
  // f(key) for every candidate key (0 = none), four independent reads per thread and iteration
  auto for_keys = [&](auto&& f) {
    int i = tid;
    for (; i + 3 * NT < M; i += 4 * NT) {
      const unsigned long long k0 = key_at(i), k1 = key_at(i + NT), k2 = key_at(i + 2 * NT), k3 = key_at(i + 3 * NT);
      f(k0);
      f(k1);
      f(k2);
      f(k3);
    }
    for (; i < M; i += NT) f(key_at(i));
  };
  auto hist_add = [&](unsigned int bin) {  // called by the lanes that hold a key of the current prefix
    const unsigned int b0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)bin);
    if (bin == b0) {
      const unsigned long long m = __ballot(1);
      if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&s_hist[b0], (unsigned int)__popcll(m));
    } else {
      atomicAdd(&s_hist[bin], 1u);
    }
  };

  for (;;) {  // one round per chunk; a single round unless GREEDY
    unsigned long long T = 1ull;  // keys are never 0
    int K = 0;
    {
      __syncthreads();  // (GREEDY: the discs of the chunk before are visible, the shared scalars are free)
      if (tid == 0) {
        s_prefix = 0ull;
        s_stop = 0;
      }
      unsigned long long himask = 0ull;
      for (int b = 7; b >= 0; b--) {
        for (int i = tid; i < 256; i += NT) s_hist[i] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        for_keys([&](unsigned long long k) {
          if (k != 0ull && (k & himask) == prefix) hist_add((unsigned int)(k >> (8 * b)) & 255u);
        });
        __syncthreads();
        if (tid == 0) {
          if (b == 7) {  // the first pass saw every key: the number of candidates
            int nc = 0;
            for (int q = 0; q < 256; q++) nc += (int)s_hist[q];
            s_ncand = nc;
            s_remaining = min(nc, want);
            if (nc <= want) s_stop = 1;  // every key is kept: T stays 1
          }
          if (!s_stop) {
            int rem = s_remaining, bin = 255;
            for (; bin > 0; bin--) {
              const int hcnt = (int)s_hist[bin];
              if (hcnt >= rem) break;
              rem -= hcnt;
            }
            s_prefix = prefix | ((unsigned long long)bin << (8 * b));
            s_remaining = rem;
            // the bin holds exactly the keys still wanted: all keys at or above the prefix are the top K
            if ((int)s_hist[bin] == rem) s_stop = 2;
          }
        }
        himask |= 0xffull << (8 * b);
        __syncthreads();
        if (s_stop) break;
      }
      K = min(s_ncand, want);
      if (s_stop != 1) T = s_prefix;
    }
    const bool last_chunk = s_ncand <= want;
    // gather and sort
    __syncthreads();
    for (int i = tid; i < arr; i += NT) s_key[i] = 0ull;
    if (tid == 0) s_count = 0;
    __syncthreads();
    for_keys([&](unsigned long long k) {
      if (k != 0ull && k >= T) {
        const int slot = atomicAdd(&s_count, 1);
        if (slot < arr) s_key[slot] = k;
      }
    });
    __syncthreads();
    // the slots beyond the gathered keys hold 0 and sort to the end: sort the smallest power of two that holds them
    int n2 = 1;
    while (n2 < K) n2 <<= 1;
    for (int size = 2; size <= n2; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = tid; i < n2 / 2; i += NT) {
          const int lo = ((i / stride) * stride * 2) + (i % stride), hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const unsigned long long a = s_key[lo], b2 = s_key[hi];
          if (desc ? (a < b2) : (a > b2)) {
            s_key[lo] = b2;
            s_key[hi] = a;
          }
        }
        __syncthreads();
      }
    if (!GREEDY) {
      for (int i = tid; i < K; i += NT) {
        const unsigned int yx = (unsigned int)s_key[i];  // y << 16 | x
        out[i] = make_short2((short)(yx & 0xffffu), (short)(yx >> 16));
      }
      if (tid == 0) ndet[stream] = K;
      return;
    }
    // ---- GREEDY sweep of the sorted chunk
    const int lane = tid & 63;
    const int side = 2 * ap.grad + 1;
    if (tid == 0) s_first[0] = s_first[1] = s_first[2] = 0x7fffffff;
    int it = 0;  // s_first[it % 3] belongs to iteration `it`; the slot two iterations ahead is reset meanwhile
    for (int cursor = 0; cursor < K && accepted < ap.nfeatures; cursor += NT) {
      __syncthreads();  // the discs painted so far are visible (and s_first's initial values)
      const int i = cursor + tid;
      bool alive = false;
      int mx = 0, my = 0;
      if (i < K) {
        const unsigned int yx = (unsigned int)s_key[i];
        my = (int)(yx >> 16);
        mx = (int)(yx & 0xffffu);
        alive = F[my * ap.w + mx] != 0;
      }
      while (accepted < ap.nfeatures) {
        const int slot = it % 3;
        const unsigned long long bal = __ballot(alive);
        if (bal != 0ull && lane == 0) atomicMin(&s_first[slot], cursor + (tid & ~63) + __ffsll((long long)bal) - 1);
        __syncthreads();
        const int first = s_first[slot];
        if (tid == 0) s_first[(it + 2) % 3] = 0x7fffffff;
        it++;
        if (first == 0x7fffffff) break;  // the window holds no alive key
        const unsigned int pyx = (unsigned int)s_key[first];
        const int py = (int)(pyx >> 16), px = (int)(pyx & 0xffffu);
        if (tid == 0) out[accepted] = make_short2((short)px, (short)py);
        accepted++;
        const float ddx = (float)mx - (float)px, ddy = (float)my - (float)py;
        if ((double)(ddx * ddx + ddy * ddy) < ap.gmd2) alive = false;  // (the accepted key itself: distance 0)
        for (int t = tid; t < side * side; t += NT) {
          const int dy = t / side - ap.grad, dx = t - (t / side) * side - ap.grad;
          const int x = px + dx, y = py + dy;
          const float fx = (float)dx, fy = (float)dy;
          if (x >= 0 && x < ap.w && y >= 0 && y < ap.h && (double)(fx * fx + fy * fy) < ap.gmd2) F[y * ap.w + x] = 0;
        }
      }
    }
    if (accepted >= ap.nfeatures || last_chunk) break;
    U = s_key[K - 1];
  }
  if (tid == 0) ndet[stream] = accepted;
}

// Position, before libstdc++'s std::sort, of the element std::sort leaves at position p when every
// comparison is false (all keys equal, Q9).  introsort: while a segment is longer than 16, swap its
// first and middle element, then the unguarded partition reverses [first + 1, last) and returns its
// middle; recurse right, loop left; the final insertion sort moves nothing.  The segment tree depends
// on n only, so the source is found by walking the path of p and undoing the transforms leaf to root.
__device__ __forceinline__ int introsort_equal_src(int n, int p) {
  int segf[32], segl[32], d = 0;
  int first = 0, last = n;
  while (last - first > 16) {
    segf[d] = first;
    segl[d] = last;
    d++;
    const int cut = first + 1 + (last - first - 1) / 2;
    if (p >= cut)
      first = cut;
    else
      last = cut;
  }
  int q = p;
  for (int k = d - 1; k >= 0; k--) {
    const int f = segf[k], l = segl[k], mid = f + (l - f) / 2;
    if (q == f) {
      q = mid;
    } else {
      const int x = f + l - q;
      q = (x == mid) ? f : x;
    }
  }
  return q;
}

// suppressNonMax + anms::Ssc + the need / detected branches of FeatureDetectorANMS::Detect (:174-186).
// One block per stream (any multiple of 64 threads up to 1024).  The sweep "accept a keypoint iff its cell is uncovered, then cover the cells
// within +-2" is sequential in the reference; here only the FIRST keypoint of a cell can ever be
// accepted, and among those owners the accepted set is the greedy independent set in index order for
// the relation |d_row| <= 2 && |d_col| <= 2, which is reached by rounds: an owner turns REJECTED when an
// earlier conflicting owner is ACCEPTED, ACCEPTED when all earlier conflicting owners are REJECTED.
// Cells: row = floor(y / (w/2)) = (2 y) / w in integers (y, w integers, w >= 1), likewise col.
__global__ __launch_bounds__(1024) void anms_ssc_kernel(const short2* __restrict__ kp, const int* __restrict__ ndet,
                                                       const int* __restrict__ n_tab, AnmsParams ap,
                                                       double2* __restrict__ sel_xy, int* __restrict__ sel_cnt) {
  // LDS, 17 bytes per keypoint slot (139 KB at 8192): positions, the cell of every keypoint at the current width,
  // a hash of occupied cells whose entries hold only the index of the cell's first keypoint (its key is read back
  // from that keypoint's cell), the per-keypoint state.  The result list of a width shares the hash's space: the
  // hash is dead once the rounds are over, and a later width rebuilds it only when its own result replaces the list.
  extern __shared__ __align__(16) uint8_t smem[];
  unsigned int* cell = reinterpret_cast<unsigned int*>(smem);                    // [kcap] row << 16 | col
  unsigned int* hown = cell + ap.kcap;                                           // [hash_size] owner index or EMPTY
  unsigned short* res = reinterpret_cast<unsigned short*>(hown);                 // [kcap], after the rounds
  short* sx = reinterpret_cast<short*>(hown + ap.hash_size);
  short* sy = sx + ap.kcap;
  volatile uint8_t* state = reinterpret_cast<volatile uint8_t*>(sy + ap.kcap);
  constexpr unsigned int EMPTY = 0xffffffffu;
  __shared__ int s_low, s_high, s_width, s_prev, s_done, s_nres, s_base;
  __shared__ unsigned int s_kmin, s_kmax;
  __shared__ int s_wave[16];
  const int NT = (int)blockDim.x, NW = NT >> 6;
  const int stream = blockIdx.x;
  const int n = ndet[stream];
  int need = ap.max_features - n_tab[stream];  // :174-175
  need = need > 0 ? need : 0;
  const short2* K = kp + (long long)stream * ap.kcap;
  double2* out = sel_xy + (long long)stream * ap.kcap;
  if (need >= n) {  // :184-186 every detected keypoint, in GFTT order
    for (int i = threadIdx.x; i < n; i += NT) out[i] = make_double2((double)K[i].x, (double)K[i].y);
    if (threadIdx.x == 0) sel_cnt[stream] = n;
    return;
  }
  if (need == 0) {  // the reference calls Ssc(numRetPoints = 0): undefined there; nothing is added here
    if (threadIdx.x == 0) sel_cnt[stream] = 0;
    return;
  }
  // NonMaximumSuppression.cpp:43-53: responses truncate to int 0, cv::sortIdx DESCENDING = reversed std::sort
  for (int i = threadIdx.x; i < n; i += NT) {
    const short2 p = K[introsort_equal_src(n, n - 1 - i)];
    sx[i] = p.x;
    sy[i] = p.y;
  }
  if (threadIdx.x == 0) {
    const int rows = ap.h, cols = ap.w, num = need;
    const int exp1 = rows + cols + 2 * num;  // anms.cpp:350-366
    const long long exp2 = ((long long)4 * cols + (long long)4 * num + (long long)4 * rows * num +
                            (long long)rows * rows + (long long)cols * cols - (long long)2 * rows * cols +
                            (long long)4 * rows * cols * num);
    const double exp3 = sqrt((double)exp2);
    const double exp4 = num - 1;
    const double sol1 = -round((exp1 + exp3) / exp4);
    const double sol2 = -round((exp1 - exp3) / exp4);
    s_high = (int)((sol1 > sol2) ? sol1 : sol2);
    s_low = (int)floor(sqrt((double)n / num));
    s_prev = -1;
    s_nres = 0;
    s_done = 0;
    const float Kf = (float)(unsigned int)num;  // :374-375 unsigned * float
    s_kmin = (unsigned int)roundf(Kf - (Kf * ap.tolerance));
    s_kmax = (unsigned int)roundf(Kf + (Kf * ap.tolerance));
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (;;) {
    if (threadIdx.x == 0) {
      const int width = s_low + (s_high - s_low) / 2;  // :380
      if (width == s_prev || s_low > s_high) s_done = 1;  // :381-386: keep the previous result
      s_width = width;
    }
    __syncthreads();
    if (s_done) break;
    const int width = s_width;
    // The widths the search visits are mostly large: the whole cell grid then fits the table and a cell's slot is
    // its raster index (no probing, no key compare: the neighbourhood test of the rounds is 24 plain reads);
    // small widths hash the occupied cells.
    const unsigned int nrow = (unsigned int)(2 * (ap.h - 1)) / (unsigned int)width + 1u;
    const unsigned int ncol = (unsigned int)(2 * (ap.w - 1)) / (unsigned int)width + 1u;
    const bool dense = (unsigned long long)nrow * ncol <= (unsigned long long)ap.hash_size;
    const int nslot = dense ? (int)(nrow * ncol) : ap.hash_size;
    for (int i = threadIdx.x; i < nslot; i += NT) hown[i] = EMPTY;
    for (int i = threadIdx.x; i < n; i += NT) {
      const unsigned int row = (unsigned int)(2 * (int)sy[i]) / (unsigned int)width;
      const unsigned int col = (unsigned int)(2 * (int)sx[i]) / (unsigned int)width;
      cell[i] = (row << 16) | col;
    }
    __syncthreads();
    // every owner a slot ever has lies in the same cell, so the slot's key can be read through any of them
    auto find_or_claim = [&](unsigned int key, unsigned int i) -> unsigned int {
      unsigned int h = (key * 2654435761u) >> (32 - ap.hash_bits);
      for (;;) {
        unsigned int e = hown[h];
        if (e == EMPTY) {
          e = atomicCAS(&hown[h], EMPTY, i);
          if (e == EMPTY) return h;
        }
        if (cell[e] == key) {
          atomicMin(&hown[h], i);
          return h;
        }
        h = (h + 1) & (unsigned int)(ap.hash_size - 1);
      }
    };
    auto owner_of = [&](unsigned int key) -> unsigned int {
      if (dense) {
        const unsigned int row = key >> 16, col = key & 0xffffu;
        return (row < nrow && col < ncol) ? hown[row * ncol + col] : EMPTY;
      }
      unsigned int h = (key * 2654435761u) >> (32 - ap.hash_bits);
      for (;;) {
        const unsigned int e = hown[h];
        if (e == EMPTY) return EMPTY;
        if (cell[e] == key) return e;
        h = (h + 1) & (unsigned int)(ap.hash_size - 1);
      }
    };
    if (dense) {
      for (int i = threadIdx.x; i < n; i += NT) atomicMin(&hown[(cell[i] >> 16) * ncol + (cell[i] & 0xffffu)], (unsigned int)i);
    } else {
      for (int i = threadIdx.x; i < n; i += NT) (void)find_or_claim(cell[i], (unsigned int)i);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += NT) state[i] = (owner_of(cell[i]) == (unsigned int)i) ? 0 : 2;  // 0 U, 1 A, 2 R
    __syncthreads();
    for (;;) {
      int undecided = 0;
      if (dense) {
        // the 24 neighbour slots are plain addresses: read all owners, then all their states, then decide (one
        // probe after the other -- owner, state, branch -- an undecided owner waited for 48 dependent LDS reads a
        // round, and with a fine grid nearly every keypoint owns its cell).  A state read here is at least as new
        // as the last barrier, which is all the rounds need.
        const uint8_t* st_r = const_cast<const uint8_t*>(state);
        for (int i = threadIdx.x; i < n; i += NT) {
          if (st_r[i] != 0) continue;
          const int row = (int)(cell[i] >> 16), col = (int)(cell[i] & 0xffffu);
          unsigned int own[24];
          {
            int q = 0;
#pragma unroll
            for (int dr = -2; dr <= 2; dr++)
#pragma unroll
              for (int dc = -2; dc <= 2; dc++) {
                if (dr == 0 && dc == 0) continue;
                const int rr = row + dr, cc = col + dc;
                const bool in = (unsigned int)rr < nrow && (unsigned int)cc < ncol;
                const unsigned int o = hown[in ? rr * (int)ncol + cc : 0];
                own[q++] = in ? o : EMPTY;
              }
          }
          bool rejected = false, blocked = false;
#pragma unroll
          for (int q = 0; q < 24; q++) {
            const bool earlier = own[q] < (unsigned int)i;
            const uint8_t st = st_r[earlier ? own[q] : (unsigned int)i];
            rejected |= earlier && st == 1;
            blocked |= earlier && st == 0;
          }
          if (rejected)
            state[i] = 2;
          else if (!blocked)
            state[i] = 1;
          else
            undecided = 1;
        }
      } else {
        for (int i = threadIdx.x; i < n; i += NT) {
          if (state[i] != 0) continue;
          const int row = (int)(cell[i] >> 16), col = (int)(cell[i] & 0xffffu);
          bool rejected = false, blocked = false;
          for (int dr = -2; dr <= 2 && !rejected; dr++)
            for (int dc = -2; dc <= 2; dc++) {
              if (dr == 0 && dc == 0) continue;
              const int rr = row + dr, cc = col + dc;
              if (rr < 0 || cc < 0) continue;
              const unsigned int j = owner_of(((unsigned int)rr << 16) | (unsigned int)cc);
              if (j < (unsigned int)i) {
                const uint8_t st = state[j];
                if (st == 1) {
                  rejected = true;
                  break;
                }
                if (st == 0) blocked = true;
              }
            }
          if (rejected)
            state[i] = 2;
          else if (!blocked)
            state[i] = 1;
          else
            undecided = 1;
        }
      }
      if (!__syncthreads_or(undecided)) break;
    }
    // result = accepted keypoints in sweep order
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += NT) {
      const int i = c0 + threadIdx.x;
      const bool acc = (i < n) && state[i] == 1;
      const unsigned long long bal = __ballot(acc);
      if (lane == 0) s_wave[wv] = __popcll(bal);
      __syncthreads();
      int off = s_base;
      for (int q = 0; q < wv; q++) off += s_wave[q];
      off += __popcll(bal & ((1ull << lane) - 1ull));
      if (acc) res[off] = (unsigned short)i;
      __syncthreads();
      if (threadIdx.x == 0) {
        int t = 0;
        for (int q = 0; q < NW; q++) t += s_wave[q];
        s_base += t;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const unsigned int nres = (unsigned int)s_base;
      s_nres = s_base;
      if (nres >= s_kmin && nres <= s_kmax)  // :423-430
        s_done = 1;
      else if (nres < s_kmin)
        s_high = width - 1;
      else
        s_low = width + 1;
      s_prev = width;
    }
    __syncthreads();
    if (s_done) break;
  }
  const int nres = s_nres;
  for (int k = threadIdx.x; k < nres; k += NT) out[k] = make_double2((double)sx[res[k]], (double)sy[res[k]]);
  if (threadIdx.x == 0) sel_cnt[stream] = nres;
}

}  // namespace vsf
