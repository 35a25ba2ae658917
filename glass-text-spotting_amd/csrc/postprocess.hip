// Word post-processing on device: PostProcessorAcademic for every image of a step in one launch.
//
// Behavioural mirror of reference glass/postprocess/post_processor_rotated_boxes.py
// (`PostProcessorRotatedBoxes.__call__` :66-87: filter_small_boxes :89-94, post_process_word_preds :96-106,
//  merge_intersecting_boxes :108-184, _merge_rotated_boxes :187-216, boxes_to_polygons :219-250,
//  polygons_to_rotated_boxes :253-286) and glass/postprocess/post_processor_academic.py:26-35 +
// glass/evaluation/text_evaluator.py:323-348 / glass/modeling/recognition/text_encoder.py:81-151
// (argmax text decode, word score = product of the probabilities up to and including the stop symbol,
// text-score filter).  On the host this tail costs ~56 ms per image (Python merge loop with a device
// round trip per iteration), i.e. 6x the whole network; here it is one workgroup per image, everything in
// LDS (<= 128 boxes), no host involvement until the final read of the compact results.
//
// Latency (round 4): the merge loop is a chain of small dependent phases, so what matters is the length of ONE thread's
// work in each.  The dynamically indexed point lists of the rotated IoU (24 points) and of the minimum-area rectangle
// (32 double points) used to live in private memory = scratch, a memory round trip per access: 25-30 us per phase,
// ~85 us per merge iteration, 0.4-1.0 ms per launch on the bench's 32 overlapping words per image and 10 ms on 128
// dense ones.  They are LDS columns now (one per thread, 64 KB shared by the phases), the candidate filter is one box
// per thread instead of a loop of thread 0, and the per-(box, step) argmax over the characters runs before this kernel
// on the whole chip (text_argmax_kernel: a wavefront per row) instead of at its end on one CU.
//
// Semantics kept on purpose: the merge loop works on a SNAPSHOT of the boxes, writes merged boxes back
// first through all first indices then through all second indices of the valid pairs in row-major pair
// order ("last pair wins", the CPU result of the reference's repeated-index assignment), re-orders the
// survivors by descending score through the 0.99 NMS, and passes the merged angle in RADIANS to the
// degree-valued orientation correction (reference quirk, :203-206 vs :267).
#include "rotated_iou.h"

// No fused multiply-add anywhere in this file: the reference computes these boxes with separate torch float32 ops and the
// pinned restatement of cv2.minAreaRect (glass_amd/postprocess/post_processor_rotated_boxes.py:min_area_rect) in numpy
// float64, neither of which fuses.  With hipcc's default (fuse wherever it can) WHICH product of `a*b + c*d` is fused
// depends on the surrounding code, so two builds of the same formulas differ by an ulp in one merge in a few thousand -
// enough to tip the choice between two equal-area hull edges, or a threshold test three merges later.
#pragma clang fp contract(off)

constexpr int PP_KMAX = 128;
constexpr int PP_THREADS = 256;
constexpr int PP_TMAX = 64;
constexpr int PP_WORK_BYTES = 65536;     // max(24 Pt x 256 threads, 32 DPt x 128 boxes)

struct PPParams {
  const float* boxes; const float* scores; const int* counts; const int* text_arg; const float* text_max; const float* scale_xy;
  int N, K, T;
  float min_box_dim, valid_score, detect_thr, merge_ioa, height_ratio, max_angle_diff, minimal_ioa, text_thr;
  int stop_index, do_text;
  float* out_boxes; float* out_scores; float* out_poly; int* out_src; int* out_char; float* out_text_score;
  int* out_text_len; int* out_count;
};

// ---- minimum-area enclosing rectangle of <= 8 points (double precision, same algorithm and tie order as
// glass_amd/postprocess/post_processor_rotated_boxes.py:min_area_rect): monotone-chain hull of the unique
// points sorted by (x, y), then the first hull edge of minimal bounding-rectangle area.
struct DPt { double x, y; };
__device__ inline double dcross(DPt o, DPt a, DPt b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

// per-thread point lists in LDS: element i of thread t at base[i * stride + t] (conflict-free across a wavefront)
struct LdsPts {
  Pt* base;
  __device__ __forceinline__ Pt& operator[](int i) const { return base[i * PP_THREADS]; }
};
struct LdsDPts {
  DPt* base;
  __device__ __forceinline__ DPt& operator[](int i) const { return base[i * PP_KMAX]; }
};

// Hull of the 8 corner points of two boxes.  `A`: 32 points of storage (sorted points 0..7, hull 8..23, upper chain 24..31);
// returns the number of hull points (<= 8).
__device__ int merge_hull(const float* pts /*[8][2]*/, LdsDPts A) {
  const LdsDPts p{A.base}, hull{A.base + 8 * PP_KMAX}, upper{A.base + 24 * PP_KMAX};
  int n = 8;
  for (int i = 0; i < 8; ++i) { p[i].x = (double)pts[2 * i]; p[i].y = (double)pts[2 * i + 1]; }
  // insertion sort by (x, y), then drop exact duplicates
  for (int i = 1; i < n; ++i) {
    const DPt key = p[i];
    int j = i - 1;
    while (j >= 0 && (p[j].x > key.x || (p[j].x == key.x && p[j].y > key.y))) { p[j + 1] = p[j]; --j; }
    p[j + 1] = key;
  }
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (m == 0 || p[i].x != p[m - 1].x || p[i].y != p[m - 1].y) p[m++] = p[i];
  n = m;
  int hn = 0;
  if (n <= 2) {
    for (int i = 0; i < n; ++i) hull[hn++] = p[i];
  } else {
    int nl = 0, nu = 0;                           // (the lower chain is built in place at the head of `hull`)
    for (int i = 0; i < n; ++i) {
      const DPt pi = p[i];
      while (nl >= 2 && dcross(hull[nl - 2], hull[nl - 1], pi) <= 0) --nl;
      hull[nl++] = pi;
    }
    for (int i = n - 1; i >= 0; --i) {
      const DPt pi = p[i];
      while (nu >= 2 && dcross(upper[nu - 2], upper[nu - 1], pi) <= 0) --nu;
      upper[nu++] = pi;
    }
    hn = nl - 1;
    for (int i = 0; i < nu - 1; ++i) hull[hn++] = upper[i];
  }
  return hn;
}

// bounding rectangle of the hull with one side along hull edge i (false: zero-length edge)
struct EdgeRect { double area, cx, cy, w, h, ang; };
__device__ bool hull_edge_rect(LdsDPts hull, int hn, int i, EdgeRect& r) {
  const DPt a = hull[i], b = hull[(i + 1) % hn];
  const double ex = b.x - a.x, ey = b.y - a.y;
  const double nrm = hypot(ex, ey);
  if (nrm == 0) return false;
  const double ux = ex / nrm, uy = ey / nrm, vx = -uy, vy = ux;
  double pumin = 1e300, pumax = -1e300, pvmin = 1e300, pvmax = -1e300;
  for (int k = 0; k < hn; ++k) {
    const DPt hk = hull[k];
    const double pu = hk.x * ux + hk.y * uy, pv = hk.x * vx + hk.y * vy;
    pumin = fmin(pumin, pu); pumax = fmax(pumax, pu); pvmin = fmin(pvmin, pv); pvmax = fmax(pvmax, pv);
  }
  const double ww = pumax - pumin, hh = pvmax - pvmin;
  r.area = ww * hh;
  const double cu = (pumax + pumin) / 2, cv = (pvmax + pvmin) / 2;
  r.cx = ux * cu + vx * cv; r.cy = uy * cu + vy * cv; r.w = ww; r.h = hh;
  r.ang = atan2(uy, ux) * 57.29577951308232;
  return true;
}

__device__ inline float floor_mod_pp(float a, float b) {   // torch.remainder semantics
  float m = fmodf(a, b);
  if (m != 0.f && ((b < 0.f) != (m < 0.f))) m += b;
  return m;
}

__device__ inline double pymod(double a, double b) {   // Python float % for b > 0
  double m = fmod(a, b);
  if (m != 0 && m < 0) m += b;
  return m;
}

// boxes_to_polygons (:219-250) for one box, float32 like the reference's torch ops
__device__ inline void box_polygon(const float* b, float* poly /*[4][2]*/) {
  const float cx = b[0], cy = b[1], w = b[2], h = b[3], a = b[4];
  const float t = (-a / 180.f) * 3.14159265358979323846f;
  float s, c;
  sincosf(t, &s, &c);
  poly[0] = cx + (h * s - w * c) / 2;  poly[1] = cy - (h * c + w * s) / 2;
  poly[2] = cx + (h * s + w * c) / 2;  poly[3] = cy - (h * c - w * s) / 2;
  poly[4] = cx - (h * s - w * c) / 2;  poly[5] = cy + (h * c + w * s) / 2;
  poly[6] = cx - (h * s + w * c) / 2;  poly[7] = cy + (h * c - w * s) / 2;
}

// _merge_rotated_boxes (:187-216) + polygons_to_rotated_boxes (:253-286) for one pair, in two parts around the minimum-area
// rectangle (cx, cy, w, h, ang) of the pair's 8 corners: merge_corners before, merge_finish after
__device__ inline void merge_corners(const float* b1, const float* b2, float* pts /*[8][2]*/) {
  box_polygon(b1, pts);
  box_polygon(b2, pts + 8);
}
__device__ void merge_finish(const float* b1, const float* b2, float s1, float s2, double cx, double cy, double w, double h, double ang,
                             float* out) {
  const float a1 = b1[4] * 3.14159265358979323846f / 180.f, a2 = b2[4] * 3.14159265358979323846f / 180.f;
  const double orient = (double)(s1 >= s2 ? a1 : a2);          // radians (reference quirk)
  double angle = 90.0 - ang;
  double diff = pymod((orient - angle) + 180.0, 360.0) - 180.0;
  double width, height;
  if (-45 < diff && diff <= 45) { width = h; height = w; }
  else if (45 < diff && diff <= 135) { width = w; height = h; angle += 90; }
  else if (-135 < diff && diff <= -45) { width = w; height = h; angle -= 90; }
  else { width = h; height = w; angle += 180; }
  angle = pymod(angle + 180.0, 360.0) - 180.0;
  out[0] = (float)cx; out[1] = (float)cy; out[2] = (float)width; out[3] = (float)height; out[4] = (float)angle;
}

// Circumscribed circles disjoint (with slack) -> the rectangles cannot intersect -> IoU is exactly 0; skips the
// polygon clipping for the (vast majority of) far-apart word pairs.
__device__ __forceinline__ bool pp_far_apart(const float* a, const float* b) {
  const float dx = a[0] - b[0], dy = a[1] - b[1];
  const float r = 0.5f * (sqrtf(a[2] * a[2] + a[3] * a[3]) + sqrtf(b[2] * b[2] + b[3] * b[3]));
  return dx * dx + dy * dy > r * r * 1.001f + 1e-2f;
}

// q-th pair (i < j) of the strict upper triangle of an n x n matrix, row-major
__device__ __forceinline__ void pp_pair(int q, int n, int& i, int& j) {
  const float b = (float)(2 * n - 1);
  int r = (int)((b - sqrtf(fmaxf(b * b - 8.f * (float)q, 0.f))) * 0.5f);
  r = max(0, min(r, n - 2));
  while (r + 1 <= n - 2 && (r + 1) * (2 * n - (r + 1) - 1) / 2 <= q) ++r;
  while (r > 0 && r * (2 * n - r - 1) / 2 > q) --r;
  i = r;
  j = q - r * (2 * n - r - 1) / 2 + r + 1;
}

// All pairs (i < j) of n boxes: on_far(i, j) for the far-apart ones right away; the others - the ones that run the long
// polygon-clipping code - are first collected per wavefront (ballot compaction into `queue`, 128 entries per wavefront)
// and handed to on_near(i, j) 64 at a time, so that a wavefront runs the long path with all lanes busy and about
// (near pairs / 256) times instead of once per loop trip with whatever lanes happen to need it.
template <class Far, class Near>
__device__ __forceinline__ void pp_for_pairs(int n, const float (*B)[5], unsigned short* queue, int tid, Far on_far, Near on_near) {
  const int npair = n * (n - 1) / 2, lane = tid & 63;
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int qn = 0;                                                       // (wavefront-uniform)
  for (int base = tid & ~63; base < npair; base += PP_THREADS) {
    const int q = base + lane;
    int i = 0, j = 0;
    bool near = false;
    if (q < npair) {
      pp_pair(q, n, i, j);
      near = !pp_far_apart(B[i], B[j]);
      if (!near) on_far(i, j);
    }
    const unsigned long long m = __ballot(near);
    if (near) queue[qn + __popcll(m & below)] = (unsigned short)(i * PP_KMAX + j);
    qn += __popcll(m);
    if (qn >= 64) {
      qn -= 64;
      const int e = queue[qn + lane];
      on_near(e / PP_KMAX, e % PP_KMAX);
    }
  }
  if (lane < qn) {
    const int e = queue[lane];
    on_near(e / PP_KMAX, e % PP_KMAX);
  }
}

__global__ __launch_bounds__(PP_THREADS) void postprocess_words_kernel(PPParams p) {
  __shared__ float bx[PP_KMAX][5], snap[PP_KMAX][5], tmpb[PP_KMAX][5];
  __shared__ float sc[PP_KMAX], tmps[PP_KMAX];
  __shared__ int src[PP_KMAX], tmpi[PP_KMAX], order[PP_KMAX];
  __shared__ float ioa[PP_KMAX][PP_KMAX + 1];      // IoA (merge) / IoU (NMS) matrix
  __shared__ unsigned char flag[PP_KMAX];
  __shared__ int s_n, s_any, s_wave_cnt[2], s_reuse;
  __shared__ float rc2[PP_KMAX], rs2[PP_KMAX], tmpc2[PP_KMAX], tmps2[PP_KMAX];   // make_rbox's half cos / sin of bx (= of snap until the write-back)
  __shared__ int old_of[PP_KMAX], tmpo[PP_KMAX], s_hn[PP_KMAX];
  __shared__ unsigned short queue[PP_THREADS / 64][128];
  // the threads' point lists (rotated IoU: 24 points per thread; merge: 32 double points per box); the phases that use the
  // two views are separated by barriers
  __shared__ __attribute__((aligned(16))) unsigned char work[PP_WORK_BYTES];
  const int n_img = blockIdx.x, tid = threadIdx.x;
  const int cnt = min(p.counts[n_img], min(p.K, PP_KMAX));
  const float* gb = p.boxes + (long)n_img * p.K * 5;
  const float* gs = p.scores + (long)n_img * p.K;
  LdsPts iou_pts{reinterpret_cast<Pt*>(work) + tid};
#ifdef GLASS_PP_STAMPS      // phase clocks of every workgroup, printed at its end (scripts/build_variant_lib.sh ppst -DGLASS_PP_STAMPS)
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime(), tstart = tlast;
  int iters = 0;
#define PP_STAMP(k) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); st[k] += tn - tlast; tlast = tn; }
#else
#define PP_STAMP(k)
#endif

  // ---- load (+ optional RotatedBoxes.scale of the runner's un-scaling), filter_small_boxes, score >= valid:
  // candidate j on thread j (wavefronts 0 and 1), survivors compacted in order through the two ballots
  {
    const int j = tid;
    float b[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, sj = 0.f;
    bool keep = false;
    if (j < cnt) {
      const float sx = p.scale_xy ? p.scale_xy[2 * n_img] : 1.f, sy = p.scale_xy ? p.scale_xy[2 * n_img + 1] : 1.f;
#pragma unroll
      for (int e = 0; e < 5; ++e) b[e] = gb[5 * j + e];
      sj = gs[j];
      if (p.scale_xy && (sx != 1.f || sy != 1.f)) {     // GlassRunner un-scales only when the ratio != 1
        b[0] *= sx; b[1] *= sy;
        const float theta = b[4] * 3.14159265358979323846f / 180.0f;
        float sn, cs;
        sincosf(theta, &sn, &cs);
        b[2] *= sqrtf((sx * cs) * (sx * cs) + (sy * sn) * (sy * sn));
        b[3] *= sqrtf((sx * sn) * (sx * sn) + (sy * cs) * (sy * cs));
        b[4] = atan2f(sx * sn, sy * cs) * 180.0f / 3.14159265358979323846f;
      }
      keep = fminf(b[2], b[3]) >= p.min_box_dim && sj >= p.valid_score;
    }
    const unsigned long long kept = __ballot(keep);                 // (of this wavefront)
    if (tid < PP_KMAX && (tid & 63) == 0) s_wave_cnt[tid >> 6] = __popcll(kept);
    __syncthreads();
    if (keep) {
      const int lane = tid & 63;
      const int d = (tid >= 64 ? s_wave_cnt[0] : 0) + __popcll(kept & (lane == 0 ? 0ull : (~0ull >> (64 - lane))));
#pragma unroll
      for (int e = 0; e < 5; ++e) bx[d][e] = b[e];
      sc[d] = sj;
      src[d] = j;
    }
    if (tid == 0) { s_n = s_wave_cnt[0] + s_wave_cnt[1]; s_reuse = 0; }
  }
  __syncthreads();
  for (int i = tid; i < s_n; i += PP_THREADS) {
    const RBox r = make_rbox(bx[i][0], bx[i][1], bx[i][2], bx[i][3], bx[i][4]);
    rc2[i] = r.c2; rs2[i] = r.s2;
  }
  auto rbox_of = [&](const float (*B)[5], int i) { return RBox{B[i][0], B[i][1], B[i][2], B[i][3], rc2[i], rs2[i]}; };
  PP_STAMP(0)

  // ---- merge_intersecting_boxes
  for (int iter = 0; iter < 4 * PP_KMAX; ++iter) {
    const int n = s_n;
    if (n == 0) break;
    for (int i = tid; i < n * 5; i += PP_THREADS) snap[i / 5][i % 5] = bx[i / 5][i % 5];
    if (tid == 0) s_any = 0;
    __syncthreads();
    // IoA matrix (upper triangle), same algebra as pairwise_ioa_rotated (glass/structures/boxes.py:33-48): from the pairs'
    // IoU.  From the second iteration on that IoU is the one the previous iteration's NMS computed on these very boxes
    // (same arguments in the same order as long as the score sort did not swap anybody: `s_reuse`), only re-indexed.
    const int npair = n * (n - 1) / 2;
    auto ioa_of = [&](int i, int j, float iou) {
      const float a1 = snap[i][2] * snap[i][3], a2 = snap[j][2] * snap[j][3];
      const float inter = (a1 + a2) * iou / (1.f + iou);
      return inter / fminf(a1, a2);
    };
    // Storage: the NMS writes its IoU into the UPPER triangle (row < column, in its own box order); the IoA of this
    // iteration goes into the LOWER triangle ([j][i]), so the re-indexing reads and writes never touch the same entry
    // and needs no staging; the pair flags below then overwrite the upper triangle, which nobody reads any more.
    if (s_reuse) {
      for (int q = tid; q < npair; q += PP_THREADS) {
        int i, j;
        pp_pair(q, n, i, j);
        ioa[j][i] = pp_far_apart(snap[i], snap[j]) ? 0.f : ioa_of(i, j, ioa[old_of[i]][old_of[j]]);
      }
    } else {
      pp_for_pairs(n, snap, queue[tid >> 6], tid, [&](int i, int j) { ioa[j][i] = 0.f; },
                   [&](int i, int j) { ioa[j][i] = ioa_of(i, j, rotated_iou_in(rbox_of(snap, i), rbox_of(snap, j), iou_pts)); });
    }
    __syncthreads();
    PP_STAMP(1)
    // valid pair mask -> ioa[i][j] (upper triangle as flag storage: 1.0 = valid)
    for (int q = tid; q < npair; q += PP_THREADS) {
      int i, j;
      pp_pair(q, n, i, j);
      bool ok = false;
      const float v = ioa[j][i];
      if (v >= p.minimal_ioa) {
        float ad = snap[j][4] - snap[i][4];
        ad = fabsf(floor_mod_pp(ad + 180.f, 360.f) - 180.f);
        const bool sim_angle = (ad < p.max_angle_diff) || (ad > (180.f - p.max_angle_diff));
        const float hr = snap[j][3] / snap[i][3];
        const bool sim_h = (p.height_ratio < hr) && (hr < (1.f / (p.height_ratio + 1e-6f)));
        const bool vs = fminf(sc[i], sc[j]) >= p.valid_score;
        ok = sim_angle && sim_h && vs && (v >= p.merge_ioa);
      }
      ioa[i][j] = ok ? 1.f : 0.f;
      if (ok) s_any = 1;
    }
    __syncthreads();
    PP_STAMP(2)
    if (!s_any) break;
#ifdef GLASS_PP_STAMPS
    ++iters;
#endif
    // write-back: box b takes the merge of its LAST valid pair as second element (largest i), else of its
    // last valid pair as first element (largest j); all merges computed from the snapshot
    // Eight lanes per box: lane 0 builds the hull of the pair's corners (LDS column b), every lane evaluates the bounding
    // rectangle along one hull edge (<= 8), and the reference loop's choice - the first edge of strictly smaller area -
    // is folded over the lanes' areas in edge order.
    for (int b0 = 0; b0 < n; b0 += PP_THREADS / 8) {
      const int b = b0 + (tid >> 3), gl = tid & 7;
      int pi = -1, pj = -1;
      if (b < n) {
        for (int i = b - 1; i >= 0; --i)
          if (ioa[i][b] == 1.f) { pi = i; pj = b; break; }
        if (pi < 0)
          for (int j = n - 1; j > b; --j)
            if (ioa[b][j] == 1.f) { pi = b; pj = j; break; }
      }
      const bool act = pi >= 0;
      const LdsDPts A{reinterpret_cast<DPt*>(work) + (act ? b : 0)}, hull{A.base + 8 * PP_KMAX};
      if (act && gl == 0) {
        float pts[16];
        merge_corners(snap[pi], snap[pj], pts);
        s_hn[b] = merge_hull(pts, A);
      }
      __syncthreads();
      const int hn = act ? s_hn[b] : 0;
      EdgeRect r{0., 0., 0., 0., 0., 0.};
      const bool ok = act && hn >= 3 && gl < hn && hull_edge_rect(hull, hn, gl, r);
      int win = 0;
      double best = -1.0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const double ae = __shfl(r.area, e, 8);
        const int oke = __shfl((int)ok, e, 8);
        if (oke && (best < 0 || ae < best)) { best = ae; win = e; }
      }
      double cx = __shfl(r.cx, win, 8), cy = __shfl(r.cy, win, 8), w = __shfl(r.w, win, 8), h = __shfl(r.h, win, 8),
             ang = __shfl(r.ang, win, 8);
      if (act && gl == 0) {
        if (hn == 1) {
          const DPt h0 = hull[0];
          cx = h0.x; cy = h0.y; w = 0; h = 0; ang = 0;
        } else if (hn == 2) {
          const DPt h0 = hull[0], h1 = hull[1];
          const double dx = h1.x - h0.x, dy = h1.y - h0.y;
          cx = (h0.x + h1.x) / 2; cy = (h0.y + h1.y) / 2;
          w = hypot(dx, dy); h = 0; ang = atan2(dy, dx) * 57.29577951308232;
        }
        merge_finish(snap[pi], snap[pj], sc[pi], sc[pj], cx, cy, w, h, ang, bx[b]);
        const RBox rb = make_rbox(bx[b][0], bx[b][1], bx[b][2], bx[b][3], bx[b][4]);
        rc2[b] = rb.c2; rs2[b] = rb.s2;
      }
    }
    __syncthreads();
    PP_STAMP(3)
    // nms_rotated(0.99): IoU matrix, stable descending-score order, greedy suppression, reorder survivors
    pp_for_pairs(n, bx, queue[tid >> 6], tid, [&](int i, int j) { ioa[i][j] = 0.f; },
                 [&](int i, int j) { ioa[i][j] = rotated_iou_in(rbox_of(bx, i), rbox_of(bx, j), iou_pts); });
    // stable descending order by rank counting (ties keep the lower index first, as the host's stable sort)
    for (int i = tid; i < n; i += PP_THREADS) {
      const float si = sc[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += (sc[j] > si || (sc[j] == si && j < i)) ? 1 : 0;
      order[rank] = i;
    }
    __syncthreads();
    PP_STAMP(4)
    // greedy suppression by ONE wavefront (no barriers): lane l owns sorted positions l and l + 64
    if (tid < 64) {
      const int c0 = tid, c1 = tid + 64;
      const int o0 = c0 < n ? order[c0] : 0, o1 = c1 < n ? order[c1] : 0;
      // who could suppress my two positions: bit a of (s0l, s0h) / (s1l, s1h) = the box at sorted position a < c has
      // IoU >= 0.99 with mine (independent LDS reads); then the sequential part runs on wavefront-uniform 128-bit masks:
      // position a, if still alive, removes every later position that has bit a set (one ballot each)
      unsigned long long s0l = 0, s0h = 0, s1l = 0, s1h = 0;
      for (int a0 = 0; a0 < n; a0 += 8) {                     // (8 positions per trip: 8 + 16 independent LDS reads in flight)
        int ia[8];
        float v0[8], v1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ia[u] = order[min(a0 + u, n - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v0[u] = ioa[min(ia[u], o0)][max(ia[u], o0)];
          v1[u] = ioa[min(ia[u], o1)][max(ia[u], o1)];
        }
        unsigned int m0 = 0, m1 = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int a = a0 + u;
          m0 |= (a < n && a < c0 && c0 < n && v0[u] >= 0.99f) ? (1u << u) : 0u;
          m1 |= (a < n && a < c1 && c1 < n && v1[u] >= 0.99f) ? (1u << u) : 0u;
        }
        if (a0 < 64) { s0l |= (unsigned long long)m0 << a0; s1l |= (unsigned long long)m1 << a0; }
        else { s0h |= (unsigned long long)m0 << (a0 - 64); s1h |= (unsigned long long)m1 << (a0 - 64); }
      }
      unsigned long long reml = n >= 64 ? 0ull : (~0ull << n), remh = n >= 128 ? 0ull : (n <= 64 ? ~0ull : (~0ull << (n - 64)));
      for (int a = 0; a < n; ++a) {
        const bool gone = a < 64 ? (reml >> a) & 1 : (remh >> (a - 64)) & 1;
        if (gone) continue;                                  // wave-uniform
        const bool h0 = a < 64 ? (s0l >> a) & 1 : (s0h >> (a - 64)) & 1, h1 = a < 64 ? (s1l >> a) & 1 : (s1h >> (a - 64)) & 1;
        reml |= __ballot(h0);
        remh |= __ballot(h1);
      }
      const int rem0 = (int)((reml >> tid) & 1), rem1 = (int)((remh >> tid) & 1);
      // ordered compaction of the survivors (sorted order)
      const unsigned long long k0 = __ballot(c0 < n && !rem0), k1 = __ballot(c1 < n && !rem1);
      const unsigned long long below = tid == 0 ? 0ull : (~0ull >> (64 - tid));
      const int d0 = __popcll(k0 & below), d1 = __popcll(k0) + __popcll(k1 & below);
      if (c0 < n && !rem0) {
        for (int e = 0; e < 5; ++e) tmpb[d0][e] = bx[o0][e];
        tmps[d0] = sc[o0];
        tmpi[d0] = src[o0];
        tmpo[d0] = o0; tmpc2[d0] = rc2[o0]; tmps2[d0] = rs2[o0];
      }
      if (c1 < n && !rem1) {
        for (int e = 0; e < 5; ++e) tmpb[d1][e] = bx[o1][e];
        tmps[d1] = sc[o1];
        tmpi[d1] = src[o1];
        tmpo[d1] = o1; tmpc2[d1] = rc2[o1]; tmps2[d1] = rs2[o1];
      }
      if (tid == 0) { s_n = __popcll(k0) + __popcll(k1); s_reuse = 1; }
    }
    __syncthreads();
    {
      const int m = s_n;
      for (int i = tid; i < m; i += PP_THREADS) {
        for (int e = 0; e < 5; ++e) bx[i][e] = tmpb[i][e];
        sc[i] = tmps[i];
        src[i] = tmpi[i];
        old_of[i] = tmpo[i]; rc2[i] = tmpc2[i]; rs2[i] = tmps2[i];
        if (i > 0 && tmpo[i - 1] > tmpo[i]) s_reuse = 0;      // the sort moved a lower-scored box behind: argument orders flip
      }
    }
    __syncthreads();
    PP_STAMP(5)
  }
  __syncthreads();
  PP_STAMP(5)

  // ---- text decode: the survivors' rows of the per-(box, step) argmax / maximum (text_argmax_kernel) into LDS (the
  // IoA matrix storage is free now)
  int* chr = reinterpret_cast<int*>(&ioa[0][0]);                  // [n][T]
  float* prb = &ioa[0][0] + PP_KMAX * PP_TMAX;                    // [n][T], T <= PP_TMAX (2 x 128 x 64 <= 128 x 129 floats)
  const int n_fin = s_n;
  if (p.do_text) {
    for (int e = tid; e < n_fin * p.T; e += PP_THREADS) {
      const int i = e / p.T, t = e - i * p.T;
      const long g = ((long)n_img * p.K + src[i]) * p.T + t;
      chr[e] = p.text_arg[g];
      prb[e] = p.text_max[g];
    }
  }
  __syncthreads();
  PP_STAMP(6)
  // per box: word score = product of the probabilities before the first stop symbol and at it (or of all T
  // when there is none), text length = characters before the stop
  for (int i = tid; i < n_fin; i += PP_THREADS) {
    float tscore = 1.f;
    int tlen = 0;
    if (p.do_text) {
      bool stopped = false;
      for (int t = 0; t < p.T; ++t) {
        const int bi = chr[i * p.T + t];
        bool take = false;
        if (!stopped) {
          take = true;
          if (bi == p.stop_index) stopped = true; else ++tlen;
        }
        if (take) tscore *= prb[i * p.T + t];
      }
    }
    tmps[i] = tscore;
    tmpi[i] = tlen;
    flag[i] = (sc[i] >= p.detect_thr) && (!p.do_text || tscore >= p.text_thr);
  }
  __syncthreads();
  // ---- ordered compaction + outputs
  if (tid == 0) {
    int m = 0;
    for (int i = 0; i < n_fin; ++i)
      if (flag[i]) order[m++] = i;
    s_n = m;
  }
  __syncthreads();
  const int m_out = s_n;
  for (int d = tid; d < m_out; d += PP_THREADS) {
    const int i = order[d];
    const long o = (long)n_img * p.K + d;
    for (int e = 0; e < 5; ++e) p.out_boxes[o * 5 + e] = bx[i][e];
    p.out_scores[o] = sc[i];
    p.out_src[o] = src[i];
    box_polygon(bx[i], p.out_poly + o * 8);
    p.out_text_score[o] = tmps[i];
    p.out_text_len[o] = tmpi[i];
    if (p.do_text)
      for (int t = 0; t < p.T; ++t) p.out_char[o * p.T + t] = chr[i * p.T + t];
  }
  if (tid == 0) p.out_count[n_img] = m_out;
#ifdef GLASS_PP_STAMPS
  PP_STAMP(7)
  if (tid == 0)
    printf("[pp] img %d n0 %d iters %d | filter %llu ioa %llu mask %llu merge %llu nms-iou %llu greedy %llu text %llu out %llu | total %llu cycles\n",
           n_img, cnt, iters, st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7], tlast - tstart);
#endif
}

// Per-(box, step) argmax over the characters (reference text_encoder.py:81-151 `preds_prob.max(dim=2)`): one wavefront
// per row of `text` [N,K,T,C], the first maximum wins (as torch.max); rows of padding boxes (k >= counts[n]) are skipped
// and their outputs left untouched.
__global__ __launch_bounds__(256) void text_argmax_kernel(const float* __restrict__ text, const int* __restrict__ counts, int N, int K,
                                                          int T, int C, int* __restrict__ out_arg, float* __restrict__ out_max) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long)N * K * T) return;
  const int n = (int)(row / ((long)K * T)), k = (int)((row / T) % K);
  if (k >= min(counts[n], K)) return;
  const float* r = text + row * (long)C;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = r[c];
    if (bi == 0x7fffffff || v > best) { best = v; bi = c; }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
  }
  if (lane == 0) { out_arg[row] = bi; out_max[row] = best; }
}

extern "C" int glass_text_argmax(const float* text, const int* counts, int N, int K, int T, int C, int* out_arg, float* out_max,
                                 glass_stream_t stream) {
  if ((long)N * K * T == 0) return GLASS_OK;
  GLASS_CHECK_ARG(text && counts && out_arg && out_max, "glass_text_argmax: null pointer");
  GLASS_CHECK_ARG(N > 0 && K > 0 && T > 0 && C > 0, "glass_text_argmax: bad shape N=%d K=%d T=%d C=%d", N, K, T, C);
  const long rows = (long)N * K * T;
  GLASS_CHECK_ARG((rows + 3) / 4 <= 0x7fffffffL, "glass_text_argmax: too many rows");
  hipLaunchKernelGGL(text_argmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, text, counts, N, K, T,
                     C, out_arg, out_max);
  GLASS_CHECK_LAUNCH("glass_text_argmax");
  return GLASS_OK;
}

extern "C" int glass_postprocess_words(const float* boxes, const float* scores, const int* counts, const int* text_arg,
                                       const float* text_max, const float* scale_xy, int N, int K, int T,
                                       const float* thresholds8_host, int stop_index, float* out_boxes, float* out_scores,
                                       float* out_polygons, int* out_src, int* out_char, float* out_text_score, int* out_text_len,
                                       int* out_count, glass_stream_t stream) {
  if (N == 0) return GLASS_OK;
  GLASS_CHECK_ARG(K >= 0 && K <= PP_KMAX, "glass_postprocess_words: K=%d (max %d)", K, PP_KMAX);
  GLASS_CHECK_ARG(counts && thresholds8_host && out_count, "glass_postprocess_words: null pointer");
  GLASS_CHECK_ARG(K == 0 || (boxes && scores && out_boxes && out_scores && out_polygons && out_src && out_char &&
                             out_text_score && out_text_len), "glass_postprocess_words: null pointer");
  GLASS_CHECK_ARG((text_arg == nullptr) == (text_max == nullptr), "glass_postprocess_words: text_arg and text_max go together");
  GLASS_CHECK_ARG(!text_arg || (T > 0 && T <= PP_TMAX), "glass_postprocess_words: text needs 0 < T <= %d", PP_TMAX);
  PPParams p;
  p.boxes = boxes; p.scores = scores; p.counts = counts; p.text_arg = text_arg; p.text_max = text_max; p.scale_xy = scale_xy;
  p.N = N; p.K = K; p.T = text_arg ? T : 1;
  p.min_box_dim = thresholds8_host[0]; p.valid_score = thresholds8_host[1]; p.detect_thr = thresholds8_host[2];
  p.merge_ioa = thresholds8_host[3]; p.height_ratio = thresholds8_host[4]; p.max_angle_diff = thresholds8_host[5];
  p.minimal_ioa = thresholds8_host[6]; p.text_thr = thresholds8_host[7];
  p.stop_index = stop_index; p.do_text = text_arg ? 1 : 0;
  p.out_boxes = out_boxes; p.out_scores = out_scores; p.out_poly = out_polygons; p.out_src = out_src; p.out_char = out_char;
  p.out_text_score = out_text_score; p.out_text_len = out_text_len; p.out_count = out_count;
  hipLaunchKernelGGL(postprocess_words_kernel, dim3(N), dim3(PP_THREADS), 0, (hipStream_t)stream, p);
  GLASS_CHECK_LAUNCH("glass_postprocess_words");
  return GLASS_OK;
}
