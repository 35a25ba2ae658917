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
// Semantics kept on purpose: the merge loop works on a SNAPSHOT of the boxes, writes merged boxes back
// first through all first indices then through all second indices of the valid pairs in row-major pair
// order ("last pair wins", the CPU result of the reference's repeated-index assignment), re-orders the
// survivors by descending score through the 0.99 NMS, and passes the merged angle in RADIANS to the
// degree-valued orientation correction (reference quirk, :203-206 vs :267).
#include "rotated_iou.h"

constexpr int PP_KMAX = 128;
constexpr int PP_THREADS = 256;

struct PPParams {
  const float* boxes; const float* scores; const int* counts; const float* text; const float* scale_xy;
  int N, K, T, C;
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

__device__ void min_area_rect8(const float* pts /*[8][2]*/, double& cx, double& cy, double& w, double& h, double& ang) {
  DPt p[8];
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
  DPt hull[16];
  int hn = 0;
  if (n <= 2) {
    for (int i = 0; i < n; ++i) hull[hn++] = p[i];
  } else {
    DPt lower[8], upper[8];
    int nl = 0, nu = 0;
    for (int i = 0; i < n; ++i) {
      while (nl >= 2 && dcross(lower[nl - 2], lower[nl - 1], p[i]) <= 0) --nl;
      lower[nl++] = p[i];
    }
    for (int i = n - 1; i >= 0; --i) {
      while (nu >= 2 && dcross(upper[nu - 2], upper[nu - 1], p[i]) <= 0) --nu;
      upper[nu++] = p[i];
    }
    for (int i = 0; i < nl - 1; ++i) hull[hn++] = lower[i];
    for (int i = 0; i < nu - 1; ++i) hull[hn++] = upper[i];
  }
  if (hn == 1) { cx = hull[0].x; cy = hull[0].y; w = 0; h = 0; ang = 0; return; }
  if (hn == 2) {
    const double dx = hull[1].x - hull[0].x, dy = hull[1].y - hull[0].y;
    cx = (hull[0].x + hull[1].x) / 2; cy = (hull[0].y + hull[1].y) / 2;
    w = hypot(dx, dy); h = 0; ang = atan2(dy, dx) * 57.29577951308232;
    return;
  }
  double best = -1.0;
  for (int i = 0; i < hn; ++i) {
    const DPt a = hull[i], b = hull[(i + 1) % hn];
    const double ex = b.x - a.x, ey = b.y - a.y;
    const double nrm = hypot(ex, ey);
    if (nrm == 0) continue;
    const double ux = ex / nrm, uy = ey / nrm, vx = -uy, vy = ux;
    double pumin = 1e300, pumax = -1e300, pvmin = 1e300, pvmax = -1e300;
    for (int k = 0; k < hn; ++k) {
      const double pu = hull[k].x * ux + hull[k].y * uy, pv = hull[k].x * vx + hull[k].y * vy;
      pumin = fmin(pumin, pu); pumax = fmax(pumax, pu); pvmin = fmin(pvmin, pv); pvmax = fmax(pvmax, pv);
    }
    const double ww = pumax - pumin, hh = pvmax - pvmin;
    if (best < 0 || ww * hh < best) {
      best = ww * hh;
      const double cu = (pumax + pumin) / 2, cv = (pvmax + pvmin) / 2;
      cx = ux * cu + vx * cv; cy = uy * cu + vy * cv; w = ww; h = hh;
      ang = atan2(uy, ux) * 57.29577951308232;
    }
  }
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

// _merge_rotated_boxes (:187-216) + polygons_to_rotated_boxes (:253-286) for one pair
__device__ void merge_pair(const float* b1, const float* b2, float s1, float s2, float* out) {
  float pts[16];
  box_polygon(b1, pts);
  box_polygon(b2, pts + 8);
  const float a1 = b1[4] * 3.14159265358979323846f / 180.f, a2 = b2[4] * 3.14159265358979323846f / 180.f;
  const double orient = (double)(s1 >= s2 ? a1 : a2);          // radians (reference quirk)
  double cx, cy, w, h, ang;
  min_area_rect8(pts, cx, cy, w, h, ang);
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

__global__ __launch_bounds__(PP_THREADS) void postprocess_words_kernel(PPParams p) {
  __shared__ float bx[PP_KMAX][5], snap[PP_KMAX][5], tmpb[PP_KMAX][5];
  __shared__ float sc[PP_KMAX], tmps[PP_KMAX];
  __shared__ int src[PP_KMAX], tmpi[PP_KMAX], order[PP_KMAX];
  __shared__ float ioa[PP_KMAX][PP_KMAX + 1];      // IoA (merge) / IoU (NMS) matrix
  __shared__ unsigned char flag[PP_KMAX];
  __shared__ int s_n, s_any;
  const int n_img = blockIdx.x, tid = threadIdx.x;
  const int cnt = min(p.counts[n_img], min(p.K, PP_KMAX));
  const float* gb = p.boxes + (long)n_img * p.K * 5;
  const float* gs = p.scores + (long)n_img * p.K;

  // ---- load (+ optional RotatedBoxes.scale of the runner's un-scaling), filter_small_boxes, score >= valid
  if (tid == 0) {
    int n = 0;
    const float sx = p.scale_xy ? p.scale_xy[2 * n_img] : 1.f, sy = p.scale_xy ? p.scale_xy[2 * n_img + 1] : 1.f;
    for (int j = 0; j < cnt; ++j) {
      float b[5] = {gb[5 * j], gb[5 * j + 1], gb[5 * j + 2], gb[5 * j + 3], gb[5 * j + 4]};
      if (p.scale_xy && (sx != 1.f || sy != 1.f)) {     // GlassRunner un-scales only when the ratio != 1
        b[0] *= sx; b[1] *= sy;
        const float theta = b[4] * 3.14159265358979323846f / 180.0f;
        float sn, cs;
        sincosf(theta, &sn, &cs);
        b[2] *= sqrtf((sx * cs) * (sx * cs) + (sy * sn) * (sy * sn));
        b[3] *= sqrtf((sx * sn) * (sx * sn) + (sy * cs) * (sy * cs));
        b[4] = atan2f(sx * sn, sy * cs) * 180.0f / 3.14159265358979323846f;
      }
      if (fminf(b[2], b[3]) >= p.min_box_dim && gs[j] >= p.valid_score) {
        for (int e = 0; e < 5; ++e) bx[n][e] = b[e];
        sc[n] = gs[j];
        src[n] = j;
        ++n;
      }
    }
    s_n = n;
  }
  __syncthreads();

  // ---- merge_intersecting_boxes
  for (int iter = 0; iter < 4 * PP_KMAX; ++iter) {
    const int n = s_n;
    if (n == 0) break;
    for (int i = tid; i < n * 5; i += PP_THREADS) snap[i / 5][i % 5] = bx[i / 5][i % 5];
    if (tid == 0) s_any = 0;
    __syncthreads();
    // IoA matrix (upper triangle), same algebra as pairwise_ioa_rotated (glass/structures/boxes.py:33-48)
    const int npair = n * (n - 1) / 2;
    for (int q = tid; q < npair; q += PP_THREADS) {
      int i, j;
      pp_pair(q, n, i, j);
      float v = 0.f;
      if (!pp_far_apart(snap[i], snap[j])) {
        const float iou = rotated_iou(make_rbox(snap[i][0], snap[i][1], snap[i][2], snap[i][3], snap[i][4]),
                                      make_rbox(snap[j][0], snap[j][1], snap[j][2], snap[j][3], snap[j][4]));
        const float a1 = snap[i][2] * snap[i][3], a2 = snap[j][2] * snap[j][3];
        const float inter = (a1 + a2) * iou / (1.f + iou);
        v = inter / fminf(a1, a2);
      }
      ioa[i][j] = v;
    }
    __syncthreads();
    // valid pair mask -> ioa[j][i] (lower triangle reused as flag storage: 1.0 = valid)
    for (int q = tid; q < npair; q += PP_THREADS) {
      int i, j;
      pp_pair(q, n, i, j);
      bool ok = false;
      const float v = ioa[i][j];
      if (v >= p.minimal_ioa) {
        float ad = snap[j][4] - snap[i][4];
        ad = fabsf(floor_mod_pp(ad + 180.f, 360.f) - 180.f);
        const bool sim_angle = (ad < p.max_angle_diff) || (ad > (180.f - p.max_angle_diff));
        const float hr = snap[j][3] / snap[i][3];
        const bool sim_h = (p.height_ratio < hr) && (hr < (1.f / (p.height_ratio + 1e-6f)));
        const bool vs = fminf(sc[i], sc[j]) >= p.valid_score;
        ok = sim_angle && sim_h && vs && (v >= p.merge_ioa);
      }
      ioa[j][i] = ok ? 1.f : 0.f;
      if (ok) s_any = 1;
    }
    __syncthreads();
    if (!s_any) break;
    // write-back: box b takes the merge of its LAST valid pair as second element (largest i), else of its
    // last valid pair as first element (largest j); all merges computed from the snapshot
    for (int b = tid; b < n; b += PP_THREADS) {
      int pi = -1, pj = -1;
      for (int i = b - 1; i >= 0; --i)
        if (ioa[b][i] == 1.f) { pi = i; pj = b; break; }
      if (pi < 0)
        for (int j = n - 1; j > b; --j)
          if (ioa[j][b] == 1.f) { pi = b; pj = j; break; }
      if (pi >= 0) merge_pair(snap[pi], snap[pj], sc[pi], sc[pj], bx[b]);
    }
    __syncthreads();
    // nms_rotated(0.99): IoU matrix, stable descending-score order, greedy suppression, reorder survivors
    for (int q = tid; q < npair; q += PP_THREADS) {
      int i, j;
      pp_pair(q, n, i, j);
      float iou = 0.f;
      if (!pp_far_apart(bx[i], bx[j]))
        iou = rotated_iou(make_rbox(bx[i][0], bx[i][1], bx[i][2], bx[i][3], bx[i][4]),
                          make_rbox(bx[j][0], bx[j][1], bx[j][2], bx[j][3], bx[j][4]));
      ioa[i][j] = iou;
      ioa[j][i] = iou;
    }
    // stable descending order by rank counting (ties keep the lower index first, as the host's stable sort)
    for (int i = tid; i < n; i += PP_THREADS) {
      const float si = sc[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += (sc[j] > si || (sc[j] == si && j < i)) ? 1 : 0;
      order[rank] = i;
    }
    __syncthreads();
    // greedy suppression by ONE wavefront (no barriers): lane l owns sorted positions l and l + 64
    if (tid < 64) {
      const int c0 = tid, c1 = tid + 64;
      const int o0 = c0 < n ? order[c0] : 0, o1 = c1 < n ? order[c1] : 0;
      int rem0 = c0 < n ? 0 : 1, rem1 = c1 < n ? 0 : 1;
      for (int a = 0; a < n; ++a) {
        const int ra = a < 64 ? __shfl(rem0, a) : __shfl(rem1, a - 64);
        if (ra) continue;                                    // wave-uniform
        const int i = order[a];
        if (c0 > a && !rem0 && ioa[i][o0] >= 0.99f) rem0 = 1;
        if (c1 > a && c1 < n && !rem1 && ioa[i][o1] >= 0.99f) rem1 = 1;
      }
      // ordered compaction of the survivors (sorted order)
      const unsigned long long k0 = __ballot(c0 < n && !rem0), k1 = __ballot(c1 < n && !rem1);
      const unsigned long long below = tid == 0 ? 0ull : (~0ull >> (64 - tid));
      const int d0 = __popcll(k0 & below), d1 = __popcll(k0) + __popcll(k1 & below);
      if (c0 < n && !rem0) {
        for (int e = 0; e < 5; ++e) tmpb[d0][e] = bx[o0][e];
        tmps[d0] = sc[o0];
        tmpi[d0] = src[o0];
      }
      if (c1 < n && !rem1) {
        for (int e = 0; e < 5; ++e) tmpb[d1][e] = bx[o1][e];
        tmps[d1] = sc[o1];
        tmpi[d1] = src[o1];
      }
      if (tid == 0) s_n = __popcll(k0) + __popcll(k1);
    }
    __syncthreads();
    {
      const int m = s_n;
      for (int i = tid; i < m; i += PP_THREADS) {
        for (int e = 0; e < 5; ++e) bx[i][e] = tmpb[i][e];
        sc[i] = tmps[i];
        src[i] = tmpi[i];
      }
    }
    __syncthreads();
  }
  __syncthreads();

  // ---- text decode: argmax per (box, step) by all threads (the IoA matrix storage is free now)
  int* chr = reinterpret_cast<int*>(&ioa[0][0]);                  // [n][T]
  float* prb = &ioa[0][0] + PP_KMAX * 32;                         // [n][T], T <= 32
  const int n_fin = s_n;
  if (p.do_text) {
    // 4 lanes per (box, step) row, 64 rows in flight per pass; every lane's loads of a pass are independent
    // (one memory latency per pass); first maximum wins, as torch.max.  (A whole wavefront per row serialises
    // 208 row latencies per wavefront: 3x slower than even one thread per row.)
    constexpr int LPR = 4, CMAX_L = (256 + LPR - 1) / LPR;
    const int sub = tid & (LPR - 1), grp = tid / LPR;
    const int nrows = n_fin * p.T;
    constexpr int G = PP_THREADS / LPR;
    for (int pass = 0; pass * G < nrows; ++pass) {
      const int pr = pass * G + grp;
      const bool live = pr < nrows;                      // every lane stays in the loop for the shuffles
      const int prc = live ? pr : nrows - 1;
      const int i = prc / p.T, t = prc - i * p.T;
      const float* row = p.text + (((long)n_img * p.K + src[i]) * p.T + t) * (long)p.C;
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll 8
      for (int k = 0; k < CMAX_L; ++k) {
        const int c = sub + LPR * k;
        if (c >= p.C) break;
        const float v = row[c];
        if (bi == 0x7fffffff || v > best) { best = v; bi = c; }
      }
#pragma unroll
      for (int off = 1; off < LPR; off <<= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
      }
      if (live && sub == 0) {
        chr[i * p.T + t] = bi;
        prb[i * p.T + t] = best;
      }
    }
  }
  __syncthreads();
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
}

extern "C" int glass_postprocess_words(const float* boxes, const float* scores, const int* counts, const float* text,
                                       const float* scale_xy, int N, int K, int T, int C, const float* thresholds8_host,
                                       int stop_index, float* out_boxes, float* out_scores, float* out_polygons, int* out_src,
                                       int* out_char, float* out_text_score, int* out_text_len, int* out_count,
                                       glass_stream_t stream) {
  if (N == 0) return GLASS_OK;
  GLASS_CHECK_ARG(K >= 0 && K <= PP_KMAX, "glass_postprocess_words: K=%d (max %d)", K, PP_KMAX);
  GLASS_CHECK_ARG(counts && thresholds8_host && out_count, "glass_postprocess_words: null pointer");
  GLASS_CHECK_ARG(K == 0 || (boxes && scores && out_boxes && out_scores && out_polygons && out_src && out_char &&
                             out_text_score && out_text_len), "glass_postprocess_words: null pointer");
  GLASS_CHECK_ARG(!text || (T > 0 && T <= 32 && C > 0), "glass_postprocess_words: text needs 0 < T <= 32, C > 0");
  PPParams p;
  p.boxes = boxes; p.scores = scores; p.counts = counts; p.text = text; p.scale_xy = scale_xy;
  p.N = N; p.K = K; p.T = text ? T : 1; p.C = C;
  p.min_box_dim = thresholds8_host[0]; p.valid_score = thresholds8_host[1]; p.detect_thr = thresholds8_host[2];
  p.merge_ioa = thresholds8_host[3]; p.height_ratio = thresholds8_host[4]; p.max_angle_diff = thresholds8_host[5];
  p.minimal_ioa = thresholds8_host[6]; p.text_thr = thresholds8_host[7];
  p.stop_index = stop_index; p.do_text = text ? 1 : 0;
  p.out_boxes = out_boxes; p.out_scores = out_scores; p.out_poly = out_polygons; p.out_src = out_src; p.out_char = out_char;
  p.out_text_score = out_text_score; p.out_text_len = out_text_len; p.out_count = out_count;
  hipLaunchKernelGGL(postprocess_words_kernel, dim3(N), dim3(PP_THREADS), 0, (hipStream_t)stream, p);
  GLASS_CHECK_LAUNCH("glass_postprocess_words");
  return GLASS_OK;
}
