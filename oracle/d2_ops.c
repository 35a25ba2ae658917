/*
 * ORACLE (test infrastructure, NOT product code): plain-C CPU restatement of the
 * detectron2 v0.6 native ops the GLASS inference path reaches through Python.
 *
 * detectron2 is a third-party dependency of the reference, pinned ==0.6
 * (reference README.md:36; demo/glass_demo.ipynb cell 4) and NOT vendored under
 * /root/reference, so these follow the published v0.6 algorithms [d2-recall]:
 *   - detectron2/layers/csrc/box_iou_rotated/box_iou_rotated_utils.h   (rotated IoU)
 *   - detectron2/layers/csrc/nms_rotated/nms_rotated_cpu.cpp          (greedy NMS, `>=`)
 *   - detectron2/layers/csrc/ROIAlignRotated/ROIAlignRotated_cpu.cpp  (rotated RoIAlign)
 * anchored on the reference's own call sites:
 *   glass/modeling/roi_heads/rotated_fast_rcnn.py:131   (batched_nms_rotated)
 *   glass/modeling/fusion/recognizers_hybrid_head.py:320,550,556 (ROIPooler -> ROIAlignRotated)
 *   glass/structures/boxes.py:33                          (pairwise_iou_rotated)
 *   glass/postprocess/post_processor_rotated_boxes.py:120,181
 * The reference holds no golden vectors for these: parity is pinned by the
 * analytic known-answer tests in tests/test_oracle_d2ops.py instead.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; } pt;

static inline pt pt_sub(pt a, pt b) { pt r = {a.x - b.x, a.y - b.y}; return r; }
static inline pt pt_add(pt a, pt b) { pt r = {a.x + b.x, a.y + b.y}; return r; }
static inline pt pt_mul(pt a, float s) { pt r = {a.x * s, a.y * s}; return r; }
static inline float dot2(pt a, pt b) { return a.x * b.x + a.y * b.y; }
static inline float cross2(pt a, pt b) { return a.x * b.y - b.x * a.y; }

typedef struct { float x_ctr, y_ctr, w, h, a; } rbox;

static void rotated_vertices(const rbox* box, pt pts[4]) {
  double theta = box->a * 0.01745329251;
  float cosTheta2 = (float)cos(theta) * 0.5f;
  float sinTheta2 = (float)sin(theta) * 0.5f;
  pts[0].x = box->x_ctr + sinTheta2 * box->h + cosTheta2 * box->w;
  pts[0].y = box->y_ctr + cosTheta2 * box->h - sinTheta2 * box->w;
  pts[1].x = box->x_ctr - sinTheta2 * box->h + cosTheta2 * box->w;
  pts[1].y = box->y_ctr - cosTheta2 * box->h - sinTheta2 * box->w;
  pts[2].x = 2 * box->x_ctr - pts[0].x;
  pts[2].y = 2 * box->y_ctr - pts[0].y;
  pts[3].x = 2 * box->x_ctr - pts[1].x;
  pts[3].y = 2 * box->y_ctr - pts[1].y;
}

/* exported for the direction-pinning tests: the 4 vertices of (cx, cy, w, h, angle_deg) in d2's order */
void d2o_rotated_vertices(const float* r, float* out8) {
  rbox b = {r[0], r[1], r[2], r[3], r[4]};
  pt p[4];
  rotated_vertices(&b, p);
  for (int i = 0; i < 4; i++) { out8[2 * i] = p[i].x; out8[2 * i + 1] = p[i].y; }
}

static int intersection_points(const pt p1[4], const pt p2[4], pt out[24]) {
  pt v1[4], v2[4];
  for (int i = 0; i < 4; i++) {
    v1[i] = pt_sub(p1[(i + 1) % 4], p1[i]);
    v2[i] = pt_sub(p2[(i + 1) % 4], p2[i]);
  }
  int num = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float det = cross2(v2[j], v1[i]);
      if (fabsf(det) <= 1e-14f) continue;  /* parallel edges */
      pt v12 = pt_sub(p2[j], p1[i]);
      float t1 = cross2(v2[j], v12) / det;
      float t2 = cross2(v1[i], v12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f)
        out[num++] = pt_add(p1[i], pt_mul(v1[i], t1));
    }
  { /* vertices of rect1 inside rect2 */
    pt AB = v2[0], DA = v2[3];
    float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
    for (int i = 0; i < 4; i++) {
      pt AP = pt_sub(p1[i], p2[0]);
      float APdotAB = dot2(AP, AB), APdotAD = -dot2(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD) out[num++] = p1[i];
    }
  }
  { /* vertices of rect2 inside rect1 */
    pt AB = v1[0], DA = v1[3];
    float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
    for (int i = 0; i < 4; i++) {
      pt AP = pt_sub(p2[i], p1[0]);
      float APdotAB = dot2(AP, AB), APdotAD = -dot2(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD) out[num++] = p2[i];
    }
  }
  return num;
}

/* comparator of the CPU build of convex_hull_graham: polar order around q[0],
 * collinear points by distance */
static int hull_less(pt A, pt B) {
  float t = cross2(A, B);
  if (fabsf(t) < 1e-6f) return dot2(A, A) < dot2(B, B);
  return t > 0;
}

static int convex_hull_graham(const pt* p, int n, pt* q) {
  int t = 0;
  for (int i = 1; i < n; i++)
    if (p[i].y < p[t].y || (p[i].y == p[t].y && p[i].x < p[t].x)) t = i;
  pt start = p[t];
  for (int i = 0; i < n; i++) q[i] = pt_sub(p[i], start);
  pt tmp = q[0]; q[0] = q[t]; q[t] = tmp;
  /* insertion sort of q[1..n) (<= 23 elements) with the comparator above */
  for (int i = 2; i < n; i++) {
    pt key = q[i];
    int j = i - 1;
    while (j >= 1 && hull_less(key, q[j])) { q[j + 1] = q[j]; j--; }
    q[j + 1] = key;
  }
  float dist[24];
  for (int i = 0; i < n; i++) dist[i] = dot2(q[i], q[i]);
  int k;
  for (k = 1; k < n; k++) if (dist[k] > 1e-8f) break;
  if (k == n) { q[0] = p[t]; return 1; }
  q[1] = q[k];
  int m = 2;
  for (int i = k + 1; i < n; i++) {
    while (m > 1 && cross2(pt_sub(q[i], q[m - 2]), pt_sub(q[m - 1], q[m - 2])) >= 0) m--;
    q[m++] = q[i];
  }
  return m;  /* shift_to_zero = true: only the area is needed */
}

static float polygon_area(const pt* q, int m) {
  if (m <= 2) return 0;
  float area = 0;
  for (int i = 1; i < m - 1; i++) area += fabsf(cross2(pt_sub(q[i], q[0]), pt_sub(q[i + 1], q[0])));
  return area / 2.0f;
}

static float boxes_intersection(const rbox* b1, const rbox* b2) {
  pt ipts[24], ordered[24], p1[4], p2[4];
  rotated_vertices(b1, p1);
  rotated_vertices(b2, p2);
  int num = intersection_points(p1, p2, ipts);
  if (num <= 2) return 0.0f;
  int nc = convex_hull_graham(ipts, num, ordered);
  return polygon_area(ordered, nc);
}

float d2o_single_box_iou_rotated(const float* r1, const float* r2) {
  rbox b1, b2;
  float csx = (r1[0] + r2[0]) / 2.0f, csy = (r1[1] + r2[1]) / 2.0f;
  b1.x_ctr = r1[0] - csx; b1.y_ctr = r1[1] - csy; b1.w = r1[2]; b1.h = r1[3]; b1.a = r1[4];
  b2.x_ctr = r2[0] - csx; b2.y_ctr = r2[1] - csy; b2.w = r2[2]; b2.h = r2[3]; b2.a = r2[4];
  float area1 = b1.w * b1.h, area2 = b2.w * b2.h;
  if (area1 < 1e-14f || area2 < 1e-14f) return 0.f;
  float inter = boxes_intersection(&b1, &b2);
  return inter / (area1 + area2 - inter);
}

void d2o_pairwise_iou_rotated(const float* b1, int n1, const float* b2, int n2, float* out) {
  for (int i = 0; i < n1; i++)
    for (int j = 0; j < n2; j++) out[(size_t)i * n2 + j] = d2o_single_box_iou_rotated(b1 + 5 * i, b2 + 5 * j);
}

/* greedy NMS, nms_rotated_cpu.cpp semantics: visit in descending-score order (stable),
 * suppress j when iou(i, j) >= thr.  Returns number kept; keep[] holds input indices. */
int d2o_nms_rotated(const float* boxes, const float* scores, int n, float thr, int64_t* keep) {
  if (n == 0) return 0;
  int* order = (int*)malloc(sizeof(int) * n);
  int* tmp = (int*)malloc(sizeof(int) * n);
  unsigned char* sup = (unsigned char*)calloc(n, 1);
  for (int i = 0; i < n; i++) order[i] = i;
  /* stable bottom-up merge sort, descending */
  for (int w = 1; w < n; w *= 2) {
    for (int lo = 0; lo < n; lo += 2 * w) {
      int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      int a = lo, b = mid, k = lo;
      while (a < mid && b < hi) tmp[k++] = (scores[order[b]] > scores[order[a]]) ? order[b++] : order[a++];
      while (a < mid) tmp[k++] = order[a++];
      while (b < hi) tmp[k++] = order[b++];
    }
    memcpy(order, tmp, sizeof(int) * n);
  }
  int nk = 0;
  for (int _i = 0; _i < n; _i++) {
    int i = order[_i];
    if (sup[i]) continue;
    keep[nk++] = i;
    for (int _j = _i + 1; _j < n; _j++) {
      int j = order[_j];
      if (sup[j]) continue;
      if (d2o_single_box_iou_rotated(boxes + 5 * i, boxes + 5 * j) >= thr) sup[j] = 1;
    }
  }
  free(order); free(tmp); free(sup);
  return nk;
}

/* ROIAlignRotated forward, NCHW input, rois = (batch_idx, cx, cy, w, h, angle_deg) */
void d2o_roi_align_rotated(const float* input, int N, int C, int H, int W, const float* rois, int R,
                           float spatial_scale, int PH, int PW, int sampling_ratio, float* out) {
  (void)N;
  for (int n = 0; n < R; n++) {
    const float* roi = rois + 6 * n;
    int b = (int)roi[0];
    float offset = 0.5f;
    float cw = roi[1] * spatial_scale - offset;
    float ch = roi[2] * spatial_scale - offset;
    float rw = roi[3] * spatial_scale;
    float rh = roi[4] * spatial_scale;
    float theta = (float)(roi[5] * M_PI / 180.0);
    float cos_t = cosf(theta), sin_t = sinf(theta);
    float bin_h = rh / (float)PH, bin_w = rw / (float)PW;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / PH);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / PW);
    float count = (float)(gh * gw > 1 ? gh * gw : 1);
    float start_h = -rh / 2.0f, start_w = -rw / 2.0f;
    for (int c = 0; c < C; c++) {
      const float* src = input + ((size_t)b * C + c) * H * W;
      float* dst = out + ((size_t)n * C + c) * PH * PW;
      for (int ph = 0; ph < PH; ph++)
        for (int pw = 0; pw < PW; pw++) {
          float acc = 0;
          for (int iy = 0; iy < gh; iy++) {
            float yy = start_h + ph * bin_h + (iy + .5f) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ix++) {
              float xx = start_w + pw * bin_w + (ix + .5f) * bin_w / (float)gw;
              float y = yy * cos_t - xx * sin_t + ch;
              float x = yy * sin_t + xx * cos_t + cw;
              if (y < -1.0f || y > H || x < -1.0f || x > W) continue;
              if (y < 0) y = 0;
              if (x < 0) x = 0;
              int yl = (int)y, xl = (int)x, yh, xh;
              if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
              if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
              float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
              acc += hy * hx * src[yl * W + xl] + hy * lx * src[yl * W + xh] +
                     ly * hx * src[yh * W + xl] + ly * lx * src[yh * W + xh];
            }
          }
          dst[ph * PW + pw] = acc / count;
        }
    }
  }
}
