// Rotated-box IoU on device: restatement of detectron2 v0.6 layers/csrc/box_iou_rotated/box_iou_rotated_utils.h
// (CPU variant of the convex-hull sort), shared by the NMS, pairwise-IoU and post-processing kernels.
// Contraction is OFF in every function of this file (`#pragma clang fp contract(off)`): d2's CPU op is compiled for baseline
// x86-64, i.e. every product and sum of the cross / dot products is rounded separately; with hipcc's default (fused
// multiply-add wherever it can) the near-degenerate cases - collinear edges, |det| ~ 1e-14 - took the other branch of the
// algorithm's epsilon tests about twice as often as the CPU code does (tests/test_gpu_d_known_answers.py: shared-edge family).
#pragma once
#include "common.h"

struct Pt { float x, y; };
__device__ __forceinline__ Pt psub(Pt a, Pt b) { return Pt{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ float dot2(Pt a, Pt b) {
#pragma clang fp contract(off)
  return a.x * b.x + a.y * b.y;
}
__device__ __forceinline__ float cross2(Pt a, Pt b) {
#pragma clang fp contract(off)
  return a.x * b.y - b.x * a.y;
}

// box = (cx, cy, w, h) + half-extent cos/sin (cos(theta)*0.5, sin(theta)*0.5 computed in double like d2)
struct RBox { float cx, cy, w, h, c2, s2; };

__device__ __forceinline__ void rbox_vertices(float cx, float cy, const RBox& b, Pt* pts) {
#pragma clang fp contract(off)
  pts[0].x = cx + b.s2 * b.h + b.c2 * b.w;
  pts[0].y = cy + b.c2 * b.h - b.s2 * b.w;
  pts[1].x = cx - b.s2 * b.h + b.c2 * b.w;
  pts[1].y = cy - b.c2 * b.h - b.s2 * b.w;
  pts[2].x = 2 * cx - pts[0].x;
  pts[2].y = 2 * cy - pts[0].y;
  pts[3].x = 2 * cx - pts[1].x;
  pts[3].y = 2 * cy - pts[1].y;
}

__device__ __forceinline__ bool hull_less(Pt A, Pt B) {
  const float t = cross2(A, B);
  if (fabsf(t) < 1e-6f) return dot2(A, A) < dot2(B, B);
  return t > 0;
}

// The algorithm's point list (up to 24 entries, indexed dynamically).  PrivPts keeps it in private memory (scratch: every
// access is a memory round trip); a kernel with LDS to spare passes its own type with the same operator[] (postprocess.hip:
// one column per thread) - the arithmetic, and so the result, is the same bit for bit.
struct PrivPts {
  Pt a[24];
  __device__ __forceinline__ Pt& operator[](int i) { return a[i]; }
};

template <class PtArr>
__device__ inline float rotated_iou_in(const RBox& r1, const RBox& r2, PtArr& ip) {
#pragma clang fp contract(off)
  const float area1 = r1.w * r1.h, area2 = r2.w * r2.h;
  if (area1 < 1e-14f || area2 < 1e-14f) return 0.f;
  {
    // circumscribed circles disjoint (with slack) -> no intersection point exists -> the algorithm below returns
    // exactly 0; skipping it matters because its point lists are indexed dynamically (400 B of scratch per lane)
    // and almost every pair in NMS / the word post-processor is far apart
    const float dx = r1.cx - r2.cx, dy = r1.cy - r2.cy;
    const float rr = 0.5f * (sqrtf(r1.w * r1.w + r1.h * r1.h) + sqrtf(r2.w * r2.w + r2.h * r2.h));
    if (dx * dx + dy * dy > rr * rr * 1.001f + 1e-2f) return 0.f;
  }
  const float sx = (r1.cx + r2.cx) / 2.0f, sy = (r1.cy + r2.cy) / 2.0f;
  Pt p1[4], p2[4];
  rbox_vertices(r1.cx - sx, r1.cy - sy, r1, p1);
  rbox_vertices(r2.cx - sx, r2.cy - sy, r2, p2);
  Pt v1[4], v2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v1[i] = psub(p1[(i + 1) & 3], p1[i]);
    v2[i] = psub(p2[(i + 1) & 3], p2[i]);
  }
  int num = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float det = cross2(v2[j], v1[i]);
      if (fabsf(det) <= 1e-14f) continue;
      const Pt v12 = psub(p2[j], p1[i]);
      const float t1 = cross2(v2[j], v12) / det;
      const float t2 = cross2(v1[i], v12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        ip[num].x = p1[i].x + v1[i].x * t1;
        ip[num].y = p1[i].y + v1[i].y * t1;
        ++num;
      }
    }
  {
    const Pt AB = v2[0], DA = v2[3];
    const float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const Pt AP = psub(p1[i], p2[0]);
      const float APdotAB = dot2(AP, AB), APdotAD = -dot2(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD) ip[num++] = p1[i];
    }
  }
  {
    const Pt AB = v1[0], DA = v1[3];
    const float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const Pt AP = psub(p2[i], p1[0]);
      const float APdotAB = dot2(AP, AB), APdotAD = -dot2(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD) ip[num++] = p2[i];
    }
  }
  if (num <= 2) return 0.f;
  // Graham scan (CPU variant of d2's convex_hull_graham, shift_to_zero = true)
  int t = 0;
  for (int i = 1; i < num; ++i)
    if (ip[i].y < ip[t].y || (ip[i].y == ip[t].y && ip[i].x < ip[t].x)) t = i;
  const Pt start = ip[t];
  PtArr& q = ip;                                   // (shifted in place)
  for (int i = 0; i < num; ++i) q[i] = psub(ip[i], start);
  { const Pt tmp = q[0]; q[0] = q[t]; q[t] = tmp; }
  for (int i = 2; i < num; ++i) {
    const Pt key = q[i];
    int j = i - 1;
    while (j >= 1 && hull_less(key, q[j])) { q[j + 1] = q[j]; --j; }
    q[j + 1] = key;
  }
  int k;
  for (k = 1; k < num; ++k)
    if (dot2(q[k], q[k]) > 1e-8f) break;
  float inter = 0.f;
  if (k < num) {
    q[1] = q[k];
    int m = 2;
    for (int i = k + 1; i < num; ++i) {
      while (m > 1 && cross2(psub(q[i], q[m - 2]), psub(q[m - 1], q[m - 2])) >= 0) --m;
      q[m++] = q[i];
    }
    if (m > 2) {
      float area = 0.f;
      const Pt q0 = q[0];
      for (int i = 1; i < m - 1; ++i) area += fabsf(cross2(psub(q[i], q0), psub(q[i + 1], q0)));
      inter = area / 2.0f;
    }
  }
  return inter / (area1 + area2 - inter);
}

__device__ inline float rotated_iou(const RBox& r1, const RBox& r2) {
  PrivPts pts;
  return rotated_iou_in(r1, r2, pts);
}

__device__ __forceinline__ RBox make_rbox(float cx, float cy, float w, float h, float a) {
  const double theta = (double)a * 0.01745329251;
  RBox b;
  b.cx = cx; b.cy = cy; b.w = w; b.h = h;
  b.c2 = (float)cos(theta) * 0.5f;
  b.s2 = (float)sin(theta) * 0.5f;
  return b;
}

