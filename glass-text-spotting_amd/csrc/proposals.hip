// Rotated-box proposal machinery: per-level top-k + anchor decode, rotated IoU, blocked greedy
// NMS, box-head decode.  These stages are latency-bound integer/compare work, not GEMMs:
// one workgroup (1024 threads = 16 wavefronts) per image keeps the whole candidate list in
// LDS, uses LDS radix histograms for the top-k select and 64-wide ballots' worth of
// candidates per NMS chunk (one wavefront-width bitmask row per candidate).
#include "common.h"

#include "proposal_common.h"

#include "rotated_iou.h"

// pairwise IoU matrix (post-processing's pairwise_iou_rotated, glass/structures/boxes.py:33)
__global__ void pairwise_iou_kernel(const float* b1, int n1, const float* b2, int n2, float* out) {
  const long total = (long)n1 * n2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int i = (int)(idx / n2), j = (int)(idx % n2);
    const float* x = b1 + 5 * i;
    const float* y = b2 + 5 * j;
    out[idx] = rotated_iou(make_rbox(x[0], x[1], x[2], x[3], x[4]), make_rbox(y[0], y[1], y[2], y[3], y[4]));
  }
}

extern "C" int glass_pairwise_iou_rotated(const float* boxes1, int n1, const float* boxes2, int n2, float* out,
                                          glass_stream_t stream) {
  if (n1 == 0 || n2 == 0) return GLASS_OK;
  GLASS_CHECK_ARG(boxes1 && boxes2 && out, "glass_pairwise_iou_rotated: null pointer");
  const long total = (long)n1 * n2;
  long g = (total + 127) / 128;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pairwise_iou_kernel, dim3((unsigned)g), dim3(128), 0, (hipStream_t)stream, boxes1, n1, boxes2, n2, out);
  GLASS_CHECK_LAUNCH("glass_pairwise_iou_rotated");
  return GLASS_OK;
}

// ------------------------------------------------------------------ filter + sort + greedy NMS
constexpr int NMS_SMAX = 8192;
constexpr int NMS_KEEP_MAX = 1024;
constexpr int NMS_THREADS = 1024;

struct NmsParams {
  const float* boxes; const float* scores; const int* cat; const int* valid_count; const int* image_hw;
  int N, S;
  float score_thresh, nms_thresh;
  int post_topk, flags;
  float* out_boxes; float* out_scores; int* out_index; int* out_count;
};

// the early exit of rotated_iou (rotated_iou.h: circumscribed circles disjoint -> IoU exactly 0), as a predicate of its own
__device__ __forceinline__ bool rbox_far_apart(const RBox& a, const RBox& b) {
#pragma clang fp contract(off)
  const float dx = a.cx - b.cx, dy = a.cy - b.cy;
  const float rr = 0.5f * (sqrtf(a.w * a.w + a.h * a.h) + sqrtf(b.w * b.w + b.h * b.h));
  return dx * dx + dy * dy > rr * rr * 1.001f + 1e-2f;
}

// `total` box pairs, pair pr -> (i, j) by `decode`; the pairs that are not far apart - the ones that run the polygon clipping,
// whose point lists live in scratch - are collected per wavefront (ballot compaction, 128 entries each) and evaluated 64 at a
// time with every lane busy, instead of by whichever lanes of a round happen to hold one (postprocess.hip: pp_for_pairs).
template <class Decode, class Far, class Heavy>
__device__ __forceinline__ void nms_for_pairs(int total, unsigned short* queue, Decode decode, Far far, Heavy heavy) {
  const int lane = threadIdx.x & 63;
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int qn = 0;                                                       // (wavefront-uniform)
  for (int base = threadIdx.x & ~63; base < total; base += NMS_THREADS) {
    const int pr = base + lane;
    int i = 0, j = 0;
    bool near = false;
    if (pr < total) {
      decode(pr, i, j);
      near = !far(i, j);
    }
    const unsigned long long m = __ballot(near);
    if (near) queue[qn + __popcll(m & below)] = (unsigned short)((i << 6) | j);
    qn += __popcll(m);
    if (qn >= 64) {
      qn -= 64;
      const int e = queue[qn + lane];
      heavy(e >> 6, e & 63);
    }
  }
  if (lane < qn) {
    const int e = queue[lane];
    heavy(e >> 6, e & 63);
  }
}

__global__ __launch_bounds__(NMS_THREADS) void nms_select_kernel(NmsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  // carve: sorted composites [npad] u64 | kept boxes | chunk boxes | masks
  const int n = blockIdx.x;
  const int S = p.S;
  int npad = 1;
  while (npad < S) npad <<= 1;
  u64* order = reinterpret_cast<u64*>(dyn_smem);
  RBox* kept = reinterpret_cast<RBox*>(order + npad);
  RBox* chunk = kept + NMS_KEEP_MAX;
  u64* cmask = reinterpret_cast<u64*>(chunk + 64);
  int* csup = reinterpret_cast<int*>(cmask + 64);
  int* cslot = csup + 64;
  __shared__ float red_max[NMS_THREADS / 64], red_min[NMS_THREADS / 64];
  __shared__ float s_span;
  __shared__ int s_nvalid, s_nkept, s_stop;
  __shared__ float cbox[64][5], cscore[64];      // the chunk's (clipped) boxes and scores as they go to the outputs
  __shared__ int ckept[64];                      // output slot of a chunk candidate, -1: suppressed
  __shared__ unsigned short pair_queue[NMS_THREADS / 64][128];

  const float* boxes = p.boxes + (long)n * S * 5;
  const float* scores = p.scores + (long)n * S;
  const int* cat = p.cat ? p.cat + (long)n * S : nullptr;
  const int cnt = p.valid_count ? min(p.valid_count[n], S) : S;
  const float img_h = (float)p.image_hw[2 * n], img_w = (float)p.image_hw[2 * n + 1];

  // clipped box of a slot (recomputed where needed; 5 floats, cheap)
  auto load_box = [&](int s, float* b) {
    b[0] = boxes[5 * s]; b[1] = boxes[5 * s + 1]; b[2] = boxes[5 * s + 2]; b[3] = boxes[5 * s + 3]; b[4] = boxes[5 * s + 4];
    if (p.flags & GLASS_NMS_CLIP) {
      // RotatedBoxes.clip: normalise angle, then clip only nearly-horizontal boxes
      b[4] = floor_mod(b[4] + 180.0f, 360.0f) - 180.0f;
      if (fabsf(b[4]) <= 1.0f) {
        float x1 = b[0] - b[2] / 2.0f, y1 = b[1] - b[3] / 2.0f, x2 = b[0] + b[2] / 2.0f, y2 = b[1] + b[3] / 2.0f;
        x1 = fminf(fmaxf(x1, 0.f), img_w); x2 = fminf(fmaxf(x2, 0.f), img_w);
        y1 = fminf(fmaxf(y1, 0.f), img_h); y2 = fminf(fmaxf(y2, 0.f), img_h);
        b[0] = (x1 + x2) / 2.0f;
        b[1] = (y1 + y2) / 2.0f;
        b[2] = fminf(b[2], x2 - x1);
        b[3] = fminf(b[3], y2 - y1);
      }
    }
  };

  // 1. validity, sort keys, coordinate span for the category offsets
  float lmax = -INFINITY, lmin = INFINITY;
  for (int s = threadIdx.x; s < npad; s += blockDim.x) {
    u64 c = 0;
    if (s < cnt) {
      const float sc = scores[s];
      bool ok = isfinite(sc) && isfinite(boxes[5 * s]) && isfinite(boxes[5 * s + 1]) && isfinite(boxes[5 * s + 2]) &&
                isfinite(boxes[5 * s + 3]) && isfinite(boxes[5 * s + 4]);
      if (ok) {
        float b[5];
        load_box(s, b);
        if ((p.flags & GLASS_NMS_DROP_EMPTY) && !(b[2] > 0.f && b[3] > 0.f)) ok = false;
        if (!(sc > p.score_thresh)) ok = false;
        if (ok) {
          c = ((u64)float_key(sc) << 32) | (u64)(0xffffffffu - (u32)s);
          const float ext = fmaxf(b[2], b[3]) / 2.0f;
          lmax = fmaxf(lmax, fmaxf(b[0], b[1]) + ext);
          lmin = fminf(lmin, fminf(b[0], b[1]) - ext);
        }
      }
    }
    order[s] = c;
  }
  for (int off = 32; off > 0; off >>= 1) {
    lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    lmin = fminf(lmin, __shfl_xor(lmin, off));
  }
  if ((threadIdx.x & 63) == 0) { red_max[threadIdx.x >> 6] = lmax; red_min[threadIdx.x >> 6] = lmin; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float mx = -INFINITY, mn = INFINITY;
    for (int i = 0; i < NMS_THREADS / 64; ++i) { mx = fmaxf(mx, red_max[i]); mn = fminf(mn, red_min[i]); }
    s_span = mx - mn + 1.0f;
    s_nkept = 0;
    s_stop = 0;
  }
  __syncthreads();
  bitonic_sort_desc(order, npad);
  // number of valid = first zero composite (valid composites are never 0: key has the top bit
  // set for non-negative scores, and for negative scores the index part is non-zero unless
  // s = 2^32-1)
  if (threadIdx.x == 0) {
    int lo = 0, hi = npad;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (order[mid] != 0) lo = mid + 1; else hi = mid; }
    s_nvalid = lo;
  }
  __syncthreads();
  const int nvalid = s_nvalid;
  const float span = s_span;
  const int keep_cap = p.post_topk < NMS_KEEP_MAX ? p.post_topk : NMS_KEEP_MAX;

  // 2. blocked greedy NMS over chunks of 64 sorted candidates
  for (int c0 = 0; c0 < nvalid; c0 += 64) {
    const int csize = min(64, nvalid - c0);
    if (threadIdx.x < 64) {
      cmask[threadIdx.x] = 0;
      csup[threadIdx.x] = 0;
      if ((int)threadIdx.x < csize) {
        const int s = (int)(0xffffffffu - (u32)(order[c0 + threadIdx.x] & 0xffffffffu));
        float b[5];
        load_box(s, b);
        const float off = cat ? (float)cat[s] * span : 0.f;
        chunk[threadIdx.x] = make_rbox(b[0] + off, b[1] + off, b[2], b[3], b[4]);
        cslot[threadIdx.x] = s;
#pragma unroll
        for (int e = 0; e < 5; ++e) cbox[threadIdx.x][e] = b[e];
        cscore[threadIdx.x] = scores[s];
      }
    }
    __syncthreads();
    const int nk = s_nkept;
    unsigned short* queue = pair_queue[threadIdx.x >> 6];
    // (a) candidates vs already kept boxes (i: kept box < 1024, j: candidate of the chunk)
    nms_for_pairs(csize * nk, queue, [&](int pr, int& i, int& j) { j = pr / nk; i = pr - j * nk; },
                  [&](int i, int j) { return rbox_far_apart(kept[i], chunk[j]); },
                  [&](int i, int j) { if (rotated_iou(kept[i], chunk[j]) >= p.nms_thresh) csup[j] = 1; });
    // (b) intra-chunk pairs i < j
    nms_for_pairs(csize * csize, queue, [&](int pr, int& i, int& j) { i = pr / csize; j = pr - i * csize; },
                  [&](int i, int j) { return i >= j || rbox_far_apart(chunk[i], chunk[j]); },
                  [&](int i, int j) { if (rotated_iou(chunk[i], chunk[j]) >= p.nms_thresh) atomicOr(&cmask[i], 1ull << j); });
    __syncthreads();
    // (c) serial resolve inside the chunk: LDS only (the survivors' output rows - which used to be re-read from global memory
    // inside this one-thread loop, a dependent round trip per kept box - are written by 64 threads afterwards)
    if (threadIdx.x == 0) {
      u64 sup = 0;
      int kcount = nk;
      for (int i = 0; i < csize; ++i) {
        ckept[i] = -1;
        if (csup[i] || ((sup >> i) & 1ull)) continue;
        if (kcount >= keep_cap) { s_stop = 1; for (int j = i + 1; j < csize; ++j) ckept[j] = -1; break; }
        kept[kcount] = chunk[i];
        ckept[i] = kcount;
        ++kcount;
        sup |= cmask[i];
      }
      if (kcount >= keep_cap) s_stop = 1;
      s_nkept = kcount;
    }
    __syncthreads();
    if ((int)threadIdx.x < csize && ckept[threadIdx.x] >= 0) {
      const long o = (long)n * p.post_topk + ckept[threadIdx.x];
#pragma unroll
      for (int e = 0; e < 5; ++e) p.out_boxes[o * 5 + e] = cbox[threadIdx.x][e];
      p.out_scores[o] = cscore[threadIdx.x];
      p.out_index[o] = cslot[threadIdx.x];
    }
    if (s_stop) break;
  }
  if (threadIdx.x == 0) p.out_count[n] = s_nkept;
}

extern "C" int glass_rotated_nms_select(const float* boxes, const float* scores, const int* cat, const int* valid_count,
                                        int N, int S, const int* image_hw, float score_thresh, float nms_thresh,
                                        int post_topk, int flags, float* out_boxes, float* out_scores, int* out_index,
                                        int* out_count, glass_stream_t stream) {
  GLASS_CHECK_ARG(out_boxes && out_scores && out_index && out_count && image_hw, "glass_rotated_nms_select: null pointer");
  GLASS_CHECK_ARG(S >= 0 && S <= NMS_SMAX, "glass_rotated_nms_select: S=%d (max %d)", S, NMS_SMAX);
  GLASS_CHECK_ARG(post_topk > 0 && post_topk <= NMS_KEEP_MAX, "glass_rotated_nms_select: post_topk=%d (max %d)", post_topk,
                  NMS_KEEP_MAX);
  if (N == 0) return GLASS_OK;
  if (S == 0) {
    hipError_t e = hipMemsetAsync(out_count, 0, sizeof(int) * N, (hipStream_t)stream);
    if (e != hipSuccess) { glass_set_error("glass_rotated_nms_select: memset: %s", hipGetErrorString(e)); return GLASS_EHIP; }
    return GLASS_OK;
  }
  GLASS_CHECK_ARG(boxes && scores, "glass_rotated_nms_select: null boxes");
  int npad = 1;
  while (npad < S) npad <<= 1;
  const size_t smem = sizeof(u64) * npad + sizeof(RBox) * (NMS_KEEP_MAX + 64) + sizeof(u64) * 64 + sizeof(int) * 128;
  NmsParams p;
  p.boxes = boxes; p.scores = scores; p.cat = cat; p.valid_count = valid_count; p.image_hw = image_hw; p.N = N; p.S = S;
  p.score_thresh = score_thresh; p.nms_thresh = nms_thresh; p.post_topk = post_topk; p.flags = flags;
  p.out_boxes = out_boxes; p.out_scores = out_scores; p.out_index = out_index; p.out_count = out_count;
  static const int attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_select_kernel),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);      // S <= 8192: 64 KB of sort keys + 26 KB of boxes + masks (the kernel's static LDS is ~6.5 KB)
  if (attr_rc != 0) {
    glass_set_error("glass_rotated_nms_select: cannot raise the dynamic LDS limit (hip error %d)", attr_rc);
    return GLASS_EHIP;
  }
  hipLaunchKernelGGL(nms_select_kernel, dim3(N), dim3(NMS_THREADS), smem, (hipStream_t)stream, p);
  GLASS_CHECK_LAUNCH("glass_rotated_nms_select");
  return GLASS_OK;
}

// ------------------------------------------------------------------ box-head decode
__global__ void box_decode_kernel(const float* cls, const float* deltas, const float* orient, const float* props, int R,
                                  float wx, float wy, float ww, float wh, float wa, float* out_boxes, float* out_fg,
                                  float* out_orient) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  // softmax over (fg, bg) logits; fg is column 0 (rotated_fast_rcnn.py:109 drops the LAST column)
  const float l0 = cls[2 * r], l1 = cls[2 * r + 1];
  const float m = fmaxf(l0, l1);
  const float e0 = expf(l0 - m), e1 = expf(l1 - m);
  out_fg[r] = e0 / (e0 + e1);
  const float* d = deltas + 5 * r;
  const float* b = props + 5 * r;
  const float dx = d[0] / wx, dy = d[1] / wy;
  float dw = d[2] / ww, dh = d[3] / wh;
  const float da = d[4] / wa;
  dw = fminf(dw, SCALE_CLAMP);
  dh = fminf(dh, SCALE_CLAMP);
  float* ob = out_boxes + 5 * r;
  ob[0] = dx * b[2] + b[0];
  ob[1] = dy * b[3] + b[1];
  ob[2] = expf(dw) * b[2];
  ob[3] = expf(dh) * b[3];
  const float pa = da * 180.0f / 3.14159265358979323846f + b[4];
  ob[4] = floor_mod(pa + 180.0f, 360.0f) - 180.0f;
  // orientation softmax over 4 logits -> (argmax, max prob); first max wins like torch.max
  const float* o = orient + 4 * r;
  float om = fmaxf(fmaxf(o[0], o[1]), fmaxf(o[2], o[3]));
  float e[4], sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { e[i] = expf(o[i] - om); sum += e[i]; }
  int am = 0;
  float pm = e[0] / sum;
#pragma unroll
  for (int i = 1; i < 4; ++i) { const float pi = e[i] / sum; if (pi > pm) { pm = pi; am = i; } }
  out_orient[2 * r] = (float)am;
  out_orient[2 * r + 1] = pm;
}

extern "C" int glass_box_decode(const float* cls_logits, const float* deltas, const float* orient_logits,
                                const float* proposals, int R, const float* weights5_host, float* out_boxes,
                                float* out_fg_prob, float* out_orient2, glass_stream_t stream) {
  if (R == 0) return GLASS_OK;
  GLASS_CHECK_ARG(cls_logits && deltas && orient_logits && proposals && weights5_host && out_boxes && out_fg_prob && out_orient2,
                  "glass_box_decode: null pointer");
  hipLaunchKernelGGL(box_decode_kernel, dim3(cdiv(R, 128)), dim3(128), 0, (hipStream_t)stream, cls_logits, deltas,
                     orient_logits, proposals, R, weights5_host[0], weights5_host[1], weights5_host[2], weights5_host[3],
                     weights5_host[4], out_boxes, out_fg_prob, out_orient2);
  GLASS_CHECK_LAUNCH("glass_box_decode");
  return GLASS_OK;
}
