// Batched tail of the meta-architecture: GlassRCNN._postprocess for every image of the step in ONE
// launch (small-box filter -> RotatedBoxes.scale -> clip -> drop empty), with ordered compaction of
// all per-detection fields (boxes, scores, orientations, the [T,C] text-probability rows), so the host
// only reads N counts and slices views — instead of ~60 tiny elementwise launches per image.
// HBM-bound copy work: the text rows (T*C floats each) are moved with float4-free scalar coalesced
// loops (C = 97 is odd).  DET_SPLIT workgroups per image: each repeats the (cheap) keep decisions; workgroup 0 of the image
// writes the compacted boxes / scores / orientations / count, and the text rows - 10 KB each, 32-100 per image, the bulk of
// the bytes - are dealt round-robin to all of them (one workgroup per image copied them one row at a time: 80 of its 90 us).
#include "common.h"

struct DetParams {
  const float* boxes; const float* scores; const float* orient; const float* text;
  const int* counts; const int* roi_start; const float* scale_xy; const int* out_hw;
  int N, K, TC;
  float min_box_dim;
  int do_filter_small;
  float* out_boxes; float* out_scores; float* out_orient; float* out_text; int* out_count;
};

__device__ __forceinline__ float floor_mod_f(float a, float b) {
  float m = fmodf(a, b);
  if (m != 0.f && ((b < 0.f) != (m < 0.f))) m += b;
  return m;
}

constexpr int DET_SPLIT = 8;

__global__ __launch_bounds__(256) void detections_finalize_kernel(DetParams p) {
  __shared__ int keep_src[1024];     // compacted source slot of each output slot
  __shared__ int s_cnt;
  __shared__ int wave_cnt[4];
  const int n = blockIdx.x;
  const bool lead = blockIdx.y == 0;             // the workgroup of the image that writes everything but its share of the text rows
  const int cnt = min(p.counts[n], p.K);
  const float sx = p.scale_xy[2 * n], sy = p.scale_xy[2 * n + 1];
  const float out_h = (float)p.out_hw[2 * n], out_w = (float)p.out_hw[2 * n + 1];
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  // pass 1: per-slot keep decision + new box (written to out_boxes at its compacted position later)
  for (int base = 0; base < cnt; base += 256) {
    const int j = base + threadIdx.x;
    bool keep = false;
    float b[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (j < cnt) {
      const float* s = p.boxes + ((long)n * p.K + j) * 5;
      b[0] = s[0]; b[1] = s[1]; b[2] = s[2]; b[3] = s[3]; b[4] = s[4];
      keep = !p.do_filter_small || (fminf(b[2], b[3]) >= p.min_box_dim);       // filter_small_boxes
      // RotatedBoxes.scale(sx, sy)
      b[0] *= sx;
      b[1] *= sy;
      const float theta = b[4] * 3.14159265358979323846f / 180.0f;
      float sn, cs;
      sincosf(theta, &sn, &cs);
      b[2] *= sqrtf((sx * cs) * (sx * cs) + (sy * sn) * (sy * sn));
      b[3] *= sqrtf((sx * sn) * (sx * sn) + (sy * cs) * (sy * cs));
      b[4] = atan2f(sx * sn, sy * cs) * 180.0f / 3.14159265358979323846f;
      // RotatedBoxes.clip(out size): normalise angle; clip only |angle| <= 1 deg
      b[4] = floor_mod_f(b[4] + 180.0f, 360.0f) - 180.0f;
      if (fabsf(b[4]) <= 1.0f) {
        float x1 = b[0] - b[2] / 2.0f, y1 = b[1] - b[3] / 2.0f, x2 = b[0] + b[2] / 2.0f, y2 = b[1] + b[3] / 2.0f;
        x1 = fminf(fmaxf(x1, 0.f), out_w); x2 = fminf(fmaxf(x2, 0.f), out_w);
        y1 = fminf(fmaxf(y1, 0.f), out_h); y2 = fminf(fmaxf(y2, 0.f), out_h);
        b[0] = (x1 + x2) / 2.0f;
        b[1] = (y1 + y2) / 2.0f;
        b[2] = fminf(b[2], x2 - x1);
        b[3] = fminf(b[3], y2 - y1);
      }
      keep = keep && (b[2] > 0.f) && (b[3] > 0.f);                               // nonempty()
    }
    // ordered compaction across the 256 threads (4 wavefronts)
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = s_cnt;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (keep) {
      const int dst = off + before;
      keep_src[dst] = j;
      if (lead) {
        float* o = p.out_boxes + ((long)n * p.K + dst) * 5;
        o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3]; o[4] = b[4];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_cnt += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  const int kept = s_cnt;
  if (lead && threadIdx.x == 0) p.out_count[n] = kept;
  for (int d = threadIdx.x; lead && d < kept; d += 256) {
    const int j = keep_src[d];
    p.out_scores[(long)n * p.K + d] = p.scores[(long)n * p.K + j];
    if (p.orient) {
      p.out_orient[((long)n * p.K + d) * 2] = p.orient[((long)n * p.K + j) * 2];
      p.out_orient[((long)n * p.K + d) * 2 + 1] = p.orient[((long)n * p.K + j) * 2 + 1];
    }
  }
  if (p.text) {
    const long r0 = p.roi_start[n];
    for (int d = blockIdx.y; d < kept; d += DET_SPLIT) {
      const float* src = p.text + (r0 + keep_src[d]) * (long)p.TC;
      float* dst = p.out_text + ((long)n * p.K + d) * p.TC;
      for (int i = threadIdx.x; i < p.TC; i += 256) dst[i] = src[i];
    }
  }
}

extern "C" int glass_detections_finalize(const float* boxes, const float* scores, const float* orient, const float* text,
                                         const int* counts, const int* roi_start, const float* scale_xy, const int* out_hw,
                                         int N, int K, int TC, float min_box_dim, int do_filter_small, float* out_boxes,
                                         float* out_scores, float* out_orient, float* out_text, int* out_count,
                                         glass_stream_t stream) {
  if (N == 0) return GLASS_OK;
  GLASS_CHECK_ARG(K >= 0 && K <= 1024, "glass_detections_finalize: K=%d (max 1024)", K);
  GLASS_CHECK_ARG(counts && scale_xy && out_hw && out_count, "glass_detections_finalize: null pointer");
  GLASS_CHECK_ARG(K == 0 || (boxes && scores && out_boxes && out_scores), "glass_detections_finalize: null boxes");
  GLASS_CHECK_ARG(!orient || out_orient, "glass_detections_finalize: orient without out_orient");
  GLASS_CHECK_ARG(!text || (out_text && roi_start && TC > 0), "glass_detections_finalize: text without out_text/roi_start");
  DetParams p;
  p.boxes = boxes; p.scores = scores; p.orient = orient; p.text = text; p.counts = counts; p.roi_start = roi_start;
  p.scale_xy = scale_xy; p.out_hw = out_hw; p.N = N; p.K = K; p.TC = TC; p.min_box_dim = min_box_dim;
  p.do_filter_small = do_filter_small; p.out_boxes = out_boxes; p.out_scores = out_scores; p.out_orient = out_orient;
  p.out_text = out_text; p.out_count = out_count;
  hipLaunchKernelGGL(detections_finalize_kernel, dim3(N, text ? DET_SPLIT : 1), dim3(256), 0, (hipStream_t)stream, p);
  GLASS_CHECK_LAUNCH("glass_detections_finalize");
  return GLASS_OK;
}

// ---------------------------------------------------------------------------------------------- word records
// The fixed-size per-image record the ranks exchange (glass_amd/distributed.py words_record_size; replaces the reference's
// pickled comm.gather, glass/evaluation/text_evaluator.py:246-249) straight from the padded outputs of
// glass_postprocess_words:  [count | boxes 5D | score D | text score D | polygon 8D | text length D | character D*T], all
// float32, rows beyond min(count, K, D) and steps beyond min(Tw, T) zero.  One launch instead of a fill + eight strided
// copies and casts.
__global__ __launch_bounds__(256) void pack_word_records_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                                const float* __restrict__ tscore, const float* __restrict__ polys,
                                                                const int* __restrict__ tlen, const int* __restrict__ chars,
                                                                const int* __restrict__ count, int K, int Tw, int D, int T,
                                                                float* __restrict__ rec) {
  const int n = blockIdx.x;
  const int L = 1 + D * (5 + 1 + 1 + 8 + 1 + T);
  const int c = min(count[n], D);          // what the record says ...
  const int k = min(K, D);                 // ... and the rows the inputs really hold
  float* r = rec + (long)n * L;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    float v = 0.f;
    if (i == 0) v = (float)c;
    else {
      int o = i - 1;
      if (o < 5 * D) { const int d = o / 5; if (d < k) v = boxes[((long)n * K + d) * 5 + o % 5]; }
      else if ((o -= 5 * D) < D) { if (o < k) v = scores[(long)n * K + o]; }
      else if ((o -= D) < D) { if (o < k) v = tscore[(long)n * K + o]; }
      else if ((o -= D) < 8 * D) { const int d = o / 8; if (d < k) v = polys[((long)n * K + d) * 8 + o % 8]; }
      else if ((o -= 8 * D) < D) { if (o < k) v = (float)tlen[(long)n * K + o]; }
      else { o -= D; const int d = o / T, t = o % T; if (d < k && t < Tw) v = (float)chars[((long)n * K + d) * Tw + t]; }
    }
    r[i] = v;
  }
}

extern "C" int glass_pack_word_records(const float* boxes, const float* scores, const float* text_score, const float* polygons,
                                       const int* text_len, const int* chars, const int* count, int N, int K, int Tw, int max_det,
                                       int steps, float* records, glass_stream_t stream) {
  if (N == 0) return GLASS_OK;
  GLASS_CHECK_ARG(boxes && scores && text_score && polygons && text_len && chars && count && records, "glass_pack_word_records: null pointer");
  GLASS_CHECK_ARG(K >= 0 && Tw > 0 && max_det > 0 && steps > 0, "glass_pack_word_records: bad sizes");
  hipLaunchKernelGGL(pack_word_records_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, boxes, scores, text_score, polygons, text_len,
                     chars, count, K, Tw, max_det, steps, records);
  GLASS_CHECK_LAUNCH("glass_pack_word_records");
  return GLASS_OK;
}
