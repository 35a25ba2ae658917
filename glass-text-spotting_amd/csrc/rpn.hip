// RRPN proposal selection for a whole batch and all pyramid levels: per (image, level) the k highest
// objectness logits in descending order (ties -> lower flat index, i.e. a stable descending sort),
// anchors generated analytically for the winners only, Box2BoxTransformRotated deltas applied.
//
// HBM-bound select over ~1 M logits per image.  A single workgroup per image would stream that at one
// CU's ~50 GB/s, so the select is a chip-wide 2-level radix histogram on the order-preserving float
// key (12 + 12 bits):
//   hist1   (all CUs)  4096-bin LDS histogram per 8K-element chunk -> global histogram
//   select1 (1 WG per image x level): bin holding the k-th largest, count above it
//   hist2 / select2: the same on the next 12 key bits inside that bin
//   compact (all CUs): keys above the 24-bit threshold -> list A (all winners), keys equal to it ->
//           list T (candidates for the remaining places)
//   finalize (1 WG per image x level): order T by (key, lower index first), take what is missing, sort
//           the k winners (bitonic, LDS), decode boxes.  If T overflows its capacity (pathological
//           ties, e.g. constant maps) the workgroup falls back to an exact 64-bit radix select over
//           the whole level, so the result is always exact and deterministic.
#include "proposal_common.h"

constexpr int RPN_MAX_LEVELS = 8;
constexpr int RPN_TOPK_MAX = 2048;
constexpr int RPN_TIE_CAP = 4096;
constexpr int RPN_CHUNK = 8192;
constexpr int RPN_BINS = 4096;

struct RpnLevel {
  const float* logits; const float* deltas; const float* cell;
  int ldl, ldd, H, W, stride, k, slot_off, total, chunk0;   // chunk0: first chunk id of this level
};
struct RpnBatch {
  RpnLevel lv[RPN_MAX_LEVELS];
  int L, N, A, slots, nchunks;
  float anchor_offset, wx, wy, ww, wh, wa;
  u32* hist1; u32* hist2; int* sel;      // [N][L][4096], [N][L][4096], [N][L][8]
  u64* listA; u64* listT;               // [N][L][RPN_TOPK_MAX], [N][L][RPN_TIE_CAP]
  float* out_boxes; float* out_scores; int* out_level;
};
// sel slots: 0 bin1, 1 above1, 2 bin2, 3 above2, 4 cntA, 5 cntT

__device__ __forceinline__ int find_level(const RpnBatch& p, int chunk) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < RPN_MAX_LEVELS; ++i)
    if (i < p.L && chunk >= p.lv[i].chunk0) l = i;
  return l;
}

__device__ __forceinline__ u32 logit_key(const RpnLevel& lv, int A, const float* lg, int i) {
  const float v = (lv.ldl == A) ? lg[i] : lg[(long)(i / A) * lv.ldl + (i % A)];
  return (v != v) ? 0xffffffffu : float_key(v);   // NaN sorts first in torch.sort(descending)
}

// PASS 1: histogram of key >> 20;  PASS 2: histogram of (key >> 8) & 0xfff among keys whose top 12 bits == bin1
template <int PASS>
__global__ __launch_bounds__(256) void rpn_hist_kernel(RpnBatch p) {
  __shared__ u32 h[RPN_BINS];
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int l = find_level(p, chunk);
  const RpnLevel& lv = p.lv[l];
  const int begin = (chunk - lv.chunk0) * RPN_CHUNK;
  const int end = min(begin + RPN_CHUNK, lv.total);
  const float* lg = lv.logits + (long)n * lv.H * lv.W * lv.ldl;
  const int* sel = p.sel + ((long)n * p.L + l) * 8;
  for (int i = threadIdx.x; i < RPN_BINS; i += 256) h[i] = 0;
  __syncthreads();
  const u32 bin1 = PASS == 2 ? (u32)sel[0] : 0u;
  for (int i = begin + threadIdx.x; i < end; i += 256) {
    const u32 key = logit_key(lv, p.A, lg, i);
    if (PASS == 1) atomicAdd(&h[key >> 20], 1u);
    else if ((key >> 20) == bin1) atomicAdd(&h[(key >> 8) & 0xfffu], 1u);
  }
  __syncthreads();
  u32* gh = (PASS == 1 ? p.hist1 : p.hist2) + ((long)n * p.L + l) * RPN_BINS;
  for (int i = threadIdx.x; i < RPN_BINS; i += 256)
    if (h[i]) atomicAdd(&gh[i], h[i]);
}

// find the bin that holds the k-th largest key: scan the 4096-bin histogram from the top
template <int PASS>
__global__ __launch_bounds__(256) void rpn_select_kernel(RpnBatch p) {
  __shared__ u32 part[256];
  const int l = blockIdx.x, n = blockIdx.y;
  int* sel = p.sel + ((long)n * p.L + l) * 8;
  const u32* gh = (PASS == 1 ? p.hist1 : p.hist2) + ((long)n * p.L + l) * RPN_BINS;
  const int k = PASS == 1 ? p.lv[l].k : p.lv[l].k - sel[1];
  // thread t owns bins [16 t, 16 t + 16)
  u32 loc[16], s = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) { loc[j] = gh[threadIdx.x * 16 + j]; s += loc[j]; }
  part[threadIdx.x] = s;
  __syncthreads();
  u32 above = 0;   // elements in bins of higher threads
  for (int t = threadIdx.x + 1; t < 256; ++t) above += part[t];
  if (above < (u32)k && above + s >= (u32)k) {
    u32 cum = above;
#pragma unroll
    for (int j = 15; j >= 0; --j) {
      if (cum + loc[j] >= (u32)k) {
        sel[PASS == 1 ? 0 : 2] = threadIdx.x * 16 + j;
        sel[PASS == 1 ? 1 : 3] = (int)cum;
        break;
      }
      cum += loc[j];
    }
  }
}

__global__ __launch_bounds__(256) void rpn_compact_kernel(RpnBatch p) {
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int l = find_level(p, chunk);
  const RpnLevel& lv = p.lv[l];
  const int begin = (chunk - lv.chunk0) * RPN_CHUNK;
  const int end = min(begin + RPN_CHUNK, lv.total);
  const float* lg = lv.logits + (long)n * lv.H * lv.W * lv.ldl;
  int* sel = p.sel + ((long)n * p.L + l) * 8;
  const u32 thr24 = ((u32)sel[0] << 12) | (u32)sel[2];
  u64* A = p.listA + ((long)n * p.L + l) * RPN_TOPK_MAX;
  u64* T = p.listT + ((long)n * p.L + l) * RPN_TIE_CAP;
  for (int i = begin + threadIdx.x; i < end; i += 256) {
    const u32 key = logit_key(lv, p.A, lg, i);
    const u32 k24 = key >> 8;
    if (k24 >= thr24) {
      const u64 c = ((u64)key << 32) | (u64)(0xffffffffu - (u32)i);
      if (k24 > thr24) {
        const int pos = atomicAdd(&sel[4], 1);
        if (pos < RPN_TOPK_MAX) A[pos] = c;
      } else {
        const int pos = atomicAdd(&sel[5], 1);
        if (pos < RPN_TIE_CAP) T[pos] = c;
      }
    }
  }
}

__global__ __launch_bounds__(1024) void rpn_finalize_kernel(RpnBatch p) {
  __shared__ u64 buf[RPN_TIE_CAP];        // tie list, then the k winners
  __shared__ u64 win[RPN_TOPK_MAX];
  __shared__ u32 hist[256];
  __shared__ u64 s_prefix;
  __shared__ int s_krem, s_done, s_cnt;
  const int l = blockIdx.x, n = blockIdx.y;
  const RpnLevel& lv = p.lv[l];
  const int k = lv.k;
  const int* sel = p.sel + ((long)n * p.L + l) * 8;
  const float* lg = lv.logits + (long)n * lv.H * lv.W * lv.ldl;
  const int nA = sel[4], nT = sel[5];
  const int need = k - nA;
  int npad = 1;
  while (npad < k) npad <<= 1;

  if (nT <= RPN_TIE_CAP && nA < RPN_TOPK_MAX) {
    // fast path: winners = list A + the `need` best of the tie list
    const u64* T = p.listT + ((long)n * p.L + l) * RPN_TIE_CAP;
    const u64* A = p.listA + ((long)n * p.L + l) * RPN_TOPK_MAX;
    int tpad = 1;
    while (tpad < nT) tpad <<= 1;
    for (int i = threadIdx.x; i < tpad; i += blockDim.x) buf[i] = i < nT ? T[i] : 0;
    __syncthreads();
    bitonic_sort_desc(buf, tpad);
    for (int i = threadIdx.x; i < npad; i += blockDim.x) win[i] = i < nA ? A[i] : (i - nA < need ? buf[i - nA] : 0);
    __syncthreads();
  } else {
    // exact fallback: MSD radix select on the 64-bit (key, ~index) composite over the whole level
    if (threadIdx.x == 0) { s_prefix = 0; s_krem = k; s_done = 0; }
    __syncthreads();
    u64 thr = 0;
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
      for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
      __syncthreads();
      const u64 prefix = s_prefix;
      for (int i = threadIdx.x; i < lv.total; i += blockDim.x) {
        const u64 c = ((u64)logit_key(lv, p.A, lg, i) << 32) | (u64)(0xffffffffu - (u32)i);
        if (pass == 0 || (c >> (shift + 8)) == prefix) atomicAdd(&hist[(u32)(c >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        int krem = s_krem, b = 255;
        u32 cum = 0;
        for (; b >= 0; --b) {
          if (cum + hist[b] >= (u32)krem) break;
          cum += hist[b];
        }
        s_prefix = (prefix << 8) | (u64)b;
        s_krem = krem - (int)cum;
        if (hist[b] == (u32)(krem - (int)cum)) s_done = 1;
      }
      __syncthreads();
      thr = s_prefix << shift;
      if (s_done || pass == 7) break;
    }
    if (threadIdx.x == 0) s_cnt = 0;
    for (int i = threadIdx.x; i < npad; i += blockDim.x) win[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < lv.total; i += blockDim.x) {
      const u64 c = ((u64)logit_key(lv, p.A, lg, i) << 32) | (u64)(0xffffffffu - (u32)i);
      if (c >= thr) {
        const int slot = atomicAdd(&s_cnt, 1);
        if (slot < RPN_TOPK_MAX) win[slot] = c;
      }
    }
    __syncthreads();
  }
  bitonic_sort_desc(win, npad);
  for (int j = threadIdx.x; j < k; j += blockDim.x) {
    const u64 c = win[j];
    const int i = (int)(0xffffffffu - (u32)(c & 0xffffffffu));
    const int a = i % p.A;
    const int cellidx = i / p.A;
    const int w = cellidx % lv.W, h = cellidx / lv.W;
    const float score = lg[(long)cellidx * lv.ldl + a];
    const float* d = lv.deltas + ((long)n * lv.H * lv.W + cellidx) * lv.ldd + a * 5;
    const float acx = (float)w * (float)lv.stride + p.anchor_offset * (float)lv.stride;
    const float acy = (float)h * (float)lv.stride + p.anchor_offset * (float)lv.stride;
    const float aw = lv.cell[a * 5 + 2], ah = lv.cell[a * 5 + 3], aa = lv.cell[a * 5 + 4];
    const float dx = d[0] / p.wx, dy = d[1] / p.wy;
    float dw = d[2] / p.ww, dh = d[3] / p.wh;
    const float da = d[4] / p.wa;
    dw = fminf(dw, SCALE_CLAMP);
    dh = fminf(dh, SCALE_CLAMP);
    const long slot = (long)n * p.slots + lv.slot_off + j;
    float* ob = p.out_boxes + slot * 5;
    ob[0] = dx * aw + acx;
    ob[1] = dy * ah + acy;
    ob[2] = expf(dw) * aw;
    ob[3] = expf(dh) * ah;
    const float pa = da * 180.0f / 3.14159265358979323846f + aa;
    ob[4] = floor_mod(pa + 180.0f, 360.0f) - 180.0f;
    p.out_scores[slot] = score;
    p.out_level[slot] = l;
  }
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" int64_t glass_rpn_workspace_bytes(int N, int num_levels) {
  const size_t nl = (size_t)N * num_levels;
  return (int64_t)(align256(nl * RPN_BINS * 4) * 2 + align256(nl * 8 * 4) + align256(nl * RPN_TOPK_MAX * 8) +
                   align256(nl * RPN_TIE_CAP * 8));
}

extern "C" int glass_rpn_topk_decode(const glass_rpn_level* levels, int num_levels, int N, int A, float anchor_offset,
                                     const float* weights5_host, int slots_per_image, float* out_boxes, float* out_scores,
                                     int* out_level, void* workspace, int64_t workspace_bytes, glass_stream_t stream) {
  GLASS_CHECK_ARG(levels && weights5_host && out_boxes && out_scores && out_level && workspace, "glass_rpn_topk_decode: null pointer");
  GLASS_CHECK_ARG(num_levels >= 1 && num_levels <= RPN_MAX_LEVELS && A > 0, "glass_rpn_topk_decode: num_levels=%d A=%d", num_levels, A);
  GLASS_CHECK_ARG(workspace_bytes >= glass_rpn_workspace_bytes(N, num_levels), "glass_rpn_topk_decode: workspace too small");
  if (N == 0) return GLASS_OK;
  RpnBatch p;
  p.L = num_levels; p.N = N; p.A = A; p.slots = slots_per_image; p.anchor_offset = anchor_offset;
  p.wx = weights5_host[0]; p.wy = weights5_host[1]; p.ww = weights5_host[2]; p.wh = weights5_host[3]; p.wa = weights5_host[4];
  int chunks = 0;
  for (int l = 0; l < num_levels; ++l) {
    const glass_rpn_level& s = levels[l];
    GLASS_CHECK_ARG(s.logits && s.deltas && s.cell_anchors && s.H > 0 && s.W > 0 && s.ldl >= A && s.ldd >= 5 * A,
                    "glass_rpn_topk_decode: level %d bad", l);
    RpnLevel& d = p.lv[l];
    d.logits = s.logits; d.deltas = s.deltas; d.cell = s.cell_anchors; d.ldl = s.ldl; d.ldd = s.ldd; d.H = s.H; d.W = s.W;
    d.stride = s.stride; d.total = s.H * s.W * A;
    d.k = s.topk < d.total ? s.topk : d.total;
    GLASS_CHECK_ARG(d.k > 0 && d.k <= RPN_TOPK_MAX, "glass_rpn_topk_decode: topk=%d (max %d)", s.topk, RPN_TOPK_MAX);
    d.slot_off = s.slot_off;
    GLASS_CHECK_ARG(s.slot_off >= 0 && s.slot_off + d.k <= slots_per_image, "glass_rpn_topk_decode: level %d slots overflow", l);
    d.chunk0 = chunks;
    chunks += cdiv(d.total, RPN_CHUNK);
  }
  for (int l = num_levels; l < RPN_MAX_LEVELS; ++l) p.lv[l] = p.lv[0];
  p.nchunks = chunks;
  const size_t nl = (size_t)N * num_levels;
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  p.hist1 = reinterpret_cast<u32*>(ws); ws += align256(nl * RPN_BINS * 4);
  p.hist2 = reinterpret_cast<u32*>(ws); ws += align256(nl * RPN_BINS * 4);
  p.sel = reinterpret_cast<int*>(ws); ws += align256(nl * 8 * 4);
  const size_t zero_bytes = ws - static_cast<unsigned char*>(workspace);
  p.listA = reinterpret_cast<u64*>(ws); ws += align256(nl * RPN_TOPK_MAX * 8);
  p.listT = reinterpret_cast<u64*>(ws);
  p.out_boxes = out_boxes; p.out_scores = out_scores; p.out_level = out_level;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(workspace, 0, zero_bytes, s);     // histograms + counters, every call
  if (e != hipSuccess) { glass_set_error("glass_rpn_topk_decode: memset: %s", hipGetErrorString(e)); return GLASS_EHIP; }
  hipLaunchKernelGGL(rpn_hist_kernel<1>, dim3(chunks, N), dim3(256), 0, s, p);
  hipLaunchKernelGGL(rpn_select_kernel<1>, dim3(num_levels, N), dim3(256), 0, s, p);
  hipLaunchKernelGGL(rpn_hist_kernel<2>, dim3(chunks, N), dim3(256), 0, s, p);
  hipLaunchKernelGGL(rpn_select_kernel<2>, dim3(num_levels, N), dim3(256), 0, s, p);
  hipLaunchKernelGGL(rpn_compact_kernel, dim3(chunks, N), dim3(256), 0, s, p);
  hipLaunchKernelGGL(rpn_finalize_kernel, dim3(num_levels, N), dim3(1024), 0, s, p);
  GLASS_CHECK_LAUNCH("glass_rpn_topk_decode");
  return GLASS_OK;
}
