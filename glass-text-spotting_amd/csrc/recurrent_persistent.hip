// One launch per recurrent layer: the BiLSTM recurrence with its recurrent weights resident in registers and the hidden
// state handed between workgroups inside the launch (reference
// glass/modeling/recognition/recognizer_encoder.py:118-144, nn.LSTM gate order i,f,g,o, zero initial state).
//
// recognition.hip runs the same recurrence as T launches per layer (lstm_step_kernel): every step re-streams the 2 x 1 MiB
// recurrent matrix from L2 and pays a kernel boundary, 8-10 us per step for ~1.7 us of matrix-core work.  Here a CHAIN =
// (group of 16 RoIs, direction) is served by a SET of 8 workgroups; workgroup `ub` of the set owns hidden units
// [32 ub, 32 ub + 32): its 4 gates x 32 units x 256 slice of W_hh (128 KiB per direction) stays in registers as
// v_mfma_f32_16x16x4_f32 A fragments for the whole layer (8 wavefronts x 16 gate rows, 64 registers per direction), the cell
// state of its (RoI, unit) elements stays in one register per thread, and only h_t - 16 x 256 floats per chain and step -
// crosses workgroups: every thread publishes its element as ONE naturally aligned 8-byte {step tag, value} granule with a
// device-scope (sc1, write-through) store, and the 8 workgroups of the set sweep the chain's 4096 granules with device-scope
// loads until every tag carries the step they wait for - the data IS the flag, no fence, one L2 / fabric round trip per step
// (cdna_hip_programming.md Guideline 16 form R2).  Granules are double-buffered by step parity: a workgroup can run at most
// one step ahead of a peer of its set (its step s+1 needs every peer's h_s), so a slot is never overwritten before every
// reader has taken the value of two steps earlier.
//
// A workgroup serves ND directions x G RoI groups = ND*G independent chains round-robin, so the sweep of one chain finds its
// granules already landed while the matrix cores ran the other chains (ND = 2, G = 1: 128 workgroups at R = 256, half the
// chip left to the other pipeline stream's convolutions).
//
// Residency: nothing assumes that the whole grid is resident or dispatched in blockIdx order.  A workgroup takes a ticket
// when it STARTS (ticket / 8 = set, ticket % 8 = unit block), so the workgroups that are running always form complete sets
// in start order plus at most one incomplete set whose missing members start as soon as any slot frees up - they wait for
// nothing but already-running workgroups.  Every spin is bounded: a wavefront that gives up raises the library's status
// word (glass_recurrence_status) and stops waiting, so a broken hand-off shows as an error, never as a hung GPU.
//
// Arithmetic is lstm_step_kernel's, instruction for instruction (same MFMA operand order and accumulator pairing, same
// gate functions): the two paths are bit-identical (tests/test_gpu_g_persistent_rnn.py).
#include <atomic>
#include <mutex>
#include "recognition_common.h"

typedef __attribute__((address_space(1))) unsigned long long gu64;

namespace {

constexpr int PL_HD = 256;            // hidden size
constexpr int PL_RB = 16;             // RoIs per chain (the N of a 16x16x4 MFMA)
constexpr int PL_UB = 32;             // hidden units per workgroup
constexpr int PL_NUB = PL_HD / PL_UB; // workgroups per set
constexpr int PL_THREADS = 512;
constexpr int PL_GRAN = PL_RB * PL_HD;        // granules per chain and parity
constexpr unsigned PL_SPIN_LIMIT = 1u << 21;  // x (one sweep + s_sleep) ~ 2-4 s: far beyond any wait for a peer to be scheduled
constexpr int PL_CTRL_BYTES = 256;

// what a bounded wait reports to, and how long it waits
struct PlGuard {
  int* status;         // library-owned, sticky (glass_recurrence_status): bit 0 = an LSTM hand-off gave up, bit 1 = a decoder hand-off
  int* call_status;    // caller-owned word of THIS call (may be null): the same bits - what the host checks at its next read-back
  unsigned spin_limit; // sweeps a wavefront waits for a peer before it gives up (PL_SPIN_LIMIT; lowered by the test hook only)
  int withhold;        // test hook: the workgroup that drew this start ticket never publishes (-1: none)
};

struct PlParams {
  const float* xg;
  const float* whh;
  float* out;
  unsigned long long* gran;
  unsigned* ctrl;      // [0] start tickets
  PlGuard guard;
  int R, T, NG, nsets;
};

// a wavefront stops waiting for a hand-off: both status words get the bit, the wavefront skips every later wait
__device__ __forceinline__ void give_up(const PlGuard& g, int bit, int lane, bool& dead) {
  dead = true;
  if (lane == 0) {
    atomicOr(g.status, bit);
    if (g.call_status) atomicOr(g.call_status, bit);
  }
}

// workgroup barrier for LDS hand-offs only: waits for this wavefront's LDS operations, NOT for its global loads -
// __syncthreads() drains vmcnt too, which would stall every step on the prefetches that are meant to stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// this thread's 8 granules of a chain slot: issue the loads (device scope: served by L2 / the fabric, never a stale L1 line)
__device__ __forceinline__ void sweep_issue(const gu64* src, int tid, unsigned long long (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = __hip_atomic_load(src + tid + PL_THREADS * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... and take them: re-read until every tag is `epoch`, then the values go to LDS as the chain's h rows
__device__ __forceinline__ void sweep_finish(const gu64* src, unsigned epoch, int tid, int lane, unsigned long long (&v)[8],
                                             float (*hs)[PL_HD + 4], bool& dead, const PlGuard& guard, int bit) {
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) ok &= (unsigned)(v[j] >> 32) == epoch;
    if (__all(ok) || dead) break;
    __builtin_amdgcn_s_sleep(2);
    if (++spins > guard.spin_limit) give_up(guard, bit, lane, dead);          // wavefront-uniform
    sweep_issue(src, tid, v);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = tid + PL_THREADS * j;
    hs[i >> 8][i & 255] = __uint_as_float((unsigned)v[j]);
  }
}

template <int ND, int G>
__global__ __launch_bounds__(PL_THREADS) void lstm_persistent_kernel(PlParams p) {
  __shared__ __attribute__((aligned(16))) float hs[PL_RB][PL_HD + 4];   // +4: 16-byte row skew against bank conflicts
  __shared__ float gates[PL_RB][4 * PL_UB + 4];
  __shared__ unsigned s_ticket;
  constexpr int NCH = ND * G;                       // chains a workgroup interleaves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_ticket = atomicAdd(p.ctrl, 1u);
  __syncthreads();
  const int set = (int)(s_ticket >> 3), ub = (int)(s_ticket & 7);
  if (set >= p.nsets) return;
  const int gb = ND == 2 ? set : set >> 1;          // block of G RoI groups
  const int dir0 = ND == 2 ? 0 : (set & 1);
  const int ncv = ND * min(G, p.NG - gb * G);       // chains that exist (the last block of an odd group count has fewer)
  // this wavefront's 16 gate rows: tile = gate * 2 + half-of-32-units (lstm_step_kernel's tiling, one tile per wavefront)
  const int g = wave >> 1, uh = wave & 1;
  const int row0 = g * PL_HD + ub * PL_UB + uh * 16;
  float4 a[ND][PL_HD / 16];
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    const float* wp = p.whh + ((long)(dir0 + d) * 4 * PL_HD + row0 + (lane & 15)) * PL_HD + (lane >> 4) * 4;
#pragma unroll
    for (int S = 0; S < PL_HD / 16; ++S) {
      a[d][S] = *reinterpret_cast<const float4*>(wp + S * 16);
      // resident for the whole layer: keep the compiler from re-loading them inside the step loop
      asm volatile("" : "+v"(a[d][S].x), "+v"(a[d][S].y), "+v"(a[d][S].z), "+v"(a[d][S].w));
    }
  }
  const int pr = tid >> 5, ul = tid & 31, u = ub * PL_UB + ul;   // the (RoI row, unit) this thread owns in every chain
  float c[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) c[i] = 0.f;
  bool dead = false;
  const int T = p.T;
  // chain ci of this workgroup = (RoI group gb*G + ci / ND, direction dir0 + ci % ND); its granules: [parity][16 x 256]
  gu64* const gbase = (gu64*)p.gran + (long)((gb * G) * 2 + dir0) * 2 * PL_GRAN;    // chain (gi, d) at + (gi*2 + d) * 2 * PL_GRAN
  unsigned long long v[8];
  // input projection (4 gate pre-activations of this thread's element) of a chain-step: independent of h, so the NEXT
  // chain-step's values are fetched one chain-step ahead
  auto load_x = [&](int s_, int ci_, float (&x)[4]) {
    const int dir = dir0 + ci_ % ND;
    const int t = dir == 0 ? s_ : T - 1 - s_;
    const int rr = (gb * G + ci_ / ND) * PL_RB + pr;
    x[0] = x[1] = x[2] = x[3] = 0.f;
    if (rr < p.R) {
      const float* xr = p.xg + (((long)rr * T + t) * 2 + dir) * (4 * PL_HD) + u;
      x[0] = xr[0]; x[1] = xr[PL_HD]; x[2] = xr[2 * PL_HD]; x[3] = xr[3 * PL_HD];
    }
  };
  float xn[4];
  load_x(0, 0, xn);
#ifdef GLASS_PL_STAMPS   // timing build (scripts/build_variant_lib.sh plst -DGLASS_PL_STAMPS): phase totals of ticket 0's wavefront 0
  unsigned long long st[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define PL_STAMP(k) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); st[k] += tn - tlast; tlast = tn; }
#else
#define PL_STAMP(k)
#endif
  for (int s = 0; s < T; ++s) {
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
      if (ci >= ncv) continue;
      const int gi = ci / ND, d = ci % ND;
      const int dir = dir0 + d;
      const int t = dir == 0 ? s : T - 1 - s;
      const int rr = (gb * G + gi) * PL_RB + pr;
      const bool valid = rr < p.R;
      const float x0 = xn[0], x1 = xn[1], x2 = xn[2], x3 = xn[3];
      const int cn = ci + 1 < ncv ? ci + 1 : 0;            // the chain-step after this one
      const int sn = ci + 1 < ncv ? s : s + 1;
      gu64* gp = gbase + (long)(gi * 2 + d) * 2 * PL_GRAN;
      float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
      if (s > 0) {
        const gu64* src = gp + ((s - 1) & 1) * PL_GRAN;
        if (NCH == 1 || (s == 1 && ci == 0)) sweep_issue(src, tid, v);       // (otherwise issued during the previous chain-step)
        sweep_finish(src, (unsigned)s, tid, lane, v, hs, dead, p.guard, 1);
        PL_STAMP(0)
        lds_barrier();
        PL_STAMP(1)
      }
      if (sn < T) {
        load_x(sn, cn, xn);
        // the NEXT chain-step's granules were published while this workgroup worked on the other chains: fetch them under
        // this step's matrix-core phase (re-read at its turn only if a tag is still old)
        if (NCH > 1 && ncv > 1 && sn > 0 && !(sn == 1 && cn == 0))
          sweep_issue(gbase + (long)((cn / ND) * 2 + cn % ND) * 2 * PL_GRAN + ((sn - 1) & 1) * PL_GRAN, tid, v);
      }
      if (s > 0) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
        const float* hp = &hs[lane & 15][(lane >> 4) * 4];
#pragma unroll
        for (int S = 0; S < PL_HD / 16; ++S) {
          const float4 b = *reinterpret_cast<const float4*>(hp + S * 16);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[d][S].x, b.x, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[d][S].y, b.y, acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[d][S].z, b.z, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[d][S].w, b.w, acc2, 0, 0, 0);
        }
        acc += acc2;
        // C layout: col = lane & 15 (RoI), row = (lane >> 4) * 4 + e
#pragma unroll
        for (int e = 0; e < 4; ++e) gates[lane & 15][g * PL_UB + uh * 16 + (lane >> 4) * 4 + e] = acc[e];
        PL_STAMP(2)
        lds_barrier();
        PL_STAMP(3)
        g0 = gates[pr][ul]; g1 = gates[pr][PL_UB + ul]; g2 = gates[pr][2 * PL_UB + ul]; g3 = gates[pr][3 * PL_UB + ul];
      }
      const LstmCell cell = lstm_cell(x0 + g0, x1 + g1, x2 + g2, x3 + g3, c[ci]);
      c[ci] = cell.c;
      const float hn = valid ? cell.h : 0.f;
      if (s + 1 < T && (int)s_ticket != p.guard.withhold)
        __hip_atomic_store(gp + (s & 1) * PL_GRAN + pr * PL_HD + u, ((unsigned long long)(unsigned)(s + 1) << 32) | __float_as_uint(hn),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (valid) p.out[((long)rr * T + t) * (2 * PL_HD) + dir * PL_HD + u] = hn;
      PL_STAMP(4)
    }
  }
#ifdef GLASS_PL_STAMPS
  if (s_ticket == 0 && tid == 0)
    for (int k = 0; k < 6; ++k) reinterpret_cast<unsigned long long*>(p.ctrl + 16)[k] = st[k];
#endif
}

std::mutex g_status_mutex;
int* g_status[64] = {};

}  // namespace

// the sticky status word of the calling thread's current device (allocated on first use; never freed: process lifetime)
static int* recurrence_status_word() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(g_status_mutex);
  if (!g_status[dev]) {
    int* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), 64) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 64) != hipSuccess) { (void)hipFree(p); return nullptr; }
    g_status[dev] = p;
  }
  return g_status[dev];
}

// test hook state (glass_recurrence_test_hook): read by the launch wrappers, process-wide
static std::atomic<unsigned> g_spin_limit{PL_SPIN_LIMIT};
static std::atomic<int> g_withhold{-1};

static bool fill_guard(PlGuard& g, int* call_status) {
  g.status = recurrence_status_word();
  g.call_status = call_status;
  g.spin_limit = g_spin_limit.load();
  g.withhold = g_withhold.load();
  return g.status != nullptr;
}

extern "C" int glass_recurrence_test_hook(int64_t spin_limit, int withhold_ticket) {
  GLASS_CHECK_ARG(spin_limit <= (int64_t)PL_SPIN_LIMIT, "glass_recurrence_test_hook: spin_limit above the built-in bound");
  g_spin_limit.store(spin_limit <= 0 ? PL_SPIN_LIMIT : (unsigned)spin_limit);
  g_withhold.store(withhold_ticket < 0 ? -1 : withhold_ticket);
  return GLASS_OK;
}

extern "C" int glass_recurrence_status(int* status_out, int reset) {
  GLASS_CHECK_ARG(status_out, "glass_recurrence_status: null pointer");
  int* w = recurrence_status_word();
  if (!w) { glass_set_error("glass_recurrence_status: no status word (hipMalloc failed)"); return GLASS_EHIP; }
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(status_out, w, sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess && reset) e = hipMemset(w, 0, sizeof(int));
  if (e != hipSuccess) { glass_set_error("glass_recurrence_status: %s", hipGetErrorString(e)); return GLASS_EHIP; }
  return GLASS_OK;
}

extern "C" int64_t glass_bilstm_persistent_workspace_bytes(int R, int Hd) {
  (void)Hd;
  return (int64_t)PL_CTRL_BYTES + (int64_t)cdiv(R, PL_RB) * 2 * 2 * PL_GRAN * (int64_t)sizeof(unsigned long long);
}

extern "C" int glass_bilstm_recurrence_persistent(const float* xg, const float* w_hh, float* out, int R, int T, int Hd,
                                                  int dirs_per_workgroup, int groups_per_workgroup, int* call_status,
                                                  void* workspace, int64_t workspace_bytes, glass_stream_t stream) {
  GLASS_CHECK_ARG(Hd == PL_HD, "glass_bilstm_recurrence_persistent: only Hd=256 is built (got %d)", Hd);
  if (R == 0) return GLASS_OK;
  GLASS_CHECK_ARG(xg && w_hh && out && T > 0 && workspace, "glass_bilstm_recurrence_persistent: bad args");
  GLASS_CHECK_ARG(workspace_bytes >= glass_bilstm_persistent_workspace_bytes(R, Hd),
                  "glass_bilstm_recurrence_persistent: workspace too small");
  GLASS_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "glass_bilstm_recurrence_persistent: workspace must be 16-byte aligned");
  const int nd = dirs_per_workgroup == 0 ? 2 : dirs_per_workgroup;
  const int ng = groups_per_workgroup == 0 ? 1 : groups_per_workgroup;
  GLASS_CHECK_ARG((nd == 1 || nd == 2) && (ng == 1 || ng == 2) && !(nd == 1 && ng == 2),
                  "glass_bilstm_recurrence_persistent: (directions, groups) per workgroup must be (2,1), (2,2) or (1,1) (got %d,%d)", nd, ng);
  hipStream_t s = (hipStream_t)stream;
  PlParams p;
  p.xg = xg; p.whh = w_hh; p.out = out; p.R = R; p.T = T;
  p.NG = cdiv(R, PL_RB);
  p.nsets = cdiv(p.NG, ng) * (nd == 2 ? 1 : 2);
  p.ctrl = static_cast<unsigned*>(workspace);
  p.gran = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + PL_CTRL_BYTES);
  if (!fill_guard(p.guard, call_status)) { glass_set_error("glass_bilstm_recurrence_persistent: no status word (hipMalloc failed)"); return GLASS_EHIP; }
  // every polled word (tickets, granule tags) is zero before every launch; tags count steps from 1
  hipError_t e = hipMemsetAsync(workspace, 0, (size_t)glass_bilstm_persistent_workspace_bytes(R, Hd), s);
  if (e != hipSuccess) { glass_set_error("glass_bilstm_recurrence_persistent: memset: %s", hipGetErrorString(e)); return GLASS_EHIP; }
  const dim3 grid(p.nsets * PL_NUB), block(PL_THREADS);
  if (nd == 2 && ng == 1) hipLaunchKernelGGL((lstm_persistent_kernel<2, 1>), grid, block, 0, s, p);
  else if (nd == 2) hipLaunchKernelGGL((lstm_persistent_kernel<2, 2>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((lstm_persistent_kernel<1, 1>), grid, block, 0, s, p);
  GLASS_CHECK_LAUNCH("glass_bilstm_recurrence_persistent");
  return GLASS_OK;
}

// ================================================================== greedy attention-GRU decoder, one launch
// Reference: AttentionRecognitionHead.sample (glass/modeling/recognition/prediction_aster.py:63-99), AttentionUnit :247-266,
// DecoderUnit :291-302.  recognition.hip runs a decoding step as two launches (dec_fc_att_kernel, dec_gru_kernel) that re-stream
// 2.7 MB of weights from L2 per RoI group and step: 24 us per step, 0.64 ms per image of a 7.9 ms one-image step.  Here a
// group of 16 RoIs is served by a SET of 16 workgroups for all max_len steps; workgroup j of the set owns
//   * rows [16 j, 16 j + 16) of every matrix that multiplies a vector of ALL 16 RoIs - the three GRU gates (W_hh and the
//     context half W_ih[:, D:2D]) and sEmbed - as v_mfma_f32_16x16x4_f32 A fragments in registers, cut into K-quarters
//     (wavefronts 4-7: W_hh and sEmbed, which need h_i only; wavefronts 0-3: the context half);
//   * rows [8 j, 8 j + 8) of the classifier fc the same way (wavefronts 0-3, under the others' W_hh / sEmbed products);
//   * the attention, the soft-max / arg-max and the outputs of RoI j of the group: its x and xEmbed(x) rows in LDS (2 x 32 KB,
//     step-invariant).  (A first form streamed the whole 97 KB of fc from L2 per workgroup and step: 5.8 K of 25 K cycles.)
// The embedding half of the GRU input needs no arithmetic at run time: W_ih[:, :D] emb[y] + b_ih is a [C, 3D] table
// (`emb_gi`, built once at load), gathered by the arg-max.
// Three hand-offs per step inside the set, all as 8-byte {step tag, value} granules (Guideline 16 R2), double-buffered by
// step parity like the BiLSTM's: h_i (16 x 256 per consumer, from the owners of its unit slices), sEmbed(h_i) + fc(h_i)
// (256 + C per consumer: a RoI's rows from the 16 row owners) and ctx_i + arg-max (16 x 257, from the owners of the RoIs).
// Step i:  sweep h_i | sEmbed(h_i) (wavefronts 4-7) + fc slice (0-3) for the 16 RoIs | publish both slices | W_hh h_i
//   (wavefronts 4-7, in the shadow of the hand-off); sweep sEmbed of RoI j (wavefronts 0-3); its logits -> soft-max /
//   arg-max / output row of step i-1 (wavefront 4) |
//   energies, soft-max, context | publish ctx_j | sweep ctx, y | W_ih[:, D:] ctx (wavefronts 0-3) | gates -> h_{i+1} slice.
// (A first form kept sEmbed's whole 256 x 256 matrix in 128 registers per thread and skipped the second hand-off: 44
//  registers spilled to scratch and every phase paid for it - 32 K cycles per step.)
// Start tickets, bounded spins and the status word are the BiLSTM kernel's (bit 1 of the status word).
namespace {

constexpr int PD_D = 256, PD_RB = 16, PD_UB = 16, PD_NS = PD_D / PD_UB;   // hidden size, RoIs per group, rows / workgroups per set
constexpr int PD_T = 32, PD_CMAX = 128, PD_THREADS = 512;
constexpr int PD_GRAN = PD_RB * PD_D;
constexpr int PD_CS = PD_CMAX / PD_NS;   // classes per workgroup of the distributed classifier (8: 16 x 8 >= C)

struct PdParams {
  const float *x, *xproj;
  const float *sW, *sB, *wW, *wB, *emb_gi, *w_ih, *w_hh, *b_hh, *fcW, *fcB;
  float temperature;
  int R, T, C, max_len, nsets;
  float* out; int* pred;
  unsigned long long *hgran, *cgran, *sgran, *ygran, *lgran;   // per set: [parity][16 x 256] x 3, [parity][16], [parity][16 x 128]
  unsigned* ctrl; PlGuard guard;
};

__device__ __forceinline__ unsigned long long granule(unsigned epoch, float v) {
  return ((unsigned long long)epoch << 32) | __float_as_uint(v);
}

// 16 rows x (K-quarter kq) of a row-major matrix . the 16 RoI rows of `bs`: one 16 x 16 partial tile
__device__ __forceinline__ f32x4 quarter_job(const float4 (&a)[4], const float (*bs)[PD_D + 4], int kq, int lane) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int S = 0; S < 4; ++S) {
    const float4 b = *reinterpret_cast<const float4*>(&bs[lane & 15][64 * kq + 16 * S + 4 * (lane >> 4)]);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[S].x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[S].y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[S].z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[S].w, b.w, acc, 0, 0, 0);
  }
  return acc;
}

__global__ __launch_bounds__(PD_THREADS) void decode_persistent_kernel(PdParams p) {
  __shared__ __attribute__((aligned(16))) float hs[PD_RB][PD_D + 4];       // h_i of the group's 16 RoIs (B operand of W_hh, sEmbed)
  __shared__ __attribute__((aligned(16))) float cs[PD_RB][PD_D + 4];       // ctx_i of the 16 RoIs (B operand of the context half)
  __shared__ __attribute__((aligned(16))) float xs[PD_T][PD_D];            // x of this workgroup's RoI
  __shared__ __attribute__((aligned(16))) float xps[PD_T][PD_D + 4];       // xEmbed(x) of this workgroup's RoI (+4: row skew, 16 lanes per row)
  __shared__ float gpart[2][4][3][PD_RB][PD_UB];                            // [context | hidden][K-quarter][gate][RoI][unit]
  __shared__ float spart[4][PD_RB][PD_UB];                                  // sEmbed: [K-quarter][RoI][row]
  __shared__ __attribute__((aligned(16))) float sproj[PD_D];
  __shared__ __attribute__((aligned(16))) float wws[PD_D];                  // wEmbed's weight row
  __shared__ float energy[PD_T];
  __shared__ float alpha[PD_THREADS / 64][PD_T];                            // per wavefront: its own copy of the attention weights
  __shared__ float ctxp[2][PD_D];
  __shared__ float lpart[4][PD_RB][PD_UB];                                  // fc: [K-quarter][RoI][class of this workgroup's slice]
  __shared__ float fcbs[PD_CMAX];                                           // fc bias
  __shared__ unsigned s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_ticket = atomicAdd(p.ctrl, 1u);
  __syncthreads();
  const int set = (int)(s_ticket >> 4), j = (int)(s_ticket & 15);
  if (set >= p.nsets) return;
  const int T = p.T, C = p.C, L = p.max_len;
  const int rr = set * PD_RB + j;                 // the RoI whose attention / classifier this workgroup runs
  const bool roi_ok = rr < p.R;
  gu64* const hg = (gu64*)p.hgran + (long)set * 2 * PD_GRAN;
  gu64* const cg = (gu64*)p.cgran + (long)set * 2 * PD_GRAN;
  gu64* const sg = (gu64*)p.sgran + (long)set * 2 * PD_GRAN;
  gu64* const yg = (gu64*)p.ygran + (long)set * 2 * PD_RB;
  gu64* const lg = (gu64*)p.lgran + (long)set * 2 * PD_RB * PD_CMAX;

  // ---- resident operands: wavefront w = (part = w >> 2: 0 the context half W_ih[:, D:2D], 1 W_hh + sEmbed; K-quarter w & 3)
  const int part = wave >> 2, kq = wave & 3;
  float4 ga[3][4], sa[4];
  {
    const float* Wp = part == 0 ? p.w_ih + PD_D : p.w_hh;
    const int ld = part == 0 ? 2 * PD_D : PD_D;
    const int koff = 64 * kq + 4 * (lane >> 4);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int S = 0; S < 4; ++S) {
        ga[g][S] = *reinterpret_cast<const float4*>(Wp + (long)(g * PD_D + j * PD_UB + (lane & 15)) * ld + koff + 16 * S);
        asm volatile("" : "+v"(ga[g][S].x), "+v"(ga[g][S].y), "+v"(ga[g][S].z), "+v"(ga[g][S].w));
      }
    // the fourth row block: sEmbed's rows [16 j, 16 j + 16) on wavefronts 4-7; on wavefronts 0-3 the classifier's rows of THIS
    // workgroup's class slice [PD_CS j, PD_CS j + PD_CS) (rows beyond the slice or beyond C are zero) - fc is k-blocked [D/4][C][4]
    const int crow = j * PD_CS + (lane & 15);
    const bool crow_ok = (lane & 15) < PD_CS && crow < C;
#pragma unroll
    for (int S = 0; S < 4; ++S) {
      if (part == 1) sa[S] = *reinterpret_cast<const float4*>(p.sW + (long)(j * PD_UB + (lane & 15)) * PD_D + koff + 16 * S);
      else sa[S] = crow_ok ? reinterpret_cast<const float4*>(p.fcW)[(long)(16 * kq + 4 * S + (lane >> 4)) * C + crow] : make_float4(0.f, 0.f, 0.f, 0.f);
      asm volatile("" : "+v"(sa[S].x), "+v"(sa[S].y), "+v"(sa[S].z), "+v"(sa[S].w));
    }
  }
  // the RoI's step-invariant rows
  for (int i = tid; i < PD_T * (PD_D / 4); i += PD_THREADS) {
    const int t = i / (PD_D / 4), d4 = i % (PD_D / 4);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (roi_ok && t < T) {
      a = reinterpret_cast<const float4*>(p.x + ((long)rr * T + t) * PD_D)[d4];
      b = reinterpret_cast<const float4*>(p.xproj + ((long)rr * T + t) * PD_D)[d4];
    }
    *reinterpret_cast<float4*>(&xs[t][d4 * 4]) = a;
    *reinterpret_cast<float4*>(&xps[t][d4 * 4]) = b;
  }
  for (int i = tid; i < PD_RB * (PD_D + 4); i += PD_THREADS) (&hs[0][0])[i] = 0.f;        // h_0 = 0
  if (tid < PD_D) wws[tid] = p.wW[tid];
  if (tid < PD_CMAX) fcbs[tid] = tid < p.C ? p.fcB[tid] : 0.f;
  // element (RoI row pr, row / unit pu of this workgroup's 16) of the threads that finish the row-owner products
  const int pr = (tid >> 4) & 15, pu = tid & 15, u = j * PD_UB + pu;
  const float bh0 = p.b_hh[u], bh1 = p.b_hh[PD_D + u], bh2 = p.b_hh[2 * PD_D + u], sbias = p.sB[u];
  const float wbias = p.wB[0];
  bool dead = false;
  __syncthreads();
#ifdef GLASS_PL_STAMPS
  unsigned long long st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define PD_STAMP(k) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); st[k] += tn - tlast; tlast = tn; }
#else
#define PD_STAMP(k)
#endif

  for (int i = 0; i <= L; ++i) {
    // ---- A: h_i of the whole group (h_0 = 0 is in LDS already)
    if (i > 0) {
      unsigned long long v[8];
      const gu64* src = hg + (i & 1) * PD_GRAN;
      sweep_issue(src, tid, v);
      sweep_finish(src, (unsigned)i, tid, lane, v, hs, dead, p.guard, 2);
    }
    lds_barrier();
    PD_STAMP(0)
    const float hprev = hs[pr][u];                  // for the cell update at the end of the step
    // ---- B: wavefronts 4-7: W_hh h_i and sEmbed(h_i) for the 16 RoIs (this workgroup's 16 rows, one K-quarter each);
    //         wavefronts 0-3: classifier partial sums on h_i of this RoI (= the output of step i - 1)
    if (part == 1) {
      if (i < L) {
        const f32x4 acc = quarter_job(sa, hs, kq, lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) spart[kq][lane & 15][(lane >> 4) * 4 + e] = acc[e];
      }
    } else if (i > 0) {
      // classifier on h_i (= the output of step i - 1): this workgroup's class slice for all 16 RoIs, one K-quarter per wavefront
      const f32x4 acc = quarter_job(sa, hs, kq, lane);
#pragma unroll
      for (int e = 0; e < 4; ++e) lpart[kq][lane & 15][(lane >> 4) * 4 + e] = acc[e];
    }
    lds_barrier();
    PD_STAMP(1)
    // ---- publish this workgroup's rows of sEmbed(h_i) and its class slice of fc(h_i) for the 16 RoIs (threads 0-255)
    if (tid < PD_RB * PD_UB) {
      if (i < L && (int)s_ticket != p.guard.withhold) {
        const float sv = ((spart[0][pr][pu] + spart[1][pr][pu]) + (spart[2][pr][pu] + spart[3][pr][pu])) + sbias;
        __hip_atomic_store(sg + (i & 1) * PD_GRAN + pr * PD_D + u, granule((unsigned)(i + 1), sv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (i > 0 && pu < PD_CS && j * PD_CS + pu < C) {
        const float lv = (lpart[0][pr][pu] + lpart[1][pr][pu]) + (lpart[2][pr][pu] + lpart[3][pr][pu]);
        __hip_atomic_store(lg + ((i & 1) * PD_RB + pr) * PD_CMAX + j * PD_CS + pu, granule((unsigned)(i + 1), lv), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      // ---- C: sEmbed(h_i) of THIS RoI, one row from each of the 16 row owners (one granule per thread)
      if (i < L) {
        const gu64* src = sg + (i & 1) * PD_GRAN + j * PD_D + tid;
        unsigned spins = 0;
        unsigned long long sv;
        for (;;) {
          sv = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__all((unsigned)(sv >> 32) == (unsigned)(i + 1)) || dead) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > p.guard.spin_limit) give_up(p.guard, 2, lane, dead);
        }
        sproj[tid] = __uint_as_float((unsigned)sv);
      }
    } else {
      // ---- wavefronts 4-7: W_hh h_i (needed by the cell update at the end of the step only) in the shadow of the hand-off
      if (i < L) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const f32x4 acc = quarter_job(ga[g], hs, kq, lane);
#pragma unroll
          for (int e = 0; e < 4; ++e) gpart[1][kq][g][lane & 15][(lane >> 4) * 4 + e] = acc[e];
        }
      }
    }
    if (wave == 4) {
      // ---- wavefront 4: the logits of THIS RoI from the 16 class-slice owners -> soft-max, arg-max, output row of step i - 1
      if (i > 0) {
        float vv[2], m = -INFINITY;
        {
          const gu64* src = lg + ((i & 1) * PD_RB + j) * PD_CMAX;
          unsigned spins = 0;
          unsigned long long l0, l1;
          for (;;) {
            l0 = __hip_atomic_load(src + min(lane, C - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            l1 = __hip_atomic_load(src + min(lane + 64, C - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(l0 >> 32) == (unsigned)(i + 1) && (unsigned)(l1 >> 32) == (unsigned)(i + 1)) || dead) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > p.guard.spin_limit) give_up(p.guard, 2, lane, dead);
          }
          vv[0] = lane < C ? (fcbs[lane] + __uint_as_float((unsigned)l0)) * p.temperature : -INFINITY;
          vv[1] = lane + 64 < C ? (fcbs[lane + 64] + __uint_as_float((unsigned)l1)) * p.temperature : -INFINITY;
          m = fmaxf(vv[0], vv[1]);
        }
        m = wave_max(m);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) { vv[e] = (lane + 64 * e) < C ? __expf(vv[e] - m) : 0.f; s += vv[e]; }
        s = wave_sum(s);
        const float inv = 1.f / s;
        float best = -1.f;
        int besti = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = lane + 64 * e;
          if (c < C) {
            const float pv = vv[e] * inv;
            if (roi_ok) p.out[((long)rr * L + (i - 1)) * C + c] = pv;
            if (pv > best) { best = pv; besti = c; }
          }
        }
        wave_argmax_first(best, besti);
        if (lane == 0) {
          if (roi_ok) p.pred[(long)rr * L + (i - 1)] = besti;
          // the symbol of step i - 1 feeds step i: tagged i + 1 like the context it is consumed with
          if (i < L) __hip_atomic_store(yg + (i & 1) * PD_RB + j, granule((unsigned)(i + 1), __int_as_float(roi_ok ? besti : 0)),
                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else if (lane == 0) {
        __hip_atomic_store(yg + j, granule(1u, __int_as_float(0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // before step 0: [GO]
      }
    }
    if (i == L) break;
    lds_barrier();
    PD_STAMP(2)
    //      energies e[t] = wEmbed(tanh(sProj + xProj[t])): thread (t = tid >> 4, 16 channels 4 dq + 64 m), 16 lanes per t
    {
      const int t = tid >> 4, dq = tid & 15;
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float4 sp = *reinterpret_cast<const float4*>(&sproj[64 * m + 4 * dq]);
        const float4 xp = *reinterpret_cast<const float4*>(&xps[t][64 * m + 4 * dq]);
        const float4 wq = *reinterpret_cast<const float4*>(&wws[64 * m + 4 * dq]);
        s += wq.x * tanh_fast(sp.x + xp.x) + wq.y * tanh_fast(sp.y + xp.y) + wq.z * tanh_fast(sp.z + xp.z) + wq.w * tanh_fast(sp.w + xp.w);
      }
      s = row16_sum(s);
      if (dq == 0) energy[t] = s + wbias;
    }
    lds_barrier();
    PD_STAMP(3)
    //      soft-max over T, every wavefront for itself (its own LDS copy: no barrier), then context = alpha . x:
    //      thread (channel d = tid & 255, time half = tid >> 8)
    {
      const float v = lane < T ? energy[lane & (PD_T - 1)] : -INFINITY;
      const float m = wave_max(v);
      const float e = lane < T ? __expf(v - m) : 0.f;
      const float s = wave_sum(e);
      if (lane < PD_T) alpha[wave][lane] = e / s;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int d = tid & 255, th = tid >> 8;
      float c = 0.f;
#pragma unroll
      for (int t = 0; t < PD_T / 2; ++t) c += alpha[wave][2 * t + th] * xs[2 * t + th][d];
      ctxp[th][d] = c;
    }
    lds_barrier();
    PD_STAMP(4)
    if (tid < PD_D)
      __hip_atomic_store(cg + (i & 1) * PD_GRAN + j * PD_D + tid, granule((unsigned)(i + 1), roi_ok ? ctxp[0][tid] + ctxp[1][tid] : 0.f),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- E: the contexts and symbols of the whole group
    float eg0 = 0.f, eg1 = 0.f, eg2 = 0.f;
    {
      unsigned long long v[8];
      const gu64* src = cg + (i & 1) * PD_GRAN;
      sweep_issue(src, tid, v);
      sweep_finish(src, (unsigned)(i + 1), tid, lane, v, cs, dead, p.guard, 2);
      PD_STAMP(5)
      // the symbols: thread (pr, pu) needs y of RoI pr for its row of the embedding table
      if (tid < PD_RB * PD_UB) {
        unsigned spins = 0;
        unsigned long long yv;
        for (;;) {
          yv = __hip_atomic_load(yg + (i & 1) * PD_RB + pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__all((unsigned)(yv >> 32) == (unsigned)(i + 1)) || dead) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > p.guard.spin_limit) give_up(p.guard, 2, lane, dead);
        }
        int y = (int)(unsigned)yv;
        y = min(max(y, 0), C - 1);
        const float* er = p.emb_gi + (long)y * (3 * PD_D) + u;
        eg0 = er[0]; eg1 = er[PD_D]; eg2 = er[2 * PD_D];          // in flight during the context half's MFMAs
      }
    }
    lds_barrier();
    PD_STAMP(6)
    // ---- F: context half (wavefronts 0-3), then the cell update
    if (part == 0) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const f32x4 acc = quarter_job(ga[g], cs, kq, lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) gpart[0][kq][g][lane & 15][(lane >> 4) * 4 + e] = acc[e];
      }
    }
    lds_barrier();
    PD_STAMP(7)
    if (tid < PD_RB * PD_UB) {
      float gi[3], gh[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        gi[g] = (gpart[0][0][g][pr][pu] + gpart[0][1][g][pr][pu]) + (gpart[0][2][g][pr][pu] + gpart[0][3][g][pr][pu]);
        gh[g] = (gpart[1][0][g][pr][pu] + gpart[1][1][g][pr][pu]) + (gpart[1][2][g][pr][pu] + gpart[1][3][g][pr][pu]);
      }
      const float rg = sigmoid_fast((gi[0] + eg0) + (gh[0] + bh0));
      const float zg = sigmoid_fast((gi[1] + eg1) + (gh[1] + bh1));
      const float ng = tanh_fast((gi[2] + eg2) + rg * (gh[2] + bh2));
      const float hn = (1.f - zg) * ng + zg * hprev;
      __hip_atomic_store(hg + ((i + 1) & 1) * PD_GRAN + pr * PD_D + u, granule((unsigned)(i + 1), hn), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    PD_STAMP(8)
    // (no barrier here: the next step's sweep rewrites hs - read above only before barrier B - and its barrier A orders the
    //  gpart / spart reads of this cell update before the next step's writes)
  }
#ifdef GLASS_PL_STAMPS
  if (s_ticket == 0 && (tid == 0 || tid == 256))
    for (int k = 0; k < 12; ++k) reinterpret_cast<unsigned long long*>(p.ctrl + 16)[k + (tid ? 12 : 0)] = st[k];
#endif
}

}  // namespace

extern "C" int64_t glass_decode_persistent_workspace_bytes(int R) {
  const int64_t sets = cdiv(R, PD_RB);
  return (int64_t)PL_CTRL_BYTES + sets * (3 * 2 * PD_GRAN + 2 * PD_RB + 2 * PD_RB * PD_CMAX) * (int64_t)sizeof(unsigned long long);
}

extern "C" int glass_decode_persistent_supported(int T, int D, int C, int max_len) {
  return D == PD_D && T > 0 && T <= PD_T && C > 0 && C <= PD_CMAX && max_len > 0;
}

extern "C" int glass_attention_decode_persistent(const float* x, const float* xproj, const glass_decoder_weights* w,
                                                 const float* sW_rowmajor, const float* emb_gi, const int* roi_image, int R,
                                                 int num_images, int T, int D, int C, int max_len, int eos, float* out,
                                                 int* pred_scratch, int* call_status, void* workspace, int64_t workspace_bytes,
                                                 glass_stream_t stream) {
  GLASS_CHECK_ARG(glass_decode_persistent_supported(T, D, C, max_len),
                  "glass_attention_decode_persistent: needs D=256, T<=32, C<=128 (got D=%d T=%d C=%d)", D, T, C);
  if (R == 0) return GLASS_OK;
  GLASS_CHECK_ARG(x && xproj && w && sW_rowmajor && emb_gi && roi_image && out && pred_scratch && num_images > 0 && workspace,
                  "glass_attention_decode_persistent: null pointer");
  GLASS_CHECK_ARG(workspace_bytes >= glass_decode_persistent_workspace_bytes(R), "glass_attention_decode_persistent: workspace too small");
  GLASS_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "glass_attention_decode_persistent: workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  PdParams p;
  p.x = x; p.xproj = xproj; p.sW = sW_rowmajor; p.sB = w->sB; p.wW = w->wW; p.wB = w->wB; p.emb_gi = emb_gi; p.w_ih = w->w_ih;
  p.w_hh = w->w_hh; p.b_hh = w->b_hh; p.fcW = w->fcW; p.fcB = w->fcB; p.temperature = w->temperature;
  p.R = R; p.T = T; p.C = C; p.max_len = max_len; p.nsets = cdiv(R, PD_RB); p.out = out; p.pred = pred_scratch;
  char* ws = static_cast<char*>(workspace);
  p.ctrl = reinterpret_cast<unsigned*>(ws);
  p.hgran = reinterpret_cast<unsigned long long*>(ws + PL_CTRL_BYTES);
  p.cgran = p.hgran + (size_t)p.nsets * 2 * PD_GRAN;
  p.sgran = p.cgran + (size_t)p.nsets * 2 * PD_GRAN;
  p.ygran = p.sgran + (size_t)p.nsets * 2 * PD_GRAN;
  p.lgran = p.ygran + (size_t)p.nsets * 2 * PD_RB;
  if (!fill_guard(p.guard, call_status)) { glass_set_error("glass_attention_decode_persistent: no status word (hipMalloc failed)"); return GLASS_EHIP; }
  hipError_t e = hipMemsetAsync(workspace, 0, (size_t)glass_decode_persistent_workspace_bytes(R), s);
  if (e != hipSuccess) { glass_set_error("glass_attention_decode_persistent: memset: %s", hipGetErrorString(e)); return GLASS_EHIP; }
  hipLaunchKernelGGL(decode_persistent_kernel, dim3(p.nsets * PD_NS), dim3(PD_THREADS), 0, s, p);
  GLASS_CHECK_LAUNCH("glass_attention_decode_persistent");
  hipLaunchKernelGGL(decode_break_mask_kernel, dim3(num_images), dim3(256), 0, s, pred_scratch, roi_image, R, max_len, C, eos, out);
  GLASS_CHECK_LAUNCH("glass_attention_decode_persistent(mask)");
  return GLASS_OK;
}
