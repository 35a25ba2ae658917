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

struct PlParams {
  const float* xg;
  const float* whh;
  float* out;
  unsigned long long* gran;
  unsigned* ctrl;      // [0] start tickets
  int* status;         // library-owned, sticky: bit 0 = an LSTM hand-off gave up, bit 1 = a decoder hand-off gave up
  int R, T, NG, nsets;
};

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
                                             float (*hs)[PL_HD + 4], bool& dead, int* status, int bit) {
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) ok &= (unsigned)(v[j] >> 32) == epoch;
    if (__all(ok) || dead) break;
    __builtin_amdgcn_s_sleep(2);
    if (++spins > PL_SPIN_LIMIT) {          // wavefront-uniform
      dead = true;
      if (lane == 0) atomicOr(status, bit);
    }
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
        sweep_finish(src, (unsigned)s, tid, lane, v, hs, dead, p.status, 1);
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
      if (s + 1 < T)
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
                                                  int dirs_per_workgroup, int groups_per_workgroup, void* workspace,
                                                  int64_t workspace_bytes, glass_stream_t stream) {
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
  p.status = recurrence_status_word();
  if (!p.status) { glass_set_error("glass_bilstm_recurrence_persistent: no status word (hipMalloc failed)"); return GLASS_EHIP; }
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
