// What does a wave-wide buffer_store_dwordx4 (1 KiB) cost when its 64 lanes cover 16 half cache lines (64 B each, the F(4x4)
// epilogue's pattern: 16 tiles x 4 lanes x 16 B) vs 8 full lines (128 B each) vs 4 x 256 B?  One workgroup of 4 waves per CU, 32 stores
// (or loads) per wave back to back, s_memtime around them incl. the drain.   hipcc --offload-arch=gfx950 -O3 store_lines.hip -o store_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int SEG, bool LOAD>   // SEG = contiguous bytes per group of lanes: 64 (4 lanes), 128 (8 lanes), 256 (16 lanes)
__global__ __launch_bounds__(256, 1) void k(float* buf, unsigned bytes, unsigned long long* out, long row_stride) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int LPG = SEG / 16;                       // lanes per contiguous segment
  const int grp = lane / LPG, sub = lane % LPG;       // 64 / LPG segments per instruction
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)bytes, 0x00020000);
  // segment g of instruction i of wave wv of block b: its own "pixel row": far apart (row_stride bytes), like tiles 4 pixels apart
  const unsigned base = (unsigned)(((long)blockIdx.x * 4 + wv) * 32 * 64 / LPG * 0 + ((long)(blockIdx.x * 4 + wv) * (64 / LPG) + grp) * row_stride + sub * 16);
  u32x4 v = {1u, 2u, 3u, (unsigned)tid};
  u32x4 acc = {0, 0, 0, 0};
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    if constexpr (LOAD) acc += __builtin_amdgcn_raw_buffer_load_b128(r, base + i * SEG, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(v, r, base + i * SEG, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (LOAD && acc.x == 0xdeadbeef) buf[0] = 1.f;
  if (lane == 0) out[blockIdx.x * 4 + wv] = t1 - t0;
}
template <int SEG, bool LOAD> void run(float* buf, unsigned bytes, unsigned long long* d, int blocks, const char* what) {
  // a segment row holds 32 instructions x SEG bytes; rows 16 KiB apart
  const long row_stride = 16384;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<SEG, LOAD>), dim3(blocks), dim3(256), 0, 0, buf, bytes, d, row_stride);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto x : h) s += (double)x;
  printf("%-6s %3d-byte segments (%2d lines per instruction), %4d workgroups: %8.0f cycles per wave for 32 instructions = %6.1f per instruction\n",
         what, SEG, SEG >= 128 ? 1024 / SEG * (SEG / 128) : 16, blocks, s / h.size(), s / h.size() / 32);
}
int main() {
  const unsigned bytes = 1u << 30;
  float* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
  unsigned long long* d; hipMalloc(&d, 8 * 4 * 4096);
  for (int blocks : {32, 256}) {
    run<64, false>(buf, bytes, d, blocks, "store"); run<128, false>(buf, bytes, d, blocks, "store"); run<256, false>(buf, bytes, d, blocks, "store");
    run<64, true>(buf, bytes, d, blocks, "load"); run<128, true>(buf, bytes, d, blocks, "load"); run<256, true>(buf, bytes, d, blocks, "load");
  }
  return 0;
}
