// l2_fill_probe.hip - how many bytes per clock a CU pulls out of L2 / the Infinity Cache / HBM through (a) LDS-DMA (global_load_lds, 16 B per
// lane: the path every GEMM tile of this library takes) and (b) plain vector loads into registers (global_load_dwordx4), with the access pattern
// of a GEMM operand tile: a wave instruction = 8 rows x 128 bytes at a row pitch of `pitch` bytes.  Question behind it (EXPERIMENTS.md, round 5):
// the few-row GEMMs move 28-32 KB per K-step and CU and take 900-1000 cycles for it whatever the ring depth - is ~32 B/clk the LDS-DMA ceiling?
//   hipcc --offload-arch=gfx950 -O3 tools/l2_fill_probe.hip -o build_lab/l2_fill_probe && build_lab/l2_fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))

// MODE 0: LDS-DMA, MODE 1: VGPR loads.  Each wave streams `steps` tiles of 8 KB (8 instructions x 1 KB) from its own region; `inflight` tiles are kept
// in flight (ring).  region per workgroup = rows x pitch; footprint decides which level of the hierarchy answers.
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void fill(const char* base, size_t wg_stride, int pitch, int steps, int wrap_steps, unsigned* sink, long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* src = base + (size_t)blockIdx.x * wg_stride + (size_t)(wave * 8 + (lane >> 3)) * pitch + (lane & 7) * 16;   // 8 rows x 128 B per instruction
  const size_t rowblock = (size_t)NW * 8 * pitch;       // the workgroup's next 8*NW rows
  u32x4 acc = {0, 0, 0, 0};
  long long t0 = 0;
  if (threadIdx.x == 0) t0 = clock64();
  for (int s = 0; s < steps; ++s) {
    const int ks = s % wrap_steps;                       // k-step: advances 128 B along the rows; wraps so that the footprint stays `wrap_steps` x tile
    const char* p = src + (size_t)ks * 128;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_global_load_lds(GP(p + (size_t)j * rowblock), LP(smem + ((s & 1) * 8 * NW + j * NW + wave) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");      // up to five tiles in flight per wave (a bandwidth probe: nobody reads the slots)
    } else {
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load((const u32x4*)(p + (size_t)j * rowblock));
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc[0] ^= v[j][0]; acc[1] ^= v[j][1]; acc[2] ^= v[j][2]; acc[3] ^= v[j][3]; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) cycles[blockIdx.x] = clock64() - t0;
  if (MODE == 0) acc[0] = *(unsigned*)(smem + threadIdx.x * 4);
  if (acc[0] == 0x12345 && acc[1] == 7) sink[threadIdx.x] = acc[0] ^ acc[2] ^ acc[3];
}

template <int MODE, int NW>
static void run(const char* name, const char* buf, size_t bytes, int wgs, int pitch, int rows_per_wg, int wrap_steps, int steps, unsigned* sink, long long* cyc) {
  const size_t wg_stride = (size_t)rows_per_wg * pitch;
  if ((size_t)wgs * wg_stride > bytes) { printf("%s: buffer too small\n", name); return; }
  const int smem = MODE == 0 ? 2 * 8 * NW * 1024 : 1024;
  CHECK(hipFuncSetAttribute((const void*)fill<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((fill<MODE, NW>), dim3(wgs), dim3(NW * 64), smem, 0, buf, wg_stride, pitch, steps, wrap_steps, sink, cyc);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((fill<MODE, NW>), dim3(wgs), dim3(NW * 64), smem, 0, buf, wg_stride, pitch, steps, wrap_steps, sink, cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h(wgs);
  CHECK(hipMemcpy(h.data(), cyc, wgs * sizeof(long long), hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto c : h) mean += (double)c;
  mean /= wgs;
  const double bytes_wg = (double)steps * 8 * NW * 1024;
  printf("%-46s %4d wgs x %d waves  %8.1f us  %7.2f TB/s  %6.1f B/clk per workgroup (mean over workgroups)\n", name, wgs, NW, ms * 1e3, bytes_wg * wgs / ms / 1e9, bytes_wg / mean);
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  char* buf; unsigned* sink; long long* cyc;
  CHECK(hipMalloc(&buf, bytes)); CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipMalloc(&sink, 4096)); CHECK(hipMalloc(&cyc, 8192 * 8));
  const int pitch = 6144;                                   // a K = 3072 bf16 operand row
  // footprint per workgroup: rows_per_wg x pitch bytes, k wraps after wrap_steps x 128 B: 4 waves x 8 instr x 8 rows = 256 rows -> 32 KB per step
  printf("== L2-resident: each workgroup re-reads 256 rows x 2 KB (16 k-steps, 512 KB) ==\n");
  run<0, 4>("LDS-DMA  1 wg/CU", buf, bytes, 256, pitch, 256, 16, 512, sink, cyc);
  run<1, 4>("VGPR     1 wg/CU", buf, bytes, 256, pitch, 256, 16, 512, sink, cyc);
  run<0, 4>("LDS-DMA  2 wg/CU", buf, bytes, 512, pitch, 256, 16, 512, sink, cyc);
  run<1, 4>("VGPR     2 wg/CU", buf, bytes, 512, pitch, 256, 16, 512, sink, cyc);
  run<1, 4>("VGPR     4 wg/CU", buf, bytes, 1024, pitch, 256, 16, 512, sink, cyc);
  run<0, 8>("LDS-DMA  1 wg/CU, 8 waves", buf, bytes, 256, pitch, 512, 16, 256, sink, cyc);
  run<1, 8>("VGPR     1 wg/CU, 8 waves", buf, bytes, 256, pitch, 512, 16, 256, sink, cyc);
  printf("== streaming (every byte once: 48 k-steps of 256 rows = the whole K of a 3072-wide operand, 1.5 MB per workgroup) ==\n");
  run<0, 4>("LDS-DMA  2 wg/CU stream", buf, bytes, 512, pitch, 256, 48, 48, sink, cyc);
  run<1, 4>("VGPR     2 wg/CU stream", buf, bytes, 512, pitch, 256, 48, 48, sink, cyc);
  printf("== one workgroup per CU sharing: 18 workgroups read the SAME 512 KB (a GEMM operand panel shared inside an XCD) ==\n");
  run<0, 4>("LDS-DMA  shared panel, 2 wg/CU", buf, bytes, 512, pitch, 0, 16, 512, sink, cyc);
  run<1, 4>("VGPR     shared panel, 2 wg/CU", buf, bytes, 512, pitch, 0, 16, 512, sink, cyc);
  return 0;
}
