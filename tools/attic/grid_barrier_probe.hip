// grid_barrier_probe.hip - what a phase boundary INSIDE one persistent launch costs on MI355X, against two launches.
// ("two-phase fused MMA kernel: phase 1 = QKV+RoPE GEMM, phase 2 = attention core behind a grid barrier".)
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier_probe tools/grid_barrier_probe.hip && /tmp/grid_barrier_probe
//
// Phase 1: every workgroup writes its slice of a 96 MB buffer (the size of Q + K + V at B8 H32 L655).  Phase 2: every
// workgroup reads the slice of ANOTHER workgroup (one that ran on a different XCD) and folds it into a checksum.
//   two launches      : write kernel, read kernel (stream order is the hand-off)
//   one launch        : write; release (agent scope) ; grid barrier on an atomic counter ; acquire ; read
// Also: the bare grid barrier (no data), per crossing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int THREADS = 256;

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // bounded: nothing guarantees that every workgroup of a plain launch is resident at once (a shared or CU-masked GPU would
    // otherwise spin here for ever and wedge the device) - after ~50 ms of wall clock the barrier gives up and traps
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > 5000000ull) __builtin_trap();      // 100 MHz ticks
    }
  }
  __syncthreads();
}

__global__ void write_kernel(uint4* buf, size_t per_wg) {
  uint4* p = buf + (size_t)blockIdx.x * per_wg;
  for (size_t i = threadIdx.x; i < per_wg; i += THREADS) p[i] = uint4{(unsigned)i, blockIdx.x, 3u, 4u};
}

__global__ void read_kernel(const uint4* buf, size_t per_wg, unsigned* out) {
  const int other = (blockIdx.x + 3) % gridDim.x;          // blocks b and b+8 share an XCD: +3 is another one
  const uint4* p = buf + (size_t)other * per_wg;
  unsigned acc = 0;
  for (size_t i = threadIdx.x; i < per_wg; i += THREADS) { const uint4 v = p[i]; acc += v.x ^ v.y; }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

__global__ void fused_kernel(uint4* buf, size_t per_wg, unsigned* out, unsigned* counter, unsigned target) {
  uint4* p = buf + (size_t)blockIdx.x * per_wg;
  for (size_t i = threadIdx.x; i < per_wg; i += THREADS) p[i] = uint4{(unsigned)i, blockIdx.x, 3u, 4u};
  __threadfence();                                          // release: this XCD's dirty lines must reach memory
  grid_barrier(counter, target);
  __threadfence();                                          // acquire side: drop what this L2 holds of the buffer
  const int other = (blockIdx.x + 3) % gridDim.x;
  const uint4* q = buf + (size_t)other * per_wg;
  unsigned acc = 0;
  for (size_t i = threadIdx.x; i < per_wg; i += THREADS) { const uint4 v = q[i]; acc += v.x ^ v.y; }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

__global__ void barrier_only_kernel(unsigned* counter, int rounds) {
  for (int r = 1; r <= rounds; ++r) grid_barrier(counter, (unsigned)r * gridDim.x);
}

int main() {
  const size_t bytes = 96ull << 20;
  uint4* buf;
  unsigned *out, *counter;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&out, 4096 * 4));
  CK(hipMalloc(&counter, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int grid : {256, 512}) {
    const size_t per_wg = bytes / 16 / grid;
    float best2 = 1e9f, best1 = 1e9f, bestb = 1e9f;
    for (int rep = 0; rep < 7; ++rep) {
      float ms;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(THREADS), 0, 0, buf, per_wg);
      hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(THREADS), 0, 0, buf, per_wg, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best2 = ms < best2 ? ms : best2;
      CK(hipMemset(counter, 0, 4));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(fused_kernel, dim3(grid), dim3(THREADS), 0, 0, buf, per_wg, out, counter, (unsigned)grid);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best1 = ms < best1 ? ms : best1;
      CK(hipMemset(counter, 0, 4));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(barrier_only_kernel, dim3(grid), dim3(THREADS), 0, 0, counter, 100);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) bestb = ms < bestb ? ms : bestb;
    }
    printf("grid %4d x %d threads, 96 MB written then read across XCDs: two launches %7.1f us | one launch + fences + grid barrier %7.1f us | bare grid barrier %5.2f us per crossing\n",
           grid, THREADS, best2 * 1e3f, best1 * 1e3f, bestb * 1e3f / 100.f);
  }
  return 0;
}
