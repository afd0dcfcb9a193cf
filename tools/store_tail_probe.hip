// store_tail_probe.hip - how long a workgroup's 128 KB output tile (256 rows x 512 B, row pitch 6144 B) takes to leave a CU,
// by the shape of the store instruction: 16 rows x 64 B (what the MFMA fragment layout gives with 16-byte stores), 4 rows x 256 B,
// 2 rows x 512 B (whole rows, as after a transposition through LDS), and 64 rows x 16 B.  512 threads, 16 x global_store_dwordx4 per lane.
// Cycles from the first store to the return of s_waitcnt vmcnt(0) (wave 0 of workgroup 0; every wave drains before the kernel ends).
//   hipcc --offload-arch=gfx950 -O3 -o store_tail_probe tools/store_tail_probe.hip && ./store_tail_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(512) void k(char* y, long long* out, int tiles_n, size_t pitch) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  char* base = y + (size_t)tm * 256 * pitch + (size_t)tn * 512;
  u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
  __syncthreads();
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int row, col;
    if (PAT == 0) {        // 16 rows x 64 B: wave (wn = wave&1, wm = wave>>1): rows wm*64 + (i>>2)*16 + (lane&15), 64-B group (i&3) of the wave's 256 B
      row = (wave >> 1) * 64 + (i >> 2) * 16 + (lane & 15); col = (wave & 1) * 256 + (i & 3) * 64 + (lane >> 4) * 16;
    } else if (PAT == 1) { // 4 rows x 256 B
      row = wave * 32 + (i >> 1) * 4 + (lane >> 4); col = (i & 1) * 256 + (lane & 15) * 16;
    } else if (PAT == 2) { // 2 rows x 512 B
      row = wave * 32 + i * 2 + (lane >> 5); col = (lane & 31) * 16;
    } else {               // 64 rows x 16 B
      row = (wave & 3) * 64 + lane; col = (wave >> 2) * 256 + i * 16;
    }
    *(u32x4*)(base + (size_t)row * pitch + col) = v;
  }
  long long t1 = clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t2 = clock64();
  __syncthreads();
  long long t3 = clock64();
  if (blockIdx.x == 0 && tid == 0) { out[0] = t1 - t0; out[1] = t2 - t0; out[2] = t3 - t0; }
}
int main() {
  const size_t pitch = 6144;  // 3072 bf16
  const int tiles_m = 21, tiles_n = 12;
  char* y; long long* out;
  hipMalloc(&y, (size_t)tiles_m * 256 * pitch); hipMalloc(&out, 64);
  long long h[3];
  const char* names[4] = {"16 rows x 64 B", "4 rows x 256 B", "2 rows x 512 B", "64 rows x 16 B"};
  for (int grid : {1, 252}) {
    for (int pat = 0; pat < 4; ++pat) {
      for (int rep = 0; rep < 3; ++rep) {
        if (pat == 0) k<0><<<grid, 512>>>(y, out, tiles_n, pitch);
        if (pat == 1) k<1><<<grid, 512>>>(y, out, tiles_n, pitch);
        if (pat == 2) k<2><<<grid, 512>>>(y, out, tiles_n, pitch);
        if (pat == 3) k<3><<<grid, 512>>>(y, out, tiles_n, pitch);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
      printf("%3d workgroups, %-15s: issued after %6lld cycles, wave 0 drained after %6lld, workgroup after %6lld\n", grid, names[pat], h[0], h[1], h[2]);
    }
  }
  return 0;
}
