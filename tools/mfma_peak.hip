// mfma_peak.hip - bare MFMA issue-rate microbenchmark (random operands in registers), to calibrate the practical
// bf16 MFMA ceiling of this device/clock: waves per SIMD, 32x32x16 vs 16x16x32, accumulators in flight, barrier period.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int NACC, int BAR>   // BAR: barrier every BAR iterations (0 = never); each iteration = NACC MFMAs
__global__ __launch_bounds__(512) void k32(const bf16x8* in, float* out, int iters) {
  bf16x8 a = in[threadIdx.x], b = in[threadIdx.x + 512];
  f32x16 acc[NACC];
_Pragma("unroll") for (int i = 0; i < NACC; ++i) _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    if (BAR && (it % BAR) == BAR - 1) __builtin_amdgcn_s_barrier();
  }
  float s = 0; _Pragma("unroll") for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k16(const bf16x8* in, float* out, int iters) {
  bf16x8 a = in[threadIdx.x], b = in[threadIdx.x + 512];
  f32x4 acc[NACC];
_Pragma("unroll") for (int i = 0; i < NACC; ++i) _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0; _Pragma("unroll") for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int BAR>
__global__ __launch_bounds__(512) void k16b(const bf16x8* in, float* out, int iters) {
  bf16x8 a = in[threadIdx.x], b = in[threadIdx.x + 512];
  f32x4 acc[NACC];
_Pragma("unroll") for (int i = 0; i < NACC; ++i) _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    if (BAR && (it % BAR) == BAR - 1) __builtin_amdgcn_s_barrier();
  }
  float s = 0; _Pragma("unroll") for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  f(); CHECK(hipDeviceSynchronize());
  float best = 1e9;
  for (int r = 0; r < 5; ++r) { CHECK(hipEventRecord(e0)); f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  return best;
}
int main() {
  bf16x8* in; float* out;
  CHECK(hipMalloc(&in, 1024 * 16)); CHECK(hipMalloc(&out, 256 * 8 * 1024 * 4));
  unsigned short h[8192]; srand(3); for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  const int iters = 4000;
  auto rep = [&](const char* n, float ms, double mf, int threads, int blocks) {
    double fl = (double)blocks * (threads / 64) * iters * mf;
    printf("%-44s %8.3f ms  %8.1f TF/s\n", n, ms, fl / ms / 1e9);
  };
  const double F32 = 32.0 * 32 * 16 * 2, F16 = 16.0 * 16 * 32 * 2;
  rep("32x32x16 4acc 1 wave/SIMD (256 thr x 256)", timeit([&] { hipLaunchKernelGGL((k32<4, 0>), dim3(256), dim3(256), 0, 0, in, out, iters); }), 4 * F32, 256, 256);
  rep("32x32x16 8acc 1 wave/SIMD", timeit([&] { hipLaunchKernelGGL((k32<8, 0>), dim3(256), dim3(256), 0, 0, in, out, iters); }), 8 * F32, 256, 256);
  rep("32x32x16 8acc 2 waves/SIMD (512 thr x 256)", timeit([&] { hipLaunchKernelGGL((k32<8, 0>), dim3(256), dim3(512), 0, 0, in, out, iters); }), 8 * F32, 512, 256);
  rep("32x32x16 8acc 2 w/SIMD barrier/4 iters(32 mfma)", timeit([&] { hipLaunchKernelGGL((k32<8, 4>), dim3(256), dim3(512), 0, 0, in, out, iters); }), 8 * F32, 512, 256);
  rep("32x32x16 8acc 2 w/SIMD barrier/1 iter(8 mfma)", timeit([&] { hipLaunchKernelGGL((k32<8, 1>), dim3(256), dim3(512), 0, 0, in, out, iters); }), 8 * F32, 512, 256);
  rep("16x16x32 8acc 1 wave/SIMD", timeit([&] { hipLaunchKernelGGL((k16<8>), dim3(256), dim3(256), 0, 0, in, out, iters); }), 8 * F16, 256, 256);
  rep("16x16x32 16acc 2 waves/SIMD", timeit([&] { hipLaunchKernelGGL((k16<16>), dim3(256), dim3(512), 0, 0, in, out, iters); }), 16 * F16, 512, 256);
  rep("16x16x32 32acc 2 w/SIMD barrier/2 it (64 mfma)", timeit([&] { hipLaunchKernelGGL((k16b<32, 2>), dim3(256), dim3(512), 0, 0, in, out, iters); }), 32 * F16, 512, 256);
  rep("16x16x32 32acc 2 w/SIMD no barrier", timeit([&] { hipLaunchKernelGGL((k16b<32, 0>), dim3(256), dim3(512), 0, 0, in, out, iters); }), 32 * F16, 512, 256);
  rep("32x32x16 8acc 2 w/SIMD, 2 blocks/CU (512 blk)", timeit([&] { hipLaunchKernelGGL((k32<8, 0>), dim3(512), dim3(256), 0, 0, in, out, iters); }), 8 * F32, 256, 512);
  return 0;
}
