// fp8_probe.hip - exact-integer check of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit block scales,
// loading both operands as "lane (r = lane&15, g = lane>>4) holds bytes [32g, 32g+32) of row r" (K-contiguous rows).
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_probe.hip -o build_lab/fp8_probe && build_lab/fp8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(const uint8_t* A, const uint8_t* B, float* C) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  v8i a = *(const v8i*)(A + r * 128 + g * 32);
  v8i b = *(const v8i*)(B + r * 128 + g * 32);
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = c[i];      // row = 4*(lane>>4)+reg (A index), col = lane&15 (B index)
}

__global__ void cvt(const float* x, uint8_t* y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 < n) {
    const int p = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
    y[2 * i] = p & 0xff; y[2 * i + 1] = (p >> 8) & 0xff;
  }
}

int main() {
  float ha[16 * 128], hb[16 * 128];
  srand(1);
  for (int i = 0; i < 16 * 128; ++i) { ha[i] = (float)(rand() % 7 - 3); hb[i] = (float)(rand() % 9 - 4) * 0.5f; }
  float *da, *db, *dc; uint8_t *qa, *qb;
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dc, 256 * 4); hipMalloc(&qa, 2048); hipMalloc(&qb, 2048);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  cvt<<<8, 128>>>(da, qa, 2048); cvt<<<8, 128>>>(db, qb, 2048);
  probe<<<1, 64>>>(qa, qb, dc);
  float hc[256]; uint8_t hq[2048];
  hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost); hipMemcpy(hq, qa, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    float ref = 0; for (int k = 0; k < 128; ++k) ref += ha[i * 128 + k] * hb[j * 128 + k];
    if (ref != hc[i * 16 + j]) { if (bad < 5) printf("mismatch C[%d][%d] = %g, want %g\n", i, j, hc[i * 16 + j], ref); ++bad; }
  }
  printf("fp8 encodings of -3..3: "); for (int v = -3; v <= 3; ++v) { for (int i = 0; i < 2048; ++i) if (ha[i] == v) { printf("%d->0x%02x ", v, hq[i]); break; } } printf("\n");
  printf("%s (%d mismatches of 256)\n", bad ? "FAIL" : "PASS", bad);
  return bad != 0;
}
