// Probe of v_mfma_f32_4x4x1_16B_f32 on gfx950: (1) lane / register map, checked against the hypothesis the conv head
// kernel relies on -- lane l: block = l / 4; A supplies row i = l % 4, B supplies column j = l % 4; D[i][j] of the
// block sits in register i of lane 4 * block + j -- with asymmetric exact-integer data; (2) issue rate.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma4x4.hip -o tools/ubench/mfma4x4 && tools/ubench/mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float* out) {
  const int l = threadIdx.x, blk = l >> 2, r = l & 3;
  const float a = (float)(1 + r + 10 * blk);        // A_blk[i = r]
  const float b = (float)(100 + 7 * r + 1000 * blk);  // B_blk[j = r]
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = c[e];
}

__global__ void rate(float* out, int iters) {
  f32x4 c[8];
  for (int k = 0; k < 8; ++k) c[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += c[k][0] + c[k][1] + c[k][2] + c[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* d;
  hipMalloc(&d, 1 << 24);
  probe<<<1, 64>>>(d);
  std::vector<float> h(256);
  hipMemcpy(h.data(), d, 256 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 4; ++e) {
      const int blk = l >> 2, j = l & 3, i = e;
      const float want = (float)(1 + i + 10 * blk) * (float)(100 + 7 * j + 1000 * blk);
      if (h[l * 4 + e] != want) ++bad;
    }
  printf("layout hypothesis (D[i][j] of block l/4 in register i of lane 4*blk+j; A row i = l%%4, B col j = l%%4): %s (%d mismatches)\n",
         bad ? "FAIL" : "PASS", bad);
  if (bad) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  const int iters = 20000, blocks = 256 * 8, thr = 256;
  rate<<<blocks, thr>>>(d, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  rate<<<blocks, thr>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * (thr / 64) * iters * 8 * 2.0 * 256;   // 16 blocks x 4x4x1 MACs per instruction
  printf("4x4x1_16B rate: %.1f TFLOP/s (%.3f ms)\n", flop / ms / 1e9, ms);
  return bad ? 1 : 0;
}
