// Micro-benchmark: sustained f32 matrix-core rate with operands held in registers (no memory traffic in
// the loop).  Run bare it shows the clock the power manager sustains under MFMA load; run under
// `rocprofv3 --pmc ...` (which pins the profiling clocks) it shows the 2.4 GHz figure.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(512, 2) void mfma_loop(const float* a, const float* b, float* out, int iters,
                                                    long long* clk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const long long c0 = clock64(), w0 = wall_clock64();   // shader-clock counter vs constant-rate counter
  float av[8], bv[8];
  for (int i = 0; i < 8; ++i) { av[i] = a[(t * 8 + i) & 0xFFFFF]; bv[i] = b[(t * 8 + i) & 0xFFFFF]; }
  float s = 0.f;
  if constexpr (KIND == 0) {            // 32x32x2: 4096 FLOP / instruction
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k], bv[k], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k], bv[(k + 1) & 7], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(k + 1) & 7], bv[k], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(k + 3) & 7], bv[(k + 2) & 7], acc[3], 0, 0, 0);
      }
    }
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {                              // 16x16x4: 2048 FLOP / instruction, 2 per 32x32x2 slot
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(k + j) & 7], bv[(k + 3 * j) & 7], acc[j], 0, 0, 0);
      }
    }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[t] = s;
  if (t == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
int main() {
  const int N = 1 << 20, blocks = 512, thr = 512, iters = 4000;
  std::vector<float> h(N);
  float *a, *b, *o;
  long long* clk;
  CK(hipMalloc(&clk, 16));
  int wall_khz = 0;
  CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
  CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&o, blocks * thr * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[3] = {"zeros", "half zeros", "dense random"};
  for (int round = 0; round < 6; ++round)
    for (int mode = 1; mode < 3; ++mode) {
      const int kind = (round & 1) ^ 1;   // interleave the two instructions: rules out warm-up order effects
      srand(1);
      for (int i = 0; i < N; ++i) {
        float r = (rand() / (float)RAND_MAX) * 2.f - 1.f;
        h[i] = mode == 0 ? 0.f : (mode == 1 ? (r > 0 ? r : 0.f) : r) * 0.01f;
      }
      CK(hipMemcpy(a, h.data(), N * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(b, h.data(), N * 4, hipMemcpyHostToDevice));
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int l = 0; l < 10; ++l) {
          if (kind == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(thr), 0, 0, a, b, o, iters, clk);
          else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(thr), 0, 0, a, b, o, iters, clk);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double fl = 10.0 * blocks * (thr / 64) * (double)iters * (kind == 0 ? 32 * 4096.0 : 64 * 2048.0);
        long long hc[2];
        CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
        const double mhz = (double)hc[0] / ((double)hc[1] / wall_khz) / 1e3;
        if (rep) printf("%-10s %-14s %8.2f ms  %7.1f TFLOP/s  shader clock %6.0f MHz\n", kind == 0 ? "32x32x2" : "16x16x4", names[mode], ms, fl / ms / 1e9, mhz);
      }
    }
  return 0;
}
