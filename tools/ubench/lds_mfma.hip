// Micro-benchmark: the inner loop of conv_igemm_kernel in isolation -- per 32-wide K-tile and wave, 12
// ds_read_b128 operand fetches from a [row][36] LDS tile and 64 v_mfma_f32_16x16x4_f32 -- without global
// loads, LDS writes or barriers.  Variants order the reads differently against the MFMAs.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_mfma.hip -o tools/ubench/lds_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int LD = 36;

template <int VARIANT>
__global__ __launch_bounds__(512, 4) void loop_kernel(const float* src, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float As[2 * 128 * LD];
  __shared__ __attribute__((aligned(16))) float Bs[2 * 128 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 128 * LD; i += 512) { As[i] = src[i & 0xFFFF]; Bs[i] = src[(i + 77) & 0xFFFF]; }
  __syncthreads();
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fk = (lane >> 4) * 4;
  f32x4 acc[4][2];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
  const float* Ab = As + (wm * 64 + fr) * LD + fk;
  const float* Bb = Bs + (wn * 32 + fr) * LD + fk;
  auto rd = [&](f32x4* a, f32x4* b, int buf, int kg) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(Bb + buf * 128 * LD + nt * 16 * LD + kg * 16);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(Ab + buf * 128 * LD + mt * 16 * LD + kg * 16);
  };
  auto mm = [&](const f32x4* a, const f32x4* b) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][t], b[nt][t], acc[mt][nt], 0, 0, 0);
  };
  if constexpr (VARIANT == 0) {          // as the kernel: read a half, multiply it
    for (int it = 0; it < iters; ++it) {
      f32x4 a[4], b[2];
      rd(a, b, it & 1, 0); mm(a, b);
      rd(a, b, it & 1, 1); mm(a, b);
    }
  } else if constexpr (VARIANT == 1) {   // double-buffered halves: next half's reads in flight behind the MFMAs
    f32x4 a0[4], b0[2], a1[4], b1[2];
    rd(a0, b0, 0, 0);
    for (int it = 0; it < iters; ++it) {
      rd(a1, b1, it & 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      rd(a0, b0, (it + 1) & 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (VARIANT == 2) {   // no LDS reads at all (operands loaded once): the MFMA-only ceiling
    f32x4 a[4], b[2];
    rd(a, b, 0, 0);
    for (int it = 0; it < iters; ++it) { mm(a, b); mm(a, b); }
  } else {                               // reads only (no MFMA): LDS time alone
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      f32x4 a[4], b[2];
      rd(a, b, it & 1, 0); s += a[0] + a[1] + a[2] + a[3] + b[0] + b[1];
      rd(a, b, it & 1, 1); s += a[0] + a[1] + a[2] + a[3] + b[0] + b[1];
    }
    acc[0][0] = s;
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
  out[blockIdx.x * 512 + tid] = s;
}

int main() {
  const int blocks = 512, iters = 2000;
  float *src, *out;
  CK(hipMalloc(&src, 65536 * 4)); CK(hipMalloc(&out, blocks * 512 * 4));
  static float h[65536];
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[4] = {"read half, multiply half (kernel order)", "double-buffered halves", "MFMA only", "LDS reads only"};
  for (int round = 0; round < 4; ++round)
    for (int v = 0; v < 3; ++v) {
      if (v == 0) {   // operand data: zeros on even rounds, N(0,1)-like random on odd rounds (data-dependent power)
        srand(1);
        for (int i = 0; i < 65536; ++i) h[i] = (round & 1) ? ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 1.7f : 0.f;
        CK(hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice));
        printf("-- operand data: %s\n", (round & 1) ? "random" : "zeros");
      }
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int l = 0; l < 10; ++l) {
          if (v == 0) hipLaunchKernelGGL(loop_kernel<0>, dim3(blocks), dim3(512), 0, 0, src, out, iters);
          if (v == 1) hipLaunchKernelGGL(loop_kernel<1>, dim3(blocks), dim3(512), 0, 0, src, out, iters);
          if (v == 2) hipLaunchKernelGGL(loop_kernel<2>, dim3(blocks), dim3(512), 0, 0, src, out, iters);
          if (v == 3) hipLaunchKernelGGL(loop_kernel<3>, dim3(blocks), dim3(512), 0, 0, src, out, iters);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double fl = 10.0 * blocks * 8 * (double)iters * 64 * 2048.0;
        if (rep) printf("%-44s %8.2f ms  %7.1f TFLOP/s-equivalent\n", names[v], ms, fl / ms / 1e9);
      }
    }
  return 0;
}
