// Micro-benchmark: does work of the PARTNER wave on a SIMD hide behind v_mfma_f32_16x16x4_f32 of the other wave?
// The f32 matrix instructions run at the f32 vector rate (64 FLOP/clk/SIMD); if they occupy the vector ALUs, a partner's VALU
// instructions are additive to the matrix time, not hidden -- which decides how the Winograd kernels must split their work.
// 512-thread blocks = 2 waves per SIMD: waves 0-3 issue MFMAs (64 per iteration, 16 independent accumulators), waves 4-7 run
// the partner body.  Reported: matrix waves alone, partner alone, both -- "both" near max(...) = hidden, near the sum = additive.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_coissue.hip -o tools/ubench/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// PARTNER: 1 v_add_f32, 2 v_pk_add_f32, 3 v_fma_f32, 4 ds_write_b64 + ds_read_b64, 5 MFMA as well, 6 ds_read_b64 only,
// 7 scalar ALU (s_add / s_mul), 8 global_load_dword (L2 hits, 8 in flight)
template <int PARTNER>
__global__ __launch_bounds__(512) void coissue(const float* a, float* out, int iters, int run_mm, int run_partner, int per_iter, int swap, int prio) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int t = blockIdx.x * blockDim.x + tid;
  float av[8];
  for (int i = 0; i < 8; ++i) av[i] = a[(t * 8 + i) & 0xFFFFF];
  float s = 0.f;
  const bool mm = swap ? wave >= 4 : wave < 4;   // swap: the matrix waves are the YOUNGER half
  if (!mm && prio) __builtin_amdgcn_s_setprio(3);
  if ((mm && run_mm) || (!mm && run_partner && PARTNER == 5)) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(k + j) & 7], av[(k + 3 * j) & 7], acc[j], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else if (!mm && run_partner) {
    float r0 = av[0], r1 = av[1], r2 = av[2], r3 = av[3], r4 = av[4], r5 = av[5], r6 = av[6], r7 = av[7];
    f32x2 p0 = {av[0], av[1]}, p1 = {av[2], av[3]}, p2 = {av[4], av[5]}, p3 = {av[6], av[7]};
    const int n = iters * per_iter / 8;
    if constexpr (PARTNER == 1) {
      for (int i = 0; i < n; ++i)
        asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %7\n"
                     "v_add_f32 %4, %4, %0\n v_add_f32 %5, %5, %1\n v_add_f32 %6, %6, %2\n v_add_f32 %7, %7, %3\n"
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
    } else if constexpr (PARTNER == 2) {
      for (int i = 0; i < n; ++i)
        asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3\n v_pk_add_f32 %2, %2, %0\n v_pk_add_f32 %3, %3, %1\n"
                     "v_pk_add_f32 %0, %0, %3\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %1\n v_pk_add_f32 %3, %3, %0\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    } else if constexpr (PARTNER == 3) {
      for (int i = 0; i < n; ++i)
        asm volatile("v_fma_f32 %0, %0, %4, %1\n v_fma_f32 %1, %1, %5, %2\n v_fma_f32 %2, %2, %6, %3\n v_fma_f32 %3, %3, %7, %0\n"
                     "v_fma_f32 %4, %4, %0, %5\n v_fma_f32 %5, %5, %1, %6\n v_fma_f32 %6, %6, %2, %7\n v_fma_f32 %7, %7, %3, %4\n"
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
    } else if constexpr (PARTNER == 7) {      // scalar ALU: wave-uniform address arithmetic
      int s0 = __builtin_amdgcn_readfirstlane(tid), s1 = 3, s2 = 5, s3 = 7;
      for (int i = 0; i < n; ++i)
        asm volatile("s_add_i32 %0, %0, %1\n s_mul_i32 %1, %1, %2\n s_add_i32 %2, %2, %3\n s_sub_i32 %3, %3, %0\n"
                     "s_add_i32 %0, %0, %2\n s_mul_i32 %1, %1, %3\n s_add_i32 %2, %2, %0\n s_sub_i32 %3, %3, %1\n"
                     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
      r0 += (float)(s0 + s1 + s2 + s3);
    } else if constexpr (PARTNER == 8) {      // vector memory: L2-resident dword loads, 8 in flight
      const float* q = a + (tid & 255);
      for (int i = 0; i < n; ++i) {
        float t0, t1, t2, t3, t4, t5, t6, t7;
        asm volatile("global_load_dword %0, %8, off\n global_load_dword %1, %8, off offset:1024\n global_load_dword %2, %8, off offset:2048\n"
                     "global_load_dword %3, %8, off offset:3072\n global_load_dword %4, %8, off offset:64\n global_load_dword %5, %8, off offset:1088\n"
                     "global_load_dword %6, %8, off offset:2112\n global_load_dword %7, %8, off offset:3136\n s_waitcnt vmcnt(0)\n"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(q) : "memory");
        r0 += t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7;
      }
    } else if constexpr (PARTNER == 4 || PARTNER == 6) {
      float* q = lds + (tid & 255) * 2;
      for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (PARTNER == 4) *reinterpret_cast<volatile f32x2*>(q + j * 1024) = p0;
          const f32x2 v = *reinterpret_cast<volatile f32x2*>(q + ((j + 1) & 3) * 1024);
          p1 += v;
        }
      }
    }
    s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
  }
  out[t] = s;
}

int g_swap = 0, g_prio = 0;
template <int P>
float run(const float* a, float* o, int iters, int mmf, int pf, int per_iter) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int l = 0; l < 2; ++l) hipLaunchKernelGGL(coissue<P>, dim3(512), dim3(512), 0, 0, a, o, iters, mmf, pf, per_iter, g_swap, g_prio);
  hipEventRecord(e0);
  for (int l = 0; l < 5; ++l) hipLaunchKernelGGL(coissue<P>, dim3(512), dim3(512), 0, 0, a, o, iters, mmf, pf, per_iter, g_swap, g_prio);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / 5 * 1e3f;
}

int main() {
  const int N = 1 << 20, iters = 400;
  std::vector<float> h(N);
  srand(1);
  for (int i = 0; i < N; ++i) h[i] = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.01f;
  float *a, *o;
  CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&o, 512 * 512 * 4));
  CK(hipMemcpy(a, h.data(), N * 4, hipMemcpyHostToDevice));
  const char* names[9] = {"", "v_add_f32", "v_pk_add_f32", "v_fma_f32", "ds_write_b64+ds_read_b64", "mfma (partner too)", "ds_read_b64", "s_add/s_mul (SALU)", "global_load_dword x8"};
  // matrix waves: 2 blocks per CU in sequence x iters x 64 MFMAs x 32 cycles
  printf("ideal matrix time at 2.4 GHz: %.1f us\n", 2.0 * iters * 64 * 32 / 2400.0);
  for (int cfg = 0; cfg < 1; ++cfg) {
  g_swap = cfg & 1; g_prio = cfg >> 1;
  printf("==== matrix waves are the %s half of the block, partner at s_setprio %d\n", g_swap ? "YOUNGER (4-7)" : "OLDER (0-3)", g_prio ? 3 : 0);
  for (int per_iter = 128; per_iter <= 128; per_iter *= 2) {
    printf("-- partner issues %d instructions per 64 MFMAs\n", per_iter);
    for (int p = 1; p <= 8; ++p) {
      float ta, tc, tb;
      switch (p) {
        case 1: ta = run<1>(a, o, iters, 1, 0, per_iter); tc = run<1>(a, o, iters, 0, 1, per_iter); tb = run<1>(a, o, iters, 1, 1, per_iter); break;
        case 2: ta = run<2>(a, o, iters, 1, 0, per_iter); tc = run<2>(a, o, iters, 0, 1, per_iter); tb = run<2>(a, o, iters, 1, 1, per_iter); break;
        case 3: ta = run<3>(a, o, iters, 1, 0, per_iter); tc = run<3>(a, o, iters, 0, 1, per_iter); tb = run<3>(a, o, iters, 1, 1, per_iter); break;
        case 4: ta = run<4>(a, o, iters, 1, 0, per_iter); tc = run<4>(a, o, iters, 0, 1, per_iter); tb = run<4>(a, o, iters, 1, 1, per_iter); break;
        case 5: ta = run<5>(a, o, iters, 1, 0, per_iter); tc = run<5>(a, o, iters, 0, 1, per_iter); tb = run<5>(a, o, iters, 1, 1, per_iter); break;
        case 7: ta = run<7>(a, o, iters, 1, 0, per_iter); tc = run<7>(a, o, iters, 0, 1, per_iter); tb = run<7>(a, o, iters, 1, 1, per_iter); break;
        case 8: ta = run<8>(a, o, iters, 1, 0, per_iter); tc = run<8>(a, o, iters, 0, 1, per_iter); tb = run<8>(a, o, iters, 1, 1, per_iter); break;
        default: ta = run<6>(a, o, iters, 1, 0, per_iter); tc = run<6>(a, o, iters, 0, 1, per_iter); tb = run<6>(a, o, iters, 1, 1, per_iter); break;
      }
      printf("%-26s matrix alone %8.1f us   partner alone %8.1f us   both %8.1f us   (max %.1f, sum %.1f)\n", names[p], ta, tc, tb,
             ta > tc ? ta : tc, ta + tc);
    }
  }
  }
  return 0;
}
