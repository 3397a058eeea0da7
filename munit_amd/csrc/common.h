// Shared host/device helpers for libmunit_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "../../include/munit_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// bf16 storage (MUNIT_DTYPE_BF16 tensors): activations of the bf16 mode live in HBM as bf16, every statistic,
// accumulator, bias, parameter and loss stays fp32.  Conversions: plain casts (v_cvt_pk_bf16_f32, round-to-nearest-even,
// NaN stays NaN).
typedef __bf16 bf16_t;
typedef __bf16 bf16v4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));

// 4 consecutive elements of a float / bf16 tensor as fp32, and back
__device__ inline f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ inline f32x4 ld4(const bf16_t* p) {
  const bf16v4 h = *reinterpret_cast<const bf16v4*>(p);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
__device__ inline void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ inline void st4(bf16_t* p, f32x4 v) {
  *reinterpret_cast<bf16v4*>(p) = bf16v4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
}
__device__ inline float ld1(const float* p) { return *p; }
__device__ inline float ld1(const bf16_t* p) { return (float)*p; }
__device__ inline void st1(float* p, float v) { *p = v; }
__device__ inline void st1(bf16_t* p, float v) { *p = (bf16_t)v; }

void munit_set_error(const char* fmt, ...);

#define MUNIT_CHECK_ARG(cond, ...)         \
  do {                                     \
    if (!(cond)) {                         \
      munit_set_error(__VA_ARGS__);        \
      return MUNIT_ERR_ARG;                \
    }                                      \
  } while (0)

#define MUNIT_CHECK_LAUNCH(what)                                                   \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) {                                                        \
      munit_set_error("%s: launch failed: %s", what, hipGetErrorString(e_));       \
      return MUNIT_ERR_LAUNCH;                                                     \
    }                                                                              \
  } while (0)

// Debug switches (MUNIT_DEBUG_*) are read ONCE per process: getenv on every launch path is host time that all eight
// data-parallel ranks pay on every one of ~5 k launches per step.
#define MUNIT_ENV_FLAG(name) ([]() -> bool { static const bool v_ = getenv(name) != nullptr; return v_; }())

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Wave64 sum (all lanes get the total).
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ inline float apply_act(float v, int act, float slope) {
  if (act == MUNIT_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == MUNIT_ACT_LRELU) return v > 0.f ? v : v * slope;
  if (act == MUNIT_ACT_TANH) return tanhf(v);
  return v;
}

// Map a coordinate of the padded (and optionally x2-upsampled) domain back to the source.
// v is in upsampled-domain coordinates (may be out of range); returns source index or -1.
__device__ inline int src_coord(int v, int Hu, int ups, int reflect) {
  // branch-free: reflect about the edges (|v| and 2Hu-2-v), or -1 outside in zero-pad mode
  const bool inside = v >= 0 && v < Hu;
  int r = v < 0 ? -v : v;
  r = r >= Hu ? 2 * Hu - 2 - r : r;
  r = reflect ? r : (inside ? v : -1);
  return r < 0 ? -1 : (r >> ups);
}

// conv_igemm.hip: executed (not algorithmic) FLOPs of the forward / backward-data pass of a layer
double munit_igemm_executed_flops(const munit_conv_desc* d, int pass);
const char* munit_igemm_kernel_name(const munit_conv_desc* d, int pass);

// conv_small.hip: channel-per-lane kernels for convolutions with 3 channels on one side
bool munit_small_fwd_supported(const munit_conv_desc* d);
bool munit_small_wgrad_supported(const munit_conv_desc* d);
// x is read as d->in_dtype (fp32 or bf16); y is always fp32 (3 channels); ws: munit_small_fwd_workspace(d) bytes
size_t munit_small_fwd_workspace(const munit_conv_desc* d);
int munit_small_fwd(const munit_conv_desc* d, int Ho, int Wo, const void* x, const float* w, const float* bias,
                    float* y, void* ws, hipStream_t st);
size_t munit_small_wgrad_workspace(const munit_conv_desc* d, int Ho);
// x is read as d->in_dtype; dy (3 channels) is fp32
int munit_small_wgrad(const munit_conv_desc* d, int Ho, int Wo, const void* x, const float* dy, float* dw,
                      float* db, float beta, void* ws, hipStream_t st);
