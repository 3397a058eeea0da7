// Small HBM-bound kernels of the MUNIT step: activation backward, the discriminator's
// 3x3/s2 average-pool pyramid, the style encoder's global average pool, the L1 / LSGAN loss
// reductions with their gradient seeds, the fused flat-buffer Adam update and scaling.
#include "common.h"

namespace {

constexpr int NT = 256;

inline int grid_for(long long n, int per_thread = 1) {
  long long b = (n + (long long)NT * per_thread - 1) / ((long long)NT * per_thread);
  return (int)std::max<long long>(1, std::min<long long>(b, 2048));
}

__global__ void act_bwd_kernel(int act, float slope, const float* __restrict__ y, const float* __restrict__ dy,
                               float* __restrict__ dx, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float yy = y[i], g = dy[i];
    float r;
    if (act == MUNIT_ACT_RELU) r = yy > 0.f ? g : 0.f;
    else if (act == MUNIT_ACT_LRELU) r = yy > 0.f ? g : g * slope;
    else if (act == MUNIT_ACT_TANH) r = g * (1.f - yy * yy);
    else r = g;
    dx[i] = r;
  }
}

// AvgPool2d(3, 2, 1, count_include_pad=False): Ho = (H - 1) / 2 + 1
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C,
                                   int Ho, int Wo) {
  const long long total = (long long)B * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    int c = (int)(r % C); r /= C;
    int ow = (int)(r % Wo); r /= Wo;
    int oh = (int)(r % Ho); r /= Ho;
    int b = (int)r;
    float s = 0.f;
    int cnt = 0;
    for (int dh = -1; dh <= 1; ++dh) {
      int ih = 2 * oh + dh;
      if (ih < 0 || ih >= H) continue;
      for (int dw = -1; dw <= 1; ++dw) {
        int iw = 2 * ow + dw;
        if (iw < 0 || iw >= W) continue;
        s += x[(((long long)b * H + ih) * W + iw) * C + c];
        ++cnt;
      }
    }
    y[i] = s / (float)cnt;
  }
}

__device__ inline int pool_cnt(int o, int n) {  // valid taps of window o along an axis of length n
  int lo = 2 * o - 1, hi = 2 * o + 1;
  if (lo < 0) lo = 0;
  if (hi > n - 1) hi = n - 1;
  return hi - lo + 1;
}

__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W,
                                   int C, int Ho, int Wo) {
  const long long total = (long long)B * H * W * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    int c = (int)(r % C); r /= C;
    int iw = (int)(r % W); r /= W;
    int ih = (int)(r % H); r /= H;
    int b = (int)r;
    float s = 0.f;
    // windows oh with |2*oh - ih| <= 1
    for (int oh = (ih) / 2; oh <= (ih + 1) / 2; ++oh) {
      if (oh >= Ho || 2 * oh - 1 > ih) continue;
      for (int ow = (iw) / 2; ow <= (iw + 1) / 2; ++ow) {
        if (ow >= Wo || 2 * ow - 1 > iw) continue;
        s += dy[(((long long)b * Ho + oh) * Wo + ow) * C + c] / (float)(pool_cnt(oh, H) * pool_cnt(ow, W));
      }
    }
    dx[i] = s;
  }
}

// global average pool: y[b][c] = mean_p x[b][p][c]; one block per (b, 64-channel group)
__global__ __launch_bounds__(NT) void gap_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int HW,
                                                     int C) {
  __shared__ float red[NT];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (c < C)
    for (int p = pl; p < HW; p += NT / 64) s += x[((long long)b * HW + p) * C + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    for (int l = 1; l < NT / 64; ++l) s += red[l * 64 + cl];
    y[(long long)b * C + c] = s / (float)HW;
  }
}

__global__ void gap_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int HW, int C, long long total) {
  const float inv = 1.f / (float)HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long b = i / ((long long)HW * C);
    dx[i] = dy[b * C + c] * inv;
  }
}

// ---- deterministic two-stage scalar reductions ----
__device__ inline float block_sum256(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ __launch_bounds__(NT) void l1_partial_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                        const float* __restrict__ mask, long long npix, int C,
                                                        float* __restrict__ partial) {
  __shared__ float red[4];
  const long long n = npix * C;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
    float d = fabsf(ld1(a + i) - ld1(b + i));
    if (mask != nullptr) d *= (1.f - mask[i / C]);
    s += d;
  }
  s = block_sum256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(NT) void mse_partial_kernel(const float* __restrict__ x, float target, long long n,
                                                         float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
    float d = x[i] - target;
    s += d * d;
  }
  s = block_sum256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(NT) void finish_mean_kernel(const float* __restrict__ partial, int nparts, double inv_n,
                                                         float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += NT) s += partial[i];
  s = block_sum256(s, red);
  if (threadIdx.x == 0) out[0] = (float)((double)s * inv_n);
}

template <typename T>
__global__ void l1_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ mask,
                              long long npix, int C, const float* __restrict__ gout, T* __restrict__ da,
                              T* __restrict__ db) {
  const long long n = npix * C;
  const float g = gout[0] / (float)n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float d = ld1(a + i) - ld1(b + i);
    float s = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    if (mask != nullptr) s *= (1.f - mask[i / C]);
    if (da != nullptr) st1(da + i, s);
    if (db != nullptr) st1(db + i, -s);
  }
}

__global__ void mse_bwd_kernel(const float* __restrict__ x, float target, long long n, const float* __restrict__ gout,
                               float* __restrict__ dx) {
  const float g = 2.f * gout[0] / (float)n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dx[i] = g * (x[i] - target);
}

struct WsumArgs {
  const float* t[32];
  float w[32];
  int n;
};
__global__ void weighted_sum_kernel(WsumArgs a, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < a.n; ++i) s += a.w[i] * a.t[i][0];
    out[0] = s;
  }
}

// torch.optim.Adam single-tensor update order (L2 weight decay folded into the gradient).
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float step_size, float beta1, float beta2, float eps,
                            float wd, float bc2_sqrt, float omb1, float omb2) {
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pp = *reinterpret_cast<f32x4*>(p + i * 4);
    f32x4 gg = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 mm = *reinterpret_cast<f32x4*>(m + i * 4);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gr = gg[e] + wd * pp[e];
      mm[e] = mm[e] + (gr - mm[e]) * omb1;
      vv[e] = vv[e] * beta2 + omb2 * gr * gr;
      float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pp[e] = pp[e] - step_size * (mm[e] / denom);
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pp;
    *reinterpret_cast<f32x4*>(m + i * 4) = mm;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
  }
  // tail
  for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float gr = g[i] + wd * p[i];
    float mm = m[i] + (gr - m[i]) * omb1;
    float vv = v[i] * beta2 + omb2 * gr * gr;
    float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mm / denom);
    m[i] = mm;
    v[i] = vv;
  }
}

// ExtraAdam (scripts/extraadam.py:121-168 `update`, :30-84 extrapolation/step): both phases advance the
// moments and return u = -lr*sqrt(bc2)/bc1 * m / (sqrt(v) + eps); mode 0: save p, p += u (first
// extrapolation); mode 1: p += u (further extrapolations); mode 2: p = saved + u (the update step).
__global__ void extraadam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, float* __restrict__ pc, long long n, float step_size,
                                 float beta1, float beta2, float eps, float wd, float omb1, float omb2, int mode) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float pv = p[i];
    const float gr = g[i] + wd * pv;
    const float mm = m[i] * beta1 + omb1 * gr;
    const float vv = v[i] * beta2 + omb2 * gr * gr;
    const float u = -step_size * mm / (sqrtf(vv) + eps);
    m[i] = mm;
    v[i] = vv;
    if (mode == 0) pc[i] = pv;
    p[i] = (mode == 2 ? pc[i] : pv) + u;
  }
}

__global__ void scale_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float alpha, int acc) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = alpha * x[i] + (acc ? y[i] : 0.f);
}

}  // namespace

extern "C" int munit_act_bwd(int act, float slope, const float* y, const float* dy, float* dx, size_t n,
                             munit_stream_t stream) {
  MUNIT_CHECK_ARG(y && dy && dx, "act_bwd: null pointer");
  if (n == 0) return MUNIT_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for((long long)n, 4)), dim3(NT), 0, (hipStream_t)stream, act, slope, y,
                     dy, dx, (long long)n);
  MUNIT_CHECK_LAUNCH("act_bwd");
  return MUNIT_OK;
}

extern "C" int munit_avgpool3s2_fwd(const float* x, float* y, int B, int H, int W, int C, munit_stream_t stream) {
  MUNIT_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0, "avgpool_fwd: bad args");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for((long long)B * Ho * Wo * C)), dim3(NT), 0,
                     (hipStream_t)stream, x, y, B, H, W, C, Ho, Wo);
  MUNIT_CHECK_LAUNCH("avgpool_fwd");
  return MUNIT_OK;
}

extern "C" int munit_avgpool3s2_bwd(const float* dy, float* dx, int B, int H, int W, int C, munit_stream_t stream) {
  MUNIT_CHECK_ARG(dy && dx && B > 0 && H > 0 && W > 0 && C > 0, "avgpool_bwd: bad args");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((long long)B * H * W * C)), dim3(NT), 0, (hipStream_t)stream,
                     dy, dx, B, H, W, C, Ho, Wo);
  MUNIT_CHECK_LAUNCH("avgpool_bwd");
  return MUNIT_OK;
}

extern "C" int munit_gap_fwd(const float* x, float* y, int B, int HW, int C, munit_stream_t stream) {
  MUNIT_CHECK_ARG(x && y && B > 0 && HW > 0 && C > 0, "gap_fwd: bad args");
  hipLaunchKernelGGL(gap_fwd_kernel, dim3(cdiv(C, 64), B), dim3(NT), 0, (hipStream_t)stream, x, y, HW, C);
  MUNIT_CHECK_LAUNCH("gap_fwd");
  return MUNIT_OK;
}

extern "C" int munit_gap_bwd(const float* dy, float* dx, int B, int HW, int C, munit_stream_t stream) {
  MUNIT_CHECK_ARG(dy && dx && B > 0 && HW > 0 && C > 0, "gap_bwd: bad args");
  const long long total = (long long)B * HW * C;
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, dy, dx, HW, C, total);
  MUNIT_CHECK_LAUNCH("gap_bwd");
  return MUNIT_OK;
}

namespace {
constexpr int LOSS_PARTS = 1024;
}

extern "C" size_t munit_loss_workspace_bytes(size_t) { return LOSS_PARTS * sizeof(float); }

namespace {
template <typename T>
int l1_mean_fwd_t(const T* a, const T* b, const float* mask, size_t npix, int C, float* out, void* ws, size_t ws_bytes,
                  munit_stream_t stream) {
  MUNIT_CHECK_ARG(a && b && out && ws && npix > 0 && C > 0, "l1_mean_fwd: bad args");
  MUNIT_CHECK_ARG(ws_bytes >= LOSS_PARTS * sizeof(float), "l1_mean_fwd: workspace too small");
  const long long n = (long long)npix * C;
  const int parts = (int)std::min<long long>(LOSS_PARTS, (n + NT - 1) / NT);
  float* partial = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(l1_partial_kernel<T>, dim3(parts), dim3(NT), 0, (hipStream_t)stream, a, b, mask, (long long)npix,
                     C, partial);
  MUNIT_CHECK_LAUNCH("l1_partial");
  hipLaunchKernelGGL(finish_mean_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, partial, parts, 1.0 / (double)n,
                     out);
  MUNIT_CHECK_LAUNCH("finish_mean");
  return MUNIT_OK;
}
template <typename T>
int l1_mean_bwd_t(const T* a, const T* b, const float* mask, size_t npix, int C, const float* gout, T* da, T* db,
                  munit_stream_t stream) {
  MUNIT_CHECK_ARG(a && b && gout && npix > 0 && C > 0, "l1_mean_bwd: bad args");
  const long long n = (long long)npix * C;
  hipLaunchKernelGGL(l1_bwd_kernel<T>, dim3(grid_for(n, 4)), dim3(NT), 0, (hipStream_t)stream, a, b, mask,
                     (long long)npix, C, gout, da, db);
  MUNIT_CHECK_LAUNCH("l1_bwd");
  return MUNIT_OK;
}
}  // namespace

extern "C" int munit_l1_mean_fwd(const float* a, const float* b, const float* mask, size_t npix, int C, float* out,
                                 void* ws, size_t ws_bytes, munit_stream_t stream) {
  return l1_mean_fwd_t<float>(a, b, mask, npix, C, out, ws, ws_bytes, stream);
}
extern "C" int munit_l1_mean_fwd_bf16(const void* a, const void* b, const float* mask, size_t npix, int C, float* out,
                                      void* ws, size_t ws_bytes, munit_stream_t stream) {
  return l1_mean_fwd_t<bf16_t>(reinterpret_cast<const bf16_t*>(a), reinterpret_cast<const bf16_t*>(b), mask, npix, C, out, ws,
                               ws_bytes, stream);
}
extern "C" int munit_l1_mean_bwd(const float* a, const float* b, const float* mask, size_t npix, int C,
                                 const float* gout, float* da, float* db, munit_stream_t stream) {
  return l1_mean_bwd_t<float>(a, b, mask, npix, C, gout, da, db, stream);
}
extern "C" int munit_l1_mean_bwd_bf16(const void* a, const void* b, const float* mask, size_t npix, int C,
                                      const float* gout, void* da, void* db, munit_stream_t stream) {
  return l1_mean_bwd_t<bf16_t>(reinterpret_cast<const bf16_t*>(a), reinterpret_cast<const bf16_t*>(b), mask, npix, C, gout,
                               reinterpret_cast<bf16_t*>(da), reinterpret_cast<bf16_t*>(db), stream);
}

extern "C" int munit_mse_const_fwd(const float* x, float target, size_t n, float* out, void* ws, size_t ws_bytes,
                                   munit_stream_t stream) {
  MUNIT_CHECK_ARG(x && out && ws && n > 0, "mse_const_fwd: bad args");
  MUNIT_CHECK_ARG(ws_bytes >= LOSS_PARTS * sizeof(float), "mse_const_fwd: workspace too small");
  const int parts = (int)std::min<long long>(LOSS_PARTS, ((long long)n + NT - 1) / NT);
  float* partial = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(mse_partial_kernel, dim3(parts), dim3(NT), 0, (hipStream_t)stream, x, target, (long long)n,
                     partial);
  MUNIT_CHECK_LAUNCH("mse_partial");
  hipLaunchKernelGGL(finish_mean_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, partial, parts, 1.0 / (double)n,
                     out);
  MUNIT_CHECK_LAUNCH("finish_mean");
  return MUNIT_OK;
}

extern "C" int munit_mse_const_bwd(const float* x, float target, size_t n, const float* gout, float* dx,
                                   munit_stream_t stream) {
  MUNIT_CHECK_ARG(x && gout && dx && n > 0, "mse_const_bwd: bad args");
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for((long long)n, 4)), dim3(NT), 0, (hipStream_t)stream, x, target,
                     (long long)n, gout, dx);
  MUNIT_CHECK_LAUNCH("mse_bwd");
  return MUNIT_OK;
}

extern "C" int munit_weighted_sum(const float* const* terms, const float* w, int n, float* out,
                                  munit_stream_t stream) {
  MUNIT_CHECK_ARG(terms && w && out && n > 0 && n <= 32, "weighted_sum: bad args (n=%d)", n);
  WsumArgs a{};
  a.n = n;
  for (int i = 0; i < n; ++i) { a.t[i] = terms[i]; a.w[i] = w[i]; }
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, out);
  MUNIT_CHECK_LAUNCH("weighted_sum");
  return MUNIT_OK;
}

extern "C" int munit_adam_step(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1,
                               double beta2, double eps, double weight_decay, int step, munit_stream_t stream) {
  MUNIT_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adam_step: bad args");
  MUNIT_CHECK_ARG(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                      ((uintptr_t)v % 16 == 0),
                  "adam_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for((long long)n, 4)), dim3(NT), 0, (hipStream_t)stream, p, g, m, v,
                     (long long)n, step_size, (float)beta1, (float)beta2, (float)eps, (float)weight_decay, bc2_sqrt,
                     (float)(1.0 - beta1), (float)(1.0 - beta2));
  MUNIT_CHECK_LAUNCH("adam");
  return MUNIT_OK;
}

extern "C" int munit_extraadam_step(float* p, const float* g, float* m, float* v, float* p_saved, size_t n, double lr,
                                    double beta1, double beta2, double eps, double weight_decay, int step, int mode,
                                    munit_stream_t stream) {
  MUNIT_CHECK_ARG(p && g && m && v && p_saved && n > 0 && step >= 1 && mode >= 0 && mode <= 2, "extraadam_step: bad args");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr * sqrt(bc2) / bc1);
  hipLaunchKernelGGL(extraadam_kernel, dim3(grid_for((long long)n, 4)), dim3(NT), 0, (hipStream_t)stream, p, g, m, v,
                     p_saved, (long long)n, step_size, (float)beta1, (float)beta2, (float)eps, (float)weight_decay,
                     (float)(1.0 - beta1), (float)(1.0 - beta2), mode);
  MUNIT_CHECK_LAUNCH("extraadam");
  return MUNIT_OK;
}

extern "C" int munit_scale(const float* x, float* y, size_t n, float alpha, int accumulate, munit_stream_t stream) {
  MUNIT_CHECK_ARG(x && y, "scale: null pointer");
  if (n == 0) return MUNIT_OK;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for((long long)n, 4)), dim3(NT), 0, (hipStream_t)stream, x, y,
                     (long long)n, alpha, accumulate);
  MUNIT_CHECK_LAUNCH("scale");
  return MUNIT_OK;
}
