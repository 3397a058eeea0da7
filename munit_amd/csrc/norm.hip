// Normalisation layers of the MUNIT generator, NHWC, fp32, HBM-bound:
//   instance norm / AdaIN  (networks.py:657, 810-848)   per-(b,c) stats over H*W
//   MUNIT LayerNorm        (networks.py:851-878)        per-sample stats over C*H*W
// Every pass is two kernels: a partial-statistics kernel over (pixel split, sample) blocks
// -- per-thread fp32 accumulators over a channel quad, wavefront/LDS reduction across the
// pixel lanes of the block -- and an apply kernel that folds the split partials in its
// prologue (kept in LDS) and streams float4s.  Sums are taken relative to a pivot (the
// first element of the plane) so the variance does not cancel catastrophically, and every
// statistic is accumulated in fp64 (free at HBM-bound rates): an fp32 error in a per-channel
// mean is the same for every pixel of the plane and would add up coherently in the weight
// gradients downstream.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_SPLIT = 64;

// Activation behind the normalisation (networks.py:668-681, 695-701 pair any norm with any activation).  The `relu` argument of
// the entry points is an activation code: 0 none, 1 ReLU, 2 LeakyReLU(0.2) (the reference's fixed slope, networks.py:672),
// 3 tanh = MUNIT_ACT_*.  The backward kernels re-evaluate the forward's own pre-activation expression bit for bit and
// differentiate the activation there (ReLU / LeakyReLU: the branch the forward took; tanh: 1 - tanh(pre)^2, the same tanhf).
constexpr float NORM_LRELU_SLOPE = 0.2f;
__device__ inline float norm_act(float v, int act) {
  if (act == MUNIT_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == MUNIT_ACT_LRELU) return v > 0.f ? v : v * NORM_LRELU_SLOPE;
  if (act == MUNIT_ACT_TANH) return tanhf(v);
  return v;
}
__device__ inline float norm_act_grad(float pre, float g, int act) {
  if (act == MUNIT_ACT_RELU) return pre > 0.f ? g : 0.f;
  if (act == MUNIT_ACT_LRELU) return pre > 0.f ? g : g * NORM_LRELU_SLOPE;
  if (act == MUNIT_ACT_TANH) { const float t = tanhf(pre); return g * (1.f - t * t); }
  return g;
}

// layout of the thread block over an NHWC plane: QB channel-quads x PL pixel lanes
struct Lay {
  int CQ;  // C/4
  int QB;  // quads handled side by side (<= 256)
  int PL;  // pixel lanes
};
__host__ __device__ inline Lay make_lay(int C) {
  Lay l;
  l.CQ = C >> 2;
  l.QB = l.CQ < NT ? l.CQ : NT;
  l.PL = NT / l.QB;
  return l;
}

inline int pick_split(int B, int HW) {
  // enough blocks to fill 256 CUs a few times, at least 64 pixels per split
  int s = std::max(1, std::min(MAX_SPLIT, 2048 / std::max(1, B)));
  s = std::min(s, std::max(1, HW / 64));
  return s;
}

// ---------------------------------------------------------------------------------------
// instance norm
// ---------------------------------------------------------------------------------------
// partial[b][split][2][C] : sum(x - pivot), sum((x - pivot)^2), pivot = x[b][0][c]
template <typename T>
__global__ __launch_bounds__(NT) void in_stats_kernel(const T* __restrict__ x, double* __restrict__ partial,
                                                      int HW, int C, int nsplit, int CS) {
  extern __shared__ double smd[];  // [PL][QB*4][2]
  double* sm = smd;
  // blockIdx.z = channel slice of CS channels (CS = C: one slice, the block walks all channel quads)
  const Lay L = make_lay(CS);
  const int b = blockIdx.y, sp = blockIdx.x, q0 = blockIdx.z * (CS >> 2);
  const int q = threadIdx.x % L.QB, pl = threadIdx.x / L.QB;
  const int per = (HW + nsplit - 1) / nsplit;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  const T* xb = x + (long long)b * HW * C;
  for (int qq = q0 + q; qq < q0 + L.CQ; qq += L.QB) {
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (pl < L.PL) {
      const f32x4 piv = ld4(xb + qq * 4);
      for (int p = p0 + pl; p < p1; p += L.PL) {
        f32x4 v = ld4(xb + (long long)p * C + qq * 4) - piv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1[e] += (double)v[e];
          s2[e] += (double)v[e] * (double)v[e];
        }
      }
      double* d = sm + ((pl * L.QB + q) * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { d[e] = s1[e]; d[4 + e] = s2[e]; }
    }
    __syncthreads();
    if (pl == 0) {
      for (int l = 1; l < L.PL; ++l) {
        const double* d = sm + ((l * L.QB + q) * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += d[e]; s2[e] += d[4 + e]; }
      }
      double* o = partial + ((long long)(b * nsplit + sp) * 2) * C + qq * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = s1[e]; o[C + e] = s2[e]; }
    }
    __syncthreads();
  }
}

// Sum over the splits of partial[b][split][2][C] for one (sample, channel) per 4 threads: block = 64 channels x 4 split
// groups (grid: ceil(C/64) x B), every group adds its splits in order, group 0 adds the four group sums in order
// (deterministic) and returns true with the totals.  A thread per (b, c) walking all 64 splits alone was a 19 us chain of
// dependent loads, as long as the apply kernel itself.
__device__ inline bool fold_partials(const double* __restrict__ partial, int B, int C, int nsplit, int& b, int& c,
                                     double& s1, double& s2) {
  __shared__ double red[3][64][2];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  b = blockIdx.y;
  c = blockIdx.x * 64 + cl;
  s1 = 0.0;
  s2 = 0.0;
  if (c < C) {
    for (int k = g; k < nsplit; k += 4) {
      const double* o = partial + ((long long)(b * nsplit + k) * 2) * C;
      s1 += o[c];
      s2 += o[C + c];
    }
  }
  if (g > 0) {
    red[g - 1][cl][0] = s1;
    red[g - 1][cl][1] = s2;
  }
  __syncthreads();
  if (g > 0 || c >= C) return false;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    s1 += red[j][cl][0];
    s2 += red[j][cl][1];
  }
  return true;
}

// Folds the split partials ONCE per (sample, channel): stats[b][c] = (mean, rstd) for the backward and
// coef[b][2][C] = (scale, shift) of y = x * scale + shift for the apply kernel.  (Every apply block used to repeat this
// fold in its prologue -- nsplit * 2 * C doubles per block, as many bytes from L2 as the tensor it then streamed.)
template <typename T>
__global__ __launch_bounds__(NT) void in_finalize_kernel(const T* __restrict__ x, const double* __restrict__ partial,
                                                         float* __restrict__ stats, float* __restrict__ coef, int B,
                                                         int HW, int C, int nsplit, const float* __restrict__ adain,
                                                         int ad_ld, int w_off, int b_off, float eps) {
  double s1, s2;
  int b, c;
  if (!fold_partials(partial, B, C, nsplit, b, c, s1, s2)) return;
  const int i = b * C + c;
  const double inv_n = 1.0 / (double)HW;
  const double d = s1 * inv_n;
  const float mean = (float)((double)ld1(x + (long long)b * HW * C + c) + d);   // pivot = first pixel of the plane
  double var = s2 * inv_n - d * d;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  float w = 1.f, bb = 0.f;
  if (adain != nullptr) {
    w = adain[(long long)b * ad_ld + w_off + c];
    bb = adain[(long long)b * ad_ld + b_off + c];
  }
  stats[(long long)i * 2] = mean;
  stats[(long long)i * 2 + 1] = rstd;
  coef[((long long)b * 2) * C + c] = rstd * w;
  coef[((long long)b * 2 + 1) * C + c] = bb - mean * rstd * w;
}

// y = act(x * scale + shift) + residual
template <typename T>
__global__ __launch_bounds__(NT) void in_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                      const float* __restrict__ coef, int HW, int C, int nsplit,
                                                      const T* __restrict__ residual, int relu) {
  extern __shared__ float sm[];  // scale[C], shift[C]
  float* scale = sm;
  float* shift = sm + C;
  const int b = blockIdx.y, sp = blockIdx.x;
  for (int c = threadIdx.x; c < 2 * C; c += NT) sm[c] = coef[(long long)b * 2 * C + c];
  __syncthreads();
  const int CQ = C >> 2;
  const int per = (HW + nsplit - 1) / nsplit;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  const long long i0 = (long long)p0 * CQ, i1 = (long long)p1 * CQ;
  const long long base = (long long)b * HW * C;
#pragma unroll 4
  for (long long i = i0 + threadIdx.x; i < i1; i += NT) {
    const int q = (int)(i % CQ);
    f32x4 v = ld4(x + base + i * 4);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + q * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + q * 4);
    v = v * sc + sh;
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = norm_act(v[e], relu);
    }
    if (residual != nullptr) v += ld4(residual + base + i * 4);
    st4(y + base + i * 4, v);
  }
}

// backward partials: partial[b][split][2][C] : sum(g), sum(g*xhat), g = dy * relu'(pre)
template <typename T>
__global__ __launch_bounds__(NT) void in_bwd_stats_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const float* __restrict__ stats, double* __restrict__ partial,
                                                          int HW, int C, int nsplit, const float* __restrict__ adain,
                                                          int ad_ld, int w_off, int b_off, int relu, int CS) {
  extern __shared__ double smd[];
  double* sm = smd;
  const Lay L = make_lay(CS);   // blockIdx.z = channel slice (see in_stats_kernel)
  const int b = blockIdx.y, sp = blockIdx.x, q0 = blockIdx.z * (CS >> 2);
  const int q = threadIdx.x % L.QB, pl = threadIdx.x / L.QB;
  const int per = (HW + nsplit - 1) / nsplit;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  const long long base = (long long)b * HW * C;
  for (int qq = q0 + q; qq < q0 + L.CQ; qq += L.QB) {
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (pl < L.PL) {
      f32x4 mean, rstd, w, bb;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = qq * 4 + e;
        mean[e] = stats[((long long)b * C + c) * 2];
        rstd[e] = stats[((long long)b * C + c) * 2 + 1];
        w[e] = adain ? adain[(long long)b * ad_ld + w_off + c] : 1.f;
        bb[e] = adain ? adain[(long long)b * ad_ld + b_off + c] : 0.f;
      }
      for (int p = p0 + pl; p < p1; p += L.PL) {
        const long long o = base + (long long)p * C + qq * 4;
        const f32x4 xv = ld4(x + o);
        const f32x4 xh = (xv - mean) * rstd;
        f32x4 g = ld4(dy + o);
        if (relu) {   // the branch the FORWARD took: the same x * scale + shift, bit for bit (in_finalize / in_apply)
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = norm_act_grad(xv[e] * (rstd[e] * w[e]) + (bb[e] - mean[e] * rstd[e] * w[e]), g[e], relu);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1[e] += (double)g[e];
          s2[e] += (double)g[e] * (double)xh[e];
        }
      }
      double* d = sm + ((pl * L.QB + q) * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { d[e] = s1[e]; d[4 + e] = s2[e]; }
    }
    __syncthreads();
    if (pl == 0) {
      for (int l = 1; l < L.PL; ++l) {
        const double* d = sm + ((l * L.QB + q) * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += d[e]; s2[e] += d[4 + e]; }
      }
      double* o = partial + ((long long)(b * nsplit + sp) * 2) * C + qq * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = s1[e]; o[C + e] = s2[e]; }
    }
    __syncthreads();
  }
}

// Backward fold, once per (sample, channel): d_adain weight = sum(g*xhat), bias = sum(g); coef[b][6][C] = mean, rstd,
// w, b, mean(g), mean(g*xhat) for the apply kernel.
__global__ __launch_bounds__(NT) void in_bwd_finalize_kernel(const double* __restrict__ partial,
                                                             const float* __restrict__ stats, float* __restrict__ coef,
                                                             int B, int HW, int C, int nsplit,
                                                             const float* __restrict__ adain, float* __restrict__ d_adain,
                                                             int ad_ld, int w_off, int b_off) {
  double a1, a2;
  int b, c;
  if (!fold_partials(partial, B, C, nsplit, b, c, a1, a2)) return;
  const int i = b * C + c;
  if (d_adain != nullptr) {
    d_adain[(long long)b * ad_ld + w_off + c] = (float)a2;
    d_adain[(long long)b * ad_ld + b_off + c] = (float)a1;
  }
  const double inv_n = 1.0 / (double)HW;
  float* o = coef + (long long)b * 6 * C + c;
  o[0] = stats[(long long)i * 2];
  o[C] = stats[(long long)i * 2 + 1];
  o[2 * C] = adain ? adain[(long long)b * ad_ld + w_off + c] : 1.f;
  o[3 * C] = adain ? adain[(long long)b * ad_ld + b_off + c] : 0.f;
  o[4 * C] = (float)(a1 * inv_n);
  o[5 * C] = (float)(a2 * inv_n);
}

// dx = rstd*w*(g - mean(g) - xhat*mean(g*xhat))
template <typename T>
__global__ __launch_bounds__(NT) void in_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const float* __restrict__ coef, T* __restrict__ dx,
                                                          int HW, int C, int nsplit, int relu) {
  extern __shared__ float sm[];  // mean[C], rstd[C], w[C], b[C], a1[C], a2[C]
  float* s_mean = sm;
  float* s_rstd = sm + C;
  float* s_w = sm + 2 * C;
  float* s_b = sm + 3 * C;
  float* s_a1 = sm + 4 * C;
  float* s_a2 = sm + 5 * C;
  const int b = blockIdx.y, sp = blockIdx.x;
  for (int c = threadIdx.x; c < 6 * C; c += NT) sm[c] = coef[(long long)b * 6 * C + c];
  __syncthreads();
  const int CQ = C >> 2;
  const int per = (HW + nsplit - 1) / nsplit;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  const long long i0 = (long long)p0 * CQ, i1 = (long long)p1 * CQ;
  const long long base = (long long)b * HW * C;
#pragma unroll 4
  for (long long i = i0 + threadIdx.x; i < i1; i += NT) {
    const int q = (int)(i % CQ);
    const f32x4 mean = *reinterpret_cast<const f32x4*>(s_mean + q * 4);
    const f32x4 rstd = *reinterpret_cast<const f32x4*>(s_rstd + q * 4);
    const f32x4 w = *reinterpret_cast<const f32x4*>(s_w + q * 4);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(s_b + q * 4);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(s_a1 + q * 4);
    const f32x4 a2 = *reinterpret_cast<const f32x4*>(s_a2 + q * 4);
    const f32x4 xv = ld4(x + base + i * 4);
    const f32x4 xh = (xv - mean) * rstd;
    f32x4 g = ld4(dy + base + i * 4);
    if (relu) {   // the forward's own x * scale + shift (see in_bwd_stats_kernel)
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = norm_act_grad(xv[e] * (rstd[e] * w[e]) + (bb[e] - mean[e] * rstd[e] * w[e]), g[e], relu);
    }
    const f32x4 r = rstd * w * (g - a1 - xh * a2);
    st4(dx + base + i * 4, r);
  }
}

// ---------------------------------------------------------------------------------------
// Channel-sliced form (C a multiple of 64): a block = (pixel split, sample, slice of 64 channels).  Statistics and apply use
// the same grid, nsplit x B x C/64 with nsplit * C/64 ~ 64, so an apply block needs the statistics of only ITS 64 channels
// and folds their nsplit partial rows itself (nsplit * 1 KiB from L2, a few per cent of the bytes it streams): the separate
// fold launch between the two -- 7 us of latency, 200 times per step, on the dependent chain conv -> norm -> conv -- is gone.
// A pixel's 64-channel slice is 256 contiguous bytes (two full cache lines), a wave's float4 loads cover four of them.
// ---------------------------------------------------------------------------------------
constexpr int SLICE = 64;

// sums over the splits of partial[b][split][2][C] for channel c0 + (tid & 63): four groups of threads add every fourth
// split in order, then the groups are added in order (the order fold_partials uses); valid in threads < 64
__device__ inline void fold_slice(const double* __restrict__ partial, int b, int C, int nsplit, int c0, double (*red)[SLICE][2],
                                  double& s1, double& s2) {
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, c = c0 + cl;
  s1 = 0.0;
  s2 = 0.0;
  for (int k = g; k < nsplit; k += 4) {
    const double* o = partial + ((long long)(b * nsplit + k) * 2) * C;
    s1 += o[c];
    s2 += o[C + c];
  }
  if (g > 0) {
    red[g - 1][cl][0] = s1;
    red[g - 1][cl][1] = s2;
  }
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      s1 += red[j][cl][0];
      s2 += red[j][cl][1];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void in_apply_sliced_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                             const double* __restrict__ partial, float* __restrict__ stats,
                                                             int HW, int C, int nsplit, const float* __restrict__ adain,
                                                             int ad_ld, int w_off, int b_off, float eps,
                                                             const T* __restrict__ residual, int relu) {
  __shared__ double red[3][SLICE][2];
  __shared__ __attribute__((aligned(16))) float scale[SLICE], shift[SLICE];
  const int b = blockIdx.y, sp = blockIdx.x, c0 = blockIdx.z * SLICE;
  double s1, s2;
  fold_slice(partial, b, C, nsplit, c0, red, s1, s2);
  if (threadIdx.x < SLICE) {   // the arithmetic of in_finalize_kernel
    const int c = c0 + threadIdx.x;
    const double inv_n = 1.0 / (double)HW;
    const double d = s1 * inv_n;
    const float mean = (float)((double)ld1(x + (long long)b * HW * C + c) + d);   // pivot = first pixel of the plane
    double var = s2 * inv_n - d * d;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    float w = 1.f, bb = 0.f;
    if (adain != nullptr) {
      w = adain[(long long)b * ad_ld + w_off + c];
      bb = adain[(long long)b * ad_ld + b_off + c];
    }
    if (sp == 0) {
      stats[((long long)b * C + c) * 2] = mean;
      stats[((long long)b * C + c) * 2 + 1] = rstd;
    }
    scale[threadIdx.x] = rstd * w;
    shift[threadIdx.x] = bb - mean * rstd * w;
  }
  __syncthreads();
  const int per = (HW + nsplit - 1) / nsplit;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  const long long base = (long long)b * HW * C + c0;
  const int q = threadIdx.x & 15;   // quad of the slice: fixed per thread (NT / 16 pixels per sweep)
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + q * 4);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + q * 4);
#pragma unroll 4
  for (int p = p0 + (threadIdx.x >> 4); p < p1; p += NT / 16) {
    const long long o = base + (long long)p * C + q * 4;
    f32x4 v = ld4(x + o);
    v = v * sc + sh;
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = norm_act(v[e], relu);
    }
    if (residual != nullptr) v += ld4(residual + o);
    st4(y + o, v);
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void in_bwd_apply_sliced_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                 const double* __restrict__ partial,
                                                                 const float* __restrict__ stats, T* __restrict__ dx, int HW,
                                                                 int C, int nsplit, const float* __restrict__ adain,
                                                                 float* __restrict__ d_adain, int ad_ld, int w_off, int b_off,
                                                                 int relu) {
  __shared__ double red[3][SLICE][2];
  __shared__ __attribute__((aligned(16))) float s_co[6][SLICE];   // mean, rstd, w, b, mean(g), mean(g*xhat)
  const int b = blockIdx.y, sp = blockIdx.x, c0 = blockIdx.z * SLICE;
  double a1, a2;
  fold_slice(partial, b, C, nsplit, c0, red, a1, a2);
  if (threadIdx.x < SLICE) {   // the arithmetic of in_bwd_finalize_kernel
    const int c = c0 + threadIdx.x;
    if (d_adain != nullptr && sp == 0) {
      d_adain[(long long)b * ad_ld + w_off + c] = (float)a2;
      d_adain[(long long)b * ad_ld + b_off + c] = (float)a1;
    }
    const double inv_n = 1.0 / (double)HW;
    s_co[0][threadIdx.x] = stats[((long long)b * C + c) * 2];
    s_co[1][threadIdx.x] = stats[((long long)b * C + c) * 2 + 1];
    s_co[2][threadIdx.x] = adain ? adain[(long long)b * ad_ld + w_off + c] : 1.f;
    s_co[3][threadIdx.x] = adain ? adain[(long long)b * ad_ld + b_off + c] : 0.f;
    s_co[4][threadIdx.x] = (float)(a1 * inv_n);
    s_co[5][threadIdx.x] = (float)(a2 * inv_n);
  }
  __syncthreads();
  const int per = (HW + nsplit - 1) / nsplit;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  const long long base = (long long)b * HW * C + c0;
  const int q = threadIdx.x & 15;
  const f32x4 mean = *reinterpret_cast<const f32x4*>(&s_co[0][q * 4]);
  const f32x4 rstd = *reinterpret_cast<const f32x4*>(&s_co[1][q * 4]);
  const f32x4 w = *reinterpret_cast<const f32x4*>(&s_co[2][q * 4]);
  const f32x4 bb = *reinterpret_cast<const f32x4*>(&s_co[3][q * 4]);
  const f32x4 a1v = *reinterpret_cast<const f32x4*>(&s_co[4][q * 4]);
  const f32x4 a2v = *reinterpret_cast<const f32x4*>(&s_co[5][q * 4]);
#pragma unroll 4
  for (int p = p0 + (threadIdx.x >> 4); p < p1; p += NT / 16) {
    const long long o = base + (long long)p * C + q * 4;
    const f32x4 xv = ld4(x + o);
    const f32x4 xh = (xv - mean) * rstd;
    f32x4 g = ld4(dy + o);
    if (relu) {   // the forward's own x * scale + shift (see in_bwd_stats_kernel)
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = norm_act_grad(xv[e] * (rstd[e] * w[e]) + (bb[e] - mean[e] * rstd[e] * w[e]), g[e], relu);
    }
    const f32x4 r = rstd * w * (g - a1v - xh * a2v);
    st4(dx + o, r);
  }
}

// ---------------------------------------------------------------------------------------
// MUNIT LayerNorm
// ---------------------------------------------------------------------------------------
__device__ inline double block_sum(double v, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}

// partial[b][split][2]: sum(x-pivot), sum((x-pivot)^2), pivot = x[b][0]
template <typename T>
__global__ __launch_bounds__(NT) void ln_stats_kernel(const T* __restrict__ x, double* __restrict__ partial,
                                                      long long n_per_sample, int nsplit) {
  __shared__ double red[NT / 64];
  const int b = blockIdx.y, sp = blockIdx.x;
  const T* xb = x + (long long)b * n_per_sample;
  const long long nq = n_per_sample >> 2;
  const long long per = (nq + nsplit - 1) / nsplit;
  const long long q0 = sp * per, q1 = min(nq, q0 + per);
  const float piv = ld1(xb);
  double s1 = 0.0, s2 = 0.0;
  for (long long i = q0 + threadIdx.x; i < q1; i += NT) {
    f32x4 v = ld4(xb + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double d = (double)(v[e] - piv);
      s1 += d;
      s2 += d * d;
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    partial[((long long)b * nsplit + sp) * 2] = s1;
    partial[((long long)b * nsplit + sp) * 2 + 1] = s2;
  }
}

template <typename T>
__device__ inline void ln_finish(const double* partial, const T* xb, int b, int nsplit, long long n,
                                 float* mean, float* sigma) {
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < nsplit; ++k) {
    s1 += partial[((long long)b * nsplit + k) * 2];
    s2 += partial[((long long)b * nsplit + k) * 2 + 1];
  }
  const double d = s1 / (double)n;
  double var = (s2 - s1 * d) / (double)(n - 1);  // unbiased (torch.std default), networks.py:868/871
  if (var < 0.0) var = 0.0;
  *mean = (float)((double)ld1(xb) + d);
  *sigma = (float)sqrt(var);
}

template <typename T>
__global__ __launch_bounds__(NT) void ln_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                      const double* __restrict__ partial, float* __restrict__ stats,
                                                      int HW, int C, int nsplit, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int relu, float eps) {
  extern __shared__ float sm[];  // scale[C], shift[C]
  float* scale = sm;
  float* shift = sm + C;
  const int b = blockIdx.y, sp = blockIdx.x;
  const long long n = (long long)HW * C;
  const T* xb = x + (long long)b * n;
  float mean, sigma;
  ln_finish(partial, xb, b, nsplit, n, &mean, &sigma);
  const float inv = 1.f / (sigma + eps);
  if (sp == 0 && threadIdx.x == 0) {
    stats[b * 2] = mean;
    stats[b * 2 + 1] = sigma;
  }
  for (int c = threadIdx.x; c < C; c += NT) {
    scale[c] = inv * gamma[c];
    shift[c] = beta[c] - mean * inv * gamma[c];
  }
  __syncthreads();
  const int CQ = C >> 2;
  const long long nq = n >> 2;
  const long long per = (nq + nsplit - 1) / nsplit;
  const long long q0 = sp * per, q1 = min(nq, q0 + per);
  for (long long i = q0 + threadIdx.x; i < q1; i += NT) {
    const int q = (int)(i % CQ);
    f32x4 v = ld4(xb + i * 4);
    v = v * *reinterpret_cast<const f32x4*>(scale + q * 4) + *reinterpret_cast<const f32x4*>(shift + q * 4);
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = norm_act(v[e], relu);
    }
    st4(y + (long long)b * n + i * 4, v);
  }
}

// backward partials. per (b, split): chan[2][C] = sum(g*xn), sum(g) per channel (g = dy*relu');
// samp[2] = sum(g*gamma), sum(g*gamma*xn).
// cpart layout [b][split][2][C], spart layout [b][split][2]
template <typename T>
__global__ __launch_bounds__(NT) void ln_bwd_stats_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const float* __restrict__ stats, double* __restrict__ cpart,
                                                          double* __restrict__ spart, int HW, int C, int nsplit,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int relu, float eps) {
  extern __shared__ double smd[];  // [PL][QB*4][2]
  double* sm = smd;
  __shared__ double red[NT / 64];
  const Lay L = make_lay(C);
  const int b = blockIdx.y, sp = blockIdx.x;
  const int q = threadIdx.x % L.QB, pl = threadIdx.x / L.QB;
  const int per = (HW + nsplit - 1) / nsplit;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  const long long base = (long long)b * HW * C;
  const float mean = stats[b * 2];
  const float inv = 1.f / (stats[b * 2 + 1] + eps);
  double t1 = 0.0, t2 = 0.0;
  for (int qq = q; qq < L.CQ; qq += L.QB) {
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (pl < L.PL) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + qq * 4);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + qq * 4);
      for (int p = p0 + pl; p < p1; p += L.PL) {
        const long long o = base + (long long)p * C + qq * 4;
        const f32x4 xv = ld4(x + o);
        const f32x4 xn = (xv - mean) * inv;
        f32x4 g = ld4(dy + o);
        if (relu) {   // the branch the forward took: x * (inv * gamma) + (beta - mean * inv * gamma), bit for bit (ln_apply)
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = norm_act_grad(xv[e] * (inv * gm[e]) + (bt[e] - mean * inv * gm[e]), g[e], relu);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1[e] += (double)g[e] * (double)xn[e];
          s2[e] += (double)g[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t1 += s2[e] * (double)gm[e];
        t2 += s1[e] * (double)gm[e];
      }
      double* d = sm + ((pl * L.QB + q) * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { d[e] = s1[e]; d[4 + e] = s2[e]; }
    }
    __syncthreads();
    if (pl == 0) {
      for (int l = 1; l < L.PL; ++l) {
        const double* d = sm + ((l * L.QB + q) * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += d[e]; s2[e] += d[4 + e]; }
      }
      double* o = cpart + ((long long)(b * nsplit + sp) * 2) * C + qq * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = s1[e]; o[C + e] = s2[e]; }
    }
    __syncthreads();
  }
  t1 = block_sum(t1, red);
  t2 = block_sum(t2, red);
  if (threadIdx.x == 0) {
    spart[((long long)b * nsplit + sp) * 2] = t1;
    spart[((long long)b * nsplit + sp) * 2 + 1] = t2;
  }
}

// dx = inv*(h - S1/N) - S2*xn/((N-1)*sigma), h = g*gamma
template <typename T>
__global__ __launch_bounds__(NT) void ln_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const float* __restrict__ stats,
                                                          const double* __restrict__ spart, T* __restrict__ dx,
                                                          int HW, int C, int nsplit, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int relu, float eps) {
  const int b = blockIdx.y, sp = blockIdx.x;
  const long long n = (long long)HW * C;
  const float mean = stats[b * 2];
  const float sigma = stats[b * 2 + 1];
  const float inv = 1.f / (sigma + eps);
  double S1 = 0.0, S2 = 0.0;
  for (int k = 0; k < nsplit; ++k) {
    S1 += spart[((long long)b * nsplit + k) * 2];
    S2 += spart[((long long)b * nsplit + k) * 2 + 1];
  }
  const float c1 = (float)(S1 / (double)n);
  const float c2 = sigma > 0.f ? (float)(S2 / ((double)(n - 1) * (double)sigma)) : 0.f;
  const int CQ = C >> 2;
  const long long nq = n >> 2;
  const long long per = (nq + nsplit - 1) / nsplit;
  const long long q0 = sp * per, q1 = min(nq, q0 + per);
  const long long base = (long long)b * n;
  for (long long i = q0 + threadIdx.x; i < q1; i += NT) {
    const int q = (int)(i % CQ);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + q * 4);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + q * 4);
    const f32x4 xv = ld4(x + base + i * 4);
    const f32x4 xn = (xv - mean) * inv;
    f32x4 g = ld4(dy + base + i * 4);
    if (relu) {   // the forward's own expression (see ln_bwd_stats_kernel)
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = norm_act_grad(xv[e] * (inv * gm[e]) + (bt[e] - mean * inv * gm[e]), g[e], relu);
    }
    const f32x4 r = (g * gm - c1) * inv - xn * c2;
    st4(dx + base + i * 4, r);
  }
}

// dgamma[c] = acc*dgamma[c] + sum_{b,split} cpart[..][0][c]; dbeta likewise with [1].  One block per
// channel, threads stride over the (sample, split) partial rows.
__global__ __launch_bounds__(NT) void ln_bwd_param_kernel(const double* __restrict__ cpart, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, int C, int nblk, float acc) {
  __shared__ double red[NT / 64];
  const int c = blockIdx.x;
  double g = 0.0, bt = 0.0;
  for (int k = threadIdx.x; k < nblk; k += NT) {
    g += cpart[((long long)k * 2) * C + c];
    bt += cpart[((long long)k * 2 + 1) * C + c];
  }
  g = block_sum(g, red);
  bt = block_sum(bt, red);
  if (threadIdx.x == 0) {
    dgamma[c] = (acc != 0.f ? acc * dgamma[c] : 0.f) + (float)g;
    dbeta[c] = (acc != 0.f ? acc * dbeta[c] : 0.f) + (float)bt;
  }
}

}  // namespace

namespace {
size_t in_partial_bytes(int B, int C) { return align_up((size_t)B * MAX_SPLIT * 2 * C * sizeof(double), 256); }
bool in_sliced(int C) { return C % SLICE == 0 && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_SLICED_NORM"); }
// pixel splits of the sliced grid: pick_split's block count shared out over the C / 64 slices
int sliced_split(int B, int HW, int C) { return std::max(1, pick_split(B, HW) / (C / SLICE)); }
}
// [split partials (fp64)][per-(sample, channel) coefficients of the apply kernels: 6 floats]
extern "C" size_t munit_instnorm_workspace_bytes(int B, int HW, int C) {
  return in_partial_bytes(B, C) + align_up((size_t)B * 6 * C * sizeof(float), 256);
}

namespace {
template <typename T>
int instnorm_fwd_t(const T* x, T* y, float* stats, int B, int HW, int C, const float* adain, int ad_ld, int w_off,
                   int b_off, const T* residual, int relu, float eps, void* ws, size_t ws_bytes, munit_stream_t stream) {
  MUNIT_CHECK_ARG(x && y && stats && ws, "instnorm_fwd: null pointer");
  MUNIT_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % 4 == 0, "instnorm_fwd: bad dims B=%d HW=%d C=%d", B, HW, C);
  MUNIT_CHECK_ARG(C <= 4096, "instnorm_fwd: C=%d too large", C);
  if (ws_bytes < munit_instnorm_workspace_bytes(B, HW, C)) {
    munit_set_error("instnorm_fwd: workspace too small");
    return MUNIT_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  double* partial = reinterpret_cast<double*>(ws);
  if (in_sliced(C)) {
    const int ns = sliced_split(B, HW, C);
    const Lay Ls = make_lay(SLICE);
    hipLaunchKernelGGL(in_stats_kernel<T>, dim3(ns, B, C / SLICE), dim3(NT), (size_t)Ls.PL * Ls.QB * 8 * sizeof(double), st, x,
                       partial, HW, C, ns, SLICE);
    MUNIT_CHECK_LAUNCH("in_stats");
    hipLaunchKernelGGL(in_apply_sliced_kernel<T>, dim3(ns, B, C / SLICE), dim3(NT), 0, st, x, y, partial, stats, HW, C, ns, adain,
                       ad_ld, w_off, b_off, eps, residual, relu);
    MUNIT_CHECK_LAUNCH("in_apply_sliced");
    return MUNIT_OK;
  }
  const int ns = pick_split(B, HW);
  const Lay L = make_lay(C);
  hipLaunchKernelGGL(in_stats_kernel<T>, dim3(ns, B), dim3(NT), (size_t)L.PL * L.QB * 8 * sizeof(double), st, x,
                     partial, HW, C, ns, C);
  MUNIT_CHECK_LAUNCH("in_stats");
  float* coef = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + in_partial_bytes(B, C));
  hipLaunchKernelGGL(in_finalize_kernel<T>, dim3(cdiv(C, 64), B), dim3(NT), 0, st, x, partial, stats, coef, B,
                     HW, C, ns, adain, ad_ld, w_off, b_off, eps);
  MUNIT_CHECK_LAUNCH("in_finalize");
  hipLaunchKernelGGL(in_apply_kernel<T>, dim3(ns, B), dim3(NT), (size_t)2 * C * sizeof(float), st, x, y, coef, HW, C,
                     ns, residual, relu);
  MUNIT_CHECK_LAUNCH("in_apply");
  return MUNIT_OK;
}

template <typename T>
int instnorm_bwd_t(const T* x, const T* dy, const float* stats, T* dx, int B, int HW, int C, const float* adain,
                   float* d_adain, int ad_ld, int w_off, int b_off, int relu, void* ws, size_t ws_bytes,
                   munit_stream_t stream) {
  MUNIT_CHECK_ARG(x && dy && stats && dx && ws, "instnorm_bwd: null pointer");
  MUNIT_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 4096, "instnorm_bwd: bad dims");
  if (ws_bytes < munit_instnorm_workspace_bytes(B, HW, C)) {
    munit_set_error("instnorm_bwd: workspace too small");
    return MUNIT_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  double* partial = reinterpret_cast<double*>(ws);
  if (in_sliced(C)) {
    const int ns = sliced_split(B, HW, C);
    const Lay Ls = make_lay(SLICE);
    hipLaunchKernelGGL(in_bwd_stats_kernel<T>, dim3(ns, B, C / SLICE), dim3(NT), (size_t)Ls.PL * Ls.QB * 8 * sizeof(double), st, x,
                       dy, stats, partial, HW, C, ns, adain, ad_ld, w_off, b_off, relu, SLICE);
    MUNIT_CHECK_LAUNCH("in_bwd_stats");
    hipLaunchKernelGGL(in_bwd_apply_sliced_kernel<T>, dim3(ns, B, C / SLICE), dim3(NT), 0, st, x, dy, partial, stats, dx, HW, C, ns,
                       adain, d_adain, ad_ld, w_off, b_off, relu);
    MUNIT_CHECK_LAUNCH("in_bwd_apply_sliced");
    return MUNIT_OK;
  }
  const int ns = pick_split(B, HW);
  const Lay L = make_lay(C);
  hipLaunchKernelGGL(in_bwd_stats_kernel<T>, dim3(ns, B), dim3(NT), (size_t)L.PL * L.QB * 8 * sizeof(double), st, x,
                     dy, stats, partial, HW, C, ns, adain, ad_ld, w_off, b_off, relu, C);
  MUNIT_CHECK_LAUNCH("in_bwd_stats");
  float* coef = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + in_partial_bytes(B, C));
  hipLaunchKernelGGL(in_bwd_finalize_kernel, dim3(cdiv(C, 64), B), dim3(NT), 0, st, partial, stats, coef, B, HW,
                     C, ns, adain, d_adain, ad_ld, w_off, b_off);
  MUNIT_CHECK_LAUNCH("in_bwd_finalize");
  hipLaunchKernelGGL(in_bwd_apply_kernel<T>, dim3(ns, B), dim3(NT), (size_t)6 * C * sizeof(float), st, x, dy, coef, dx,
                     HW, C, ns, relu);
  MUNIT_CHECK_LAUNCH("in_bwd_apply");
  return MUNIT_OK;
}
}  // namespace

extern "C" int munit_instnorm_fwd(const float* x, float* y, float* stats, int B, int HW, int C,
                                  const float* adain, int ad_ld, int w_off, int b_off, const float* residual,
                                  int relu, float eps, void* ws, size_t ws_bytes, munit_stream_t stream) {
  return instnorm_fwd_t<float>(x, y, stats, B, HW, C, adain, ad_ld, w_off, b_off, residual, relu, eps, ws, ws_bytes, stream);
}
extern "C" int munit_instnorm_fwd_bf16(const void* x, void* y, float* stats, int B, int HW, int C,
                                       const float* adain, int ad_ld, int w_off, int b_off, const void* residual,
                                       int relu, float eps, void* ws, size_t ws_bytes, munit_stream_t stream) {
  return instnorm_fwd_t<bf16_t>(reinterpret_cast<const bf16_t*>(x), reinterpret_cast<bf16_t*>(y), stats, B, HW, C, adain, ad_ld,
                                w_off, b_off, reinterpret_cast<const bf16_t*>(residual), relu, eps, ws, ws_bytes, stream);
}
extern "C" int munit_instnorm_bwd(const float* x, const float* dy, const float* stats, float* dx, int B, int HW,
                                  int C, const float* adain, float* d_adain, int ad_ld, int w_off, int b_off,
                                  int relu, void* ws, size_t ws_bytes, munit_stream_t stream) {
  return instnorm_bwd_t<float>(x, dy, stats, dx, B, HW, C, adain, d_adain, ad_ld, w_off, b_off, relu, ws, ws_bytes, stream);
}
extern "C" int munit_instnorm_bwd_bf16(const void* x, const void* dy, const float* stats, void* dx, int B, int HW,
                                       int C, const float* adain, float* d_adain, int ad_ld, int w_off, int b_off,
                                       int relu, void* ws, size_t ws_bytes, munit_stream_t stream) {
  return instnorm_bwd_t<bf16_t>(reinterpret_cast<const bf16_t*>(x), reinterpret_cast<const bf16_t*>(dy), stats,
                                reinterpret_cast<bf16_t*>(dx), B, HW, C, adain, d_adain, ad_ld, w_off, b_off, relu, ws, ws_bytes,
                                stream);
}

extern "C" size_t munit_layernorm_workspace_bytes(int B, int HW, int C) {
  return align_up((size_t)B * MAX_SPLIT * 2 * C * sizeof(double), 256) +
         align_up((size_t)B * MAX_SPLIT * 2 * sizeof(double), 256);
}

namespace {
template <typename T>
int layernorm_fwd_t(const T* x, T* y, float* stats, int B, int HW, int C, const float* gamma, const float* beta, int relu,
                    float eps, void* ws, size_t ws_bytes, munit_stream_t stream) {
  MUNIT_CHECK_ARG(x && y && stats && gamma && beta && ws, "layernorm_fwd: null pointer");
  MUNIT_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 4096, "layernorm_fwd: bad dims");
  MUNIT_CHECK_ARG((long long)HW * C > 1, "layernorm_fwd: unbiased std needs more than one element");
  if (ws_bytes < munit_layernorm_workspace_bytes(B, HW, C)) {
    munit_set_error("layernorm_fwd: workspace too small");
    return MUNIT_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int ns = pick_split(B, HW);
  double* spart = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(ln_stats_kernel<T>, dim3(ns, B), dim3(NT), 0, st, x, spart, (long long)HW * C, ns);
  MUNIT_CHECK_LAUNCH("ln_stats");
  hipLaunchKernelGGL(ln_apply_kernel<T>, dim3(ns, B), dim3(NT), (size_t)2 * C * sizeof(float), st, x, y, spart, stats,
                     HW, C, ns, gamma, beta, relu, eps);
  MUNIT_CHECK_LAUNCH("ln_apply");
  return MUNIT_OK;
}

template <typename T>
int layernorm_bwd_t(const T* x, const T* dy, const float* stats, T* dx, int B, int HW, int C, const float* gamma,
                    const float* beta, float* dgamma, float* dbeta, float acc, int relu, float eps, void* ws,
                    size_t ws_bytes, munit_stream_t stream) {
  MUNIT_CHECK_ARG(x && dy && stats && dx && gamma && beta && dgamma && dbeta && ws, "layernorm_bwd: null pointer");
  MUNIT_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 4096, "layernorm_bwd: bad dims");
  if (ws_bytes < munit_layernorm_workspace_bytes(B, HW, C)) {
    munit_set_error("layernorm_bwd: workspace too small");
    return MUNIT_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int ns = pick_split(B, HW);
  double* cpart = reinterpret_cast<double*>(ws);
  double* spart = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) +
                                            align_up((size_t)B * MAX_SPLIT * 2 * C * sizeof(double), 256));
  const Lay L = make_lay(C);
  hipLaunchKernelGGL(ln_bwd_stats_kernel<T>, dim3(ns, B), dim3(NT), (size_t)L.PL * L.QB * 8 * sizeof(double), st, x,
                     dy, stats, cpart, spart, HW, C, ns, gamma, beta, relu, eps);
  MUNIT_CHECK_LAUNCH("ln_bwd_stats");
  hipLaunchKernelGGL(ln_bwd_apply_kernel<T>, dim3(ns, B), dim3(NT), 0, st, x, dy, stats, spart, dx, HW, C, ns, gamma,
                     beta, relu, eps);
  MUNIT_CHECK_LAUNCH("ln_bwd_apply");
  hipLaunchKernelGGL(ln_bwd_param_kernel, dim3(C), dim3(NT), 0, st, cpart, dgamma, dbeta, C, B * ns,
                     acc);
  MUNIT_CHECK_LAUNCH("ln_bwd_param");
  return MUNIT_OK;
}
}  // namespace

extern "C" int munit_layernorm_fwd(const float* x, float* y, float* stats, int B, int HW, int C,
                                   const float* gamma, const float* beta, int relu, float eps, void* ws,
                                   size_t ws_bytes, munit_stream_t stream) {
  return layernorm_fwd_t<float>(x, y, stats, B, HW, C, gamma, beta, relu, eps, ws, ws_bytes, stream);
}
extern "C" int munit_layernorm_fwd_bf16(const void* x, void* y, float* stats, int B, int HW, int C,
                                        const float* gamma, const float* beta, int relu, float eps, void* ws,
                                        size_t ws_bytes, munit_stream_t stream) {
  return layernorm_fwd_t<bf16_t>(reinterpret_cast<const bf16_t*>(x), reinterpret_cast<bf16_t*>(y), stats, B, HW, C, gamma, beta,
                                 relu, eps, ws, ws_bytes, stream);
}
extern "C" int munit_layernorm_bwd(const float* x, const float* dy, const float* stats, float* dx, int B, int HW,
                                   int C, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                                   float acc, int relu, float eps, void* ws, size_t ws_bytes,
                                   munit_stream_t stream) {
  return layernorm_bwd_t<float>(x, dy, stats, dx, B, HW, C, gamma, beta, dgamma, dbeta, acc, relu, eps, ws, ws_bytes, stream);
}
extern "C" int munit_layernorm_bwd_bf16(const void* x, const void* dy, const float* stats, void* dx, int B, int HW,
                                        int C, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                                        float acc, int relu, float eps, void* ws, size_t ws_bytes,
                                        munit_stream_t stream) {
  return layernorm_bwd_t<bf16_t>(reinterpret_cast<const bf16_t*>(x), reinterpret_cast<const bf16_t*>(dy), stats,
                                 reinterpret_cast<bf16_t*>(dx), B, HW, C, gamma, beta, dgamma, dbeta, acc, relu, eps, ws,
                                 ws_bytes, stream);
}
