// Backward-weight of the convolutions (autograd of networks.py:691-696) as a split-K
// MFMA GEMM:  dw[co][k] = sum_m dy[m][co] * A[m][k],  k = (kh*KW+kw)*Cin+ci,
// A = the same on-the-fly im2col gather the forward uses (reflect/zero pad, stride,
// fused nearest upsample).  The reduction runs over m = output pixels (up to 524288*B),
// so the grid is (k tiles) x (cout tiles) x (pixel splits); each split writes a partial
// slab and slab_reduce_kernel sums them in a fixed order (deterministic, no atomics).
// The bias gradient (column sums of dy) rides along in the k-tile-0 blocks.
//
// Tile: BC (cout) x 128 (k) accumulators, 32 pixels per step.  dy and A tiles are staged
// in LDS as [pixel][channel]; the MFMA operand fetch is one ds_read_b32 per lane
// (lanes 0-31 = consecutive channels => conflict-free).
#include "common.h"
#include <cstdlib>

namespace {

struct WgradParams {
  const float* x;
  const float* dy;
  float* slab;       // [nsplit][Cout][Ktot]
  float* bias_slab;  // [nsplit][Cout] or null
  int B, H, W, Cin, Hu, Wu, ups;
  int Ho, Wo, Cout;
  int KH, KW, stride, pad, reflect;
  int Ktot, M;
  int pix_per_split;  // multiple of 32
  // dy addressing: offset(b, oh, ow) = b*dy_sb + oh*dy_sh + ow*dy_sw + dy_off (floats); the plain layer
  // has dy_sw = Cout etc.; the sub-pixel phases of an up-sampling conv read every other row/column
  long long dy_sb, dy_sh;
  int dy_sw;
  long long dy_off;
  int frame;  // GEMM rows enumerate only the 2-pixel border frame of the Ho x Wo output (see decode_pixel)
};

// row index -> (sample, output row, output column); same enumeration as conv_igemm.hip
__device__ inline void decode_pixel(int m, int Ho, int Wo, int frame, int& b, int& oh, int& ow) {
  if (!frame) {
    const int HoWo = Ho * Wo;
    b = m / HoWo;
    const int rem = m - b * HoWo;
    oh = rem / Wo;
    ow = rem - oh * Wo;
  } else {
    const int nb = 4 * Wo + 4 * (Ho - 4);
    b = m / nb;
    int r = m - b * nb;
    if (r < 4 * Wo) {
      const int q = r / Wo;
      ow = r - q * Wo;
      oh = q < 2 ? q : Ho - 4 + q;
    } else {
      r -= 4 * Wo;
      const int q = r & 3;
      oh = 2 + (r >> 2);
      ow = q < 2 ? q : Wo - 4 + q;
    }
  }
}

// advance a (b, oh, ow) pixel of the linear Ho x Wo enumeration by WPX pixels
template <int WPX>
__device__ inline void advance_pixel(int& b, int& oh, int& ow, int Ho, int Wo) {
  if (Wo >= WPX) {  // wave-uniform: at most one row wrap per step, done with selects
    int w = ow + WPX;
    const bool wrap = w >= Wo;
    w -= wrap ? Wo : 0;
    const int h = oh + (wrap ? 1 : 0);
    const bool wrap2 = h >= Ho;
    ow = w;
    oh = wrap2 ? 0 : h;
    b += wrap2 ? 1 : 0;
  } else {
    ow += WPX;
    while (ow >= Wo) {
      ow -= Wo;
      if (++oh == Ho) { oh = 0; ++b; }
    }
  }
}


constexpr int WP = 32;   // pixels per step

constexpr int WTHR = 512;

template <int BC, bool ALIGNED>
__global__ __launch_bounds__(WTHR, 4) void conv_wgrad_kernel(WgradParams p) {
  // 8 waves: 2 along cout x 4 along k.  Tile BC x WKT with WKT = 128 (BC=128) or 256 (BC=64), so every
  // wave owns two 32x32 accumulators either way (2 cout tiles x 1 k tile, or 1 x 2).
  constexpr int WKT = BC == 128 ? 128 : 256;
  constexpr int NTK = WKT / 128;        // 32-col MFMA tiles per wave along k
  constexpr int XQ = WKT / 4;           // float4 per gathered row
  constexpr int XRPT = WTHR / XQ;       // gather rows covered per pass of the block: 16 or 8
  constexpr int XROWS = WP / XRPT;      // gather rows per loader thread: 2 or 4
  constexpr int WC = BC / 2;            // cout rows per wave
  constexpr int MT = WC / 32;           // 32-row MFMA tiles per wave along cout: 2 (BC=128) or 1
  constexpr int DQ = BC / 4;            // float4 per dy row
  constexpr int DRPT = WTHR / DQ;       // dy rows covered per pass of the block: 16 or 32
  constexpr int DROWS = WP / DRPT;      // dy rows per loader thread: 2 or 1
  __shared__ __attribute__((aligned(16))) float Ds[2 * WP * BC];   // double-buffered: one barrier per step
  __shared__ __attribute__((aligned(16))) float Xs[2 * WP * WKT];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int kc0 = blockIdx.x * WKT;
  const int co0 = blockIdx.y * BC;
  const int split = blockIdx.z;
  const int m_begin = split * p.pix_per_split;
  const int m_end = min(p.M, m_begin + p.pix_per_split);

  // X-tile loader: 32 pixels x WKT k columns; thread -> (row = tid / XQ + XRPT*i, q = tid % XQ)
  const int xq = tid % XQ;
  const int xr0 = tid / XQ;
  int ekh[4], ekw[4], eci[4];
  bool eok[4];
  if constexpr (ALIGNED) {
    int k = kc0 + xq * 4;
    eok[0] = k < p.Ktot;
    int kk = eok[0] ? k : 0;
    int tap = kk / p.Cin;
    eci[0] = kk - tap * p.Cin;
    ekh[0] = tap / p.KW;
    ekw[0] = tap - ekh[0] * p.KW;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int k = kc0 + xq * 4 + e;
      eok[e] = k < p.Ktot;
      int kk = eok[e] ? k : 0;
      int tap = kk / p.Cin;
      eci[e] = kk - tap * p.Cin;
      ekh[e] = tap / p.KW;
      ekw[e] = tap - ekh[e] * p.KW;
    }
  }
  // dy-tile loader: thread -> (row = tid / DQ + DRPT * i, q = tid % DQ)
  const int dq = tid % DQ;
  const int dr0 = tid / DQ;

  f32x4 rx[XROWS], rd[DROWS];

  // pixel coordinates of this thread's gather rows and dy rows, advanced incrementally by 32 pixels per
  // step (one division per row up front instead of per step); frame mode decodes each step
  int px_oh[XROWS], px_ow[XROWS], px_b[XROWS];
#pragma unroll
  for (int i = 0; i < XROWS; ++i) {
    int m = m_begin + xr0 + XRPT * i;
    decode_pixel(m < p.M ? m : 0, p.Ho, p.Wo, p.frame, px_b[i], px_oh[i], px_ow[i]);
  }
  int dp_oh[DROWS], dp_ow[DROWS], dp_b[DROWS];
#pragma unroll
  for (int i = 0; i < DROWS; ++i) {
    int m = m_begin + dr0 + DRPT * i;
    decode_pixel(m < p.M ? m : 0, p.Ho, p.Wo, p.frame, dp_b[i], dp_oh[i], dp_ow[i]);
  }

  auto load_tiles = [&](int mbase) {
#pragma unroll
    for (int i = 0; i < XROWS; ++i) {
      const int m = mbase + xr0 + XRPT * i;
      const bool mok = m < m_end;
      if (p.frame) decode_pixel(mok ? m : 0, p.Ho, p.Wo, 1, px_b[i], px_oh[i], px_ow[i]);
      const int oh = px_oh[i], ow = px_ow[i];
      const long long base = (long long)px_b[i] * p.H * p.W;
      if (!p.frame) advance_pixel<WP>(px_b[i], px_oh[i], px_ow[i], p.Ho, p.Wo);
      int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if constexpr (ALIGNED) {
        int ih = src_coord(ih0 + ekh[0], p.Hu, p.ups, p.reflect);
        int iw = src_coord(iw0 + ekw[0], p.Wu, p.ups, p.reflect);
        if (mok && eok[0] && ih >= 0 && iw >= 0)
          v = *reinterpret_cast<const f32x4*>(p.x + (base + (long long)ih * p.W + iw) * p.Cin + eci[0]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int ih = src_coord(ih0 + ekh[e], p.Hu, p.ups, p.reflect);
          int iw = src_coord(iw0 + ekw[e], p.Wu, p.ups, p.reflect);
          float s = 0.f;
          if (mok && eok[e] && ih >= 0 && iw >= 0) s = p.x[(base + (long long)ih * p.W + iw) * p.Cin + eci[e]];
          v[e] = s;
        }
      }
      rx[i] = v;
    }
#pragma unroll
    for (int i = 0; i < DROWS; ++i) {
      const int m = mbase + dr0 + DRPT * i;
      const bool mok = m < m_end;
      if (p.frame) decode_pixel(mok ? m : 0, p.Ho, p.Wo, 1, dp_b[i], dp_oh[i], dp_ow[i]);
      const float* ptr = p.dy + (long long)dp_b[i] * p.dy_sb + (long long)dp_oh[i] * p.dy_sh +
                         (long long)dp_ow[i] * p.dy_sw + p.dy_off + co0 + dq * 4;
      if (!p.frame) advance_pixel<WP>(dp_b[i], dp_oh[i], dp_ow[i], p.Ho, p.Wo);
      const int co = co0 + dq * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (mok) {
        if ((p.Cout & 3) == 0) {
          if (co < p.Cout) v = *reinterpret_cast<const f32x4*>(ptr);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < p.Cout) v[e] = ptr[e];
        }
      }
      rd[i] = v;
    }
  };
  auto store_tiles = [&](int buf) {
    float* Xd = Xs + buf * (WP * WKT);
    float* Dd = Ds + buf * (WP * BC);
#pragma unroll
    for (int i = 0; i < XROWS; ++i) *reinterpret_cast<f32x4*>(&Xd[(xr0 + XRPT * i) * WKT + xq * 4]) = rx[i];
#pragma unroll
    for (int i = 0; i < DROWS; ++i) *reinterpret_cast<f32x4*>(&Dd[(dr0 + DRPT * i) * BC + dq * 4]) = rd[i];
  };

  f32x16 acc[MT][NTK];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int t = 0; t < NTK; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;
  float bsum = 0.f;  // bias column sum: thread tid < BC owns channel co0 + tid
  const bool do_bias = p.bias_slab != nullptr && blockIdx.x == 0;

  const int fr = lane & 31;
  const int fk = lane >> 5;
  auto compute_half = [&](int buf, int k0) {
    const float* Dc = Ds + buf * (WP * BC);
    const float* Xc = Xs + buf * (WP * WKT);
#pragma unroll
    for (int kk = k0; kk < k0 + WP / 2; kk += 2) {
      float a[MT], b[NTK];
#pragma unroll
      for (int t = 0; t < MT; ++t) a[t] = Dc[(kk + fk) * BC + wm * WC + t * 32 + fr];
#pragma unroll
      for (int t = 0; t < NTK; ++t) b[t] = Xc[(kk + fk) * WKT + wn * (32 * NTK) + t * 32 + fr];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NTK; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[u], acc[t][u], 0, 0, 0);
    }
  };

  // same pipeline as conv_igemm_kernel: registers hold step s+1 while step s is multiplied out of LDS
  // buffer s&1; half-way they are written to the other buffer and the loads of step s+2 are issued.
  if (m_begin < m_end) {
    load_tiles(m_begin);
    store_tiles(0);
  }
  __syncthreads();
  if (m_begin + WP < m_end) load_tiles(m_begin + WP);
  int cur = 0;
  for (int mb = m_begin; mb < m_end; mb += WP) {
    compute_half(cur, 0);
    if (mb + WP < m_end) {
      store_tiles(cur ^ 1);
      if (mb + 2 * WP < m_end) load_tiles(mb + 2 * WP);
    }
    compute_half(cur, WP / 2);
    if (do_bias && tid < BC) {
      const float* Dc = Ds + cur * (WP * BC);
#pragma unroll 8
      for (int r = 0; r < WP; ++r) bsum += Dc[r * BC + tid];
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- write the partial tile: slab[split][co][k] ----
  float* out = p.slab + (long long)split * p.Cout * p.Ktot;
#pragma unroll
  for (int s = 0; s < MT; ++s) {
#pragma unroll
    for (int u = 0; u < NTK; ++u) {
      const int k = kc0 + wn * (32 * NTK) + u * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int co = co0 + wm * WC + s * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < p.Cout && k < p.Ktot) out[(long long)co * p.Ktot + k] = acc[s][u][r];
      }
    }
  }
  if (do_bias && tid < BC && co0 + tid < p.Cout) p.bias_slab[(long long)split * p.Cout + co0 + tid] = bsum;
}

// dw[i] = beta*dw[i] + sum_s slab[s][i] (i < n) and, in the same launch, db[j] likewise from bias_slab
__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, long long n,
                                   const float* __restrict__ bias_slab, float* __restrict__ db, int nb,
                                   int nsplit, float beta, float beta_b) {
  const long long total = n + (db != nullptr ? nb : 0);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    if (i < n) {
      for (int k = 0; k < nsplit; ++k) s += slab[(long long)k * n + i];
      dw[i] = (beta != 0.f ? beta * dw[i] : 0.f) + s;
    } else {
      const long long j = i - n;
      for (int k = 0; k < nsplit; ++k) s += bias_slab[(long long)k * nb + j];
      db[j] = (beta_b != 0.f ? beta_b * db[j] : 0.f) + s;
    }
  }
}

// resident 512-thread blocks per CU of each instantiation (occupancy query, cached; 2 when unknown)
int wgrad_blocks_per_cu(int bc, bool aligned) {
  static int cache[2][2] = {{0, 0}, {0, 0}};
  int& c = cache[bc == 128][aligned];
  if (c == 0) {
    int n = 0;
    hipError_t e;
    if (bc == 128) {
      if (aligned) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_wgrad_kernel<128, true>, WTHR, 0);
      else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_wgrad_kernel<128, false>, WTHR, 0);
    } else {
      if (aligned) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_wgrad_kernel<64, true>, WTHR, 0);
      else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_wgrad_kernel<64, false>, WTHR, 0);
    }
    if (e != hipSuccess) (void)hipGetLastError();  // e.g. no device in the build container
    c = (e == hipSuccess && n > 0) ? std::min(n, 8) : 2;
  }
  return c;
}

// dw[co][kh][kw][ci] += sum over the 4 phases of dwc[phase][co][dh(a,kh)][dw(b,kw)][ci]: chain rule of the
// weight merge of upw_combine_kernel (conv_igemm.hip): kernel row kh feeds tap dh = {0,0,1,1,2} (a=0) or
// {0,1,1,2,2} (a=1) of phase a.
__global__ void upw_scatter_kernel(const float* __restrict__ dwc, float* __restrict__ dw, int Cout, int Cin) {
  const long long total = (long long)Cout * 25 * Cin;
  const long long per_phase = (long long)Cout * 9 * Cin;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int ci = (int)(r % Cin); r /= Cin;
    const int kw = (int)(r % 5); r /= 5;
    const int kh = (int)(r % 5); r /= 5;
    const int co = (int)r;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int dh = a == 0 ? (kh >> 1) : ((kh + 1) >> 1);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int dw_ = b == 0 ? (kw >> 1) : ((kw + 1) >> 1);
        s += dwc[(a * 2 + b) * per_phase + (((long long)co * 3 + dh) * 3 + dw_) * Cin + ci];
      }
    }
    dw[i] += s;
  }
}

struct WgradPlan {
  int bc, k_tiles, c_tiles, nsplit, pix_per_split;
  size_t slab_bytes, bias_bytes;
};

// split-K plan of one launch: M pixels reduced, Cout x Ktot outputs
void plan_launch(int M, int Ktot, int Cout, bool aligned, WgradPlan* pl) {
  pl->bc = Cout <= 64 ? 64 : 128;
  pl->k_tiles = cdiv(Ktot, pl->bc == 128 ? 128 : 256);
  pl->c_tiles = cdiv(Cout, pl->bc);
  const int tiles = pl->k_tiles * pl->c_tiles;
  // One full round of resident blocks: 256 CUs x blocks/CU the register budget admits.  A grid of e.g.
  // 1025 equal blocks on 1024 slots costs two rounds, so the split count is floored to fit one round;
  // at least 128 pixels per split, at most 512 splits.
  const int slots = 256 * wgrad_blocks_per_cu(pl->bc, aligned);
  int want = std::max(1, slots / tiles);
  int max_by_pix = std::max(1, M / 128);
  int ns = std::max(1, std::min(std::min(want, max_by_pix), 512));
  int pps = cdiv(M, ns);
  pps = (pps + WP - 1) / WP * WP;
  ns = cdiv(M, pps);
  pl->nsplit = ns;
  pl->pix_per_split = pps;
  pl->slab_bytes = align_up((size_t)ns * Cout * Ktot * sizeof(float), 256);
  pl->bias_bytes = align_up((size_t)ns * Cout * sizeof(float), 256);
}

// one split-K launch + deterministic slab reduction into dw (and db)
int run_wgrad(WgradParams p, const WgradPlan& pl, bool aligned, float* dw, float* db, float beta, float beta_b,
              void* ws, hipStream_t st) {
  p.slab = reinterpret_cast<float*>(ws);
  p.bias_slab = db ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + pl.slab_bytes) : nullptr;
  p.pix_per_split = pl.pix_per_split;
  dim3 grid((unsigned)pl.k_tiles, (unsigned)pl.c_tiles, (unsigned)pl.nsplit);
  if (pl.bc == 64) {
    if (aligned) hipLaunchKernelGGL((conv_wgrad_kernel<64, true>), grid, dim3(WTHR), 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<64, false>), grid, dim3(WTHR), 0, st, p);
  } else {
    if (aligned) hipLaunchKernelGGL((conv_wgrad_kernel<128, true>), grid, dim3(WTHR), 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<128, false>), grid, dim3(WTHR), 0, st, p);
  }
  MUNIT_CHECK_LAUNCH("conv_wgrad");
  const long long n = (long long)p.Cout * p.Ktot;
  const int blocks = (int)std::min<long long>((n + p.Cout + 255) / 256, 4096);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.slab, dw, n, p.bias_slab, db, p.Cout,
                     pl.nsplit, beta, beta_b);
  MUNIT_CHECK_LAUNCH("slab_reduce");
  return MUNIT_OK;
}

bool subpixel_wgrad_ok(const munit_conv_desc* d) {
  return d->upsample == 1 && d->KH == 5 && d->KW == 5 && d->pad == 2 && d->stride == 1 &&
         d->pad_mode == MUNIT_PAD_REFLECT && d->Cin % 4 == 0 && d->H >= 3 && d->W >= 3 &&
         !getenv("MUNIT_DEBUG_NO_SUBPIXEL");
}

// sub-pixel wgrad of an up-sampling conv: workspace = [dwc: 4*Cout*9*Cin][slabs of the largest launch]
struct SubpixelPlan {
  WgradPlan phase, frame;
  size_t dwc_bytes, slab_bytes;
};
void plan_subpixel(const munit_conv_desc* d, SubpixelPlan* sp) {
  const bool aligned = d->Cin % 4 == 0;
  plan_launch(d->B * (d->H - 2) * (d->W - 2), 9 * d->Cin, d->Cout, aligned, &sp->phase);
  const int Ho = 2 * d->H, Wo = 2 * d->W;
  plan_launch(d->B * (4 * Wo + 4 * (Ho - 4)), 25 * d->Cin, d->Cout, aligned, &sp->frame);
  sp->dwc_bytes = align_up((size_t)4 * d->Cout * 9 * d->Cin * sizeof(float), 256);
  sp->slab_bytes = std::max(sp->phase.slab_bytes + sp->phase.bias_bytes, sp->frame.slab_bytes + sp->frame.bias_bytes);
}

}  // namespace

extern "C" size_t munit_conv2d_wgrad_workspace_bytes(const munit_conv_desc* d) {
  int Ho, Wo;
  if (munit_conv2d_out_hw(d, &Ho, &Wo)) return 0;
  if (munit_small_wgrad_supported(d) && !getenv("MUNIT_DEBUG_NO_SMALL_WGRAD")) return munit_small_wgrad_workspace(d, Ho);
  if (subpixel_wgrad_ok(d)) {
    SubpixelPlan sp;
    plan_subpixel(d, &sp);
    return sp.dwc_bytes + sp.slab_bytes;
  }
  WgradPlan pl;
  plan_launch(d->B * Ho * Wo, d->KH * d->KW * d->Cin, d->Cout, d->Cin % 4 == 0, &pl);
  return pl.slab_bytes + pl.bias_bytes;
}

extern "C" int munit_conv2d_wgrad(const munit_conv_desc* d, const float* x, const float* dy, float* dw,
                                  float* db, float beta, void* ws, size_t ws_bytes,
                                  munit_stream_t stream) {
  int Ho, Wo;
  int rc = munit_conv2d_out_hw(d, &Ho, &Wo);
  if (rc) return rc;
  MUNIT_CHECK_ARG(x && dy && dw && ws, "conv2d_wgrad: null pointer");
  if (ws_bytes < munit_conv2d_wgrad_workspace_bytes(d)) {
    munit_set_error("conv2d_wgrad: workspace %zu < %zu", ws_bytes, munit_conv2d_wgrad_workspace_bytes(d));
    return MUNIT_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (munit_small_wgrad_supported(d) && !getenv("MUNIT_DEBUG_NO_SMALL_WGRAD"))
    return munit_small_wgrad(d, Ho, Wo, x, dy, dw, db, beta, ws, st);
  const bool aligned = d->Cin % 4 == 0;
  WgradParams p{};
  p.x = x; p.dy = dy;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin;
  p.ups = d->upsample; p.Hu = d->H << p.ups; p.Wu = d->W << p.ups;
  p.Ho = Ho; p.Wo = Wo; p.Cout = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  p.reflect = d->pad_mode == MUNIT_PAD_REFLECT;
  p.Ktot = d->KH * d->KW * d->Cin; p.M = d->B * Ho * Wo;
  p.dy_sw = d->Cout; p.dy_sh = (long long)Wo * d->Cout; p.dy_sb = (long long)Ho * Wo * d->Cout; p.dy_off = 0;
  if (subpixel_wgrad_ok(d)) {
    // dw = (frame pixels, generic 25-tap gather) + scatter of the 4 phase gradients (interior pixels, 3x3
    // VALID conv over the source against every other dy row/column): 36 instead of 100 MACs per source
    // pixel and channel pair.
    SubpixelPlan sp;
    plan_subpixel(d, &sp);
    float* dwc = reinterpret_cast<float*>(ws);
    void* slabs = reinterpret_cast<char*>(ws) + sp.dwc_bytes;
    WgradParams f = p;
    f.frame = 1;
    f.M = d->B * (4 * Wo + 4 * (Ho - 4));
    rc = run_wgrad(f, sp.frame, aligned, dw, db, beta, beta, slabs, st);
    if (rc) return rc;
    for (int ph = 0; ph < 4; ++ph) {
      const int a = ph >> 1, b = ph & 1;
      WgradParams q = p;
      q.ups = 0; q.Hu = d->H; q.Wu = d->W;
      q.Ho = d->H - 2; q.Wo = d->W - 2;          // interior source pixels i = oh+1, j = ow+1
      q.KH = 3; q.KW = 3; q.pad = 0; q.reflect = 0;  // taps i-1..i+1 = oh..oh+2: a VALID 3x3 gather
      q.Ktot = 9 * d->Cin; q.M = d->B * q.Ho * q.Wo;
      q.dy_sw = 2 * d->Cout; q.dy_sh = (long long)2 * Wo * d->Cout;
      q.dy_off = ((long long)(2 + a) * Wo + 2 + b) * d->Cout;
      rc = run_wgrad(q, sp.phase, aligned, dwc + (long long)ph * d->Cout * 9 * d->Cin, db, 0.0f, 1.0f, slabs, st);
      if (rc) return rc;
    }
    const long long total = (long long)d->Cout * 25 * d->Cin;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(upw_scatter_kernel, dim3(blocks), dim3(256), 0, st, dwc, dw, d->Cout, d->Cin);
    MUNIT_CHECK_LAUNCH("upw_scatter");
    return MUNIT_OK;
  }
  WgradPlan pl;
  plan_launch(p.M, p.Ktot, d->Cout, aligned, &pl);
  return run_wgrad(p, pl, aligned, dw, db, beta, beta, ws, st);
}
