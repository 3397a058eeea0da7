// Backward-weight of the convolutions (autograd of networks.py:691-696) as a split-K
// MFMA GEMM:  dw[co][k] = sum_m dy[m][co] * A[m][k],  k = (kh*KW+kw)*Cin+ci,
// A = the same on-the-fly im2col gather the forward uses (reflect/zero pad, stride,
// fused nearest upsample).  The reduction runs over m = output pixels (up to 524288*B),
// so the grid is (k tiles) x (cout tiles) x (pixel splits); each split writes a partial
// slab and slab_reduce_kernel sums them in a fixed order (deterministic, no atomics).
// The bias gradient (column sums of dy) rides along in the k-tile-0 blocks.
//
// Tile: BC (cout) x 128 or 256 (k) accumulators, 32 pixels per step.  dy and A tiles are staged
// in LDS as [pixel][channel]; v_mfma_f32_16x16x4_f32 operands are fetched with ds_read_b128 / b64
// (one wide read feeds 4 / 2 MFMA tiles, see Frag).
#include "common.h"
#include "wino.h"
#include <cstdlib>
#include <type_traits>

namespace {

struct WgradParams {
  const void* x;     // fp32, or bf16 when the kernel's XB is set
  const void* dy;    // fp32, or bf16 when DYB is set
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
  int frame;  // > 0: GEMM rows enumerate only the border frame of the Ho x Wo output, `frame` pixels wide (see decode_pixel)
  unsigned x_bytes, dy_bytes;  // extents for the buffer descriptors of the FAST loader (0 when >= 2 GiB)
  int ct;                      // compute type (FAST variants only): 0 fp32 MFMA, 1 bf16 operands, 2 f32x3
  int x_bf16, dy_bf16;         // element types in HBM (FAST loader only)
};

// 16 bytes of zeros: rows past the end of a split / taps in zero padding are pointed here by the direct-to-LDS loader
__device__ const float munit_wgrad_zero16[4] = {0.f, 0.f, 0.f, 0.f};
constexpr int DMA_MAX_HW = 1024;   // Ho + Wo entries of the direct-to-LDS loader's offset tables

// row index -> (sample, output row, output column); same enumeration as conv_igemm.hip
__device__ inline void decode_pixel(int m, int Ho, int Wo, int frame, int& b, int& oh, int& ow) {
  if (!frame) {
    const int HoWo = Ho * Wo;
    b = m / HoWo;
    const int rem = m - b * HoWo;
    oh = rem / Wo;
    ow = rem - oh * Wo;
  } else {
    const int F = frame, F2 = 2 * frame;   // ring width in pixels
    const int nb = F2 * Wo + F2 * (Ho - F2);
    b = m / nb;
    int r = m - b * nb;
    if (r < F2 * Wo) {
      const int q = r / Wo;
      ow = r - q * Wo;
      oh = q < F ? q : Ho - F2 + q;
    } else {
      r -= F2 * Wo;
      const int rr = r / F2, q = r - rr * F2;
      oh = F + rr;
      ow = q < F ? q : Wo - F2 + q;
    }
  }
}

constexpr int WP = 32;   // pixels per step

constexpr int WTHR = 512;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;   // buffer-load offset past any (< 2 GiB) tensor: the load returns zeros

// Operand fragments of v_mfma_f32_16x16x4_f32 straight out of a [pixel][channel] LDS tile.  Lane l
// supplies (row j = l&15, contraction index g = l>>4).  One wide read of NTILE consecutive channels
// starting at NTILE*j feeds NTILE MFMA tiles at once: tile t holds channels {NTILE*j + t}, a permutation
// of the wave's channel range that the epilogue undoes.  b128 rows (NTILE=4) are conflict-free as they
// are; b64 rows (NTILE=2) XOR the column with 32*(pixel&1) so that the two pixel rows of a 32-lane group
// fall on different banks.
template <int NTILE>
struct Frag;
template <>
struct Frag<4> {
  f32x4 v;
  __device__ inline void load(const float* row, int col, int /*swz*/) { v = *reinterpret_cast<const f32x4*>(row + col); }
  __device__ inline float at(int t) const { return v[t]; }
};
template <>
struct Frag<2> {
  f32x2 v;
  __device__ inline void load(const float* row, int col, int swz) { v = *reinterpret_cast<const f32x2*>(row + (col ^ swz)); }
  __device__ inline float at(int t) const { return v[t]; }
};

// FAST = Cin % 4 == 0 && Cout % 4 == 0 && linear pixel enumeration && tensors < 2 GiB: the tile loads are
// straight-line buffer loads with 32-bit offsets (invalid rows get an out-of-range offset and read as
// zero), the pixel walk is a carry chain without divisions -- ~1/4 of the generic loader's VALU work,
// which at 16 MFMAs per wave and step was the limiter.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// bf16 operand images (BF16 mode): [32 pixel rows][256 B = 128 channels], the 16-byte chunk ch of a row stored at
// ch ^ (((row&3)<<2) | ((row>>2)&3)).  The contraction index (pixel) is the slow dimension, so the MFMA
// operands (8 consecutive pixels of one channel per lane) are fetched with ds_read_b64_tr_b16, which hands
// each lane a COLUMN of a 4-row x 16-channel block; this XOR keeps both those reads and the 8-byte stores
// conflict-free (layout (b) of cdna_hip_programming.md T10).
__device__ inline int img_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// CT: 0 fp32 MFMA, 1 bf16 operands, 2 f32x3 (three exact bf16 planes per operand, six product terms; see
// conv_igemm.hip).  f32x3 keeps one LDS buffer of 3 planes (two barriers per step).
// CT == 3 (fp32, FAST conditions, Cin % WKT == 0 so that one filter tap serves the whole k-tile of a block): the tiles
// travel global -> LDS directly (global_load_lds_dwordx4: no VGPR staging, no ds_write), and the gathered source offset
// of an output pixel is rowoff[oh] + coloff[ow] from two small LDS tables built once per block (the reflect / zero-pad /
// upsample coordinate map of that tap) instead of ~45 VALU instructions per row and step.
// XB / DYB (FAST loader only): x / dy are bf16 tensors in HBM (bf16 storage mode); they are widened on load, so every
// compute type works on them (CT 1 rounds back to the same bf16 values: exact).
template <int BC, bool ALIGNED, bool FAST, int CT = 0, bool XB = false, bool DYB = false>
__global__ __launch_bounds__(WTHR, 4) void conv_wgrad_kernel(WgradParams p) {
  constexpr bool DMA = CT == 3;
  static_assert(!(XB || DYB) || (FAST && !DMA), "bf16 tensors: FAST register loader only");
  constexpr bool BF16 = CT == 1 || CT == 2;
  static_assert(!DMA || (FAST && ALIGNED), "direct-to-LDS loads: FAST variants only");
  constexpr int NPL = CT == 2 ? 3 : 1;
  static_assert(!BF16 || FAST, "the bf16 variants use the FAST loader");
  // 8 waves: 2 along cout x 4 along k.  Tile BC x WKT with WKT = 128 (BC=128) or 256 (BC=64): every wave
  // owns eight 16x16 accumulators either way (4 cout tiles x 2 k tiles, or 2 x 4).
  constexpr int WKT = BC == 128 ? 128 : 256;
  constexpr int WC = BC / 2;            // cout rows per wave: 64 or 32
  constexpr int WK = WKT / 4;           // k columns per wave: 32 or 64
  constexpr int MT = WC / 16;           // 16-row MFMA tiles per wave along cout: 4 or 2
  constexpr int NT = WK / 16;           // 16-col MFMA tiles per wave along k: 2 or 4
  constexpr int XQ = WKT / 4;           // float4 per gathered row
  constexpr int XRPT = WTHR / XQ;       // gather rows covered per pass of the block: 16 or 8
  constexpr int XROWS = WP / XRPT;      // gather rows per loader thread: 2 or 4
  constexpr int DQ = BC / 4;            // float4 per dy row
  constexpr int DRPT = WTHR / DQ;       // dy rows covered per pass of the block: 16 or 32
  constexpr int DROWS = WP / DRPT;      // dy rows per loader thread: 2 or 1
  constexpr int SWZ_D = MT == 2 ? 32 : 0;
  constexpr int SWZ_X = NT == 2 ? 32 : 0;
  static_assert(XRPT % 2 == 0 && DRPT % 2 == 0, "row parity of a loader thread must not change between passes");
  // one allocation: fp32 mode carves double-buffered dy / x tiles (one barrier per step), the bf16 modes their images
  __shared__ __attribute__((aligned(16))) float smem[2 * WP * BC + 2 * WP * WKT + (DMA ? DMA_MAX_HW : 0)];
  float* const Ds = smem;
  float* const Xs = smem + 2 * WP * BC;
  int* const offtab = reinterpret_cast<int*>(smem + 2 * WP * BC + 2 * WP * WKT);   // DMA: rowoff[Ho], coloff[Wo]

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
    // DMA: the lane's LDS position is fixed (wave base + 16 B * lane), so the swizzle is applied to what it fetches
    int k = kc0 + (DMA ? (xq ^ (((xr0 & 1) ? SWZ_X : 0) >> 2)) : xq) * 4;
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
  const int dq_g = DMA ? (dq ^ (((dr0 & 1) ? SWZ_D : 0) >> 2)) : dq;   // global float4 column this thread fetches
  // LDS destinations of this thread's float4s (row parity is the same for all of its rows)
  const int xs_col = (xq * 4) ^ ((xr0 & 1) ? SWZ_X : 0);
  const int ds_col = (dq * 4) ^ ((dr0 & 1) ? SWZ_D : 0);

  f32x4 rx[XROWS], rd[DROWS];

  // FAST: pixel coordinates of this thread's gather rows and dy rows, walked 32 pixels per step (one
  // division per row up front); the generic loader decodes every step instead (fewer live registers)
  int px_oh[FAST ? XROWS : 1], px_ow[FAST ? XROWS : 1], px_b[FAST ? XROWS : 1];
  int dp_oh[FAST ? DROWS : 1], dp_ow[FAST ? DROWS : 1], dp_b[FAST ? DROWS : 1];
  if constexpr (FAST) {
#pragma unroll
    for (int i = 0; i < XROWS; ++i) {
      int m = m_begin + xr0 + XRPT * i;
      decode_pixel(m < p.M ? m : 0, p.Ho, p.Wo, 0, px_b[i], px_oh[i], px_ow[i]);
    }
#pragma unroll
    for (int i = 0; i < DROWS; ++i) {
      int m = m_begin + dr0 + DRPT * i;
      decode_pixel(m < p.M ? m : 0, p.Ho, p.Wo, 0, dp_b[i], dp_oh[i], dp_ow[i]);
    }
  }
  // FAST: a step of 32 pixels = step_b samples + step_h rows + step_w columns (uniform), applied with carries
  const int hw = p.Ho * p.Wo;
  const int step_b = WP / hw;
  const int step_h = (WP - step_b * hw) / p.Wo;
  const int step_w = WP - step_b * hw - step_h * p.Wo;
  // FAST: float offset of the dy row, walked with the same carries
  int dp_off[DROWS];
  const int d_step = step_b * (int)p.dy_sb + step_h * (int)p.dy_sh + step_w * p.dy_sw;
  const int d_carry_w = (int)p.dy_sh - p.Wo * p.dy_sw;          // column wrapped: next row
  const int d_carry_h = (int)p.dy_sb - p.Ho * (int)p.dy_sh;     // row wrapped: next sample
  if constexpr (FAST) {
#pragma unroll
    for (int i = 0; i < DROWS; ++i)
      dp_off[i] = dp_b[i] * (int)p.dy_sb + dp_oh[i] * (int)p.dy_sh + dp_ow[i] * p.dy_sw + (int)p.dy_off + co0 + dq_g * 4;
  }
  const bool d_col_ok = co0 + dq_g * 4 < p.Cout;

  auto load_tiles = [&](int mbase) {
    if constexpr (FAST) {
      const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, p.dy_bytes, 0x00020000);
      // 4 consecutive elements at element offset `off` (or zeros): 16 bytes of fp32, or 8 bytes of bf16 widened
      auto fetch4 = [&](const __amdgpu_buffer_rsrc_t& res, bool ok, unsigned off, auto is_bf16) -> f32x4 {
        if constexpr (decltype(is_bf16)::value) {
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(res, ok ? off * 2u : OOB, 0, 0);
          const bf16v4 h = __builtin_bit_cast(bf16v4, raw);
          return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        } else {
          return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res, ok ? off * 4u : OOB, 0, 0));
        }
      };
#pragma unroll
      for (int i = 0; i < XROWS; ++i) {
        const int m = mbase + xr0 + XRPT * i;
        const int ih = src_coord(__mul24(px_oh[i], p.stride) - p.pad + ekh[0], p.Hu, p.ups, p.reflect);
        const int iw = src_coord(__mul24(px_ow[i], p.stride) - p.pad + ekw[0], p.Wu, p.ups, p.reflect);
        const bool ok = m < m_end && eok[0] && ih >= 0 && iw >= 0;
        // 24-bit multiplies: the host checked B*H*W < 2^23 and x_bytes < 2^31
        const int pix = __mul24(__mul24(px_b[i], p.H) + ih, p.W) + iw;
        const unsigned off = (unsigned)(__mul24(pix, p.Cin) + eci[0]);
        rx[i] = fetch4(xres, ok, off, std::integral_constant<bool, XB>{});
        // walk 32 pixels on
        int ow = px_ow[i] + step_w;
        const bool cw = ow >= p.Wo;
        ow -= cw ? p.Wo : 0;
        int oh = px_oh[i] + step_h + (cw ? 1 : 0);
        const bool ch = oh >= p.Ho;
        oh -= ch ? p.Ho : 0;
        px_ow[i] = ow;
        px_oh[i] = oh;
        px_b[i] += step_b + (ch ? 1 : 0);
      }
#pragma unroll
      for (int i = 0; i < DROWS; ++i) {
        const int m = mbase + dr0 + DRPT * i;
        const bool ok = m < m_end && d_col_ok;
        rd[i] = fetch4(dres, ok, (unsigned)dp_off[i], std::integral_constant<bool, DYB>{});
        int ow = dp_ow[i] + step_w;
        const bool cw = ow >= p.Wo;
        ow -= cw ? p.Wo : 0;
        int oh = dp_oh[i] + step_h + (cw ? 1 : 0);
        const bool ch = oh >= p.Ho;
        oh -= ch ? p.Ho : 0;
        dp_ow[i] = ow;
        dp_oh[i] = oh;
        dp_off[i] += d_step + (cw ? d_carry_w : 0) + (ch ? d_carry_h : 0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < XROWS; ++i) {
      const int m = mbase + xr0 + XRPT * i;
      const bool mok = m < m_end;
      int b, oh, ow;
      decode_pixel(mok ? m : 0, p.Ho, p.Wo, p.frame, b, oh, ow);
      const long long base = (long long)b * p.H * p.W;
      int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if constexpr (ALIGNED) {
        int ih = src_coord(ih0 + ekh[0], p.Hu, p.ups, p.reflect);
        int iw = src_coord(iw0 + ekw[0], p.Wu, p.ups, p.reflect);
        if (mok && eok[0] && ih >= 0 && iw >= 0)
          v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.x) + (base + (long long)ih * p.W + iw) * p.Cin + eci[0]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int ih = src_coord(ih0 + ekh[e], p.Hu, p.ups, p.reflect);
          int iw = src_coord(iw0 + ekw[e], p.Wu, p.ups, p.reflect);
          float s = 0.f;
          if (mok && eok[e] && ih >= 0 && iw >= 0) s = reinterpret_cast<const float*>(p.x)[(base + (long long)ih * p.W + iw) * p.Cin + eci[e]];
          v[e] = s;
        }
      }
      rx[i] = v;
    }
#pragma unroll
    for (int i = 0; i < DROWS; ++i) {
      const int m = mbase + dr0 + DRPT * i;
      const bool mok = m < m_end;
      int b, oh, ow;
      decode_pixel(mok ? m : 0, p.Ho, p.Wo, p.frame, b, oh, ow);
      const float* ptr = reinterpret_cast<const float*>(p.dy) + (long long)b * p.dy_sb + (long long)oh * p.dy_sh +
                         (long long)ow * p.dy_sw + p.dy_off + co0 + dq * 4;
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
  // ---- direct-to-LDS loader -------------------------------------------------------------------------------------
  constexpr int OFF_NONE = -(1 << 30);   // rowoff / coloff of a tap position in zero padding: any sum with it stays negative
  if constexpr (DMA) {
    // one tap per block (Cin % WKT == 0): float offset of the gathered row / column inside a sample
    for (int t = tid; t < p.Ho + p.Wo; t += WTHR) {
      if (t < p.Ho) {
        const int ih = src_coord(t * p.stride - p.pad + ekh[0], p.Hu, p.ups, p.reflect);
        offtab[t] = ih >= 0 ? ih * p.W * p.Cin : OFF_NONE;
      } else {
        const int iw = src_coord((t - p.Ho) * p.stride - p.pad + ekw[0], p.Wu, p.ups, p.reflect);
        offtab[t] = iw >= 0 ? iw * p.Cin : OFF_NONE;
      }
    }
    __syncthreads();
  }
  const int hwc = p.H * p.W * p.Cin;
  auto dma_tiles = [&](int mbase, int buf) {
    if constexpr (DMA) {
      const int w = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
      for (int i = 0; i < XROWS; ++i) {
        const int m = mbase + xr0 + XRPT * i;
        const int off = px_b[i] * hwc + offtab[px_oh[i]] + offtab[p.Ho + px_ow[i]] + eci[0];
        const bool ok = m < m_end && eok[0] && off >= 0;
        const float* g = ok ? reinterpret_cast<const float*>(p.x) + off : munit_wgrad_zero16;
        // wave w fills rows [w * 64 / XQ, ...) of pass i: 1 KiB per wave instruction, lane l lands at base + 16 B * l
        float* l = Xs + buf * (WP * WKT) + (XRPT * i) * WKT + w * 256;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        int ow = px_ow[i] + step_w;
        const bool cw = ow >= p.Wo;
        ow -= cw ? p.Wo : 0;
        int oh = px_oh[i] + step_h + (cw ? 1 : 0);
        const bool ch = oh >= p.Ho;
        oh -= ch ? p.Ho : 0;
        px_ow[i] = ow;
        px_oh[i] = oh;
        px_b[i] += step_b + (ch ? 1 : 0);
      }
#pragma unroll
      for (int i = 0; i < DROWS; ++i) {
        const int m = mbase + dr0 + DRPT * i;
        const bool ok = m < m_end && d_col_ok;
        const float* g = ok ? reinterpret_cast<const float*>(p.dy) + dp_off[i] : munit_wgrad_zero16;
        float* l = Ds + buf * (WP * BC) + (DRPT * i) * BC + w * 256;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        int ow = dp_ow[i] + step_w;
        const bool cw = ow >= p.Wo;
        ow -= cw ? p.Wo : 0;
        int oh = dp_oh[i] + step_h + (cw ? 1 : 0);
        const bool ch = oh >= p.Ho;
        oh -= ch ? p.Ho : 0;
        dp_ow[i] = ow;
        dp_oh[i] = oh;
        dp_off[i] += d_step + (cw ? d_carry_w : 0) + (ch ? d_carry_h : 0);
      }
    }
  };
  constexpr int NXI = WKT / 128;              // X images per buffer (BF16 mode); image 0 of a buffer is dy
  static_assert((CT == 2 ? 3 : 2) * (1 + NXI) * 8192 <= (int)sizeof(float) * (2 * WP * BC + 2 * WP * WKT), "bf16 images must fit the LDS allocation");
  char* const img_base = reinterpret_cast<char*>(smem);
  auto store_tiles = [&](int buf) {
    if constexpr (BF16) {
      // buffer = NPL planes x (1 dy image + NXI x images) x 8 KiB
      char* ib = img_base + buf * (NPL * (1 + NXI) * 8192);
      auto put = [&](char* dst, f32x4 v) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          bf16x4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
          *reinterpret_cast<bf16x4*>(dst + pl * ((1 + NXI) * 8192)) = h;
          if (pl + 1 < NPL) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] -= (float)h[e];
          }
        }
      };
#pragma unroll
      for (int i = 0; i < XROWS; ++i) {
        const int quad = xq & 31;
        put(ib + (1 + (xq >> 5)) * 8192 + img_off(xr0 + XRPT * i, quad >> 1) + 8 * (quad & 1), rx[i]);
      }
#pragma unroll
      for (int i = 0; i < DROWS; ++i) put(ib + img_off(dr0 + DRPT * i, dq >> 1) + 8 * (dq & 1), rd[i]);
      return;
    }
    float* Xd = Xs + buf * (WP * WKT);
    float* Dd = Ds + buf * (WP * BC);
#pragma unroll
    for (int i = 0; i < XROWS; ++i) *reinterpret_cast<f32x4*>(&Xd[(xr0 + XRPT * i) * WKT + xs_col]) = rx[i];
#pragma unroll
    for (int i = 0; i < DROWS; ++i) *reinterpret_cast<f32x4*>(&Dd[(dr0 + DRPT * i) * BC + ds_col]) = rd[i];
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][t][r] = 0.f;
  float bsum = 0.f;  // bias column sum: thread tid < BC owns channel co0 + tid
  const bool do_bias = p.bias_slab != nullptr && blockIdx.x == 0;

  const int fj = lane & 15;   // operand row / column inside a 16x16 tile
  const int fg = lane >> 4;   // pixel inside a group of 4
  const int a_col = wm * WC + MT * fj;
  const int b_col = wn * WK + NT * fj;
  const int a_swz = (fg & 1) ? SWZ_D : 0;
  const int b_swz = (fg & 1) ? SWZ_X : 0;
  // 16 of the 32 pixels of a step: wide LDS reads for GB pixel groups at a time, then GB * MT * NT MFMAs
  // (the generic loader keeps more per-row state in registers and takes smaller batches)
  constexpr int GB = (FAST || BC == 128) ? 4 : 2;
  // BF16: operand of the 16-channel tile starting at channel c0 of image `im`: pixels 8*fg .. 8*fg+7 of channel
  // c0 + fj, from two transposed 4x16 block reads (lane 4q+p of a 16-lane group addresses row q, channels 4p..4p+3)
  auto tr_frag = [&](const char* im, int c0) -> bf16x8 {
    const int q = fj >> 2, pp = lane & 3;
    const int ch = (c0 >> 3) + (pp >> 1);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(im + img_off(8 * fg + q, ch) + 8 * (pp & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(im + img_off(8 * fg + 4 + q, ch) + 8 * (pp & 1)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };
  auto compute_half = [&](int buf, int p0) {
    if constexpr (BF16) {
      // one v_mfma_f32_16x16x32_bf16 per tile pair (and product term) contracts all 32 pixels of the step; half
      // 0 / 1 = first / second half of the tiles of the operand that is NOT kept resident
      const char* ib = img_base + buf * (NPL * (1 + NXI) * 8192);
      constexpr int PLB = (1 + NXI) * 8192;
      const int h = p0 ? 1 : 0;
      if constexpr (NT <= MT) {   // keep the k-side fragments resident, stream the cout tiles
        bf16x8 hb[NPL][NT];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int u = 0; u < NT; ++u) {
            const int col = wn * WK + 16 * u;
            hb[pl][u] = tr_frag(ib + pl * PLB + (1 + (col >> 7)) * 8192, col & 127);
          }
#pragma unroll
        for (int t = h * (MT / 2); t < (h + 1) * (MT / 2); ++t) {
          bf16x8 ha[NPL];
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) ha[pl] = tr_frag(ib + pl * PLB, wm * WC + 16 * t);
#pragma unroll
          for (int sum = NPL - 1; sum >= 0; --sum)
#pragma unroll
            for (int i = 0; i <= sum; ++i)
#pragma unroll
              for (int u = 0; u < NT; ++u)
                acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[i], hb[sum - i][u], acc[t][u], 0, 0, 0);
        }
      } else {                    // Cout <= 64 tile: two cout tiles resident, stream the four k tiles
        bf16x8 ha[NPL][MT];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int t = 0; t < MT; ++t) ha[pl][t] = tr_frag(ib + pl * PLB, wm * WC + 16 * t);
#pragma unroll
        for (int u = h * (NT / 2); u < (h + 1) * (NT / 2); ++u) {
          const int col = wn * WK + 16 * u;
          bf16x8 hb[NPL];
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) hb[pl] = tr_frag(ib + pl * PLB + (1 + (col >> 7)) * 8192, col & 127);
#pragma unroll
          for (int sum = NPL - 1; sum >= 0; --sum)
#pragma unroll
            for (int i = 0; i <= sum; ++i)
#pragma unroll
              for (int t = 0; t < MT; ++t)
                acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[i][t], hb[sum - i], acc[t][u], 0, 0, 0);
        }
      }
      return;
    }
    const float* Dc = Ds + buf * (WP * BC);
    const float* Xc = Xs + buf * (WP * WKT);
#pragma unroll
    for (int s0 = 0; s0 < 4; s0 += GB) {
      Frag<MT> a[GB];
      Frag<NT> b[GB];
#pragma unroll
      for (int s = 0; s < GB; ++s) {
        a[s].load(Dc + (p0 + 4 * (s0 + s) + fg) * BC, a_col, a_swz);
        b[s].load(Xc + (p0 + 4 * (s0 + s) + fg) * WKT, b_col, b_swz);
      }
#pragma unroll
      for (int s = 0; s < GB; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int u = 0; u < NT; ++u)
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].at(t), b[s].at(u), acc[t][u], 0, 0, 0);
      if constexpr (GB != 4) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // same pipeline as conv_igemm_kernel: registers hold step s+1 while step s is multiplied out of LDS
  // buffer s&1; half-way they are written to the other buffer and the loads of step s+2 are issued.
  if constexpr (DMA) {
    // as conv_igemm_kernel's direct-to-LDS pipeline: the tile of step s+1 is issued into the other buffer at the top of
    // step s and published by s_waitcnt vmcnt(0) + the one barrier
    if (m_begin < m_end) dma_tiles(m_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int mb = m_begin; mb < m_end; mb += WP) {
      if (mb + WP < m_end) dma_tiles(mb + WP, cur ^ 1);
      compute_half(cur, 0);
      compute_half(cur, WP / 2);
      if (do_bias && tid < BC) {
        const float* Dc = Ds + cur * (WP * BC);
#pragma unroll 8
        for (int r = 0; r < WP; ++r) bsum += Dc[r * BC + (tid ^ ((r & 1) ? SWZ_D : 0))];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  } else if constexpr (CT == 2) {
    // one buffer of three planes: barrier, split the registers into the planes, issue the next loads, barrier, multiply
    if (m_begin < m_end) load_tiles(m_begin);
    for (int mb = m_begin; mb < m_end; mb += WP) {
      __syncthreads();
      store_tiles(0);
      if (mb + WP < m_end) load_tiles(mb + WP);
      __syncthreads();
      compute_half(0, 0);
      compute_half(0, WP / 2);
      if (do_bias && tid < BC) {
        const char* ib = img_base;
#pragma unroll 8
        for (int r = 0; r < WP; ++r) {
          float v = 0.f;
#pragma unroll
          for (int pl = NPL - 1; pl >= 0; --pl)
            v += (float)*reinterpret_cast<const __bf16*>(ib + pl * ((1 + NXI) * 8192) + img_off(r, tid >> 3) + 2 * (tid & 7));
          bsum += v;
        }
      }
    }
  } else {
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
        if constexpr (BF16) {
          const char* ib = img_base + cur * (NPL * (1 + NXI) * 8192);
  #pragma unroll 8
          for (int r = 0; r < WP; ++r) {
            float v = 0.f;
  #pragma unroll
            for (int pl = NPL - 1; pl >= 0; --pl)   // planes summed small to large: reconstructs the fp32 value exactly
              v += (float)*reinterpret_cast<const __bf16*>(ib + pl * ((1 + NXI) * 8192) + img_off(r, tid >> 3) + 2 * (tid & 7));
            bsum += v;
          }
        } else {
          const float* Dc = Ds + cur * (WP * BC);
  #pragma unroll 8
          for (int r = 0; r < WP; ++r) bsum += Dc[r * BC + (tid ^ ((r & 1) ? SWZ_D : 0))];
        }
      }
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- write the partial tile: slab[split][co][k] ----
  // accumulator (t, u), register r of lane (fj, fg): cout row MT*(4*fg + r) + t, k column NT*fj + u
  float* out = p.slab + (long long)split * p.Cout * p.Ktot;
  if constexpr (BF16) {
    // natural C/D map: accumulator (t, u), register r of lane (fj, fg) = cout row 16t + 4*fg + r, k column 16u + fj
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wm * WC + 16 * t + 4 * fg + r;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          const int k = kc0 + wn * WK + 16 * u + fj;
          if (co < p.Cout && k < p.Ktot) out[(long long)co * p.Ktot + k] = acc[t][u][r];
        }
      }
    if (do_bias && tid < BC && co0 + tid < p.Cout) p.bias_slab[(long long)split * p.Cout + co0 + tid] = bsum;
    return;
  }
  const int k_base = kc0 + wn * WK + NT * fj;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + wm * WC + MT * (4 * fg + r) + t;
      if (co >= p.Cout) continue;
      float* dst = out + (long long)co * p.Ktot + k_base;
      if (ALIGNED && k_base + NT <= p.Ktot) {   // Ktot % 4 == 0 when aligned: NT consecutive columns, vector store
        if constexpr (NT == 2) {
          f32x2 v = {acc[t][0][r], acc[t][1][r]};
          *reinterpret_cast<f32x2*>(dst) = v;
        } else {
          f32x4 v = {acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
          *reinterpret_cast<f32x4*>(dst) = v;
        }
      } else {
#pragma unroll
        for (int u = 0; u < NT; ++u)
          if (k_base + u < p.Ktot) dst[u] = acc[t][u][r];
      }
    }
  }
  if (do_bias && tid < BC && co0 + tid < p.Cout) p.bias_slab[(long long)split * p.Cout + co0 + tid] = bsum;
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 STORAGE backward-weight (x and dy are bf16 tensors): tile 128 (cout) x 128 (k), 64 pixels per step, both
// operands travel global -> LDS directly (global_load_lds_dwordx4) into the [32 pixels][256 B] images that
// ds_read_b64_tr_b16 reads (img_off: the lane picks the global 16-byte chunk that belongs at its fixed landing spot),
// gathered source offsets come from the per-block LDS tables (one filter tap per block: Cin % 128 == 0),
// v_mfma_f32_16x16x32_bf16 with fp32 accumulators, partial slabs in fp32.  Against the register loader of the CT == 1
// kernel (bf16 -> fp32 -> bf16 through VGPRs, 32-pixel steps) this is what a 16x faster matrix pipe needs: the loop is
// bound by how fast tiles arrive, not by arithmetic.  Cout is padded to the 128-wide tile with zero rows.
// ---------------------------------------------------------------------------------------------------------------
constexpr int BS_WP = 64;        // pixels per step
__device__ const float munit_wgrad_zero16b[4] = {0.f, 0.f, 0.f, 0.f};
#ifdef WGB_STAMP   // diagnostic build (tools/wgrad_bf16_stamps.py): shader-clock accounting of the main loop per wave
__device__ long long g_wgb_stamps[8 * 8 * 4];   // [block < 8][wave][DMA issue, fragments + MFMA issue, vmcnt wait, barrier]
#define WGB(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); wgb_acc[i] += t_ - wgb_last; wgb_last = t_; } while (0)
#else
#define WGB(i) do { } while (0)
#endif

// TC = 2: tile 256 (cout) x 256 (k), one block per CU.  At the bf16 matrix rate the 128 x 128 tile is bound by what the L2
// delivers (32 KiB per 2.1 MFLOP step: the trunk layer at B = 32 moved 2.4 GB per launch through the L2 at 6.8 TB/s, 0.18 of the
// matrix peak); the larger tile halves the bytes per FLOP.  Images stay 32 pixels x 128 channels: a stage holds, per operand and
// 32-pixel group, TC of them side by side.
template <int TC>
__global__ __launch_bounds__(WTHR, TC == 1 ? 4 : 2) void conv_wgrad_bf16s_kernel(WgradParams p) {
  constexpr int BC = 128 * TC, WKT = 128 * TC, WC = 64 * TC, WK = 32 * TC, MT = 4 * TC, NT = 2 * TC;
  constexpr int IMG = 8192;                 // one 32-pixel x 128-channel bf16 image
  constexpr int STAGE = 4 * TC * IMG;       // image ((operand * 2 + pixel group) * TC + channel half): dy first, then x
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + 4 * DMA_MAX_HW];
  int* const offtab = reinterpret_cast<int*>(smem + 2 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  // XCD-aware order: workgroups go to the 8 XCDs round-robin by linear id, so the k / cout tiles of ONE pixel split -- which
  // read the same dy and x rows -- would fetch them through eight different L2s.  Every XCD gets a contiguous run of logical
  // blocks instead (tile index fastest), so a split's tiles meet in one L2.
  int blk = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  {
    const int nb = gridDim.x * gridDim.y * gridDim.z, q = nb >> 3, rr = nb & 7, xcd = blk & 7, idx = blk >> 3;
    blk = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  const int bx = blk % gridDim.x, by = (blk / gridDim.x) % gridDim.y, split = blk / (gridDim.x * gridDim.y);
  const int kc0 = bx * WKT, co0 = by * BC;
  const int m_begin = split * p.pix_per_split;
  const int m_end = min(p.M, m_begin + p.pix_per_split);
  const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(p.x);
  const bf16_t* __restrict__ dyg = reinterpret_cast<const bf16_t*>(p.dy);

  // the filter tap and first input channel of this block's k-columns (Cin % WKT == 0: one tap per block)
  const int tap = kc0 / p.Cin, ci0 = kc0 - tap * p.Cin;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  constexpr int OFF_NONE = -(1 << 30);
  for (int t = tid; t < p.Ho + p.Wo; t += WTHR) {
    if (t < p.Ho) {
      const int ih = src_coord(t * p.stride - p.pad + kh, p.Hu, p.ups, p.reflect);
      offtab[t] = ih >= 0 ? ih * p.W * p.Cin : OFF_NONE;
    } else {
      const int iw = src_coord((t - p.Ho) * p.stride - p.pad + kw, p.Wu, p.ups, p.reflect);
      offtab[t] = iw >= 0 ? iw * p.Cin : OFF_NONE;
    }
  }

  // loader: pass i (rows 32i .. 32i+31 = image i), row r = tid / 16, landing position pos = tid % 16 (16 bytes each);
  // the image keeps global chunk c of row r at position c ^ f(r), so this lane fetches chunk pos ^ f(r)
  const int r = tid >> 4, pos = tid & 15;
  const int chunk = pos ^ (((r & 3) << 2) | ((r >> 2) & 3));
  const int x_col = ci0 + 8 * chunk;                       // channel of x   (+ 128 per channel half)
  const int d_col = co0 + 8 * chunk;                       // channel of dy  (+ 128 per channel half)
  int px_b[2], px_oh[2], px_ow[2], dp_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m_begin + 32 * i + r;
    decode_pixel(m < p.M ? m : 0, p.Ho, p.Wo, p.frame, px_b[i], px_oh[i], px_ow[i]);
    dp_off[i] = px_b[i] * (int)p.dy_sb + px_oh[i] * (int)p.dy_sh + px_ow[i] * p.dy_sw + (int)p.dy_off + d_col;
  }
  const int hw = p.Ho * p.Wo;
  const int step_b = BS_WP / hw;
  const int step_h = (BS_WP - step_b * hw) / p.Wo;
  const int step_w = BS_WP - step_b * hw - step_h * p.Wo;
  const int d_step = step_b * (int)p.dy_sb + step_h * (int)p.dy_sh + step_w * p.dy_sw;
  const int d_carry_w = (int)p.dy_sh - p.Wo * p.dy_sw;
  const int d_carry_h = (int)p.dy_sb - p.Ho * (int)p.dy_sh;
  const int hwc = p.H * p.W * p.Cin;
  __syncthreads();   // tables

  auto dma = [&](int mbase, int stage) {
    const int w = __builtin_amdgcn_readfirstlane(wave);
    char* sb = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mbase + 32 * i + r;
      const bool mok = m < m_end;
      const int xo = px_b[i] * hwc + offtab[px_oh[i]] + offtab[p.Ho + px_ow[i]] + x_col;
#pragma unroll
      for (int c = 0; c < TC; ++c) {
        const void* gx = (mok && xo >= 0) ? (const void*)(xg + xo + 128 * c) : (const void*)munit_wgrad_zero16b;
        const void* gd = (mok && d_col + 128 * c < p.Cout) ? (const void*)(dyg + dp_off[i] + 128 * c) : (const void*)munit_wgrad_zero16b;
        // wave w lands rows 4w .. 4w+3 of the image: 1 KiB per wave instruction, lane l at base + 16 l
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gd,
                                         (__attribute__((address_space(3))) void*)(sb + (i * TC + c) * IMG + 1024 * w), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gx,
                                         (__attribute__((address_space(3))) void*)(sb + ((2 + i) * TC + c) * IMG + 1024 * w), 16, 0, 0);
      }
      if (p.frame) {
        // frame enumeration (the 2-pixel border of the sub-pixel form): rows are not a raster walk, decode the next one
        const int mn = m + BS_WP;
        decode_pixel(mn < p.M ? mn : 0, p.Ho, p.Wo, p.frame, px_b[i], px_oh[i], px_ow[i]);
        dp_off[i] = px_b[i] * (int)p.dy_sb + px_oh[i] * (int)p.dy_sh + px_ow[i] * p.dy_sw + (int)p.dy_off + d_col;
        continue;
      }
      int ow = px_ow[i] + step_w;
      const bool cw = ow >= p.Wo;
      ow -= cw ? p.Wo : 0;
      int oh = px_oh[i] + step_h + (cw ? 1 : 0);
      const bool ch = oh >= p.Ho;
      oh -= ch ? p.Ho : 0;
      px_ow[i] = ow;
      px_oh[i] = oh;
      px_b[i] += step_b + (ch ? 1 : 0);
      dp_off[i] += d_step + (cw ? d_carry_w : 0) + (ch ? d_carry_h : 0);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[a][t][e] = 0.f;
  float bsum = 0.f;
  const bool do_bias = p.bias_slab != nullptr && bx == 0;
  const int fj = lane & 15, fg = lane >> 4;
  // operand of the 16-channel tile starting at channel c0: pixels 8 fg .. 8 fg + 7 of channel c0 + fj (see tr_frag of
  // conv_wgrad_kernel: two transposed 4 x 16 block reads)
  auto tr_frag = [&](const char* im0, int c0) -> bf16x8 {   // im0: first of the TC images of (operand, pixel group)
    const int q = fj >> 2, pp = lane & 3;
    const char* im = im0 + (c0 >> 7) * IMG;
    const int ch = ((c0 & 127) >> 3) + (pp >> 1);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(im + img_off(8 * fg + q, ch) + 8 * (pp & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(im + img_off(8 * fg + 4 + q, ch) + 8 * (pp & 1)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };

  if (m_begin < m_end) dma(m_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef WGB_STAMP
  long long wgb_acc[4] = {0, 0, 0, 0};
  long long wgb_last = __builtin_amdgcn_s_memtime();
#endif
  // Stamps (tools/wgrad_bf16_stamps.py, L2-warm loop, 256 x 256 tile): issuing the 8 loads of a stage costs a wave ~1 300 of a
  // step's 4 900 cycles (the CU's address unit takes ~165 cycles per 1 KiB piece when eight waves push at once), the MFMA phase
  // 2 600-3 300 for 2 x 1 024 cycles of matrix work per SIMD, the final wait 60: the compiler has already put an s_waitcnt
  // vmcnt(0) in front of the first LDS read after the loads (it cannot tell the stage being filled from the one being read).
  // Tried on that basis, none kept: staggering the issue between the two waves of a SIMD (-15 % in the warm loop, -2 % inside
  // the step, where the operands come from HBM), one piece between every two row tiles (+25 %), the loads as inline assembly
  // outside the compiler's wait bookkeeping (+5 %: its memory clobbers pin the fragment reads), a ring of four 32-pixel
  // groups (+3 %).
  int cur = 0;
  for (int mb = m_begin; mb < m_end; mb += BS_WP) {
    if (mb + BS_WP < m_end) dma(mb + BS_WP, cur ^ 1);
    WGB(0);
    const char* sb = smem + cur * STAGE;
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // the two 32-pixel images of the stage: one MFMA contracts a whole image
      bf16x8 hb[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) hb[u] = tr_frag(sb + (2 + h) * TC * IMG, wn * WK + 16 * u);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const bf16x8 ha = tr_frag(sb + h * TC * IMG, wm * WC + 16 * t);
#pragma unroll
        for (int u = 0; u < NT; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb[u], acc[t][u], 0, 0, 0);
      }
    }
    if (do_bias && tid < BC) {
#pragma unroll 8
      for (int rr = 0; rr < BS_WP; ++rr)
        bsum += (float)*reinterpret_cast<const bf16_t*>(sb + ((rr >> 5) * TC + (tid >> 7)) * IMG + img_off(rr & 31, (tid & 127) >> 3) +
                                                       2 * (tid & 7));
    }
    WGB(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WGB(2);
    __syncthreads();
    WGB(3);
    cur ^= 1;
  }
#ifdef WGB_STAMP
  if (blk < 8 && lane == 0)
    for (int i = 0; i < 4; ++i) g_wgb_stamps[(blk * 8 + wave) * 4 + i] = wgb_acc[i];
#endif

  // natural C/D map: accumulator (t, u), register e of lane (fj, fg) = cout row 16 t + 4 fg + e, k column 16 u + fj
  float* out = p.slab + (long long)split * p.Cout * p.Ktot;
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = co0 + wm * WC + 16 * t + 4 * fg + e;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int k = kc0 + wn * WK + 16 * u + fj;
        if (co < p.Cout && k < p.Ktot) out[(long long)co * p.Ktot + k] = acc[t][u][e];
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

// Same reduction for small outputs with many splits (the 3-channel first layers: 9.5 k outputs x 512 splits, where
// one thread per output loops over every split for 130 us): 64 outputs per block, the splits dealt to 16 waves in
// a fixed pattern and combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void slab_reduce_wide_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                long long n, const float* __restrict__ bias_slab,
                                                                float* __restrict__ db, int nb, int nsplit, float beta,
                                                                float beta_b) {
  __shared__ float red[16][64];
  const int il = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const long long total = n + (db != nullptr ? nb : 0);
  const long long i = (long long)blockIdx.x * 64 + il;
  float s = 0.f;
  if (i < total) {
    if (i < n) {
      for (int k = kg; k < nsplit; k += 16) s += slab[(long long)k * n + i];
    } else {
      const long long j = i - n;
      for (int k = kg; k < nsplit; k += 16) s += bias_slab[(long long)k * nb + j];
    }
  }
  red[kg][il] = s;
  __syncthreads();
  if (kg == 0 && i < total) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][il];
    if (i < n) dw[i] = (beta != 0.f ? beta * dw[i] : 0.f) + t;
    else db[i - n] = (beta_b != 0.f ? beta_b * db[i - n] : 0.f) + t;
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
      if (aligned) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_wgrad_kernel<128, true, true>, WTHR, 0);
      else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_wgrad_kernel<128, false, false>, WTHR, 0);
    } else {
      if (aligned) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_wgrad_kernel<64, true, true>, WTHR, 0);
      else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_wgrad_kernel<64, false, false>, WTHR, 0);
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

// 3-channel inputs (the image layers): x is re-laid with a zero 4th channel so that the gather is one aligned
// 16-byte load per tap (FAST loader) instead of twelve scalar ones, and the weight gradient is compacted back.
__global__ void pad3to4_kernel(const float* __restrict__ x, f32x4* __restrict__ x4, long long npix) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = {x[3 * i], x[3 * i + 1], x[3 * i + 2], 0.f};
    x4[i] = v;
  }
}
__global__ void compact4to3_kernel(const float* __restrict__ dw4, float* __restrict__ dw, long long n3, float beta) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (long long)gridDim.x * blockDim.x) {
    const long long t = i / 3;
    const int c = (int)(i - 3 * t);
    dw[i] = (beta != 0.f ? beta * dw[i] : 0.f) + dw4[4 * t + c];
  }
}

struct WgradPlan {
  int bc, k_tiles, c_tiles, nsplit, pix_per_split;
  size_t slab_bytes, bias_bytes;
};

// split-K plan of one launch: M pixels reduced, Cout x Ktot outputs
// big256: the bf16-storage kernel's 256 x 256 tile (conv_wgrad_bf16s_kernel<2>, one block per CU)
void plan_launch(int M, int Ktot, int Cout, bool aligned, WgradPlan* pl, bool big256 = false) {
  pl->bc = big256 ? 256 : (Cout <= 64 ? 64 : 128);
  pl->k_tiles = cdiv(Ktot, pl->bc == 128 ? 128 : 256);
  pl->c_tiles = cdiv(Cout, pl->bc);
  const int tiles = pl->k_tiles * pl->c_tiles;
  // One full round of resident blocks: 256 CUs x blocks/CU the register budget admits.  A grid of e.g.
  // 1025 equal blocks on 1024 slots costs two rounds, so the split count is floored to fit one round;
  // at least 128 pixels per split, at most 512 splits.
  const int slots = big256 ? 256 : 256 * wgrad_blocks_per_cu(pl->bc, aligned);
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
  // FAST loader: 32-bit byte offsets and 24-bit multiplies (see the kernel)
  const long long xb = (long long)p.B * p.H * p.W * p.Cin * (p.x_bf16 ? 2 : 4), db_ = (long long)p.B * p.dy_sb * (p.dy_bf16 ? 2 : 4);
  const bool any_bf16 = p.x_bf16 || p.dy_bf16;
  const bool fast_any = aligned && (p.Cout % 4 == 0) && xb < (1ll << 31) && db_ < (1ll << 31) &&
                        (long long)p.B * p.H * p.W < (1ll << 23) && (any_bf16 || !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_FAST_WGRAD"));
  const bool fast = fast_any && !p.frame;
  p.x_bytes = fast ? (unsigned)xb : 0u;
  p.dy_bytes = fast ? (unsigned)db_ : 0u;
  if (any_bf16) {
    // bf16 storage: only the FAST register loader reads bf16 tensors
    const bool dma_ok = p.x_bf16 && p.dy_bf16 && p.Cin % 128 == 0 && p.Cout % 8 == 0 && p.Ho + p.Wo <= DMA_MAX_HW;
    if (!fast && !(fast_any && dma_ok)) {   // (the frame enumeration exists in the direct-to-LDS kernel only)
      munit_set_error("conv2d_wgrad: bf16 tensors need Cin %% 4 == 0, Cout %% 4 == 0 and tensors below 2 GiB");
      return MUNIT_ERR_ARG;
    }
    if (p.x_bf16 && p.dy_bf16 && p.Cin % 128 == 0 && p.Cout % 8 == 0 && p.Ho + p.Wo <= DMA_MAX_HW &&
        !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WGRAD_DMA")) {
      // direct-to-LDS bf16 kernel: 128 x 128 tiles (Cout padded), 64-pixel steps; never more splits than the plan's slabs hold
      // (pl.bc == 256: the plan was made for the 256 x 256 tile, see plan_launch)
      int pps = (cdiv(p.M, pl.nsplit) + BS_WP - 1) / BS_WP * BS_WP;
      p.pix_per_split = pps;
      const int ns = cdiv(p.M, pps);
      if (pl.bc == 256) {
        dim3 g2((unsigned)(p.Ktot / 256), (unsigned)(p.Cout / 256), (unsigned)ns);
        hipLaunchKernelGGL(conv_wgrad_bf16s_kernel<2>, g2, dim3(WTHR), 0, st, p);
      } else {
        dim3 g2((unsigned)cdiv(p.Ktot, 128), (unsigned)cdiv(p.Cout, 128), (unsigned)ns);
        hipLaunchKernelGGL(conv_wgrad_bf16s_kernel<1>, g2, dim3(WTHR), 0, st, p);
      }
      MUNIT_CHECK_LAUNCH("conv_wgrad_bf16s");
      const long long n = (long long)p.Cout * p.Ktot;
      const int blocks = (int)std::min<long long>((n + p.Cout + 255) / 256, 4096);
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.slab, dw, n, p.bias_slab, db, p.Cout,
                         ns, beta, beta_b);
      MUNIT_CHECK_LAUNCH("slab_reduce");
      return MUNIT_OK;
    }
    if (p.x_bf16 && p.dy_bf16) {        // trunk layers: bf16 MFMA (operands are already bf16 values)
      if (pl.bc == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, true, true, 1, true, true>), grid, dim3(WTHR), 0, st, p);
      else hipLaunchKernelGGL((conv_wgrad_kernel<128, true, true, 1, true, true>), grid, dim3(WTHR), 0, st, p);
    } else if (p.dy_bf16) {             // first layer: fp32 image (4-channel re-layout) against a bf16 dy, fp32 MFMA
      if (pl.bc == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, true, true, 0, false, true>), grid, dim3(WTHR), 0, st, p);
      else hipLaunchKernelGGL((conv_wgrad_kernel<128, true, true, 0, false, true>), grid, dim3(WTHR), 0, st, p);
    } else {
      munit_set_error("conv2d_wgrad: bf16 x with fp32 dy exists for the 3-channel image head only");
      return MUNIT_ERR_ARG;
    }
  } else if (fast && p.ct == 0 && pl.bc == 128 && p.Cin % 128 == 0 && p.Ho + p.Wo <= DMA_MAX_HW &&
      !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WGRAD_DMA")) {
    hipLaunchKernelGGL((conv_wgrad_kernel<128, true, true, 3>), grid, dim3(WTHR), 0, st, p);
  } else if (fast && p.ct == 1) {
    if (pl.bc == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, true, true, 1>), grid, dim3(WTHR), 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<128, true, true, 1>), grid, dim3(WTHR), 0, st, p);
  } else if (fast && p.ct == 2) {
    if (pl.bc == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, true, true, 2>), grid, dim3(WTHR), 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<128, true, true, 2>), grid, dim3(WTHR), 0, st, p);
  } else if (pl.bc == 64) {
    if (fast) hipLaunchKernelGGL((conv_wgrad_kernel<64, true, true>), grid, dim3(WTHR), 0, st, p);
    else if (aligned) hipLaunchKernelGGL((conv_wgrad_kernel<64, true, false>), grid, dim3(WTHR), 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<64, false, false>), grid, dim3(WTHR), 0, st, p);
  } else {
    if (fast) hipLaunchKernelGGL((conv_wgrad_kernel<128, true, true>), grid, dim3(WTHR), 0, st, p);
    else if (aligned) hipLaunchKernelGGL((conv_wgrad_kernel<128, true, false>), grid, dim3(WTHR), 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<128, false, false>), grid, dim3(WTHR), 0, st, p);
  }
  MUNIT_CHECK_LAUNCH("conv_wgrad");
  const long long n = (long long)p.Cout * p.Ktot;
  if (pl.nsplit >= 32 && n + p.Cout <= 64 * 2048) {
    hipLaunchKernelGGL(slab_reduce_wide_kernel, dim3(cdiv(n + p.Cout, 64)), dim3(1024), 0, st, p.slab, dw, n,
                       p.bias_slab, db, p.Cout, pl.nsplit, beta, beta_b);
    MUNIT_CHECK_LAUNCH("slab_reduce_wide");
    return MUNIT_OK;
  }
  const int blocks = (int)std::min<long long>((n + p.Cout + 255) / 256, 4096);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.slab, dw, n, p.bias_slab, db, p.Cout,
                     pl.nsplit, beta, beta_b);
  MUNIT_CHECK_LAUNCH("slab_reduce");
  return MUNIT_OK;
}

bool cin3_padded_ok(const munit_conv_desc* d) {
  return d->Cin == 3 && d->Cout % 4 == 0 && d->upsample == 0 && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_CIN3_PAD");
}
struct Cin3Plan {
  WgradPlan pl;
  size_t x4_bytes, dw4_bytes;
};
void plan_cin3(const munit_conv_desc* d, int Ho, int Wo, Cin3Plan* cp) {
  plan_launch(d->B * Ho * Wo, d->KH * d->KW * 4, d->Cout, true, &cp->pl);
  cp->x4_bytes = align_up((size_t)d->B * d->H * d->W * 4 * sizeof(float), 256);
  cp->dw4_bytes = align_up((size_t)d->Cout * d->KH * d->KW * 4 * sizeof(float), 256);
}

// bf16-storage layers whose backward-weight takes the 256 x 256 tile: both tensors bf16, whole 256-channel tiles on both sides
// (one filter tap per block), the direct-to-LDS conditions of run_wgrad, and enough pixels that 256 / tiles splits keep >= 16 steps
bool bf16s_big_tile(const munit_conv_desc* d, int Ho, int Wo) {
  if (MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WGRAD_BIG_TILE") || MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WGRAD_DMA")) return false;
  if (d->in_dtype != MUNIT_DTYPE_BF16 || d->out_dtype != MUNIT_DTYPE_BF16) return false;
  if (d->Cin % 256 != 0 || d->Cout % 256 != 0 || Ho + Wo > DMA_MAX_HW) return false;
  const long long tiles = (long long)(d->KH * d->KW * d->Cin / 256) * (d->Cout / 256);
  const long long M = (long long)d->B * Ho * Wo;
  return tiles <= 256 && M * tiles >= 256ll * 1024;
}

bool subpixel_wgrad_ok(const munit_conv_desc* d) {
  // bf16 tensors: only where the direct-to-LDS bf16 kernel applies (it alone enumerates the frame on bf16 data)
  const bool f32 = d->in_dtype == MUNIT_DTYPE_F32 && d->out_dtype == MUNIT_DTYPE_F32;
  const bool b16 = d->in_dtype == MUNIT_DTYPE_BF16 && d->out_dtype == MUNIT_DTYPE_BF16 && d->Cin % 128 == 0 &&
                   d->Cout % 8 == 0 && 2 * (d->H + d->W) <= DMA_MAX_HW && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WGRAD_DMA");
  return (f32 || b16) && d->upsample == 1 && d->KH == 5 && d->KW == 5 && d->pad == 2 && d->stride == 1 &&
         d->pad_mode == MUNIT_PAD_REFLECT && d->Cin % 4 == 0 && d->H >= 3 && d->W >= 3 &&
         !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_SUBPIXEL");
}

// sub-pixel wgrad of an up-sampling conv: workspace = [dwc: 4*Cout*9*Cin][slabs of the largest launch]
struct SubpixelPlan {
  WgradPlan phase, frame;
  size_t dwc_bytes, slab_bytes;
  bool wino;   // the four phase gradients through the Winograd kernel
  int ring;    // width of the output frame the generic 25-tap launch covers
};
void plan_subpixel(const munit_conv_desc* d, SubpixelPlan* sp) {
  const bool aligned = d->Cin % 4 == 0;
  plan_launch(d->B * (d->H - 2) * (d->W - 2), 9 * d->Cin, d->Cout, aligned, &sp->phase);
  const int Ho = 2 * d->H, Wo = 2 * d->W;
  sp->wino = d->compute == MUNIT_COMPUTE_F32 && d->in_dtype == MUNIT_DTYPE_F32 && d->out_dtype == MUNIT_DTYPE_F32 && d->H >= 4 &&
             d->W >= 4 && munit_wino_wgrad_ok(d->B, d->H, d->W, d->Cin, d->Cout) &&
             (long long)d->B * 4 * d->H * d->W * d->Cout < (1ll << 29);
  // Winograd phases run over ALL source pixels with a replicated edge and the outermost ring of dy masked (see
  // munit_conv2d_wgrad): the generic 25-tap launch then covers a ring one pixel wide instead of two
  sp->ring = sp->wino ? 1 : 2;
  plan_launch(d->B * (2 * sp->ring * Wo + 2 * sp->ring * (Ho - 2 * sp->ring)), 25 * d->Cin, d->Cout, aligned, &sp->frame);
  sp->dwc_bytes = align_up((size_t)4 * d->Cout * 9 * d->Cin * sizeof(float), 256);
  sp->slab_bytes = std::max(sp->phase.slab_bytes + sp->phase.bias_bytes, sp->frame.slab_bytes + sp->frame.bias_bytes);
  if (sp->wino)
    sp->slab_bytes = std::max(sp->slab_bytes, munit_wino_wgrad_workspace((long long)d->B * (d->H / 2) * (d->W / 2), d->Cin, d->Cout, 4));
}

// 3x3 / stride 1 / pad 1 fp32 layers with 64-multiples of channels: Winograd backward-weight (conv_wino.hip)
bool wino_wgrad_layer(const munit_conv_desc* d) {
  return d->compute == MUNIT_COMPUTE_F32 && d->in_dtype == MUNIT_DTYPE_F32 && d->out_dtype == MUNIT_DTYPE_F32 && d->KH == 3 &&
         d->KW == 3 && d->stride == 1 && d->pad == 1 && d->upsample == 0 && munit_wino_wgrad_ok(d->B, d->H, d->W, d->Cin, d->Cout);
}
// 4x4 / stride 2 / pad 1 fp32 layers: F(3x3, 2x2) backward-weight, one launch phase per filter-tap parity
bool wino_s2_wgrad_layer(const munit_conv_desc* d) {
  return d->compute == MUNIT_COMPUTE_F32 && d->in_dtype == MUNIT_DTYPE_F32 && d->out_dtype == MUNIT_DTYPE_F32 && d->KH == 4 &&
         d->KW == 4 && d->stride == 2 && d->pad == 1 && d->upsample == 0 && d->H >= 4 && d->W >= 4 &&
         munit_wino_wgrad_ok(d->B, d->H, d->W, d->Cin, d->Cout) && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WINOGRAD_S2") &&
         (long long)d->B * cdiv(d->H / 2, 3) * cdiv(d->W / 2, 3) >= (getenv("MUNIT_WINO_S2_MIN_BLOCKS") ? 1 : 512);
}
long long s2_tiles(const munit_conv_desc* d) { return (long long)d->B * cdiv(d->H / 2, 3) * cdiv(d->W / 2, 3); }
}  // namespace

extern "C" size_t munit_conv2d_wgrad_workspace_bytes(const munit_conv_desc* d) {
  int Ho, Wo;
  if (munit_conv2d_out_hw(d, &Ho, &Wo)) return 0;
  if (wino_s2_wgrad_layer(d)) return munit_wino_wgrad_workspace(s2_tiles(d), d->Cin, d->Cout, 4);
  if (wino_wgrad_layer(d)) return munit_wino_wgrad_workspace((long long)d->B * (d->H / 2) * (d->W / 2), d->Cin, d->Cout, 1);
  if (munit_small_wgrad_supported(d) && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_SMALL_WGRAD")) return munit_small_wgrad_workspace(d, Ho);
  if (subpixel_wgrad_ok(d)) {
    SubpixelPlan sp;
    plan_subpixel(d, &sp);
    return sp.dwc_bytes + sp.slab_bytes;
  }
  if (cin3_padded_ok(d)) {
    Cin3Plan cp;
    plan_cin3(d, Ho, Wo, &cp);
    return cp.x4_bytes + cp.dw4_bytes + cp.pl.slab_bytes + cp.pl.bias_bytes;
  }
  WgradPlan pl;
  plan_launch(d->B * Ho * Wo, d->KH * d->KW * d->Cin, d->Cout, d->Cin % 4 == 0, &pl, bf16s_big_tile(d, Ho, Wo));
  return pl.slab_bytes + pl.bias_bytes;
}

extern "C" double munit_conv2d_executed_flops(const munit_conv_desc* d, int pass) {
  if (pass == MUNIT_PASS_FWD || pass == MUNIT_PASS_DGRAD) return munit_igemm_executed_flops(d, pass);
  int Ho, Wo;
  if (pass != MUNIT_PASS_WGRAD || munit_conv2d_out_hw(d, &Ho, &Wo)) return 0.0;
  const double cc = 2.0 * d->Cin * d->Cout;
  if (wino_wgrad_layer(d)) return cc * d->B * (d->H / 2) * (d->W / 2) * 16;
  if (wino_s2_wgrad_layer(d)) return 4 * cc * (double)s2_tiles(d) * 16;
  if (munit_small_wgrad_supported(d) && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_SMALL_WGRAD")) return cc * d->B * Ho * Wo * d->KH * d->KW;
  if (subpixel_wgrad_ok(d)) {  // 4 phase gradients over the interior source pixels + the 25-tap frame
    SubpixelPlan sp;
    plan_subpixel(d, &sp);
    if (sp.wino) return cc * d->B * ((double)d->H * d->W * 4 * 4 + (2.0 * Wo + 2.0 * (Ho - 2)) * 25);
    return cc * d->B * ((double)(d->H - 2) * (d->W - 2) * 4 * 9 + (4.0 * Wo + 4.0 * (Ho - 4)) * 25);
  }
  if (cin3_padded_ok(d)) return 2.0 * 4 * d->Cout * d->B * Ho * Wo * d->KH * d->KW;   // zero 4th input channel
  return cc * d->B * Ho * Wo * d->KH * d->KW;
}

extern "C" const char* munit_conv2d_kernel_name(const munit_conv_desc* d, int pass) {
  if (pass == MUNIT_PASS_FWD || pass == MUNIT_PASS_DGRAD) return munit_igemm_kernel_name(d, pass);
  int Ho, Wo;
  if (pass != MUNIT_PASS_WGRAD || munit_conv2d_out_hw(d, &Ho, &Wo)) return "invalid";
  // (the second template argument mirrors the FAST decision of munit_wino_wgrad_launch)
  if (wino_s2_wgrad_layer(d)) {
    const bool fast = s2_tiles(d) % 8 == 0 && d->pad_mode == MUNIT_PAD_REFLECT && d->H % 6 == 0 && d->W % 6 == 0 && Ho % 3 == 0 && Wo % 3 == 0;
    const bool xclamp = !fast && s2_tiles(d) % 8 == 0 && d->pad_mode == MUNIT_PAD_REFLECT;   // mirrors munit_wino_wgrad_launch
    return fast ? "conv_wino_wgrad_kernel<true, true, false> + wino_wgrad_reduce_kernel"
                : xclamp ? "conv_wino_wgrad_kernel<true, false, true> + wino_wgrad_reduce_kernel"
                         : "conv_wino_wgrad_kernel<true, false, false> + wino_wgrad_reduce_kernel";
  }
  if (wino_wgrad_layer(d)) {
    const bool fast = ((long long)d->B * (d->H / 2) * (d->W / 2)) % 8 == 0 && d->pad_mode == MUNIT_PAD_REFLECT;
    return fast ? "conv_wino_wgrad_kernel<false, true, false> + wino_wgrad_reduce_kernel" : "conv_wino_wgrad_kernel<false, false, false> + wino_wgrad_reduce_kernel";
  }
  if (munit_small_wgrad_supported(d) && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_SMALL_WGRAD"))   // mirrors munit_small_wgrad's choice
    return MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WGRAD_PK") ? "conv_lanes_wgrad_kernel" : "conv_lanes_wgrad_pk_kernel";
  if (subpixel_wgrad_ok(d)) {
    SubpixelPlan sp;
    plan_subpixel(d, &sp);
    return sp.wino ? "conv_wino_wgrad_kernel<false, ., true> x4 sub-pixel phases + frame + reduce" : "conv_wgrad_kernel x4 sub-pixel phases + frame";
  }
  if (cin3_padded_ok(d)) return "conv_wgrad_kernel (3 input channels padded to 4)";
  return "conv_wgrad_kernel + slab_reduce_kernel";
}

extern "C" int munit_conv2d_wgrad(const munit_conv_desc* d, const void* x, const void* dy, float* dw,
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
  if (wino_s2_wgrad_layer(d)) {
    WinoWgradParams q{};
    q.x = reinterpret_cast<const float*>(x); q.dy = reinterpret_cast<const float*>(dy);
    q.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->Cin * 4); q.dy_bytes = (unsigned)((size_t)d->B * Ho * Wo * d->Cout * 4);
    q.dy_sw = d->Cout; q.dy_sh = (long long)Wo * d->Cout; q.dy_sb = (long long)Ho * Wo * d->Cout;
    q.B = d->B; q.H = d->H; q.W = d->W; q.Cin = d->Cin; q.Cout = d->Cout;
    q.reflect = d->pad_mode == MUNIT_PAD_REFLECT;
    q.th = cdiv(Ho, 3); q.tw = cdiv(Wo, 3); q.phases = 4;
    q.s2 = 1; q.Ho = Ho; q.Wo = Wo;
    return munit_wino_wgrad_launch(q, dw, 0, db, beta, beta, ws, st);
  }
  if (wino_wgrad_layer(d)) {
    WinoWgradParams q{};
    q.x = reinterpret_cast<const float*>(x); q.dy = reinterpret_cast<const float*>(dy);
    q.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->Cin * 4); q.dy_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->Cout * 4);
    q.dy_sw = d->Cout; q.dy_sh = (long long)d->W * d->Cout; q.dy_sb = (long long)d->H * d->W * d->Cout;
    q.B = d->B; q.H = d->H; q.W = d->W; q.Cin = d->Cin; q.Cout = d->Cout;
    q.reflect = d->pad_mode == MUNIT_PAD_REFLECT; q.xo = -1;
    q.th = d->H / 2; q.tw = d->W / 2; q.phases = 1;
    return munit_wino_wgrad_launch(q, dw, 0, db, beta, beta, ws, st);
  }
  if (munit_small_wgrad_supported(d) && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_SMALL_WGRAD"))
  {
    MUNIT_CHECK_ARG(d->out_dtype == MUNIT_DTYPE_F32, "conv2d_wgrad: the 3-channel image head has an fp32 dy");
    return munit_small_wgrad(d, Ho, Wo, x, reinterpret_cast<const float*>(dy), dw, db, beta, ws, st);
  }
  const bool aligned = d->Cin % 4 == 0;
  WgradParams p{};
  p.x = x; p.dy = dy;
  p.x_bf16 = d->in_dtype == MUNIT_DTYPE_BF16; p.dy_bf16 = d->out_dtype == MUNIT_DTYPE_BF16;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin;
  p.ups = d->upsample; p.Hu = d->H << p.ups; p.Wu = d->W << p.ups;
  p.Ho = Ho; p.Wo = Wo; p.Cout = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  p.reflect = d->pad_mode == MUNIT_PAD_REFLECT;
  p.ct = d->compute;
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
    f.frame = sp.ring;
    f.M = d->B * (2 * sp.ring * Wo + 2 * sp.ring * (Ho - 2 * sp.ring));
    rc = run_wgrad(f, sp.frame, aligned, dw, db, beta, beta, slabs, st);
    if (rc) return rc;
    if (sp.wino) {
      WinoWgradParams q{};
      q.x = reinterpret_cast<const float*>(x); q.dy = reinterpret_cast<const float*>(dy);
      q.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->Cin * 4); q.dy_bytes = (unsigned)((size_t)d->B * Ho * Wo * d->Cout * 4);
      q.dy_sw = 2 * d->Cout; q.dy_sh = (long long)2 * Wo * d->Cout; q.dy_sb = (long long)Ho * Wo * d->Cout;
      q.dy_off = 0; q.dy_prow = (long long)Wo * d->Cout; q.dy_pcol = d->Cout;
      q.B = d->B; q.H = d->H; q.W = d->W; q.Cin = d->Cin; q.Cout = d->Cout;
      // every source pixel i: output (2i + a, 2j + b) of phase (a, b) against taps i - 1 .. i + 1 of the source with its edge
      // REPLICATED -- exact for all outputs but the outermost ring (conv_igemm.hip, forward), whose dy the kernel reads as 0
      // and whose contribution the 25-tap frame launch above has added
      q.reflect = 2; q.xo = -1; q.ring_mask = 1;
      q.th = d->H / 2; q.tw = d->W / 2; q.phases = 4;
      rc = munit_wino_wgrad_launch(q, dwc, (long long)d->Cout * 9 * d->Cin, db, 0.0f, 1.0f, slabs, st);
      if (rc) return rc;
    }
    for (int ph = 0; ph < 4 && !sp.wino; ++ph) {
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
  if (cin3_padded_ok(d)) {
    Cin3Plan cp;
    plan_cin3(d, Ho, Wo, &cp);
    float* x4 = reinterpret_cast<float*>(ws);
    float* dw4 = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + cp.x4_bytes);
    void* slabs = reinterpret_cast<char*>(ws) + cp.x4_bytes + cp.dw4_bytes;
    const long long npix = (long long)d->B * d->H * d->W;
    hipLaunchKernelGGL(pad3to4_kernel, dim3((unsigned)std::min<long long>((npix + 255) / 256, 8192)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(x), reinterpret_cast<f32x4*>(x4), npix);
    MUNIT_CHECK_ARG(d->in_dtype == MUNIT_DTYPE_F32, "conv2d_wgrad: 3-channel inputs are fp32");
    MUNIT_CHECK_LAUNCH("pad3to4");
    WgradParams q = p;
    q.x = x4; q.Cin = 4; q.Ktot = d->KH * d->KW * 4; q.x_bf16 = 0;
    q.ct = 0;   // 4 channels per tap: not a multiple of the bf16 K granularity, stays fp32
    rc = run_wgrad(q, cp.pl, true, dw4, db, 0.0f, beta, slabs, st);
    if (rc) return rc;
    const long long n3 = (long long)d->Cout * d->KH * d->KW * 3;
    hipLaunchKernelGGL(compact4to3_kernel, dim3((unsigned)std::min<long long>((n3 + 255) / 256, 4096)), dim3(256), 0, st, dw4, dw,
                       n3, beta);
    MUNIT_CHECK_LAUNCH("compact4to3");
    return MUNIT_OK;
  }
  WgradPlan pl;
  plan_launch(p.M, p.Ktot, d->Cout, aligned, &pl, bf16s_big_tile(d, Ho, Wo));
  return run_wgrad(p, pl, aligned, dw, db, beta, beta, ws, st);
}

#ifdef WGB_STAMP
extern "C" int munit_debug_wgrad_bf16_stamps(long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wgb_stamps), sizeof(long long) * 8 * 8 * 4) == hipSuccess ? 0 : -1;
}
#endif
