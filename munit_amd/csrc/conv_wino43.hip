// Winograd F(4x4, 3x3) forward convolution for gfx950 on v_mfma_f32_16x16x4_f32 (round 3): the 3x3 / stride 1 / pad 1 layers
// of the residual trunk (scripts/networks.py:603-624 -> nn.Conv2d) and the four 3x3 phase convolutions of the sub-pixel
// up-sampling layers.
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A      per 4x4 output tile, 6x6 input patch d, 3x3 filter g   (points 0, +-1, +-2, inf)
//
// spends 36 multiply-accumulates per 16 outputs and channel pair where F(2x2, 3x3) spends 64 and the direct form 144.  Why it
// pays HERE: on this chip nothing a SIMD's other wave issues is hidden behind the fp32 matrix instruction
// (tools/ubench/mfma_coissue.hip), and the whole step is bound by the matrix work it issues (bench.py: the step time follows
// the sum of the kernels' times) -- so the only large lever left is fewer multiplies.  The price is arithmetic: the transforms
// multiply by 4, 5, 8 and 1/24, and the result is ~8e-6 normalised max error against fp64 on the trunk shape where F(2x2, 3x3)
// has 7e-7 and the direct fp32 form 1.7e-6 -- inside the 2e-5 the op tests assert and far inside the stated tolerance
// (forward 1e-4).  cuDNN, which the reference runs these layers on, picks among the same family.
//
// One block = 4x4 tiles (16x16 output pixels) of one image x 64 output channels x all 36 frequencies, 8 waves:
//   * a 16-tile block is exactly ONE 16-row MFMA tile per frequency, so a frequency costs 4 (channel tiles) x 2 MFMAs per 8
//     input channels;
//   * K runs in chunks of 32 channels (one barrier each), four sub-steps of 8.  Waves 0-3 are the loaders: wave w transforms
//     the 8 channels 8w..8w+7 of the chunk -- a lane = (tile, channel pair) reads its 6x6 patch with 36 buffer_load_dwordx2
//     (reflect padding = mirrored offsets, zero padding = an out-of-range offset), applies B^T . B with packed FMAs (144) and
//     writes 36 pairs to V[f][sub-step w][tile][k pair];
//   * frequencies: the loader wave of a SIMD owns 3, its multiply-only partner (wave w + 4) 6 -- nine per SIMD, and the
//     loader's smaller share pays for its transform work;
//   * the transformed weights U[f][n][k] go from the prepared image (fragment order, L2) straight to registers, two
//     global_load_dwordx4 per frequency and sub-step, requested one sub-step ahead;
//   * V is double buffered (2 x 72 KiB);
//   * epilogue: the 36 planes meet in LDS, one thread per (tile, channel) folds them into the 4x4 pixels (A^T . A), adds bias,
//     applies the activation and stores 256-byte row segments.
#include "wino.h"
#include <algorithm>
#include <type_traits>

namespace {

constexpr int F43 = 36;
constexpr int T43 = 16;                          // tiles per block (4 x 4)
constexpr int KC43 = 32;                         // channels per chunk
constexpr int VSUB43 = F43 * T43 * 8 + 8;        // floats per sub-step slab [f][tile 16][8 k] + 8: the pad keeps the four
                                                 // sub-steps' reads of a frequency from being a multiple of 512 B apart, or the
                                                 // compiler fuses pairs into ds_read2st64_b64, which banks mod 32 in 16-lane groups
constexpr int VBUF43 = 4 * VSUB43;               // floats per V buffer: [sub-step 4][f][tile][8 k]  (72 KiB)
constexpr int MLD43 = 68;                        // row stride of the epilogue planes M[f][tile][64 channels]
constexpr int SMEM43 = F43 * T43 * MLD43;        // 39 168 floats = 156 672 B  (>= 2 * VBUF43 = 147 456 B)
static_assert(SMEM43 >= 2 * VBUF43, "operand buffers must fit");

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ inline float act43(float v, int act, float slope) {
  return act == MUNIT_ACT_NONE ? v : (v > 0.f ? v : (act == MUNIT_ACT_RELU ? 0.f : v * slope));
}
__device__ inline f32x2 fma2(float c, f32x2 a, f32x2 b) { return __builtin_elementwise_fma(f32x2{c, c}, a, b); }

// y = B^T x for the 6-vector x (stride st in the array), in place.  B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0;
// 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
template <int ST>
__device__ inline void bt6(f32x2* x) {
  const f32x2 x0 = x[0], x1 = x[ST], x2 = x[2 * ST], x3 = x[3 * ST], x4 = x[4 * ST], x5 = x[5 * ST];
  const f32x2 a = fma2(-4.f, x2, x4), b = fma2(-4.f, x1, x3), c = x4 - x2, d = x3 - x1;
  x[0] = fma2(4.f, x0, fma2(-5.f, x2, x4));
  x[ST] = a + b;
  x[2 * ST] = a - b;
  x[3 * ST] = fma2(2.f, d, c);
  x[4 * ST] = fma2(-2.f, d, c);
  x[5 * ST] = fma2(4.f, x1, fma2(-5.f, x3, x5));
}
// z = A^T m for the 6-vector m: A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ inline void at6(const float (&m)[6], float (&z)[4]) {
  const float s = m[1] + m[2], d = m[1] - m[2], u = m[3] + m[4], v = m[3] - m[4];
  z[0] = (m[0] + s) + u;
  z[1] = __builtin_fmaf(2.f, v, d);
  z[2] = __builtin_fmaf(4.f, u, s);
  z[3] = __builtin_fmaf(8.f, v, d) + m[5];
}

// MODE 0: reflect padding.  1: zero padding.
template <int MODE>
__global__ __launch_bounds__(512) void conv_wino43_kernel(WinoParams p) {
  constexpr bool REFLECT = MODE == 0;
  __shared__ __attribute__((aligned(16))) float smem[SMEM43];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware order (blocks b and b + 8 share an XCD): every XCD gets one contiguous run of blocks, so the N-blocks of a tile
  // block and neighbouring tile blocks (shared halo) meet in one L2, and the blocks of an XCD walk the weight image together
  int blk = blockIdx.x;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blk & 7, idx = blk >> 3;
    blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int n_blk = blk % p.NB;
  int m_blk = blk / p.NB;
  const int btx = m_blk % p.btw; m_blk /= p.btw;
  const int bty = m_blk % p.bth;
  const int b = m_blk / p.bth;
  const int phase = blockIdx.y;

  const bool loader = wave < 4;
  const int wq = wave & 3;                       // SIMD pair index: frequencies 9 wq .. 9 wq + 8
  const int f0 = 9 * wq + (loader ? 0 : 3);      // its first frequency

  // ---- loader: lane = (tile tl, channel pair cp) of sub-step `wave` of every chunk ----
  const int tl = lane >> 2, cp = lane & 3;
  const int gy = min(bty * 4 + (tl >> 2), p.th - 1), gx = min(btx * 4 + (tl & 3), p.tw - 1);   // clamped (stores are predicated)
  // byte offsets of the 6 patch rows and columns; anything outside the image that is not a reflection gets an offset that
  // keeps the sum beyond the buffer (x_bytes < 2^30 is a precondition), which the load returns as 0
  unsigned roff[6], coff[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    int ih = 4 * gy - 1 + i, iw = 4 * gx - 1 + i;
    if (REFLECT) {
      ih = ih < 0 ? -ih : (ih >= p.H ? 2 * p.H - 2 - ih : ih);
      iw = iw < 0 ? -iw : (iw >= p.W ? 2 * p.W - 2 - iw : iw);
    }
    roff[i] = (unsigned)ih < (unsigned)p.H ? (unsigned)((b * p.H + ih) * p.W) * (unsigned)p.xc * 4u : 0x80000000u;
    coff[i] = (unsigned)iw < (unsigned)p.W ? (unsigned)(iw * p.xc + 8 * wave + 2 * cp) * 4u : 0x40000000u;
  }
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  f32x2 d[36];
  auto load_raw = [&](int c) {
#ifdef ABL43_NO_RAW     // timing-only ablation: the patch is loaded once
    if (c > 1) return;
#endif
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j)
        d[i * 6 + j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xres, roff[i] + coff[j], c * (KC43 * 4), 0));
  };
  // V position of this loader lane in plane f of sub-step slab `wave`: [tile][8 k] with the k-pair slot swizzled by the tile's bit 3
  // (the half-wave fragment reads below then cover all 64 banks)
  const int vpos = wave * VSUB43 + tl * 8 + ((cp ^ (((tl >> 3) & 1) << 1)) << 1);
  auto transform_store = [&](int buf) {
#ifndef ABL43_NO_XFORM   // timing-only ablation: no transform arithmetic
#pragma unroll
    for (int j = 0; j < 6; ++j) bt6<6>(d + j);        // columns: B^T d
#pragma unroll
    for (int i = 0; i < 6; ++i) bt6<1>(d + i * 6);    // rows: (.) B
#endif
    float* V = smem + buf * VBUF43 + vpos;
#pragma unroll
    for (int f = 0; f < F43; ++f) *reinterpret_cast<f32x2*>(V + f * (T43 * 8)) = d[f];
  };

  // ---- the two roles as two instantiations of one body: NF = frequencies of the wave (3 loader / 6 multiply-only), so that
  // each role's registers are allocated for what it really keeps live (the loader: 12 accumulator quads + the 6x6 patch; the
  // multiplier: 24 accumulator quads + two sets of 12 weight fragments)
  const int nc = p.K / KC43;
  auto body = [&](auto nf_tag, auto ld_tag) {
    constexpr int NF = decltype(nf_tag)::value;
    constexpr bool LD = decltype(ld_tag)::value;
    // U: this wave's frequencies straight from the prepared image into registers, per sub-step of 8 channels: image
    // [c8][N block][f][half][lane][4]; lane (n, kq) holds (k pair kq of channel n) of channel tiles 2 half, 2 half + 1
    const f32x4* const ug = reinterpret_cast<const f32x4*>(p.u + phase * p.u_phase) + ((long long)n_blk * F43 + f0) * 128 + lane;
    const long long u_c8 = (long long)p.NB * F43 * 128;   // f32x4 per 8-channel step
    f32x4 u0[NF][2], u1[NF][2];
    auto load_u = [&](int c8, f32x4 (&u)[NF][2]) {
#ifdef ABL43_NO_U      // timing-only ablation: the weight fragments are loaded once
      if (c8 > 1) return;
#endif
#ifdef ABL43_U_SAME    // timing-only ablation: every sub-step reads the same 2 slices of the image (cache hits), same instruction count
      c8 &= 1;
#endif
      const f32x4* g = ug + c8 * u_c8;
#pragma unroll
      for (int q = 0; q < NF; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) u[q][h] = g[q * 128 + h * 64];
    };
    f32x4 acc[NF][4];
#pragma unroll
    for (int q = 0; q < NF; ++q)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment of tile r = lane & 15, k pair kq = lane >> 4 of plane f0 + q: byte position inside a V buffer, one opaque
    // register per frequency (formed once; the compiler cannot fuse reads a constant apart into ds_read2st64_b64)
    int fpos[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      fpos[q] = ((f0 + q) * (T43 * 8) + (lane & 15) * 8 + (((lane >> 4) ^ (((lane >> 3) & 1) << 1)) << 1)) * 4;
      asm("" : "+v"(fpos[q]));
    }
    auto compute = [&](const int (&fb)[NF], auto s_tag, const f32x4 (&u)[NF][2]) {
      constexpr int S = decltype(s_tag)::value;
      const char* Vb = reinterpret_cast<const char*>(smem) + S * (VSUB43 * 4);
      f32x2 a[NF];
#pragma unroll
      for (int q = 0; q < NF; ++q) a[q] = *reinterpret_cast<const f32x2*>(Vb + fb[q]);
#pragma unroll
      for (int q = 0; q < NF; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][t], u[q][nt >> 1][(nt & 1) * 2 + t], acc[q][nt], 0, 0, 0);
    };

    if constexpr (LD) {
      load_raw(0);
      load_u(0, u0);
      transform_store(0);
      if (nc > 1) load_raw(1);
    } else {
      load_u(0, u0);
    }
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
      const int cur = c & 1;
      if constexpr (LD) {
#ifndef ABL43_NO_LOADER   // timing-only ablation: the loader waves only multiply
        if (c + 1 < nc) {
          transform_store(cur ^ 1);
          if (c + 2 < nc) load_raw(c + 2);
        }
#endif
      }
      int fb[NF];
#pragma unroll
      for (int q = 0; q < NF; ++q) fb[q] = fpos[q] + cur * (VBUF43 * 4);
      // four sub-steps; U of the next sub-step (or of the next chunk's first) is requested before the current one multiplies
      load_u(4 * c + 1, u1);
      compute(fb, std::integral_constant<int, 0>{}, u0);
      load_u(4 * c + 2, u0);
      compute(fb, std::integral_constant<int, 1>{}, u1);
      load_u(4 * c + 3, u1);
      compute(fb, std::integral_constant<int, 2>{}, u0);
      if (c + 1 < nc) load_u(4 * c + 4, u0);
      compute(fb, std::integral_constant<int, 3>{}, u1);
#ifndef ABL43_NO_BARRIER  // timing-only ablation
      __syncthreads();
#endif
    }
    // epilogue, first half: this wave's planes M[f][tile][64 channels] into LDS
#pragma unroll
    for (int q = 0; q < NF; ++q)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)   // C/D map: col = lane & 15 (channel), row = 4 * (lane >> 4) + r (tile)
          smem[((f0 + q) * T43 + 4 * (lane >> 4) + r) * MLD43 + nt * 16 + (lane & 15)] = acc[q][nt][r];
  };
  if (loader) body(std::integral_constant<int, 3>{}, std::true_type{});
  else body(std::integral_constant<int, 6>{}, std::false_type{});

  // ---- epilogue: every thread folds the 36 planes of two (tile, channel) pairs into 4x4 pixels ----
#ifdef ABL43_NO_EPI       // timing-only ablation: no output transform / stores
  if (p.K > 0) return;
#endif
  __syncthreads();
  const int co = tid & 63;
  const int n = n_blk * 64 + co;
  const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
  const float slope = p.slope;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int t2 = (tid >> 6) * 2 + it;
    const int ty = bty * 4 + (t2 >> 2), tx = btx * 4 + (t2 & 3);
    float z[6][4];   // rows of A^T M: z[j][i] = (A^T m_col_j)[i]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float m[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) m[i] = smem[((i * 6 + j) * T43 + t2) * MLD43 + co];
      at6(m, z[j]);
    }
    if (ty < p.th && tx < p.tw) {
      const long long yo = (phase >> 1) * p.y_prow + (phase & 1) * p.y_pcol + b * p.y_sb + (long long)(4 * ty) * p.y_sh +
                           (long long)(4 * tx) * p.y_sw + n;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float row[6] = {z[0][i], z[1][i], z[2][i], z[3][i], z[4][i], z[5][i]};
        float o[4];
        at6(row, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = o[j] + bv;
          const long long off = yo + i * p.y_sh + j * p.y_sw;
          if (p.add != nullptr) v += p.add[off];
          p.y[off] = act43(v, p.act, slope);
        }
      }
    }
  }
}

}  // namespace

// OPT-IN (environment MUNIT_WINOGRAD43=1, read once per process).  Measured in round 3 on the trunk layer: 148 us against 176 us
// for F(2x2, 3x3) forward (-16 %; bench step 106.3 vs 107.9 ms), at 8e-6 against 7e-7 normalised max error; with it the
// step-level gradient parity sits at 1.7e-5 .. 3.5e-5 (bound 5e-5) instead of 4e-6 and the full-size adjoint test needs a looser
// bound.  Parity is the first gate of this build, so the default stays the exact-to-1e-6 F(2x2, 3x3) path.
bool munit_wino43_ok(int B, int H, int W, int K, int N) {
  if (MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WINOGRAD") || !MUNIT_ENV_FLAG("MUNIT_WINOGRAD43")) return false;
  return K % KC43 == 0 && N % 64 == 0 && H % 4 == 0 && W % 4 == 0 && H >= 4 && W >= 4 &&
         (long long)B * H * W * K < (1ll << 28) && (long long)B * H * W * N < (1ll << 40);
}

int munit_wino43_launch(const WinoParams& p, hipStream_t st) {
  const long long blocks = (long long)p.B * p.bth * p.btw * p.NB;
  MUNIT_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "conv_wino43: bad grid");
  MUNIT_CHECK_ARG(p.x_bytes < (1u << 30), "conv_wino43: input tensor too large for the padding offsets");
  const dim3 grid((unsigned)blocks, (unsigned)std::max(1, p.phases));
  if (p.mode == 0) hipLaunchKernelGGL(conv_wino43_kernel<0>, grid, dim3(512), 0, st, p);
  else hipLaunchKernelGGL(conv_wino43_kernel<1>, grid, dim3(512), 0, st, p);
  MUNIT_CHECK_LAUNCH("conv_wino43");
  return MUNIT_OK;
}
