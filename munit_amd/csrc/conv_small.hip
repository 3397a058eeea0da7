// Convolutions with a tiny channel count on one side (the 64->3 image head of the decoder,
// networks.py:548-559, its weight gradient, and the data gradient of the 3->64 first encoder
// layers, networks.py:446-450/484-488).  With N = 3 an MFMA tile would be >90 % padding and the
// im2col A-operand (49x the input) would be streamed for 6 useful FLOP per 4 bytes, so these run on
// the vector ALUs (same fp32 peak as the f32 MFMA on gfx950):
//   * forward (conv_patch_fwd_kernel): one thread per output pixel, a (4+K-1) x (64+K-1) halo patch of
//     16 input channels staged in LDS as float4 planes (adjacent lanes = adjacent pixels => conflict-free
//     ds_read_b128), the weights are wave-uniform so they arrive through scalar loads and every
//     ds_read_b128 feeds 4*CO v_fma with an SGPR operand; no cross-lane reduction at all;
//   * backward-weight (conv_lanes_wgrad_kernel): channel-per-lane; wave (b, rows, kh) owns
//     dw[:, kh, :, ci] (CO*K accumulators), slides a 1 x K window of its input row in registers
//     (one coalesced 256-byte load per pixel), dy values are fetched one pixel per lane and broadcast
//     with v_readlane; partial results go to slabs summed in a fixed order (deterministic).
#include "common.h"

namespace {

struct SmallParams {
  const void* x;      // [B][H][W][Cin], fp32 or bf16 (template parameter XT of the kernels)
  const float* w;     // [CO][K][K][Cin]
  const float* bias;  // [CO] or null
  const float* dy;    // wgrad: [B][Ho][Wo][CO]
  float* y;           // fwd:  [B][Ho][Wo][CO]
  float* slab;        // wgrad: [rows][CO*K*K*Cin + CO]
  int B, H, W, Cin, Ho, Wo;
  int pad, reflect, act;
  float slope;
  int tiles_x, tiles_y;  // fwd
  int rows_per_unit, units_per_img;  // wgrad
};

__device__ inline int map_coord(int v, int n, int reflect) {
  if (v < 0) return reflect ? -v : -1;
  if (v >= n) return reflect ? 2 * n - 2 - v : -1;
  return v;
}

// ---------------------------------------------------------------------------------------------
// forward: thread per pixel, LDS halo patch, scalar weights.  Cin % 16 == 0.
// ---------------------------------------------------------------------------------------------
template <int CO, int K, typename XT>
__global__ __launch_bounds__(256) void conv_patch_fwd_kernel(SmallParams p) {
  constexpr int TW = 64, TH = 4, PW = TW + K - 1, PH = TH + K - 1, PLANE = PH * PW;
  __shared__ f32x4 patch[4 * PLANE];  // [c4][row][col]
  const int tid = threadIdx.x;
  const int tx = tid & 63, ty = tid >> 6;
  int bid = blockIdx.x;
  const int tile_x = bid % p.tiles_x; bid /= p.tiles_x;
  const int tile_y = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int ow0 = tile_x * TW, oh0 = tile_y * TH;
  const XT* xb = reinterpret_cast<const XT*>(p.x) + (long long)b * p.H * p.W * p.Cin;
  const float* __restrict__ wg = p.w;

  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;

  for (int ch0 = 0; ch0 < p.Cin; ch0 += 16) {
    __syncthreads();
    for (int e = tid; e < 4 * PLANE; e += 256) {
      const int c4 = e & 3;
      const int pix = e >> 2;
      const int pr = pix / PW, pc = pix - pr * PW;
      const int ih = map_coord(oh0 + pr - p.pad, p.H, p.reflect);
      const int iw = map_coord(ow0 + pc - p.pad, p.W, p.reflect);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ih >= 0 && iw >= 0) v = ld4(xb + ((long long)ih * p.W + iw) * p.Cin + ch0 + c4 * 4);
      patch[c4 * PLANE + pr * PW + pc] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int kh = 0; kh < K; ++kh) {
#pragma unroll 1
      for (int c4 = 0; c4 < 4; ++c4) {   // one c4 per trip keeps the K*CO*4 scalar weights within the SGPR file
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const f32x4 xv = patch[c4 * PLANE + (ty + kh) * PW + tx + kw];
#pragma unroll
          for (int c = 0; c < CO; ++c) {
            // wave-uniform address -> scalar load
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wg + ((long long)(c * K + kh) * K + kw) * p.Cin + ch0 + c4 * 4);
            acc[c] = fmaf(wv[0], xv[0], acc[c]);
            acc[c] = fmaf(wv[1], xv[1], acc[c]);
            acc[c] = fmaf(wv[2], xv[2], acc[c]);
            acc[c] = fmaf(wv[3], xv[3], acc[c]);
          }
        }
      }
    }
  }
  const int oh = oh0 + ty, ow = ow0 + tx;
  if (oh < p.Ho && ow < p.Wo) {
    float* yo = p.y + (((long long)b * p.Ho + oh) * p.Wo + ow) * CO;
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      float v = acc[c] + (p.bias != nullptr ? p.bias[c] : 0.f);
      yo[c] = apply_act(v, p.act, p.slope);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward on the matrix pipe (round 2): v_mfma_f32_4x4x1_16B_f32.  A 16x16 MFMA tile would be 13/16 padding with three
// output channels; the 16-block 4x4x1 form wastes only one row in four: block b (lanes 4b..4b+3) computes the outer
// product D_b[i][j] += A_b[i] * B_b[j] with i = output channel (row 3 is a zero weight) and j = one of 4 adjacent
// pixels, and the sixteen blocks take sixteen different k: input channels 4b..4b+3 of the current filter tap, delivered
// by ONE ds_read_b128 per lane and tap (element t feeds MFMA t).  Lane map verified on the hardware with exact integer
// data (tools/ubench/mfma4x4.hip): A row i = l%4, B column j = l%4, D[i][j] in register i of lane 4*block + j.  The
// sixteen partial tiles are summed across blocks with four shuffles at the very end.  Same rate as every fp32 MFMA
// (64 FLOP/clk/SIMD), exact fp32 FMAs like the VALU kernel it replaces (which reaches 0.15 of that rate).
// Block: 4 waves, output tile 8 rows x 16 pixels (a wave owns rows ty and ty + 4 = 8 pixel groups), 64 input channels
// per pass, halo patch of 14 x 22 pixels in LDS (77 KiB: two blocks per CU; the halo makes the kernel read 2.4x its
// input, which is what bounds it -- a 4-row tile read 3.4x and ran no faster on a 3x shorter MFMA stream), 256 bytes per pixel with the 16-byte chunk c of pixel q stored at position
// c ^ 4*(q&3) (the four pixels of a group then hit four different bank quads for every block: conflict-free b128 reads).
// fp32 input: the patch is filled by global_load_lds_dwordx4 (one wave instruction = 4 pixels, every lane fetches the
// chunk that belongs at its landing spot; padding taps read a zero page): all 14 fills of a wave are in flight at once.
// The thread-per-chunk register loop it replaces waited for every load in turn (~19 us per tile, 6x the MFMA time).
// bf16 input is widened in registers, with the loads of a thread issued together.
// ---------------------------------------------------------------------------------------------
__device__ const float munit_head_zero16[4] = {0.f, 0.f, 0.f, 0.f};

template <typename XT>
__global__ __launch_bounds__(256, 2) void conv_head_mfma_kernel(SmallParams p) {
  constexpr int K = 7, CO = 3, TW = 16, TH = 8, PW = TW + K - 1, PH = TH + K - 1, PS = 64;
  constexpr int NG_W = 2 * (TW / 4);                  // pixel groups per wave: rows ty and ty + 4, four groups each
  constexpr int NPIX = PH * PW;                       // 220 = 55 groups of 4 pixels
  static_assert(NPIX % 4 == 0, "patch fills go four pixels at a time");
  __shared__ __attribute__((aligned(16))) float patch[NPIX * PS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int ty = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int blk = lane >> 2, j = lane & 3;
  int bid = blockIdx.x;
  const int tile_x = bid % p.tiles_x; bid /= p.tiles_x;
  const int tile_y = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int ow0 = tile_x * TW, oh0 = tile_y * TH;
  const XT* xb = reinterpret_cast<const XT*>(p.x) + (long long)b * p.H * p.W * p.Cin;
  const float* __restrict__ wg = p.w;

  f32x4 acc[NG_W];
#pragma unroll
  for (int g = 0; g < NG_W; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wi = j < CO ? j : 0;             // weight row of this lane (row 3 multiplies by zero)
  const float wmask = j < CO ? 1.f : 0.f;

  for (int ch0 = 0; ch0 < p.Cin; ch0 += 64) {
    __syncthreads();
    {
      // wave w fills pixel groups w, w+4, ...: lane l -> pixel 4*grp + (l>>4), landing position l&15, chunk (l&15) ^ 4*(l>>4)
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
      const int sub = lane >> 4, c4 = (lane & 15) ^ (4 * sub);
      constexpr int NG = NPIX / 4, PER = (NG + 3) / 4;
      if constexpr (sizeof(XT) == 4) {
#pragma unroll
        for (int it = 0; it < PER; ++it) {
          const int grp = wv + 4 * it;
          if (grp < NG) {
            const int pix = 4 * grp + sub;
            const int pr = pix / PW, pc = pix - pr * PW;
            const int ih = map_coord(oh0 + pr - p.pad, p.H, p.reflect);
            const int iw = map_coord(ow0 + pc - p.pad, p.W, p.reflect);
            const void* g = (ih >= 0 && iw >= 0) ? (const void*)(xb + ((long long)ih * p.W + iw) * p.Cin + ch0 + c4 * 4)
                                                 : (const void*)munit_head_zero16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(patch + grp * 4 * PS), 16, 0, 0);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        f32x4 stage[PER];
#pragma unroll
        for (int it = 0; it < PER; ++it) {
          const int grp = wv + 4 * it;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (grp < NG) {
            const int pix = 4 * grp + sub;
            const int pr = pix / PW, pc = pix - pr * PW;
            const int ih = map_coord(oh0 + pr - p.pad, p.H, p.reflect);
            const int iw = map_coord(ow0 + pc - p.pad, p.W, p.reflect);
            if (ih >= 0 && iw >= 0) v = ld4(xb + ((long long)ih * p.W + iw) * p.Cin + ch0 + c4 * 4);
          }
          stage[it] = v;
        }
#pragma unroll
        for (int it = 0; it < PER; ++it) {
          const int grp = wv + 4 * it;
          if (grp < NG) *reinterpret_cast<f32x4*>(&patch[grp * 4 * PS + lane * 4]) = stage[it];
        }
      }
    }
    __syncthreads();
    const float* wl = wg + (long long)wi * K * K * p.Cin + ch0 + 4 * blk;   // + tap * Cin
    // weights: one filter row (7 taps) per trip, the next row in flight meanwhile -- a single-tap prefetch distance
    // (128 cycles of MFMAs) does not cover an L2 hit, and made every tap wait for its weights
    f32x4 wrow[K], wnext[K];
#pragma unroll
    for (int kw = 0; kw < K; ++kw) wrow[kw] = *reinterpret_cast<const f32x4*>(wl + (long long)kw * p.Cin) * wmask;
#pragma unroll 1
    for (int kh = 0; kh < K; ++kh) {
      const int khn = kh + 1 < K ? kh + 1 : kh;
#pragma unroll
      for (int kw = 0; kw < K; ++kw)
        wnext[kw] = *reinterpret_cast<const f32x4*>(wl + (long long)(khn * K + kw) * p.Cin) * wmask;
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        f32x4 xf[NG_W];
#pragma unroll
        for (int g = 0; g < NG_W; ++g) {
          const int q = (ty + 4 * (g >> 2) + kh) * PW + 4 * (g & 3) + j + kw;   // patch pixel; chunk c at position c ^ 4*(q&3)
          xf[g] = *reinterpret_cast<const f32x4*>(&patch[q * PS + 4 * (blk ^ (4 * (q & 3)))]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int g = 0; g < NG_W; ++g)
            acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wrow[kw][t], xf[g][t], acc[g], 0, 0, 0);
      }
#pragma unroll
      for (int kw = 0; kw < K; ++kw) wrow[kw] = wnext[kw];
    }
  }
  // sum the sixteen blocks (lane bits 2..5); lanes 0..3 then hold pixel j of every group
#pragma unroll
  for (int g = 0; g < NG_W; ++g) {
    const int oh = oh0 + ty + 4 * (g >> 2);
    float v[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      float s = acc[g][c];
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 8, 64);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      v[c] = s;
    }
    const int ow = ow0 + 4 * (g & 3) + j;
    if (blk == 0 && oh < p.Ho && ow < p.Wo) {
      float* yo = p.y + (((long long)b * p.Ho + oh) * p.Wo + ow) * CO;
#pragma unroll
      for (int c = 0; c < CO; ++c) yo[c] = apply_act(v[c] + (p.bias != nullptr ? p.bias[c] : 0.f), p.act, p.slope);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward on the vector ALUs with packed FMAs (round 2, replaces the 4x4x1 MFMA form as the default): v_pk_fma_f32 issues
// two exact fp32 FMAs per lane and cycle -- the same 64 FLOP/clk/SIMD as the fp32 matrix pipe -- and takes a wave-uniform
// weight pair straight from SGPRs, so three output channels cost no padding at all.  The MFMA form above is bound by its
// operand traffic (one ds_read_b128 per lane for four 8-cycle MFMAs = the full LDS rate of the CU); here one lane owns
// two adjacent output pixels and every ds_read_b128 (4 channels of one patch pixel) feeds up to 2 pixels x 3 channels x 4 =
// 24 FMAs, and the 8 reads of one patch row segment serve all 7 horizontal taps.
//   per tap and channel c:        (o0, o1)[A] += xA[c] * (w0, w1)[c]     v_pk_fma_f32, xA[c] broadcast by op_sel; same for B
//   per tap and channel pair:     o2[A] (even, odd partial) += (xA[c], xA[c+1]) * (w2[c], w2[c+1])      same for B
//   = 12 packed FMAs for the 24 multiply-accumulates of a tap and channel quad: nothing is padded, nothing is moved.
// Block = 4 waves, tile 16 rows x 32 pixels (wave = 4 rows, lane = pixel pair), 8 input channels per pass; the halo patch
// of a pass (22 x 38 pixels x 32 B = 26 KiB: four to five blocks per CU cover each other's fills) is laid out
// [row][channel quad][column parity][column / 2][4] so that the 16 lanes of a row read 256 contiguous bytes (conflict-free)
// for every tap; fp32 input fills it with global_load_lds_dwordx4.  The weights are re-laid once per call into
// [pass][kh][quad][kw][12] (38 KiB in the workspace): (w0 w1) pairs of the quad's four channels, then their four w2.
// ---------------------------------------------------------------------------------------------
constexpr int HP_TW = 32, HP_TH = 16, HP_K = 7, HP_PW = HP_TW + HP_K - 1, HP_PH = HP_TH + HP_K - 1, HP_CP = 8;
constexpr int HP_IDX = 20;                                   // (HP_PW + 1) / 2 = 19 column pairs, padded to 20
constexpr int HP_ROW = (HP_CP / 4) * 2 * HP_IDX * 4;         // floats per patch row: [quad][parity][idx][4] = 320 (1280 B = 5 bank rows)
constexpr int HP_ITEMS = HP_PH * (HP_CP / 4) * 2 * HP_IDX;   // 16-byte items of the patch: 1760

// wp[((pass * 7 + kh) * 2 + quad) * 7 + kw][12] = (w0 w1)[c0] (w0 w1)[c1] (w0 w1)[c2] (w0 w1)[c3] | w2[c0] w2[c1] w2[c2] w2[c3]
// for the four input channels c = pass * 8 + quad * 4 + 0..3 of tap (kh, kw); w is [3][7][7][Cin]
__global__ void head_pk_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin) {
  const int total = (Cin / HP_CP) * HP_K * 2 * HP_K * 12;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int r = i;
    const int e = r % 12; r /= 12;
    const int kw = r % HP_K; r /= HP_K;
    const int quad = r & 1; r >>= 1;
    const int kh = r % HP_K;
    const int pass = r / HP_K;
    const int co = e < 8 ? (e & 1) : 2, c = e < 8 ? (e >> 1) : e - 8;
    wp[i] = w[((long long)(co * HP_K + kh) * HP_K + kw) * Cin + pass * HP_CP + quad * 4 + c];
  }
}

template <typename XT>
__global__ __launch_bounds__(256, 4) void conv_head_pk_kernel(SmallParams p, const float* __restrict__ wp) {
  __shared__ __attribute__((aligned(16))) float patch[HP_PH * HP_ROW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, lr = 4 * wv + (lane >> 4);        // pixel pair (columns 2j, 2j+1) of tile row lr
  int bid = blockIdx.x;
  const int tile_x = bid % p.tiles_x; bid /= p.tiles_x;
  const int tile_y = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int ow0 = tile_x * HP_TW, oh0 = tile_y * HP_TH;
  const XT* xb = reinterpret_cast<const XT*>(p.x) + (long long)b * p.H * p.W * p.Cin;

  // fill items of this thread: item n -> (row, quad, parity, idx); patch column = 2 * idx + parity
  constexpr int PER = (HP_ITEMS + 255) / 256;   // 7
  int foff[PER];                                // element offset of the item's 4 channels at pass 0, or -1 (zero)
#pragma unroll
  for (int it = 0; it < PER; ++it) {
    const int n = (wv + 4 * it) * 64 + lane;
    int r = n;
    const int idx = r % HP_IDX; r /= HP_IDX;
    const int par = r & 1; r >>= 1;
    const int quad = r & 1;
    const int row = r >> 1;
    const int pc = 2 * idx + par;
    const int ih = map_coord(oh0 + row - p.pad, p.H, p.reflect);
    const int iw = map_coord(ow0 + pc - p.pad, p.W, p.reflect);
    foff[it] = (n < HP_ITEMS && pc < HP_PW && ih >= 0 && iw >= 0) ? (ih * p.W + iw) * p.Cin + quad * 4 : -1;
  }

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 a01 = {0.f, 0.f}, b01 = {0.f, 0.f};   // output channels 0, 1 of pixel A / B
  f32x2 a2 = {0.f, 0.f}, b2 = {0.f, 0.f};     // output channel 2: partial sums over the even / odd input channels
  const float* rd = patch + lr * HP_ROW + j * 4;   // + (kh * HP_ROW) + ((quad * 2 + (i & 1)) * HP_IDX + (i >> 1)) * 4
  const int npass = p.Cin / HP_CP;
  for (int pass = 0; pass < npass; ++pass) {
    __syncthreads();
    if constexpr (sizeof(XT) == 4) {
#pragma unroll
      for (int it = 0; it < PER; ++it) {
        const int n0 = (wv + 4 * it) * 64;
        if (n0 < HP_ITEMS) {   // wave-uniform; the last wave-load runs past the patch by 32 items: clipped by the lane test below
          const void* g = foff[it] >= 0 ? (const void*)(xb + foff[it] + pass * HP_CP) : (const void*)munit_head_zero16;
          if (n0 + 64 <= HP_ITEMS || n0 + lane < HP_ITEMS)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(patch + n0 * 4), 16, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      f32x4 stage[PER];
#pragma unroll
      for (int it = 0; it < PER; ++it) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (foff[it] >= 0) v = ld4(xb + foff[it] + pass * HP_CP);
        stage[it] = v;
      }
#pragma unroll
      for (int it = 0; it < PER; ++it) {
        const int n = (wv + 4 * it) * 64 + lane;
        if (n < HP_ITEMS) *reinterpret_cast<f32x4*>(&patch[n * 4]) = stage[it];
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int kh = 0; kh < HP_K; ++kh) {
#pragma unroll
      for (int quad = 0; quad < 2; ++quad) {
        f32x4 xs[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          xs[i] = *reinterpret_cast<const f32x4*>(rd + kh * HP_ROW + ((quad * 2 + (i & 1)) * HP_IDX + (i >> 1)) * 4);
        const float* wq = wp + (((pass * HP_K + kh) * 2 + quad) * HP_K) * 12;   // wave-uniform: scalar loads
#pragma unroll
        for (int kw = 0; kw < HP_K; ++kw) {
          const f32x4 xa = xs[kw], xb2 = xs[kw + 1];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x2 w01 = {wq[kw * 12 + 2 * c], wq[kw * 12 + 2 * c + 1]};
            a01 = __builtin_elementwise_fma(f32x2{xa[c], xa[c]}, w01, a01);
            b01 = __builtin_elementwise_fma(f32x2{xb2[c], xb2[c]}, w01, b01);
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 w2 = {wq[kw * 12 + 8 + 2 * h], wq[kw * 12 + 9 + 2 * h]};
            a2 = __builtin_elementwise_fma(f32x2{xa[2 * h], xa[2 * h + 1]}, w2, a2);
            b2 = __builtin_elementwise_fma(f32x2{xb2[2 * h], xb2[2 * h + 1]}, w2, b2);
          }
        }
      }
    }
  }
  const int oh = oh0 + lr, ow = ow0 + 2 * j;
  if (oh < p.Ho) {
    const float bias0 = p.bias != nullptr ? p.bias[0] : 0.f, bias1 = p.bias != nullptr ? p.bias[1] : 0.f,
                bias2 = p.bias != nullptr ? p.bias[2] : 0.f;
    float* yo = p.y + (((long long)b * p.Ho + oh) * p.Wo + ow) * 3;
    if (ow < p.Wo) {
      yo[0] = apply_act(a01[0] + bias0, p.act, p.slope);
      yo[1] = apply_act(a01[1] + bias1, p.act, p.slope);
      yo[2] = apply_act((a2[0] + a2[1]) + bias2, p.act, p.slope);
    }
    if (ow + 1 < p.Wo) {
      yo[3] = apply_act(b01[0] + bias0, p.act, p.slope);
      yo[4] = apply_act(b01[1] + bias1, p.act, p.slope);
      yo[5] = apply_act((b2[0] + b2[1]) + bias2, p.act, p.slope);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward-weight: channel per lane; unit = (b, band of rows_per_unit output rows, kh, channel group)
// ---------------------------------------------------------------------------------------------
template <int CO, int K, typename XT>
__global__ __launch_bounds__(256) void conv_lanes_wgrad_kernel(SmallParams p) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = p.Cin >> 6;
  const long long unit = (long long)blockIdx.x * 4 + wv;
  const long long per_band = (long long)K * G;
  const long long nunits = (long long)p.B * p.units_per_img * per_band;
  if (unit >= nunits) return;  // no block-level synchronisation below
  const int g = (int)(unit % G);
  const int kh = (int)((unit / G) % K);
  const long long band = unit / per_band;          // (b, band index)
  const int bi = (int)(band % p.units_per_img);
  const int b = (int)(band / p.units_per_img);
  const int ci = g * 64 + lane;

  float acc[CO][K];
#pragma unroll
  for (int c = 0; c < CO; ++c)
#pragma unroll
    for (int kw = 0; kw < K; ++kw) acc[c][kw] = 0.f;
  float bsum[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) bsum[c] = 0.f;

  constexpr int CH = (64 / K) * K;  // pixels per dy fetch: a multiple of K keeps the rotation phase
  const int oh_begin = bi * p.rows_per_unit, oh_end = min(p.Ho, oh_begin + p.rows_per_unit);
  for (int oh = oh_begin; oh < oh_end; ++oh) {
    const int ih = map_coord(oh - p.pad + kh, p.H, p.reflect);
    const bool rowok = ih >= 0;
    const XT* rowp = reinterpret_cast<const XT*>(p.x) + ((long long)b * p.H + (rowok ? ih : 0)) * p.W * p.Cin + ci;
    const float* dyrow = p.dy + ((long long)b * p.Ho + oh) * p.Wo * CO;
    float win[K];
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
      const int iw = map_coord(j - p.pad, p.W, p.reflect);
      win[j] = (rowok && iw >= 0) ? ld1(rowp + (long long)iw * p.Cin) : 0.f;
    }
    for (int ow0 = 0; ow0 < p.Wo; ow0 += CH) {
      const int n = min(CH, p.Wo - ow0);
      float dyv[CO];
#pragma unroll
      for (int c = 0; c < CO; ++c) dyv[c] = (lane < n) ? dyrow[(long long)(ow0 + lane) * CO + c] : 0.f;
#pragma unroll 1
      for (int i0 = 0; i0 < n; i0 += K) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
          const int i = i0 + s;  // pixels past n carry dy = 0 (lanes >= n loaded 0) and clamped loads
          const int iw = map_coord(min(ow0 + i, p.Wo - 1) + (K - 1) - p.pad, p.W, p.reflect);
          win[(K - 1 + s) % K] = (rowok && iw >= 0) ? ld1(rowp + (long long)iw * p.Cin) : 0.f;
#pragma unroll
          for (int c = 0; c < CO; ++c) {
            const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dyv[c]), i));
            bsum[c] += d;
#pragma unroll
            for (int kw = 0; kw < K; ++kw) acc[c][kw] = fmaf(d, win[(kw + s) % K], acc[c][kw]);
          }
        }
      }
    }
  }
  // slab row per (b, band): [CO][K][K][Cin] + CO bias sums; this wave owns [:, kh, :, 64g..64g+63]
  const long long n_w = (long long)CO * K * K * p.Cin;
  float* out = p.slab + band * (n_w + CO);
#pragma unroll
  for (int c = 0; c < CO; ++c)
#pragma unroll
    for (int kw = 0; kw < K; ++kw) out[((long long)(c * K + kh) * K + kw) * p.Cin + ci] = acc[c][kw];
  if (g == 0 && kh == 0 && lane < CO) {
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < CO; ++c) v = (lane == c) ? bsum[c] : v;
    out[n_w + lane] = v;
  }
}

// dw = beta*dw + sum_r slab[r][0 : n_w]; db = beta*db + sum_r slab[r][n_w : n_w + CO]
__global__ void small_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                         float* __restrict__ db, long long n_w, int co, long long rows,
                                         float beta) {
  const long long stride = n_w + co;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < stride;
       i += (long long)gridDim.x * blockDim.x) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long long r = 0;
    for (; r + 3 < rows; r += 4) {
      s0 += slab[r * stride + i];
      s1 += slab[(r + 1) * stride + i];
      s2 += slab[(r + 2) * stride + i];
      s3 += slab[(r + 3) * stride + i];
    }
    for (; r < rows; ++r) s0 += slab[r * stride + i];
    const float s = (s0 + s1) + (s2 + s3);
    if (i < n_w) dw[i] = (beta != 0.f ? beta * dw[i] : 0.f) + s;
    else if (db != nullptr) db[i - n_w] = (beta != 0.f ? beta * db[i - n_w] : 0.f) + s;
  }
}

}  // namespace

// ---- entry points used by the conv dispatch in conv_igemm.hip / conv_wgrad.hip (not part of the C ABI) ----
// Round 3: the same backward-weight with PACKED fp32 FMAs.  A wave walks TWO output rows at once -- lanes 0-31 row oh, lanes
// 32-63 row oh + 1 -- and a lane owns a channel PAIR, so every multiply-accumulate of the 3 x 7 window products is one
// v_pk_fma_f32 over the pair: 21 vector instructions per two pixels where the scalar form issues 42 (+ 6 v_readlane), the
// dy value of a lane's own row arrives by ds_bpermute (LDS pipe) instead of readlane + select.  The two half-waves hold partial
// sums of the same 64 channels; they meet through one cross-half shuffle per accumulator at the end.  Same slabs, same
// reduction kernel, same summation over rows inside a band up to the pairing of rows.
namespace {
template <int CO, int K, typename XT>
__global__ __launch_bounds__(256) void conv_lanes_wgrad_pk_kernel(SmallParams p) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = p.Cin >> 6;
  const long long unit = (long long)blockIdx.x * 4 + wv;
  const long long per_band = (long long)K * G;
  const long long nunits = (long long)p.B * p.units_per_img * per_band;
  if (unit >= nunits) return;  // no block-level synchronisation below
  const int g = (int)(unit % G);
  const int kh = (int)((unit / G) % K);
  const long long band = unit / per_band;          // (b, band index)
  const int bi = (int)(band % p.units_per_img);
  const int b = (int)(band / p.units_per_img);
  const int rh = lane >> 5, cp = lane & 31;         // row of the pair, channel pair
  const int ci = g * 64 + 2 * cp;
  const bool want_b = g == 0 && kh == 0;

  f32x2 acc[CO][K];
#pragma unroll
  for (int c = 0; c < CO; ++c)
#pragma unroll
    for (int kw = 0; kw < K; ++kw) acc[c][kw] = f32x2{0.f, 0.f};
  float bsum[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) bsum[c] = 0.f;

  auto ldx = [&](const XT* q) -> f32x2 {
    if constexpr (sizeof(XT) == 4) return *reinterpret_cast<const f32x2*>(q);
    else { const bf16_t* h = reinterpret_cast<const bf16_t*>(q); return f32x2{(float)h[0], (float)h[1]}; }
  };
  constexpr int CH = (32 / K) * K;  // pixels per dy fetch (32 lanes per row): a multiple of K keeps the rotation phase
  const int oh_begin = bi * p.rows_per_unit, oh_end = min(p.Ho, oh_begin + p.rows_per_unit);
  for (int oh2 = oh_begin; oh2 < oh_end; oh2 += 2) {
    const int oh = oh2 + rh;
    const bool live = oh < oh_end;                 // an odd row count leaves the second half-wave idle in the last pair
    const int ih = map_coord(min(oh, p.Ho - 1) - p.pad + kh, p.H, p.reflect);
    const bool rowok = live && ih >= 0;
    const XT* rowp = reinterpret_cast<const XT*>(p.x) + ((long long)b * p.H + (rowok ? ih : 0)) * p.W * p.Cin + ci;
    const float* dyrow = p.dy + ((long long)b * p.Ho + min(oh, p.Ho - 1)) * p.Wo * CO;
    f32x2 win[K];
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
      const int iw = map_coord(j - p.pad, p.W, p.reflect);
      win[j] = (rowok && iw >= 0) ? ldx(rowp + (long long)iw * p.Cin) : f32x2{0.f, 0.f};
    }
    for (int ow0 = 0; ow0 < p.Wo; ow0 += CH) {
      const int n = min(CH, p.Wo - ow0);
      float dyv[CO];   // lane (rh, j) holds pixel ow0 + j of its row
#pragma unroll
      for (int c = 0; c < CO; ++c) dyv[c] = (live && cp < n) ? dyrow[(long long)(ow0 + cp) * CO + c] : 0.f;
#pragma unroll 1
      for (int i0 = 0; i0 < n; i0 += K) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
          const int i = i0 + s;  // pixels past n carry dy = 0 (lanes >= n loaded 0) and clamped loads
          const int iw = map_coord(min(ow0 + i, p.Wo - 1) + (K - 1) - p.pad, p.W, p.reflect);
          win[(K - 1 + s) % K] = (rowok && iw >= 0) ? ldx(rowp + (long long)iw * p.Cin) : f32x2{0.f, 0.f};
          const int src = ((lane & 32) + min(i, 31)) << 2;   // byte index of the lane holding pixel i of this lane's row
#pragma unroll
          for (int c = 0; c < CO; ++c) {
            const float d = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, dyv[c])));
            if (want_b) bsum[c] += d;
            const f32x2 dd = {d, d};
#pragma unroll
            for (int kw = 0; kw < K; ++kw) acc[c][kw] = __builtin_elementwise_fma(dd, win[(kw + s) % K], acc[c][kw]);
          }
        }
      }
    }
  }
  // the two half-waves hold partial sums of the same channel pairs: add across (lane ^ 32); lanes 0-31 store
  const long long n_w = (long long)CO * K * K * p.Cin;
  float* out = p.slab + band * (n_w + CO);
#pragma unroll
  for (int c = 0; c < CO; ++c)
#pragma unroll
    for (int kw = 0; kw < K; ++kw) {
      f32x2 v = acc[c][kw];
      v[0] += __shfl_xor(v[0], 32, 64);
      v[1] += __shfl_xor(v[1], 32, 64);
      if (rh == 0) *reinterpret_cast<f32x2*>(out + ((long long)(c * K + kh) * K + kw) * p.Cin + ci) = v;
    }
  if (want_b) {
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      const float t = bsum[c] + __shfl_xor(bsum[c], 32, 64);   // every lane of a half saw the same dy sequence
      v = (lane == c) ? t : v;
    }
    if (lane < CO) out[n_w + lane] = v;
  }
}
}  // namespace

bool munit_small_fwd_supported(const munit_conv_desc* d) {
  return d->Cout == 3 && d->KH == 7 && d->KW == 7 && d->stride == 1 && d->upsample == 0 && d->Cin % 16 == 0;
}

bool munit_small_wgrad_supported(const munit_conv_desc* d) {
  return d->Cout == 3 && d->KH == 7 && d->KW == 7 && d->stride == 1 && d->upsample == 0 && d->Cin % 64 == 0;
}

size_t munit_small_fwd_workspace(const munit_conv_desc* d) {
  // packed-FMA head kernel: the re-laid weights [Cin/8][7][2][7][4][4]
  if (d->Cin % 8 != 0 || MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_HEAD_PK")) return 0;
  return align_up((size_t)(d->Cin / HP_CP) * HP_K * 2 * HP_K * 12 * sizeof(float), 256);
}

int munit_small_fwd(const munit_conv_desc* d, int Ho, int Wo, const void* x, const float* w, const float* bias,
                    float* y, void* ws, hipStream_t st) {
  SmallParams p{};
  p.x = x; p.w = w; p.bias = bias; p.y = y;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = Ho; p.Wo = Wo;
  p.pad = d->pad; p.reflect = d->pad_mode == MUNIT_PAD_REFLECT; p.act = d->act; p.slope = d->slope;
  if (munit_small_fwd_workspace(d) != 0 && ws != nullptr) {
    float* wp = reinterpret_cast<float*>(ws);
    const int total = (d->Cin / HP_CP) * HP_K * 2 * HP_K * 12;
    hipLaunchKernelGGL(head_pk_weights_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, wp, d->Cin);
    MUNIT_CHECK_LAUNCH("head_pk_weights");
    p.tiles_x = cdiv(Wo, HP_TW); p.tiles_y = cdiv(Ho, HP_TH);
    const long long nb = (long long)d->B * p.tiles_x * p.tiles_y;
    if (d->in_dtype == MUNIT_DTYPE_BF16) hipLaunchKernelGGL((conv_head_pk_kernel<bf16_t>), dim3((unsigned)nb), dim3(256), 0, st, p, wp);
    else hipLaunchKernelGGL((conv_head_pk_kernel<float>), dim3((unsigned)nb), dim3(256), 0, st, p, wp);
    MUNIT_CHECK_LAUNCH("conv_head_pk");
    return MUNIT_OK;
  }
  if (d->Cin % 64 == 0 && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_HEAD_MFMA")) {
    p.tiles_x = cdiv(Wo, 16); p.tiles_y = cdiv(Ho, 8);
    const long long nb = (long long)d->B * p.tiles_x * p.tiles_y;
    if (d->in_dtype == MUNIT_DTYPE_BF16) hipLaunchKernelGGL((conv_head_mfma_kernel<bf16_t>), dim3((unsigned)nb), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_head_mfma_kernel<float>), dim3((unsigned)nb), dim3(256), 0, st, p);
    MUNIT_CHECK_LAUNCH("conv_head_mfma");
    return MUNIT_OK;
  }
  p.tiles_x = cdiv(Wo, 64); p.tiles_y = cdiv(Ho, 4);
  const long long blocks = (long long)d->B * p.tiles_x * p.tiles_y;
  if (d->in_dtype == MUNIT_DTYPE_BF16) hipLaunchKernelGGL((conv_patch_fwd_kernel<3, 7, bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((conv_patch_fwd_kernel<3, 7, float>), dim3((unsigned)blocks), dim3(256), 0, st, p);
  MUNIT_CHECK_LAUNCH("conv_patch_fwd");
  return MUNIT_OK;
}

namespace {
constexpr int WG_ROWS = 4;  // output rows per wgrad unit
}

size_t munit_small_wgrad_workspace(const munit_conv_desc* d, int Ho) {
  const long long bands = (long long)d->B * cdiv(Ho, WG_ROWS);
  return align_up((size_t)bands * ((size_t)3 * 49 * d->Cin + 3) * sizeof(float), 256);
}

int munit_small_wgrad(const munit_conv_desc* d, int Ho, int Wo, const void* x, const float* dy, float* dw,
                      float* db, float beta, void* ws, hipStream_t st) {
  SmallParams p{};
  p.x = x; p.dy = dy; p.slab = reinterpret_cast<float*>(ws);
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = Ho; p.Wo = Wo;
  p.pad = d->pad; p.reflect = d->pad_mode == MUNIT_PAD_REFLECT;
  p.rows_per_unit = WG_ROWS; p.units_per_img = cdiv(Ho, WG_ROWS);
  const int G = d->Cin / 64;
  const long long bands = (long long)d->B * p.units_per_img;
  const long long units = bands * 7 * G;
  const bool pk = !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WGRAD_PK");
  if (pk) {
    if (d->in_dtype == MUNIT_DTYPE_BF16) hipLaunchKernelGGL((conv_lanes_wgrad_pk_kernel<3, 7, bf16_t>), dim3((unsigned)cdiv(units, 4)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_lanes_wgrad_pk_kernel<3, 7, float>), dim3((unsigned)cdiv(units, 4)), dim3(256), 0, st, p);
  } else {
    if (d->in_dtype == MUNIT_DTYPE_BF16) hipLaunchKernelGGL((conv_lanes_wgrad_kernel<3, 7, bf16_t>), dim3((unsigned)cdiv(units, 4)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_lanes_wgrad_kernel<3, 7, float>), dim3((unsigned)cdiv(units, 4)), dim3(256), 0, st, p);
  }
  MUNIT_CHECK_LAUNCH("conv_lanes_wgrad");
  const long long n_w = (long long)3 * 49 * d->Cin;
  const int blocks = cdiv(n_w + 3, 256);
  hipLaunchKernelGGL(small_slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.slab, dw, db, n_w, 3, bands, beta);
  MUNIT_CHECK_LAUNCH("small_slab_reduce");
  return MUNIT_OK;
}
