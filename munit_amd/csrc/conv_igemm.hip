// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32
// (bf16 / f32x3 modes: v_mfma_f32_16x16x32_bf16).  One kernel serves
//   * forward conv (reflect/zero pad, stride 1/2, optional fused nearest x2 upsample of
//     the input, fused bias + activation)            -- networks.py:695-701, 532-546
//   * backward-data: stride-1 layers with the pad/upsample adjoint folded into the gather (ROLE 2), strided
//     layers as one launch "phase" per (row,col) residue (a transposed conv) + fold_kernel (ROLE 1).
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin with k = (kh*KW+kw)*Cin+ci.
// NHWC makes every K-tile of 32 channels one contiguous 128-byte run per output pixel.
// Tile: 128 (M) x BN (N) x 32 (K), 512 threads = 8 waves (4 x 2 at BN=128), each wave 32 channels x 64 or 32
// pixels as 16x16 MFMA tiles.  fp32 single-gather variants fill the LDS tiles directly from global memory
// (global_load_lds_dwordx4, XOR-swizzled unpadded [row][32] tiles); the folded backward-data and the bf16 modes
// stage through registers into [row][36] (pad 4 => conflict-free ds_read_b128: one b128 feeds 4 MFMA k-steps).
// Tiles are double-buffered with one barrier per K-tile.  DESIGN.md section 3.1.
#include "common.h"
#include "wino.h"
#include <cstdlib>

namespace {

struct IgemmParams {
  const float* x;     // bf16 storage (CT == 4): the bf16 tensor seen as a float tensor with Cin/2 channels
  const float* w;     // likewise the bf16 weight image
  const float* bias;
  void* y;            // float or bf16 (out_bf16) elements; all y strides / offsets below are in ELEMENTS
  int B, H, W, Cin;   // source tensor
  int Hu, Wu, ups;    // upsampled extent (H << ups)
  int Ho, Wo, Cout;   // output grid / GEMM N
  int KH, KW, stride, pad, reflect;
  int Ktot;           // KH*KW*Cin
  long long w_row;    // floats between two output-channel rows of w
  long long y_sb;     // output strides (floats)
  long long y_sh;
  int y_sw;
  int M;              // B*Ho*Wo
  int act;
  float slope;
  // phase launches (backward-data of strided convs): blockIdx.y = pa*ps + pb
  int ps;                  // phases per axis (1 for everything else)
  long long w_phase;       // floats between the weights of two phases
  long long y_phase_row;   // output offset per pa
  int y_phase_col;         // output offset per pb
  int n_tiles;             // tiles along N
  // ROLE 2 (backward-data with the pad/upsample adjoint folded into the gather): geometry of the
  // forward conv whose input gradient is being formed
  int f_pad, f_ups, f_reflect, f_Hu, f_Wu;
  // frame mode (sub-pixel up-sampling conv): GEMM rows enumerate only the border frame of the Ho x Wo output, `frame`
  // pixels wide (2: rows 0,1,Ho-2,Ho-1 in full, then columns 0,1,Wo-2,Wo-1 of the remaining rows; 1: the outermost ring)
  int frame;
  // split-K (small grids): blockIdx.z = split, K-tiles [z*kt_per_split, ...); raw partial tiles go to
  // slab[(phase*ksplit + z)][M][Cout] and splitk_epilogue_kernel sums them in order (+bias, act)
  int ksplit, kt_per_split;
  float* slab;
  int ct;    // compute type (aligned variants only): 0 fp32 MFMA, 1 bf16 operands, 2 f32x3 split
  int patch; // ROLE 2: at most two padded positions per axis fold onto a pixel -> direct-to-LDS tiles + LDS patch
  int out_bf16;   // the epilogue rounds to bf16 (bias + activation applied in fp32 first)
  int bf16s;      // x and w are bf16 in HBM (see x): direct-to-LDS tiles of 64 bf16 per row, v_mfma_f32_16x16x32_bf16
  int cin4;       // x is the 4-channel re-layout of a 3-channel tensor, w the matching padded image (CT == 5)
};

// sum of packed bf16 octets (the LDS patch of the bf16-storage backward-data): fp32 adds, one rounding
__device__ inline f32x4 bf16x8_add4(f32x4 l, f32x4 a, f32x4 b, f32x4 c) {
  const bf16v8 L = __builtin_bit_cast(bf16v8, l), A = __builtin_bit_cast(bf16v8, a), B = __builtin_bit_cast(bf16v8, b),
               C = __builtin_bit_cast(bf16v8, c);
  bf16v8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (bf16_t)((float)L[e] + (((float)A[e] + (float)B[e]) + (float)C[e]));
  return __builtin_bit_cast(f32x4, r);
}

// row index -> (sample, output row, output column)
__device__ inline void decode_pixel(int m, int Ho, int Wo, int frame, int& b, int& oh, int& ow) {
  if (!frame) {
    const int HoWo = Ho * Wo;
    b = m / HoWo;
    const int rem = m - b * HoWo;
    oh = rem / Wo;
    ow = rem - oh * Wo;
  } else {
    const int F = frame, F2 = 2 * frame;   // ring width (2, or 1 when the phase convs replicate the source edge)
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

constexpr int BM = 128;
constexpr int BK = 32;
#ifndef IGEMM_WAVES
#define IGEMM_WAVES 8
#endif
constexpr int NWAVES = IGEMM_WAVES;   // 8: wave tile 64x32 (BN=128), four waves per SIMD; 4: wave tile 64x64, two per SIMD
constexpr int NTHR = 64 * NWAVES;
constexpr int LDS_LD = BK + 4;

// Padded/up-sampled coordinates whose gradient folds onto source coordinate i (adjoint of
// nearest x2 upsample followed by reflect/zero padding): up to 4, packed 16 bits each, 0xFFFF = none.
__device__ inline uint2 fold_cands(int i, int Hu, int ups, int P, int reflect) {
  unsigned long long pk = ~0ull;  // four 16-bit slots, filled from the low end (no private array)
  int n = 0;
  auto push = [&](int v) {
    if (n < 4) {
      const int sh = 16 * n;
      pk = (pk & ~(0xFFFFull << sh)) | ((unsigned long long)(unsigned)v << sh);
      ++n;
    }
  };
  const int nu = 1 << ups;
  for (int du = 0; du < nu; ++du) {
    const int hu = (i << ups) + du;
    push(hu + P);
    if (reflect) {
      if (hu >= 1 && hu <= P) push(P - hu);
      if (hu >= Hu - 1 - P && hu <= Hu - 2) push(2 * Hu - 2 - hu + P);
    }
  }
  return make_uint2((unsigned)(pk & 0xFFFFFFFFull), (unsigned)(pk >> 32));
}
__device__ inline int cand_at(uint2 v, int a) {
  const unsigned w = (a & 2) ? v.y : v.x;
  return (int)((w >> ((a & 1) * 16)) & 0xFFFFu);
}

// ROLE: 0 = forward; 1 = backward-data as plain correlation over dy (phase launches for strided
// convs; the pad/upsample adjoint is applied afterwards by fold_kernel); 2 = backward-data of
// stride-1 convs with that adjoint folded into the A-operand gather (GEMM rows = source pixels:
// the gathered dy values of every padded/up-sampled position that maps to the pixel are summed
// before the MFMA -- exact, because the GEMM is linear in A -- which removes the padded-domain
// buffer and, for the up-sampling convs, 4x of the MFMA work).
// BF16 = operands rounded to bf16 (RNE) when they are staged into LDS and multiplied with
// v_mfma_f32_16x16x32_bf16 (fp32 accumulate; tensors stay fp32 in HBM): the "bf16 compute" mode of
// BASELINE.json config #3.  Aligned variants only.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 128 bytes of zeros in device memory: the direct-to-LDS loads cannot substitute a value, so a tap that falls into
// zero padding (or a row past M) is pointed here instead
__device__ const float munit_zero_page[32] = {0.f};

// CT (compute type): 0 = fp32 MFMA; 1 = bf16 operands; 2 = "f32x3": every fp32 operand is split exactly into
// three bf16 planes a = a0 + a1 + a2 (a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)) and the six
// products a_i * b_j with i + j <= 2 are accumulated in fp32.  The dropped terms are <= 2^-24 |a||b|; measured
// against fp64 the result is as accurate as the fp32 FMA chain (truncation 6e-9 vs 4e-7 accumulation error,
// K = 2304).  Six bf16 MFMAs cost 6/16 of one fp32 MFMA step.  Three planes are single-buffered in LDS (two
// barriers per K-tile) to keep two blocks per CU.
template <int BN, bool ALIGNED, int ROLE, int CT = 0>
__global__ __launch_bounds__(NTHR, NWAVES == 8 ? ((CT == 3 && BN == 64) ? 6 : 4) : 2) void conv_igemm_kernel(IgemmParams p) {
  // CT == 3: fp32 MFMA as CT == 0, but the tiles travel global -> LDS directly (global_load_lds_dwordx4, no VGPR
  // staging, no ds_write); taps in zero padding read munit_zero_page.  Not for ROLE 2, which adds gathers.  Tiles are unpadded [row][32] floats (each wave instruction fills 8 rows
  // = 1 KiB); bank conflicts are avoided by a swizzle instead: the lane that lands at 16-byte position p of row r
  // fetches global chunk p ^ ((r >> 1) & 7), and the fragment reads undo it.
  // CT == 4: bf16 STORAGE.  Activations and the weight image are bf16 in HBM; a 128-byte tile row is then 64 channels
  // instead of 32 floats, which is the only thing that changes for the direct-to-LDS loader (the host hands it the
  // tensors as float tensors with Cin/2 channels), the fragment ds_read_b128 now holds the 8 consecutive k of one
  // v_mfma_f32_16x16x32_bf16 operand, and one MFMA replaces four.
  // CT == 5: fp32 direct-to-LDS tiles for layers with THREE input channels (the 7x7 first layers, the 4x4 first layers of
  // the discriminators, the backward-data of the image head).  The host re-lays the image with a zero 4th channel, so
  // a 128-byte tile row is 8 filter taps x 4 channels and every lane's 16 bytes are ONE pixel of ONE tap: the lane
  // derives its tap from its chunk index and fetches that pixel -- no scalar per-element gather (the ALIGNED == false
  // kernel needed 12 loads and ~200 VALU per row and K-tile for the same bytes) -- against a weight image padded the
  // same way ([Cout][8 taps x 4] per K-tile, zero beyond the last tap).
  constexpr bool BF16 = CT == 1 || CT == 2;
  constexpr bool DMA = CT == 3 || CT == 4 || CT == 5;
  constexpr bool BF16S = CT == 4;
  constexpr bool CIN4 = CT == 5;
  static_assert(!CIN4 || ROLE != 2, "4-channel taps: single-gather roles only");
  static_assert(!DMA || (ALIGNED && NWAVES == 8), "direct-to-LDS loads: aligned variants only");
  // ROLE 2 with CT == 3 ("fold by LDS patch"): layers whose pad adjoint folds at most two padded positions per axis onto
  // a source pixel (no up-sampling; reflect pad 1 of the 3x3 resblock convs).  The primary position of every row is a
  // direct-to-LDS load like any forward tile; the up to three further combinations (column partner, row partner, both)
  // are fetched into registers next to it and added to the wave's OWN landed rows after s_waitcnt vmcnt(0), before the
  // barrier that publishes the tile.  Exact (the GEMM is linear in A); only waves that hold a border pixel take the
  // branch, and no load depends on another (the register-path variant pays a dependent round trip per K-tile in the
  // blocks that hold a corner pixel).
  constexpr bool PATCH = DMA && ROLE == 2;
  constexpr int NPL = CT == 2 ? 3 : 1;     // bf16 planes per operand
  static_assert(!BF16 || ALIGNED, "bf16 operands need Cin % 32 == 0");
  // 8 waves per block: four waves per SIMD with two blocks per CU keep the matrix pipe fed while other
  // waves gather (measured MfmaUtil 73 % with 4 waves / 210 registers -> see profiles/).
  constexpr int WN = NWAVES == 8 ? 32 : 64;   // output channels per wave
  constexpr int WAVES_N = BN / WN;
  constexpr int WAVES_M = NWAVES / WAVES_N;
  constexpr int WM = BM / WAVES_M;            // rows per wave: 64 or 32
  // v_mfma_f32_16x16x4_f32 tiles: measured on MI355X the 16x16x4 form sustains ~20 % more FLOP/s than
  // 32x32x2 at the clocks the power manager grants under matrix load (tools/ubench/mfma_peak.hip); same
  // LDS operand traffic per FLOP, same exact-f32 fma chain.
  constexpr int MT = WM / 16;                 // 16-row MFMA tiles per wave: 4 or 2
  constexpr int NT = WN / 16;
  // Direct-to-LDS single-gather fp32 variants: the tile loads of a block can be issued by its first IGEMM_LOADER_WAVES
  // waves only (default: all 8).  With 4, waves 0-3 spend the head of every K-tile on addresses and load issue while
  // waves 4-7 -- their partners on the four SIMDs -- start their fragment reads and MFMAs at once, so the two waves of a
  // SIMD stop meeting the LDS and the matrix pipe in lockstep after every barrier (MI355X_MICROARCH.md, two waves per
  // SIMD, item 9).  A/B switch: `make alt ALTFLAGS=-DIGEMM_LOADER_WAVES=4`.  Measured (tools/ab_bench.sh, one box): 352 vs
  // 344 us for the resblock layer, step 161.1 vs 161.0 ms -- no gain, so the default stays 8.
#ifndef IGEMM_LOADER_WAVES
#define IGEMM_LOADER_WAVES 8
#endif
  constexpr int LW = (CT == 3 && ROLE != 2 && NWAVES == 8) ? IGEMM_LOADER_WAVES : NWAVES;
  constexpr int RSTEP = 64 * LW / 8;          // loader rows covered per pass (shadows the file-level constant)
  constexpr int AROWS = BM / RSTEP;           // gather rows per loader thread (512 threads x float4 = 64 rows)
  constexpr int BROWS = BN / RSTEP;           // weight rows per loader thread
  // double-buffered A/B tiles: one barrier per K-tile (72 KiB at BN=128: two blocks per CU)
  constexpr int CLD = BN + 4;   // row stride of the epilogue's C staging tile (floats): conflict-free b32 writes
  static_assert(BM * CLD <= 2 * (BM + BN) * LDS_LD, "C staging tile must fit in the operand buffers");
  // direct-to-LDS variant: unpadded tiles; with BN = 64 that is 48 KiB -> three blocks (6 waves per SIMD) per CU
  constexpr int SMEM_FLOATS = DMA ? (2 * (BM + BN) * 32 > BM * CLD ? 2 * (BM + BN) * 32 : BM * CLD) : 2 * (BM + BN) * LDS_LD;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  float* const As = smem;
  float* const Bs = smem + 2 * BM * LDS_LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware order: blocks b and b+8 share an XCD (round-robin dispatch), so give every XCD one
  // contiguous run of tiles; the n-tiles of an m-tile and vertically adjacent m-tiles (shared halo rows)
  // then hit the same L2.  Pure speed heuristic -- any placement computes the same result.
  int tile = blockIdx.x;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = tile & 7, idx = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int n_tile = tile % p.n_tiles;
  const int m_tile = tile / p.n_tiles;
  const int m0 = m_tile * BM;
  const int n0 = n_tile * BN;

  const float* __restrict__ xg = p.x;
  const float* __restrict__ wg = p.w;
  long long y_base = 0;   // element offset of this launch phase inside y
  if (p.ps > 1) {
    const int pa = blockIdx.y / p.ps, pb = blockIdx.y % p.ps;
    wg += (long long)blockIdx.y * p.w_phase;
    y_base = (long long)pa * p.y_phase_row + (long long)pb * p.y_phase_col;
  }

  // ---- loader coordinates ----
  const int c4 = tid & 7;   // float4 column inside the 32-wide K tile
  const int r0 = tid >> 3;  // 0..63
  int a_base[AROWS];            // b*H*W (pixel index of the sample's first pixel)
  int a_ih0[AROWS], a_iw0[AROWS];
  bool a_ok[AROWS];
  uint2 a_ch[AROWS], a_cw[AROWS];  // ROLE 2 only: packed fold candidates per axis
  int a_nc[AROWS];                 // ROLE 2 only: candidate counts (rows | cols << 4)
  int f_h1[PATCH ? AROWS : 1], f_w1[PATCH ? AROWS : 1];   // PATCH: second padded position per axis minus (K-1), or a sentinel
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    int m = m0 + r0 + RSTEP * i;
    a_ok[i] = m < p.M;
    int mm = a_ok[i] ? m : 0;
    int b, oh, ow;
    decode_pixel(mm, p.Ho, p.Wo, p.frame, b, oh, ow);
    a_base[i] = b * p.H * p.W;
    a_ih0[i] = oh * p.stride - p.pad;
    a_iw0[i] = ow * p.stride - p.pad;
    if constexpr (PATCH) {
      // padded positions folding onto (oh, ow): oh + P always; with reflect padding the mirror image of rows 1..P and
      // Hu-1-P..Hu-2 as well (the host admits this variant only when at most one of the two applies)
      constexpr int NONE = -(1 << 20);
      const int P = p.f_pad, K1 = p.KH - 1;
      int h1 = NONE, w1 = NONE;
      if (p.f_reflect) {
        if (oh >= 1 && oh <= P) h1 = P - oh - K1;
        else if (oh >= p.f_Hu - 1 - P && oh <= p.f_Hu - 2) h1 = 2 * p.f_Hu - 2 - oh + P - K1;
        if (ow >= 1 && ow <= P) w1 = P - ow - K1;
        else if (ow >= p.f_Wu - 1 - P && ow <= p.f_Wu - 2) w1 = 2 * p.f_Wu - 2 - ow + P - K1;
      }
      a_ih0[i] = oh + P - K1;
      a_iw0[i] = ow + P - K1;
      f_h1[i] = h1;
      f_w1[i] = w1;
    } else if constexpr (ROLE == 2) {
      a_ch[i] = fold_cands(oh, p.f_Hu, p.f_ups, p.f_pad, p.f_reflect);
      a_cw[i] = fold_cands(ow, p.f_Wu, p.f_ups, p.f_pad, p.f_reflect);
      int nh = 0, nw = 0;
      for (int a = 0; a < 4; ++a) {
        nh += cand_at(a_ch[i], a) != 0xFFFF;
        nw += cand_at(a_cw[i], a) != 0xFFFF;
      }
      a_nc[i] = nh | (nw << 4);
    }
  }

  f32x4 ra[AROWS], rb[BROWS];
  f32x4 rx[ROLE == 2 ? AROWS : 1];  // ROLE 2: second folded contribution per row
  int kh = 0, kw = 0, c0 = 0;   // aligned-mode K iterator
  const int nk_total = (p.Ktot + BK - 1) / BK;
  const int kt_begin = p.ksplit > 1 ? blockIdx.z * p.kt_per_split : 0;
  const int kt_end = p.ksplit > 1 ? min(nk_total, kt_begin + p.kt_per_split) : nk_total;
  if constexpr (ALIGNED) {
    if (kt_begin > 0) {
      const int k0 = kt_begin * BK, tap = k0 / p.Cin;
      c0 = k0 - tap * p.Cin;
      kh = tap / p.KW;
      kw = tap - kh * p.KW;
    }
  }
  // aligned mode: float offset of channel 0 of the pixel each loader row reads for the CURRENT tap
  // (-1 = contributes zero); recomputed only when the tap changes, i.e. every Cin/32 K-tiles
  int aoff[AROWS];
  int aoff1[ROLE == 2 ? AROWS : 1];
  int aoffp[PATCH ? 3 : 1][AROWS];   // PATCH: (h0,w1), (h1,w0), (h1,w1)
  f32x4 xp[PATCH ? 3 : 1][AROWS];    // PATCH: their values for the tile in flight
  bool wave_patch = false;           // PATCH: some lane of this wave has a partner at the current tap (wave-uniform)

  auto tap_setup = [&]() {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      if constexpr (PATCH) {
        const int vh0 = a_ih0[i] + kh, vh1 = f_h1[i] + kh, vw0 = a_iw0[i] + kw, vw1 = f_w1[i] + kw;
        const bool h0 = a_ok[i] && (unsigned)vh0 < (unsigned)p.H, h1 = a_ok[i] && (unsigned)vh1 < (unsigned)p.H;
        const bool w0 = (unsigned)vw0 < (unsigned)p.W, w1 = (unsigned)vw1 < (unsigned)p.W;
        aoff[i] = (h0 && w0) ? (a_base[i] + vh0 * p.W + vw0) * p.Cin : -1;
        aoffp[0][i] = (h0 && w1) ? (a_base[i] + vh0 * p.W + vw1) * p.Cin : -1;
        aoffp[1][i] = (h1 && w0) ? (a_base[i] + vh1 * p.W + vw0) * p.Cin : -1;
        aoffp[2][i] = (h1 && w1) ? (a_base[i] + vh1 * p.W + vw1) * p.Cin : -1;
        if (i == 0) wave_patch = false;
        wave_patch = wave_patch || __builtin_amdgcn_ballot_w64((aoffp[0][i] & aoffp[1][i] & aoffp[2][i]) >= 0) != 0;
      } else if constexpr (ROLE == 2) {
        const int nw = a_nc[i] >> 4;
        const int ncomb = (a_nc[i] & 15) * nw;
        int o0 = -1, o1 = -1;
        {
          const int vh = cand_at(a_ch[i], 0) + kh - (p.KH - 1);
          const int vw = cand_at(a_cw[i], 0) + kw - (p.KW - 1);
          const bool ok = a_ok[i] && vh >= 0 && vh < p.H && vw >= 0 && vw < p.W;
          o0 = ok ? (a_base[i] + vh * p.W + vw) * p.Cin : -1;
        }
        {
          const int vh = cand_at(a_ch[i], nw >= 2 ? 0 : 1) + kh - (p.KH - 1);
          const int vw = cand_at(a_cw[i], nw >= 2 ? 1 : 0) + kw - (p.KW - 1);
          const bool ok = a_ok[i] && ncomb >= 2 && vh >= 0 && vh < p.H && vw >= 0 && vw < p.W;
          o1 = ok ? (a_base[i] + vh * p.W + vw) * p.Cin : -1;
        }
        aoff[i] = o0;
        aoff1[i] = o1;
      } else {
        const int ih = src_coord(a_ih0[i] + kh, p.Hu, p.ups, p.reflect);
        const int iw = src_coord(a_iw0[i] + kw, p.Wu, p.ups, p.reflect);
        const bool ok = a_ok[i] && ih >= 0 && iw >= 0;
        aoff[i] = ok ? (a_base[i] + ih * p.W + iw) * p.Cin : -1;
      }
    }
  };

  auto load_tile = [&](int kt) {
    if constexpr (ALIGNED) {
      if (c0 == 0 || kt == kt_begin) tap_setup();
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (aoff[i] >= 0) v = *reinterpret_cast<const f32x4*>(xg + (long long)aoff[i] + c0 + c4 * 4);
        if constexpr (ROLE == 2) {
          // Combination 0 (always) and 1 (border pixels / up-sampling) are plain predicated loads into
          // separate registers so they stay in flight behind the MFMAs; only pixels that fold more than
          // two padded positions (corners, up-sampling convs) take the dependent loop below.
          f32x4 x1 = {0.f, 0.f, 0.f, 0.f};
          if (aoff1[i] >= 0) x1 = *reinterpret_cast<const f32x4*>(xg + (long long)aoff1[i] + c0 + c4 * 4);
          const int nw = a_nc[i] >> 4;
          const int ncomb = (a_nc[i] & 15) * nw;
          if (a_ok[i] && ncomb > 2) {
            // remaining combinations two at a time: both loads are issued before either is consumed, so a
            // corner pixel (4 combinations) pays one dependent round trip per K-tile, not two
            for (int cidx = 2; cidx < ncomb; cidx += 2) {
              f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
              {
                const int a = cidx / nw, b2 = cidx - a * nw;
                const int vh = cand_at(a_ch[i], a) + kh - (p.KH - 1);
                const int vw = cand_at(a_cw[i], b2) + kw - (p.KW - 1);
                if (vh >= 0 && vh < p.H && vw >= 0 && vw < p.W)
                  t0 = *reinterpret_cast<const f32x4*>(xg + (long long)(a_base[i] + vh * p.W + vw) * p.Cin + c0 + c4 * 4);
              }
              if (cidx + 1 < ncomb) {
                const int a = (cidx + 1) / nw, b2 = cidx + 1 - a * nw;
                const int vh = cand_at(a_ch[i], a) + kh - (p.KH - 1);
                const int vw = cand_at(a_cw[i], b2) + kw - (p.KW - 1);
                if (vh >= 0 && vh < p.H && vw >= 0 && vw < p.W)
                  t1 = *reinterpret_cast<const f32x4*>(xg + (long long)(a_base[i] + vh * p.W + vw) * p.Cin + c0 + c4 * 4);
              }
              x1 += t0 + t1;
            }
          }
          rx[i] = x1;
        }
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        const int n = n0 + r0 + RSTEP * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        // k index is linear in the tile number: tap*Cin + c0 == kt*BK
        if (n < p.Cout) v = *reinterpret_cast<const f32x4*>(wg + (long long)n * p.w_row + kt * BK + c4 * 4);
        rb[i] = v;
      }
      c0 += BK;
      if (c0 >= p.Cin) {
        c0 = 0;
        if (++kw == p.KW) { kw = 0; ++kh; }
      }
    } else {
      int ekh[4], ekw[4], eci[4];
      bool eok[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int k = kt * BK + c4 * 4 + e;
        eok[e] = k < p.Ktot;
        int kk = eok[e] ? k : 0;
        int tap = kk / p.Cin;
        eci[e] = kk - tap * p.Cin;
        ekh[e] = tap / p.KW;
        ekw[e] = tap - ekh[e] * p.KW;
      }
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int ih = src_coord(a_ih0[i] + ekh[e], p.Hu, p.ups, p.reflect);
          int iw = src_coord(a_iw0[i] + ekw[e], p.Wu, p.ups, p.reflect);
          bool ok = a_ok[i] && eok[e] && ih >= 0 && iw >= 0;
          float s = 0.f;
          if (ok) s = xg[(a_base[i] + (long long)ih * p.W + iw) * p.Cin + eci[e]];
          v[e] = s;
        }
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        int n = n0 + r0 + RSTEP * i;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int k = kt * BK + c4 * 4 + e;
          float s = 0.f;
          if (n < p.Cout && k < p.Ktot) s = wg[(long long)n * p.w_row + k];
          v[e] = s;
        }
        rb[i] = v;
      }
    }
  };

  const int dma_col = ((c4 ^ ((r0 >> 1) & 7)) * 4);   // global chunk (floats) this lane fetches; RSTEP % 16 == 0
  auto dma_tile = [&](int kt, int buf) {
    if constexpr (DMA) {
      if (LW < NWAVES && __builtin_amdgcn_readfirstlane(wave) >= LW) return;   // not a loader wave of this variant
      if constexpr (!CIN4) {
        if (c0 == 0 || kt == kt_begin) tap_setup();
      }
      const int wrow = __builtin_amdgcn_readfirstlane(wave) * 8;
      // CIN4: this lane's tap inside the K-tile (8 taps of 4 channels) and its position in the filter
      const int tap4 = kt * 8 + (dma_col >> 2);
      const int kh4 = tap4 / p.KW, kw4 = tap4 - kh4 * p.KW;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const float* g;
        if constexpr (CIN4) {
          const int ih = src_coord(a_ih0[i] + kh4, p.Hu, p.ups, p.reflect);
          const int iw = src_coord(a_iw0[i] + kw4, p.Wu, p.ups, p.reflect);
          const bool ok = a_ok[i] && tap4 < p.KH * p.KW && ih >= 0 && iw >= 0;
          g = ok ? xg + (long long)(a_base[i] + ih * p.W + iw) * 4 : munit_zero_page;
        } else {
          g = aoff[i] >= 0 ? xg + (long long)aoff[i] + c0 + dma_col : munit_zero_page + dma_col;
        }
        float* l = smem + buf * (BM * 32) + (wrow + RSTEP * i) * 32;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        if constexpr (PATCH) {
          if (wave_patch) {   // scalar branch: interior waves issue nothing here
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              f32x4 v = {0.f, 0.f, 0.f, 0.f};
              if (aoffp[c][i] >= 0) v = *reinterpret_cast<const f32x4*>(xg + (long long)aoffp[c][i] + c0 + dma_col);
              xp[c][i] = v;
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        const int n = min(n0 + r0 + RSTEP * i, p.Cout - 1);
        const float* g = wg + (long long)n * p.w_row + kt * BK + dma_col;
        float* l = smem + 2 * BM * 32 + buf * (BN * 32) + (wrow + RSTEP * i) * 32;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
      }
      if constexpr (!CIN4) {
        c0 += BK;
        if (c0 >= p.Cin) {
          c0 = 0;
          if (++kw == p.KW) { kw = 0; ++kh; }
        }
      }
    }
  };

  // bf16 tiles: [row][32] bf16 = 64-byte rows without padding; the 16-byte chunk c of a row lives at
  // position c ^ F[(row>>2)&3], F = {0,2,3,1}, which makes the fragment ds_read_b128 conflict-free
  // (every 16-lane service group then touches 16 distinct (row&3, position) pairs = all 64 banks).
  __bf16* const Ah = reinterpret_cast<__bf16*>(smem);
  __bf16* const Bh = Ah + (CT == 2 ? 3 : 2) * BM * 32;   // bf16: two buffers; f32x3: three planes, one buffer
  const int st_swz = (0x78 >> (2 * ((r0 >> 2) & 3))) & 3;     // RSTEP % 16 == 0: the same for all rows of a thread
  const int st_col = (((c4 >> 1) ^ st_swz) * 8) + (c4 & 1) * 4;  // bf16 index inside the row
  auto store_tile = [&](int buf) {
    if constexpr (BF16) {
      // plane pl of buffer buf: A at Ah + (buf*NPL + pl) * BM*32, B at Bh + (buf*NPL + pl) * BN*32
      auto put = [&](__bf16* base, int plane_elems, int row, f32x4 v) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          bf16x4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
          *reinterpret_cast<bf16x4*>(&base[pl * plane_elems + row * 32 + st_col]) = h;
          if (pl + 1 < NPL) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] -= (float)h[e];   // exact: the remainder fits fp32
          }
        }
      };
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        if constexpr (ROLE == 2) ra[i] += rx[i];
        put(Ah + buf * NPL * (BM * 32), BM * 32, r0 + RSTEP * i, ra[i]);
      }
#pragma unroll
      for (int i = 0; i < BROWS; ++i) put(Bh + buf * NPL * (BN * 32), BN * 32, r0 + RSTEP * i, rb[i]);
      return;
    }
    float* Ad = As + buf * (BM * LDS_LD);
    float* Bd = Bs + buf * (BN * LDS_LD);
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      if constexpr (ROLE == 2) ra[i] += rx[i];
      *reinterpret_cast<f32x4*>(&Ad[(r0 + RSTEP * i) * LDS_LD + c4 * 4]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i)
      *reinterpret_cast<f32x4*>(&Bd[(r0 + RSTEP * i) * LDS_LD + c4 * 4]) = rb[i];
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mt][nt][r] = 0.f;

  // 16x16x4 operand map: lane l holds A[row l&15][k = l>>4] / B[k = l>>4][col l&15].  Each lane reads
  // four consecutive k of its row with one ds_read_b128 and uses element t in MFMA step t, i.e. step t
  // contracts k = {4g + t : g = 0..3} of the 16-wide group -- a permutation of k, identical for A and B.
  const int frag_row = lane & 15;
  const int frag_k = (lane >> 4) * 4;
  // one of the two 16-wide k groups of a K-tile: (MT + NT) ds_read_b128, 4 * MT * NT MFMAs
  const int fr_col = (((lane >> 4) ^ ((0x78 >> (2 * ((frag_row >> 2) & 3))) & 3)) * 8);   // bf16 fragment chunk
  auto compute_half = [&](int buf, int kg) {
    if constexpr (BF16) {
      // one v_mfma_f32_16x16x32_bf16 per tile pair (and per product term) covers the whole 32-deep K-tile: half
      // kg takes half of the row tiles
      const __bf16* Ac = Ah + buf * NPL * (BM * 32);
      const __bf16* Bc = Bh + buf * NPL * (BN * 32);
      bf16x8 hb[NPL][NT];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          hb[pl][nt] = *reinterpret_cast<const bf16x8*>(&Bc[pl * (BN * 32) + (wn * WN + nt * 16 + frag_row) * 32 + fr_col]);
#pragma unroll
      for (int mt = kg * (MT / 2); mt < (kg + 1) * (MT / 2); ++mt) {
        bf16x8 ha[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          ha[pl] = *reinterpret_cast<const bf16x8*>(&Ac[pl * (BM * 32) + (wm * WM + mt * 16 + frag_row) * 32 + fr_col]);
        // product terms a_i * b_j, i + j <= NPL - 1, smallest first
#pragma unroll
        for (int sum = NPL - 1; sum >= 0; --sum)
#pragma unroll
          for (int i = 0; i <= sum; ++i)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[i], hb[sum - i][nt], acc[mt][nt], 0, 0, 0);
      }
      return;
    }
    if constexpr (DMA) {
      const float* Ac = smem + buf * (BM * 32);
      const float* Bc = smem + 2 * BM * 32 + buf * (BN * 32);
      const int pos = ((kg * 4 + (lane >> 4)) ^ ((frag_row >> 1) & 7)) * 4;
      f32x4 a[MT], b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        b[nt] = *reinterpret_cast<const f32x4*>(&Bc[(wn * WN + nt * 16 + frag_row) * 32 + pos]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        a[mt] = *reinterpret_cast<const f32x4*>(&Ac[(wm * WM + mt * 16 + frag_row) * 32 + pos]);
      if constexpr (BF16S) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[mt]),
                                                                  __builtin_bit_cast(bf16x8, b[nt]), acc[mt][nt], 0, 0, 0);
        return;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][t], b[nt][t], acc[mt][nt], 0, 0, 0);
      return;
    }
    const float* Ac = As + buf * (BM * LDS_LD);
    const float* Bc = Bs + buf * (BN * LDS_LD);
    f32x4 a[MT], b[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      b[nt] = *reinterpret_cast<const f32x4*>(&Bc[(wn * WN + nt * 16 + frag_row) * LDS_LD + kg * 16 + frag_k]);
    // ROLE 2 carries two gather slots per row and has no registers to spare: it reads the A fragments in
    // two batches (the scheduling fence keeps the compiler from hoisting the second batch over the MFMAs)
    constexpr int MB = (ROLE == 2 && MT >= 4) ? MT / 2 : MT;
#pragma unroll
    for (int m0t = 0; m0t < MT; m0t += MB) {
#pragma unroll
      for (int mt = m0t; mt < m0t + MB; ++mt)
        a[mt] = *reinterpret_cast<const f32x4*>(&Ac[(wm * WM + mt * 16 + frag_row) * LDS_LD + kg * 16 + frag_k]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = m0t; mt < m0t + MB; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][t], b[nt][t], acc[mt][nt], 0, 0, 0);
      if constexpr (MB != MT) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Pipeline: while tile t is multiplied out of LDS buffer t&1, the registers hold tile t+1 (loads issued
  // during tile t-1); half-way through the MFMAs they are written to the other buffer -- safe, every wave
  // passed the barrier that ended tile t-1 and nobody reads that buffer before the next barrier -- and the
  // loads of tile t+2 are issued into the freed registers.  One barrier per K-tile.
  const int nk = kt_end - kt_begin;
  // PATCH: add the partner combinations to this lane's own landing spot (row r0 + RSTEP*i, position c4) of buffer buf;
  // called after s_waitcnt vmcnt(0), i.e. once the wave's own direct-to-LDS rows and the partner registers have arrived
  auto patch_tile = [&](int buf) {
    if constexpr (PATCH) {
      if (!wave_patch) return;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        if ((aoffp[0][i] & aoffp[1][i] & aoffp[2][i]) >= 0) {   // any of the three offsets is valid (-1 = none)
          f32x4* l = reinterpret_cast<f32x4*>(smem + buf * (BM * 32) + (r0 + RSTEP * i) * 32 + c4 * 4);
          if constexpr (BF16S) *l = bf16x8_add4(*l, xp[0][i], xp[1][i], xp[2][i]);
          else *l = *l + ((xp[0][i] + xp[1][i]) + xp[2][i]);
        }
      }
    }
  };
  if constexpr (DMA) {
    if (nk > 0) dma_tile(kt_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nk > 0) patch_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) dma_tile(kt_begin + kt + 1, cur ^ 1);   // every wave finished reading that buffer before the last barrier
      compute_half(cur, 0);
      compute_half(cur, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (kt + 1 < nk) patch_tile(cur ^ 1);
      __syncthreads();
    }
  } else if constexpr (CT == 2) {
    // three planes, one LDS buffer: barrier, split the registers (tile kt) into the planes, issue the loads of
    // tile kt+1, barrier, multiply.  The other resident block covers the store phase.
    if (nk > 0) load_tile(kt_begin);
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();
      store_tile(0);
      if (kt + 1 < nk) load_tile(kt_begin + kt + 1);
      __syncthreads();
      compute_half(0, 0);
      compute_half(0, 1);
    }
    __syncthreads();
  } else {
    if (nk > 0) {
      load_tile(kt_begin);
      store_tile(0);
    }
    __syncthreads();
    if (nk > 1) load_tile(kt_begin + 1);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      compute_half(cur, 0);
      if (kt + 1 < nk) {
        // ABL_*: timing-only ablation builds (`make alt ALTFLAGS=-DABL_NOGLOBAL`, results are wrong), DESIGN.md section 3.5
#ifndef ABL_NOLDSW
        store_tile(cur ^ 1);
#endif
#ifndef ABL_NOGLOBAL
        if (kt + 2 < nk) load_tile(kt_begin + kt + 2);
#endif
      }
      if constexpr (ROLE == 2) __builtin_amdgcn_sched_barrier(0);   // keep the fragment reads below the gather (registers)
      compute_half(cur, 1);
#ifndef ABL_NOBAR
      __syncthreads();
#endif
    }
  }

  // ---- epilogue ----
  // The accumulators go through LDS (the operand buffers are free: the K loop ended with a barrier) so that
  // every global store is a full row segment: a quarter-wave writing 64 B per pixel straight from the MFMA
  // layout left the write path at ~1.8 TB/s and the 33 MB tile flush cost ~18 us per launch.
  float* const Cs = smem;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r)   // C/D map: col = lane&15, row = 4*(lane>>4)+r
        Cs[(wm * WM + mt * 16 + 4 * (lane >> 4) + r) * CLD + wn * WN + nt * 16 + (lane & 15)] = acc[mt][nt][r];
  __syncthreads();
  constexpr int CQ = BN / 4;          // float4 per tile row
  constexpr int CRPT = NTHR / CQ;     // rows per pass
  const int cq = tid % CQ, cr0 = tid / CQ;
  const int n = n0 + cq * 4;
  const bool split = p.ksplit > 1;
  // raw partial tile -> slab[(phase*ksplit + split)][m][n], or bias + activation -> y (NHWC)
  float* const sl = split ? p.slab + ((long long)blockIdx.y * p.ksplit + blockIdx.z) * (long long)p.M * p.Cout : nullptr;
  const bool vec = (p.Cout & 3) == 0 && (split || ((p.y_sw & 3) == 0 && (p.y_sh & 3) == 0 && (p.y_sb & 3) == 0));
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (!split && p.bias != nullptr) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n + e < p.Cout) bv[e] = p.bias[n + e];
  }
#pragma unroll 2
  for (int row = cr0; row < BM; row += CRPT) {
    const int m = m0 + row;
    if (m >= p.M || n >= p.Cout) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * CLD + cq * 4]);
    if (split) {
      float* dst = sl + (long long)m * p.Cout + n;
      if (vec) {
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.Cout) dst[e] = v[e];
      }
      continue;
    }
    int b, oh, ow;
    decode_pixel(m, p.Ho, p.Wo, p.frame, b, oh, ow);
    const long long off = y_base + (long long)b * p.y_sb + (long long)oh * p.y_sh + (long long)ow * p.y_sw + n;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e] + bv[e], p.act, p.slope);
    if (p.out_bf16) {
      bf16_t* dst = reinterpret_cast<bf16_t*>(p.y) + off;
      if (vec) {
        st4(dst, v);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.Cout) dst[e] = (bf16_t)v[e];
      }
    } else {
      float* dst = reinterpret_cast<float*>(p.y) + off;
      if (vec) {
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.Cout) dst[e] = v[e];
      }
    }
  }
}

// y[pixel(m)][n] = act(sum_s slab[(phase*ksplit + s)][m][n] + bias[n])   (fixed summation order)
__global__ void splitk_epilogue_kernel(IgemmParams p, int phases) {
  const long long per = (long long)p.M * p.Cout;
  const long long total = per * phases;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ph = (int)(i / per);
    const long long r = i - (long long)ph * per;
    const int m = (int)(r / p.Cout), n = (int)(r - (long long)m * p.Cout);
    float s = 0.f;
    for (int k = 0; k < p.ksplit; ++k) s += p.slab[((long long)ph * p.ksplit + k) * per + r];
    int b, oh, ow;
    decode_pixel(m, p.Ho, p.Wo, p.frame, b, oh, ow);
    long long off = (long long)b * p.y_sb + (long long)oh * p.y_sh + (long long)ow * p.y_sw + n;
    if (p.ps > 1) off += (long long)(ph / p.ps) * p.y_phase_row + (long long)(ph % p.ps) * p.y_phase_col;
    const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
    const float res = apply_act(s + bv, p.act, p.slope);
    if (p.out_bf16) reinterpret_cast<bf16_t*>(p.y)[off] = (bf16_t)res;
    else reinterpret_cast<float*>(p.y)[off] = res;
  }
}

// ---- prepared weights -------------------------------------------------------------------------------------
// Two layers of the path multiply by a re-laid-out image of their weights: backward-data (flipped / transposed,
// one slice per stride phase) and the sub-pixel forward of the up-sampling convs (merged phase weights).  Weights
// only change at the optimizer step, so the caller may keep these images (munit_conv2d_prepare_weights[_batch])
// and pass them to the *_prepared entry points; without one the image is rebuilt into the workspace per call.
typedef munit_prep_item PrepItem;

// backward-data: w [Cout][KH][KW][Cin] -> wt [ps*ps phases][Cin][T][T][Cout], T = K/ps, phase (pa,pb), tap (t,r):
//   wt[ph][ci][t][r][co] = w[co][pa + ps*(T-1-t)][pb + ps*(T-1-r)][ci]
__device__ inline void prep_dgrad_elem(const PrepItem& it, long long i) {
  const int ps = it.ps, TH = it.KH / ps, TW = it.KW / ps;
  long long r_ = i;
  int co = (int)(r_ % it.Cout); r_ /= it.Cout;
  int r = (int)(r_ % TW); r_ /= TW;
  int t = (int)(r_ % TH); r_ /= TH;
  int ci = (int)(r_ % it.Cin); r_ /= it.Cin;
  int ph = (int)r_;
  int pa = ph / ps, pb = ph % ps;
  int kh = pa + ps * (TH - 1 - t);
  int kw = pb + ps * (TW - 1 - r);
  const float v = it.w[(((long long)co * it.KH + kh) * it.KW + kw) * it.Cin + ci];
  if (it.bf16) reinterpret_cast<bf16_t*>(it.wp)[i] = (bf16_t)v;
  else it.wp[i] = v;
}

// forward of the bf16-storage mode: the weights as they are, rounded to bf16
__device__ inline void prep_cast_elem(const PrepItem& it, long long i) { reinterpret_cast<bf16_t*>(it.wp)[i] = (bf16_t)it.w[i]; }

// Sub-pixel form of nearest-x2-upsample + 5x5 conv: output pixel (2i+a, 2j+b) reads source rows
// i-1, i, i+1 with the 5 kernel rows merged as  a=0: {0,1} {2,3} {4}   a=1: {0} {1,2} {3,4}  (same for
// columns), so each of the 4 phases is a 3x3 conv over the source with summed weights: 36 instead of
// 100 MACs per source pixel and channel pair.  wc: [phase = a*2+b][Cout][3][3][Cin].
__device__ inline void prep_subpixel_elem(const PrepItem& it, long long i) {
  const long long merged = (long long)4 * 9 * it.Cout * it.Cin;
  if (i >= merged) {   // bf16 image only: copy of the 5x5 weights
    reinterpret_cast<bf16_t*>(it.wp)[i] = (bf16_t)it.w[i - merged];
    return;
  }
  long long r = i;
  const int ci = (int)(r % it.Cin); r /= it.Cin;
  const int dw = (int)(r % 3); r /= 3;
  const int dh = (int)(r % 3); r /= 3;
  const int co = (int)(r % it.Cout); r /= it.Cout;
  const int ph = (int)r;
  const int a = ph >> 1, b = ph & 1;
  // kernel rows merged into tap dh of phase a: first row and count
  const int h0 = a == 0 ? (dh == 0 ? 0 : dh == 1 ? 2 : 4) : (dh == 0 ? 0 : dh == 1 ? 1 : 3);
  const int hn = a == 0 ? (dh == 2 ? 1 : 2) : (dh == 0 ? 1 : 2);
  const int w0 = b == 0 ? (dw == 0 ? 0 : dw == 1 ? 2 : 4) : (dw == 0 ? 0 : dw == 1 ? 1 : 3);
  const int wn = b == 0 ? (dw == 2 ? 1 : 2) : (dw == 0 ? 1 : 2);
  float s = 0.f;
  for (int kh = h0; kh < h0 + hn; ++kh)
    for (int kw = w0; kw < w0 + wn; ++kw) s += it.w[(((long long)co * 5 + kh) * 5 + kw) * it.Cin + ci];
  if (it.bf16) reinterpret_cast<bf16_t*>(it.wp)[i] = (bf16_t)s;
  else it.wp[i] = s;
}

// elements of an image.  The bf16 sub-pixel image is [4 merged 3x3 phase kernels][the 5x5 weights themselves] (the frame
// launch of the sub-pixel forward multiplies by the latter).
__host__ __device__ inline long long prep_elems(const PrepItem& it) {
  const long long cc = (long long)it.Cout * it.Cin;
  if (it.kind == MUNIT_PREP_SUBPIXEL) return it.bf16 ? (4 * 9 + 25) * cc : 4 * 9 * cc;
  if (it.kind == MUNIT_PREP_WINOGRAD || it.kind == MUNIT_PREP_WINOGRAD_DGRAD) return wino_image_elems(it.Cin, it.Cout);
  if (it.kind == MUNIT_PREP_SUBPIXEL_WINOGRAD_DGRAD) return cc * it.KH * it.KW + 4 * wino_image_elems(it.Cin, it.Cout);
  if (it.kind == MUNIT_PREP_SUBPIXEL_WINOGRAD || it.kind == MUNIT_PREP_WINOGRAD_S2 || it.kind == MUNIT_PREP_WINOGRAD_S2_DGRAD)
    return 4 * wino_image_elems(it.Cin, it.Cout);
  return cc * it.KH * it.KW;
}

// blockIdx.y = item (DEV: table in device memory, one launch re-lays every weight of an optimizer; else the one
// item passed by value), grid-stride over the item's elements in x
// loop trips of one image: its elements, or -- Winograd images -- its (k, n) channel pairs (16 elements each)
__host__ __device__ inline long long prep_trips(const PrepItem& it) {
  const bool wino = it.kind == MUNIT_PREP_WINOGRAD || it.kind == MUNIT_PREP_WINOGRAD_DGRAD || it.kind == MUNIT_PREP_SUBPIXEL_WINOGRAD ||
                    it.kind == MUNIT_PREP_WINOGRAD_S2 || it.kind == MUNIT_PREP_WINOGRAD_S2_DGRAD;
  if (it.kind == MUNIT_PREP_SUBPIXEL_WINOGRAD_DGRAD) return (long long)it.Cout * it.Cin * (it.KH * it.KW + 4);
  return wino ? prep_elems(it) / 16 : prep_elems(it);
}
template <bool DEV>
__global__ void prep_weights_kernel(const PrepItem* __restrict__ items, PrepItem one) {
  const PrepItem it = DEV ? items[blockIdx.y] : one;
  const long long total = prep_trips(it);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    if (it.kind == MUNIT_PREP_SUBPIXEL) prep_subpixel_elem(it, i);
    else if (it.kind == MUNIT_PREP_WINOGRAD || it.kind == MUNIT_PREP_WINOGRAD_DGRAD)
      wino_weight_item(it.w, it.wp, it.Cout, it.Cin, it.kind == MUNIT_PREP_WINOGRAD_DGRAD, i);
    else if (it.kind == MUNIT_PREP_SUBPIXEL_WINOGRAD) wino_subpixel_weight_item(it.w, it.wp, it.Cout, it.Cin, i);
    else if (it.kind == MUNIT_PREP_WINOGRAD_S2) wino_s2_weight_item(it.w, it.wp, it.Cout, it.Cin, i);
    else if (it.kind == MUNIT_PREP_WINOGRAD_S2_DGRAD) wino_s2_dgrad_weight_item(it.w, it.wp, it.Cout, it.Cin, i);
    else if (it.kind == MUNIT_PREP_SUBPIXEL_WINOGRAD_DGRAD) {
      const long long nd = (long long)it.Cout * it.Cin * it.KH * it.KW;   // the transposed weights first (frame launch) ...
      if (i < nd) prep_dgrad_elem(it, i);
      else wino_subpixel_dgrad_weight_item(it.w, it.wp + nd, it.Cout, it.Cin, i - nd);   // ... then the Winograd image
    }
    else if (it.kind == MUNIT_PREP_CAST) prep_cast_elem(it, i);
    else prep_dgrad_elem(it, i);
  }
}

int launch_prep_one(const PrepItem& it, hipStream_t st) {
  const long long total = prep_trips(it);
  const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(prep_weights_kernel<false>, dim3(blocks), dim3(256), 0, st, nullptr, it);
  MUNIT_CHECK_LAUNCH("prep_weights");
  return MUNIT_OK;
}

// Adjoint of (nearest x2 upsample) + (reflect | zero pad): gather-sum the padded-domain
// gradient g[B][Hq][Wq][C] (rows/cols beyond Hq/Wq are zero) into dx[B][H][W][C].
template <typename T>
__global__ void fold_kernel(const T* __restrict__ g, const T* __restrict__ add,
                            T* __restrict__ dx, int B, int H, int W, int C, int ups, int pad,
                            int reflect, int Hq, int Wq) {
  const int C4 = C >> 2;
  const long long total = (long long)B * H * W * C4;
  const int Hu = H << ups, Wu = W << ups;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r_ = i;
    int c4 = (int)(r_ % C4); r_ /= C4;
    int iw = (int)(r_ % W); r_ /= W;
    int ih = (int)(r_ % H); r_ /= H;
    int b = (int)r_;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const int nu = 1 << ups;
    for (int du = 0; du < nu; ++du) {
      int hu = (ih << ups) + du;
      int qr[3], nr = 0;
      qr[nr++] = hu + pad;
      if (reflect) {
        if (hu >= 1 && hu <= pad) qr[nr++] = pad - hu;
        if (hu >= Hu - 1 - pad && hu <= Hu - 2) qr[nr++] = 2 * Hu - 2 - hu + pad;
      }
      for (int dv = 0; dv < nu; ++dv) {
        int wu = (iw << ups) + dv;
        int qc[3], nc = 0;
        qc[nc++] = wu + pad;
        if (reflect) {
          if (wu >= 1 && wu <= pad) qc[nc++] = pad - wu;
          if (wu >= Wu - 1 - pad && wu <= Wu - 2) qc[nc++] = 2 * Wu - 2 - wu + pad;
        }
        for (int a = 0; a < nr; ++a) {
          if (qr[a] >= Hq) continue;
          for (int c = 0; c < nc; ++c) {
            if (qc[c] >= Wq) continue;
            s += ld4(g + (((long long)b * Hq + qr[a]) * Wq + qc[c]) * C + c4 * 4);
          }
        }
      }
    }
    if (add != nullptr) s += ld4(add + i * 4);
    st4(dx + i * 4, s);
  }
}

// scalar-channel variant of fold_kernel for C % 4 != 0 (3-channel images)
__global__ void fold_scalar_kernel(const float* __restrict__ g, const float* __restrict__ add,
                                   float* __restrict__ dx, int B, int H, int W, int C, int ups, int pad,
                                   int reflect, int Hq, int Wq) {
  const long long total = (long long)B * H * W * C;
  const int Hu = H << ups, Wu = W << ups;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r_ = i;
    int c = (int)(r_ % C); r_ /= C;
    int iw = (int)(r_ % W); r_ /= W;
    int ih = (int)(r_ % H); r_ /= H;
    int b = (int)r_;
    float s = 0.f;
    const int nu = 1 << ups;
    for (int du = 0; du < nu; ++du) {
      int hu = (ih << ups) + du;
      int qr[3], nr = 0;
      qr[nr++] = hu + pad;
      if (reflect) {
        if (hu >= 1 && hu <= pad) qr[nr++] = pad - hu;
        if (hu >= Hu - 1 - pad && hu <= Hu - 2) qr[nr++] = 2 * Hu - 2 - hu + pad;
      }
      for (int dv = 0; dv < nu; ++dv) {
        int wu = (iw << ups) + dv;
        int qc[3], nc = 0;
        qc[nc++] = wu + pad;
        if (reflect) {
          if (wu >= 1 && wu <= pad) qc[nc++] = pad - wu;
          if (wu >= Wu - 1 - pad && wu <= Wu - 2) qc[nc++] = 2 * Wu - 2 - wu + pad;
        }
        for (int a = 0; a < nr; ++a) {
          if (qr[a] >= Hq) continue;
          for (int cc = 0; cc < nc; ++cc) {
            if (qc[cc] >= Wq) continue;
            s += g[(((long long)b * Hq + qr[a]) * Wq + qc[cc]) * C + c];
          }
        }
      }
    }
    if (add != nullptr) s += add[i];
    dx[i] = s;
  }
}

__global__ void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ a, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] += a[i];
}

// Split-K factor for grids that cannot fill the chip: a 128x128 tile with K = 4096 is 128 sequential
// K-tiles (~150-290 us) however few blocks there are.  0 workspace -> no split.
int pick_ksplit(int M, int Cout, int Ktot, int phases) {
  const int bn = Cout <= 64 ? 64 : 128;
  const int tiles = cdiv(M, BM) * cdiv(Cout, bn) * phases;
  const int nk = cdiv(Ktot, BK);
  if (tiles >= 192 || nk < 16 || MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_SPLITK")) return 1;
  int ks = std::min(cdiv(512, tiles), nk / 4);
  return std::max(1, std::min(ks, 32));
}
size_t splitk_bytes(int M, int Cout, int Ktot, int phases) {
  const int ks = pick_ksplit(M, Cout, Ktot, phases);
  return ks > 1 ? align_up((size_t)ks * phases * M * Cout * sizeof(float), 256) : 0;
}

template <int ROLE>
int launch_igemm(const IgemmParams& p, int phases, hipStream_t st, void* slab = nullptr, size_t slab_bytes = 0) {
  const int bn = p.Cout <= 64 ? 64 : 128;
  IgemmParams q = p;
  if (p.bf16s) {
    // bf16 tensors seen as float tensors with half the channels (a tile row is 128 bytes either way)
    if (p.Cin % (2 * BK) != 0 || NWAVES != 8) {
      munit_set_error("conv_igemm: bf16 storage needs a multiple of 64 input channels (got %d)", p.Cin);
      return MUNIT_ERR_ARG;
    }
    q.Cin = p.Cin / 2; q.Ktot = p.Ktot / 2; q.w_row = p.w_row / 2; q.w_phase = p.w_phase / 2;
    slab = nullptr;   // no split-K in this mode
  }
  const bool aligned = (q.Cin % BK == 0) && (q.w_row % 4 == 0);
  q.n_tiles = cdiv(p.Cout, bn);
  const int m_tiles = cdiv(p.M, BM);
  q.ksplit = 1;
  if (slab != nullptr) {
    const int ks = pick_ksplit(p.M, p.Cout, p.Ktot, phases);
    if (ks > 1 && slab_bytes >= splitk_bytes(p.M, p.Cout, p.Ktot, phases)) {
      const int nk = cdiv(p.Ktot, BK);
      q.kt_per_split = cdiv(nk, ks);
      q.ksplit = cdiv(nk, q.kt_per_split);
      q.slab = reinterpret_cast<float*>(slab);
    }
  }
  dim3 grid((unsigned)(m_tiles * q.n_tiles), (unsigned)phases, (unsigned)q.ksplit);
  dim3 block(NTHR);
  if (p.cin4) {
    if constexpr (ROLE == 2) {
      munit_set_error("conv_igemm: 4-channel taps are not a folded-backward-data form");
      return MUNIT_ERR_ARG;
    } else {
      if (bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<64, true, ROLE, 5>), grid, block, 0, st, q);
      else hipLaunchKernelGGL((conv_igemm_kernel<128, true, ROLE, 5>), grid, block, 0, st, q);
    }
  } else if (p.bf16s) {
    if constexpr (ROLE == 2) {
      if (!p.patch || p.frame) {
        munit_set_error("conv_igemm: bf16-storage folded backward-data exists for the LDS-patch form only");
        return MUNIT_ERR_ARG;
      }
    }
    if (bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<64, true, ROLE, 4>), grid, block, 0, st, q);
    else hipLaunchKernelGGL((conv_igemm_kernel<128, true, ROLE, 4>), grid, block, 0, st, q);
  } else if constexpr (ROLE == 2) {
    if (!aligned) {
      munit_set_error("conv_igemm: folded backward-data needs Cout %% 32 == 0");
      return MUNIT_ERR_ARG;
    }
    if (p.ct == 0 && p.patch && !p.frame && NWAVES == 8 && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_PATCH")) {
      if (bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<64, true, 2, 3>), grid, block, 0, st, q);
      else hipLaunchKernelGGL((conv_igemm_kernel<128, true, 2, 3>), grid, block, 0, st, q);
    } else if (bn == 64) {
      if (p.ct == 2) hipLaunchKernelGGL((conv_igemm_kernel<64, true, 2, 2>), grid, block, 0, st, q);
      else if (p.ct == 1) hipLaunchKernelGGL((conv_igemm_kernel<64, true, 2, 1>), grid, block, 0, st, q);
      else hipLaunchKernelGGL((conv_igemm_kernel<64, true, 2>), grid, block, 0, st, q);
    } else {
      if (p.ct == 2) hipLaunchKernelGGL((conv_igemm_kernel<128, true, 2, 2>), grid, block, 0, st, q);
      else if (p.ct == 1) hipLaunchKernelGGL((conv_igemm_kernel<128, true, 2, 1>), grid, block, 0, st, q);
      else hipLaunchKernelGGL((conv_igemm_kernel<128, true, 2>), grid, block, 0, st, q);
    }
  } else if (aligned && p.ct == 0 && NWAVES == 8 && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_DMA")) {
    // single-gather fp32 variants: tiles go global -> LDS directly
    if (bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<64, true, ROLE, 3>), grid, block, 0, st, q);
    else hipLaunchKernelGGL((conv_igemm_kernel<128, true, ROLE, 3>), grid, block, 0, st, q);
  } else if (bn == 64) {
    if (aligned && p.ct == 2) hipLaunchKernelGGL((conv_igemm_kernel<64, true, ROLE, 2>), grid, block, 0, st, q);
    else if (aligned && p.ct == 1) hipLaunchKernelGGL((conv_igemm_kernel<64, true, ROLE, 1>), grid, block, 0, st, q);
    else if (aligned) hipLaunchKernelGGL((conv_igemm_kernel<64, true, ROLE>), grid, block, 0, st, q);
    else hipLaunchKernelGGL((conv_igemm_kernel<64, false, ROLE>), grid, block, 0, st, q);
  } else {
    if (aligned && p.ct == 2) hipLaunchKernelGGL((conv_igemm_kernel<128, true, ROLE, 2>), grid, block, 0, st, q);
    else if (aligned && p.ct == 1) hipLaunchKernelGGL((conv_igemm_kernel<128, true, ROLE, 1>), grid, block, 0, st, q);
    else if (aligned) hipLaunchKernelGGL((conv_igemm_kernel<128, true, ROLE>), grid, block, 0, st, q);
    else hipLaunchKernelGGL((conv_igemm_kernel<128, false, ROLE>), grid, block, 0, st, q);
  }
  MUNIT_CHECK_LAUNCH("conv_igemm");
  if (q.ksplit > 1) {
    const long long total = (long long)p.M * p.Cout * phases;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, st, q, phases);
    MUNIT_CHECK_LAUNCH("splitk_epilogue");
  }
  return MUNIT_OK;
}

int check_desc(const munit_conv_desc* d) {
  MUNIT_CHECK_ARG(d != nullptr, "conv: null descriptor");
  MUNIT_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv: bad dims");
  MUNIT_CHECK_ARG(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "conv: bad kernel geometry");
  MUNIT_CHECK_ARG(d->upsample == 0 || d->upsample == 1, "conv: upsample must be 0 or 1");
  MUNIT_CHECK_ARG(d->compute >= MUNIT_COMPUTE_F32 && d->compute <= MUNIT_COMPUTE_F32X3, "conv: bad compute mode %d", d->compute);
  MUNIT_CHECK_ARG(d->pad_mode == MUNIT_PAD_ZERO || d->pad_mode == MUNIT_PAD_REFLECT, "conv: bad pad mode");
  MUNIT_CHECK_ARG((d->in_dtype == MUNIT_DTYPE_F32 || d->in_dtype == MUNIT_DTYPE_BF16) &&
                  (d->out_dtype == MUNIT_DTYPE_F32 || d->out_dtype == MUNIT_DTYPE_BF16), "conv: bad tensor dtype");
  MUNIT_CHECK_ARG(d->in_dtype == MUNIT_DTYPE_F32 || d->Cin % 64 == 0, "conv: a bf16 input needs Cin %% 64 == 0 (got %d)", d->Cin);
  MUNIT_CHECK_ARG(d->out_dtype == MUNIT_DTYPE_F32 || d->Cout % 64 == 0, "conv: a bf16 output needs Cout %% 64 == 0 (got %d)", d->Cout);
  const int Hu = d->H << d->upsample, Wu = d->W << d->upsample;
  if (d->pad_mode == MUNIT_PAD_REFLECT)
    MUNIT_CHECK_ARG(d->pad < Hu && d->pad < Wu, "conv: reflect pad %d needs input > pad (got %dx%d)", d->pad, Hu, Wu);
  MUNIT_CHECK_ARG(Hu + 2 * d->pad >= d->KH && Wu + 2 * d->pad >= d->KW, "conv: kernel larger than padded input");
  MUNIT_CHECK_ARG((long long)d->B * Hu * Wu < (1ll << 31) / 4, "conv: too many pixels for 32-bit tile indices");
  MUNIT_CHECK_ARG((long long)d->B * (Hu + 2 * d->pad) * (Wu + 2 * d->pad) * std::max(d->Cin, d->Cout) < (1ll << 31),
                  "conv: tensor too large for the 32-bit element offsets of the tile loader");
  return MUNIT_OK;
}

}  // namespace

extern "C" int munit_conv2d_out_hw(const munit_conv_desc* d, int* Ho, int* Wo) {
  int rc = check_desc(d);
  if (rc) return rc;
  *Ho = ((d->H << d->upsample) + 2 * d->pad - d->KH) / d->stride + 1;
  *Wo = ((d->W << d->upsample) + 2 * d->pad - d->KW) / d->stride + 1;
  return MUNIT_OK;
}

namespace {
// nearest-x2 upsample + 5x5 reflect-pad-2 stride-1 conv: eligible for the sub-pixel decomposition
bool subpixel_ok(const munit_conv_desc* d) {
  return d->upsample == 1 && d->KH == 5 && d->KW == 5 && d->pad == 2 && d->stride == 1 &&
         d->pad_mode == MUNIT_PAD_REFLECT && d->Cin % 32 == 0 && d->H >= 3 && d->W >= 3 &&
         !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_SUBPIXEL");
}
}  // namespace

namespace {
// ---- three input channels on the direct-to-LDS path (CT == 5): 4-channel re-layouts built per call into the workspace
__global__ void pad3to4_image_kernel(const float* __restrict__ x, f32x4* __restrict__ x4, long long npix) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x)
    x4[i] = f32x4{x[3 * i], x[3 * i + 1], x[3 * i + 2], 0.f};
}
// w [N][taps][3] -> w4 [N][kpad], kpad = taps * 4 rounded up to the 32-wide K-tile, zero channel 3 and zero tail
__global__ void pad3to4_weight_kernel(const float* __restrict__ w, float* __restrict__ w4, int N, int taps, int kpad) {
  const long long total = (long long)N * kpad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i / kpad), k = (int)(i - (long long)n * kpad);
    const int tap = k >> 2, c = k & 3;
    w4[i] = (tap < taps && c < 3) ? w[((long long)n * taps + tap) * 3 + c] : 0.f;
  }
}
struct Cin4Plan {
  int kpad;
  size_t x4_bytes, w4_bytes;
};
// npix pixels of the 3-channel tensor, N GEMM columns, taps filter taps
Cin4Plan plan_cin4(long long npix, int N, int taps) {
  Cin4Plan c;
  c.kpad = (taps * 4 + BK - 1) / BK * BK;
  c.x4_bytes = align_up((size_t)npix * 4 * sizeof(float), 256);
  c.w4_bytes = align_up((size_t)N * c.kpad * sizeof(float), 256);
  return c;
}
int build_cin4(const Cin4Plan& c, const float* x3, long long npix, const float* w3, int N, int taps, void* ws, hipStream_t st) {
  f32x4* x4 = reinterpret_cast<f32x4*>(ws);
  float* w4 = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + c.x4_bytes);
  hipLaunchKernelGGL(pad3to4_image_kernel, dim3((unsigned)std::min<long long>((npix + 255) / 256, 8192)), dim3(256), 0, st, x3, x4, npix);
  MUNIT_CHECK_LAUNCH("pad3to4_image");
  hipLaunchKernelGGL(pad3to4_weight_kernel, dim3((unsigned)cdiv((long long)N * c.kpad, 256)), dim3(256), 0, st, w3, w4, N, taps, c.kpad);
  MUNIT_CHECK_LAUNCH("pad3to4_weight");
  return MUNIT_OK;
}
bool cin4_fwd_ok(const munit_conv_desc* d) {
  return d->Cin == 3 && d->in_dtype == MUNIT_DTYPE_F32 && d->upsample == 0 && d->KH * d->KW <= 64 && d->Cout % 4 == 0 &&
         NWAVES == 8 && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_CIN4");
}

// 3x3 / stride 1 / pad 1 fp32 layers with wide channel counts: Winograd F(2x2, 3x3) (conv_wino.hip)
bool wino_geometry_ok(const munit_conv_desc* d) {
  return d->compute == MUNIT_COMPUTE_F32 && d->in_dtype == MUNIT_DTYPE_F32 && d->out_dtype == MUNIT_DTYPE_F32 &&
         d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->upsample == 0 && d->act != MUNIT_ACT_TANH;
}
bool wino_fwd_ok(const munit_conv_desc* d) { return wino_geometry_ok(d) && munit_wino_ok(d->B, d->H, d->W, d->Cin, d->Cout); }
// MUNIT_WINO_S2_MIN_BLOCKS: developer override of the threshold (tests use 1 to push small shapes through the kernel)
long long wino_s2_min_blocks() {
  static const long long v = getenv("MUNIT_WINO_S2_MIN_BLOCKS") ? atoll(getenv("MUNIT_WINO_S2_MIN_BLOCKS")) : 192;
  return v;
}
// Backward-data takes the Winograd form further down: its alternative is four phase launches of the implicit GEMM plus the
// fold kernel.  Measured per launch (tools/time_layers.py s2small, round 4): 144 blocks (discriminator 256->512 at 32x32, fake +
// real) 282 -> 166 us, 128 blocks (128->256 at 64x64) 153 -> 93, 124 blocks 77 -> 58; 80 blocks 138 -> 160 and 72 blocks
// 85 -> 90 the other way.
long long wino_s2_dgrad_min_blocks() {
  static const long long v = getenv("MUNIT_WINO_S2_MIN_BLOCKS") ? atoll(getenv("MUNIT_WINO_S2_MIN_BLOCKS")) : 100;
  return v;
}
// 4x4 / stride 2 / pad 1 fp32 layers (encoder down-sampling, discriminators): F(3x3, 2x2) over the four input phases
bool wino_s2_fwd_ok(const munit_conv_desc* d) {
  return d->compute == MUNIT_COMPUTE_F32 && d->in_dtype == MUNIT_DTYPE_F32 && d->out_dtype == MUNIT_DTYPE_F32 && d->KH == 4 &&
         d->KW == 4 && d->stride == 2 && d->pad == 1 && d->upsample == 0 && d->act != MUNIT_ACT_TANH && d->H >= 4 && d->W >= 4 &&
         munit_wino_ok(d->B, d->H, d->W, d->Cin, d->Cout) && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WINOGRAD_S2") &&
         // one block per CU and no split over K: only where the tile list fills most of the chip (the small layers of the
         // style encoder and the discriminators stay on the implicit-GEMM kernel and its split-K)
         (long long)cdiv((long long)d->B * cdiv(d->H / 2, 3) * cdiv(d->W / 2, 3), 64) * (d->Cout / 64) >= wino_s2_min_blocks();
}
// backward-data of those layers: dy has extent H/2 x W/2 and Cout channels (the contraction), dx Cin channels
bool wino_s2_dgrad_ok(const munit_conv_desc* d) {
  if (!(d->compute == MUNIT_COMPUTE_F32 && d->in_dtype == MUNIT_DTYPE_F32 && d->out_dtype == MUNIT_DTYPE_F32 && d->KH == 4 &&
        d->KW == 4 && d->stride == 2 && d->pad == 1 && d->upsample == 0 && d->H >= 4 && d->W >= 4 && d->H % 2 == 0 && d->W % 2 == 0 &&
        d->Cout % 8 == 0 && d->Cin % 64 == 0 && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WINOGRAD_S2") && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WINOGRAD")))
    return false;
  const int Hd = d->H / 2, Wd = d->W / 2;
  if (d->pad_mode == MUNIT_PAD_REFLECT && (Hd % 3 == 0 || Wd % 3 == 0)) return false;   // fold pairs must share a 3x3 tile
  if ((long long)d->B * d->H * d->W * std::max(d->Cin, d->Cout) >= (1ll << 29)) return false;
  return (long long)cdiv((long long)d->B * cdiv(Hd + 1, 3) * cdiv(Wd + 1, 3), 64) * (d->Cin / 64) * 4 >= wino_s2_dgrad_min_blocks();
}
// the four 3x3 phase convs of a sub-pixel up-sampling layer (over the SOURCE image) through the Winograd kernel
bool subpixel_wino_ok(const munit_conv_desc* d) {
  return subpixel_ok(d) && d->compute == MUNIT_COMPUTE_F32 && d->in_dtype == MUNIT_DTYPE_F32 && d->out_dtype == MUNIT_DTYPE_F32 &&
         d->act != MUNIT_ACT_TANH && munit_wino_ok(d->B, d->H, d->W, d->Cin, d->Cout);
}

// forward: which re-laid-out weight image the pass multiplies by (MUNIT_PREP_NONE: w as it is) and its size
munit_prep_item fwd_prep_item(const munit_conv_desc* d, const float* w, float* wp) {
  munit_prep_item it{w, wp, d->Cout, d->KH, d->KW, d->Cin, MUNIT_PREP_NONE, 1, 0};
  const bool small = munit_small_fwd_supported(d) && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_SMALL_FWD");
  if (small) return it;
  it.bf16 = d->in_dtype == MUNIT_DTYPE_BF16;
  if (subpixel_ok(d)) it.kind = subpixel_wino_ok(d) ? MUNIT_PREP_SUBPIXEL_WINOGRAD : MUNIT_PREP_SUBPIXEL;
  else if (wino_fwd_ok(d)) it.kind = MUNIT_PREP_WINOGRAD;
  else if (wino_s2_fwd_ok(d)) it.kind = MUNIT_PREP_WINOGRAD_S2;
  else if (it.bf16) it.kind = MUNIT_PREP_CAST;
  return it;
}
size_t prep_bytes(const munit_prep_item& it) {
  return it.kind == MUNIT_PREP_NONE ? 0 : align_up((size_t)prep_elems(it) * (it.bf16 ? 2 : 4), 256);
}
}  // namespace

extern "C" size_t munit_conv2d_fwd_workspace_bytes(const munit_conv_desc* d) {
  int Ho, Wo;
  if (munit_conv2d_out_hw(d, &Ho, &Wo)) return 0;
  // [weight image of the pass, when the caller keeps none][split-K slabs]
  const size_t img = prep_bytes(fwd_prep_item(d, nullptr, nullptr));
  if (munit_small_fwd_supported(d) && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_SMALL_FWD")) return munit_small_fwd_workspace(d);
  if (munit_small_fwd_supported(d) || d->in_dtype == MUNIT_DTYPE_BF16 || wino_fwd_ok(d) || wino_s2_fwd_ok(d)) return img;
  if (cin4_fwd_ok(d)) {   // [4-channel image][padded weights]
    const Cin4Plan c = plan_cin4((long long)d->B * d->H * d->W, d->Cout, d->KH * d->KW);
    return c.x4_bytes + c.w4_bytes;
  }
  if (subpixel_ok(d))   // split-K slabs of the frame launch (few tiles, 25-tap K)
    return img + std::max(splitk_bytes(d->B * (4 * Wo + 4 * (Ho - 4)), d->Cout, d->KH * d->KW * d->Cin, 1),     // 2-pixel frame
                          splitk_bytes(d->B * (2 * Wo + 2 * (Ho - 2)), d->Cout, d->KH * d->KW * d->Cin, 1));    // 1-pixel ring (F(2x2,3x3) phases)
  return img + splitk_bytes(d->B * Ho * Wo, d->Cout, d->KH * d->KW * d->Cin, 1);
}

extern "C" int munit_conv2d_fwd(const munit_conv_desc* d, const void* x, const float* w,
                                const float* bias, void* y, void* ws, size_t ws_bytes,
                                munit_stream_t stream) {
  return munit_conv2d_fwd_prepared(d, x, w, nullptr, bias, y, ws, ws_bytes, stream);
}

extern "C" int munit_conv2d_fwd_prepared(const munit_conv_desc* d, const void* x, const float* w, const void* wp,
                                         const float* bias, void* y, void* ws, size_t ws_bytes,
                                         munit_stream_t stream) {
  int Ho, Wo;
  int rc = munit_conv2d_out_hw(d, &Ho, &Wo);
  if (rc) return rc;
  MUNIT_CHECK_ARG(x && w && y, "conv2d_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (munit_small_fwd_supported(d) && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_SMALL_FWD")) {
    MUNIT_CHECK_ARG(d->out_dtype == MUNIT_DTYPE_F32, "conv2d_fwd: the 3-channel image head writes fp32");
    const size_t sneed = munit_small_fwd_workspace(d);
    if (sneed != 0 && (ws == nullptr || ws_bytes < sneed)) {
      munit_set_error("conv2d_fwd: workspace %zu < %zu", ws_bytes, sneed);
      return MUNIT_ERR_WORKSPACE;
    }
    return munit_small_fwd(d, Ho, Wo, x, w, bias, reinterpret_cast<float*>(y), ws, st);
  }
  const size_t need = munit_conv2d_fwd_workspace_bytes(d);
  if (need != 0 && (ws == nullptr || ws_bytes < need)) {
    munit_set_error("conv2d_fwd: workspace %zu < %zu", ws_bytes, need);
    return MUNIT_ERR_WORKSPACE;
  }
  // weight image of the pass: the caller's, or built into the head of the workspace
  munit_prep_item it = fwd_prep_item(d, w, reinterpret_cast<float*>(ws));
  const size_t img_bytes = prep_bytes(it);
  const float* wimg = w;
  if (it.kind != MUNIT_PREP_NONE) {
    wimg = reinterpret_cast<const float*>(wp);
    if (wimg == nullptr) {
      rc = launch_prep_one(it, st);
      if (rc) return rc;
      wimg = it.wp;
    }
  }
  if (it.kind == MUNIT_PREP_WINOGRAD_S2) {
    WinoParams q{};
    q.x = reinterpret_cast<const float*>(x); q.u = wimg; q.bias = bias; q.y = reinterpret_cast<float*>(y);
    q.y_sw = d->Cout; q.y_sh = (long long)Wo * d->Cout; q.y_sb = (long long)Ho * Wo * d->Cout;
    q.B = d->B; q.H = d->H; q.W = d->W; q.K = 4 * d->Cin; q.N = d->Cout; q.xc = d->Cin; q.cpp = d->Cin / 8;
    q.s2 = 1; q.Ho = Ho; q.Wo = Wo;
    q.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->Cin * 4);
    q.mode = d->pad_mode == MUNIT_PAD_REFLECT ? 0 : 1;
    q.th = cdiv(Ho, 3); q.tw = cdiv(Wo, 3); q.bth = cdiv(q.th, 8); q.btw = cdiv(q.tw, 8); q.NB = d->Cout / 64;
    q.act = d->act; q.slope = d->slope;
    return munit_wino_launch(q, st);
  }
  if (it.kind == MUNIT_PREP_WINOGRAD) {
    WinoParams q{};
    q.x = reinterpret_cast<const float*>(x); q.u = wimg; q.bias = bias; q.y = reinterpret_cast<float*>(y);
    q.y_sw = d->Cout; q.y_sh = (long long)d->W * d->Cout; q.y_sb = (long long)d->H * d->W * d->Cout;
    q.B = d->B; q.H = d->H; q.W = d->W; q.K = d->Cin; q.N = d->Cout; q.xc = d->Cin; q.cpp = d->Cin / 8;
    q.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->Cin * 4);
    q.mode = d->pad_mode == MUNIT_PAD_REFLECT ? 0 : 1;
    q.th = d->H / 2; q.tw = d->W / 2; q.bth = cdiv(q.th, 8); q.btw = cdiv(q.tw, 8); q.NB = d->Cout / 64;
    q.act = d->act; q.slope = d->slope;
    return munit_wino_launch(q, st);
  }
  void* slabs = ws ? reinterpret_cast<char*>(ws) + img_bytes : nullptr;
  const size_t slab_bytes = ws ? ws_bytes - img_bytes : 0;
  IgemmParams p{};
  p.x = reinterpret_cast<const float*>(x); p.w = wimg; p.bias = bias; p.y = y;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin;
  p.ups = d->upsample; p.Hu = d->H << p.ups; p.Wu = d->W << p.ups;
  p.Ho = Ho; p.Wo = Wo; p.Cout = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  p.reflect = d->pad_mode == MUNIT_PAD_REFLECT;
  p.Ktot = d->KH * d->KW * d->Cin;
  p.w_row = p.Ktot;
  p.y_sw = d->Cout;
  p.y_sh = (long long)Wo * d->Cout;
  p.y_sb = (long long)Ho * Wo * d->Cout;
  p.M = d->B * Ho * Wo;
  p.act = d->act; p.slope = d->slope;
  p.ct = d->compute;
  p.ps = 1;
  p.bf16s = d->in_dtype == MUNIT_DTYPE_BF16;
  p.out_bf16 = d->out_dtype == MUNIT_DTYPE_BF16;
  if (cin4_fwd_ok(d)) {
    const long long npix = (long long)d->B * d->H * d->W;
    const Cin4Plan c = plan_cin4(npix, d->Cout, d->KH * d->KW);
    rc = build_cin4(c, reinterpret_cast<const float*>(x), npix, w, d->Cout, d->KH * d->KW, ws, st);
    if (rc) return rc;
    p.x = reinterpret_cast<const float*>(ws);
    p.w = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ws) + c.x4_bytes);
    p.Ktot = c.kpad; p.w_row = c.kpad; p.cin4 = 1;
    return launch_igemm<0>(p, 1, st);
  }
  if (it.kind == MUNIT_PREP_SUBPIXEL || it.kind == MUNIT_PREP_SUBPIXEL_WINOGRAD) {
    // (1) four phase convs (3x3 over the source, merged weights) write every output pixel; the 2-pixel
    // frame, where reflect padding breaks the merge, is then (2) recomputed by the generic 25-tap gather.
    IgemmParams q = p;
    int ring = 2;                             // width of the frame the generic gather recomputes
    q.ups = 0; q.Hu = d->H; q.Wu = d->W;
    q.Ho = d->H; q.Wo = d->W;                 // one GEMM row per source pixel and phase
    q.KH = 3; q.KW = 3; q.pad = 1; q.reflect = 0;
    q.Ktot = 9 * d->Cin; q.w_row = q.Ktot;
    q.y_sw = 2 * d->Cout;
    q.y_sh = (long long)2 * Wo * d->Cout;
    q.M = d->B * d->H * d->W;
    q.ps = 2;
    q.w_phase = (long long)d->Cout * q.Ktot;
    q.y_phase_row = (long long)Wo * d->Cout;
    q.y_phase_col = d->Cout;
    if (it.kind == MUNIT_PREP_SUBPIXEL_WINOGRAD) {
      WinoParams wq{};
      wq.x = reinterpret_cast<const float*>(x); wq.u = wimg; wq.bias = bias; wq.y = reinterpret_cast<float*>(y);
      wq.y_sw = q.y_sw; wq.y_sh = q.y_sh; wq.y_sb = p.y_sb;
      wq.u_phase = wino_image_elems(d->Cin, d->Cout); wq.y_prow = q.y_phase_row; wq.y_pcol = q.y_phase_col; wq.phases = 4;
      wq.B = d->B; wq.H = d->H; wq.W = d->W; wq.K = d->Cin; wq.N = d->Cout; wq.xc = d->Cin; wq.cpp = d->Cin / 8;
      wq.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->Cin * 4);
      wq.mode = 1;
      // the phase convolutions REPLICATE the source edge instead of padding with zeros: with x[-1] := x[0] the merged 3x3
      // filters reproduce reflect(2) o nearest-upsample exactly for every output row / column but the outermost one (row 1
      // reads up[-1] = up[1] = x[0], which is what the replicated pixel holds; row 0 reads up[-2] = up[2] = x[1], which it is
      // not), so the 25-tap frame launch below recomputes a ring one pixel wide instead of two
      wq.edge = 1;
      ring = 1;
      wq.th = d->H / 2; wq.tw = d->W / 2; wq.bth = cdiv(wq.th, 8); wq.btw = cdiv(wq.tw, 8); wq.NB = d->Cout / 64;
      wq.act = d->act; wq.slope = d->slope;
      rc = munit_wino_launch(wq, st);
    } else {
      rc = launch_igemm<0>(q, 4, st);
    }
    if (rc) return rc;
    p.frame = ring;
    p.M = d->B * (2 * ring * Wo + 2 * ring * (Ho - 2 * ring));
    // the frame launch multiplies by the original 5x5 weights: fp32 -> w itself; bf16 storage -> their bf16 copy,
    // which the image carries behind the merged phase weights
    p.w = it.bf16 ? reinterpret_cast<const float*>(reinterpret_cast<const bf16_t*>(wimg) + (size_t)4 * 9 * d->Cout * d->Cin) : w;
    return launch_igemm<0>(p, 1, st, slabs, slab_bytes);
  }
  return launch_igemm<0>(p, 1, st, slabs, slab_bytes);
}

namespace {
// S[b][u][v][c] = dy[u][v] + dy[u+1][v] + dy[u][v+1] + dy[u+1][v+1] (terms past the last row / column are zero): the four
// up-sampled positions 2i+a, 2j+b of a source pixel are a 2x2 block of dy for every tap, so the backward-data of an
// up-sampling conv over the interior becomes a single-gather stride-2 correlation over S.
__global__ void box2x2_kernel(const f32x4* __restrict__ dy, f32x4* __restrict__ s, int B, int H, int W, int C4) {
  const long long total = (long long)B * H * W * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long long r = i / C4;
    const int v = (int)(r % W); r /= W;
    const int u = (int)(r % H);
    f32x4 a = dy[i];
    if (v + 1 < W) a += dy[i + C4];
    if (u + 1 < H) {
      a += dy[i + (long long)W * C4];
      if (v + 1 < W) a += dy[i + (long long)W * C4 + C4];
    }
    s[i] = a;
    (void)c;
  }
}

struct DgradPlan {
  int Ho, Wo, ps, TH, TW, Hq, Wq;
  bool direct;  // write dx directly (no pad / upsample / add): 1x1 convs, linear layers
  bool folded;  // stride-1, Cout % 32 == 0: pad/upsample adjoint folded into the gather (ROLE 2)
  bool small;   // 3 input channels, 7x7: padded-domain correlation on the thread-per-pixel VALU kernel
  bool boxsum;  // up-sampling 5x5 conv: interior through the 2x2 box sum of dy, 2-pixel frame through the folded gather
  bool bf16s;   // dy (and the weight image) are bf16 in HBM: bf16-storage kernels (direct-to-LDS forms only)
  bool patch;   // folded with at most two padded positions per axis: the LDS-patch form
  bool upwino;  // boxsum layer whose interior runs as one Winograd launch over the four dy phases (conv_wino.hip, KIND 3)
  bool wino_s2; // 4x4 stride-2 pad-1 fp32 layer: four F(3x3, 2x2) parity phases over dy (conv_wino.hip, KIND 2)
  bool wino;    // 3x3 stride-1 pad-1 fp32 layer: Winograd F(2x2, 3x3) with the border fold in the input patch (conv_wino.hip)
  bool cin4;    // three output channels (the image head): dy re-laid with a zero 4th channel, direct-to-LDS 4-channel taps
  size_t wt_bytes, g_bytes, sk_bytes, c4_bytes;
  size_t small_ws;   // small: workspace of the 3-output-channel forward kernel that computes the padded-domain correlation
};
// number of padded/up-sampled coordinates folding onto one source coordinate (host mirror of fold_cands)
int max_fold_cands(int H, int ups, int P, int reflect) {
  const int Hu = H << ups;
  int worst = 0;
  for (int i = 0; i < H; ++i) {
    int n = 0;
    for (int du = 0; du < (1 << ups); ++du) {
      const int hu = (i << ups) + du;
      ++n;
      if (reflect) {
        if (hu >= 1 && hu <= P) ++n;
        if (hu >= Hu - 1 - P && hu <= Hu - 2) ++n;
      }
    }
    worst = std::max(worst, n);
    if (i == 2 * P + 2 && H > 4 * P + 8) i = H - 2 * P - 4;  // interior rows are all alike
  }
  return worst;
}
int plan_dgrad(const munit_conv_desc* d, DgradPlan* pl) {
  int rc = munit_conv2d_out_hw(d, &pl->Ho, &pl->Wo);
  if (rc) return rc;
  MUNIT_CHECK_ARG(d->KH % d->stride == 0 && d->KW % d->stride == 0,
                  "conv2d_dgrad: kernel %dx%d not a multiple of stride %d", d->KH, d->KW, d->stride);
  pl->ps = d->stride;
  pl->TH = d->KH / d->stride;
  pl->TW = d->KW / d->stride;
  pl->Hq = d->stride * (pl->Ho + pl->TH - 1);
  pl->Wq = d->stride * (pl->Wo + pl->TW - 1);
  pl->direct = d->pad == 0 && d->upsample == 0 && pl->Hq == d->H && pl->Wq == d->W;
  const int reflect = d->pad_mode == MUNIT_PAD_REFLECT;
  pl->folded = !pl->direct && d->stride == 1 && d->Cout % 32 == 0 && d->KH == d->KW &&
               (d->H << d->upsample) + 2 * d->pad < 0xFFFF && (d->W << d->upsample) + 2 * d->pad < 0xFFFF &&
               max_fold_cands(d->H, d->upsample, d->pad, reflect) <= 4 &&
               max_fold_cands(d->W, d->upsample, d->pad, reflect) <= 4;
  {
    munit_conv_desc t{};
    t.B = d->B; t.H = pl->Ho; t.W = pl->Wo; t.Cin = d->Cout; t.Cout = d->Cin; t.KH = pl->TH; t.KW = pl->TW;
    t.stride = 1; t.pad = pl->TH - 1; t.pad_mode = MUNIT_PAD_ZERO;
    pl->small = !pl->direct && pl->ps == 1 && munit_small_fwd_supported(&t) && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_SMALL_DGRAD");
    if (pl->small) pl->folded = false;
    pl->small_ws = pl->small ? munit_small_fwd_workspace(&t) : 0;
    if (MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_FOLD")) pl->folded = false;
  }
  pl->patch = pl->folded && d->upsample == 0 && max_fold_cands(d->H, 0, d->pad, reflect) <= 2 &&
              max_fold_cands(d->W, 0, d->pad, reflect) <= 2;
  pl->wino_s2 = wino_s2_dgrad_ok(d);
  if (pl->wino_s2) {
    pl->direct = pl->folded = pl->small = pl->patch = pl->boxsum = pl->bf16s = pl->cin4 = pl->wino = pl->upwino = false;
    pl->wt_bytes = align_up((size_t)4 * wino_image_elems(d->Cout, d->Cin) * 4, 256);
    pl->g_bytes = 256;
    pl->sk_bytes = pl->c4_bytes = pl->small_ws = 0;
    return MUNIT_OK;
  }
  pl->wino = wino_geometry_ok(d) && munit_wino_ok(d->B, d->H, d->W, d->Cout, d->Cin);
  if (pl->wino) {
    pl->direct = pl->folded = pl->small = pl->patch = pl->boxsum = pl->bf16s = pl->cin4 = pl->upwino = pl->wino_s2 = false;
    pl->wt_bytes = align_up((size_t)wino_image_elems(d->Cout, d->Cin) * 4, 256);
    pl->g_bytes = 256;
    pl->sk_bytes = pl->c4_bytes = pl->small_ws = 0;
    return MUNIT_OK;
  }
  // bf16 storage: dy is bf16 (d->out_dtype), dx / the padded-domain buffer g take d->in_dtype.  Only the direct-to-LDS
  // forms exist in this mode: folded layers must be patchable, everything else runs as a plain correlation + fold_kernel
  // (the up-sampling convs then issue all 100 MACs per source pixel: cheap on the bf16 pipe, and no box sum).
  pl->bf16s = d->out_dtype == MUNIT_DTYPE_BF16 && !pl->small;
  if (pl->bf16s && !pl->patch) pl->folded = false;
  pl->boxsum = pl->folded && !pl->bf16s && d->in_dtype == MUNIT_DTYPE_F32 && d->upsample == 1 && d->KH == 5 && d->pad == 2 &&
               reflect && d->Cout % 32 == 0 && d->Cin % 4 == 0 && d->H >= 8 && d->W >= 8 && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_BOXSUM");
  pl->upwino = pl->boxsum && d->compute == MUNIT_COMPUTE_F32 && d->out_dtype == MUNIT_DTYPE_F32 && d->H % 2 == 0 && d->W % 2 == 0 &&
               d->Cout % 8 == 0 && d->Cin % 64 == 0 && (long long)d->B * 4 * d->H * d->W * d->Cout < (1ll << 29) &&
               !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WINOGRAD");
  const size_t esz = d->in_dtype == MUNIT_DTYPE_BF16 ? 2 : 4;   // element size of dx and g
  pl->wt_bytes = align_up((size_t)d->Cout * d->KH * d->KW * d->Cin * (pl->bf16s ? 2 : 4), 256);
  if (pl->upwino) pl->wt_bytes = align_up(((size_t)d->Cout * 25 * d->Cin + (size_t)4 * wino_image_elems(d->Cout, d->Cin)) * 4, 256);
  pl->g_bytes = align_up((size_t)d->B * pl->Hq * pl->Wq * d->Cin * esz, 256);
  if (pl->folded) pl->g_bytes = 256;  // no padded-domain buffer (an `add` operand falls back, see below)
  if (pl->boxsum) pl->g_bytes = align_up((size_t)d->B * pl->Ho * pl->Wo * d->Cout * sizeof(float), 256);  // S
  pl->cin4 = false;
  pl->c4_bytes = 0;
  if (pl->bf16s) {
    pl->sk_bytes = 0;
    return MUNIT_OK;
  }
  if (!pl->direct && !pl->folded && !pl->small && pl->ps == 1 && d->Cout == 3 && d->out_dtype == MUNIT_DTYPE_F32 &&
      pl->TH * pl->TW <= 64 && d->Cin % 4 == 0 && NWAVES == 8 && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_CIN4")) {
    const Cin4Plan c = plan_cin4((long long)d->B * pl->Ho * pl->Wo, d->Cin, pl->TH * pl->TW);
    pl->cin4 = true;
    pl->c4_bytes = c.x4_bytes + c.w4_bytes;
    pl->sk_bytes = 0;
    return MUNIT_OK;
  }
  if (pl->boxsum) {
    pl->sk_bytes = splitk_bytes(d->B * (4 * d->W + 4 * (d->H - 4)), d->Cin, d->KH * d->KW * d->Cout, 1);
    return MUNIT_OK;
  }
  pl->sk_bytes = pl->small ? pl->small_ws : pl->folded ? 0
                 : splitk_bytes(d->B * (pl->Ho + pl->TH - 1) * (pl->Wo + pl->TW - 1), d->Cin,
                                pl->TH * pl->TW * d->Cout, pl->ps * pl->ps);
  return MUNIT_OK;
}
}  // namespace

extern "C" size_t munit_conv2d_dgrad_workspace_bytes(const munit_conv_desc* d) {
  DgradPlan pl;
  if (plan_dgrad(d, &pl)) return 0;
  return pl.wt_bytes + pl.g_bytes + pl.sk_bytes + pl.c4_bytes;
}

extern "C" int munit_conv2d_dgrad(const munit_conv_desc* d, const void* dy, const float* w,
                                  const void* add, void* dx, void* ws, size_t ws_bytes,
                                  munit_stream_t stream) {
  return munit_conv2d_dgrad_prepared(d, dy, w, nullptr, add, dx, ws, ws_bytes, stream);
}

namespace {
template <typename T>
int launch_fold(const munit_conv_desc* d, const DgradPlan& pl, const void* g, const void* add, void* dx, hipStream_t st) {
  const int reflect = d->pad_mode == MUNIT_PAD_REFLECT;
  long long total = (long long)d->B * d->H * d->W * (d->Cin / 4);
  int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(fold_kernel<T>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const T*>(g),
                     reinterpret_cast<const T*>(add), reinterpret_cast<T*>(dx), d->B, d->H, d->W, d->Cin, d->upsample,
                     d->pad, reflect, pl.Hq, pl.Wq);
  MUNIT_CHECK_LAUNCH("fold");
  return MUNIT_OK;
}
}  // namespace

extern "C" int munit_conv2d_dgrad_prepared(const munit_conv_desc* d, const void* dy_, const float* w, const void* wp,
                                           const void* add_, void* dx_, void* ws, size_t ws_bytes,
                                           munit_stream_t stream) {
  DgradPlan pl;
  int rc = plan_dgrad(d, &pl);
  if (rc) return rc;
  MUNIT_CHECK_ARG(dy_ && (w || wp) && dx_ && ws, "conv2d_dgrad: null pointer");
  if (ws_bytes < pl.wt_bytes + pl.g_bytes + pl.sk_bytes + pl.c4_bytes) {
    munit_set_error("conv2d_dgrad: workspace %zu < %zu", ws_bytes, pl.wt_bytes + pl.g_bytes + pl.sk_bytes + pl.c4_bytes);
    return MUNIT_ERR_WORKSPACE;
  }
  // element types: dy = d->out_dtype, dx / add / g = d->in_dtype.  The fp32 names below keep the fp32 code readable; in
  // the bf16 cases the pointers are only handed on (the kernels reinterpret them).
  const float* dy = reinterpret_cast<const float*>(dy_);
  const float* add = reinterpret_cast<const float*>(add_);
  float* dx = reinterpret_cast<float*>(dx_);
  const bool dx_bf16 = d->in_dtype == MUNIT_DTYPE_BF16;
  MUNIT_CHECK_ARG(!dx_bf16 || d->Cin % 4 == 0, "conv2d_dgrad: bf16 dx needs Cin %% 4 == 0");
  hipStream_t st = (hipStream_t)stream;
  const float* wt = reinterpret_cast<const float*>(wp);
  float* g = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + pl.wt_bytes);
  const bool direct = pl.direct && add == nullptr;
  if (wt == nullptr) {   // no prepared image from the caller: re-lay the weights into the workspace
    PrepItem it{w, reinterpret_cast<float*>(ws), d->Cout, d->KH, d->KW, d->Cin,
                pl.wino_s2 ? MUNIT_PREP_WINOGRAD_S2_DGRAD : pl.wino ? MUNIT_PREP_WINOGRAD_DGRAD
                : pl.upwino ? MUNIT_PREP_SUBPIXEL_WINOGRAD_DGRAD : MUNIT_PREP_DGRAD, pl.ps, pl.bf16s ? 1 : 0};
    rc = launch_prep_one(it, st);
    if (rc) return rc;
    wt = it.wp;
  }
  if (pl.wino_s2) {
    WinoParams q{};
    const int Hd = d->H / 2, Wd = d->W / 2;
    q.x = dy; q.u = wt; q.bias = nullptr; q.y = dx;
    q.y_sw = d->Cin; q.y_sh = (long long)d->W * d->Cin; q.y_sb = (long long)d->H * d->W * d->Cin;
    q.B = d->B; q.H = Hd; q.W = Wd; q.K = d->Cout; q.N = d->Cin; q.xc = d->Cout; q.cpp = d->Cout / 8;
    q.s2 = 2; q.Ho = d->H; q.Wo = d->W;
    q.x_bytes = (unsigned)((size_t)d->B * Hd * Wd * d->Cout * 4);
    q.mode = d->pad_mode == MUNIT_PAD_REFLECT ? 0 : 1;
    q.th = cdiv(Hd + 1, 3); q.tw = cdiv(Wd + 1, 3); q.bth = cdiv(q.th, 8); q.btw = cdiv(q.tw, 8); q.NB = d->Cin / 64;
    q.u_phase = wino_image_elems(d->Cout, d->Cin); q.phases = 4;
    q.act = MUNIT_ACT_NONE; q.slope = 0.f;
    rc = munit_wino_launch(q, st);
    if (rc) return rc;
    if (add != nullptr) {
      long long n = (long long)d->B * d->H * d->W * d->Cin;
      int blocks = (int)std::min<long long>((n + 255) / 256, 8192);
      hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks), dim3(256), 0, st, dx, add, n);
      MUNIT_CHECK_LAUNCH("add_inplace");
    }
    return MUNIT_OK;
  }
  if (pl.wino) {
    WinoParams q{};
    q.x = dy; q.u = wt; q.bias = nullptr; q.y = dx;
    q.y_sw = d->Cin; q.y_sh = (long long)d->W * d->Cin; q.y_sb = (long long)d->H * d->W * d->Cin;
    q.B = d->B; q.H = d->H; q.W = d->W; q.K = d->Cout; q.N = d->Cin; q.xc = d->Cout; q.cpp = d->Cout / 8;
    q.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->Cout * 4);
    q.mode = d->pad_mode == MUNIT_PAD_REFLECT ? 2 : 1;
    q.th = d->H / 2; q.tw = d->W / 2; q.bth = cdiv(q.th, 8); q.btw = cdiv(q.tw, 8); q.NB = d->Cin / 64;
    q.act = MUNIT_ACT_NONE; q.slope = 0.f;
    q.add = add;    // added in the kernel's epilogue (dx and add share the layout)
    return munit_wino_launch(q, st);
  }
  {
    // data gradient of a 7x7 conv with 3 input channels (first encoder layers): the padded-domain
    // correlation has N = 3 -> thread-per-pixel VALU kernel instead of a 95 %-padded MFMA tile
    munit_conv_desc t{};
    t.B = d->B; t.H = pl.Ho; t.W = pl.Wo; t.Cin = d->Cout; t.Cout = d->Cin; t.KH = pl.TH; t.KW = pl.TW;
    t.stride = 1; t.pad = pl.TH - 1; t.pad_mode = MUNIT_PAD_ZERO; t.upsample = 0; t.act = MUNIT_ACT_NONE;
    t.in_dtype = d->out_dtype; t.out_dtype = MUNIT_DTYPE_F32;
    if (pl.small) {
      MUNIT_CHECK_ARG(!dx_bf16, "conv2d_dgrad: the 3-channel data gradient is fp32");
      rc = munit_small_fwd(&t, pl.Hq, pl.Wq, dy_, wt, nullptr, g, reinterpret_cast<char*>(ws) + pl.wt_bytes + pl.g_bytes, st);
      if (rc) return rc;
      const int reflect = d->pad_mode == MUNIT_PAD_REFLECT;
      long long total = (long long)d->B * d->H * d->W * d->Cin;
      int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
      hipLaunchKernelGGL(fold_scalar_kernel, dim3(blocks), dim3(256), 0, st, g, add, dx, d->B, d->H, d->W,
                         d->Cin, d->upsample, d->pad, reflect, pl.Hq, pl.Wq);
      MUNIT_CHECK_LAUNCH("fold");
      return MUNIT_OK;
    }
  }
  if (pl.folded) {
    IgemmParams p{};
    p.x = dy; p.w = wt; p.bias = nullptr; p.y = dx;
    p.B = d->B; p.H = pl.Ho; p.W = pl.Wo; p.Cin = d->Cout;   // GEMM "input" = dy
    p.ups = 0; p.Hu = pl.Ho; p.Wu = pl.Wo;
    p.Ho = d->H; p.Wo = d->W; p.Cout = d->Cin;               // GEMM rows = source pixels of dx
    p.KH = d->KH; p.KW = d->KW; p.stride = 1; p.pad = d->KH - 1; p.reflect = 0;
    p.Ktot = d->KH * d->KW * d->Cout;
    p.w_row = p.Ktot;
    p.y_sw = d->Cin;
    p.y_sh = (long long)d->W * d->Cin;
    p.y_sb = (long long)d->H * d->W * d->Cin;
    p.M = d->B * d->H * d->W;
    p.act = MUNIT_ACT_NONE; p.slope = 0.f;
    p.ct = d->compute;
    p.ps = 1;
    p.f_pad = d->pad; p.f_ups = d->upsample; p.f_reflect = d->pad_mode == MUNIT_PAD_REFLECT;
    p.f_Hu = d->H << d->upsample; p.f_Wu = d->W << d->upsample;
    p.patch = pl.patch;
    p.bf16s = pl.bf16s; p.out_bf16 = dx_bf16;
    MUNIT_CHECK_ARG(!dx_bf16 || pl.bf16s || pl.patch, "conv2d_dgrad: unsupported dtype combination");
    if (pl.boxsum && pl.upwino) {
      // interior source pixels 2..H-3 x 2..W-3 as ONE Winograd launch: the four output phases of dy are 3x3-correlated with the
      // rotated merged filters and summed (K = 4 Cout); no box sum, no padding
      WinoParams wq{};
      wq.x = dy; wq.u = wt + (size_t)d->Cout * 25 * d->Cin; wq.bias = nullptr;
      wq.y = dx + ((long long)2 * d->W + 2) * d->Cin;
      wq.y_sw = d->Cin; wq.y_sh = (long long)d->W * d->Cin; wq.y_sb = (long long)d->H * d->W * d->Cin;
      wq.B = d->B; wq.H = pl.Ho; wq.W = pl.Wo; wq.K = 4 * d->Cout; wq.N = d->Cin; wq.xc = d->Cout; wq.cpp = d->Cout / 8;
      wq.s2 = 3;
      wq.x_bytes = (unsigned)((size_t)d->B * pl.Ho * pl.Wo * d->Cout * 4);
      wq.mode = 1;
      wq.th = (d->H - 4) / 2; wq.tw = (d->W - 4) / 2; wq.bth = cdiv(wq.th, 8); wq.btw = cdiv(wq.tw, 8); wq.NB = d->Cin / 64;
      wq.act = MUNIT_ACT_NONE; wq.slope = 0.f;
      rc = munit_wino_launch(wq, st);
      if (rc) return rc;
      p.frame = 2;
      p.M = d->B * (4 * d->W + 4 * (d->H - 4));
      rc = launch_igemm<2>(p, 1, st, reinterpret_cast<char*>(ws) + pl.wt_bytes + pl.g_bytes, pl.sk_bytes);
    } else if (pl.boxsum) {
      // interior source pixels 2..H-3 x 2..W-3: dx[i][j] = sum_taps wt[t][r] . S[2i-2+t][2j-2+r] -- one gather per element
      {
        const long long total = (long long)d->B * pl.Ho * pl.Wo * (d->Cout / 4);
        hipLaunchKernelGGL(box2x2_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 16384)), dim3(256), 0, st,
                           reinterpret_cast<const f32x4*>(dy), reinterpret_cast<f32x4*>(g), d->B, pl.Ho, pl.Wo, d->Cout / 4);
        MUNIT_CHECK_LAUNCH("box2x2");
      }
      IgemmParams q = p;
      q.x = g;
      q.y = dx + ((long long)2 * d->W + 2) * d->Cin;
      q.Ho = d->H - 4; q.Wo = d->W - 4;
      q.stride = 2; q.pad = -2;
      q.M = d->B * q.Ho * q.Wo;
      rc = launch_igemm<0>(q, 1, st);
      if (rc) return rc;
      // the 2-pixel frame keeps the general folded gather (reflections add further positions there), split over K
      p.frame = 2;
      p.M = d->B * (4 * d->W + 4 * (d->H - 4));
      rc = launch_igemm<2>(p, 1, st, reinterpret_cast<char*>(ws) + pl.wt_bytes + pl.g_bytes, pl.sk_bytes);
    } else {
      rc = launch_igemm<2>(p, 1, st);
    }
    if (rc) return rc;
    if (add != nullptr) {
      MUNIT_CHECK_ARG(!dx_bf16, "conv2d_dgrad: `add` with a bf16 dx is not supported on the folded path");
      long long n = (long long)d->B * d->H * d->W * d->Cin;
      int blocks = (int)std::min<long long>((n + 255) / 256, 8192);
      hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks), dim3(256), 0, st, dx, add, n);
      MUNIT_CHECK_LAUNCH("add_inplace");
    }
    return MUNIT_OK;
  }
  IgemmParams p{};
  p.x = dy; p.w = wt; p.bias = nullptr; p.y = direct ? (void*)dx : (void*)g;
  p.B = d->B; p.H = pl.Ho; p.W = pl.Wo; p.Cin = d->Cout;
  p.ups = 0; p.Hu = pl.Ho; p.Wu = pl.Wo;
  p.Ho = pl.Ho + pl.TH - 1; p.Wo = pl.Wo + pl.TW - 1; p.Cout = d->Cin;
  p.KH = pl.TH; p.KW = pl.TW; p.stride = 1; p.pad = pl.TH - 1;  // TH == TW for every layer here
  MUNIT_CHECK_ARG(pl.TH == pl.TW, "conv2d_dgrad: non-square kernels are not supported");
  p.reflect = 0;
  p.Ktot = pl.TH * pl.TW * d->Cout;
  p.w_row = p.Ktot;
  p.y_sw = pl.ps * d->Cin;
  p.y_sh = (long long)pl.ps * pl.Wq * d->Cin;
  p.y_sb = (long long)pl.Hq * pl.Wq * d->Cin;
  p.M = d->B * p.Ho * p.Wo;
  p.act = MUNIT_ACT_NONE; p.slope = 0.f;
  p.ct = d->compute;
  p.ps = pl.ps;
  p.w_phase = (long long)d->Cin * p.Ktot;
  p.y_phase_row = (long long)pl.Wq * d->Cin;
  p.y_phase_col = d->Cin;
  p.bf16s = pl.bf16s; p.out_bf16 = dx_bf16;
  if (pl.cin4) {
    // image head: dy has three channels -> 4-channel re-layout of dy and of the transposed weights, direct-to-LDS taps
    char* c4 = reinterpret_cast<char*>(ws) + pl.wt_bytes + pl.g_bytes;
    const long long npix = (long long)d->B * pl.Ho * pl.Wo;
    const Cin4Plan c = plan_cin4(npix, d->Cin, pl.TH * pl.TW);
    rc = build_cin4(c, dy, npix, wt, d->Cin, pl.TH * pl.TW, c4, st);
    if (rc) return rc;
    p.x = reinterpret_cast<const float*>(c4);
    p.w = reinterpret_cast<const float*>(c4 + c.x4_bytes);
    p.Ktot = c.kpad; p.w_row = c.kpad; p.cin4 = 1;
    rc = launch_igemm<1>(p, 1, st);
  } else {
    rc = launch_igemm<1>(p, pl.ps * pl.ps, st, reinterpret_cast<char*>(ws) + pl.wt_bytes + pl.g_bytes, pl.sk_bytes);
  }
  if (rc) return rc;
  if (!direct) {
    const int reflect = d->pad_mode == MUNIT_PAD_REFLECT;
    if (d->Cin % 4 == 0) {
      rc = dx_bf16 ? launch_fold<bf16_t>(d, pl, g, add_, dx_, st) : launch_fold<float>(d, pl, g, add_, dx_, st);
      if (rc) return rc;
    } else {
      long long total = (long long)d->B * d->H * d->W * d->Cin;
      int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
      hipLaunchKernelGGL(fold_scalar_kernel, dim3(blocks), dim3(256), 0, st, g, add, dx, d->B, d->H, d->W,
                         d->Cin, d->upsample, d->pad, reflect, pl.Hq, pl.Wq);
      MUNIT_CHECK_LAUNCH("fold");
    }
  }
  return MUNIT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Prepared weight images (see prep_weights_kernel)
// ---------------------------------------------------------------------------------------------------------
extern "C" int munit_conv2d_prep_item(const munit_conv_desc* d, int pass, const float* w, float* wp, munit_prep_item* out) {
  int Ho, Wo;
  int rc = munit_conv2d_out_hw(d, &Ho, &Wo);
  if (rc) return rc;
  MUNIT_CHECK_ARG(out != nullptr, "conv2d_prep_item: null item");
  MUNIT_CHECK_ARG(pass == MUNIT_PASS_FWD || pass == MUNIT_PASS_DGRAD, "conv2d_prep_item: pass must be MUNIT_PASS_FWD or _DGRAD");
  munit_prep_item it{w, wp, d->Cout, d->KH, d->KW, d->Cin, MUNIT_PREP_NONE, 1, 0};
  if (pass == MUNIT_PASS_FWD) {
    it = fwd_prep_item(d, w, wp);
  } else {
    DgradPlan pl;
    rc = plan_dgrad(d, &pl);
    if (rc) return rc;
    it.kind = pl.wino_s2 ? MUNIT_PREP_WINOGRAD_S2_DGRAD : pl.wino ? MUNIT_PREP_WINOGRAD_DGRAD
              : pl.upwino ? MUNIT_PREP_SUBPIXEL_WINOGRAD_DGRAD : MUNIT_PREP_DGRAD;
    it.ps = pl.ps;
    it.bf16 = pl.bf16s ? 1 : 0;
  }
  *out = it;
  return MUNIT_OK;
}

extern "C" size_t munit_conv2d_prepared_weight_bytes(const munit_conv_desc* d, int pass) {
  munit_prep_item it;
  if (munit_conv2d_prep_item(d, pass, nullptr, nullptr, &it) || it.kind == MUNIT_PREP_NONE) return 0;
  return (size_t)prep_elems(it) * (it.bf16 ? 2 : 4);
}

extern "C" int munit_conv2d_prepare_weights(const munit_prep_item* item, munit_stream_t stream) {
  MUNIT_CHECK_ARG(item && item->w && item->wp, "conv2d_prepare_weights: null pointer");
  const bool wino = item->kind == MUNIT_PREP_WINOGRAD || item->kind == MUNIT_PREP_WINOGRAD_DGRAD;
  const bool spw = item->kind == MUNIT_PREP_SUBPIXEL_WINOGRAD || item->kind == MUNIT_PREP_SUBPIXEL_WINOGRAD_DGRAD;
  const bool ws2 = item->kind == MUNIT_PREP_WINOGRAD_S2 || item->kind == MUNIT_PREP_WINOGRAD_S2_DGRAD;
  MUNIT_CHECK_ARG(!ws2 || (item->KH == 4 && item->KW == 4 && !item->bf16 &&
                           (item->kind == MUNIT_PREP_WINOGRAD_S2 ? item->Cin % 8 == 0 && item->Cout % 64 == 0
                                                                 : item->Cout % 8 == 0 && item->Cin % 64 == 0)),
                  "conv2d_prepare_weights: stride-2 Winograd image needs a 4x4 fp32 filter, K %% 8 == 0, N %% 64 == 0");
  MUNIT_CHECK_ARG(!spw || (item->KH == 5 && item->KW == 5 && !item->bf16 &&
                           (item->kind != MUNIT_PREP_SUBPIXEL_WINOGRAD_DGRAD ? item->Cin % 8 == 0 && item->Cout % 64 == 0
                                                                       : item->Cout % 8 == 0 && item->Cin % 64 == 0)),
                  "conv2d_prepare_weights: sub-pixel Winograd image needs a 5x5 fp32 filter, Cin %% 8 == 0, Cout %% 64 == 0");
  MUNIT_CHECK_ARG(item->kind == MUNIT_PREP_DGRAD || item->kind == MUNIT_PREP_SUBPIXEL || wino || spw || ws2 || (item->kind == MUNIT_PREP_CAST && item->bf16),
                  "conv2d_prepare_weights: bad kind %d", item->kind);
  MUNIT_CHECK_ARG(!wino || (item->KH == 3 && item->KW == 3 && !item->bf16 &&
                            (item->kind != MUNIT_PREP_WINOGRAD_DGRAD ? item->Cin % 8 == 0 && item->Cout % 64 == 0
                                                               : item->Cout % 8 == 0 && item->Cin % 64 == 0)),
                  "conv2d_prepare_weights: Winograd image needs a 3x3 fp32 filter, K %% 8 == 0, N %% 64 == 0");
  MUNIT_CHECK_ARG(item->ps >= 1 && item->KH % item->ps == 0 && item->KW % item->ps == 0, "conv2d_prepare_weights: bad phase count");
  MUNIT_CHECK_ARG(item->kind != MUNIT_PREP_SUBPIXEL || (item->KH == 5 && item->KW == 5), "conv2d_prepare_weights: sub-pixel needs 5x5");
  return launch_prep_one(*item, (hipStream_t)stream);
}

extern "C" int munit_conv2d_prepare_weights_batch(const munit_prep_item* items_dev, int n, munit_stream_t stream) {
  MUNIT_CHECK_ARG(items_dev != nullptr && n > 0 && n <= 65535, "conv2d_prepare_weights_batch: bad table (n=%d)", n);
  hipLaunchKernelGGL(prep_weights_kernel<true>, dim3(64, (unsigned)n), dim3(256), 0, (hipStream_t)stream, items_dev, PrepItem{});
  MUNIT_CHECK_LAUNCH("prep_weights_batch");
  return MUNIT_OK;
}

// Multiply-accumulates the kernels actually issue for one call (x2 = FLOPs), as opposed to the algorithmic
// 2*B*Ho*Wo*Cout*KH*KW*Cin: the sub-pixel forward runs 4 merged 3x3 phases + the 25-tap frame, the box-sum
// backward-data one 25-tap row per interior SOURCE pixel + the frame, strided backward-data its phases over the
// padded domain.  Valid GEMM rows only (tile padding is not counted).  bench.py reports both totals.
// Name (as a profiler shows it) of the kernel that carries a pass of this layer; mirrors the dispatch of the entry points.
const char* munit_igemm_kernel_name(const munit_conv_desc* d, int pass) {
  int Ho, Wo;
  if (munit_conv2d_out_hw(d, &Ho, &Wo)) return "invalid";
  const bool refl = d->pad_mode == MUNIT_PAD_REFLECT;
  if (pass == MUNIT_PASS_FWD) {
    if (munit_small_fwd_supported(d) && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_SMALL_FWD")) return "conv_head_pk_kernel";
    if (subpixel_wino_ok(d)) return "conv_wino_kernel<1, 0> x4 sub-pixel phases + conv_igemm_kernel frame";
    if (subpixel_ok(d)) return "conv_igemm_kernel x4 sub-pixel phases + frame";
    if (wino_fwd_ok(d)) return refl ? "conv_wino_kernel<0, 0>" : "conv_wino_kernel<1, 0>";
    if (wino_s2_fwd_ok(d)) return refl ? "conv_wino_kernel<0, 1>" : "conv_wino_kernel<1, 1>";
    if (cin4_fwd_ok(d)) return "conv_igemm_kernel<.., 5> (3 input channels as 4-channel taps)";
    return "conv_igemm_kernel<fwd>";
  }
  DgradPlan pl;
  if (plan_dgrad(d, &pl)) return "invalid";
  if (pl.wino) return refl ? "conv_wino_kernel<2, 0>" : "conv_wino_kernel<1, 0>";
  if (pl.wino_s2) return refl ? "conv_wino_kernel<0, 2>" : "conv_wino_kernel<1, 2>";
  if (pl.boxsum && pl.upwino) return "conv_wino_kernel<1, 3> + conv_igemm_kernel frame";
  if (pl.boxsum) return "box2x2_kernel + conv_igemm_kernel (box-sum backward-data)";
  if (pl.small) return "conv_head_pk_kernel (padded-domain correlation)";
  if (pl.patch) return "conv_igemm_kernel<.., 2, 3> (LDS-patch fold)";
  if (pl.folded) return "conv_igemm_kernel<.., 2, 0> (folded gather)";
  if (pl.direct) return "conv_igemm_kernel<dgrad direct>";
  if (pl.cin4) return "conv_igemm_kernel<.., 1, 5> (3 output channels as 4-channel taps)";
  return "conv_igemm_kernel<.., 1, .> phases + fold_kernel";
}

double munit_igemm_executed_flops(const munit_conv_desc* d, int pass) {
  int Ho, Wo;
  if (munit_conv2d_out_hw(d, &Ho, &Wo)) return 0.0;
  const double cc = 2.0 * d->Cin * d->Cout;
  if (pass == MUNIT_PASS_FWD) {
    if (subpixel_wino_ok(d)) return cc * d->B * ((double)(d->H / 2) * (d->W / 2) * 4 * 16 + (2.0 * Wo + 2.0 * (Ho - 2)) * 25);
    if (subpixel_ok(d)) return cc * d->B * ((double)d->H * d->W * 4 * 9 + (4.0 * Wo + 4.0 * (Ho - 4)) * 25);
    if (wino_fwd_ok(d)) return cc * d->B * (d->H / 2) * (d->W / 2) * 16;   // 16 products per 2x2 tile instead of 36
    if (wino_s2_fwd_ok(d)) return 4 * cc * d->B * cdiv(Ho, 3) * cdiv(Wo, 3) * 16;   // per 3x3 tile and input phase
    if (cin4_fwd_ok(d) && !munit_small_fwd_supported(d))   // zero 4th input channel, K padded to the 32-wide tile
      return 2.0 * d->Cout * d->B * Ho * Wo * plan_cin4(1, 1, d->KH * d->KW).kpad;
    return cc * d->B * Ho * Wo * d->KH * d->KW;
  }
  DgradPlan pl;
  if (plan_dgrad(d, &pl)) return 0.0;
  if (pl.wino) return cc * d->B * (d->H / 2) * (d->W / 2) * 16;
  if (pl.wino_s2) return 4 * cc * d->B * cdiv(d->H / 2 + 1, 3) * cdiv(d->W / 2 + 1, 3) * 16;
  if (pl.boxsum && pl.upwino) return cc * d->B * ((double)((d->H - 4) / 2) * ((d->W - 4) / 2) * 4 * 16 + (4.0 * d->W + 4.0 * (d->H - 4)) * 25);
  if (pl.boxsum) return cc * d->B * ((double)(d->H - 4) * (d->W - 4) + 4.0 * d->W + 4.0 * (d->H - 4)) * d->KH * d->KW;
  if (pl.folded) return cc * d->B * d->H * d->W * d->KH * d->KW;
  if (pl.direct) return cc * d->B * Ho * Wo * d->KH * d->KW;
  if (pl.cin4) return 2.0 * d->Cin * d->B * (double)(pl.Ho + pl.TH - 1) * (pl.Wo + pl.TW - 1) * plan_cin4(1, 1, pl.TH * pl.TW).kpad;
  // phase launches over the padded domain (also the 3-channel first layer through the thread-per-pixel kernel)
  return cc * d->B * (double)(pl.Ho + pl.TH - 1) * (pl.Wo + pl.TW - 1) * pl.TH * pl.TW * pl.ps * pl.ps;
}

// ---------------------------------------------------------------------------------------------------------
// nn.Linear (LinearBlock, scripts/networks.py:712, 743-749) as named entry points: y[B][N] = act(x[B][K] w[N][K]^T + b).
// They are the 1x1 convolution on a [B][1][1][K] image -- same kernels, same workspaces.
// ---------------------------------------------------------------------------------------------------------
namespace {
munit_conv_desc linear_desc(int B, int K, int N, int act, float slope, int compute) {
  munit_conv_desc d{};
  d.B = B; d.H = 1; d.W = 1; d.Cin = K; d.Cout = N; d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
  d.pad_mode = MUNIT_PAD_ZERO; d.upsample = 0; d.act = act; d.slope = slope; d.compute = compute;
  return d;
}
}  // namespace

extern "C" size_t munit_linear_workspace_bytes(int B, int K, int N) {
  const munit_conv_desc d = linear_desc(B, K, N, MUNIT_ACT_NONE, 0.f, MUNIT_COMPUTE_F32);
  return std::max(std::max(munit_conv2d_fwd_workspace_bytes(&d), munit_conv2d_dgrad_workspace_bytes(&d)),
                  munit_conv2d_wgrad_workspace_bytes(&d));
}

extern "C" int munit_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, int act,
                                float slope, void* ws, size_t ws_bytes, munit_stream_t stream) {
  const munit_conv_desc d = linear_desc(B, K, N, act, slope, MUNIT_COMPUTE_F32);
  return munit_conv2d_fwd(&d, x, w, bias, y, ws, ws_bytes, stream);
}

extern "C" int munit_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B,
                                int K, int N, float beta, void* ws, size_t ws_bytes, munit_stream_t stream) {
  const munit_conv_desc d = linear_desc(B, K, N, MUNIT_ACT_NONE, 0.f, MUNIT_COMPUTE_F32);
  if (dx != nullptr) {
    const int rc = munit_conv2d_dgrad(&d, dy, w, nullptr, dx, ws, ws_bytes, stream);
    if (rc) return rc;
  }
  if (dw != nullptr) return munit_conv2d_wgrad(&d, x, dy, dw, db, beta, ws, ws_bytes, stream);
  return MUNIT_OK;
}

